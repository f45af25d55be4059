// kernels_gemm.hip -- the SRP-PHAT steering scan as a dense contraction on the matrix cores.
//
//   C[f][d] = sum_{g,k} ( Re Ghat_g[f][k] cos(phi_{g,d,k}) - Im Ghat_g[f][k] sin(phi_{g,d,k}) )
//           = A[f][:] . B[:][d]          A: [frames][Kp]  (k_stft_phat),  B: steering table (host)
//
// which is sum_p R_p[d] of SteeringBeamforming::computeCorrelations (SteeringBeamforming.cpp:104-130)
// with pairs of identical delay tables pre-summed.  D = 361 angles x depth 7182 (8-mic ULA) makes
// this >95 % of the path's flops and compute-bound, so it is the one stage on MFMA.
//
//   k_srp_gemm_f32   v_mfma_f32_32x32x2_f32: bit-exact fp32 fma chain, parity anchor
//   k_srp_gemm_f16   v_mfma_f32_32x32x16_f16 with 1 (fp16) or 3 (fp16x3 hi/lo split) products
#include "mca_internal.h"
#include "cand_unit.h"

namespace mca {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// ---------------------------------------------------------------------------------------
// fp32: block tile 128 frames x 192 angles, 4 waves as 2 x 2, wave tile 64 x 96 = 2 x 3 MFMA
// tiles (96 accumulator VGPRs).  BK = 16; global -> registers -> LDS with the next tile's
// loads in flight during the MFMAs.  A is stored transposed in LDS ([k][row]) so that both
// operand reads are conflict-free ds_read_b32.
// ---------------------------------------------------------------------------------------
constexpr int G32_BM = 128, G32_BN = 192, G32_BK = 16;

__global__ __launch_bounds__(256) void k_srp_gemm_f32(GemmArgs p)
{
    __shared__ float As[G32_BK][G32_BM + 4];
    __shared__ float Bs[G32_BK][G32_BN + 4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int row0 = blockIdx.x * G32_BM, col0 = blockIdx.y * G32_BN;
    const float *A = reinterpret_cast<const float *>(p.A);
    const float *B = reinterpret_cast<const float *>(p.B);

    f32x16 acc[2][3];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int ar = tid >> 2, akq = tid & 3;
    const int arow_a = min(row0 + ar, p.rows - 1), arow_b = min(row0 + ar + 64, p.rows - 1);
    const float *pa0 = A + (long long)arow_a * p.a_row_elems + akq * 4;
    const float *pa1 = A + (long long)arow_b * p.a_row_elems + akq * 4;
    const int bkr[3] = {tid / 48, (tid + 256) / 48, (tid + 512) / 48};
    const int bc4[3] = {tid % 48, (tid + 256) % 48, (tid + 512) % 48};

    float4 ra0, ra1, rb0, rb1, rb2;
    const float *pb0 = B + (long long)bkr[0] * p.Dp + col0 + bc4[0] * 4;
    const float *pb1 = B + (long long)bkr[1] * p.Dp + col0 + bc4[1] * 4;
    const float *pb2 = B + (long long)bkr[2] * p.Dp + col0 + bc4[2] * 4;
#define G32_GLOAD(k0)                                                        \
    do {                                                                     \
        ra0 = *reinterpret_cast<const float4 *>(pa0 + (k0));                 \
        ra1 = *reinterpret_cast<const float4 *>(pa1 + (k0));                 \
        rb0 = *reinterpret_cast<const float4 *>(pb0 + (long long)(k0) * p.Dp); \
        rb1 = *reinterpret_cast<const float4 *>(pb1 + (long long)(k0) * p.Dp); \
        rb2 = *reinterpret_cast<const float4 *>(pb2 + (long long)(k0) * p.Dp); \
    } while (0)
    // K range of this workgroup (split-K over blockIdx.z: partial map z goes to C + z * c_plane_elems)
    const int nk_all = p.Kp / G32_BK, per = (nk_all + gridDim.z - 1) / gridDim.z;
    const int kt0 = blockIdx.z * per, nk = min(kt0 + per, nk_all);
    G32_GLOAD(kt0 * G32_BK);
    for (int kt = kt0; kt < nk; ++kt) {
        As[akq * 4 + 0][ar] = ra0.x; As[akq * 4 + 1][ar] = ra0.y; As[akq * 4 + 2][ar] = ra0.z; As[akq * 4 + 3][ar] = ra0.w;
        As[akq * 4 + 0][ar + 64] = ra1.x; As[akq * 4 + 1][ar + 64] = ra1.y; As[akq * 4 + 2][ar + 64] = ra1.z; As[akq * 4 + 3][ar + 64] = ra1.w;
        *reinterpret_cast<float4 *>(&Bs[bkr[0]][bc4[0] * 4]) = rb0;
        *reinterpret_cast<float4 *>(&Bs[bkr[1]][bc4[1] * 4]) = rb1;
        *reinterpret_cast<float4 *>(&Bs[bkr[2]][bc4[2] * 4]) = rb2;
        __syncthreads();
        if (kt + 1 < nk) G32_GLOAD((kt + 1) * G32_BK);
#pragma unroll
        for (int kk = 0; kk < G32_BK; kk += 2) {
            const int kl = kk + (lane >> 5);
            float af[2], bf[3];
#pragma unroll
            for (int i = 0; i < 2; ++i) af[i] = As[kl][wm * 64 + i * 32 + (lane & 31)];
#pragma unroll
            for (int j = 0; j < 3; ++j) bf[j] = Bs[kl][wn * 96 + j * 32 + (lane & 31)];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
    }
    // epilogue: C/D layout of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int frow = row0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            if (frow < p.rows) {
                const int arr = frow / p.chunk_frames, fl = frow - arr * p.chunk_frames;
                float *crow = p.C + (long long)blockIdx.z * p.c_plane_elems + ((long long)arr * p.total_frames + p.frame0 + fl) * p.Dp + col0 + wn * 96 + (lane & 31);
#pragma unroll
                for (int j = 0; j < 3; ++j) crow[j * 32] = acc[i][j][r];
            }
        }
}

// ---------------------------------------------------------------------------------------
// fp16 operands, fp32 accumulate: v_mfma_f32_32x32x16_f16.  A: [rows][planes*Kp] halves
// (k contiguous, lo plane after hi plane), B: [planes][Dp][Kp] halves (k contiguous).
// Block tile 128 x BN (192 or 64), BK = 32 halves.  Fragments: lane l holds
// A[row l&31][k = 8 (l>>5) + 0..7] (one 16-byte LDS read).  LDS rows are 64 B + 16 B pad
// (stride 80 B): ds_read_b128 of 16-lane groups then covers all 64 banks without conflict
// (rows r..r+15 at stride 20 dwords hit distinct 4-bank slots).
// SPLIT = false: C += Ahi Bhi.   SPLIT = true: C += Ahi Bhi + Alo Bhi + Ahi Blo  (~fp32 accuracy).
// ---------------------------------------------------------------------------------------
constexpr int G16_BM = 128, G16_BK = 32, G16_LD = 40;   // LD in halves (80 B)

// BN = 192: 4 waves as 2 x 2, wave tile 64 x 96.  BN = 64 (grids of up to 64 angles: the reference's own 37 and
// the 2-microphone 61): 4 waves as 4 x 1, wave tile 32 x 64 -- a third of the B traffic, LDS and MFMA work of a
// 192-wide tile whose columns would mostly be padding.
// one output tile (128 rows x BN columns) over the K range of split z out of ksplit; As / Bs: the workgroup's LDS tiles
template <bool SPLIT, int BN>
__device__ __forceinline__ void gemm_f16_tile(const GemmArgs &p, int row0, int col0, int z, int ksplit, int n_rows, long long plane_elems,
                                              _Float16 (*As)[G16_BM][G16_LD], _Float16 (*Bs)[BN][G16_LD])
{
    constexpr int NP = SPLIT ? 2 : 1;
    constexpr int WN = BN == 192 ? 2 : 1, WM = 4 / WN;            // waves across columns / rows
    constexpr int NI = G16_BM / WM / 32, NJ = BN / WN / 32;       // 32 x 32 MFMA tiles per wave
    constexpr int NBL = BN * 4 / 256;                             // 16-byte B chunks per thread and plane
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const _Float16 *A = reinterpret_cast<const _Float16 *>(p.A);
    const _Float16 *B = reinterpret_cast<const _Float16 *>(p.B);

    f32x16 acc[NI][NJ];
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // global loads: 16 B = 8 halves per thread; a BK=32 row is 64 B = 4 chunks.
    // A tile: 128 rows x 4 chunks = 512 chunks -> 2 per thread; B tile: BN x 4 chunks -> NBL per thread.
    const int lr = tid >> 2, lc = tid & 3;
    f16x8 ra[NP][2], rb[NP][NBL];
    const int ar0 = min(row0 + lr, n_rows - 1), ar1 = min(row0 + lr + 64, n_rows - 1);
#define G16_GLOAD(k0)                                                                                                          \
    _Pragma("unroll") for (int pl = 0; pl < NP; ++pl) {                                                                        \
        ra[pl][0] = *reinterpret_cast<const f16x8 *>(A + (long long)ar0 * p.a_row_elems + pl * p.Kp + (k0) + lc * 8);          \
        ra[pl][1] = *reinterpret_cast<const f16x8 *>(A + (long long)ar1 * p.a_row_elems + pl * p.Kp + (k0) + lc * 8);          \
        _Pragma("unroll") for (int i = 0; i < NBL; ++i)                                                                        \
            rb[pl][i] = *reinterpret_cast<const f16x8 *>(B + ((long long)pl * p.Dp + col0 + lr + 64 * i) * p.Kp + (k0) + lc * 8); \
    }
    const int nk_all = p.Kp / G16_BK, per = (nk_all + ksplit - 1) / ksplit;   // split-K over blockIdx.z
    const int kt0 = z * per, nk = min(kt0 + per, nk_all);
    if (kt0 < nk) { G16_GLOAD(kt0 * G16_BK) }     // (an empty split -- more segments than K steps -- loads nothing and leaves a zero tile)
    for (int kt = kt0; kt < nk; ++kt) {
#pragma unroll
        for (int pl = 0; pl < NP; ++pl) {
            *reinterpret_cast<f16x8 *>(&As[pl][lr][lc * 8]) = ra[pl][0];
            *reinterpret_cast<f16x8 *>(&As[pl][lr + 64][lc * 8]) = ra[pl][1];
#pragma unroll
            for (int i = 0; i < NBL; ++i) *reinterpret_cast<f16x8 *>(&Bs[pl][lr + 64 * i][lc * 8]) = rb[pl][i];
        }
        __syncthreads();
        if (kt + 1 < nk) { G16_GLOAD((kt + 1) * G16_BK) }
#pragma unroll
        for (int kk = 0; kk < G16_BK; kk += 16) {
            const int ko = kk + 8 * (lane >> 5);
            f16x8 af[NP][NI], bf[NP][NJ];
#pragma unroll
            for (int pl = 0; pl < NP; ++pl) {
#pragma unroll
                for (int i = 0; i < NI; ++i) af[pl][i] = *reinterpret_cast<const f16x8 *>(&As[pl][wm * (32 * NI) + i * 32 + (lane & 31)][ko]);
#pragma unroll
                for (int j = 0; j < NJ; ++j) bf[pl][j] = *reinterpret_cast<const f16x8 *>(&Bs[pl][wn * (32 * NJ) + j * 32 + (lane & 31)][ko]);
            }
#pragma unroll
            for (int i = 0; i < NI; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    if constexpr (SPLIT) {
                        // small terms first so they are not absorbed by the large partial sum
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[1][i], bf[0][j], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[0][i], bf[1][j], acc[i][j], 0, 0, 0);
                    }
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[0][i], bf[0][j], acc[i][j], 0, 0, 0);
                }
        }
        __syncthreads();
    }
#undef G16_GLOAD
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int frow = row0 + wm * (32 * NI) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            if (frow < n_rows) {
                const int arr = frow / p.chunk_frames, fl = frow - arr * p.chunk_frames;
                float *crow = p.C + (long long)z * plane_elems + ((long long)arr * p.total_frames + p.frame0 + fl) * p.Dp + col0 + wn * (32 * NJ) + (lane & 31);
#pragma unroll
                for (int j = 0; j < NJ; ++j) crow[j * 32] = acc[i][j][r];
            }
        }
}


template <bool SPLIT, int BN>
__global__ __launch_bounds__(256) void k_srp_gemm_f16(GemmArgs p)
{
    __shared__ __attribute__((aligned(16))) _Float16 As[SPLIT ? 2 : 1][G16_BM][G16_LD];
    __shared__ __attribute__((aligned(16))) _Float16 Bs[SPLIT ? 2 : 1][BN][G16_LD];
    gemm_f16_tile<SPLIT, BN>(p, blockIdx.x * G16_BM, blockIdx.y * BN, blockIdx.z, gridDim.z, p.rows, p.c_plane_elems, As, Bs);
}

// The repair contraction of the adaptive SRP precision: the three-product kernel above on however many rows the plan listed.
// The row count lives on the device, so the launch cannot be sized for it; a fixed, moderate grid walks the work items
// (row tile, column tile, K split) instead -- a grid sized for the worst case would spend ~90 us retiring empty
// workgroups.  Partial maps: [repair_ksplit][repair_plane_stride] (mca_internal.h), summed by k_repair_patch.
template <int BN>
__global__ __launch_bounds__(256) void k_srp_gemm_repair(GemmArgs p)
{
    __shared__ __attribute__((aligned(16))) _Float16 As[2][G16_BM][G16_LD];
    __shared__ __attribute__((aligned(16))) _Float16 Bs[2][BN][G16_LD];
    const int n_rows = min(p.rows, (*p.n_list - p.list0) * REPAIR_GROUP);
    if (n_rows <= 0) return;
    const int col_tiles = p.Dp / BN, row_tiles = (n_rows + G16_BM - 1) / G16_BM;
    const int ksplit = repair_ksplit_eff(p.repair_ksplit, n_rows, p.Dp, p.repair_items);
    const long long plane_elems = repair_plane_stride(n_rows, p.Dp);
    const int n_work = row_tiles * col_tiles * ksplit;
    for (int w = blockIdx.x; w < n_work; w += gridDim.x) {
        // K split fastest: the workgroups that share an output tile (and its A rows) run at the same time
        const int z = w % ksplit, t = w / ksplit;
        const int ct = t % col_tiles, rt = t / col_tiles;
        gemm_f16_tile<true, BN>(p, rt * G16_BM, ct * BN, z, ksplit, n_rows, plane_elems, As, Bs);
    }
}

// ---------------------------------------------------------------------------------------
// k_srp_cand -- the repair contraction at the CANDIDATE COLUMNS only (CandArgs, mca_internal.h): one workgroup of sixteen waves per
// listed unit, cand_unit.h.  A fixed grid walks the units.  (4 / 8-microphone contexts do this inside the list-mode analysis launch,
// StftPhatArgs::cand_on; this launch serves the contexts whose list-mode analysis is another kernel.)
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void k_srp_cand(CandArgs p)
{
    __shared__ __attribute__((aligned(16))) unsigned char lds[cand_unit_lds_bytes<16>()];
    const int n_here = min(*p.n_list - p.list0, p.pass_rows / REPAIR_GROUP);
    for (int g = blockIdx.x; g < n_here; g += gridDim.x) cand_unit<16>(p, g, p.list[p.list0 + g], lds, (int)threadIdx.x);
}

template __global__ void k_srp_gemm_repair<192>(GemmArgs);
template __global__ void k_srp_gemm_repair<64>(GemmArgs);

template __global__ void k_srp_gemm_f16<false, 192>(GemmArgs);
template __global__ void k_srp_gemm_f16<true, 192>(GemmArgs);
template __global__ void k_srp_gemm_f16<false, 64>(GemmArgs);
template __global__ void k_srp_gemm_f16<true, 64>(GemmArgs);

// ---------------------------------------------------------------------------------------
// The 256 x 384 contraction kernel for Dp == 384 (the 361-angle grid).  Measured on MI355X this contraction is
// bound by operand delivery into LDS, not by the MFMA pipe (ablation of its 16-deep predecessor, ms per launch
// of the bench shape: full 0.60, no MFMAs 0.43, no DMA 0.37; HISTORY.md).  The design goal is therefore
// the fewest operand bytes per CU, i.e. the largest output tile the register file can hold, and request shapes
// the memory system likes:
//   * a 256 x 384 output tile per workgroup (8 waves as 4 x 2, wave tile 64 x 192 = 2 x 6 MFMA tiles, 192
//     accumulator registers = 75 % of the CU's register file); the 128 x 192 kernels above (two workgroups
//     per CU) move twice the bytes per flop;
//   * K split over blockIdx.y (two partial maps, summed by the scan kernels) so that 32 768 rows still
//     give one workgroup per CU;
//   * operands by direct global->LDS loads (no staging registers, no ds_write) through a two-stage ring of
//     32-deep K stages that uses the whole 160 KiB of LDS, ONE barrier per stage.  The stage depth is about
//     the request shape: out of the row-major A a 16-deep stage gives every LDS-DMA instruction 32 pieces of
//     32 B, a 32-deep stage 16 pieces of 64 B (0.49 -> 0.46 ms per launch).  The steering table is stored
//     tiled per stage ([plane][stage][384][32]) so that each of its DMA instructions is one contiguous KiB (out
//     of a row-major table: 0.26 instead of 0.16 ms per launch for B alone); A stays row-major because a tiled
//     A costs its producer more than it saves here;
//   * LDS rows are 64 B = four 16-B chunks; the physical chunk is the logical one XOR ((row >> 2) & 3) --
//     applied on the per-lane SOURCE address of the LDS-DMA and on the ds_read_b128 address -- which makes the
//     16 rows of a read group hit 16 distinct 4-bank slots;
//   * B fragments in two rolling register slots (see the loop).
// A variant that computed the steering operand into LDS with sincospif + rotations instead of loading it
// passed every parity test and ran at the same speed (its VALU work took the place of the loads).
// ---------------------------------------------------------------------------------------
constexpr int V2_BM = 256, V2_BN = 384;

typedef __attribute__((address_space(3))) void lds_void_t;

constexpr int V3_BK = 32, V3_ROWB = 64;

template <bool SPLIT>
__global__ __launch_bounds__(512) void k_srp_gemm_f16_v2(GemmArgs p)
{
    constexpr int NP = SPLIT ? 2 : 1;
    constexpr int A_BYTES = V2_BM * V3_ROWB, B_BYTES = V2_BN * V3_ROWB;      // per plane: 16 KiB, 24 KiB
    constexpr int STAGE = NP * (A_BYTES + B_BYTES);                         // 80 KiB (SPLIT)
    // stages in LDS: two with the hi + lo planes (all 160 KiB); the one-plane kernel has room for four, so that the loads of a
    // stage are issued three stages (~2 us) ahead of their use instead of one
    constexpr int NS = SPLIT ? 2 : 4;
    constexpr int LPS = NP * 5;                                             // load instructions per wave and stage (2 A + 3 B per plane)
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem_g[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int row0 = blockIdx.x * V2_BM;
    const unsigned char *A = reinterpret_cast<const unsigned char *>(p.A);
    const unsigned char *B = reinterpret_cast<const unsigned char *>(p.Bt);  // [plane][stage][384][32]

    const int nst_all = p.Kp / V3_BK;
    const int per = (nst_all + gridDim.y - 1) / gridDim.y;
    const int s_beg = blockIdx.y * per, s_end = min(s_beg + per, nst_all);
    const int ns = s_end - s_beg;

    // one LDS-DMA instruction = 16 rows x 64 B: lane -> (row lane >> 2, physical chunk lane & 3)
    const int lrow = lane >> 2, pch = lane & 3;
    // A: 16 instructions per plane and stage, wave w takes row blocks w and w + 8 (16 rows each)
    // B: 24 instructions per plane and stage, wave w takes row blocks w, w + 8, w + 16
    // (the swizzle term (row >> 2) & 3 only depends on lrow: row blocks start at multiples of 16 rows, so one
    // per-lane pointer per operand plus uniform block offsets is enough)
    const int r0 = wave * 16 + lrow;
    const int lch = pch ^ ((r0 >> 2) & 3);
    const unsigned char *a_lane = A + ((long long)(row0 + r0) * p.a_row_elems + lch * 8) * 2;
    const unsigned char *b_lane = B + r0 * V3_ROWB + lch * 16;
    const long long a_blk = (long long)128 * p.a_row_elems * 2;                  // 8 row blocks of 16 rows further
    const int dst0 = wave * 1024;
    const long long a_pl = (long long)p.Kp * 2, b_pl = (long long)nst_all * B_BYTES;

    f32x16 acc[2][6];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 6; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // The loads of the next stage are issued in two bursts (A at the top of a stage, B between its two k-steps):
    // one burst of 80 instructions per CU queues up behind itself while the LDS reads of the stage start
    // (0.455 -> 0.428 ms per launch; one load after every MFMA group is no better and costs registers).
    auto issue_a = [&](int s, int buf) {
        const long long koff_a = (long long)(s_beg + s) * V3_BK * 2;
        unsigned char *sb = smem_g + buf * STAGE;
#pragma unroll
        for (int pl = 0; pl < NP; ++pl)
#pragma unroll
            for (int i = 0; i < 2; ++i)
                __builtin_amdgcn_global_load_lds(reinterpret_cast<const void *>(a_lane + pl * a_pl + i * a_blk + koff_a),
                                                 (lds_void_t *)(sb + pl * A_BYTES + dst0 + i * 8192), 16, 0, 0);
    };
    auto issue_b = [&](int s, int buf) {
        const long long koff_b = (long long)(s_beg + s) * B_BYTES;
        unsigned char *sb = smem_g + buf * STAGE;
#pragma unroll
        for (int pl = 0; pl < NP; ++pl)
#pragma unroll
            for (int i = 0; i < 3; ++i)
                __builtin_amdgcn_global_load_lds(reinterpret_cast<const void *>(b_lane + pl * b_pl + i * (128 * V3_ROWB) + koff_b),
                                                 (lds_void_t *)(sb + NP * A_BYTES + pl * B_BYTES + dst0 + i * 8192), 16, 0, 0);
    };

    // fragment addresses: lane (row l & 31, k half l >> 5); k-step kk of the stage is logical chunk 2 kk + (l >> 5);
    // row blocks are 32 rows = 2 KiB apart, and the swizzle term (row >> 2) & 3 is the same for rows 32 apart
    const int ra = wm * 64 + (lane & 31), rb = wn * 192 + (lane & 31);
    // (k-step 1 is logical chunk 2 + (l >> 5): its address is that of k-step 0 with byte-offset bit 5 flipped)
    const int a_off0 = ra * V3_ROWB + ((((lane >> 5)) ^ ((ra >> 2) & 3)) << 4);
    const int b_off0 = rb * V3_ROWB + ((((lane >> 5)) ^ ((rb >> 2) & 3)) << 4);
    const unsigned lds0 = (unsigned)(uintptr_t)((lds_void_t *)smem_g);

#pragma unroll
    for (int q = 0; q < NS - 1; ++q)
        if (q < ns) { issue_a(q, q); issue_b(q, q); }
    for (int s = 0; s < ns; ++s) {
        // the loads of stage s have landed when at most those of the stages issued after it are outstanding (loads of one
        // wave return in order)
        const int later = min(NS - 2, ns - 1 - s);
        if (later >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * LPS) : "memory");
        else if (later == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPS) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (s + NS - 1 < ns) issue_a(s + NS - 1, (s + NS - 1) % NS);      // that buffer was last read in stage s-1
#define LDS_RD(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            if (kk == 1 && s + NS - 1 < ns) issue_b(s + NS - 1, (s + NS - 1) % NS);
            // Rolling B fragments: two register slots (a third would spill).  A and column blocks 0, 1 are requested
            // up front; as soon as the MFMAs of block j have issued, block j+2 is requested into the slot they read, so
            // the LDS reads run under the matrix work.  The reads and their waits are inline asm: the compiler waits
            // for lgkmcnt(0) before every MFMA group (it does not count outstanding ds_read_b128), which would
            // serialise read and compute phases.  LDS reads of one wave return in order, so "at most N younger reads
            // outstanding" is exact; every wait lists the registers it releases as in/out operands so that nothing
            // can read them earlier.
            f16x8 af[NP][2], bs[2][NP];
            const unsigned a_addr = lds0 + (s % NS) * STAGE + (a_off0 ^ (kk << 5));
            const unsigned b_addr = lds0 + (s % NS) * STAGE + NP * A_BYTES + (b_off0 ^ (kk << 5));
#pragma unroll
            for (int pl = 0; pl < NP; ++pl) {
                LDS_RD(af[pl][0], a_addr, pl * A_BYTES);
                LDS_RD(af[pl][1], a_addr, pl * A_BYTES + 32 * V3_ROWB);
            }
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int pl = 0; pl < NP; ++pl) LDS_RD(bs[j][pl], b_addr, pl * B_BYTES + j * 32 * V3_ROWB);
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                constexpr int Y1 = NP;                 // younger reads allowed in flight: block j+1
                if constexpr (SPLIT) {
                    if (j == 0) asm volatile("s_waitcnt lgkmcnt(%6)" : "+v"(af[0][0]), "+v"(af[0][1]), "+v"(af[1][0]), "+v"(af[1][1]), "+v"(bs[0][0]), "+v"(bs[0][1]) : "n"(Y1));
                    else if (j < 5) asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(bs[j % 2][0]), "+v"(bs[j % 2][1]) : "n"(Y1));
                    else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(bs[j % 2][0]), "+v"(bs[j % 2][1]));
                    acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[1][0], bs[j % 2][0], acc[0][j], 0, 0, 0);
                    acc[1][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[1][1], bs[j % 2][0], acc[1][j], 0, 0, 0);
                    acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[0][0], bs[j % 2][1], acc[0][j], 0, 0, 0);
                    acc[1][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[0][1], bs[j % 2][1], acc[1][j], 0, 0, 0);
                } else {
                    if (j == 0) asm volatile("s_waitcnt lgkmcnt(%3)" : "+v"(af[0][0]), "+v"(af[0][1]), "+v"(bs[0][0]) : "n"(Y1));
                    else if (j < 5) asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(bs[j % 2][0]) : "n"(Y1));
                    else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(bs[j % 2][0]));
                }
                acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[0][0], bs[j % 2][0], acc[0][j], 0, 0, 0);
                acc[1][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[0][1], bs[j % 2][0], acc[1][j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (j + 2 < 6) {
#pragma unroll
                    for (int pl = 0; pl < NP; ++pl) LDS_RD(bs[j % 2][pl], b_addr, pl * B_BYTES + (j + 2) * 32 * V3_ROWB);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#undef LDS_RD
    }
    float *Cp = p.C + (long long)blockIdx.y * p.c_plane_elems;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int frow = row0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            if (frow < p.rows) {
                const int arr = frow / p.chunk_frames, fl = frow - arr * p.chunk_frames;
                float *crow = Cp + ((long long)arr * p.total_frames + p.frame0 + fl) * p.Dp + wn * 192 + (lane & 31);
#pragma unroll
                for (int j = 0; j < 6; ++j) crow[j * 32] = acc[i][j][r];
            }
        }
    if constexpr (!SPLIT) if (p.part) {      // (the two-plane kernel has no registers to spare for it)
        // Every 32-row block of the tile is one chunk of the scan over frames (the host only asks for this when that holds):
        // its chunk-local recursion result is a weighted sum over the block's rows, 16 of them in this lane and 16 in
        // lane ^ 32, so k_scan_partial's pass over the map is not needed.
        const int h = lane >> 5;
        float wt[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) wt[r] = h ? p.scan_w[(r & 3) + 8 * (r >> 2) + 4] : p.scan_w[(r & 3) + 8 * (r >> 2)];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int frow0 = row0 + wm * 64 + i * 32;
            if (frow0 < p.rows) {
                const int arr = frow0 / p.chunk_frames, chunk = (p.frame0 + frow0 - arr * p.chunk_frames) >> 5;
                float *out = p.part + blockIdx.y * p.part_plane_stride + ((long long)arr * p.n_chunks + chunk) * p.D + wn * 192 + (lane & 31);
#pragma unroll
                for (int j = 0; j < 6; ++j) {
                    float sacc = 0.f;
#pragma unroll
                    for (int r = 0; r < 16; ++r) sacc = fmaf(wt[r], acc[i][j][r], sacc);
                    sacc += __shfl_xor(sacc, 32);
                    if (h == 0 && wn * 192 + j * 32 + (lane & 31) < p.D) out[j * 32] = sacc;
                }
                if (wn == 0 && lane == 0 && blockIdx.y == 0) p.nvoiced[(long long)arr * p.n_chunks + chunk] = 32;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------
// k_srp_gemm_f16_v3: the one-plane 256 x 384 contraction on v_mfma_f32_16x16x32_f16.
// Same tile, stages, DMA and split-K as k_srp_gemm_f16_v2<false>; the 16x16x32 form needs the same fragment bytes per
// MAC (a fragment spans the stage's whole 32-deep K: 4 A + 12 B reads and 48 MFMAs per stage and wave instead of
// 2 x (2 + 6) reads and 24 MFMAs) and runs the matrix pipe at a higher clock under the same load (timing probe with the
// 32x32 kernel's operand stream: 168.6 -> 155.4 us).  Fragment: lane l holds row (l & 15), K chunk (l >> 4) of the
// 64-byte LDS row, i.e. a ds_read_b128 covers 16 whole rows; with the v2 swizzle its lane groups would hit banks
// twice, so the physical chunk is the logical one XOR g[(row >> 2) & 3], g = {0, 2, 3, 1} (every 16-lane read group
// then takes 16 distinct (row % 4, chunk) slots), on the DMA source address and on the read address alike.
// C/D of the MFMA: col = lane & 15, row = 4 (lane >> 4) + r.
// ---------------------------------------------------------------------------------------
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ int v3_swz(int row) { return (0x78 >> (2 * ((row >> 2) & 3))) & 3; }

__global__ __launch_bounds__(512) void k_srp_gemm_f16_v3(GemmArgs p)
{
    constexpr int A_BYTES = V2_BM * V3_ROWB, B_BYTES = V2_BN * V3_ROWB;      // 16 KiB, 24 KiB
    constexpr int STAGE = A_BYTES + B_BYTES;                                // 40 KiB: four stages in LDS
    constexpr int NS = 4, LPS = 5;                                          // load instructions per wave and stage (2 A + 3 B)
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem_g[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int row0 = blockIdx.x * V2_BM;
    const unsigned char *A = reinterpret_cast<const unsigned char *>(p.A);
    const unsigned char *B = reinterpret_cast<const unsigned char *>(p.Bt);  // [stage][384][32]

    const int nst_all = p.Kp / V3_BK;
    const int per = (nst_all + gridDim.y - 1) / gridDim.y;
    const int s_beg = blockIdx.y * per, s_end = min(s_beg + per, nst_all);
    const int ns = s_end - s_beg;

    // one LDS-DMA instruction = 16 rows x 64 B: lane -> (row lane >> 2, physical chunk lane & 3); wave w takes the A row
    // blocks w, w + 8 and the B row blocks w, w + 8, w + 16 (16 rows each; blocks start at multiples of 16 rows, so the
    // swizzle term only depends on the lane's row inside the block)
    const int lrow = lane >> 2, pch = lane & 3;
    const int r0 = wave * 16 + lrow;
    const int lch = pch ^ v3_swz(r0);
    const unsigned char *a_lane = A + ((long long)(row0 + r0) * p.a_row_elems + lch * 8) * 2;
    const unsigned char *b_lane = B + r0 * V3_ROWB + lch * 16;
    const long long a_blk = (long long)128 * p.a_row_elems * 2;
    const int dst0 = wave * 1024;

    f32x4 acc[4][12];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 12; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;

    auto issue_a = [&](int s, int buf) {
        const long long koff_a = (long long)(s_beg + s) * V3_BK * 2;
        unsigned char *sb = smem_g + buf * STAGE;
#pragma unroll
        for (int i = 0; i < 2; ++i)
            __builtin_amdgcn_global_load_lds(reinterpret_cast<const void *>(a_lane + i * a_blk + koff_a), (lds_void_t *)(sb + dst0 + i * 8192), 16, 0, 0);
    };
    auto issue_b = [&](int s, int buf) {
        const long long koff_b = (long long)(s_beg + s) * B_BYTES;
        unsigned char *sb = smem_g + buf * STAGE;
#pragma unroll
        for (int i = 0; i < 3; ++i)
            __builtin_amdgcn_global_load_lds(reinterpret_cast<const void *>(b_lane + i * (128 * V3_ROWB) + koff_b), (lds_void_t *)(sb + A_BYTES + dst0 + i * 8192), 16, 0, 0);
    };

    // fragment addresses: lane (row l & 15, K chunk l >> 4); row blocks are 16 rows = 1 KiB apart
    const int ra = wm * 64 + (lane & 15), rb = wn * 192 + (lane & 15);
    const int a_off0 = ra * V3_ROWB + (((lane >> 4) ^ v3_swz(ra)) << 4);
    const int b_off0 = rb * V3_ROWB + (((lane >> 4) ^ v3_swz(rb)) << 4);
    const unsigned lds0 = (unsigned)(uintptr_t)((lds_void_t *)smem_g);

#pragma unroll
    for (int q = 0; q < NS - 1; ++q)
        if (q < ns) { issue_a(q, q); issue_b(q, q); }
    for (int s = 0; s < ns; ++s) {
        const int later = min(NS - 2, ns - 1 - s);                 // stages issued after this one (loads of a wave return in order)
        if (later >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * LPS) : "memory");
        else if (later == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPS) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        const bool more = s + NS - 1 < ns;
        if (more) issue_a(s + NS - 1, (s + NS - 1) % NS);          // that buffer was last read in stage s-1
#define LDS_RD(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
        f16x8 af[4], bs[2];
        const unsigned a_addr = lds0 + (s % NS) * STAGE + a_off0;
        const unsigned b_addr = lds0 + (s % NS) * STAGE + A_BYTES + b_off0;
        LDS_RD(af[0], a_addr, 0); LDS_RD(af[1], a_addr, 1024); LDS_RD(af[2], a_addr, 2048); LDS_RD(af[3], a_addr, 3072);
        LDS_RD(bs[0], b_addr, 0); LDS_RD(bs[1], b_addr, 1024);
        // rolling B fragments as in v2: block j's MFMAs issue, then block j + 2 is requested into the slot they read
#pragma unroll
        for (int j = 0; j < 12; ++j) {
            // (the B burst behind the last column block's reads: measured at column block 2 / 4 / 6 / 9 / 11: 159.7 / 157 / 156 / 154.5 / 153.5 us)
            if (j == 11 && more) issue_b(s + NS - 1, (s + NS - 1) % NS);
            if (j == 0) asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(af[0]), "+v"(af[1]), "+v"(af[2]), "+v"(af[3]), "+v"(bs[0]));
            else if (j < 11) asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(bs[j % 2]));
            else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(bs[j % 2]));
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[i], bs[j % 2], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (j + 2 < 12) LDS_RD(bs[j % 2], b_addr, (j + 2) * 1024);
            __builtin_amdgcn_sched_barrier(0);
        }
#undef LDS_RD
    }
    float *Cp = p.C + (long long)blockIdx.y * p.c_plane_elems;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int frow = row0 + wm * 64 + i * 16 + 4 * (lane >> 4) + r;
            if (frow < p.rows) {
                const int arr = frow / p.chunk_frames, fl = frow - arr * p.chunk_frames;
                float *crow = Cp + ((long long)arr * p.total_frames + p.frame0 + fl) * p.Dp + wn * 192 + (lane & 15);
#pragma unroll
                for (int j = 0; j < 12; ++j) crow[j * 16] = acc[i][j][r];
            }
        }
    if (p.part) {
        // Every 32-row block of the tile is one chunk of the scan over frames (the host only asks for this when that holds):
        // its chunk-local recursion result is a weighted sum over the block's rows -- 8 of them in this lane (two 16-row MFMA
        // blocks x 4 registers), the rest in the lanes 16, 32 and 48 further -- so k_scan_partial's pass over the map is not needed.
        float wt[2][4];
#pragma unroll
        for (int b2 = 0; b2 < 2; ++b2)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int g4 = lane >> 4;                           // (selects, not an indexed read of the argument block)
                const float w01 = g4 == 0 ? p.scan_w[b2 * 16 + r] : p.scan_w[b2 * 16 + 4 + r];
                const float w23 = g4 == 2 ? p.scan_w[b2 * 16 + 8 + r] : p.scan_w[b2 * 16 + 12 + r];
                wt[b2][r] = g4 < 2 ? w01 : w23;
            }
#pragma unroll
        for (int ch = 0; ch < 2; ++ch) {
            const int frow0 = row0 + wm * 64 + ch * 32;
            if (frow0 < p.rows) {
                const int arr = frow0 / p.chunk_frames, chunk = (p.frame0 + frow0 - arr * p.chunk_frames) >> 5;
                float *out = p.part + blockIdx.y * p.part_plane_stride + ((long long)arr * p.n_chunks + chunk) * p.D + wn * 192 + (lane & 15);
#pragma unroll
                for (int j = 0; j < 12; ++j) {
                    float sacc = 0.f;
#pragma unroll
                    for (int b2 = 0; b2 < 2; ++b2)
#pragma unroll
                        for (int r = 0; r < 4; ++r) sacc = fmaf(wt[b2][r], acc[2 * ch + b2][j][r], sacc);
                    sacc += __shfl_xor(sacc, 16);
                    sacc += __shfl_xor(sacc, 32);
                    if (lane < 16 && wn * 192 + j * 16 + lane < p.D) out[j * 16] = sacc;
                }
                if (wn == 0 && lane == 0 && blockIdx.y == 0) p.nvoiced[(long long)arr * p.n_chunks + chunk] = 32;
            }
        }
    }
}

template __global__ void k_srp_gemm_f16_v2<true>(GemmArgs);
template __global__ void k_srp_gemm_f16_v2<false>(GemmArgs);

}  // namespace mca
