// kernels_2048.hip -- 2048-sample frames (96 kHz at the reference's 0.025 s frame rate; the frame length of the reference's own
// beamformer test, test/test_mcarray.cpp:660-662: fftOrder = 11) on the wave-level 1024-point complex transform (round 6).
//
// A real 2048-sample frame is ONE 1024-point complex transform of z[n] = x[2n] + j x[2n+1] plus a split step:
//     2 E[k] = Z[k] + conj Z[1024 - k]        (the transform of the even samples)
//     2 O[k] = -j (Z[k] - conj Z[1024 - k])   (... of the odd samples)
//     2 X[k] = 2 E[k] + W^k 2 O[k],   2 X[1024 - k] = conj(2 E[k] - W^k 2 O[k]),   W = exp(-j 2 pi / 2048),   k = 0 .. 512
// so a wave transforms one CHANNEL per pass (no two channels share a transform here: nothing of pair_balance.h applies) at the cost per
// sample of the 1024-sample path, where the any-length kernels (kernels_generic.hip: a block-cooperative radix-2 transform, one barrier
// per stage) took 4 x as long per sample.  Lane layout as in k_stft_phat_wave: the transform leaves the bins lam + 64 s in lane l
// (lam = l for l <= 32, 96 - l above), the mirror Z[1024 - k] of a lane's bin k is brought over by v_permlane32_swap, and the lane ends up
// with X at its 8 low bins k = lam + 64 s (s < 8) and at their 8 mirrors 1024 - k; lane 0 also holds k = 512 (its own mirror) and, as the
// mirror of k = 0, the Nyquist bin 1024.
//
//   k_stft_phat_2048        PCM -> spectra of the frame's channels in LDS (wave = channel) -> PHAT -> pair / delay-group sums -> A
//                           (SteeringBeamforming.cpp:104-130 up to the steering sum; M <= 8)
//   k_bf_table_2048         steering rows of the delay-and-sum stage per grid angle and channel
//   k_beamform_wave_2048    PCM -> per channel: transform, split, Y += X_c T_c -> inverse split, inverse transform, overlap-add
//                           (Beamformer.cpp:51-71), one wave per run of frames, any M <= 16
#include "fft1024c.h"
#include "mca_internal.h"
#include "phat_pairs.h"

namespace mca {

constexpr int N2K = 2048, H2K = 1024, K2K = 1025;
constexpr int ROW2K = 1026;               // float2 words per spectrum row in LDS (rows 4 banks apart: the pair stage reads 8 rows at one bin)
constexpr int TROW2K = 1032;              // float2 words per steering row

// After fft1024c(..., row = lam): z[dr16(s)] = Z[lam + 64 s].  Brings the mirrors over: on return the mirror Z[1024 - k] of the lane's bin
// k = lam + 64 s, s < 8, sits in z[mirror_slot(s)] (the lane's own bins s >= 8 are gone: they are the partner lane's mirrors).  Lanes 0 and
// 32 are their own mirrors (lane 0: register (16 - s) & 15, lane 32: register 15 - s).  z512: Z[512] of lane 0 (its own mirror), saved first.
__device__ __forceinline__ constexpr int mirror_slot(int s) { return dr16(15 - s < 12 ? 15 - s + 4 : 15 - s - 4); }
__device__ __forceinline__ void mirror_exchange(float2 (&z)[16], int lane, float2 &z512)
{
    z512 = z[dr16(8)];
    if (lane == 0) {
        float2 t[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) t[j] = z[dr16(j)];
#pragma unroll
        for (int j = 8; j < 16; ++j) z[dr16(j < 12 ? j + 4 : j - 4)] = t[(j + 1) & 15];       // slot mate(j) <- register j + 1
    } else if ((lane & 31) == 0) {
#pragma unroll
        for (int j = 8; j < 12; ++j) { const float2 t = z[dr16(j)]; z[dr16(j)] = z[dr16(j + 4)]; z[dr16(j + 4)] = t; }
    } else {
#pragma unroll
        for (int j = 8; j < 12; ++j) {
            float2 &u = z[dr16(j)], &w = z[dr16(j + 4)];
            swap_rows32(u.x, w.x); swap_rows32(w.x, u.x);
            swap_rows32(u.y, w.y); swap_rows32(w.y, u.y);
        }
    }
}

// the split step for the lane's pair (k, 1024 - k): lo = 2 X[k], hi = 2 X[1024 - k];  wk = W^k
__device__ __forceinline__ void split_pair(float2 zk, float2 zm, float2 wk, float2 &lo, float2 &hi)
{
    const float2 e2 = make_float2(zk.x + zm.x, zk.y - zm.y);          // 2 E[k]
    const float2 o2 = make_float2(zk.y + zm.y, zm.x - zk.x);          // 2 O[k]
    const float2 t = cmul(o2, wk);
    lo = cadd(e2, t);
    hi = make_float2(e2.x - t.x, t.y - e2.y);                         // conj(2 E - t)
}

// W^(lam + 64 s), s < 8
__device__ __forceinline__ void split_twiddles(float2 (&wk)[8], int lam)
{
#pragma unroll
    for (int s = 0; s < 8; ++s) wk[s] = twiddle(lam + 64 * s, N2K, false);
}

// --------------------------------------------------------------------------------------
// k_stft_phat_2048: grid (ceil(frames / fpb), arrays) x 512 threads.  Wave w transforms channel w of the frame (MT = 4: two frames per
// pass, wave w -> frame slot w >> 2, channel w & 3); the spectra go to LDS, [slot][channel][ROW2K]; then thread t whitens bins t and
// t + 512 of every channel and forms the pair products (pair_stage of phat_pairs.h, as k_stft_phat does for 1024-sample frames); the
// Nyquist bins of the run are parked and finished at the end, lane = frame.  Each PCM sample is read once per workgroup: the second half
// of a frame stays in registers as the first half of the next.
// LDS: spectra FP x M rows | twiddle table | 8 transform scratches | [fpb][M] Nyquist bins | [fpb][8] power partials.
// --------------------------------------------------------------------------------------
// MERGE (ULA of 4 / 8 microphones, one fp16 plane: the ADAPTIVE coarse pass, plain FP16): the contraction index is the product m = k (j - i)
// -- all (bin, spacing) combinations of equal product share one steering column (api.hip, build_merged_tables) -- so their PHAT sums are added
// up before the row is stored: 3 924 complex values per row instead of 7 x 1 025.  The 512 threads add their 2 x (M - 1) sums into ONE region
// per frame slot with ds_add_u32 on 2^26 fixed point (|sum| <= 28: 31 bits; the step 1.5e-8 is four decades below the fp16 rounding the row
// gets anyway), so the order in which the waves arrive does not matter: the row is the same bits every run.  The region lies on the
// transforms' scratches (idle during the pair stage); every wave zeroes its own scratch after its transform.
constexpr float MERGE_SCALE = 67108864.f, MERGE_UNSCALE = 1.f / 67108864.f;
// (the LDS atomics are what the merged stage costs -- +50 us per 16 384 frames against -68 us of contraction, profiles/r06_ab_merge_2048.log; one
// ds_add_u64 per complex addend, re + 2^32 im, measured slower than the two 32-bit ones: 0.430 against 0.415 ms)

template <int MT, bool ULA, typename OutT, bool MERGE>
__global__ __launch_bounds__(512) void k_stft_phat_2048(StftPhatArgs p)
{
    static_assert(!MERGE || (ULA && (MT == 4 || MT == 8) && sizeof(OutT) == 2), "the merged index serves the one-plane fp16 rows of a 4 / 8 microphone ULA");
    constexpr int FP = MT == 4 ? 2 : 1;                                   // frames per pass
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int M = MT > 0 ? MT : p.M;
    const int MR = MT == 4 ? 4 : 8;                                       // spectrum rows per frame slot
    float2 *spec = reinterpret_cast<float2 *>(smem_raw);                  // [FP][MR][ROW2K]
    float2 *tab = spec + FP * MR * ROW2K;
    float2 *scr = tab + F1K_TWORDS;                                       // [8][F1K_SCRATCH]
    float2 *nyq = scr + 8 * F1K_SCRATCH;                                  // [fpb][M]
    float *spow = reinterpret_cast<float *>(nyq + p.fpb * M);             // [fpb][8]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int slot = MT == 4 ? wave >> 2 : 0, ch = MT == 4 ? wave & 3 : wave;
    const bool fft_wave = ch < M;
    f1k_table_init(tab, tid, 512);
    F1kLane lc;
    lc.init(lane);
    const int lam = lane <= 32 ? lane : 96 - lane;
    float2 wk[8], win[16];
    split_twiddles(wk, lam);
#pragma unroll
    for (int i = 0; i < 16; ++i) win[i] = reinterpret_cast<const float2 *>(p.window)[lane + 64 * i];     // (w[2n], w[2n+1]), n = lane + 64 i
    for (int e = tid; e < p.fpb * 8; e += 512) spow[e] = 0.f;
    // MERGE: the region offsets of the thread's products k g (k = tid, tid + 512; g = 1 .. M - 1) and, threads 0 .. M - 2, of the Nyquist bin's
    const int nmp = MERGE ? (p.n_merged + 63) & ~63 : 0;
    int *msum = reinterpret_cast<int *>(scr);                             // [FP][2][nmp]: real parts | imaginary parts
    unsigned short rk[2][MT > 1 ? MT - 1 : 1], rkn = 0;
    if constexpr (MERGE) {
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int g = 0; g < MT - 1; ++g) rk[h][g] = p.mrank[(tid + 512 * h) * (g + 1)];
        if (tid < MT - 1) rkn = p.mrank[H2K * (tid + 1)];
    }
    // list mode (the repair pass of the adaptive SRP precision, as in k_stft_phat): the workgroups walk the listed units of REPAIR_GROUP
    // frames; unit number li - list0 of the pass writes the A rows (li - list0) * REPAIR_GROUP ...  Otherwise one run of fpb frames.
    const int li_end = p.list ? min(*p.n_list, p.list0 + p.list_cap) : 1, li_step = p.list ? (int)gridDim.x : 1;
    for (int li = p.list ? p.list0 + (int)blockIdx.x : 0; li < li_end; li += li_step) {
    int a = blockIdx.y;
    int f_begin = blockIdx.x * p.fpb;
    long long row_base = (long long)a * p.n_frames;                       // A row of frame f = row_base + f
    if (p.list) {
        const int e = p.list[li];
        a = e / p.groups_per_array;
        f_begin = (e - a * p.groups_per_array) * REPAIR_GROUP;
        row_base = (long long)(li - p.list0) * REPAIR_GROUP - f_begin;
    }
    const int f_end = min(f_begin + p.fpb, p.n_frames);
    __syncthreads();

    // cur[i] = (x[2n], x[2n+1]) of the wave's frame, n = lane + 64 i; the frames of a slot are FP apart
    const float2 *src = reinterpret_cast<const float2 *>(p.pcm + (long long)a * p.array_stride + (long long)ch * p.mic_stride) + lane;
    float2 cur[16];
    if (fft_wave && f_begin + slot < f_end) {
        const float2 *s0 = src + (long long)(p.frame0 + f_begin + slot) * (H2K / 2);
#pragma unroll
        for (int i = 0; i < 16; ++i) cur[i] = s0[64 * i];
    }
    for (int f = f_begin; f < f_end; f += FP) {
        const int nfr = min(FP, f_end - f);
        if (fft_wave && slot < nfr) {
            float2 z[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) z[i] = make_float2(cur[i].x * win[i].x, cur[i].y * win[i].y);
            // the next frame of this slot: FP = 1 keeps the second half, FP = 2 (frames two apart) reloads the whole frame
            if (f + FP + slot < f_end) {
                const float2 *s1 = src + (long long)(p.frame0 + f + FP + slot) * (H2K / 2);
                if (FP == 1) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) { cur[i] = cur[i + 8]; cur[i + 8] = s1[64 * (i + 8)]; }
                } else {
#pragma unroll
                    for (int i = 0; i < 16; ++i) cur[i] = s1[64 * i];
                }
            }
            fft1024c<false, 3>(z, scr + wave * F1K_SCRATCH, lane, tab, lc, F1kNoMid(), lam);
            float2 z512;
            mirror_exchange(z, lane, z512);
            float2 *row = spec + (slot * MR + ch) * ROW2K;
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                float2 lo, hi;
                split_pair(z[dr16(s)], z[mirror_slot(s)], wk[s], lo, hi);
                row[lam + 64 * s] = make_float2(0.5f * lo.x, 0.5f * lo.y);
                row[H2K - lam - 64 * s] = make_float2(0.5f * hi.x, 0.5f * hi.y);
            }
            if (lane == 0) row[512] = make_float2(z512.x, -z512.y);              // X[512] = conj Z[512]
        }
        if constexpr (MERGE) {
            wave_lds_fence();
#pragma unroll
            for (int i = 0; i < F1K_SCRATCH / 64; ++i)
                if (wave * F1K_SCRATCH + 64 * i < FP * nmp) scr[wave * F1K_SCRATCH + lane + 64 * i] = make_float2(0.f, 0.f);     // (the region's words only)
        }
        __syncthreads();
#pragma unroll
        for (int sl = 0; sl < FP; ++sl) {
            const int fr = f + sl;
            if (fr < f_end) {
                const float2 *xs = spec + sl * MR * ROW2K;
                OutT *arow = reinterpret_cast<OutT *>(p.A) + (row_base + fr) * (long long)p.a_row_elems;
                if (tid < M) nyq[(MERGE ? 0 : (fr - f_begin) * M) + tid] = whiten(xs[tid * ROW2K + H2K]);
                if (p.power) {
                    // dsp::SignalPower::FFTPower [INFERRED, SURVEY A.8]: (1/N^2) sum_k w_k |X[k]|^2, w = 2 except DC and Nyquist
                    float acc = 0.f, acc2 = 0.f;
                    for (int m = 0; m < M; ++m) {
                        const float2 u = xs[m * ROW2K + tid], v = xs[m * ROW2K + tid + 512];
                        acc += u.x * u.x + u.y * u.y; acc2 += v.x * v.x + v.y * v.y;
                    }
                    acc = (tid == 0 ? acc : 2.f * acc) + 2.f * acc2;
                    if (tid < M) { const float2 u = xs[tid * ROW2K + H2K]; acc += u.x * u.x + u.y * u.y; }
#pragma unroll
                    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off);
                    if (lane == 0) spow[(fr - f_begin) * 8 + wave] = acc;         // (one slot per wave, summed in wave order below)
                }
                if constexpr (MERGE) {
                    int *ms = msum + sl * nmp * 2;
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        float2 r[MT], out[MT - 1];
#pragma unroll
                        for (int m = 0; m < MT; ++m) r[m] = whiten(xs[m * ROW2K + tid + 512 * h]);
                        pair_products<MT, true>(r, out);
#pragma unroll
                        for (int g = 0; g < MT - 1; ++g) {               // (real parts | imaginary parts: an instruction's 64 words spread over all banks)
                            atomicAdd(ms + rk[h][g], __float2int_rn(out[g].x * MERGE_SCALE));
                            atomicAdd(ms + nmp + rk[h][g], __float2int_rn(out[g].y * MERGE_SCALE));
                        }
                    }
                    if (tid < MT - 1) {                                   // the Nyquist bin: spacing tid + 1 (its whitened values were written by this wave)
                        wave_lds_fence();
                        float2 acc = make_float2(0.f, 0.f);
                        for (int i = 0; i + tid + 1 < MT; ++i) acc = cmacc(acc, nyq[i], nyq[i + tid + 1]);
                        atomicAdd(ms + rkn, __float2int_rn(acc.x * MERGE_SCALE));
                        atomicAdd(ms + nmp + rkn, __float2int_rn(acc.y * MERGE_SCALE));
                    }
                    __syncthreads();
                    typedef _Float16 h2v __attribute__((ext_vector_type(2)));
                    h2v *arow2 = reinterpret_cast<h2v *>(arow);
                    for (int r = tid; r < nmp; r += 512) {
                        arow2[r] = h2v{(_Float16)((float)ms[r] * MERGE_UNSCALE), (_Float16)((float)ms[nmp + r] * MERGE_UNSCALE)};
                    }
                } else {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int k = tid + 512 * h;
                    if constexpr (MT == 0) {
                        float2 *xw = spec + sl * MR * ROW2K;
                        for (int m = 0; m < M; ++m) xw[m * ROW2K + k] = whiten(xw[m * ROW2K + k]);
                    }
                    pair_stage<MT, ULA, true, OutT>(xs + k, ROW2K, M, arow, p, k, K2K);
                }
                }
            }
        }
        __syncthreads();
    }
    if (p.power && tid < f_end - f_begin) {
        const float *s = spow + tid * 8;
        p.power[(long long)a * p.total_frames + p.frame0 + f_begin + tid] = (((((((s[0] + s[1]) + s[2]) + s[3]) + s[4]) + s[5]) + s[6]) + s[7]) / ((float)N2K * (float)N2K) / (float)M;
    }
    if (!MERGE && tid < f_end - f_begin) {
        OutT *arow = reinterpret_cast<OutT *>(p.A) + (row_base + f_begin + tid) * (long long)p.a_row_elems;
        pair_stage<MT, ULA, false, OutT>(nyq + tid * M, 1, M, arow, p, H2K, K2K);
    }
    }                                                                     // (list mode: the barrier at the top of the next unit keeps nyq until every thread is through)
}

#define INST_2048(MT, ULA, T) template __global__ void k_stft_phat_2048<MT, ULA, T, false>(StftPhatArgs);
INST_2048(0, false, float) INST_2048(0, true, float) INST_2048(4, false, float) INST_2048(4, true, float) INST_2048(8, false, float) INST_2048(8, true, float)
INST_2048(0, false, _Float16) INST_2048(0, true, _Float16) INST_2048(4, false, _Float16) INST_2048(4, true, _Float16) INST_2048(8, false, _Float16) INST_2048(8, true, _Float16)
template __global__ void k_stft_phat_2048<4, true, _Float16, true>(StftPhatArgs);
template __global__ void k_stft_phat_2048<8, true, _Float16, true>(StftPhatArgs);

// --------------------------------------------------------------------------------------
// k_bf_table_2048: grid (D + 1, M) x 256.  Row 0: DOA = 0 rad (the module's initial _currentDOA, BeamformingSeparationAndLocalisation.cpp:51),
// row 1 + d: grid[d].  T_c[k] = P_c[k] / (4096 M), P_c[k] = exp(j k s_c) (Beamformer.cpp:59-60), k = 0 .. 1024: the 1/M of :70, the two
// halvings of the forward and the inverse split step and the 1/1024 of the inverse transform in one factor.  Im P_c[1024] = 0: the
// reference's CCS inverse ignores the imaginary part of the Nyquist bin.
// --------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_bf_table_2048(float2 *tab, const float *grid, const double *mic_x, int M, double unit)
{
    const int d = blockIdx.x, c = blockIdx.y;
    const double doa = d == 0 ? 0.0 : (double)grid[d - 1];
    const double cd = cos(doa + 1.57079632679489661923);                   // cos(DOA + M_PI/2), Beamformer.cpp:59
    const double sc = 1.0 / ((double)M * 4096.0);
    float2 *row = tab + ((long long)d * M + c) * TROW2K;
    for (int k = threadIdx.x; k < TROW2K; k += 256) {
        double sn = 0.0, cs = 0.0;
        if (k <= H2K) {
            double turns = (double)k * (unit * mic_x[c] * cd);               // k s_c / (2 pi)
            turns -= rint(turns);
            sincospi(2.0 * turns, &sn, &cs);
            if (k == H2K) sn = 0.0;
        }
        row[k] = make_float2((float)(cs * sc), (float)(sn * sc));
    }
}

// --------------------------------------------------------------------------------------
// k_beamform_wave_2048: grid (workgroups per array, arrays, sources) x 256 threads = 4 waves, one wave per run of frames.  Overlap-add carries
// as in k_beamform_wave: a workgroup covers 4 ft - 1 consecutive frames; wave 0 takes the frame BEFORE them too (only its second half
// counts: the carry into the workgroup's first hop) and ft - 1 frames, waves 1..3 take ft frames each.  A wave does not wait for its
// predecessor: it stores its first hop without a carry, the waves leave their final carries in LDS, and after one barrier at the very end
// each wave adds its predecessor's carry to that hop.  One frame in 4 ft is analysed twice.
// Per frame and channel: windowed samples -> transform -> mirrors -> split -> Y += 2 X_c T_c[doa bin] at the lane's 17 bins; then the
// inverse split
//     2 E'[k] = Y[k] + conj Y[1024 - k],   2 O'[k] = (Y[k] - conj Y[1024 - k]) conj(W^k),   Z'[k] = E' + j O',   Z'[1024 - k] = conj E' + j conj O'
// through the wave's scratch into the transform's input order, the inverse transform (y[2n] + j y[2n+1] at n = lane + 64 i), overlap-add.
// --------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void k_beamform_wave_2048(BeamformWaveArgs p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float2 *tab = reinterpret_cast<float2 *>(smem_raw);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float2 *buf = tab + F1K_TWORDS + wave * F1K_SCRATCH;
    float2 *xcarry = tab + F1K_TWORDS + 4 * F1K_SCRATCH;                  // [4 waves][512] final carries
    f1k_table_init(tab, tid, 256);
    F1kLane lc;
    lc.init(lane);
    __syncthreads();
    const int a = blockIdx.y, S = p.S, M = p.M;
    const long long as = (long long)a * S + blockIdx.z;
    const int w0 = (int)blockIdx.x * (4 * p.ft - 1);                      // the workgroup's first frame
    const int t0 = wave == 0 ? w0 : w0 + wave * p.ft - 1, t1 = min(w0 + (wave + 1) * p.ft - 1, p.n_frames);
    const bool active = t0 < t1;
    const int tfirst = (wave == 0 && t0 > 0) ? t0 - 1 : t0;               // wave 0 analyses the frame before its run for the carry
    const int lam = lane <= 32 ? lane : 96 - lane;
    if (active) {
    float2 wk[8], win[16];
    split_twiddles(wk, lam);
#pragma unroll
    for (int i = 0; i < 16; ++i) win[i] = reinterpret_cast<const float2 *>(p.window)[lane + 64 * i];
    float2 carry[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) carry[i] = t0 == 0 ? reinterpret_cast<const float2 *>(p.tail_in + as * H2K)[lane + 64 * i] : make_float2(0.f, 0.f);
    const float2 *base = reinterpret_cast<const float2 *>(p.pcm + (long long)a * p.array_stride) + lane;
    const int *bins = p.doa_bin + (long long)a * p.n_frames * S + blockIdx.z;      // [frame][source]
    float2 x[16];
    auto load_ch = [&](int t, int c) {
        const float2 *s = base + ((long long)c * p.mic_stride + (long long)t * H2K) / 2;
#pragma unroll
        for (int i = 0; i < 16; ++i) x[i] = s[64 * i];
    };
    load_ch(tfirst, 0);
    for (int t = tfirst; t < t1; ++t) {
        const float2 *trow = p.table + (long long)(bins[(long long)t * S] + 1) * M * TROW2K;
        float2 Ylo[8], Yhi[8], Y512 = make_float2(0.f, 0.f);
#pragma unroll
        for (int s = 0; s < 8; ++s) { Ylo[s] = make_float2(0.f, 0.f); Yhi[s] = make_float2(0.f, 0.f); }
        for (int c = 0; c < M; ++c) {
            float2 z[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) z[i] = make_float2(x[i].x * win[i].x, x[i].y * win[i].y);
            const float2 *tr = trow + (long long)c * TROW2K;
            float2 Tlo[8], Thi[8];
            // the next channel's samples (the run's last step reloads its own) and this channel's steering row are requested in the
            // middle of the transform, where the fewest registers are live
            fft1024c<false, 3>(z, buf, lane, tab, lc, [&]() {
                const bool lastc = c == M - 1, last = lastc && t + 1 >= t1;
                load_ch(last ? t : (lastc ? t + 1 : t), last ? c : (lastc ? 0 : c + 1));
#pragma unroll
                for (int s = 0; s < 8; ++s) { Tlo[s] = tr[lam + 64 * s]; Thi[s] = tr[H2K - lam - 64 * s]; }
            }, lam);
            float2 z512;
            mirror_exchange(z, lane, z512);
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                float2 lo, hi;
                split_pair(z[dr16(s)], z[mirror_slot(s)], wk[s], lo, hi);
                Ylo[s] = cmac(Ylo[s], lo, Tlo[s]);
                Yhi[s] = cmac(Yhi[s], hi, Thi[s]);
            }
            // lane 0: 2 X[512] = 2 conj Z[512]
            Y512 = cmac(Y512, make_float2(2.f * z512.x, -2.f * z512.y), tr[512]);
        }
        // inverse split into the wave's scratch at the transform's input order: word n = bin
        wave_lds_fence();
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const float2 yk = Ylo[s], ym = Yhi[s];
            const float2 e2 = make_float2(yk.x + ym.x, yk.y - ym.y);          // 2 E'
            const float2 d = make_float2(yk.x - ym.x, yk.y + ym.y);           // Y[k] - conj Y[1024 - k]
            const float2 o2 = cmulc(d, wk[s]);                                // 2 O'
            const int k = lam + 64 * s;
            buf[k] = make_float2(e2.x - o2.y, e2.y + o2.x);                   // E' + j O'
            if (k != 0) buf[H2K - k] = make_float2(e2.x + o2.y, o2.x - e2.y); // conj E' + j conj O'
        }
        if (lane == 0) buf[512] = make_float2(2.f * Y512.x, -2.f * Y512.y);   // 2 Z'[512] = 2 conj Y[512]  (2 E' = 2 Re Y, 2 O' = -2 Im Y; every word holds 2 E' + j 2 O')
        wave_lds_fence();
        float2 y[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) y[i] = buf[lane + 64 * i];
        wave_lds_fence();
        fft1024c<true, 3>(y, buf, lane, tab, lc);
        // y[p] = (out[2n], out[2n+1]), n = lane + 64 dr16(p)
        if (t >= t0) {
            float2 *o = reinterpret_cast<float2 *>(p.out + as * p.n_frames * H2K + (long long)t * H2K) + lane;
#pragma unroll
            for (int i = 0; i < 8; ++i) o[64 * i] = make_float2(carry[i].x + y[dr16(i)].x, carry[i].y + y[dr16(i)].y);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) carry[i] = y[dr16(i + 8)];
    }
    if (t1 == p.n_frames) {
#pragma unroll
        for (int i = 0; i < 8; ++i) reinterpret_cast<float2 *>(p.tail_out + as * H2K)[lane + 64 * i] = carry[i];
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) xcarry[wave * 512 + lane + 64 * i] = carry[i];
    }
    __syncthreads();
    if (active && wave > 0) {                                             // (the hop was stored without a carry: the wave's own store, read back)
        float2 *o = reinterpret_cast<float2 *>(p.out + as * p.n_frames * H2K + (long long)t0 * H2K) + lane;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float2 c = xcarry[(wave - 1) * 512 + lane + 64 * i], v = o[64 * i];
            o[64 * i] = make_float2(v.x + c.x, v.y + c.y);
        }
    }
}

}  // namespace mca
