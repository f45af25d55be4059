// cand_unit.h -- the repair contraction of ONE listed unit (4 rows) at the unit's candidate columns (CandArgs, mca_internal.h), by the NW
// waves of a workgroup.  Shared by k_srp_cand (NW = 16: a launch of its own) and the list mode of k_stft_phat_wave (NW = 4: the workgroup
// that wrote the unit's rows contracts them right away).
//
// The shape is 4 x 8 with a depth of thousands, so the matrix instruction is the 16-block 4 x 4 x 4 one with the BLOCKS AS DEPTH SLICES:
// lane 4 b + q holds row q of A / column q of B over the eight depth positions of slice b of a 128-deep step (one 16-byte load per operand
// and plane; the sixteen lanes of a row read 256 contiguous bytes), two instructions per load.  The waves interleave the steps; their
// NW x 16 partial 4 x 4 tiles are summed in LDS in a fixed order (wave, slice) and the exact values go straight into the map (plane 0;
// zeros into the other planes): no partial maps, no patch kernel, and a value's bits depend on nothing but the row, the column and NW.
// Three products (lo hi, hi lo, hi hi) as the whole-row kernel.  The unit is the workgroup's alone: its column mask is taken and cleared,
// its test-and-set word released.  A unit's column groups (<= 8 columns each; the usual flat-topped peak needs six: one group) are taken
// one after the other -- a unit that asked for every column (46 groups) keeps its workgroup for a long time; by construction that is a
// frame whose coarse map has no guaranteed peak, and rows of exact zeros are not listed at all.
#pragma once
#include "mca_internal.h"

namespace mca {

typedef _Float16 cu_f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 cu_f16x4 __attribute__((ext_vector_type(4)));
typedef float cu_f32x4 __attribute__((ext_vector_type(4)));

template <int NW> constexpr size_t cand_unit_lds_bytes() { return (size_t)(NW * 2 * 4 * 64 + 32 * 8) * 4 + (CAND_WORDS_MAX + CAND_WORDS_MAX + 1) * 4 + CAND_WORDS_MAX * 32 * 2; }

// g: the unit's position among the rows of p.A (row = 4 g + frame), e: the unit.  Every thread of the workgroup calls it (barriers inside).
template <int NW>
__device__ __forceinline__ void cand_unit(const CandArgs &p, int g, int e, unsigned char *lds, int tid)
{
    static_assert(NW == 4 || NW == 8 || NW == 16, "eight threads per result element share the NW x 16 partial tiles");
    float (*red)[2][4][64] = reinterpret_cast<float (*)[2][4][64]>(lds);                     // [wave][column group][register = row][lane = 4 slice + column]
    float (*red2)[8] = reinterpret_cast<float (*)[8]>(lds + (size_t)NW * 2 * 4 * 64 * 4);     // [32 elements][8 parts]
    unsigned *s_mask = reinterpret_cast<unsigned *>(lds + (size_t)(NW * 2 * 4 * 64 + 32 * 8) * 4);
    int *s_pre = reinterpret_cast<int *>(s_mask + CAND_WORDS_MAX);
    unsigned short *s_cols = reinterpret_cast<unsigned short *>(s_pre + CAND_WORDS_MAX + 1);
    const int lane = tid & 63, wave = tid >> 6, words = p.umask_words;
    const _Float16 *A = reinterpret_cast<const _Float16 *>(p.A);
    const _Float16 *B = reinterpret_cast<const _Float16 *>(p.B);
    const int b = lane >> 2, q = lane & 3;
    const int nst = (p.Kp + 127) / 128;
    const long long bplane = (long long)p.Dp * p.Kp;
    if (tid < words) { s_mask[tid] = p.umask[(long long)e * words + tid]; p.umask[(long long)e * words + tid] = 0u; }
    if (tid == 0) p.need[e] = 0;
    __syncthreads();
    if (tid == 0) {
        int n = 0;
        for (int w = 0; w < words; ++w) { s_pre[w] = n; n += __popc(s_mask[w]); }
        s_pre[words] = n;
    }
    __syncthreads();
    const int ncols = s_pre[words];
    for (int c = tid; c < words * 32; c += NW * 64) {
        const unsigned m = s_mask[c >> 5];
        if ((m >> (c & 31)) & 1u) s_cols[s_pre[c >> 5] + __popc(m & ((1u << (c & 31)) - 1u))] = (unsigned short)c;
    }
    __syncthreads();
    // where the unit's rows live (as k_repair_patch): a unit of the previous call's last frames goes into the history's own map
    const bool hist = p.hist_C != nullptr && e >= p.hist_base;
    const int eu = hist ? e - p.hist_base : e, upa = hist ? HIST_UNITS : p.groups_per_array;
    const int arr = eu / upa, f0 = (eu - arr * upa) * REPAIR_GROUP, f_lim = hist ? HIST_FRAMES : p.n_frames;
    float *crow0 = hist ? p.hist_C + ((long long)arr * HIST_FRAMES + f0) * p.Dp : p.C + ((long long)arr * p.n_frames + f0) * p.Dp;
    const _Float16 *pa = A + (long long)(g * REPAIR_GROUP + q) * p.a_row_elems + b * 8;
    for (int c0 = 0; c0 < ncols; c0 += 8) {
        const _Float16 *pb1 = B + (long long)s_cols[min(c0 + q, ncols - 1)] * p.Kp + b * 8;
        const _Float16 *pb2 = B + (long long)s_cols[min(c0 + 4 + q, ncols - 1)] * p.Kp + b * 8;
        cu_f32x4 acc1 = {0.f, 0.f, 0.f, 0.f}, acc2 = {0.f, 0.f, 0.f, 0.f};
        for (int st0 = wave; st0 < nst; st0 += 4 * NW) {
            // four steps of this wave in flight: 24 sixteen-byte loads, then their 48 matrix instructions
            cu_f16x8 ah[4], al[4], b1h[4], b1l[4], b2h[4], b2l[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int k = (st0 + NW * u) * 128;
                const bool on = k + b * 8 < p.Kp;                    // (the depth is a multiple of 32, a step takes 128; st0 + NW u may be past the end)
                const int ks = on ? k : 0;
                ah[u] = *reinterpret_cast<const cu_f16x8 *>(pa + ks); al[u] = *reinterpret_cast<const cu_f16x8 *>(pa + p.Kp + ks);
                b1h[u] = *reinterpret_cast<const cu_f16x8 *>(pb1 + ks); b1l[u] = *reinterpret_cast<const cu_f16x8 *>(pb1 + bplane + ks);
                b2h[u] = *reinterpret_cast<const cu_f16x8 *>(pb2 + ks); b2l[u] = *reinterpret_cast<const cu_f16x8 *>(pb2 + bplane + ks);
                if (!on) {
#pragma unroll
                    for (int x = 0; x < 8; ++x) { ah[u][x] = (_Float16)0.f; al[u][x] = (_Float16)0.f; }
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
                    const cu_f16x4 xh = {ah[u][4 * hh], ah[u][4 * hh + 1], ah[u][4 * hh + 2], ah[u][4 * hh + 3]}, xl = {al[u][4 * hh], al[u][4 * hh + 1], al[u][4 * hh + 2], al[u][4 * hh + 3]};
                    const cu_f16x4 y1h = {b1h[u][4 * hh], b1h[u][4 * hh + 1], b1h[u][4 * hh + 2], b1h[u][4 * hh + 3]}, y1l = {b1l[u][4 * hh], b1l[u][4 * hh + 1], b1l[u][4 * hh + 2], b1l[u][4 * hh + 3]};
                    const cu_f16x4 y2h = {b2h[u][4 * hh], b2h[u][4 * hh + 1], b2h[u][4 * hh + 2], b2h[u][4 * hh + 3]}, y2l = {b2l[u][4 * hh], b2l[u][4 * hh + 1], b2l[u][4 * hh + 2], b2l[u][4 * hh + 3]};
                    // small terms first so they are not absorbed by the large partial sum
                    acc1 = __builtin_amdgcn_mfma_f32_4x4x4f16(xl, y1h, acc1, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_4x4x4f16(xh, y1l, acc1, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_4x4x4f16(xh, y1h, acc1, 0, 0, 0);
                    acc2 = __builtin_amdgcn_mfma_f32_4x4x4f16(xl, y2h, acc2, 0, 0, 0);
                    acc2 = __builtin_amdgcn_mfma_f32_4x4x4f16(xh, y2l, acc2, 0, 0, 0);
                    acc2 = __builtin_amdgcn_mfma_f32_4x4x4f16(xh, y2h, acc2, 0, 0, 0);
                }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) { red[wave][0][i][lane] = acc1[i]; red[wave][1][i][lane] = acc2[i]; }
        __syncthreads();
        // element el = (group, row i, column j) of the 4 x 8 result: NW x 16 partial values (waves x slices), summed in the order
        // (wave, slice) -- eight threads take an eighth each, one adds the eight
        if (tid < 256) {
            const int el = tid >> 3, part = tid & 7, grp = el >> 4, i = (el >> 2) & 3, j = el & 3;
            float v = 0.f;
#pragma unroll
            for (int x = 0; x < 2 * NW; ++x) {
                const int pidx = part * (2 * NW) + x;                 // = 16 wave + slice
                v += red[pidx >> 4][grp][i][4 * (pidx & 15) + j];
            }
            red2[el][part] = v;
        }
        __syncthreads();
        if (tid < 32) {
            const int grp = tid >> 4, i = (tid >> 2) & 3, j = tid & 3, ci = c0 + 4 * grp + j;
            float v = red2[tid][0];
#pragma unroll
            for (int part = 1; part < 8; ++part) v += red2[tid][part];
            if (ci < ncols && f0 + i < f_lim) {
                float *dst = crow0 + (long long)i * p.Dp + s_cols[ci];
                *dst = v;
                if (!hist)
                    for (int pl = 1; pl < p.c_planes; ++pl) dst[pl * p.c_plane_stride] = 0.f;
            }
        }
        __syncthreads();
    }
    __syncthreads();                                            // (s_mask / s_cols are rewritten by the next unit)
}

}  // namespace mca
