// phat_pairs.h -- PHAT whitening and the per-bin pair products of the GCC-PHAT stage (SteeringBeamforming.cpp:104-130 up to
// the steering sum), shared by the analysis kernels of kernels_stream.hip and kernels_wave.hip.
#pragma once
#include "fft512.h"
#include "mca_internal.h"

namespace mca {

__device__ __forceinline__ float2 whiten(float2 z)
{
    const float pw = z.x * z.x + z.y * z.y;
    const float s = pw > 1e-30f ? rsqrtf(pw) : 0.f;
    return make_float2(z.x * s, z.y * s);
}

// The pair products of one bin from whitened spectra in registers: out[g] = sum over the pairs of delay group g (ULA: pairs
// with equal j - i, MT - 1 groups) or one product per pair (i < j, lexicographic).
template <int MT, bool ULA>
struct PairOut { static constexpr int N = ULA ? MT - 1 : MT * (MT - 1) / 2; };
template <int MT, bool ULA>
__device__ __forceinline__ void pair_products(const float2 (&r)[MT], float2 (&out)[PairOut<MT, ULA>::N])
{
    if constexpr (ULA) {
#pragma unroll
        for (int g = 0; g < MT - 1; ++g) out[g] = make_float2(0.f, 0.f);
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = i + 1; j < MT; ++j) out[j - i - 1] = cmacc(out[j - i - 1], r[i], r[j]);
    } else {
        int pi = 0;
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = i + 1; j < MT; ++j) { out[pi] = cmulc(r[i], r[j]); ++pi; }
    }
}

// pair products of one bin.  x: whitened spectra of the bin, element m at x[m * xstride].
template <int MT, bool ULA, bool WHITEN, typename OutT>
__device__ __forceinline__ void pair_stage(const float2 *x, int xstride, int M, OutT *arow, const StftPhatArgs &p, int k, int kg = KG)
{
    if constexpr (MT > 0) {
        float2 r[MT];
#pragma unroll
        for (int m = 0; m < MT; ++m) r[m] = WHITEN ? whiten(x[m * xstride]) : x[m * xstride];
        if constexpr (ULA) {
            float2 acc[MT - 1];
#pragma unroll
            for (int g = 0; g < MT - 1; ++g) acc[g] = make_float2(0.f, 0.f);
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = i + 1; j < MT; ++j) acc[j - i - 1] = cmacc(acc[j - i - 1], r[i], r[j]);
#pragma unroll
            for (int g = 0; g < MT - 1; ++g) store_a(arow, p, g * kg + k, acc[g]);
        } else {
            int pi = 0;
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = i + 1; j < MT; ++j) { store_a(arow, p, pi * kg + k, cmulc(r[i], r[j])); ++pi; }
        }
    } else {
        if (ULA) {
            for (int g = 0; g < M - 1; ++g) {
                float2 acc = make_float2(0.f, 0.f);
                for (int i = 0; i + g + 1 < M; ++i) acc = cmacc(acc, x[i * xstride], x[(i + g + 1) * xstride]);
                store_a(arow, p, g * kg + k, acc);
            }
        } else {
            int pi = 0;
            for (int i = 0; i < M; ++i)
                for (int j = i + 1; j < M; ++j) { store_a(arow, p, pi * kg + k, cmulc(x[i * xstride], x[j * xstride])); ++pi; }
        }
    }
}

}  // namespace mca
