// fft_block.h -- block-cooperative power-of-two real FFT in LDS (any H = N/2 = 2^logH), used by the
// any-length stream kernels (kernels_generic.hip) and the multiband localiser (kernels_multiband.hip).
// All threads of the workgroup work on all channels; one barrier per radix-2 stage; twiddles
// exp(-j 2 pi i / N), i < N/2, from a table in global memory.  Conventions as in fft512.h / SURVEY A.1.
#pragma once
#include "fft512.h"

namespace mca {

__device__ __forceinline__ float2 whiten_g(float2 z)
{
    const float pw = z.x * z.x + z.y * z.y;
    const float s = pw > 1e-30f ? rsqrtf(pw) : 0.f;
    return make_float2(z.x * s, z.y * s);
}

// nch interleaved H-point complex transforms in LDS, channel c at z + c * zs.
// DIT: input in bit-reversed order, output natural.  Ends with a barrier.
__device__ __forceinline__ void block_fft_dit(float2 *z, int zs, int nch, int logH, const float2 *tw, int N, int tid, int nthr)
{
    const int halfH = 1 << (logH - 1);
    for (int s = 0; s < logH; ++s) {
        const int half = 1 << s;
        for (int e = tid; e < nch * halfH; e += nthr) {
            const int ch = e >> (logH - 1), j = e & (halfH - 1);
            const int pos = j & (half - 1);
            const int i0 = ((j >> s) << (s + 1)) + pos, i1 = i0 + half;
            const float2 w = tw[pos * (N >> (s + 1))];               // exp(-j 2 pi pos / (2 half))
            float2 *zz = z + ch * zs;
            const float2 a = zz[i0], b = cmul(zz[i1], w);
            zz[i0] = cadd(a, b); zz[i1] = csub(a, b);
        }
        __syncthreads();
    }
}

// inverse, DIF: natural input, bit-reversed output, unnormalised.  Ends with a barrier.
__device__ __forceinline__ void block_ifft_dif(float2 *z, int zs, int nch, int logH, const float2 *tw, int N, int tid, int nthr)
{
    const int halfH = 1 << (logH - 1);
    for (int s = logH - 1; s >= 0; --s) {
        const int half = 1 << s;
        for (int e = tid; e < nch * halfH; e += nthr) {
            const int ch = e >> (logH - 1), j = e & (halfH - 1);
            const int pos = j & (half - 1);
            const int i0 = ((j >> s) << (s + 1)) + pos, i1 = i0 + half;
            const float2 w = cconj(tw[pos * (N >> (s + 1))]);
            float2 *zz = z + ch * zs;
            const float2 a = zz[i0], b = zz[i1];
            zz[i0] = cadd(a, b); zz[i1] = cmul(csub(a, b), w);
        }
        __syncthreads();
    }
}

// windowed frame t of nch channels -> packed z[n] = x[2n] + j x[2n+1] at bit-reversed n
__device__ __forceinline__ void load_frames(float2 *z, int zs, int nch, int logH, const float *base, long long mic_stride,
                                            long long t, const float *window, int tid, int nthr)
{
    const int H = 1 << logH;
    for (int e = tid; e < nch * H; e += nthr) {
        const int ch = e >> logH, n = e & (H - 1);
        const float2 x = reinterpret_cast<const float2 *>(base + (long long)ch * mic_stride + t * H)[n];
        const float2 w = reinterpret_cast<const float2 *>(window)[n];
        z[ch * zs + (int)(__brev((unsigned)n) >> (32 - logH))] = make_float2(x.x * w.x, x.y * w.y);
    }
    __syncthreads();
}

// Z (H-point transform of the packed sequence) -> one-sided spectrum X[0..H], in place.  Ends with a barrier.
__device__ __forceinline__ void split_forward(float2 *z, int zs, int nch, int logH, const float2 *tw, int tid, int nthr)
{
    const int H = 1 << logH, per = H / 2 + 1;
    for (int e = tid; e < nch * per; e += nthr) {
        const int ch = e / per, k = e - ch * per;
        float2 *zz = z + ch * zs;
        if (k == 0) {
            const float2 z0 = zz[0];
            zz[0] = make_float2(z0.x + z0.y, 0.f);
            zz[H] = make_float2(z0.x - z0.y, 0.f);
        } else {
            const float2 zk = zz[k], zp = zz[H - k];
            const float2 ev = make_float2(0.5f * (zk.x + zp.x), 0.5f * (zk.y - zp.y));     // (Zk + conj Zp) / 2
            const float2 od = make_float2(0.5f * (zk.y + zp.y), -0.5f * (zk.x - zp.x));    // (Zk - conj Zp) / 2j
            const float2 wo = cmul(od, tw[k]);                                              // W_N^k O[k]
            zz[k] = cadd(ev, wo);
            zz[H - k] = cconj(csub(ev, wo));
        }
    }
    __syncthreads();
}

}  // namespace mca
