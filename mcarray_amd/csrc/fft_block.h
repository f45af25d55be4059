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
// DIT: input in bit-reversed order, output natural.  Two radix-2 levels are fused per pass (four points in
// registers: half the LDS traffic and half the barriers); an odd log2 H ends with one plain level.
// Every pass ends with a barrier.
__device__ __forceinline__ void block_fft_dit(float2 *z, int zs, int nch, int logH, const float2 *tw, int N, int tid, int nthr)
{
    int s = 0;
    for (; s + 1 < logH; s += 2) {
        const int half = 1 << s, quarterH = 1 << (logH - 2);
        for (int e = tid; e < nch * quarterH; e += nthr) {
            const int ch = e >> (logH - 2), j = e & (quarterH - 1);
            const int pos = j & (half - 1);
            const int i0 = ((j >> s) << (s + 2)) + pos;                   // groups of 4 * half points
            float2 *zz = z + ch * zs;
            const float2 w1 = tw[pos * (N >> (s + 1))];                   // level s:   exp(-j 2 pi pos / (2 half))
            const float2 w2 = tw[pos * (N >> (s + 2))];                   // level s+1: exp(-j 2 pi pos / (4 half))
            const float2 w3 = tw[(pos + half) * (N >> (s + 2))];          //            exp(-j 2 pi (pos + half) / (4 half))
            const float2 a = zz[i0], b = cmul(zz[i0 + half], w1), c = zz[i0 + 2 * half], d = cmul(zz[i0 + 3 * half], w1);
            const float2 p0 = cadd(a, b), p1 = csub(a, b), q0 = cmul(cadd(c, d), w2), q1 = cmul(csub(c, d), w3);
            zz[i0] = cadd(p0, q0); zz[i0 + 2 * half] = csub(p0, q0);
            zz[i0 + half] = cadd(p1, q1); zz[i0 + 3 * half] = csub(p1, q1);
        }
        __syncthreads();
    }
    if (s < logH) {
        const int half = 1 << s, halfH = 1 << (logH - 1);
        for (int e = tid; e < nch * halfH; e += nthr) {
            const int ch = e >> (logH - 1), j = e & (halfH - 1);
            const int pos = j & (half - 1);
            const int i0 = ((j >> s) << (s + 1)) + pos, i1 = i0 + half;
            const float2 w = tw[pos * (N >> (s + 1))];
            float2 *zz = z + ch * zs;
            const float2 a = zz[i0], b = cmul(zz[i1], w);
            zz[i0] = cadd(a, b); zz[i1] = csub(a, b);
        }
        __syncthreads();
    }
}

// inverse, DIF: natural input, bit-reversed output, unnormalised; two levels fused per pass like the forward.
__device__ __forceinline__ void block_ifft_dif(float2 *z, int zs, int nch, int logH, const float2 *tw, int N, int tid, int nthr)
{
    int s = logH - 1;
    for (; s >= 1; s -= 2) {
        const int half = 1 << (s - 1), quarterH = 1 << (logH - 2);       // levels s (span 2^s) then s-1 (span 2^(s-1))
        for (int e = tid; e < nch * quarterH; e += nthr) {
            const int ch = e >> (logH - 2), j = e & (quarterH - 1);
            const int pos = j & (half - 1);
            const int i0 = ((j >> (s - 1)) << (s + 1)) + pos;
            float2 *zz = z + ch * zs;
            const float2 w2 = cconj(tw[pos * (N >> (s + 1))]);            // level s:   exp(+j 2 pi pos / (4 half))
            const float2 w3 = cconj(tw[(pos + half) * (N >> (s + 1))]);   //            exp(+j 2 pi (pos + half) / (4 half))
            const float2 w1 = cconj(tw[pos * (N >> s)]);                  // level s-1: exp(+j 2 pi pos / (2 half))
            const float2 a = zz[i0], b = zz[i0 + half], c = zz[i0 + 2 * half], d = zz[i0 + 3 * half];
            const float2 p0 = cadd(a, c), q0 = cmul(csub(a, c), w2), p1 = cadd(b, d), q1 = cmul(csub(b, d), w3);
            zz[i0] = cadd(p0, p1); zz[i0 + half] = cmul(csub(p0, p1), w1);
            zz[i0 + 2 * half] = cadd(q0, q1); zz[i0 + 3 * half] = cmul(csub(q0, q1), w1);
        }
        __syncthreads();
    }
    if (s == 0) {
        const int halfH = 1 << (logH - 1);
        for (int e = tid; e < nch * halfH; e += nthr) {
            const int ch = e >> (logH - 1), j = e & (halfH - 1);
            float2 *zz = z + ch * zs;
            const float2 a = zz[2 * j], b = zz[2 * j + 1];                // level 0: span 1, twiddle 1
            zz[2 * j] = cadd(a, b); zz[2 * j + 1] = csub(a, b);
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
