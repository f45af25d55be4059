// kernels_frame.hip -- single-frame kernels behind the frame-level module API
// (mca::SteeringBeamforming::processFrame, mca::Beamformer::processFrame).  These are the
// drop-in replacements for the reference's per-frame calls, so they run in the reference's own
// arithmetic type (double) -- MI355X has a full-rate fp64 vector pipe, and a single frame is
// latency-bound anyway.  Templated so the same code also exists in fp32.
#include "mca_internal.h"

namespace mca {

template <typename T> struct C2 { T x, y; };

template <typename T> __device__ __forceinline__ void sincos2pi(double turns, T *s, T *c);
template <> __device__ __forceinline__ void sincos2pi<double>(double turns, double *s, double *c) { sincospi(2.0 * turns, s, c); }
template <> __device__ __forceinline__ void sincos2pi<float>(double turns, float *s, float *c) { sincospif(2.0f * (float)turns, s, c); }

template <typename T>
__device__ __forceinline__ T block_sum(T v, T *sred)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    __syncthreads();
    if (lane == 0) sred[wave] = v;
    __syncthreads();
    T r = sred[0];
    for (int w = 1; w < nw; ++w) r += sred[w];
    return r;
}

// computeCorrelations + computeEnergyInDOA (SteeringBeamforming.cpp:104-144) for one frame.
// grid = D blocks (one steering angle each), 256 threads over the K bins.
// X: [M][K] complex T (CCS order), delays: [P][D] float samples, E_in/E_out: [D].
template <typename T>
__global__ __launch_bounds__(256) void k_frame_srp(const C2<T> *X, int K, int D, int P, const int2 *pairs,
                                                   const float *delays, const T *E_in, T *E_out, T mu, T omu, int no_phat)
{
    __shared__ T sred[4];
    const int d = blockIdx.x, tid = threadIdx.x;
    const double N = 2.0 * (double)(K - 1);
    T e = mu * E_in[d];                                                   // :134
    for (int p = 0; p < P; ++p) {
        const C2<T> *A = X + (long long)pairs[p].x * K;
        const C2<T> *B = X + (long long)pairs[p].y * K;
        const double tau = (double)delays[(long long)p * D + d];
        T part = 0;
        for (int k = tid; k < K; k += 256) {
            C2<T> a = A[k], b = B[k];
            T gr = a.x * b.x + a.y * b.y, gi = a.y * b.x - a.x * b.y;     // A conj(B)
            T mag = sqrt(gr * gr + gi * gi);
            T inv = no_phat ? (T)1 : (mag > (T)1e-30 ? (T)1 / mag : (T)0);     // gcc_weighting: PHAT / NONE
            double turns = (double)k * tau / N;
            turns -= rint(turns);
            T sn, cs;
            sincos2pi<T>(turns, &sn, &cs);
            part += (gr * cs - gi * sn) * inv;                           // Re(Ghat * exp(+j 2 pi k tau / N))
        }
        T rp = block_sum<T>(part, sred);
        e += omu * rp;                                                    // :139-140, pair order p = 0..P-1
    }
    if (tid == 0) E_out[d] = e;                                           // :143
}

template <typename T> __device__ __forceinline__ T med3(T a, T b, T c)
{
    T lo = a < b ? a : b, hi = a < b ? b : a;
    return c < lo ? lo : (c > hi ? hi : c);
}

// selectDOA (SteeringBeamforming.cpp:146-195), one block of 512 threads, D <= 512.
template <typename T>
__global__ __launch_bounds__(512) void k_frame_pick(const T *E, int D, int P, int S, const float *grid,
                                                    T *doa, T *prob, int *bins)
{
    __shared__ T sEn[520], sFd[520], sFm[520], sV[8];
    __shared__ int sI[8];
    const int d = threadIdx.x, lane = d & 63, wave = d >> 6, nw = blockDim.x >> 6;
    const T mn = (T)(-15 * P);
    if (d < D) sEn[d] = (E[d] - mn) / ((T)-2 * mn);
    __syncthreads();
    if (d < D - 1) { T df = sEn[d + 1] - sEn[d]; sFd[d] = df < (T)0 ? (T)1 : (T)0; }
    __syncthreads();
    if (d < D - 1) sFm[d] = med3<T>(sFd[d > 0 ? d - 1 : 0], sFd[d], sFd[d + 1 < D - 1 ? d + 1 : D - 2]);
    __syncthreads();
    T sd = -INFINITY;
    if (d < D - 2) sd = (sFm[d + 1] - sFm[d]) * sEn[d + 1];
    for (int s = 0; s < S; ++s) {
        T bv = sd; int bi = d;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            T ov = __shfl_down(bv, off); int oi = __shfl_down(bi, off);
            if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
        }
        if (lane == 0) { sV[wave] = bv; sI[wave] = bi; }
        __syncthreads();
        bv = sV[0]; bi = sI[0];
        for (int w = 1; w < nw; ++w) { T ov = sV[w]; int oi = sI[w]; if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; } }
        if (d == bi) sd = (T)0;
        if (d == 0) { doa[s] = (T)grid[bi + 1]; prob[s] = bv; bins[s] = bi + 1; }
        __syncthreads();
    }
}

// Beamformer::processFrame (Beamformer.cpp:51-71): Y[k] = (1/M) sum_c X_c[k] exp(j k s_c)
template <typename T>
__global__ __launch_bounds__(256) void k_frame_beamform(const C2<T> *X, int M, int K, int fs, const double *mic_x,
                                                        double doa, C2<T> *Y)
{
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= K) return;
    const double cd = cos(doa + 1.57079632679489661923);
    const double unit = (double)fs / (2.0 * (double)(K - 1)) / 346.1;     // slope / (2 pi), :59
    T ar = 0, ai = 0;
    for (int c = 0; c < M; ++c) {
        double turns = (double)k * (unit * mic_x[c] * cd);
        turns -= rint(turns);
        T sn, cs;
        sincos2pi<T>(turns, &sn, &cs);
        C2<T> x = X[(long long)c * K + k];
        ar += x.x * cs - x.y * sn;
        ai += x.x * sn + x.y * cs;
    }
    Y[k].x = ar / (T)M;                                                   // divC :70
    Y[k].y = ai / (T)M;
}

// dsp::SignalPower::FFTPower [INFERRED, SURVEY A.8]: mean over channels of (1/N^2) sum_k w_k |X[k]|^2
template <typename T>
__global__ __launch_bounds__(256) void k_frame_power(const C2<T> *X, int M, int K, T *out)
{
    __shared__ T sred[4];
    T part = 0;
    for (int i = threadIdx.x; i < M * K; i += 256) {
        const int k = i % K;
        C2<T> x = X[i];
        T w = (k == 0 || k == K - 1) ? (T)1 : (T)2;
        part += w * (x.x * x.x + x.y * x.y);
    }
    T tot = block_sum<T>(part, sred);
    const T N = (T)(2 * (K - 1));
    if (threadIdx.x == 0) out[0] = tot / (N * N) / (T)M;
}

#define INST_FRAME(T)                                                                                            \
    template __global__ void k_frame_srp<T>(const C2<T> *, int, int, int, const int2 *, const float *, const T *, T *, T, T, int); \
    template __global__ void k_frame_pick<T>(const T *, int, int, int, const float *, T *, T *, int *);            \
    template __global__ void k_frame_beamform<T>(const C2<T> *, int, int, int, const double *, double, C2<T> *);     \
    template __global__ void k_frame_power<T>(const C2<T> *, int, int, T *);
INST_FRAME(double)
INST_FRAME(float)

}  // namespace mca
