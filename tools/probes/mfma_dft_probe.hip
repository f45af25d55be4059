// Probe: a 1024-point complex DFT per wave on the matrix cores of gfx950 -- 1024 = 32 x 32, two DFT-32 stages as
// v_mfma_f32_32x32x16_f16 products (fp16 operands, fp32 accumulation), the accumulator layout of stage 1 being the B-operand
// layout of stage 2 up to a permutation of the contraction index, so that nothing is exchanged between lanes -- against the
// vector-ALU transform of the library (mcarray_amd/csrc/fft1024c.h).  Prints the time per transform and the error of both
// against a double-precision DFT.  Input: n = 32 n1 + n2; output: k = k1 + 32 k2;
//     Z[k1 + 32 k2] = sum_n2 W32^(n2 k2) W1024^(n2 k1) sum_n1 W32^(n1 k1) z[32 n1 + n2].
// Build (from the repository root): hipcc --offload-arch=gfx950 -O3 -std=c++17 -Imcarray_amd/csrc -Iinclude tools/probes/mfma_dft_probe.hip
#include <hip/hip_runtime.h>

#include <cmath>
#include <complex>
#include <cstdio>
#include <vector>

#include "fft1024c.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// the contraction index behind slot s of lane group h in product q: the accumulator rows a lane holds in registers 8 q + s
__host__ __device__ inline int slot_index(int q, int h, int s) { return (s & 3) + 8 * (2 * q + (s >> 2)) + 4 * h; }

// MODE 0: store the spectra (verification); 1: keep a checksum only (timing)
template <int MODE>
__global__ __launch_bounds__(256, 2) void k_mfma_dft(const float2 *in, float2 *out, float *chk, int frames_per_wave)
{
    const int lane = threadIdx.x & 63, wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int c = lane & 31, h = lane >> 5;
    // constants: F[slot_index(q, h, s)][c], F[a][b] = exp(-j 2 pi a b / 32): the B operand of stage 1 and the A operand of stage 2
    f16x8 Fr[2], Fi[2], nFi[2];
    for (int q = 0; q < 2; ++q)
        for (int s = 0; s < 8; ++s) {
            float sn, cs;
            sincospif(-2.0f * (float)((slot_index(q, h, s) * c) & 31) / 32.0f, &sn, &cs);
            Fr[q][s] = (_Float16)cs; Fi[q][s] = (_Float16)sn; nFi[q][s] = (_Float16)(-sn);
        }
    // twiddles W1024^(n2 k1) of the accumulator registers: row n2 = slot_index(r >> 3, h, r & 7), column k1 = c
    float2 tw[16];
    for (int r = 0; r < 16; ++r) {
        float sn, cs;
        sincospif(-2.0f * (float)(slot_index(r >> 3, h, r & 7) * c) / 1024.0f, &sn, &cs);
        tw[r] = make_float2(cs, sn);
    }
    float acc_chk = 0.f;
    for (int f = 0; f < frames_per_wave; ++f) {
        const float2 *z = in + ((long long)wave * frames_per_wave + (MODE == 1 ? (f & 1) : f)) * 1024;     // (timing: two frames per wave, from L2)
        // stage 1 operand: lane (n2 = c, h) holds z[32 n1 + n2], n1 = slot_index(q, h, s)
        f16x8 xr[2], xi[2];
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                const float2 v = z[32 * slot_index(q, h, s) + c];
                xr[q][s] = (_Float16)v.x; xi[q][s] = (_Float16)v.y;
            }
        f32x16 yr = {0}, yi = {0};
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            yr = __builtin_amdgcn_mfma_f32_32x32x16_f16(xr[q], Fr[q], yr, 0, 0, 0);
            yr = __builtin_amdgcn_mfma_f32_32x32x16_f16(xi[q], nFi[q], yr, 0, 0, 0);
            yi = __builtin_amdgcn_mfma_f32_32x32x16_f16(xr[q], Fi[q], yi, 0, 0, 0);
            yi = __builtin_amdgcn_mfma_f32_32x32x16_f16(xi[q], Fr[q], yi, 0, 0, 0);
        }
        // twiddle in fp32, then fp16 operands of stage 2: B[column k1 = c][slot s of product q] = Y[n2 = slot_index(q, h, s)][k1]
        f16x8 br[2], bi[2];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float re = yr[r] * tw[r].x - yi[r] * tw[r].y, im = yr[r] * tw[r].y + yi[r] * tw[r].x;
            br[r >> 3][r & 7] = (_Float16)re; bi[r >> 3][r & 7] = (_Float16)im;
        }
        f32x16 zr = {0}, zi = {0};
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            zr = __builtin_amdgcn_mfma_f32_32x32x16_f16(Fr[q], br[q], zr, 0, 0, 0);
            zr = __builtin_amdgcn_mfma_f32_32x32x16_f16(nFi[q], bi[q], zr, 0, 0, 0);
            zi = __builtin_amdgcn_mfma_f32_32x32x16_f16(Fi[q], br[q], zi, 0, 0, 0);
            zi = __builtin_amdgcn_mfma_f32_32x32x16_f16(Fr[q], bi[q], zi, 0, 0, 0);
        }
        if (MODE == 0) {
            float2 *o = out + ((long long)wave * frames_per_wave + f) * 1024;
#pragma unroll
            for (int r = 0; r < 16; ++r) o[c + 32 * slot_index(r >> 3, h, r & 7)] = make_float2(zr[r], zi[r]);   // Z[k1 + 32 k2]
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) acc_chk += zr[r] * 1e-3f + zi[r] * 1e-4f;
        }
    }
    if (MODE == 1) chk[blockIdx.x * blockDim.x + threadIdx.x] = acc_chk;
}

// the library's vector-ALU transform in the same harness (input and output: index = lane + 64 register, output in its dr16 order)
template <int MODE>
__global__ __launch_bounds__(256, 2) void k_valu_dft(const float2 *in, float2 *out, float *chk, int frames_per_wave)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float2 *tab = reinterpret_cast<float2 *>(smem_raw);
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    float2 *buf = tab + mca::F1K_TWORDS + w * mca::F1K_SCRATCH;
    mca::f1k_table_init(tab, tid, 256);
    mca::F1kLane lc;
    lc.init(lane);
    __syncthreads();
    float acc_chk = 0.f;
    for (int f = 0; f < frames_per_wave; ++f) {
        const float2 *z = in + ((long long)wave * frames_per_wave + (MODE == 1 ? (f & 1) : f)) * 1024;
        float2 v[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = z[lane + 64 * i];
        mca::fft1024c<false, 3>(v, buf, lane, tab, lc, []() {});
        if (MODE == 0) {
            float2 *o = out + ((long long)wave * frames_per_wave + f) * 1024;
#pragma unroll
            for (int i = 0; i < 16; ++i) o[lane + 64 * mca::dr16(i)] = v[i];
        } else {
#pragma unroll
            for (int i = 0; i < 16; ++i) acc_chk += v[i].x * 1e-3f + v[i].y * 1e-4f;
        }
    }
    if (MODE == 1) chk[blockIdx.x * blockDim.x + threadIdx.x] = acc_chk;
}

int main()
{
    const int WGS = 512, FPW = 64;                                   // 2048 waves (two per SIMD) x 64 transforms = 131 072 = the pair transforms of a bench step
    const long long n_tr = (long long)WGS * 4 * FPW;
    std::vector<float2> h((size_t)n_tr * 1024);
    unsigned long long s = 88172645463325252ull;
    for (auto &v : h) {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17; v.x = (float)((double)(s >> 11) / 9007199254740992.0 - 0.5);
        s ^= s << 13; s ^= s >> 7; s ^= s << 17; v.y = (float)((double)(s >> 11) / 9007199254740992.0 - 0.5);
    }
    float2 *d_in, *d_out; float *d_chk;
    (void)hipMalloc(&d_in, h.size() * 8); (void)hipMalloc(&d_out, h.size() * 8); (void)hipMalloc(&d_chk, (size_t)WGS * 256 * 4);
    (void)hipMemcpy(d_in, h.data(), h.size() * 8, hipMemcpyHostToDevice);
    const size_t smem = (size_t)(mca::F1K_TWORDS + 4 * mca::F1K_SCRATCH) * sizeof(float2);
    // accuracy on the first transforms
    std::vector<float2> o(8 * 1024);
    for (int which = 0; which < 2; ++which) {
        if (which == 0) hipLaunchKernelGGL(k_mfma_dft<0>, dim3(WGS), dim3(256), 0, 0, d_in, d_out, d_chk, FPW);
        else hipLaunchKernelGGL(k_valu_dft<0>, dim3(WGS), dim3(256), smem, 0, d_in, d_out, d_chk, FPW);
        (void)hipDeviceSynchronize();
        (void)hipMemcpy(o.data(), d_out, o.size() * 8, hipMemcpyDeviceToHost);
        double err2 = 0, ref2 = 0, emax = 0;
        for (int t = 0; t < 8; ++t)
            for (int k = 0; k < 1024; k += 7) {
                std::complex<double> a(0, 0);
                for (int n = 0; n < 1024; ++n)
                    a += std::complex<double>(h[(size_t)t * 1024 + n].x, h[(size_t)t * 1024 + n].y) * std::polar(1.0, -2.0 * M_PI * (double)((n * k) & 1023) / 1024.0);
                const std::complex<double> g(o[(size_t)t * 1024 + k].x, o[(size_t)t * 1024 + k].y);
                err2 += std::norm(g - a); ref2 += std::norm(a); emax = std::fmax(emax, std::abs(g - a));
            }
        printf("%s: rms error / rms spectrum = %.2e, largest error / rms spectrum = %.2e\n", which == 0 ? "MFMA fp16 DFT-32 x DFT-32" : "vector-ALU fp32 (fft1024c)  ",
               std::sqrt(err2 / ref2), emax / std::sqrt(ref2 / (8 * 147)));
    }
    // time (inputs from L2, no stores)
    for (int which = 0; which < 2; ++which) {
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        float best = 1e9f;
        for (int rep = 0; rep < 5; ++rep) {
            (void)hipEventRecord(e0, 0);
            if (which == 0) hipLaunchKernelGGL(k_mfma_dft<1>, dim3(WGS), dim3(256), 0, 0, d_in, d_out, d_chk, FPW);
            else hipLaunchKernelGGL(k_valu_dft<1>, dim3(WGS), dim3(256), smem, 0, d_in, d_out, d_chk, FPW);
            (void)hipEventRecord(e1, 0);
            (void)hipDeviceSynchronize();
            float ms = 0.f;
            (void)hipEventElapsedTime(&ms, e0, e1);
            best = std::fmin(best, ms);
        }
        printf("%s: %.3f ms per %lld transforms (load 8 KB each from L2 + transform, 2 waves per SIMD) = %.1f ns per transform per SIMD\n",
               which == 0 ? "MFMA fp16 DFT-32 x DFT-32" : "vector-ALU fp32 (fft1024c)  ", best, n_tr, best * 1e6 / ((double)n_tr / 1024.0));
    }
    return 0;
}
