// round 6: what v_mfma_f32_4x4x1_16B_f32 could buy k_mvdr_solve.
//   (1) layout: lane l = 4 b + q: A holds a[b][row q], B holds b[b][col q]; D register i of lane l = a[b][i] * b[b][q] (+ C) -- the 16 blocks are
//       the 16 (stream, bin) problems of a wave in k_mvdr_solve's four-lanes-per-problem layout, a block = one 4 x 4 outer product per problem
//   (2) rate: wave-instructions per SIMD cycle of (a) the MFMA alone, (b) v_pk_fma_f32 alone, (c) both interleaved in ONE wave, with two waves
//       per SIMD resident as in the solve kernel -- does the matrix pipe run beside the vector ALU, and at what cost per instruction
// build: hipcc --offload-arch=gfx950 -O2 -o /tmp/mfma4x4x1_probe tools/probes/mfma4x4x1_f32_probe.hip && /tmp/mfma4x4x1_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

__global__ void k_layout(const float *A, const float *B, float *D)
{
    const int l = threadIdx.x;
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    acc = __builtin_amdgcn_mfma_f32_4x4x1f32(A[l], B[l], acc, 0, 0, 0);
    for (int i = 0; i < 4; ++i) D[l * 4 + i] = acc[i];
}

template <int MODE>      // 0: MFMA only, 1: packed fma only, 2: both interleaved
__global__ __launch_bounds__(256, 2) void k_rate(float *out, int iters, float seed)
{
    f4 acc[8];
    f2 v[16];
    for (int i = 0; i < 8; ++i) acc[i] = f4{seed, seed, seed, seed};
    for (int i = 0; i < 16; ++i) v[i] = f2{seed + i, seed - i};
    const float a = seed * 0.5f + threadIdx.x, b = seed * 0.25f;
    const f2 m = {1.0001f, 0.9999f}, c = {1e-3f, -1e-3f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (MODE != 1) acc[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[i], 0, 0, 0);
            if (MODE != 0) {
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(v[2 * i]) : "v"(m), "v"(c));
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(v[2 * i + 1]) : "v"(m), "v"(c));
            }
        }
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    for (int i = 0; i < 16; ++i) s += v[i][0] + v[i][1];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
static double run(float *d_out, int iters)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k_rate<MODE>, dim3(512), dim3(256), 0, 0, d_out, 16, 1.0f);      // warm-up
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_rate<MODE>, dim3(512), dim3(256), 0, 0, d_out, iters, 1.0f);   // 512 x 4 waves = 2 waves per SIMD on 256 CUs
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main()
{
    float hA[64], hB[64], hD[256];
    for (int i = 0; i < 64; ++i) { hA[i] = (float)((i * 7 % 13) - 6); hB[i] = (float)((i * 5 % 11) - 5); }
    float *dA, *dB, *dD, *d_out;
    hipMalloc(&dA, 256); hipMalloc(&dB, 256); hipMalloc(&dD, 1024); hipMalloc(&d_out, 512 * 256 * 4);
    hipMemcpy(dA, hA, 256, hipMemcpyHostToDevice); hipMemcpy(dB, hB, 256, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_layout, dim3(1), dim3(64), 0, 0, dA, dB, dD);
    hipMemcpy(hD, dD, 1024, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l) for (int i = 0; i < 4; ++i) bad += hD[l * 4 + i] != hA[(l & ~3) + i] * hB[l];
    printf("mfma 4x4x1 f32 layout (lane 4 b + q: A = row q, B = column q; D register i = row i, column q of block b): %s (%d of 256 differ)\n", bad ? "WRONG" : "as assumed", bad);
    const int iters = 20000;
    const double t0 = run<0>(d_out, iters), t1 = run<1>(d_out, iters), t2 = run<2>(d_out, iters);
    int clk_khz = 0;
    hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeClockRate, 0);
    const double ghz = clk_khz / 1e6;
    // per SIMD: 2 waves x iters x 8 MFMA (and / or 16 packed fma)
    printf("clock (attribute) %.2f GHz; per SIMD, two waves resident:\n", ghz);
    printf("  8 MFMA 4x4x1 per iteration alone:            %.3f ms = %.2f cycles per MFMA\n", t0, t0 * 1e-3 * ghz * 1e9 / (2.0 * iters * 8));
    printf("  16 v_pk_fma_f32 per iteration alone:         %.3f ms = %.2f cycles per packed fma\n", t1, t1 * 1e-3 * ghz * 1e9 / (2.0 * iters * 16));
    printf("  8 MFMA + 16 v_pk_fma_f32 interleaved:        %.3f ms = %.2f of the sum of the two alone (1.0: no overlap, max/sum = %.2f: full overlap)\n", t2, t2 / (t0 + t1),
           fmax(t0, t1) / (t0 + t1));
    return bad != 0;
}
