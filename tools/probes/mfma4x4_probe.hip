// Probe: issue cost of v_mfma_f32_4x4x1_16B_f32 against v_pk_fma_f32 / DPP moves on gfx950, alone and mixed, with one
// and two waves per SIMD.  Prints shader cycles per instruction (s_memtime).  Build: hipcc --offload-arch=gfx950 -O2
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

#define REP 64
template <int MODE>
__global__ void k(float *out, long long *cyc, int iters)
{
    f4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0, c4 = c0, c5 = c0, c6 = c0, c7 = c0;
    f2 p0 = {1, 1}, p1 = p0, p2 = p0, p3 = p0, p4 = p0, p5 = p0, p6 = p0, p7 = p0;
    float a = threadIdx.x * 1e-3f, b = 1.0001f;
    f2 pa = {a, b};
    float m0 = a, m1 = b;
    __syncthreads();
    const long long w0 = (long long)wall_clock64();
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < REP / 8; ++r) {
            if (MODE == 0) {        // 8 independent MFMA chains
                asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %8, %9, %0\n v_mfma_f32_4x4x1_16b_f32 %1, %8, %9, %1\n v_mfma_f32_4x4x1_16b_f32 %2, %8, %9, %2\n v_mfma_f32_4x4x1_16b_f32 %3, %8, %9, %3\n"
                             "v_mfma_f32_4x4x1_16b_f32 %4, %8, %9, %4\n v_mfma_f32_4x4x1_16b_f32 %5, %8, %9, %5\n v_mfma_f32_4x4x1_16b_f32 %6, %8, %9, %6\n v_mfma_f32_4x4x1_16b_f32 %7, %8, %9, %7\n"
                             : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7) : "v"(a), "v"(b));
            } else if (MODE == 1) { // one dependent MFMA chain (compiler-visible so that it adds the required nops)
#pragma unroll
                for (int q = 0; q < 8; ++q) c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c0, 0, 0, 0);
            } else if (MODE == 2) { // 8 independent pk_fma chains
                asm volatile("v_pk_fma_f32 %0, %8, %8, %0\n v_pk_fma_f32 %1, %8, %8, %1\n v_pk_fma_f32 %2, %8, %8, %2\n v_pk_fma_f32 %3, %8, %8, %3\n"
                             "v_pk_fma_f32 %4, %8, %8, %4\n v_pk_fma_f32 %5, %8, %8, %5\n v_pk_fma_f32 %6, %8, %8, %6\n v_pk_fma_f32 %7, %8, %8, %7\n"
                             : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pa));
            } else if (MODE == 3) { // 4 MFMA + 8 pk_fma interleaved, all independent
                asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %12, %13, %0\n v_pk_fma_f32 %4, %14, %14, %4\n v_pk_fma_f32 %5, %14, %14, %5\n"
                             "v_mfma_f32_4x4x1_16b_f32 %1, %12, %13, %1\n v_pk_fma_f32 %6, %14, %14, %6\n v_pk_fma_f32 %7, %14, %14, %7\n"
                             "v_mfma_f32_4x4x1_16b_f32 %2, %12, %13, %2\n v_pk_fma_f32 %8, %14, %14, %8\n v_pk_fma_f32 %9, %14, %14, %9\n"
                             "v_mfma_f32_4x4x1_16b_f32 %3, %12, %13, %3\n v_pk_fma_f32 %10, %14, %14, %10\n v_pk_fma_f32 %11, %14, %14, %11\n"
                             : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7)
                             : "v"(a), "v"(b), "v"(pa));
            } else if (MODE == 4) { // 8 plain v_fma_f32, independent
                asm volatile("v_fma_f32 %0, %8, %8, %0\n v_fma_f32 %1, %8, %8, %1\n v_fma_f32 %2, %8, %8, %2\n v_fma_f32 %3, %8, %8, %3\n"
                             "v_fma_f32 %4, %8, %8, %4\n v_fma_f32 %5, %8, %8, %5\n v_fma_f32 %6, %8, %8, %6\n v_fma_f32 %7, %8, %8, %7\n"
                             : "+v"(p0.x), "+v"(p1.x), "+v"(p2.x), "+v"(p3.x), "+v"(p4.x), "+v"(p5.x), "+v"(p6.x), "+v"(p7.x) : "v"(a));
            } else if (MODE == 5) { // 8 DPP quad broadcasts, independent
                asm volatile("v_mov_b32_dpp %0, %8 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp %1, %8 quad_perm:[2,2,2,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                             "v_mov_b32_dpp %2, %8 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp %3, %8 quad_perm:[2,2,2,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                             "v_mov_b32_dpp %4, %8 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp %5, %8 quad_perm:[2,2,2,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                             "v_mov_b32_dpp %6, %8 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp %7, %8 quad_perm:[2,2,2,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                             : "+v"(p0.x), "+v"(p1.x), "+v"(p2.x), "+v"(p3.x), "+v"(p4.x), "+v"(p5.x), "+v"(p6.x), "+v"(p7.x) : "v"(a));
            } else if (MODE == 6) { // dependent pair chain: pk_fma -> pk_fma on the same register (the cmacc pattern), 4 chains
                asm volatile("v_pk_fma_f32 %0, %4, %4, %0\n v_pk_fma_f32 %1, %4, %4, %1\n v_pk_fma_f32 %2, %4, %4, %2\n v_pk_fma_f32 %3, %4, %4, %3\n"
                             "v_pk_fma_f32 %0, %4, %4, %0\n v_pk_fma_f32 %1, %4, %4, %1\n v_pk_fma_f32 %2, %4, %4, %2\n v_pk_fma_f32 %3, %4, %4, %3\n"
                             : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pa));
            } else if (MODE == 7) { // fully dependent pk_fma chain
                asm volatile("v_pk_fma_f32 %0, %1, %1, %0\n v_pk_fma_f32 %0, %1, %1, %0\n v_pk_fma_f32 %0, %1, %1, %0\n v_pk_fma_f32 %0, %1, %1, %0\n"
                             "v_pk_fma_f32 %0, %1, %1, %0\n v_pk_fma_f32 %0, %1, %1, %0\n v_pk_fma_f32 %0, %1, %1, %0\n v_pk_fma_f32 %0, %1, %1, %0\n"
                             : "+v"(p0) : "v"(pa));
            } else if (MODE == 8) { // DPP mov reading the result of the previous VALU op, then feeding a pk_fma (the broadcast pattern)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    m0 = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(p0.x), 0x55, 0xf, 0xf, true));
                    m1 = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(p0.y), 0x55, 0xf, 0xf, true));
                    f2 mm = {m0, m1};
                    asm volatile("v_pk_fma_f32 %0, %1, %1, %0" : "+v"(p0) : "v"(mm));
                }
            }
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0 && blockIdx.x == 0) { cyc[0] = t1 - t0; cyc[1] = (long long)wall_clock64() - w0; }
    f4 s = c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7;
    f2 ps = p0 + p1 + p2 + p3 + p4 + p5 + p6 + p7;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s[0] + s[1] + s[2] + s[3] + ps.x + ps.y + m0 + m1;
}

template <int MODE>
void run(const char *name, int per_rep_instr, float *out, long long *cyc)
{
    const int iters = 2000;
    for (int threads = 256; threads <= 1024; threads *= 2) {
        hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, out, cyc, iters);
        hipDeviceSynchronize();
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, out, cyc, iters);
        hipEventRecord(e1, 0);
        hipDeviceSynchronize();
        float ms = 0.f;
        hipEventElapsedTime(&ms, e0, e1);
        long long h[2];
        hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
        const double n = (double)iters * (REP / 8) * per_rep_instr;
        printf("%-46s waves/SIMD %d: %.2f ticks per instr per wave; wall %.1f ns per instr per SIMD (kernel %.3f ms); tick = %.2f ns\n", name, threads / 256,
               h[0] / n, ms * 1e6 / (n * (threads / 256)), ms, h[1] * 10.0 / h[0]);
    }
}

int main()
{
    float *out; long long *cyc;
    hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&cyc, 16 * 8);
    run<0>("mfma 4x4x1 f32, 8 independent chains", 8, out, cyc);
    run<1>("mfma 4x4x1 f32, one dependent chain", 8, out, cyc);
    run<2>("v_pk_fma_f32, 8 independent chains", 8, out, cyc);
    run<4>("v_fma_f32, 8 independent chains", 8, out, cyc);
    run<5>("v_mov_b32_dpp quad broadcast, independent", 8, out, cyc);
    run<6>("v_pk_fma_f32, 4 chains of dependent ops", 8, out, cyc);
    run<7>("v_pk_fma_f32, one dependent chain", 8, out, cyc);
    run<3>("4 mfma + 8 v_pk_fma_f32 interleaved (12 instr)", 12, out, cyc);
    run<8>("2 dpp of a fresh result + dependent pk_fma (3 instr)", 12, out, cyc);
    return 0;
}
