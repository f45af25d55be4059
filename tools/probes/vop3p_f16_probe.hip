// Probe: issue cost on gfx950 of the packed-fp16 / dot2 instructions a half-precision transform would be made of, next to the
// packed-fp32 ones the fp32 transform uses (8 independent chains per wave, 1 / 2 / 4 waves per SIMD), and the operand-modifier
// semantics of v_dot2_f32_f16 / v_pk_add_f16 / v_pk_fma_f16.  Build: hipcc --offload-arch=gfx950 -O2
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));

#define R8(S) S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7)
#define OUT8 "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7)
#define OUT8P "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7)

template <int MODE>
__global__ void k(float *out, long long *cyc, int iters)
{
    unsigned r0 = threadIdx.x * 3 + 0x3c003c00u, r1 = r0 + 1, r2 = r0 + 2, r3 = r0 + 3, r4 = r0 + 4, r5 = r0 + 5, r6 = r0 + 6, r7 = r0 + 7;
    f2 p0 = {1.f, 1.f}, p1 = p0, p2 = p0, p3 = p0, p4 = p0, p5 = p0, p6 = p0, p7 = p0;
    const unsigned a = 0x38003800u + threadIdx.x;                 // (0.5, 0.5) in fp16
    const f2 pa = {0.999f, 1.001f};
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
#define I(n) "v_pk_add_f32 %" #n ", %" #n ", %8\n"
            if (MODE == 0) asm volatile(R8(I) : OUT8P : "v"(pa));
#undef I
#define I(n) "v_add_f32 %" #n ", %" #n ", %8\n"
            if (MODE == 1) asm volatile(R8(I) : OUT8 : "v"(a));
#undef I
#define I(n) "v_pk_add_f16 %" #n ", %" #n ", %8 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]\n"
            if (MODE == 2) asm volatile(R8(I) : OUT8 : "v"(a));
#undef I
#define I(n) "v_pk_fma_f16 %" #n ", %8, %8, %" #n " op_sel:[0,0,0] op_sel_hi:[0,1,1]\n"
            if (MODE == 3) asm volatile(R8(I) : OUT8 : "v"(a));
#undef I
#define I(n) "v_dot2_f32_f16 %" #n ", %8, %8, %" #n " neg_hi:[1,0,0]\n"
            if (MODE == 4) asm volatile(R8(I) : OUT8 : "v"(a));
#undef I
#define I(n) "v_cvt_pk_f16_f32 %" #n ", %" #n ", %8\n"
            if (MODE == 5) asm volatile(R8(I) : OUT8 : "v"(a));
#undef I
            if (MODE == 6) asm volatile("v_permlane32_swap_b32 %0, %1\n v_permlane32_swap_b32 %2, %3\n v_permlane32_swap_b32 %4, %5\n v_permlane32_swap_b32 %6, %7\n"
                                        "v_permlane32_swap_b32 %0, %1\n v_permlane32_swap_b32 %2, %3\n v_permlane32_swap_b32 %4, %5\n v_permlane32_swap_b32 %6, %7\n" : OUT8);
            if (MODE == 7) asm volatile("v_permlane16_swap_b32 %0, %1\n v_permlane16_swap_b32 %2, %3\n v_permlane16_swap_b32 %4, %5\n v_permlane16_swap_b32 %6, %7\n"
                                        "v_permlane16_swap_b32 %0, %1\n v_permlane16_swap_b32 %2, %3\n v_permlane16_swap_b32 %4, %5\n v_permlane16_swap_b32 %6, %7\n" : OUT8);
#define I(n) "v_rsq_f32 %" #n ", %" #n "\n"
            if (MODE == 8) asm volatile(R8(I) : OUT8);
#undef I
#define I(n) "v_alignbit_b32 %" #n ", %" #n ", %" #n ", 16\n"
            if (MODE == 9) asm volatile(R8(I) : OUT8);
#undef I
#define I(n) "v_pk_fma_f32 %" #n ", %8, %8, %" #n "\n"
            if (MODE == 10) asm volatile(R8(I) : OUT8P : "v"(pa));
#undef I
#define I(n) "v_pk_mul_f32 %" #n ", %" #n ", %8\n"
            if (MODE == 11) asm volatile(R8(I) : OUT8P : "v"(pa));
#undef I
#define I(n) "v_fma_f32 %" #n ", %8, %8, %" #n "\n"
            if (MODE == 12) asm volatile(R8(I) : OUT8 : "v"(a));
#undef I
#define I(n) "v_cvt_f32_f16 %" #n ", %" #n "\n"
            if (MODE == 13) asm volatile(R8(I) : OUT8);
#undef I
#define I(n) "v_pk_mul_f16 %" #n ", %" #n ", %8 op_sel_hi:[1,0]\n"
            if (MODE == 14) asm volatile(R8(I) : OUT8 : "v"(a));
#undef I
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
    const f2 ps = p0 + p1 + p2 + p3 + p4 + p5 + p6 + p7;
    out[blockIdx.x * blockDim.x + threadIdx.x] = __uint_as_float(r0 ^ r1 ^ r2 ^ r3 ^ r4 ^ r5 ^ r6 ^ r7) + ps.x + ps.y;
}

template <int MODE>
void run(const char *name, float *out, long long *cyc)
{
    const int iters = 4000;
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
        const double n = (double)iters * 64;
        printf("%-44s waves/SIMD %d: %.2f ns per instruction per SIMD (kernel %.3f ms)\n", name, threads / 256, ms * 1e6 / (n * (threads / 256)), ms);
    }
}

// semantics: complex numbers as (re, im) halves of a register
__global__ void sem(float *o)
{
    const h2 a = {(_Float16)1.5f, (_Float16)-0.25f}, b = {(_Float16)0.75f, (_Float16)2.0f};
    float re = 0.f, im = 0.f;
    // a conj(b) = (ar br + ai bi, ai br - ar bi): dot2(a, b), dot2(swap(a), b) with the high product negated
    unsigned as;
    asm volatile("v_alignbit_b32 %0, %1, %1, 16" : "=v"(as) : "v"(a));
    asm volatile("v_dot2_f32_f16 %0, %1, %2, %0" : "+v"(re) : "v"(a), "v"(b));
    asm volatile("v_dot2_f32_f16 %0, %1, %2, %0 neg_hi:[1,0,0]" : "+v"(im) : "v"(as), "v"(b));
    o[0] = re; o[1] = im;
    // a - j b = (ar + bi, ai - br): v_pk_add_f16 with b's halves swapped and the second (hi) result's operand negated
    h2 s;
    asm volatile("v_pk_add_f16 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(s) : "v"(a), "v"(b));
    o[2] = (float)s.x; o[3] = (float)s.y;
    // a * b (complex): t = a * b.re (broadcast lo), then t += (-ai, ar) * b.im
    h2 t;
    asm volatile("v_pk_mul_f16 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(t) : "v"(a), "v"(b));
    asm volatile("v_pk_fma_f16 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]" : "+v"(t) : "v"(a), "v"(b));
    o[4] = (float)t.x; o[5] = (float)t.y;
}

int main()
{
    float *out; long long *cyc;
    hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&cyc, 16 * 8);
    run<0>("v_pk_add_f32", out, cyc);
    run<11>("v_pk_mul_f32", out, cyc);
    run<10>("v_pk_fma_f32", out, cyc);
    run<1>("v_add_f32", out, cyc);
    run<12>("v_fma_f32", out, cyc);
    run<2>("v_pk_add_f16 (op_sel, neg)", out, cyc);
    run<14>("v_pk_mul_f16 (op_sel_hi broadcast)", out, cyc);
    run<3>("v_pk_fma_f16 (op_sel)", out, cyc);
    run<4>("v_dot2_f32_f16 (neg_hi)", out, cyc);
    run<5>("v_cvt_pk_f16_f32", out, cyc);
    run<13>("v_cvt_f32_f16", out, cyc);
    run<6>("v_permlane32_swap_b32", out, cyc);
    run<7>("v_permlane16_swap_b32", out, cyc);
    run<8>("v_rsq_f32", out, cyc);
    run<9>("v_alignbit_b32", out, cyc);
    float *o; hipMalloc(&o, 64);
    hipLaunchKernelGGL(sem, dim3(1), dim3(1), 0, 0, o);
    float h[6]; hipMemcpy(h, o, sizeof(h), hipMemcpyDeviceToHost);
    // a = 1.5 - 0.25j, b = 0.75 + 2j: a conj(b) = 0.625 - 3.1875j; a - jb = 3.5 - 1.0j; a b = 1.625 + 2.8125j
    printf("a conj(b) = %g %+gj (want 0.625 -3.1875j)\na - j b   = %g %+gj (want 3.5 -1j)\na b       = %g %+gj (want 1.625 +2.8125j)\n", h[0], h[1], h[2], h[3], h[4], h[5]);
    return 0;
}
