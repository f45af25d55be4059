// round 6: issue cost of the vector instructions the transform and MVDR kernels are made of, per SIMD with two waves resident (as those kernels run).
// v_pk_fma_f32 / v_pk_mul_f32 cost TWO plain fma slots on this chip (no throughput gain from packing, only fewer instructions); v_pk_add_f32 1.6,
// a DPP move 1.5, v_rsq_f32 2.7.  profiles/r06_mvdr_floor_model.md uses these costs.
// build: hipcc --offload-arch=gfx950 -O2 -o /tmp/valu_rate tools/probes/valu_rate_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
#define BODY(ASM) for (int it = 0; it < iters; ++it) { _Pragma("unroll") for (int i = 0; i < 16; ++i) { ASM; } }
template <int MODE>
__global__ __launch_bounds__(256, 2) void k(float *out, int iters, float seed)
{
    f2 v[16]; float s_[16];
    for (int i = 0; i < 16; ++i) { v[i] = f2{seed + i, seed - i}; s_[i] = seed * i; }
    f2 m = {1.0001f + threadIdx.x * 1e-9f, 0.9999f}, c = {1e-3f, -1e-3f};
    float ms = m[0], cs = c[0];
    if (MODE == 0) BODY(asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(m), "v"(c)))
    if (MODE == 1) BODY(asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(v[i]) : "v"(m), "v"(c)))
    if (MODE == 2) BODY(asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(v[i]) : "v"(m)))
    if (MODE == 3) BODY(asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(v[i]) : "v"(c)))
    if (MODE == 4) BODY(asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(s_[i]) : "v"(ms), "v"(cs)))
    if (MODE == 5) BODY(asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(s_[i]) : "v"(ms), "v"(cs)))
    if (MODE == 6) BODY(asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(v[i]) : "v"(m)))
    if (MODE == 7) BODY(asm volatile("v_pk_fma_f32 %0, %0, %0, %1" : "+v"(v[i]) : "v"(c)))
    if (MODE == 8) BODY(asm volatile("v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(s_[i])))
    if (MODE == 9) BODY(asm volatile("v_rsq_f32 %0, %0" : "+v"(s_[i])))
    if (MODE == 10) BODY(asm volatile("v_pk_fma_f32 %0, %0, %1, %2 op_sel:[1,0,1] op_sel_hi:[1,1,0]" : "+v"(v[i]) : "v"(m), "v"(c)))
    if (MODE == 11) BODY(asm volatile("v_add_f32 %0, %0, %1" : "+v"(s_[i]) : "v"(cs)))
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += v[i][0] + v[i][1] + s_[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MODE> void run(float *d, const char *name)
{
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 20000;
    hipLaunchKernelGGL(k<MODE>, dim3(512), dim3(256), 0, 0, d, 16, 1.0f);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(512), dim3(256), 0, 0, d, iters, 1.0f);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-44s %.3f ms  -> %.2f ns per wave-instruction per SIMD (two waves resident)\n", name, ms, ms * 1e6 / (2.0 * iters * 16));
}
int main()
{
    float *d; (void)hipMalloc(&d, 512 * 256 * 4);
    run<4>(d, "v_fma_f32 (3 VGPR sources)");
    run<5>(d, "v_fmac_f32");
    run<11>(d, "v_add_f32");
    run<0>(d, "v_pk_fma_f32 d = d*m + c");
    run<1>(d, "v_pk_fma_f32 d = m*c + d");
    run<6>(d, "v_pk_fma_f32 d = d*m + m (2 distinct)");
    run<7>(d, "v_pk_fma_f32 d = d*d + c (2 distinct)");
    run<10>(d, "v_pk_fma_f32 with op_sel crossing");
    run<2>(d, "v_pk_mul_f32");
    run<3>(d, "v_pk_add_f32");
    run<8>(d, "v_mov_b32_dpp quad_perm");
    run<9>(d, "v_rsq_f32");
    return 0;
}
