// coresidency_probe.hip -- which kind of neighbour on the same CU makes k_beamform_wave return wrong hops?
// Round 4 found the beamformer wrong whenever one of its workgroups shared a CU with k_srp_gemm_repair running on another stream
// (DESIGN.md section 7).  This file builds "hog" kernels that each exercise ONE resource the repair contraction uses -- the matrix
// cores with accumulators in AGPRs, static LDS traffic, plain vector work -- one 256-thread workgroup per CU, running for a few
// milliseconds on their own stream while the library's serial path (beamformer on the caller's stream) runs beside them;
// tools/probes/coresidency_probe.py compares the audio with a run that had no neighbour.
// build: hipcc -O3 --offload-arch=gfx950 -shared -fPIC tools/probes/coresidency_probe.hip -o abtest/libhog.so
#include <hip/hip_runtime.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

__global__ __launch_bounds__(256) void hog_mfma(float *sink, long long iters)
{
    f32x16 acc[6];
    for (int j = 0; j < 6; ++j)
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.001f * (threadIdx.x + i)); b[i] = (_Float16)(0.002f * (threadIdx.x - i)); }
    for (long long it = 0; it < iters; ++it)
#pragma unroll
        for (int j = 0; j < 6; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[j], 0, 0, 0);
    float s = 0.f;
    for (int j = 0; j < 6; ++j)
        for (int r = 0; r < 16; ++r) s += acc[j][r];
    sink[blockIdx.x * 256 + threadIdx.x] = s;
}

__global__ __launch_bounds__(256) void hog_lds(float *sink, long long iters)
{
    __shared__ __attribute__((aligned(16))) float buf[12800];          // 50 KiB static, as the repair contraction's tiles
    for (int i = threadIdx.x; i < 12800; i += 256) buf[i] = (float)i;
    __syncthreads();
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    for (long long it = 0; it < iters; ++it) {
        const int o = (int)((threadIdx.x * 4 + it * 1024) % 12796) & ~3;
        const float4 w = *reinterpret_cast<const float4 *>(buf + o);
        v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w;
        *reinterpret_cast<float4 *>(buf + ((o + 2048) % 12796 & ~3)) = v;
        __syncthreads();
    }
    sink[blockIdx.x * 256 + threadIdx.x] = v.x + v.y + v.z + v.w;
}

__global__ __launch_bounds__(256) void hog_valu(float *sink, long long iters)
{
    float v[32];
    for (int i = 0; i < 32; ++i) v[i] = 0.001f * (threadIdx.x + i);
    for (long long it = 0; it < iters; ++it)
#pragma unroll
        for (int i = 0; i < 32; ++i) v[i] = fmaf(v[i], 1.0001f, 0.5f);
    float s = 0.f;
    for (int i = 0; i < 32; ++i) s += v[i];
    sink[blockIdx.x * 256 + threadIdx.x] = s;
}

extern "C" int hog_launch(int kind, int n_wg, long long iters, float *sink, void *stream)
{
    hipStream_t st = (hipStream_t)stream;
    if (kind == 0) hipLaunchKernelGGL(hog_mfma, dim3(n_wg), dim3(256), 0, st, sink, iters);
    else if (kind == 1) hipLaunchKernelGGL(hog_lds, dim3(n_wg), dim3(256), 0, st, sink, iters);
    else hipLaunchKernelGGL(hog_valu, dim3(n_wg), dim3(256), 0, st, sink, iters);
    return (int)hipGetLastError();
}
