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

// kind 3: everything the repair contraction does at once -- 16-byte global loads into registers, ds_write_b128 into 50 KiB of static LDS,
// barrier, ds_read_b128 fragments, 32x32x16 fp16 MFMAs into AGPR accumulators, barrier -- on dummy data (src: >= 64 KiB of zeros)
// FLAGS (kinds 16..31 = 16 + FLAGS) switch the ingredients off one at a time: 1 the global loads, 2 the LDS staging writes, 4 the MFMAs (a vector sum
// of the fragments takes their place), 8 the barriers.  Kind 3 is FLAGS = 15.
template <int FLAGS>
__global__ __launch_bounds__(256) void hog_gemm(float *sink, const _Float16 *src, long long iters)
{
    __shared__ __attribute__((aligned(16))) _Float16 As[2][128][40];
    __shared__ __attribute__((aligned(16))) _Float16 Bs[2][192][40];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    const int lr = tid >> 2, lc = tid & 3;
    f32x16 acc[2][3];
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 3; ++j)
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    f16x8 ra[2][2], rb[2][3];
    for (long long it = 0; it < iters; ++it) {
        const _Float16 *g = src + ((it * 37 + blockIdx.x) & 63) * 512;
        if ((FLAGS & 1) || it == 0)
        for (int pl = 0; pl < 2; ++pl) {
            ra[pl][0] = *reinterpret_cast<const f16x8 *>(g + lr * 8 + pl * 4096);
            ra[pl][1] = *reinterpret_cast<const f16x8 *>(g + lr * 8 + 2048 + pl * 4096);
            for (int i = 0; i < 3; ++i) rb[pl][i] = *reinterpret_cast<const f16x8 *>(g + lr * 8 + 1024 * i + lc * 8 + pl * 4096);
        }
        if ((FLAGS & 2) || it == 0)
        for (int pl = 0; pl < 2; ++pl) {
            *reinterpret_cast<f16x8 *>(&As[pl][lr][lc * 8]) = ra[pl][0];
            *reinterpret_cast<f16x8 *>(&As[pl][lr + 64][lc * 8]) = ra[pl][1];
            for (int i = 0; i < 3; ++i) *reinterpret_cast<f16x8 *>(&Bs[pl][lr + 64 * i][lc * 8]) = rb[pl][i];
        }
        if ((FLAGS & 8) || it == 0) __syncthreads();
        for (int kk = 0; kk < 32; kk += 16) {
            const int ko = kk + 8 * (lane >> 5);
            f16x8 af[2][2], bf[2][3];
            for (int pl = 0; pl < 2; ++pl) {
                for (int i = 0; i < 2; ++i) af[pl][i] = *reinterpret_cast<const f16x8 *>(&As[pl][wm * 64 + i * 32 + (lane & 31)][ko]);
                for (int j = 0; j < 3; ++j) bf[pl][j] = *reinterpret_cast<const f16x8 *>(&Bs[pl][wn * 96 + j * 32 + (lane & 31)][ko]);
            }
            if (!(FLAGS & 4)) {
                for (int i = 0; i < 2; ++i)
                    for (int j = 0; j < 3; ++j)
                        for (int r = 0; r < 8; ++r) acc[i][j][r] += (float)af[0][i][r] * (float)bf[1][j][r] + (float)af[1][i][r] * (float)bf[0][j][r];
                asm volatile("" ::: "memory");
            } else
            for (int i = 0; i < 2; ++i)
                for (int j = 0; j < 3; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[1][i], bf[0][j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[0][i], bf[1][j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[0][i], bf[0][j], acc[i][j], 0, 0, 0);
                }
        }
        if (FLAGS & 8) __syncthreads();
    }
    float s = 0.f;
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 3; ++j)
            for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    sink[blockIdx.x * 256 + threadIdx.x] = s;
}

extern "C" int hog_launch(int kind, int n_wg, long long iters, float *sink, void *stream)
{
    hipStream_t st = (hipStream_t)stream;
    if (kind == 0) hipLaunchKernelGGL(hog_mfma, dim3(n_wg), dim3(256), 0, st, sink, iters);
    else if (kind == 1) hipLaunchKernelGGL(hog_lds, dim3(n_wg), dim3(256), 0, st, sink, iters);
    else if (kind == 3 || kind >= 16) {
        const _Float16 *src = reinterpret_cast<const _Float16 *>(sink);
        float *out = sink + 512 * 1024;
        switch (kind == 3 ? 15 : kind - 16) {
#define HOG_CASE(f) case f: hipLaunchKernelGGL(hog_gemm<f>, dim3(n_wg), dim3(256), 0, st, out, src, iters); break;
            HOG_CASE(0) HOG_CASE(1) HOG_CASE(2) HOG_CASE(3) HOG_CASE(4) HOG_CASE(5) HOG_CASE(6) HOG_CASE(7)
            HOG_CASE(8) HOG_CASE(9) HOG_CASE(10) HOG_CASE(11) HOG_CASE(12) HOG_CASE(13) HOG_CASE(14) HOG_CASE(15)
        }
    }
    else hipLaunchKernelGGL(hog_valu, dim3(n_wg), dim3(256), 0, st, sink, iters);
    return (int)hipGetLastError();
}
