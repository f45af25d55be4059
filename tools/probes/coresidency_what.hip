// coresidency_what.hip -- WHAT does a packed-fp32 instruction whose LOW lane takes the HIGH half of src1 (op_sel:[0,1]) return when it goes wrong
// beside the MFMA + LDS-read neighbour of coresidency_standalone.hip?  Every iteration writes fresh, recognisable values into the
// register pair b = (it, it + 0.5) and, GAP independent instructions later, computes r = (0,0) + b with the halves crossed: expected
// (it + 0.5, it).  The first wrong r of each thread is recorded with its iteration.
// build + run: hipcc -O3 --offload-arch=gfx950 tools/probes/coresidency_what.hip -o abtest/what && abtest/what
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <map>
#include <string>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); std::exit(2); } } while (0)

__global__ __launch_bounds__(256) void neighbour(float *sink, long long iters, float lds_fill, float reg_fill)
{
    __shared__ __attribute__((aligned(16))) _Float16 As[2][128][40];
    __shared__ __attribute__((aligned(16))) _Float16 Bs[2][192][40];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    for (int i = tid; i < 2 * 128 * 40; i += 256) (&As[0][0][0])[i] = (_Float16)lds_fill;
    for (int i = tid; i < 2 * 192 * 40; i += 256) (&Bs[0][0][0])[i] = (_Float16)lds_fill;
    __syncthreads();
    f32x16 acc[2][3];
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 3; ++j)
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    f16x8 zero;
    for (int r = 0; r < 8; ++r) zero[r] = (_Float16)reg_fill;
    float vs = 0.f;
    for (long long it = 0; it < iters; ++it) {
        asm volatile("" ::: "memory");
        for (int kk = 0; kk < 32; kk += 16) {
            const int ko = kk + 8 * (lane >> 5);
            f16x8 af[2][2], bf[2][3];
            for (int pl = 0; pl < 2; ++pl) {
                for (int i = 0; i < 2; ++i) af[pl][i] = *reinterpret_cast<const f16x8 *>(&As[pl][wm * 64 + i * 32 + (lane & 31)][ko]);
                for (int j = 0; j < 3; ++j) bf[pl][j] = *reinterpret_cast<const f16x8 *>(&Bs[pl][wn * 96 + j * 32 + (lane & 31)][ko]);
            }
            for (int pl = 0; pl < 2; ++pl) {
                for (int i = 0; i < 2; ++i) vs += (float)af[pl][i][0] + (float)af[pl][i][7];
                for (int j = 0; j < 3; ++j) vs += (float)bf[pl][j][0] + (float)bf[pl][j][7];
            }
            for (int i = 0; i < 2; ++i)
                for (int j = 0; j < 3; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(zero, zero, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(zero, zero, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(zero, zero, acc[i][j], 0, 0, 0);
                }
        }
    }
    float s = vs;
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 3; ++j)
            for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    sink[blockIdx.x * 256 + threadIdx.x] = s;
}

struct Rec { int count, first_it; float rx, ry; };

// WRITER: how b is produced -- 0 two v_mov/cvt of scalar-width ops (one per half), 1 one packed op writing both halves
template <int GAP, int WRITER>
__global__ __launch_bounds__(256) void victim(Rec *out, int iters)
{
    Rec rec = {0, -1, 0.f, 0.f};
    const f2 zero = {0.f, 0.f}, ones = {1.f, 1.f};
    f2 b = {0.f, 0.5f};
    float pad[8];
    for (int i = 0; i < 8; ++i) pad[i] = (float)(threadIdx.x + i);
    for (int it = 0; it < iters; ++it) {
        if (WRITER == 0) {
            float lo, hi;
            asm volatile("v_cvt_f32_i32 %0, %1" : "=v"(lo) : "v"(it));
            asm volatile("v_add_f32 %0, 0.5, %1" : "=v"(hi) : "v"(lo));
            b.x = lo; b.y = hi;
        } else if (it > 0)
            asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(b) : "v"(b), "v"(ones));
#pragma unroll
        for (int g = 0; g < GAP; ++g) asm volatile("v_add_f32 %0, 1.0, %0" : "+v"(pad[g & 7]));
        f2 r;
        asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(r) : "v"(zero), "v"(b));
        const float want_x = (float)it + 0.5f, want_y = (float)it;
        if (r.x != want_x || r.y != want_y) {
            if (rec.count == 0) { rec.first_it = it; rec.rx = r.x; rec.ry = r.y; }
            ++rec.count;
        }
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += pad[i];
    if (s == -1.f) rec.count = -1;
    out[blockIdx.x * 256 + threadIdx.x] = rec;
}

int main(int argc, char **argv)
{
    const int V = 4096, iters = 4000;
    hipStream_t main_s, side_s;
    CK(hipStreamCreateWithFlags(&main_s, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&side_s, hipStreamNonBlocking));
    float *sink; Rec *out;
    CK(hipMalloc(&sink, 1024 * 256 * sizeof(float)));
    CK(hipMalloc(&out, (size_t)V * 256 * sizeof(Rec)));
    std::vector<Rec> h((size_t)V * 256);
    auto launch = [&](int which) {
        switch (which) {
        case 0: hipLaunchKernelGGL((victim<0, 0>), dim3(V), dim3(256), 0, main_s, out, iters); break;
        case 1: hipLaunchKernelGGL((victim<4, 0>), dim3(V), dim3(256), 0, main_s, out, iters); break;
        case 2: hipLaunchKernelGGL((victim<32, 0>), dim3(V), dim3(256), 0, main_s, out, iters); break;
        case 3: hipLaunchKernelGGL((victim<0, 1>), dim3(V), dim3(256), 0, main_s, out, iters); break;
        case 4: hipLaunchKernelGGL((victim<4, 1>), dim3(V), dim3(256), 0, main_s, out, iters); break;
        default: hipLaunchKernelGGL((victim<32, 1>), dim3(V), dim3(256), 0, main_s, out, iters); break;
        }
        CK(hipGetLastError());
    };
    const char *names[6] = {"halves written one at a time, read at once", "... read 4 instructions later", "... read 32 instructions later",
                            "pair written by one packed add, read at once", "... read 4 instructions later", "... read 32 instructions later"};
    for (int which = 0; which < 6; ++which)
        for (int with = 0; with < 3; ++with) {
            CK(hipDeviceSynchronize());
            // with == 2: the neighbour works on recognisable non-zero data (its accumulators overflow to inf; nobody reads them)
            if (with) { hipLaunchKernelGGL(neighbour, dim3(256), dim3(256), 0, side_s, sink, 30000LL, with == 2 ? 2.0f : 0.f, with == 2 ? 1.0f : 0.f); CK(hipGetLastError()); }
            launch(which);
            CK(hipDeviceSynchronize());
            CK(hipMemcpy(h.data(), out, h.size() * sizeof(Rec), hipMemcpyDeviceToHost));
            size_t threads = 0, total = 0;
            std::map<std::string, size_t> kinds;
            for (const Rec &r : h) {
                if (r.count <= 0) continue;
                ++threads; total += r.count;
                const float it = (float)r.first_it;
                char buf[160];
                auto name = [&](float v) -> std::string {
                    if (v == it + 0.5f) return "b.hi"; if (v == it) return "b.lo"; if (v == it - 0.5f) return "previous b.hi"; if (v == it - 1.f) return "previous b.lo";
                    if (v == 0.f) return "0";
                    if (__builtin_bit_cast(unsigned, v) == 0x40004000u) return "the neighbour's LDS data (fp16 2.0 pairs)";
                    if (__builtin_bit_cast(unsigned, v) == 0x3c003c00u) return "the neighbour's MFMA operand (fp16 1.0 pairs)"; std::snprintf(buf, sizeof buf, "other(%g at it %g)", v, it); return buf; };
                kinds["(" + name(r.rx) + ", " + name(r.ry) + ")"]++;
            }
            std::printf("%-48s %s: threads with a wrong result %zu of %d, wrong results %zu of %lld\n", names[which], with == 2 ? "beside the neighbour (non-zero data)" : with ? "beside the neighbour (all zeros)   " : "alone                              ",
                        threads, V * 256, total, (long long)V * 256 * iters);
            int shown = 0;
            for (auto &kv : kinds) { if (shown++ < 8) std::printf("      first wrong (lo, hi) = %s   [expected (b.hi, b.lo)]: %zu threads\n", kv.first.c_str(), kv.second); }
        }
    return 0;
}
