// coresidency_standalone.hip -- is the co-residency hazard of DESIGN.md section 7 a property of this library's kernels, or of the machine?
// No library code here: a "neighbour" (LDS fragment reads feeding 32x32x16 fp16 MFMAs, the inner loop of any tiled contraction, on zeros)
// runs on one stream with one workgroup per CU; a "victim" of one of three trivial kinds runs on another stream at the same time, and its
// output is compared bit for bit with a run that had no neighbour.  The victims:
//   0 registers only   -- a dependent fma chain per thread
//   1 LDS round trips  -- each workgroup writes a pattern to dynamic LDS, barriers, reads it back permuted, accumulates
//   2 global loads     -- each thread sums a strided slice of a constant buffer
//   3 LDS, no barriers -- each WAVE owns a slice of LDS (the wave-level FFT's exchange pattern: ds_write then ds_read of the same wave)
// build + run on the GPU box:  hipcc -O3 --offload-arch=gfx950 tools/probes/coresidency_standalone.hip -o abtest/cores && abtest/cores
#include <hip/hip_runtime.h>
#ifdef WITH_LIBRARY_FFT                 // -DWITH_LIBRARY_FFT -Imcarray_amd/csrc: victims 15.. run this library's wave-level transform on made-up data
#include "fft1024c.h"
#endif
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); std::exit(2); } } while (0)

// mode 0: LDS reads feed the MFMAs; 1: the MFMAs take register operands and the LDS reads are summed by vector adds; 2: LDS reads only
template <int MODE>
__global__ __launch_bounds__(256) void neighbour(float *sink, long long iters)
{
    __shared__ __attribute__((aligned(16))) _Float16 As[2][128][40];
    __shared__ __attribute__((aligned(16))) _Float16 Bs[2][192][40];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    for (int i = tid; i < 2 * 128 * 40; i += 256) (&As[0][0][0])[i] = (_Float16)0.f;
    for (int i = tid; i < 2 * 192 * 40; i += 256) (&Bs[0][0][0])[i] = (_Float16)0.f;
    __syncthreads();
    f32x16 acc[2][3];
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 3; ++j)
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    f16x8 zero;
    for (int r = 0; r < 8; ++r) zero[r] = (_Float16)0.f;
    float vs = 0.f;
    for (long long it = 0; it < iters; ++it) {
        asm volatile("" ::: "memory");                    // the LDS reads stay inside the loop
        for (int kk = 0; kk < 32; kk += 16) {
            const int ko = kk + 8 * (lane >> 5);
            f16x8 af[2][2], bf[2][3];
            for (int pl = 0; pl < 2; ++pl) {
                for (int i = 0; i < 2; ++i) af[pl][i] = *reinterpret_cast<const f16x8 *>(&As[pl][wm * 64 + i * 32 + (lane & 31)][ko]);
                for (int j = 0; j < 3; ++j) bf[pl][j] = *reinterpret_cast<const f16x8 *>(&Bs[pl][wn * 96 + j * 32 + (lane & 31)][ko]);
            }
            if (MODE != 0)
                for (int pl = 0; pl < 2; ++pl) {
                    for (int i = 0; i < 2; ++i) vs += (float)af[pl][i][0] + (float)af[pl][i][7];
                    for (int j = 0; j < 3; ++j) vs += (float)bf[pl][j][0] + (float)bf[pl][j][7];
                }
            if (MODE != 2)
                for (int i = 0; i < 2; ++i)
                    for (int j = 0; j < 3; ++j) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(MODE ? zero : af[1][i], MODE ? zero : bf[0][j], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(MODE ? zero : af[0][i], MODE ? zero : bf[1][j], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(MODE ? zero : af[0][i], MODE ? zero : bf[0][j], acc[i][j], 0, 0, 0);
                    }
        }
    }
    float s = vs;
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 3; ++j)
            for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    sink[blockIdx.x * 256 + threadIdx.x] = s;
}

__global__ __launch_bounds__(256) void victim_regs(unsigned *out, int iters)
{
    unsigned v[8];
    for (int i = 0; i < 8; ++i) v[i] = blockIdx.x * 2654435761u + threadIdx.x * 40503u + i;
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = v[i] * 1664525u + 1013904223u + v[(i + 1) & 7];
    unsigned s = 0;
    for (int i = 0; i < 8; ++i) s ^= v[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

__global__ __launch_bounds__(256) void victim_lds(unsigned *out, int iters, int words)
{
    extern __shared__ unsigned buf[];
    unsigned acc = blockIdx.x * 2654435761u + threadIdx.x;
    for (int it = 0; it < iters; ++it) {
        for (int i = threadIdx.x; i < words; i += 256) buf[i] = acc + i * 2246822519u + it;
        __syncthreads();
        for (int i = threadIdx.x; i < words; i += 256) acc = acc * 1664525u + buf[(i * 33 + 7 * it) % words];
        __syncthreads();
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

__global__ __launch_bounds__(256) void victim_global(unsigned *out, const unsigned *src, int n_words, int iters)
{
    unsigned acc = 0;
    size_t at = (size_t)(blockIdx.x * 256 + threadIdx.x) * 4;
    for (int it = 0; it < iters; ++it) {
        const uint4 v = *reinterpret_cast<const uint4 *>(src + at % (size_t)(n_words - 4));
        acc = acc * 1664525u + v.x + 3u * v.y + 5u * v.z + 7u * v.w;
        at += 256u * 4u * 977u;
        at &= ~(size_t)3;
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

// each wave has words_per_wave of LDS to itself: lane l writes slot l + 64 r, then reads the slot another lane wrote, with no barrier --
// the same wave's LDS operations are ordered, which the wave-level transforms of this library rely on
__global__ __launch_bounds__(256) void victim_wave_lds(unsigned *out, int iters, int words_per_wave)
{
    extern __shared__ unsigned buf[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned *mine = buf + wave * words_per_wave;
    unsigned acc = blockIdx.x * 2654435761u + threadIdx.x;
    const int rows = words_per_wave / 64;
    for (int it = 0; it < iters; ++it) {
        for (int r = 0; r < rows; ++r) mine[r * 64 + lane] = acc + r * 2246822519u + it;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        for (int r = 0; r < rows; ++r) acc = acc * 1664525u + mine[((r * 7 + it) % rows) * 64 + ((lane * 5 + r) & 63)];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

typedef float f2 __attribute__((ext_vector_type(2)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(256) void victim_f32(unsigned *out, int iters)
{
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = 0.25f + 1e-3f * (float)((blockIdx.x * 7 + threadIdx.x * 3 + i) & 1023);
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = fmaf(v[i], 0.73f, 0.31f * v[(i + 1) & 7]) + 0.01f;
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += v[i];
    out[blockIdx.x * 256 + threadIdx.x] = __float_as_uint(s);
}

__global__ __launch_bounds__(256) void victim_pk_f32(unsigned *out, int iters)
{
    f2 v[8];
    for (int i = 0; i < 8; ++i) { v[i].x = 0.25f + 1e-3f * (float)((blockIdx.x * 7 + threadIdx.x * 3 + i) & 1023); v[i].y = 0.5f - v[i].x; }
    const f2 a = {0.73f, 0.69f}, b = {0.31f, -0.29f}, c = {0.01f, 0.02f};
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = __builtin_elementwise_fma(v[i], a, __builtin_elementwise_fma(v[(i + 1) & 7], b, c));
    f2 s = {0.f, 0.f};
    for (int i = 0; i < 8; ++i) s += v[i];
    out[blockIdx.x * 256 + threadIdx.x] = __float_as_uint(s.x) ^ (__float_as_uint(s.y) * 3u);
}

__global__ __launch_bounds__(256) void victim_trans(unsigned *out, int iters)
{
    float v = 0.1f + 1e-3f * (float)((blockIdx.x * 7 + threadIdx.x * 3) & 1023), s = 0.f;
    for (int it = 0; it < iters; ++it) {
        float sn, cs;
        __sincosf(v * 3.0f, &sn, &cs);
        s += sn * 0.5f + cs * 0.25f + __frsqrt_rn(1.0f + v * v) + __builtin_amdgcn_exp2f(-v);
        v = 0.1f + 0.9f * fabsf(sn);
    }
    out[blockIdx.x * 256 + threadIdx.x] = __float_as_uint(s);
}

__global__ __launch_bounds__(256) void victim_sincos_precise(unsigned *out, int iters)
{
    float v = 0.1f + 1e-3f * (float)((blockIdx.x * 7 + threadIdx.x * 3) & 1023), s = 0.f;
    for (int it = 0; it < iters; ++it) {
        float sn, cs;
        sincosf(v * 1000.0f, &sn, &cs);
        s += sn * 0.5f + cs * 0.25f;
        v = 0.1f + 0.9f * fabsf(sn);
    }
    out[blockIdx.x * 256 + threadIdx.x] = __float_as_uint(s);
}

__global__ __launch_bounds__(256) void victim_crosslane(unsigned *out, int iters)
{
    unsigned v = blockIdx.x * 2654435761u + threadIdx.x * 40503u;
    const int lane = threadIdx.x & 63;
    for (int it = 0; it < iters; ++it) {
        v = v * 1664525u + (unsigned)__builtin_amdgcn_ds_bpermute(((lane * 5 + it) & 63) * 4, (int)v);
        v += (unsigned)__builtin_amdgcn_mov_dpp((int)v, 0x4E, 0xF, 0xF, true);               // quad_perm [2,3,0,1]
        v = v * 22695477u + (unsigned)__builtin_amdgcn_mov_dpp((int)v, 0x141, 0xF, 0xF, true);   // row_mirror... (row_half_mirror 0x141)
        v += (unsigned)__shfl_xor((int)v, 32);
        v += (unsigned)__builtin_amdgcn_readlane((int)v, it & 63);
    }
    out[blockIdx.x * 256 + threadIdx.x] = v;
}

__global__ __launch_bounds__(256) void victim_f16(unsigned *out, int iters)
{
    h2 v[8];
    for (int i = 0; i < 8; ++i) { v[i].x = (_Float16)(0.25f + 1e-3f * (float)((blockIdx.x * 7 + threadIdx.x * 3 + i) & 255)); v[i].y = (_Float16)0.5f - v[i].x; }
    const h2 a = {(_Float16)0.73f, (_Float16)0.69f}, b = {(_Float16)0.31f, (_Float16)-0.29f};
    float d = 0.f;
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            v[i] = v[i] * a + v[(i + 1) & 7] * b;
            d = __builtin_amdgcn_fdot2(v[i], a, d * 0.5f, false);
        }
    out[blockIdx.x * 256 + threadIdx.x] = __float_as_uint(d);
}

// a victim that itself uses the matrix cores on register operands (the library's contraction and its 3-product split do)
__global__ __launch_bounds__(256) void victim_mfma(unsigned *out, int iters)
{
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.01f * ((threadIdx.x + i + blockIdx.x) & 63)); b[i] = (_Float16)(0.02f * ((threadIdx.x * 3 - i) & 31)); }
    for (int it = 0; it < iters; ++it) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
        for (int r = 0; r < 16; ++r) acc[r] *= 0.5f;
    }
    float s = 0.f;
    for (int r = 0; r < 16; ++r) s += acc[r];
    out[blockIdx.x * 256 + threadIdx.x] = __float_as_uint(s);
}

// wider LDS accesses: T = uint2 (ds_*_b64), uint4 (ds_*_b128); PAIRS also touches two far-apart slots per thread, which the compiler merges
// into the two-address forms (ds_read2 / ds_write2 / read2st64) that transform kernels are full of
template <typename T, bool PAIRS>
__global__ __launch_bounds__(256) void victim_lds_wide(unsigned *out, int iters, int slots)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char raw[];
    T *buf = reinterpret_cast<T *>(raw);
    unsigned acc = blockIdx.x * 2654435761u + threadIdx.x;
    const int half = slots / 2;
    for (int it = 0; it < iters; ++it) {
        if (PAIRS) {
            for (int i = threadIdx.x; i < half; i += 256) {
                T a, b;
                a.x = acc + i * 2246822519u + it; a.y = a.x * 3u + 1u; b.x = a.x ^ 0x9e3779b9u; b.y = b.x * 5u + 7u;
                if constexpr (sizeof(T) == 16) { a.z = a.x + 11u; a.w = a.y + 13u; b.z = b.x + 17u; b.w = b.y + 19u; }
                buf[i] = a; buf[i + half] = b;
            }
        } else
            for (int i = threadIdx.x; i < slots; i += 256) {
                T a;
                a.x = acc + i * 2246822519u + it; a.y = a.x * 3u + 1u;
                if constexpr (sizeof(T) == 16) { a.z = a.x + 11u; a.w = a.y + 13u; }
                buf[i] = a;
            }
        __syncthreads();
        if (PAIRS) {
            for (int i = threadIdx.x; i < half; i += 256) {
                const int j = (i * 33 + 7 * it) % half;
                const T a = buf[j], b = buf[j + half];
                acc = acc * 1664525u + a.x + 3u * a.y + 5u * b.x + 7u * b.y;
                if constexpr (sizeof(T) == 16) acc += a.z + a.w * 11u + b.z * 13u + b.w * 17u;
            }
        } else
            for (int i = threadIdx.x; i < slots; i += 256) {
                const T a = buf[(i * 33 + 7 * it) % slots];
                acc = acc * 1664525u + a.x + 3u * a.y;
                if constexpr (sizeof(T) == 16) acc += a.z + a.w * 11u;
            }
        __syncthreads();
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

// one packed-fp32 instruction form per victim, written by hand so that the operand selects are exactly these
template <int FORM>
__global__ __launch_bounds__(256) void victim_pk_form(unsigned *out, int iters)
{
    f2 v[8];
    for (int i = 0; i < 8; ++i) { v[i].x = 0.25f + 1e-3f * (float)((blockIdx.x * 7 + threadIdx.x * 3 + i) & 1023); v[i].y = 0.5f - v[i].x; }
    const f2 half = {0.5f, 0.5f}, w = {0.6f, -0.8f};
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            f2 r, a = v[i], b = v[(i + 1) & 7];
            if (FORM == 0) asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
            else if (FORM == 1) asm volatile("v_pk_add_f32 %0, %1, %2 neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
            else if (FORM == 2) asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(r) : "v"(a), "v"(b));
            else if (FORM == 3) asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
            else if (FORM == 4) asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0]" : "=v"(r) : "v"(a), "v"(w));
            else if (FORM == 5) asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[0,1,1] neg_lo:[0,0,1]" : "=v"(r) : "v"(a), "v"(w), "v"(b));
            else if (FORM == 6) asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(w), "v"(b));
            else if (FORM == 7) asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(w));
            else if (FORM == 8) asm volatile("v_pk_add_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(r) : "v"(a), "v"(b));
            else if (FORM == 9) asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[1,0]" : "=v"(r) : "v"(a), "v"(b));
            else if (FORM == 10) asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1]" : "=v"(r) : "v"(a), "v"(b));
            else if (FORM == 11) asm volatile("v_pk_add_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
            else if (FORM == 12) asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
            else if (FORM == 13) asm volatile("v_pk_fma_f32 %0, %2, 1.0, %1 op_sel:[1,0,0] op_sel_hi:[0,0,1] neg_hi:[1,0,0]" : "=v"(r) : "v"(a), "v"(b));   // = form 3, a - j b
            else if (FORM == 14) asm volatile("v_pk_fma_f32 %0, %2, 1.0, %1 op_sel:[1,0,0] op_sel_hi:[0,0,1] neg_lo:[1,0,0]" : "=v"(r) : "v"(a), "v"(b));   // a + j b
            else if (FORM == 15) asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(r) : "v"(a), "v"(b));
            else if (FORM == 16) asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,0,1]" : "=v"(r) : "v"(a), "v"(b), "v"(w));
            else if (FORM == 17) asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,1] op_sel_hi:[1,1,0]" : "=v"(r) : "v"(a), "v"(w), "v"(b));
            else if (FORM == 18) {   // a complex product with crossings on src0 and src2 only
                f2 t;
                asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,1]" : "=v"(t) : "v"(a), "v"(b));
                asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,1] op_sel_hi:[0,1,0] neg_lo:[0,0,1]" : "=v"(r) : "v"(a), "v"(b), "v"(t));
            } else if (FORM == 19) {   // packed fp16: the halves of ONE register crossed on src1
                unsigned ra, rb, rr;
                h2 ah = {(_Float16)a.x, (_Float16)a.y}, bh = {(_Float16)b.x, (_Float16)b.y}, rh;
                ra = __builtin_bit_cast(unsigned, ah); rb = __builtin_bit_cast(unsigned, bh);
                asm volatile("v_pk_add_f16 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(rr) : "v"(ra), "v"(rb));
                rh = __builtin_bit_cast(h2, rr); r.x = (float)rh.x; r.y = (float)rh.y;
            } else asm volatile("v_pk_mov_b32 %0, %1, %2 op_sel:[1,0]" : "=v"(r) : "v"(b), "v"(b));   // (b.hi, b.lo)
            asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(v[i]) : "v"(r), "v"(half));
            v[i].x += 0.125f;
        }
    unsigned s = 0;
    for (int i = 0; i < 8; ++i) s = s * 31u + __float_as_uint(v[i].x) + 7u * __float_as_uint(v[i].y);
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

#ifdef WITH_LIBRARY_FFT
// PART: 0 the whole 1024-point transform; 1 only its register butterflies (no LDS, no lane swaps); 2 butterflies + lane swaps; 3 butterflies + the LDS exchange
template <int PART>
__global__ __launch_bounds__(256) void victim_fft(unsigned *out, int iters)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char raw[];
    float2 *tab = reinterpret_cast<float2 *>(raw);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float2 *buf = tab + mca::F1K_TWORDS + wave * mca::F1K_SCRATCH;
    mca::f1k_table_init(tab, threadIdx.x, 256);
    __syncthreads();
    mca::F1kLane lc;
    lc.init(lane);
    float2 v[16];
    unsigned h = blockIdx.x * 2654435761u + threadIdx.x * 40503u;
    for (int i = 0; i < 16; ++i) {
        h = h * 1664525u + 1013904223u; v[i].x = (float)(h >> 8) * (1.0f / 16777216.0f) - 0.5f;
        h = h * 1664525u + 1013904223u; v[i].y = (float)(h >> 8) * (1.0f / 16777216.0f) - 0.5f;
    }
    for (int it = 0; it < iters; ++it) {
        if (PART == 0) mca::fft1024c<false, 3>(v, buf, lane, tab, lc);
        else {
            mca::fft16<false>(v);
            if (PART == 2)
                for (int q = 0; q < 4; ++q) mca::transpose_rows4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
            if (PART == 3) {
                float2 *wr = buf + (lane >> 4) * (4 * mca::F1K_ROW) + (lane & 15);
                for (int q = 0; q < 4; ++q)
                    for (int kb = 0; kb < 4; ++kb) wr[(q + 16 * kb) * mca::F1K_ROW] = v[4 * q + kb];
                mca::wave_lds_fence();
                mca::lds_read16_b128(v, buf + lane * mca::F1K_ROW);
                mca::wave_lds_fence();
            }
            for (int p = 1; p < 16; ++p) v[p] = mca::cmul(v[p], lc.wb[p & 3 ? p & 3 : 1]);
        }
        for (int p = 0; p < 16; ++p) v[p] = mca::cscale(v[p], PART == 0 ? 1.0f / 32.0f : 0.25f);
    }
    unsigned s = 0;
    for (int p = 0; p < 16; ++p) s = s * 31u + __float_as_uint(v[p].x) + 7u * __float_as_uint(v[p].y);
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
#endif

extern "C" int neighbour_launch(int mode, int n_wg, long long iters, float *sink, void *stream)
{
    hipStream_t st = (hipStream_t)stream;
    if (mode == 0) hipLaunchKernelGGL(neighbour<0>, dim3(n_wg), dim3(256), 0, st, sink, iters);
    else if (mode == 1) hipLaunchKernelGGL(neighbour<1>, dim3(n_wg), dim3(256), 0, st, sink, iters);
    else hipLaunchKernelGGL(neighbour<2>, dim3(n_wg), dim3(256), 0, st, sink, iters);
    return (int)hipGetLastError();
}

#ifndef AS_LIB
int main(int argc, char **argv)
{
    const int n_wg = argc > 1 ? std::atoi(argv[1]) : 256;
    const long long hog_iters = argc > 2 ? std::atoll(argv[2]) : 40000;
    const int V = 8192;                                   // victim workgroups
    const int first_kind = argc > 3 ? std::atoi(argv[3]) : 0;
    hipStream_t main_s, side_s;
    CK(hipStreamCreateWithFlags(&main_s, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&side_s, hipStreamNonBlocking));
    float *sink; unsigned *out, *src;
    const int n_src = 1 << 22;
    CK(hipMalloc(&sink, 1024 * 256 * sizeof(float)));
    CK(hipMalloc(&out, (size_t)V * 256 * 4));
    CK(hipMalloc(&src, (size_t)n_src * 4));
    std::vector<unsigned> h(n_src);
    for (int i = 0; i < n_src; ++i) h[i] = (unsigned)i * 2654435761u + 12345u;
    CK(hipMemcpy(src, h.data(), (size_t)n_src * 4, hipMemcpyHostToDevice));
    const int lds_words = 13824;                          // 54 KiB, the beamformer's footprint
#ifdef WITH_LIBRARY_FFT
    const int fft_lds = (mca::F1K_TWORDS + 4 * mca::F1K_SCRATCH) * 8;
#endif
    const int n_kinds = 40;
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(victim_lds), hipFuncAttributeMaxDynamicSharedMemorySize, lds_words * 4));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(victim_wave_lds), hipFuncAttributeMaxDynamicSharedMemorySize, lds_words * 4));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(victim_lds_wide<uint2, false>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_words * 4));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(victim_lds_wide<uint4, false>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_words * 4));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(victim_lds_wide<uint2, true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_words * 4));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(victim_lds_wide<uint4, true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_words * 4));

    auto victim = [&](int kind) {
        if (kind == 0) hipLaunchKernelGGL(victim_regs, dim3(V), dim3(256), 0, main_s, out, 2000);
        else if (kind == 1) hipLaunchKernelGGL(victim_lds, dim3(V), dim3(256), lds_words * 4, main_s, out, 6, lds_words);
        else if (kind == 2) hipLaunchKernelGGL(victim_global, dim3(V), dim3(256), 0, main_s, out, src, n_src, 400);
        else if (kind == 3) hipLaunchKernelGGL(victim_wave_lds, dim3(V), dim3(256), lds_words * 4, main_s, out, 12, lds_words / 4);
        else if (kind == 4) hipLaunchKernelGGL(victim_f32, dim3(V), dim3(256), 0, main_s, out, 2000);
        else if (kind == 5) hipLaunchKernelGGL(victim_pk_f32, dim3(V), dim3(256), 0, main_s, out, 2000);
        else if (kind == 6) hipLaunchKernelGGL(victim_trans, dim3(V), dim3(256), 0, main_s, out, 1000);
        else if (kind == 7) hipLaunchKernelGGL(victim_sincos_precise, dim3(V), dim3(256), 0, main_s, out, 300);
        else if (kind == 8) hipLaunchKernelGGL(victim_crosslane, dim3(V), dim3(256), 0, main_s, out, 2000);
        else if (kind == 9) hipLaunchKernelGGL(victim_f16, dim3(V), dim3(256), 0, main_s, out, 2000);
        else if (kind == 10) hipLaunchKernelGGL(victim_mfma, dim3(V), dim3(256), 0, main_s, out, 2000);
        else if (kind == 11) hipLaunchKernelGGL((victim_lds_wide<uint2, false>), dim3(V), dim3(256), lds_words * 4, main_s, out, 8, lds_words / 2);
        else if (kind == 12) hipLaunchKernelGGL((victim_lds_wide<uint4, false>), dim3(V), dim3(256), lds_words * 4, main_s, out, 8, lds_words / 4);
        else if (kind == 13) hipLaunchKernelGGL((victim_lds_wide<uint2, true>), dim3(V), dim3(256), lds_words * 4, main_s, out, 8, lds_words / 2);
        else if (kind == 14) hipLaunchKernelGGL((victim_lds_wide<uint4, true>), dim3(V), dim3(256), lds_words * 4, main_s, out, 8, lds_words / 4);
#ifdef WITH_LIBRARY_FFT
        else if (kind == 15) hipLaunchKernelGGL(victim_fft<0>, dim3(V), dim3(256), fft_lds, main_s, out, 24);
        else if (kind == 16) hipLaunchKernelGGL(victim_fft<1>, dim3(V), dim3(256), fft_lds, main_s, out, 48);
        else if (kind == 17) hipLaunchKernelGGL(victim_fft<2>, dim3(V), dim3(256), fft_lds, main_s, out, 48);
        else if (kind == 18) hipLaunchKernelGGL(victim_fft<3>, dim3(V), dim3(256), fft_lds, main_s, out, 48);
#endif
        else if (kind >= 19) {
            switch (kind - 19) {
#define FORM_CASE(f) case f: hipLaunchKernelGGL(victim_pk_form<f>, dim3(V), dim3(256), 0, main_s, out, 1500); break;
                FORM_CASE(0) FORM_CASE(1) FORM_CASE(2) FORM_CASE(3) FORM_CASE(4) FORM_CASE(5) FORM_CASE(6) FORM_CASE(7) FORM_CASE(8) FORM_CASE(9) FORM_CASE(10) FORM_CASE(11)
                FORM_CASE(12) FORM_CASE(13) FORM_CASE(14) FORM_CASE(15) FORM_CASE(16) FORM_CASE(17) FORM_CASE(18) FORM_CASE(19) FORM_CASE(20)
            }
        }
        CK(hipGetLastError());
    };
    auto hog = [&](int mode) {
        if (mode == 0) hipLaunchKernelGGL(neighbour<0>, dim3(n_wg), dim3(256), 0, side_s, sink, hog_iters);
        else if (mode == 1) hipLaunchKernelGGL(neighbour<1>, dim3(n_wg), dim3(256), 0, side_s, sink, hog_iters);
        else hipLaunchKernelGGL(neighbour<2>, dim3(n_wg), dim3(256), 0, side_s, sink, hog_iters);
        CK(hipGetLastError());
    };
    const char *vname[40] = {"registers only (integer)", "LDS round trips with barriers", "global loads", "LDS exchanges inside one wave", "fp32 fma chain", "packed fp32 fma",
                             "fast sin/cos/rsq/exp2", "library sincosf (large arguments)", "bpermute / DPP / readlane", "packed fp16 + dot2", "MFMA on register operands",
                             "LDS 8-byte accesses", "LDS 16-byte accesses", "LDS 8-byte accesses, two far slots each", "LDS 16-byte accesses, two far slots each",
                             "this library's wave-level 1024-point transform", "its register butterflies alone", "butterflies + lane swaps", "butterflies + the LDS exchange",
                             "v_pk_add_f32", "v_pk_add_f32 neg_hi", "v_pk_add_f32 crossed op_sel", "v_pk_add_f32 crossed op_sel + neg_hi", "v_pk_mul_f32 op_sel [1,1]/[1,0]",
                             "v_pk_fma_f32 op_sel_hi [0,1,1] neg_lo", "v_pk_fma_f32", "v_pk_mul_f32",
                             "v_pk_add_f32 op_sel_hi:[1,0]", "v_pk_add_f32 op_sel:[1,0]", "v_pk_add_f32 op_sel:[0,1]", "v_pk_add_f32 op_sel_hi:[0,1]", "v_pk_add_f32 op_sel:[1,0] op_sel_hi:[0,1]",
                             "v_pk_fma_f32 b, 1.0, a crossed b, neg_hi (a - jb)", "v_pk_fma_f32 b, 1.0, a crossed b, neg_lo (a + jb)", "v_pk_mul_f32 crossed src1 (two registers)",
                             "v_pk_fma_f32 crossed src1 (two registers)", "v_pk_fma_f32 crossed src2", "complex product, crossings on src0 and src2 only", "v_pk_add_f16 crossed src1", "v_pk_mov_b32 swap (src0 hi, src1 lo)"};
    const char *hname[3] = {"LDS reads feed the MFMAs", "MFMAs on registers + LDS reads summed by vector adds", "LDS reads only"};
    std::vector<unsigned> ref((size_t)V * 256), got((size_t)V * 256);
    for (int kind = first_kind; kind < n_kinds; ++kind) {
#ifndef WITH_LIBRARY_FFT
        if (kind >= 15 && kind < 19) continue;
#endif
        CK(hipMemsetAsync(out, 0, (size_t)V * 256 * 4, main_s));
        victim(kind);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(ref.data(), out, ref.size() * 4, hipMemcpyDeviceToHost));
        CK(hipMemsetAsync(out, 0, (size_t)V * 256 * 4, main_s));
        victim(kind);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(got.data(), out, got.size() * 4, hipMemcpyDeviceToHost));
        std::printf("victim: %-32s alone twice: %s\n", vname[kind], std::memcmp(ref.data(), got.data(), ref.size() * 4) ? "DIFFERENT" : "identical");
        for (int mode = 0; mode < 3; ++mode)
            for (int rep = 0; rep < 2; ++rep) {
                CK(hipMemsetAsync(out, 0, (size_t)V * 256 * 4, main_s));
                CK(hipDeviceSynchronize());
                hipEvent_t e0, e1;
                CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
                hog(mode);
                CK(hipEventRecord(e0, main_s));
                victim(kind);
                CK(hipEventRecord(e1, main_s));
                CK(hipEventSynchronize(e1));
                const bool hog_still_running = hipStreamQuery(side_s) == hipErrorNotReady;
                CK(hipDeviceSynchronize());
                float ms = 0.f;
                CK(hipEventElapsedTime(&ms, e0, e1));
                CK(hipMemcpy(got.data(), out, got.size() * 4, hipMemcpyDeviceToHost));
                size_t bad_words = 0, bad_wgs = 0;
                for (int w = 0; w < V; ++w) {
                    size_t b = 0;
                    for (int t = 0; t < 256; ++t) b += got[(size_t)w * 256 + t] != ref[(size_t)w * 256 + t];
                    bad_words += b; bad_wgs += b != 0;
                }
                std::printf("  neighbour: %-52s run %d: victim %.3f ms (neighbour %s when it ended), wrong words %zu in %zu of %d workgroups\n",
                            hname[mode], rep, ms, hog_still_running ? "still running" : "ALREADY DONE", bad_words, bad_wgs, V);
                CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
            }
    }
    return 0;
}
#endif
