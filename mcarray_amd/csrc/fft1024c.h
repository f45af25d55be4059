// fft1024c.h -- wave-level 1024-point COMPLEX FFT for gfx950: one 64-lane wave, 16 points per lane.
//
// Two real 1024-sample frames (two microphones) ride in one complex transform, z = x_a + j x_b, so a wave analyses a
// channel pair per pass and no split step exists; the consumers either never separate the two spectra (delay-and-sum:
// y = Re IDFT(sum_p Z_p T_p), see k_beamform_wave) or separate them per bin from Z[k] and Z[1024 - k].
//
// 1024 = 16 x 4 x 16, decimation in frequency; ONE exchange goes through LDS, the other stays in registers:
//   in   lane l holds z[l + 64 i], i = 0..15                         n = 64 i + l,  l = 16 j + m  (j = lane row, m = lane & 15)
//   A    16-point DFT over i -> kA (registers);  twiddle W1024^(l kA)
//   SW   4 x 4 transposes between the lane ROW j (lane bits 4, 5) and the two high bits of kA by
//        v_permlane32_swap / v_permlane16_swap -- 32 full-rate vector instructions, no LDS, no round trip
//   B    4-point DFT over j -> kB;  twiddle W64^(m kB)  (3 per lane, kept in registers)
//   EX   through the wave's own LDS scratch: element (kA, kB, m) at word 18 (kA + 16 kB) + m; lane (kA + 16 kB) reads its
//        16 values m = 0..15 as 8 ds_read_b128 (rows of 18 words: reads and writes are bank-conflict free)
//   C    16-point DFT over m -> kC
//   out  lane (k & 63) holds Z[k], k = lane + 64 kC
// Input and output have the same distribution (index = lane + 64 * register), so the inverse is the same routine with
// conjugated twiddles.  Register order: the 16-point kernel leaves index dr16(p) = (p >> 2) + 4 (p & 3) in register p, so
// on return v[p] = Z[lane + 64 dr16(p)] (dr16 is its own inverse; the callers index at compile time).
// LDS traffic per transform: 16 ds_write_b64 + 8 ds_read_b128 + the 15 stage-A twiddles (8 ds_read_b128), about half of a
// two-exchange layout -- on this kernel family the LDS, not the vector ALU, is the first pipe to fill.
//
// Replaces the FFT inside DSPONE's dsp::STFT (call site SourceSeparationAndLocalisation.cpp:52); conventions per
// unnormalised forward, the caller applies 1/N after the inverse.
#pragma once
#include "fft512.h"

namespace mca {

constexpr int F1K_ROW = 18;                 // words (float2) per row of the exchange and of the twiddle table
constexpr int F1K_SCRATCH = 64 * F1K_ROW;   // float2 words of LDS scratch per wave
constexpr int F1K_TWORDS = 64 * F1K_ROW;    // float2 words of the twiddle table: [lane][q] = W1024^(lane q), q = 0..15 (shared by a workgroup)

// all threads of the block; the caller synchronises afterwards
__device__ __forceinline__ void f1k_table_init(float2 *tab, int tid, int nthreads)
{
    for (int e = tid; e < 1024; e += nthreads) tab[(e >> 4) * F1K_ROW + (e & 15)] = twiddle(((e >> 4) * (e & 15)) & 1023, 1024, false);
}

// a - j b and a + j b as one packed instruction each: b * 1 + a with the halves of b swapped and neg_* supplying the sign (the same bits
// as an add).  An fma because b has to ride in src0: v_pk_add_f32 would need the high half of src1 in the low result, the operand
// path that is not sound beside an MFMA + LDS neighbour (fft512.h, RULE).
__device__ __forceinline__ float2 csub_jb(float2 a, float2 b)   // (a.x + b.y, a.y - b.x)
{
    v2f av = to_v2f(a), bv = to_v2f(b), r;
    asm("v_pk_fma_f32 %0, %2, 1.0, %1 op_sel:[1,0,0] op_sel_hi:[0,0,1] neg_hi:[1,0,0]" : "=v"(r) : "v"(av), "v"(bv));
    return from_v2f(r);
}
__device__ __forceinline__ float2 cadd_jb(float2 a, float2 b)   // (a.x - b.y, a.y + b.x)
{
    v2f av = to_v2f(a), bv = to_v2f(b), r;
    asm("v_pk_fma_f32 %0, %2, 1.0, %1 op_sel:[1,0,0] op_sel_hi:[0,0,1] neg_lo:[1,0,0]" : "=v"(r) : "v"(av), "v"(bv));
    return from_v2f(r);
}
__device__ __forceinline__ float2 cscale(float2 a, float s)
{
    v2f av = to_v2f(a), sv = {s, s}, r;
    asm("v_pk_mul_f32 %0, %1, %2" : "=v"(r) : "v"(av), "v"(sv));
    return from_v2f(r);
}
// -j a = (a.y, -a.x),  +j a = (-a.y, a.x)
__device__ __forceinline__ float2 cmul_mj(float2 a)
{
    v2f av = to_v2f(a), one = {1.f, 1.f}, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[0,1] neg_hi:[1,0]" : "=v"(r) : "v"(av), "v"(one));
    return from_v2f(r);
}
__device__ __forceinline__ float2 cmul_pj(float2 a)
{
    v2f av = to_v2f(a), one = {1.f, 1.f}, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[0,1] neg_lo:[1,0]" : "=v"(r) : "v"(av), "v"(one));
    return from_v2f(r);
}

// 4-point DFT in place: (a0..a3) = x[0..3] -> X[0..3]
template <bool INV>
__device__ __forceinline__ void bfly4(float2 &a0, float2 &a1, float2 &a2, float2 &a3)
{
    const float2 t0 = cadd(a0, a2), t1 = csub(a0, a2), t2 = cadd(a1, a3), t3 = csub(a1, a3);
    a0 = cadd(t0, t2);
    a2 = csub(t0, t2);
    if (!INV) { a1 = csub_jb(t1, t3); a3 = cadd_jb(t1, t3); }
    else { a1 = cadd_jb(t1, t3); a3 = csub_jb(t1, t3); }
}

// position p of the result of fft16 holds output index dr16(p)
__device__ __forceinline__ constexpr int dr16(int p) { return (p >> 2) + 4 * (p & 3); }

// In-register 16-point DFT (radix 4 x 4, decimation in frequency).  On return v[p] = X[dr16(p)].
template <bool INV>
__device__ __forceinline__ void fft16(float2 (&v)[16])
{
    constexpr float R = 0.70710678118654752440f, C1 = 0.92387953251128675613f, S1 = 0.38268343236508977173f;
#pragma unroll
    for (int j = 0; j < 4; ++j) bfly4<INV>(v[j], v[j + 4], v[j + 8], v[j + 12]);
    // twiddles W16^(j q) on v[j + 4 q]:  W^2 = R (1 - j), W^6 = -R (1 + j), W^4 = -j
    const float2 w1 = make_float2(C1, -S1), w3 = make_float2(S1, -C1), w9 = make_float2(-C1, S1);
    if (!INV) {
        v[5] = cmul(v[5], w1);  v[9] = cscale(csub_jb(v[9], v[9]), R);   v[13] = cmul(v[13], w3);
        v[6] = cscale(csub_jb(v[6], v[6]), R);  v[10] = cmul_mj(v[10]);   v[14] = cscale(cadd_jb(v[14], v[14]), -R);
        v[7] = cmul(v[7], w3);  v[11] = cscale(cadd_jb(v[11], v[11]), -R); v[15] = cmul(v[15], w9);
    } else {
        v[5] = cmulc(v[5], w1); v[9] = cscale(cadd_jb(v[9], v[9]), R);   v[13] = cmulc(v[13], w3);
        v[6] = cscale(cadd_jb(v[6], v[6]), R);  v[10] = cmul_pj(v[10]);   v[14] = cscale(csub_jb(v[14], v[14]), -R);
        v[7] = cmulc(v[7], w3); v[11] = cscale(csub_jb(v[11], v[11]), -R); v[15] = cmulc(v[15], w9);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) bfly4<INV>(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
}

// 1024-point complex DFT of v[i] = z[lane + 64 i] through buf (this wave's scratch, F1K_SCRATCH words); on return
// v[s] = Z[lane + 64 s].  INV: conjugated twiddles (unnormalised inverse).  tab: the table of f1k_table_init.
// 16 consecutive words of p (16-byte aligned) as eight ds_read_b128 and their wait, in one statement (the compiler would
// otherwise wait for every read right in front of its first use)
__device__ __forceinline__ void lds_read16_b128(float2 (&v)[16], const float2 *p)
{
    typedef float v4f __attribute__((ext_vector_type(4)));
    v4f a[8];
    asm volatile(
        "ds_read_b128 %0, %8\n\tds_read_b128 %1, %8 offset:16\n\tds_read_b128 %2, %8 offset:32\n\tds_read_b128 %3, %8 offset:48\n\t"
        "ds_read_b128 %4, %8 offset:64\n\tds_read_b128 %5, %8 offset:80\n\tds_read_b128 %6, %8 offset:96\n\tds_read_b128 %7, %8 offset:112\n\t"
        "s_waitcnt lgkmcnt(0)"
        : "=&v"(a[0]), "=&v"(a[1]), "=&v"(a[2]), "=&v"(a[3]), "=&v"(a[4]), "=&v"(a[5]), "=&v"(a[6]), "=&v"(a[7])
        : "v"((unsigned)(uintptr_t)p)
        : "memory");
#pragma unroll
    for (int i = 0; i < 8; ++i) { v[2 * i] = make_float2(a[i].x, a[i].y); v[2 * i + 1] = make_float2(a[i].z, a[i].w); }
}

// lanes 32..63 of a <-> lanes 0..31 of b;  odd rows (of 16 lanes) of a <-> even rows of b
__device__ __forceinline__ void swap_rows32(float &a, float &b)
{
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    a = __uint_as_float(r[0]); b = __uint_as_float(r[1]);
}
__device__ __forceinline__ void swap_rows16(float &a, float &b)
{
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    a = __uint_as_float(r[0]); b = __uint_as_float(r[1]);
}
// x[r] in lane row j  <->  x[j] in lane row r
__device__ __forceinline__ void transpose_rows4(float2 &x0, float2 &x1, float2 &x2, float2 &x3)
{
    swap_rows32(x0.x, x2.x); swap_rows32(x0.y, x2.y); swap_rows32(x1.x, x3.x); swap_rows32(x1.y, x3.y);
    swap_rows16(x0.x, x1.x); swap_rows16(x0.y, x1.y); swap_rows16(x2.x, x3.x); swap_rows16(x2.y, x3.y);
}

struct F1kNoMid { __device__ __forceinline__ void operator()() const {} };

// Per-lane constants of the transform: the three stage-B twiddles W64^(m kB), m = lane & 15.
struct F1kLane {
    float2 wb[4];
    float2 t1[16];     // (fft1024c OPT bit 2) the stage-A twiddles W1024^(lane q) kept in registers: no table read per transform
    __device__ __forceinline__ void init(int lane)
    {
#pragma unroll
        for (int q = 1; q < 4; ++q) wb[q] = twiddle(((lane & 15) * q) & 63, 64, false);
    }
    __device__ __forceinline__ void load_t1(const float2 *tab, int lane)
    {
#pragma unroll
        for (int q = 0; q < 16; ++q) t1[q] = tab[lane * 18 + q];
    }
};

// 1024-point complex DFT of v[i] = z[lane + 64 i] through buf (this wave's scratch, F1K_SCRATCH words); on return
// v[p] = Z[lane + 64 dr16(p)].  INV: conjugated twiddles (unnormalised inverse).  tab: the table of f1k_table_init.
// mid(): called right behind the exchange's stores -- the place where the fewest registers are live; the callers issue
// global loads there.
template <bool INV, int OPT = 3, typename Mid = F1kNoMid>
__device__ __forceinline__ void fft1024c(float2 (&v)[16], float2 *buf, int lane, const float2 *tab, const F1kLane &lc, Mid mid = Mid(), int row = -1)
{
    // row: the residue k mod 64 this lane receives (default: its own number); any permutation of the lanes reads conflict
    // free as long as every ds_read_b128 lane group sees 16 different rows mod 16
    constexpr bool SCHED = OPT & 2, T1REG = OPT & 4;
    // the stage's twiddles are read ahead of its butterflies (T1REG: they live in registers)
    float2 tw[16];
    if (!T1REG) {
        lds_read16_b128(tw, tab + lane * F1K_ROW);
        if (SCHED) __builtin_amdgcn_sched_barrier(0);
    }
    fft16<INV>(v);
#pragma unroll
    for (int p = 1; p < 16; ++p) {
        const float2 w = T1REG ? lc.t1[dr16(p)] : tw[dr16(p)];
        v[p] = INV ? cmulc(v[p], w) : cmul(v[p], w);
    }
    // register p = 4 q + r holds kA = q + 4 r: the lane row takes r
#pragma unroll
    for (int q = 0; q < 4; ++q) transpose_rows4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
#pragma unroll
    for (int q = 0; q < 4; ++q) bfly4<INV>(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
    // register 4 q + kB, lane (m, row): kA = q + 4 row -> word 18 (kA + 16 kB) + m
    float2 *wr = buf + (lane >> 4) * (4 * F1K_ROW) + (lane & 15);
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
            if (kb != 0) v[4 * q + kb] = INV ? cmulc(v[4 * q + kb], lc.wb[kb]) : cmul(v[4 * q + kb], lc.wb[kb]);
            wr[(q + 16 * kb) * F1K_ROW] = v[4 * q + kb];
        }
    wave_lds_fence();
    mid();
    if (SCHED) __builtin_amdgcn_sched_barrier(0);
    lds_read16_b128(v, buf + (row < 0 ? lane : row) * F1K_ROW);
    wave_lds_fence();
    fft16<INV>(v);
}

}  // namespace mca
