// fft512.h -- wave-level 1024-point real FFT / inverse for gfx950 (one 64-lane wave per transform).
//
// A 1024-point real transform is done as a 512-point complex transform of the packed
// sequence z[m] = x[2m] + j x[2m+1] plus a split step.  512 = 8*8*8: every lane owns 8
// complex points and does three radix-8 butterflies in registers; between them the wave
// exchanges data through its own LDS scratch (two transposes).  The LDS layouts
// (row stride 72 complex words in the first exchange, q + 66*l0 + 8*s in the second) make every
// ds_write_b64 / ds_read_b64 of the exchanges bank-conflict free (MI355X LDS: 64 banks x 4 B;
// b64 reads are serviced in two 32-lane groups, b64 writes in four 16-lane groups).
//
// Replaces the FFT that DSPONE's dsp::STFT performs before handing frames to
// processParametrisation (call sites SourceSeparationAndLocalisation.cpp:52,
// FastBinauralMasking.cpp:57); conventions per SURVEY A.1: unnormalised forward,
// 1/N inverse, CCS bin order k = 0..N/2.
#pragma once
#include <hip/hip_runtime.h>

namespace mca {

constexpr int FFT_N = 1024;          // real transform length
constexpr int FFT_H = 512;           // complex half-length
constexpr int FFT_K = 513;           // one-sided bins
constexpr int FFT_SCRATCH = 576;     // float2 words of LDS scratch per wave (8 rows x 72)
constexpr int FFT_ROW = 72;

// Complex products as two packed-fp32 instructions each.  v_pk_mul/fma_f32 work on a VGPR pair {lo, hi};
// op_sel / op_sel_hi pick which half of every source feeds the low / high result and neg_lo / neg_hi
// negate it, so the swaps and the one-sided sign of a complex multiply cost nothing (the compiler
// otherwise emits pk_mul + 2 pk_fma + v_mov for the same thing).
// RULE (tools/probes/coresidency_standalone.hip): no instruction here lets the LOW result take the HIGH half of
// src1 (op_sel bit 1).  On this machine that one operand path returns a wrong value now and then while a wave of ANOTHER kernel on
// the same SIMD mixes MFMA with LDS reads; the high half of src0 or src2 in the low result, and any low half in the high result,
// are sound.  A product of two high halves therefore lands in the HIGH lane and crosses over through src2 of the second fma.
typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2f to_v2f(float2 a) { v2f r = {a.x, a.y}; return r; }
__device__ __forceinline__ float2 from_v2f(v2f a) { return make_float2(a.x, a.y); }

__device__ __forceinline__ float2 cmul(float2 a, float2 b)   // (a.x b.x - a.y b.y, a.x b.y + a.y b.x)
{
    v2f av = to_v2f(a), bv = to_v2f(b), t, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,1]" : "=v"(t) : "v"(av), "v"(bv));                                  // (a.y b.x, a.y b.y)
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,1] op_sel_hi:[0,1,0] neg_lo:[0,0,1]" : "=v"(r) : "v"(av), "v"(bv), "v"(t));   // (a.x b.x - t.hi, a.x b.y + t.lo)
    return from_v2f(r);
}
__device__ __forceinline__ float2 cmulc(float2 a, float2 b)  // a * conj(b) = (a.x b.x + a.y b.y, a.y b.x - a.x b.y)
{
    v2f av = to_v2f(a), bv = to_v2f(b), t, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,1]" : "=v"(t) : "v"(bv), "v"(av));                                  // (b.y a.x, b.y a.y)
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,1] op_sel_hi:[1,0,0] neg_hi:[0,0,1]" : "=v"(r) : "v"(av), "v"(bv), "v"(t));   // (a.x b.x + t.hi, a.y b.x - t.lo)
    return from_v2f(r);
}
// the accumulating forms: the first fma takes the terms with b.y (b rides in src0, its high half may go anywhere), real part in the
// HIGH lane; the second adds the terms with b.x and takes the first result crossed through src2
__device__ __forceinline__ float2 cmac(float2 acc, float2 a, float2 b)   // acc + a * b
{
    v2f av = to_v2f(a), bv = to_v2f(b), cv = to_v2f(acc), u, r;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,1] op_sel_hi:[1,1,0] neg_hi:[1,0,0]" : "=v"(u) : "v"(bv), "v"(av), "v"(cv));  // (c.y + a.x b.y, c.x - a.y b.y)
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,1] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(av), "v"(bv), "v"(u));                  // (a.x b.x + u.hi, a.y b.x + u.lo)
    return from_v2f(r);
}
__device__ __forceinline__ float2 cmacc(float2 acc, float2 a, float2 b)  // acc + a * conj(b)
{
    v2f av = to_v2f(a), bv = to_v2f(b), cv = to_v2f(acc), u, r;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,1] op_sel_hi:[1,1,0] neg_lo:[1,0,0]" : "=v"(u) : "v"(bv), "v"(av), "v"(cv));  // (c.y - a.x b.y, c.x + a.y b.y)
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,1] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(av), "v"(bv), "v"(u));                  // (a.x b.x + u.hi, a.y b.x + u.lo)
    return from_v2f(r);
}
__device__ __forceinline__ float2 cnmac(float2 acc, float2 a, float2 b)   // acc - a * b
{
    v2f av = to_v2f(a), bv = to_v2f(b), cv = to_v2f(acc), u, r;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,1] op_sel_hi:[1,1,0] neg_lo:[1,0,0]" : "=v"(u) : "v"(bv), "v"(av), "v"(cv));                  // (c.y - a.x b.y, c.x + a.y b.y)
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,1] op_sel_hi:[1,0,0] neg_lo:[1,0,0] neg_hi:[1,0,0]" : "=v"(r) : "v"(av), "v"(bv), "v"(u));    // (u.hi - a.x b.x, u.lo - a.y b.x)
    return from_v2f(r);
}
__device__ __forceinline__ float2 cnmacc(float2 acc, float2 a, float2 b)  // acc - a * conj(b)
{
    v2f av = to_v2f(a), bv = to_v2f(b), cv = to_v2f(acc), u, r;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,1] op_sel_hi:[1,1,0] neg_hi:[1,0,0]" : "=v"(u) : "v"(bv), "v"(av), "v"(cv));                  // (c.y + a.x b.y, c.x - a.y b.y)
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,1] op_sel_hi:[1,0,0] neg_lo:[1,0,0] neg_hi:[1,0,0]" : "=v"(r) : "v"(av), "v"(bv), "v"(u));    // (u.hi - a.x b.x, u.lo - a.y b.x)
    return from_v2f(r);
}
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ float2 cconj(float2 a) { return make_float2(a.x, -a.y); }

// exp(-j*2*pi*num/den) for forward (INV = false), conjugate for inverse.
__device__ __forceinline__ float2 twiddle(int num, int den, bool inv)
{
    float s, c;
    sincospif(2.0f * (float)num / (float)den, &s, &c);
    return make_float2(c, inv ? s : -s);
}

// In-register 8-point DFT (decimation in frequency).  Output index of register i is
// bitrev3(i): {0,4,2,6,1,5,3,7}.
template <bool INV>
__device__ __forceinline__ void fft8(float2 (&v)[8])
{
    constexpr float R = 0.70710678118654752440f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        float2 a = cadd(v[r], v[r + 4]);
        float2 b = csub(v[r], v[r + 4]);
        v[r] = a;
        if (r == 0) v[r + 4] = b;
        else if (r == 1) v[r + 4] = INV ? make_float2((b.x - b.y) * R, (b.x + b.y) * R)
                                        : make_float2((b.x + b.y) * R, (b.y - b.x) * R);
        else if (r == 2) v[r + 4] = INV ? make_float2(-b.y, b.x) : make_float2(b.y, -b.x);
        else v[r + 4] = INV ? make_float2((-b.x - b.y) * R, (b.x - b.y) * R)
                            : make_float2((b.y - b.x) * R, (-b.x - b.y) * R);
    }
#pragma unroll
    for (int h = 0; h < 8; h += 4) {
        float2 a0 = cadd(v[h], v[h + 2]), b0 = csub(v[h], v[h + 2]);
        float2 a1 = cadd(v[h + 1], v[h + 3]), b1 = csub(v[h + 1], v[h + 3]);
        v[h] = a0; v[h + 1] = a1; v[h + 2] = b0;
        v[h + 3] = INV ? make_float2(-b1.y, b1.x) : make_float2(b1.y, -b1.x);
    }
#pragma unroll
    for (int h = 0; h < 8; h += 2) {
        float2 a = cadd(v[h], v[h + 1]), b = csub(v[h], v[h + 1]);
        v[h] = a; v[h + 1] = b;
    }
}

__device__ __forceinline__ constexpr int br3(int i) { return ((i & 1) << 2) | (i & 2) | ((i >> 2) & 1); }

// Twiddle + window table shared by all waves of a workgroup, in LDS (read-only after init):
//   t1[q][lane] = W512^(lane*q)        q = 0..7   (index q*64 + lane)
//   t2[s][l0]   = W64^(l0*s)           s, l0 = 0..7 (index 512 + s*8 + l0)
//   ts[i][lane] = W1024^(lane + 64 i)  i = 0..3   (index 576 + i*64 + lane)
//   win[r][lane] = (w[2m], w[2m+1]) / 2, m = lane + 64 r (index 832 + r*64 + lane), periodic Hann
// Forward values; the inverse transform uses their conjugates.  Every read is lane-contiguous
// (conflict-free) or a broadcast.  Keeping them here instead of in registers frees ~58 VGPRs per
// lane, which is what lets two 512-thread workgroups share a CU.
constexpr int TW_T1 = 0, TW_T2 = 512, TW_TS = 576, TW_WIN = 832, TW_WORDS = 1344;   // float2 words

struct FftTw {
    const float2 *tab;
    __device__ __forceinline__ float2 t1(int q, int lane) const { return tab[TW_T1 + q * 64 + lane]; }
    __device__ __forceinline__ float2 t2(int s, int lane) const { return tab[TW_T2 + s * 8 + (lane & 7)]; }
    __device__ __forceinline__ float2 ts(int i, int lane) const { return tab[TW_TS + i * 64 + lane]; }
    __device__ __forceinline__ float2 win(int r, int lane) const { return tab[TW_WIN + r * 64 + lane]; }
};

// fill the table (all threads of the block; caller must __syncthreads() afterwards).  window == nullptr: twiddles
// only (TW_WIN words; the caller keeps its window samples in registers).
__device__ __forceinline__ void fft_table_init(float2 *tab, const float *window, int tid, int nthreads)
{
    for (int e = tid; e < TW_WIN; e += nthreads) {
        float2 v;
        if (e < TW_T2) { const int q = e >> 6, l = e & 63; v = twiddle((l * q) & 511, 512, false); }
        else if (e < TW_TS) { const int s = (e - TW_T2) >> 3, l0 = (e - TW_T2) & 7; v = twiddle((l0 * s) & 63, 64, false); }
        else { const int i = (e - TW_TS) >> 6, l = (e - TW_TS) & 63; v = twiddle(l + 64 * i, 1024, false); }
        tab[e] = v;
    }
    if (window)
        for (int e = tid; e < 512; e += nthreads) {   // half the window: carries the 1/2 of rfft1024's split step (exact in fp32)
            const float2 w = reinterpret_cast<const float2 *>(window)[e];
            tab[TW_WIN + e] = make_float2(0.5f * w.x, 0.5f * w.y);
        }
}

// wave-level ordering of LDS traffic: LDS instructions of one wave execute in order, so only
// the compiler has to be kept from reordering across the exchange points.
__device__ __forceinline__ void wave_lds_fence()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// 512-point complex FFT of v[r] = z[lane + 64 r] through buf (this wave's scratch, FFT_SCRATCH words).
// The result stays in registers with the same distribution as the input, register index bit-reversed:
// on return v[i] = Z[lane + 64 * br3(i)].
template <bool INV, bool AHEAD = false>
__device__ __forceinline__ void cfft512_regs(float2 (&v)[8], float2 *buf, int lane, const FftTw &tw)
{
    // stage A: DFT over r, twiddle W512^(lane*q), exchange 1: buf[q*72 + lane]
    // AHEAD: the twiddles of a stage are read ahead of its butterflies (the compiler cannot move a table read above the
    // exchange stores -- same address space -- so each read is otherwise waited for right before its product).  Worth
    // 2 % in k_stft_phat; k_beamform_ola, at 125 registers, loses as much.
    float2 twa[8];
    if (AHEAD) {
#pragma unroll
        for (int i = 1; i < 8; ++i) twa[i] = tw.t1(br3(i), lane);
    }
    fft8<INV>(v);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int q = br3(i);
        const float2 w = (q == 0) ? make_float2(1.f, 0.f) : (AHEAD ? twa[i] : tw.t1(q, lane));
        buf[q * FFT_ROW + lane] = (q == 0) ? v[i] : (INV ? cmulc(v[i], w) : cmul(v[i], w));
    }
    wave_lds_fence();
    const int qq = lane >> 3, l0 = lane & 7;
#pragma unroll
    for (int l1 = 0; l1 < 8; ++l1) v[l1] = buf[qq * FFT_ROW + l0 + 8 * l1];
    wave_lds_fence();
    // stage B: DFT over l1, twiddle W64^(l0*s), exchange 2: buf[q + 66*l0 + 8*s].  The layout makes the
    // 16-lane write groups (q pair x l0) hit 16 distinct bank pairs and turns the read-back into
    // contiguous rows: the reading lane is (s, q) = (lane >> 3, lane & 7), address lane + 66*l0.
    float2 twb[8];
    if (AHEAD) {
#pragma unroll
        for (int i = 1; i < 8; ++i) twb[i] = tw.t2(br3(i), lane);
    }
    fft8<INV>(v);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int s = br3(i);
        const float2 w = (s == 0) ? make_float2(1.f, 0.f) : (AHEAD ? twb[i] : tw.t2(s, lane));
        buf[qq + 66 * l0 + 8 * s] = (s == 0) ? v[i] : (INV ? cmulc(v[i], w) : cmul(v[i], w));
    }
    wave_lds_fence();
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = buf[lane + 66 * j];
    wave_lds_fence();
    // stage C: DFT over l0 -> t ; k = q + 8 s + 64 t = lane + 64 t
    fft8<INV>(v);
}

// Forward real FFT.  v[r] = (x[2m], x[2m+1]) windowed with HALF the analysis window (the table of
// fft_table_init carries the 1/2 of the split step), m = lane + 64 r.
// On return buf[k], k = 0..512, holds X[k] (buf must have >= FFT_SCRATCH words; word 512 is used).
template <bool AHEAD = false>
__device__ __forceinline__ void rfft1024(float2 (&v)[8], float2 *buf, int lane, const FftTw &tw)
{
    cfft512_regs<false, AHEAD>(v, buf, lane, tw);
    // split step on pairs (k, 512-k):  X[k] = a + W b,  X[512-k] = conj(a - W b),
    // a = Z[k] + conj Z[512-k],  b = -j (Z[k] - conj Z[512-k]),  W = W1024^k  (Z already halved).
    // Z[k], k = lane + 64 i < 256, is in this lane's registers; its partner lives in lane 64 - lane,
    // rows 4..7, so only those rows go through LDS.
#pragma unroll
    for (int i = 0; i < 8; ++i)
        if (br3(i) >= 4) buf[lane + 64 * br3(i)] = v[i];
    wave_lds_fence();
    float2 xk[4], xp[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int k = lane + 64 * i;
        const float2 zk = v[br3(i)];
        float2 zp = buf[(512 - k) & 511];
        if (k == 0) zp = zk;                                                    // Z[512] = Z[0]
        const float2 a = make_float2(zk.x + zp.x, zk.y - zp.y);
        const float2 d = make_float2(zk.x - zp.x, zk.y + zp.y);                 // Z[k] - conj Zp
        const float2 b = make_float2(d.y, -d.x);                                // -j d
        const float2 wb = cmul(tw.ts(i, lane), b);
        xk[i] = cadd(a, wb);
        xp[i] = cconj(csub(a, wb));
    }
    const float2 z256 = v[br3(4)];                                              // lane 0: Z[256]
    wave_lds_fence();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int k = lane + 64 * i;
        buf[k] = xk[i];
        buf[512 - k] = xp[i];      // k = 0 writes X[512]
    }
    if (lane == 0) {
        // k = 256: Zp = Zk, W = -j:  a = (2 Re z, 0), b = (2 Im z, 0); X = a + W b = 2 (Re z, -Im z)
        buf[256] = make_float2(2.f * z256.x, -2.f * z256.y);
    }
    wave_lds_fence();
}

// Inverse real FFT.  buf[k], k = 0..512 holds X[k] (imaginary parts of X[0] and X[512] are
// ignored, like a CCS->real inverse).  On return buf[m] = (x[2m], x[2m+1]), m = 0..511,
// scaled by 1/1024 overall.
__device__ __forceinline__ void irfft1024(float2 *buf, int lane, const FftTw &tw)
{
    // Z[k] = a + j b, Z[512-k] = conj(a) + j conj(b) with a = X[k] + conj X[512-k],
    // b = conj(W1024^k) (X[k] - conj X[512-k]); the 1/2 of both is folded into the final scale.
    // Z[k], k = lane + 64 i < 256, stays in this lane's registers; Z[512-k] belongs to lane 64 - lane.
    float2 v[8], zp[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int k = lane + 64 * i;
        float2 xk = buf[k];
        float2 xp = buf[512 - k];
        if (k == 0) { xk.y = 0.f; xp.y = 0.f; }
        const float2 a = make_float2(xk.x + xp.x, xk.y - xp.y);
        const float2 d = make_float2(xk.x - xp.x, xk.y + xp.y);
        const float2 b = cmulc(d, tw.ts(i, lane));
        v[i] = make_float2(a.x - b.y, a.y + b.x);           // a + j b
        zp[i] = make_float2(a.x + b.y, -a.y + b.x);         // conj(a) + j conj(b)
    }
    const float2 x256 = buf[256];
    wave_lds_fence();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int k = lane + 64 * i;
        if (k != 0) buf[512 - k] = zp[i];
    }
    if (lane == 0) {
        // k = 256: a = (2 Re x, 0), d = (0, 2 Im x), b = conj(W) d = (-2 Im x, 0); Z = a + j b = 2 (Re x, -Im x)
        buf[256] = make_float2(2.f * x256.x, -2.f * x256.y);
    }
    wave_lds_fence();
#pragma unroll
    for (int r = 4; r < 8; ++r) v[r] = buf[lane + 64 * r];
    wave_lds_fence();
    cfft512_regs<true>(v, buf, lane, tw);
    const float sc = 1.0f / 1024.0f;
#pragma unroll
    for (int i = 0; i < 8; ++i) buf[lane + 64 * br3(i)] = make_float2(v[i].x * sc, v[i].y * sc);
    wave_lds_fence();
}

// ---- 512-sample real frames, two per 512-point complex transform --------------------------------------------------
constexpr int N512_H = 256, N512_K = 257, N512_ROW = 258;

// v[r] = (a[m], b[m]) * w[m] / 2, m = lane + 64 r.  On return specA[k], specB[k], k = 0..256.
__device__ __forceinline__ void rfft512_pair(float2 (&v)[8], float2 *buf, float2 *specA, float2 *specB, int lane, const FftTw &tw)
{
    cfft512_regs<false>(v, buf, lane, tw);
#pragma unroll
    for (int i = 0; i < 8; ++i) buf[lane + 64 * br3(i)] = v[i];
    wave_lds_fence();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int k = lane + 64 * i;
        const float2 zk = buf[k], zp = buf[(512 - k) & 511];
        const float2 d = make_float2(zk.x - zp.x, zk.y + zp.y);                 // Z[k] - conj Z[512-k]
        specA[k] = make_float2(zk.x + zp.x, zk.y - zp.y);
        specB[k] = make_float2(d.y, -d.x);                                      // -j d
    }
    if (lane == 0) {
        const float2 z = buf[256];
        specA[256] = make_float2(2.f * z.x, 0.f);
        specB[256] = make_float2(2.f * z.y, 0.f);
    }
    wave_lds_fence();
}

// inverse of two one-sided spectra Ya, Yb (k = 0..256; imaginary parts of k = 0 and 256 ignored, like a CCS inverse):
// on return v[i] = (ya[n], yb[n]), n = lane + 64 br3(i), scaled by 1/512.
__device__ __forceinline__ void irfft512_pair(const float2 *Ya, const float2 *Yb, float2 *buf, float2 (&v)[8], int lane, const FftTw &tw)
{
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int k = lane + 64 * i;
        float2 a = Ya[k], b = Yb[k];
        if (k == 0) { a.y = 0.f; b.y = 0.f; }
        buf[k] = make_float2(a.x - b.y, a.y + b.x);                             // Ya + j Yb
        if (k != 0) buf[512 - k] = make_float2(a.x + b.y, -a.y + b.x);          // conj Ya + j conj Yb
    }
    if (lane == 0) buf[256] = make_float2(Ya[256].x, Yb[256].x);
    wave_lds_fence();
#pragma unroll
    for (int r = 0; r < 8; ++r) v[r] = buf[lane + 64 * r];
    wave_lds_fence();
    cfft512_regs<true>(v, buf, lane, tw);
    const float sc = 1.0f / 512.0f;
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = make_float2(v[i].x * sc, v[i].y * sc);
}

// windowed samples of channels c0, c0 + 1 (zeros beyond M) of the frame that starts at sample `start`
__device__ __forceinline__ void load_pair_512(float2 (&v)[8], const float *base, long long mic_stride, int c0, int M, long long start,
                                              const float (&wreg)[8], int lane)
{
    const float *pa = base + (long long)c0 * mic_stride + start, *pb = pa + mic_stride;
    const bool hb = c0 + 1 < M;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const float xa = pa[lane + 64 * r], xb = hb ? pb[lane + 64 * r] : 0.f;
        v[r] = make_float2(xa * wreg[r], xb * wreg[r]);
    }
}


// ---- 2048-sample real frames by decimation in time over four 512-sample sub-sequences x[4 n + r] ---------------------
// The sub-sequences are real, so two of them share one 512-point complex transform (rfft512_pair); then
//   X[k] = sum_r W^(r k) S_r[k mod 512],  W = exp(-j 2 pi / 2048),  k = 0..1024
// with the Hermitian extension S_r[m] = conj S_r[512 - m] for m > 256.  S: the four one-sided sub-spectra of a channel,
// N512_ROW words apart; tw2048[i] = exp(-j 2 pi i / 2048), i < 1024.
__device__ __forceinline__ float2 sub512_at(const float2 *S, int m) { return m <= 256 ? S[m] : cconj(S[512 - m]); }

// One radix-4 butterfly per m = k mod 512 gives X[m + 512 q]: t_r = W^(r m) S_r[m], then the 4-point DFT over r.
// Writes the one-sided part: X[m], X[m + 512], and X[1024] for m = 0.
__device__ __forceinline__ void combine2048_m(const float2 *S, int m, const float2 *tw2048, float2 *X)
{
    const float2 w1 = tw2048[m], w2 = tw2048[2 * m];                    // m < 512
    const float2 t0 = sub512_at(S, m), t1 = cmul(sub512_at(S + N512_ROW, m), w1), t2 = cmul(sub512_at(S + 2 * N512_ROW, m), w2),
                 t3 = cmul(sub512_at(S + 3 * N512_ROW, m), cmul(w1, w2));
    const float2 a = cadd(t0, t2), b = csub(t0, t2), c = cadd(t1, t3), d = csub(t1, t3);
    X[m] = cadd(a, c);
    X[m + 512] = make_float2(b.x + d.y, b.y - d.x);                     // b - j d
    if (m == 0) X[1024] = csub(a, c);
}

// inverse: the four one-sided sub-spectra at bin m = 0..256 of a spectrum Y[0..1024] (imaginary parts of Y[0], Y[1024] ignored):
//   Y_r[m] = W^(-r m) / 4 * sum_q exp(+j 2 pi r q / 4) Yext[m + 512 q],  Yext[i > 1024] = conj Y[2048 - i]
__device__ __forceinline__ void split2048_inv(const float2 *Y, int m, const float2 *tw2048, float2 (&out)[4])
{
    float2 y0 = Y[m], y1 = Y[m + 512], y2 = cconj(Y[1024 - m]), y3 = cconj(Y[512 - m]);
    if (m == 0) { y0.y = 0.f; y2.y = 0.f; }                             // DC and Nyquist (Yext[1024] = Y[1024]) are real
    const float2 t0 = cadd(y0, y2), t1 = csub(y0, y2), t2 = cadd(y1, y3), t3 = csub(y1, y3);
    const float2 jt3 = make_float2(-t3.y, t3.x);                        // j (y1 - y3)
    const float2 w1 = cconj(tw2048[m]), w2 = cconj(tw2048[2 * m]), w3 = cconj(tw2048[3 * m]);   // 3 m <= 768 < 1024
    const float2 a0 = cadd(t0, t2), a1 = cmul(cadd(t1, jt3), w1), a2 = cmul(csub(t0, t2), w2), a3 = cmul(csub(t1, jt3), w3);
    out[0] = make_float2(0.25f * a0.x, 0.25f * a0.y); out[1] = make_float2(0.25f * a1.x, 0.25f * a1.y);
    out[2] = make_float2(0.25f * a2.x, 0.25f * a2.y); out[3] = make_float2(0.25f * a3.x, 0.25f * a3.y);
}

// ---- 4096-sample real frames: eight 512-sample sub-sequences x[8 n + r] ----------------------------------------------
//   X[k] = sum_{r<8} W^(r k) S_r[k mod 512],  W = exp(-j 2 pi / 4096),  k = 0..2048;  tw4096[i] = W^i, i < 2048.
__device__ __forceinline__ float2 tw4096_at(const float2 *tw4096, int i)       // W^i for any i >= 0
{
    i &= 4095;
    if (i < 2048) return tw4096[i];
    const float2 t = tw4096[i - 2048];
    return make_float2(-t.x, -t.y);
}

// One radix-8 butterfly per m = k mod 512 gives X[m + 512 q], q = 0..7: t_r = W^(r m) S_r[m], then the 8-point DFT over r
// (W^(512 r q) = exp(-j 2 pi r q / 8)).  Writes the one-sided part, X[m + 512 q] for q = 0..3, and X[2048] for m = 0.
__device__ __forceinline__ void combine4096_m(const float2 *S, int m, const float2 *tw4096, float2 *X)
{
    float2 v[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) v[r] = sub512_at(S + r * N512_ROW, m);
    const float2 w1 = tw4096[m], w2 = tw4096[2 * m], w4 = tw4096_at(tw4096, 4 * m);    // m < 512: m, 2 m < 2048
    const float2 w3 = cmul(w1, w2), w5 = cmul(w4, w1), w6 = cmul(w4, w2), w7 = cmul(w4, w3);
    v[1] = cmul(v[1], w1); v[2] = cmul(v[2], w2); v[3] = cmul(v[3], w3); v[4] = cmul(v[4], w4);
    v[5] = cmul(v[5], w5); v[6] = cmul(v[6], w6); v[7] = cmul(v[7], w7);
    fft8<false>(v);                                                     // v[i] = X[m + 512 br3(i)]
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int q = br3(i);
        if (q < 4) X[m + 512 * q] = v[i];
        else if (q == 4 && m == 0) X[2048] = v[i];
    }
}

}  // namespace mca
