// pair_balance.h -- level balance of two channels that share one complex transform (round 6).
//
// The wave-level analysis kernels transform two channels as ONE complex sequence, z = x_a + j x_b, and separate the spectra by
// symmetry (X_a = (Z[k] + conj Z[N - k]) / 2, X_b = -j (Z[k] - conj Z[N - k]) / 2).  The fp32 rounding of Z -- about 2^-23 of the
// LARGER channel -- lands on both, and PHAT (SteeringBeamforming.cpp:115-119: every channel's own spectrum, whitened) keeps only the
// phase: a channel r times weaker than its partner would carry a phase error of ~1e-7 r where the reference, which transforms every
// channel on its own (dsp::STFT(M, order), SourceSeparationAndLocalisation.cpp:52), has none (round 5's fuzz, seed 7305: a digitally
// muted channel whose mute edge falls inside a frame, r ~ 1e6).  Whitening does not care about a channel's scale, so the weaker
// channel's windowed samples are multiplied by a power of two (exact) that brings it to its partner's level BEFORE the transform,
// and the thresholds behind the transform are scaled with it: its rounding is then its own, as in a transform of its own.
//
// Cost on balanced input (2.1 % of k_stft_phat_wave, profiles/r06_ab_r05_r06.log): the per-lane largest |windowed sample| of either channel
// (v_max3_f32, in place of the OR chain of the exact-zeros test, which it also answers) and a screen of two ballots.  The screen is NECESSARY for an imbalance: the lane that
// holds the wave maximum of the stronger channel sees it above PB_SCREEN x its own value of the other channel.  Only then are the
// two wave maxima formed (DPP) and compared by exponent; channels less than PB_MIN_SHIFT exponents apart stay untouched (the same
// bits as before this header existed), so do frames of microphones that see the same field.
#pragma once
#include "fft512.h"

namespace mca {

constexpr int PB_MIN_SHIFT = 4;          // exponents apart from which the weaker channel is scaled (levels < 16 x apart: untouched)
constexpr float PB_SCREEN = 8.f;         // = 2^(PB_MIN_SHIFT - 1): exponents PB_MIN_SHIFT apart mean maxima more than this factor apart
constexpr int PB_MAX_SHIFT = 60;         // |X|^2 > 1e-30 cuts a channel off 1e-15 below unit scale anyway; thr * 2^120 stays finite

__device__ __forceinline__ float max3abs(float a, float b, float c)
{
    float r;
    asm("v_max3_f32 %0, |%1|, |%2|, |%3|" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ float max2abs(float a, float b)
{
    float r;
    asm("v_max_f32 %0, |%1|, |%2|" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// maxima over the 64 lanes of two non-negative values, taken from lane 63 (uniform).  DPP steps as wave_sum64 of kernels_wave.hip;
// the two chains alternate, so one s_nop covers the two wait states a DPP read needs behind the vector write of its source.
__device__ __forceinline__ void wave_max64_2(float &a, float &b)
{
#define MCA_PB_STEP(ctrl) "v_max_f32_dpp %0, %0, %0 " ctrl "\n\tv_max_f32_dpp %1, %1, %1 " ctrl "\n\ts_nop 0\n\t"
    asm("s_nop 1\n\t"
        MCA_PB_STEP("quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
        MCA_PB_STEP("quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf")
        MCA_PB_STEP("row_ror:4 row_mask:0xf bank_mask:0xf")
        MCA_PB_STEP("row_ror:8 row_mask:0xf bank_mask:0xf")
        MCA_PB_STEP("row_bcast:15 row_mask:0xa bank_mask:0xf")
        MCA_PB_STEP("row_bcast:31 row_mask:0xc bank_mask:0xf")
        "s_nop 0"
        : "+v"(a), "+v"(b));
#undef MCA_PB_STEP
    a = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(a), 63));
    b = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(b), 63));
}

// Every member is wave-uniform and every derived constant is built with INTEGER arithmetic on the exponent field (scalar ALU: gfx950
// has no scalar float multiply; as floats these would occupy vector registers across the transform).
struct PairBalance {
    int na, nb;                          // the windowed samples of channel a / b go through the transform multiplied by 2^na / 2^nb (0: untouched)
    bool alive_a, alive_b;               // the channel has a non-zero windowed sample in this frame (a channel of exact zeros must
                                         // give X = 0 like the reference's own transform, not its partner's rounding noise)
    __device__ __forceinline__ bool scaled() const { return (na | nb) != 0; }
    static __device__ __forceinline__ float pow2(int n) { return __uint_as_float((unsigned)(127 + n) << 23); }
    __device__ __forceinline__ float sa() const { return pow2(na); }
    __device__ __forceinline__ float sb() const { return pow2(nb); }
    // back to the channel's own scale; 0 for a channel of exact zeros
    __device__ __forceinline__ float un_a() const { return alive_a ? pow2(-na) : 0.f; }
    __device__ __forceinline__ float un_b() const { return alive_b ? pow2(-nb) : 0.f; }
    // c 2^(-n) for a live channel, 0 for a dead one (c: a normal number well inside the exponent range)
    static __device__ __forceinline__ float down(float c, int n, bool alive) { return alive ? __uint_as_float(__float_as_uint(c) - ((unsigned)n << 23)) : 0.f; }
    // thr 2^(2 n): the threshold on |s X|^2 that stands for thr on |X|^2; +inf for a dead channel (nothing passes)
    static __device__ __forceinline__ float thr(float t, int n, bool alive) { return alive ? __uint_as_float(__float_as_uint(t) + ((unsigned)(2 * n) << 23)) : __builtin_inff(); }
};

// ma, mb: per lane, the largest |windowed sample| of channel a / b among the lane's samples of the frame
__device__ __forceinline__ PairBalance pair_balance(float ma, float mb, bool enable = true)
{
    PairBalance r;
    r.na = r.nb = 0;
    r.alive_a = __any(ma > 0.f);
    r.alive_b = __any(mb > 0.f);
    if (enable && (__any(ma > PB_SCREEN * mb) || __any(mb > PB_SCREEN * ma))) {
        wave_max64_2(ma, mb);
        const int ea = (int)(__float_as_uint(ma) >> 23), eb = (int)(__float_as_uint(mb) >> 23);      // (non-negative values: no sign bit)
        if (__float_as_uint(ma) != 0u && __float_as_uint(mb) != 0u) {          // (both alive; integer compares: scalar ALU)
            const int d = ea - eb, n = min(d < 0 ? -d : d, PB_MAX_SHIFT);
            if (n >= PB_MIN_SHIFT) { r.na = d < 0 ? n : 0; r.nb = d > 0 ? n : 0; }
        }
    }
    return r;
}

// ---- 512-sample frames: two channels per 512-point complex transform (fft512.h: load_pair_512 / rfft512_pair) -----------------
// v[r] = (a[m], b[m]) w[m] / 2: balanced in place before the transform ...
__device__ __forceinline__ PairBalance pair_balance_512(float2 (&v)[8])
{
    float ma = max3abs(v[0].x, v[1].x, v[2].x), mb = max3abs(v[0].y, v[1].y, v[2].y);
    ma = max3abs(ma, v[3].x, v[4].x); mb = max3abs(mb, v[3].y, v[4].y);
    ma = max3abs(ma, v[5].x, v[6].x); mb = max3abs(mb, v[5].y, v[6].y);
    ma = max2abs(ma, v[7].x); mb = max2abs(mb, v[7].y);
    const PairBalance pb = pair_balance(ma, mb);
    if (pb.scaled()) {
        const float sa = pb.sa(), sb = pb.sb();
#pragma unroll
        for (int r = 0; r < 8; ++r) v[r] = make_float2(v[r].x * sa, v[r].y * sb);
    }
    return pb;
}
// ... and the two spectra rfft512_pair left in LDS (k = 0..256; a lane touches the words it wrote) back at the channels' own scales;
// a channel of exact zeros gets zeros, not its partner's rounding noise
__device__ __forceinline__ void pair_restore_512(const PairBalance &pb, float2 *specA, float2 *specB, int lane)
{
    if (pb.scaled() || !pb.alive_a || !pb.alive_b) {
        const float ua = pb.un_a(), ub = pb.un_b();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int k = lane + 64 * i;
            const float2 xa = specA[k], xb = specB[k];
            specA[k] = make_float2(xa.x * ua, xa.y * ua);
            specB[k] = make_float2(xb.x * ub, xb.y * ub);
        }
        if (lane == 0) {
            const float2 xa = specA[256], xb = specB[256];
            specA[256] = make_float2(xa.x * ua, xa.y * ua);
            specB[256] = make_float2(xb.x * ub, xb.y * ub);
        }
        wave_lds_fence();
    }
}

}  // namespace mca
