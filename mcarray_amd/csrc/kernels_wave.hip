// kernels_wave.hip -- wave-per-run stream kernels on the 1024-point complex transform of fft1024c.h (gfx950).
//
//   k_bf_table        steering table of the delay-and-sum stage for every angle of the DOA grid
//   k_beamform_wave   PCM -> FFT of channel PAIRS -> delay-and-sum (Beamformer.cpp:51-71) -> inverse FFT -> overlap-add,
//                     one wave per run of frames, no workgroup barrier inside the frame loop
//
// Delay-and-sum without ever separating the two channels of a pair.  With z_p = x_a + j x_b (a = 2p, b = 2p + 1) and
// Z_p = DFT(z_p) = X_a + j X_b, the beamformed frame y = IDFT(Y), Y[k] = (1/M) sum_c X_c[k] P_c[k] extended to a
// Hermitian spectrum (P_c[1024 - k] = conj P_c[k]; the imaginary parts of Y[0] and Y[512] are ignored by the reference's
// CCS inverse, i.e. P_c[512] counts with its real part), equals
//     y = Re IDFT(W),   W[k] = sum_p Z_p[k] T_p[k],   T_p[k] = (P_a[k] - j P_b[k]) / (M N),   k = 0..1023
// because Y = W' + conj-mirror(W') for W' = W N / 2.  T depends on the steering angle only: one 8 KB row per (angle, pair),
// built once per context (k_bf_table) and read through L2 -- the kernel computes no sincos and keeps no phasor state.
#include <type_traits>

#include "fft1024c.h"
#include "mca_internal.h"
#include "cand_unit.h"
#include "phat_pairs.h"
#include "pair_balance.h"

namespace mca {

// grid (D + 1, pairs) x 256.  Row 0: DOA = 0 rad, the module's initial _currentDOA
// (BeamformingSeparationAndLocalisation.cpp:51) that frames before the first pick are steered with; row 1 + d: grid[d].
__global__ __launch_bounds__(256) void k_bf_table(float2 *tab, const float *grid, const double *mic_x, int M, int n_pairs, double unit)
{
    const int d = blockIdx.x, pr = blockIdx.y;
    const double doa = d == 0 ? 0.0 : (double)grid[d - 1];
    const double cd = cos(doa + 1.57079632679489661923);                   // cos(DOA + M_PI/2), Beamformer.cpp:59
    const double sc = 1.0 / ((double)M * 1024.0);
    float2 *row = tab + ((long long)d * n_pairs + pr) * 1024;
    for (int k = threadIdx.x; k < 1024; k += 256) {
        const int kap = k <= 512 ? k : k - 1024;
        double pa[2] = {0.0, 0.0}, pb[2] = {0.0, 0.0};
        for (int e = 0; e < 2; ++e) {
            const int c = 2 * pr + e;
            if (c >= M) continue;
            double turns = (double)kap * (unit * mic_x[c] * cd);            // k s_c / (2 pi), s_c of Beamformer.cpp:59
            turns -= rint(turns);
            double sn, cs;
            sincospi(2.0 * turns, &sn, &cs);
            if (k == 512) sn = 0.0;                                          // Im Y[512] is dropped by the CCS inverse
            (e == 0 ? pa : pb)[0] = cs; (e == 0 ? pa : pb)[1] = sn;
        }
        row[k] = make_float2((float)((pa[0] + pb[1]) * sc), (float)((pa[1] - pb[0]) * sc));     // (P_a - j P_b) / (M N)
    }
}

// (x_a, x_b) * w for two consecutive points whose window samples share a register pair: op_sel broadcasts the low / high half
__device__ __forceinline__ float2 win_lo(float a, float b, v2f w)
{
    v2f x = {a, b}, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[1,0]" : "=v"(r) : "v"(x), "v"(w));
    return from_v2f(r);
}
__device__ __forceinline__ float2 win_hi(float a, float b, v2f w)
{
    v2f x = {a, b}, r;
    asm("v_pk_mul_f32 %0, %2, %1 op_sel:[1,0] op_sel_hi:[1,1]" : "=v"(r) : "v"(x), "v"(w));   // w in src0: its high half may feed the low result there (fft512.h, RULE)
    return from_v2f(r);
}

// grid (workgroups per array, arrays) x 256 threads = 4 waves.  Per frame: for each channel pair the windowed samples
// (z = (x_a, x_b) w), the 1024-point transform, W += Z T[doa bin]; then the inverse transform of W, whose real part is the
// beamformed frame.  A wave's work is ONE loop over its (frame, pair) steps with the same loads in every step -- the next
// step's samples (the last step reloads its own) and this step's table row, requested in the middle of the transform -- so
// that the counted waits on the row leave the sample loads in flight behind the transform.
//
// Overlap-add carries (HANDOFF): a workgroup covers 4 ft - 1 consecutive frames; wave 0 takes the frame BEFORE them too (only
// its second half counts: the carry into the workgroup's first hop) and ft - 1 frames, waves 1..3 take ft frames each.  A wave
// does not wait for its predecessor's carry: it keeps the first half of its first frame in registers, the waves leave their
// final carries in LDS, and after one barrier at the very end each wave adds its predecessor's carry and stores that hop.
// One frame in 4 ft is analysed twice (without HANDOFF every wave re-analyses the frame before its run: one in ft + 1).
// ODD: the last pair has one channel (its imaginary input is zero).  VAR: bit 0 stage-A twiddles in registers, bit 1 HANDOFF,
// bit 2 table row requested in the middle of the transform, bit 3 scheduling barriers (A/B switches; api.hip picks one).
#ifndef BFW_OCC
#define BFW_OCC 2
#endif
template <bool ODD, int VAR, int ABL>
__global__ __launch_bounds__(256, BFW_OCC) void k_beamform_wave(BeamformWaveArgs p)
{
    constexpr bool T1REG = VAR & 1, HANDOFF = VAR & 2, TMID = VAR & 4;
    // (round 6 A/B; MEASURE builds) STAGE: the next step's samples go to LDS with eight global_load_lds_dwordx4 (16 bytes
    // per lane whatever the transform's register layout wants) and are read from there with ds_read_b32, instead of 32 four-byte loads into
    // registers that stay live through the transform.  The staging buffer of a wave is 8 KiB; wave 0 takes the twiddle table's place (dead
    // once the stage-A twiddles are in registers), so that two workgroups still share a CU.
    constexpr bool STAGE = (VAR & 16) != 0;
    static_assert(!STAGE || (T1REG && TMID), "the staged variant reuses the twiddle table's LDS and requests its steering rows in the middle of the transform");
    constexpr int FOPT = 1 | ((VAR >> 3) & 1) << 1 | (T1REG ? 4 : 0);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float2 *tab = reinterpret_cast<float2 *>(smem_raw);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float2 *buf = tab + F1K_TWORDS + wave * F1K_SCRATCH;
    float *xcarry = reinterpret_cast<float *>(tab + F1K_TWORDS + 4 * F1K_SCRATCH);      // HANDOFF: [4 waves][512] final carries
    f1k_table_init(tab, tid, 256);
    F1kLane lc;
    lc.init(lane);
    __syncthreads();
    if (T1REG) lc.load_t1(tab, lane);
    typedef __attribute__((address_space(3))) void lds_void_t;
    float *stage = nullptr;
    if (STAGE) {
        __syncthreads();                                                   // (every wave has its twiddles: the table's place is free)
        stage = wave == 0 ? reinterpret_cast<float *>(tab) : reinterpret_cast<float *>(tab + F1K_TWORDS + 4 * F1K_SCRATCH) + 4 * FFT_H + (wave - 1) * 2048;
    }

    const int a = p.skew ? blockIdx.x : blockIdx.y;
    const long long as = (long long)a * p.S + blockIdx.z;                // (array, source): output channel, overlap-add carry
    int t0, t1;
    if (HANDOFF) {
        int w0 = (int)blockIdx.x * (4 * p.ft - 1), ft = p.ft;            // the workgroup's first frame, frames per wave
        if (p.skew) {
            const int g = blockIdx.y, half = gridDim.y >> 1, hi = p.ft + p.skew, lo = p.ft - p.skew;
            ft = g < half ? hi : lo;
            w0 = g < half ? g * (4 * hi - 1) : half * (4 * hi - 1) + (g - half) * (4 * lo - 1);
        }
        t0 = wave == 0 ? w0 : w0 + wave * ft - 1;
        t1 = min(w0 + (wave + 1) * ft - 1, p.n_frames);
    } else {
        t0 = ((int)blockIdx.x * 4 + wave) * p.ft;
        t1 = min(t0 + p.ft, p.n_frames);
    }
    const bool active = t0 < t1;
    // (Every array is cut at the same frames and the frames of a run go in order -- the overlap-add carry --, so waves that
    // start together stream the same piece of their rows.  Layouts whose row pitch is a power of two plus a little (128 arrays
    // x 257 half frames of 512 floats: 2^19 + 2^11 bytes) then put the loads of all resident waves onto the same few memory
    // channels: 0.33 instead of 0.295 ms per 32 768 frames; 64 floats of padding per row avoid it.  Shifting the run
    // boundaries per array costs an extra, mostly empty round of waves: 0.36 ms.  k_stft_phat_wave rotates its frame order.)
    const bool lead_in = HANDOFF ? (wave == 0 && t0 > 0) : t0 > 0;       // this wave analyses frame t0 - 1 for its carry
    const bool deferred = HANDOFF && wave > 0;                           // the carry into hop t0 comes from the wave before
    const int tfirst = lead_in ? t0 - 1 : t0;
    const int NP = p.n_pairs;
    float carry[8], first[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { carry[i] = 0.f; first[i] = 0.f; }

    if (active) {
        v2f win[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) { win[i].x = p.window[lane + 128 * i]; win[i].y = p.window[lane + 128 * i + 64]; }
        if (t0 == 0) {
#pragma unroll
            for (int i = 0; i < 8; ++i) carry[i] = p.tail_in[as * FFT_H + lane + 64 * i];
        }
        const float *base = p.pcm + (long long)a * p.array_stride + lane;
        const int *bins = p.doa_bin + (long long)a * p.n_frames * p.S + blockIdx.z;     // [frame][source]
        float xa[16], xb[16];
        auto load_pair = [&](int t, int pr) {
            const float *pa = base + (long long)(2 * pr) * p.mic_stride + (long long)t * FFT_H;
            const float *pb = (ODD && pr == NP - 1) ? pa : pa + p.mic_stride;
            if (STAGE) {
                // lane l, instruction q: samples 4 (64 q + l) .. + 3 -> the same words of the buffer (a linear copy of the frame)
                const float *ga = pa - lane + 4 * lane, *gb = pb - lane + 4 * lane;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    __builtin_amdgcn_global_load_lds(reinterpret_cast<const void *>(ga + 256 * q), (lds_void_t *)(stage + 256 * q), 16, 0, 0);
                    __builtin_amdgcn_global_load_lds(reinterpret_cast<const void *>(gb + 256 * q), (lds_void_t *)(stage + 1024 + 256 * q), 16, 0, 0);
                }
            } else {
#pragma unroll
            for (int i = 0; i < 16; ++i) { xa[i] = pa[64 * i]; xb[i] = pb[64 * i]; }
            }
        };
        auto take_pair = [&]() {                                           // STAGE: the staged step into registers
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int i = 0; i < 16; ++i) { xa[i] = stage[lane + 64 * i]; xb[i] = stage[1024 + lane + 64 * i]; }
            wave_lds_fence();
        };
        load_pair(tfirst, 0);

        float2 W[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) W[i] = make_float2(0.f, 0.f);
        int t = tfirst, pr = 0;
        const float2 *trow = p.table + ((long long)(bins[(long long)t * p.S] + 1) * NP) * 1024 + lane;
        for (;;) {
            float2 z[16], T[16];
            if (STAGE) take_pair();
#pragma unroll
            for (int i = 0; i < 8; ++i) { z[2 * i] = win_lo(xa[2 * i], xb[2 * i], win[i]); z[2 * i + 1] = win_hi(xa[2 * i + 1], xb[2 * i + 1], win[i]); }
            if (ODD && pr == NP - 1) {
#pragma unroll
                for (int i = 0; i < 16; ++i) z[i].y = 0.f;
            }
            const bool last_pair = pr == NP - 1, last = last_pair && t + 1 >= t1;
            if (!TMID) {
#pragma unroll
                for (int i = 0; i < 16; ++i) T[i] = trow[pr * 1024 + 64 * dr16(i)];
                if (FOPT & 2) __builtin_amdgcn_sched_barrier(0);
            }
            if (!(ABL & 1)) load_pair(last ? t : (last_pair ? t + 1 : t), last ? pr : (last_pair ? 0 : pr + 1));
            if (FOPT & 2) __builtin_amdgcn_sched_barrier(0);
            fft1024c<false, FOPT>(z, buf, lane, tab, lc, [&]() {
                if (TMID) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) T[i] = (ABL & 2) ? make_float2(1e-3f * (float)(i + pr), 1e-3f) : trow[pr * 1024 + 64 * dr16(i)];
                }
            });
#pragma unroll
            for (int i = 0; i < 16; ++i) W[i] = cmac(W[i], z[i], T[i]);
            if (last_pair) {
                // W[p] holds bin lane + 64 dr16(p); the inverse takes register i = bin lane + 64 i and returns sample lane + 64 dr16(p)
                float2 y[16];
#pragma unroll
                for (int i = 0; i < 16; ++i) y[i] = W[dr16(i)];
                fft1024c<true, FOPT>(y, buf, lane, tab, lc);
                if (t >= t0) {
                    if (deferred && t == t0) {
#pragma unroll
                        for (int i = 0; i < 8; ++i) first[i] = y[dr16(i)].x;
                    } else {
                        float *o = p.out + as * p.n_frames * FFT_H + (long long)t * FFT_H + lane;
#pragma unroll
                        for (int i = 0; i < 8; ++i) o[64 * i] = carry[i] + y[dr16(i)].x;
                    }
                }
#pragma unroll
                for (int i = 0; i < 8; ++i) carry[i] = y[dr16(i + 8)].x;
                if (last) break;
#pragma unroll
                for (int i = 0; i < 16; ++i) W[i] = make_float2(0.f, 0.f);
                ++t; pr = 0;
                trow = p.table + ((long long)(bins[(long long)t * p.S] + 1) * NP) * 1024 + lane;
            } else {
                ++pr;
            }
        }
        if (t1 == p.n_frames) {
#pragma unroll
            for (int i = 0; i < 8; ++i) p.tail_out[as * FFT_H + lane + 64 * i] = carry[i];
        }
    }
    if (HANDOFF) {
#pragma unroll
        for (int i = 0; i < 8; ++i) xcarry[wave * FFT_H + lane + 64 * i] = carry[i];
        __syncthreads();
        if (active && deferred) {
            float *o = p.out + as * p.n_frames * FFT_H + (long long)t0 * FFT_H + lane;
#pragma unroll
            for (int i = 0; i < 8; ++i) o[64 * i] = xcarry[(wave - 1) * FFT_H + lane + 64 * i] + first[i];
        }
    }
}

#define INST_BFW(V) template __global__ void k_beamform_wave<false, V, 0>(BeamformWaveArgs); template __global__ void k_beamform_wave<true, V, 0>(BeamformWaveArgs);
INST_BFW(15)              // the shipped variant
#ifdef MCA_MEASURE        // the A/B variants and the ablations (wrong results) of DESIGN.md's measurements: make MEASURE=1 only
template __global__ void k_beamform_wave<false, 14, 1>(BeamformWaveArgs); template __global__ void k_beamform_wave<false, 14, 2>(BeamformWaveArgs);
template __global__ void k_beamform_wave<false, 14, 3>(BeamformWaveArgs);
INST_BFW(0) INST_BFW(1) INST_BFW(2) INST_BFW(3) INST_BFW(4) INST_BFW(5) INST_BFW(6) INST_BFW(7)
INST_BFW(8) INST_BFW(9) INST_BFW(10) INST_BFW(11) INST_BFW(12) INST_BFW(13) INST_BFW(14)
INST_BFW(31)              // the shipped variant with its samples staged through LDS (round 6 A/B: profiles/r06_bfw_lds_stage_negative.log)
#endif

// --------------------------------------------------------------------------------------
// k_beamform_wave_ms: several sources per array, the forward transforms SHARED (round 4; up to 8 microphones)
// --------------------------------------------------------------------------------------
// processFrameSeparation beamforms min(M, S) outputs from the SAME analysis frames (BeamformingSeparationAndLocalisation.cpp:113-114,
// Beamformer.cpp:51-71): per frame the NPT pair spectra Z_p stay in registers (NPT <= 4: 128 of them) and every source s takes
// W_s = sum_p Z_p T_p[bin_s], one inverse transform and its own overlap-add carry (LDS).  Per frame 4 forward + S inverse transforms
// instead of S x (4 + 1): 2 975 instead of 5 286 vector instructions for three sources.  The steering rows of (source, pair + 1) are
// requested element by element behind the multiply-accumulates of (source, pair) -- one buffer of 16 registers, refilled as it is
// consumed --, those of the next source's first pair and the next frame's first samples in the middle of an inverse transform, where
// the pair spectra (the last source) or the steering rows are dead.  One wave per run of ft frames; a run re-analyses the frame
// before it for its overlap-add carry (its output is dropped).
template <int NPT, bool ODD>
__global__ __launch_bounds__(256, 2) void k_beamform_wave_ms(BeamformWaveArgs p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float2 *tab = reinterpret_cast<float2 *>(smem_raw);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float2 *buf = tab + F1K_TWORDS + wave * F1K_SCRATCH;
    const int S = p.S;
    float *carry = reinterpret_cast<float *>(tab + F1K_TWORDS + 4 * F1K_SCRATCH) + wave * (S * FFT_H) + lane;   // [S][512] of this wave
    f1k_table_init(tab, tid, 256);
    F1kLane lc;
    lc.init(lane);
    __syncthreads();

    const int a = blockIdx.y;
    const int t0 = ((int)blockIdx.x * 4 + wave) * p.ft, t1 = min(t0 + p.ft, p.n_frames);
    if (t0 >= t1) return;                                                 // (no barrier below)
    const int tfirst = t0 > 0 ? t0 - 1 : 0;
    v2f win[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { win[i].x = p.window[lane + 128 * i]; win[i].y = p.window[lane + 128 * i + 64]; }
    for (int s = 0; s < S; ++s) {
#pragma unroll
        for (int i = 0; i < 8; ++i) carry[s * FFT_H + 64 * i] = t0 == 0 ? p.tail_in[((long long)a * S + s) * FFT_H + lane + 64 * i] : 0.f;
    }
    const float *base = p.pcm + (long long)a * p.array_stride + lane;
    const int *bins = p.doa_bin + (long long)a * p.n_frames * S;            // [frame][source]
    float xa[16], xb[16];
    auto load_pair = [&](int t, int pr) {
        const float *pa = base + (long long)(2 * pr) * p.mic_stride + (long long)t * FFT_H;
        const float *pb = (ODD && pr == NPT - 1) ? pa : pa + p.mic_stride;
#pragma unroll
        for (int i = 0; i < 16; ++i) { xa[i] = pa[64 * i]; xb[i] = pb[64 * i]; }
    };
    auto row_of = [&](int t, int s) { return p.table + ((long long)(bins[(long long)t * S + s] + 1) * NPT) * 1024 + lane; };
    load_pair(tfirst, 0);
    for (int t = tfirst; t < t1; ++t) {
        float2 Z[NPT][16];
#pragma unroll
        for (int pr = 0; pr < NPT; ++pr) {
            float2 z[16];
#pragma unroll
            for (int i = 0; i < 8; ++i) { z[2 * i] = win_lo(xa[2 * i], xb[2 * i], win[i]); z[2 * i + 1] = win_hi(xa[2 * i + 1], xb[2 * i + 1], win[i]); }
            if (ODD && pr == NPT - 1) {
                // a zero the compiler cannot see through: folding it into the first butterflies makes the vectoriser pack adds with the
                // operand selects fft512.h's RULE forbids (tools/check_isa.py fails the build on those)
                float zero;
                asm("v_mov_b32 %0, 0" : "=v"(zero));
#pragma unroll
                for (int i = 0; i < 16; ++i) z[i].y = zero;
            }
            fft1024c<false, 3>(z, buf, lane, tab, lc, [&]() { if (pr + 1 < NPT) load_pair(t, pr + 1); });
#pragma unroll
            for (int i = 0; i < 16; ++i) Z[pr][i] = z[i];
        }
        float2 T[16];
        const float2 *trow = row_of(t, 0);
#pragma unroll
        for (int i = 0; i < 16; ++i) T[i] = trow[64 * dr16(i)];
        // one source: W = sum_p Z_p T_p, inverse transform, overlap-add.  LAST (peeled below: the pair spectra are dead behind its
        // products, which the compiler must be able to see): the next frame's first samples are requested in the middle of its inverse
        // transform; the other sources request the next source's first row right behind theirs.
        auto source = [&](int s, auto last_tag) {
            constexpr bool LAST = decltype(last_tag)::value;
            float2 W[16];
#pragma unroll
            for (int pr = 0; pr < NPT; ++pr) {
                __builtin_amdgcn_sched_barrier(0);      // (the refills stay behind this pair's products: hoisted, every pair's row would be live at once)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    W[i] = pr == 0 ? cmul(Z[0][i], T[i]) : cmac(W[i], Z[pr][i], T[i]);
                    if (pr + 1 < NPT) T[i] = trow[(pr + 1) * 1024 + 64 * dr16(i)];   // the next pair's row, element by element behind its use
                }
            }
            // W[q] holds bin lane + 64 dr16(q); the inverse takes register i = bin lane + 64 i and returns sample lane + 64 dr16(q)
            float2 y[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) y[i] = W[dr16(i)];
            if (LAST) {
                fft1024c<true, 3>(y, buf, lane, tab, lc, [&]() { load_pair(min(t + 1, t1 - 1), 0); });   // (the run's last frame reloads its own)
            } else {
                fft1024c<true, 3>(y, buf, lane, tab, lc);
                trow = row_of(t, s + 1);
#pragma unroll
                for (int i = 0; i < 16; ++i) T[i] = trow[64 * dr16(i)];
            }
            float *cs = carry + s * FFT_H;
            if (t >= t0) {
                float *o = p.out + ((long long)a * S + s) * p.n_frames * FFT_H + (long long)t * FFT_H + lane;
#pragma unroll
                for (int i = 0; i < 8; ++i) o[64 * i] = cs[64 * i] + y[dr16(i)].x;
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) cs[64 * i] = y[dr16(i + 8)].x;
        };
        for (int s = 0; s + 1 < S; ++s) source(s, std::false_type());
        source(S - 1, std::true_type());
    }
    if (t1 == p.n_frames) {
        for (int s = 0; s < S; ++s) {
#pragma unroll
            for (int i = 0; i < 8; ++i) p.tail_out[((long long)a * S + s) * FFT_H + lane + 64 * i] = carry[s * FFT_H + 64 * i];
        }
    }
}
template __global__ void k_beamform_wave_ms<1, false>(BeamformWaveArgs);      // (two microphones: one pair)
template __global__ void k_beamform_wave_ms<2, false>(BeamformWaveArgs); template __global__ void k_beamform_wave_ms<2, true>(BeamformWaveArgs);
template __global__ void k_beamform_wave_ms<3, false>(BeamformWaveArgs); template __global__ void k_beamform_wave_ms<3, true>(BeamformWaveArgs);
template __global__ void k_beamform_wave_ms<4, false>(BeamformWaveArgs); template __global__ void k_beamform_wave_ms<4, true>(BeamformWaveArgs);

// --------------------------------------------------------------------------------------
// k_stft_phat_wave: STFT analysis + GCC-PHAT pair products (SteeringBeamforming.cpp:104-130 up to the steering sum), one
// wave per run of frames, everything between the PCM loads and the A-row stores in registers.
// --------------------------------------------------------------------------------------
// Per frame the wave transforms the MT / 2 channel pairs (z = x_a + j x_b).  Lane l of the result holds the bins
// lam + 64 s, lam = l for l <= 32 and 96 - l above: the mirror bin 1024 - k of a lane's bin k then lives in lane l ^ 32,
// register 15 - s, and ONE v_permlane32_swap per dword brings the partner's upper half over (pairs of registers swap into
// each other's places; a v_swap puts them back).  Lanes 0 and 32 are their own mirrors (lam = 0: register (16 - s) & 15,
// lam = 32: register 15 - s) and sit the exchange out.  Then per bin k < 512
//     2 X_a = Z[k] + conj Z[1024 - k],   2 X_b = -j (Z[k] - conj Z[1024 - k])
// are whitened (the factor 2 drops out) and kept: 8 bins x MT channels per lane.  After the last pair the lane forms the
// pair products of its 8 bins (pair_stage of kernels_stream.hip) and stores them.  The Nyquist bins (lane 0, register 8:
// X_a = Re Z, X_b = Im Z) are parked in LDS and finished after the run, lane = frame.
// z = 2 X: |X|^2 > 1e-30 <=> |z|^2 > 4e-30.  alive: the channel has a non-zero sample in this frame (a channel of exact
// zeros must give X = 0 like the reference's own transform; riding on its partner's transform it would come out as the
// partner's rounding noise, which the whitening would blow up to unit modulus).
// NOPHAT (gcc_weighting NONE): X itself, z / 2.
// thr, unit (wave-uniform; pair_balance.h): a channel that went through the transform multiplied by s has thr = 4e-30 s^2 and
// unit = 0.5 / s; a channel without a non-zero windowed sample thr = inf and unit = 0.
template <bool NOPHAT = false>
__device__ __forceinline__ float2 whiten4(float2 z, float &pw, float thr = 4e-30f, float unit = 0.5f)
{
    v2f zv = to_v2f(z), sq, r;
    asm("v_pk_mul_f32 %0, %1, %1" : "=v"(sq) : "v"(zv));
    pw = sq.x + sq.y;
    // (v_rsq_f32 itself: a value that passes thr >= 4e-30 is a normal number, the result of the others is dropped -- rsqrtf() with a
    // run-time threshold makes the compiler guard every call against denormal inputs, four more instructions per bin and channel)
    const float s = NOPHAT ? unit : (pw > thr ? __builtin_amdgcn_rsqf(pw) : 0.f);
    v2f sv = {s, s};
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(r) : "v"(zv), "v"(sv));
    return from_v2f(r);
}

// A-row stores: a wave-uniform row pointer, a per-lane byte offset (the lane's residue) and a compile-time element index
template <bool PL2>
__device__ __forceinline__ void store_a_wave(_Float16 *row, unsigned voff, int cidx, float2 v, int Kp)
{
    const float2_t vv = {v.x, v.y};
    const half2_t hi = __builtin_convertvector(vv, half2_t);
    char *b = reinterpret_cast<char *>(row + 2 * cidx);
    *reinterpret_cast<half2_t *>(b + voff) = hi;
    if (PL2) {
        const float2_t back = __builtin_convertvector(hi, float2_t);
        *reinterpret_cast<half2_t *>(reinterpret_cast<char *>(row + Kp + 2 * cidx) + voff) = __builtin_convertvector(vv - back, half2_t);
    }
}
template <bool PL2>
__device__ __forceinline__ void store_a_wave(float *row, unsigned voff, int cidx, float2 v, int)
{
    *reinterpret_cast<float2 *>(reinterpret_cast<char *>(row + 2 * cidx) + 2 * voff) = v;
}

__device__ __forceinline__ unsigned or3(unsigned a, unsigned b, unsigned c)
{
    unsigned r;
    asm("v_or3_b32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

typedef __attribute__((address_space(3))) float lds_float_t;

// sum over the 64 lanes on v_add_f32 with a DPP source operand (quad swaps, row rotations, the two cross-row broadcasts of gfx9);
// taken from lane 63, uniform.  (A DPP read needs two wait states after the vector write of its source.)
__device__ __forceinline__ float wave_sum64(float v)
{
    asm("s_nop 1\n\tv_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_add_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_ror:4 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
        "s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
        "s_nop 1"
        : "+v"(v));
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

// MERGE (ULA, one fp16 plane): the contraction index is the product m = k (j - i) -- all (bin, spacing) combinations of equal
// product share one steering column (api.hip, build_merged_tables), so their PHAT sums are added up before they are stored:
// the lane's 8 x (M - 1) sums go into the wave's LDS region (the transform's scratch, free by then) with ds_add_f32 at
// rank[m] -- every instruction hits 64 different words and the instructions of a wave execute in order, so the sums are
// formed in a fixed order --, and the region is read back as the row: n_merged instead of (M - 1) * 513 complex values.
template <int MT, bool ULA, typename OutT, bool PL2, bool POWER, bool NOPHAT, bool MERGE, bool CAND = false>
__global__ __launch_bounds__(512) void k_stft_phat_wave(StftPhatArgs p)      // 4 or 8 waves (256 registers either way: two waves per SIMD)
{
    static_assert(!MERGE || (ULA && !PL2 && !NOPHAT && sizeof(OutT) == 2), "the merged index serves the one-plane fp16 rows of a ULA");
    constexpr int NP = MT / 2, NOUT = PairOut<MT, ULA>::N;
    constexpr int NRANK = 2 * (MT - 1) * 64 * 4 + 8;                              // u16 entries of the offset table (+ the Nyquist bin's)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float2 *tab = reinterpret_cast<float2 *>(smem_raw);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nw = (int)blockDim.x >> 6, nthr = (int)blockDim.x;                   // waves per workgroup
    // LDS: twiddles | MERGE: rank table | per wave: transform scratch (MERGE: also the merged sums, whichever is larger) | Nyquist bins
    const unsigned short *rankl = reinterpret_cast<const unsigned short *>(tab + F1K_TWORDS);
    const int nmp = MERGE ? (p.n_merged + 63) & ~63 : 0;                           // merged sums, rounded up to whole wave rows
    const int regw = MERGE ? max(F1K_SCRATCH, nmp) : F1K_SCRATCH;                  // float2 words per wave
    float2 *wbase = tab + F1K_TWORDS + (MERGE ? NRANK / 4 : 0);
    float2 *buf = wbase + wave * regw;
    float2 *nyq = wbase + nw * regw + wave * ((p.fpb + p.skew) * NP);              // [fpb (+ skew)][NP] Z_p[512] of the run's frames (not MERGE)
    // list mode: the length of the list and this workgroup's first entry are requested before the tables are built (the barrier
    // below would hold the loads back: three dependent round trips -- length, entry, samples -- in front of a single frame per wave)
    const int n_list_now = p.list ? *p.n_list : 0;
    const int e_first = (p.list && p.list0 + (int)blockIdx.x < n_list_now) ? p.list[p.list0 + (int)blockIdx.x] : 0;
    f1k_table_init(tab, tid, nthr);
    if (MERGE) {
        // per lane, spacing g and half h: the region offsets (in words) of its four products k g, k = lam + 64 (4 h + s):
        // [2 (g - 1) + h][lane][4] u16 -- one ds_read_b64 per batch; behind them rank[512 g], the Nyquist bin's
        unsigned short *ot = reinterpret_cast<unsigned short *>(tab + F1K_TWORDS);
        for (int e = tid; e < 2 * (MT - 1) * 64 * 4; e += nthr) {
            const int s4 = e & 3, ln = (e >> 2) & 63, gh = e >> 8, g = (gh >> 1) + 1, h = gh & 1;
            const int lm = ln <= 32 ? ln : 96 - ln;
            ot[e] = p.mrank[(lm + 64 * (4 * h + s4)) * g];
        }
        if (tid < MT - 1) ot[2 * (MT - 1) * 64 * 4 + tid] = p.mrank[512 * (tid + 1)];
    }
    F1kLane lc;
    lc.init(lane);
    __syncthreads();
    const int lam = lane <= 32 ? lane : 96 - lane;
    const unsigned voff = (unsigned)lam * 4u;                                      // byte offset of the lane's residue in an fp16 row
    const bool self = (lane & 31) == 0;                                            // lanes 0 and 32
    v2f win[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { win[i].x = p.window[lane + 128 * i]; win[i].y = p.window[lane + 128 * i + 64]; }

    // regular mode: wave w of workgroup b takes the run of p.fpb frames number 4 b + w.  List mode (repair pass of the adaptive
    // SRP precision): workgroup b walks the listed groups of REPAIR_GROUP = 4 frames, wave w takes frame w of a group.
    // Dynamic runs (p.queue; see StftPhatArgs): the wave takes its runs off the device-side counter; the request for the next run goes
    // out at the top of a run's last frame and is read when the run is done.
    const int li_end = p.list ? min(n_list_now, p.list0 + p.list_cap) : 1, li_step = p.list ? (int)gridDim.x : 1;
    unsigned rq = 0, rq_pending = 0;
    const unsigned long long clock_in = p.wave_clock ? wall_clock64() : 0ull;
    int runs_taken = 0;
    if (p.queue) {
        if (lane == 0) rq_pending = atomicAdd(p.queue, 1u);
        rq = __builtin_amdgcn_readfirstlane(rq_pending);
    }
    for (int li = p.list ? p.list0 + (int)blockIdx.x : 0; p.queue ? rq < (unsigned)p.q_total : li < li_end; li += li_step) {
        int a = blockIdx.y;
        // (measurement: xcd_map -- consecutive workgroups go to consecutive XCDs; give every XCD one contiguous piece of the array instead)
        const int bxm = (p.xcd_map && (gridDim.x & 7) == 0) ? (int)(blockIdx.x & 7) * (int)(gridDim.x >> 3) + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
        int f_begin = (bxm * nw + wave) * p.fpb, f_end = min(f_begin + p.fpb, p.n_frames);
        if (p.skew) {
            const int g = blockIdx.y, half = gridDim.y >> 1, hi = p.fpb + p.skew, lo = p.fpb - p.skew;
            a = blockIdx.x;
            if (g < half) { f_begin = (g * nw + wave) * hi; f_end = f_begin + hi; }
            else { f_begin = half * nw * hi + ((g - half) * nw + wave) * lo; f_end = f_begin + lo; }
        }
        if (p.queue) {
            int rr = 0, rpa = 1, f_first = 0, len = 1, f_last = 0;
            dyn_run((int)rq, p.n_frames, p.q_arrays, p.q_sh0, rr, rpa, f_first, len, f_last, p.q_flat);
            a = __builtin_amdgcn_readfirstlane(rr / rpa);                     // (the division runs on the vector ALU: back to scalar registers, or every address below turns into vector code)
            f_begin = f_first + (rr - a * rpa) * len;
            f_end = min(f_begin + len, f_last);
        }
        long long row_base = (long long)a * p.n_frames;     // A row of frame f = row_base + f
        bool hist_unit = false;                                // list mode, lazy tails: a unit of the PREVIOUS call's last frames (StftPhatArgs::hist_in)
        int e_unit = 0;
        if (p.list) {
            const int e = li == p.list0 + (int)blockIdx.x ? e_first : p.list[li];
            e_unit = e;
            hist_unit = p.hist_in != nullptr && e >= p.hist_base;
            const int eu = hist_unit ? e - p.hist_base : e, upa = hist_unit ? HIST_UNITS : p.groups_per_array;
            a = eu / upa;
            const int g_begin = (eu - a * upa) * REPAIR_GROUP;
            row_base = (long long)(li - p.list0) * REPAIR_GROUP - g_begin;
            f_begin = g_begin + wave; f_end = min(f_begin + 1, hist_unit ? HIST_FRAMES : p.n_frames);
        }
        const bool has_frame = f_begin < f_end;
        if (!has_frame) {
            if (p.queue) break;                     // (cannot happen: every listed run holds a frame)
            if (!CAND) continue;                    // (candidate-column list mode: the wave still helps to contract the unit's rows below)
        }
        if (has_frame) {
        const float *base = hist_unit ? p.hist_in + (long long)a * MT * HIST_SAMPLES + lane : p.pcm + (long long)a * p.array_stride + lane;
        const long long mstride = hist_unit ? (long long)HIST_SAMPLES : p.mic_stride;
        const int fr0 = hist_unit ? 0 : p.frame0;
        float xa[16], xb[16];
        auto load_pair = [&](int f, int pr) {
            const float *pa = base + (long long)(2 * pr) * mstride + (long long)(fr0 + f) * FFT_H;
            const float *pb = pa + mstride;
#pragma unroll
            for (int i = 0; i < 16; ++i) { xa[i] = pa[64 * i]; xb[i] = pb[64 * i]; }
        };
        // The frames of a run are independent: each run starts at its own offset and wraps around.  Waves that start
        // together then stream different 2 KB pieces of their rows -- with every run starting at its first frame, layouts
        // whose row pitch is a power of two plus a little (128 arrays x 257 half frames: 2^19 + 2^11 bytes) put the loads of
        // all resident waves onto the same few memory channels (measured: 0.33 instead of 0.27 ms per 32 768 frames).
        const int nfr = f_end - f_begin;
        const int rot = p.list ? 0 : (int)((unsigned)(5 * a + 3 * ((int)blockIdx.x * nw + wave)) % (unsigned)nfr);
        auto frame_of = [&](int i) { const int j = i + rot; return f_begin + (j >= nfr ? j - nfr : j); };
        load_pair(frame_of(0), 0);
        for (int fi = 0; fi < nfr; ++fi) {
            const int f = frame_of(fi);
            if (p.queue && fi == nfr - 1 && lane == 0) rq_pending = atomicAdd(p.queue, 1u);     // the next run: asked for a frame ahead of its use
            float2 Xh[MT][8], zn[NP];
            v2f ptime = {0.f, 0.f};                                                // POWER: sum of the squared windowed samples (Parseval)
            bool any_alive = false;                                                // (wave-uniform)
#pragma unroll
            for (int pr = 0; pr < NP; ++pr) {
                float2 z[16];
                if (p.hist_out) {
                    // lazy tails: the call's last HIST_FRAMES frames of PCM stay behind for the next call's repair pass (frame j of the
                    // history = samples [512 j, 512 j + 1024) of a channel's HIST_SAMPLES)
                    const int j = p.frame0 + f - (p.total_frames - HIST_FRAMES);
                    if (j >= 0) {
                        float *ha = p.hist_out + ((long long)a * MT + 2 * pr) * HIST_SAMPLES + j * FFT_H + lane;
#pragma unroll
                        for (int i = 0; i < 8; ++i) { ha[64 * i] = xa[i]; ha[HIST_SAMPLES + 64 * i] = xb[i]; }
                        if (j == HIST_FRAMES - 1) {
#pragma unroll
                            for (int i = 0; i < 8; ++i) { ha[FFT_H + 64 * i] = xa[8 + i]; ha[HIST_SAMPLES + FFT_H + 64 * i] = xb[8 + i]; }
                        }
                    }
                }
#pragma unroll
                for (int i = 0; i < 8; ++i) { z[2 * i] = win_lo(xa[2 * i], xb[2 * i], win[i]); z[2 * i + 1] = win_hi(xa[2 * i + 1], xb[2 * i + 1], win[i]); }
                if (POWER) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) { const v2f zv = to_v2f(z[i]); asm("v_pk_fma_f32 %0, %1, %1, %0" : "+v"(ptime) : "v"(zv)); }
                }
                // the lane's largest |windowed sample| of either channel: a channel of exact zeros?  two channels whose levels are far
                // apart?  (pair_balance.h: the weaker one goes through the transform scaled up by a power of two)
                float ma = max3abs(z[0].x, z[1].x, z[2].x), mb = max3abs(z[0].y, z[1].y, z[2].y);
#pragma unroll
                for (int i = 3; i < 15; i += 2) { ma = max3abs(ma, z[i].x, z[i + 1].x); mb = max3abs(mb, z[i].y, z[i + 1].y); }
                ma = max2abs(ma, z[15].x); mb = max2abs(mb, z[15].y);
                const PairBalance pb = pair_balance(ma, mb, !p.no_balance);
                const bool alive_a = pb.alive_a, alive_b = pb.alive_b;
                any_alive = any_alive || alive_a || alive_b;
                if (pb.scaled()) {
                    const float sa = pb.sa(), sb = pb.sb();
#pragma unroll
                    for (int i = 0; i < 16; ++i) z[i] = make_float2(z[i].x * sa, z[i].y * sb);
                }
                // the next pair's samples (the run's last step reloads its own) are requested in the middle of the transform
                fft1024c<false, 3>(z, buf, lane, tab, lc, [&]() {
                    const bool lastp = pr == NP - 1, last = lastp && fi + 1 >= nfr;
                    load_pair(last ? f : (lastp ? frame_of(fi + 1) : f), last ? pr : (lastp ? 0 : pr + 1));
                }, lam);
                // z[q] = Z[lam + 64 dr16(q)];  the register of bin index s is dr16(s)
                // The mirror Z[1024 - k] of the lane's bin k = lam + 64 s (s < 8) is the partner lane's (lane ^ 32) register 15 - s.
                // Two v_permlane32_swap per register pair (j, j + 4), j = 8..11, leave the partner's register j in slot j + 4 and
                // the partner's j + 4 in slot j: the mirror of bin s is read from slot mate(15 - s).  Lanes 0 and 32 are their own
                // mirrors -- lane 32: register 15 - s, lane 0: register (16 - s) & 15 -- and put those into the same slots.
                const float un_a = pb.un_a(), un_b = pb.un_b();                                  // back to the channel's own scale (0: exact zeros)
                if (MERGE) zn[pr] = make_float2(z[dr16(8)].x * un_a, z[dr16(8)].y * un_b);     // (lane 0's is the Nyquist bin)
                if (lane == 0) {
                    const float2 n = z[dr16(8)];
                    if (!MERGE) nyq[(f - f_begin) * NP + pr] = make_float2(n.x * un_a, n.y * un_b);
                    float2 t[16];
#pragma unroll
                    for (int j = 0; j < 16; ++j) t[j] = z[dr16(j)];
#pragma unroll
                    for (int j = 8; j < 16; ++j) z[dr16(j < 12 ? j + 4 : j - 4)] = t[(j + 1) & 15];       // slot mate(j) <- register j + 1
                } else if (self) {
#pragma unroll
                    for (int j = 8; j < 12; ++j) { const float2 t = z[dr16(j)]; z[dr16(j)] = z[dr16(j + 4)]; z[dr16(j + 4)] = t; }
                } else {
#pragma unroll
                    for (int j = 8; j < 12; ++j) {
                        float2 &u = z[dr16(j)], &w = z[dr16(j + 4)];
                        swap_rows32(u.x, w.x); swap_rows32(w.x, u.x);
                        swap_rows32(u.y, w.y); swap_rows32(w.y, u.y);
                    }
                }
                // z = 2 s X (s: the channel's power of two): |X|^2 > 1e-30 <=> |z|^2 > 4e-30 s^2;  NOPHAT: X = z / (2 s)
                const float thr_a = PairBalance::thr(4e-30f, pb.na, alive_a), thr_b = PairBalance::thr(4e-30f, pb.nb, alive_b);
                const float half_a = PairBalance::down(0.5f, pb.na, alive_a), half_b = PairBalance::down(0.5f, pb.nb, alive_b);
#pragma unroll
                for (int s = 0; s < 8; ++s) {
                    const float2 zk = z[dr16(s)], zm = z[dr16(15 - s < 12 ? 15 - s + 4 : 15 - s - 4)];      // Z[k], Z[1024 - k] (slot mate(15 - s))
                    const float2 a2 = make_float2(zk.x + zm.x, zk.y - zm.y);                               // 2 X_a
                    const float2 b2 = make_float2(zk.y + zm.y, zm.x - zk.x);                               // 2 X_b
                    float pwa, pwb;
                    Xh[2 * pr][s] = whiten4<NOPHAT>(a2, pwa, thr_a, half_a);
                    Xh[2 * pr + 1][s] = whiten4<NOPHAT>(b2, pwb, thr_b, half_b);
                }
            }
            if (p.dead && lane == 0) p.dead[(long long)a * p.total_frames + p.frame0 + f] = any_alive ? 0 : 1;
            OutT *arow = reinterpret_cast<OutT *>(p.A) + (row_base + f) * (long long)p.a_row_elems;
            if (MERGE) {
                // One spacing g at a time: the products k g of a lane's eight bins -- of all lanes' bins -- are all different for a
                // fixed g, so the eight read-add-write sequences of a batch never touch a word twice; batches follow each other in
                // program order (the LDS executes a wave's instructions in order).  (ds_add_f32 would do the same without the
                // batches, but runs at one lane every ~2.5 cycles: 1.5 ms per 32 768 frames instead of 0.27.)
                float2 *S = buf;
                for (int j = lane; j < nmp / 2; j += 64) reinterpret_cast<float4 *>(S)[j] = make_float4(0.f, 0.f, 0.f, 0.f);
                wave_lds_fence();
                float xn[MT];                                                     // lane 0: the Nyquist bin (real: X_a = Re Z, X_b = Im Z), whitened
#pragma unroll
                for (int pr = 0; pr < NP; ++pr) {
                    float pw;
                    xn[2 * pr] = whiten4(make_float2(2.f * zn[pr].x, 0.f), pw).x;
                    xn[2 * pr + 1] = whiten4(make_float2(2.f * zn[pr].y, 0.f), pw).x;
                }
                const uint2 *offt = reinterpret_cast<const uint2 *>(rankl) + lane;
                // (no fence between the batches: the LDS executes a wave's instructions in order and the compiler keeps the
                // order of accesses to S that may alias; the offsets of the next batch can be read ahead)
#pragma unroll
                for (int g = 1; g < MT; ++g) {
#pragma unroll
                    for (int h = 0; h < 2; ++h) {                                 // four bins at a time (registers)
                        const uint2 o4 = offt[(2 * (g - 1) + h) * 64];
                        const unsigned off[4] = {o4.x & 0xffffu, o4.x >> 16, o4.y & 0xffffu, o4.y >> 16};
                        float2 acc[4], cur[4];
#pragma unroll
                        for (int s2 = 0; s2 < 4; ++s2) {
                            acc[s2] = make_float2(0.f, 0.f);
#pragma unroll
                            for (int i = 0; i + g < MT; ++i) acc[s2] = cmacc(acc[s2], Xh[i][4 * h + s2], Xh[i + g][4 * h + s2]);
                        }
#pragma unroll
                        for (int s2 = 0; s2 < 4; ++s2) cur[s2] = S[off[s2]];
#pragma unroll
                        for (int s2 = 0; s2 < 4; ++s2) S[off[s2]] = cadd(cur[s2], acc[s2]);
                    }
                }
                if (lane == 0) {                                                  // the Nyquist bin: m = 512 g, one word per spacing
                    float an[MT - 1], cn[MT - 1];
                    unsigned on[MT - 1];
#pragma unroll
                    for (int g = 1; g < MT; ++g) {
                        an[g - 1] = 0.f;
#pragma unroll
                        for (int i = 0; i + g < MT; ++i) an[g - 1] = fmaf(xn[i], xn[i + g], an[g - 1]);
                        on[g - 1] = rankl[2 * (MT - 1) * 64 * 4 + g - 1];
                    }
#pragma unroll
                    for (int g = 0; g < MT - 1; ++g) cn[g] = S[on[g]].x;
#pragma unroll
                    for (int g = 0; g < MT - 1; ++g) S[on[g]].x = cn[g] + an[g];
                }
                wave_lds_fence();
                // the row: two merged sums (16 bytes of the region, 8 bytes of fp16) per lane and step; the entries behind
                // n_merged are zero and fall into the row's padding
                typedef _Float16 half4_t __attribute__((ext_vector_type(4)));
                typedef float float4_t __attribute__((ext_vector_type(4)));
                for (int j = lane; j < nmp / 2; j += 64) {
                    const float4 v = reinterpret_cast<const float4 *>(S)[j];
                    const float4_t vv = {v.x, v.y, v.z, v.w};
                    reinterpret_cast<half4_t *>(arow)[j] = __builtin_convertvector(vv, half4_t);
                }
                wave_lds_fence();                                                  // (the next frame's transform reuses the region)
            } else {
#pragma unroll
                for (int s = 0; s < 8; ++s) {
                    if constexpr (ULA) {
                        float2 r[MT], out[NOUT];
#pragma unroll
                        for (int m = 0; m < MT; ++m) r[m] = Xh[m][s];
                        pair_products<MT, ULA>(r, out);
#pragma unroll
                        for (int g = 0; g < NOUT; ++g) store_a_wave<PL2>(arow, voff, g * KG + 64 * s, out[g], p.Kp);
                    } else {
                        // one product per pair: formed and stored one at a time (28 of them held together next to the 128
                        // registers of spectra spilled 8 ... 42 registers)
                        int pi = 0;
#pragma unroll
                        for (int i = 0; i < MT; ++i)
#pragma unroll
                            for (int j = i + 1; j < MT; ++j) { store_a_wave<PL2>(arow, voff, pi * KG + 64 * s, cmulc(Xh[i][s], Xh[j][s]), p.Kp); ++pi; }
                    }
                }
            }
            if (POWER) {
                // dsp::SignalPower::FFTPower [INFERRED, SURVEY A.8]: (1/N^2) sum_k w_k |X[k]|^2 over the channels, / M, w = 2 except
                // DC and Nyquist: the sum over the full spectrum, = N sum_n (w[n] x[n])^2 (Parseval) -- taken from the windowed
                // samples, where it costs 16 instructions per pair and no register across the transforms
                const float pacc = wave_sum64(ptime.x + ptime.y);                    // (six DPP adds: the shuffles were six round trips through the LDS crossbar)
                if (lane == 0) p.power[(long long)a * p.total_frames + p.frame0 + f] = pacc / (float)FFT_N / (float)MT;
            }
        }
        // Nyquist bins of the run: lane = frame
        wave_lds_fence();
        if (!MERGE && lane < f_end - f_begin) {
            float2 xn[MT], out[NOUT];
#pragma unroll
            for (int pr = 0; pr < NP; ++pr) {
                const float2 n = nyq[lane * NP + pr];
                float pw;
                xn[2 * pr] = whiten4<NOPHAT>(make_float2(2.f * n.x, 0.f), pw);
                xn[2 * pr + 1] = whiten4<NOPHAT>(make_float2(2.f * n.y, 0.f), pw);
            }
            OutT *arow = reinterpret_cast<OutT *>(p.A) + (row_base + f_begin + lane) * (long long)p.a_row_elems;
            pair_products<MT, ULA>(xn, out);
#pragma unroll
            for (int g = 0; g < NOUT; ++g) store_a_wave<PL2>(arow, 0u, g * KG + FFT_H, out[g], p.Kp);
        }
        }
        wave_lds_fence();
        if constexpr (CAND) {
            static_assert(!CAND || PL2, "the candidate contraction reads the two-plane rows of the list mode");
            {
                // candidate columns: the unit's four rows are written -- by this workgroup, through this CU's L1 -- and are contracted at the
                // unit's columns right here (cand_unit.h), in the transforms' scratch
                __syncthreads();
                cand_unit<4>(p.cand, li - p.list0, e_unit, reinterpret_cast<unsigned char *>(wbase), tid);
            }
        }
        if (p.queue) rq = __builtin_amdgcn_readfirstlane(rq_pending);
        ++runs_taken;
    }
    if (p.wave_clock && lane == 0) {
        unsigned long long *wc = p.wave_clock + 3ull * ((blockIdx.y * gridDim.x + blockIdx.x) * (unsigned)nw + wave);
        wc[0] = clock_in; wc[1] = wall_clock64(); wc[2] = (unsigned long long)runs_taken;
    }
    // the last wave to leave zeroes the counters (every wave has made its last request by then)
    if (p.queue && lane == 0) {
        __threadfence();
        if (atomicAdd(p.queue + 1, 1u) == gridDim.x * gridDim.y * (unsigned)nw - 1u) { p.queue[0] = 0u; p.queue[1] = 0u; }
    }
}

// --------------------------------------------------------------------------------------
// k_stft_phat_wave16: the wave-per-run analysis for a 16-microphone uniform linear array, ONE fp16 operand plane (round 4)
// --------------------------------------------------------------------------------------
// Eight pair transforms per frame; 16 channels x 8 bins of whitened spectra per lane would be 256 registers in fp32, so they are
// kept as packed fp16 pairs (128 registers) -- this kernel only serves the rows that are rounded to fp16 anyway (the ADAPTIVE
// coarse pass, plain FP16: every operand of the contraction carries that rounding, api.hip's error model) -- and the 120 pair
// products of a bin are formed on v_dot2_f32_f16 with fp32 sums: Re a conj b = dot2(a, b), Im a conj b = dot2(a, (-b.y, b.x)), the
// rotated copy one v_pk_mul_f16 per channel and bin.  The sums of the 15 spacings of a bin are converted and stored before the
// next bin's are formed (30 accumulators).  The exact rows of such an array (FP16X3, the repair pass) stay on k_stft_phat<16>.
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ h2 rot_j_conj(h2 b)                            // (-b.y, b.x)
{
    h2 r;
    const unsigned c = 0x3C00BC00u;                                        // (lo, hi) = (-1, +1)
    asm("v_pk_mul_f16 %0, %1, %2 op_sel:[1,0] op_sel_hi:[0,1]" : "=v"(r) : "v"(b), "v"(c));
    return r;
}

template <bool POWER>
__global__ __launch_bounds__(256, 2) void k_stft_phat_wave16(StftPhatArgs p)
{
    constexpr int MT = 16, NP = 8;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float2 *tab = reinterpret_cast<float2 *>(smem_raw);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float2 *buf = tab + F1K_TWORDS + wave * F1K_SCRATCH;
    float *wtab = reinterpret_cast<float *>(tab + F1K_TWORDS + 4 * F1K_SCRATCH);   // [64 lanes][16]: the lane's window samples w[lane + 64 i]
    const int fpa = p.fpb + p.skew;                                                // frames a wave's LDS slots hold (StftPhatArgs::skew)
    float2 *nyq = reinterpret_cast<float2 *>(wtab + 1024) + wave * (fpa * NP);    // [fpb][NP] Z_p[512] of the run's frames
    // (unsure frames, see StftPhatArgs: per frame and pair the channels' MEAN bin power as the scale -- Parseval: sum_n (w x)^2, wave-reduced;
    // round 4 took the power of bin 64 alone, which a notch or a tone at 3 kHz makes arbitrarily small or large, ADVICE r4 --, per frame
    // whether a DC bin fell below it)
    float2 *nref = reinterpret_cast<float2 *>(wtab + 1024) + 4 * (fpa * NP) + wave * (fpa * NP);
    float *dcbad = reinterpret_cast<float *>(reinterpret_cast<float2 *>(wtab + 1024) + 8 * (fpa * NP)) + wave * fpa;
    constexpr float UNSURE = 1e-10f;                                               // power ratio: 1e-5 of the channel's rms bin amplitude in this frame
    for (int e = tid; e < 1024; e += 256) wtab[(e & 63) * 16 + (e >> 6)] = p.window[e];
    f1k_table_init(tab, tid, 256);
    F1kLane lc;
    lc.init(lane);
    __syncthreads();
    const int lam = lane <= 32 ? lane : 96 - lane;
    const unsigned voff = (unsigned)lam * 4u;
    const bool self = (lane & 31) == 0;
    // (the window is read from LDS per pair: 16 registers that the 128 of packed spectra do not leave room for)
    const float4 *wq = reinterpret_cast<const float4 *>(wtab + lane * 16);

    const int a = p.skew ? blockIdx.x : blockIdx.y;
    int f_begin = ((int)blockIdx.x * 4 + wave) * p.fpb, f_end = min(f_begin + p.fpb, p.n_frames);
    if (p.skew) {                                                                 // (see StftPhatArgs::skew)
        const int g = blockIdx.y, half = gridDim.y >> 1, hi = p.fpb + p.skew, lo = p.fpb - p.skew;
        if (g < half) { f_begin = (g * 4 + wave) * hi; f_end = f_begin + hi; }
        else { f_begin = half * 4 * hi + ((g - half) * 4 + wave) * lo; f_end = f_begin + lo; }
    }
    if (f_begin >= f_end) return;                                                 // (no barrier below)
    const long long row_base = (long long)a * p.n_frames;
    const float *base = p.pcm + (long long)a * p.array_stride + lane;
    float xa[16], xb[16];
    auto load_pair = [&](int f, int pr) {
        const float *pa = base + (long long)(2 * pr) * p.mic_stride + (long long)(p.frame0 + f) * FFT_H;
        const float *pb = pa + p.mic_stride;
#pragma unroll
        for (int i = 0; i < 16; ++i) { xa[i] = pa[64 * i]; xb[i] = pb[64 * i]; }
    };
    const int nfr = f_end - f_begin;
    const int rot = (int)((unsigned)(5 * a + 3 * ((int)blockIdx.x * 4 + wave)) % (unsigned)nfr);      // (row pitch: see k_stft_phat_wave)
    auto frame_of = [&](int i) { const int j = i + rot; return f_begin + (j >= nfr ? j - nfr : j); };
    load_pair(frame_of(0), 0);
    for (int fi = 0; fi < nfr; ++fi) {
        const int f = frame_of(fi);
        h2 Xh[MT][8];
        v2f ptime = {0.f, 0.f};
        bool dc_unsure = false;                                                    // (lane 0's: it holds bin 0)
#pragma unroll
        for (int pr = 0; pr < NP; ++pr) {
            float2 z[16];
#pragma unroll
            for (int i4 = 0; i4 < 4; ++i4) {
                const float4 w4 = wq[i4];
                const v2f w01 = {w4.x, w4.y}, w23 = {w4.z, w4.w};
                z[4 * i4] = win_lo(xa[4 * i4], xb[4 * i4], w01); z[4 * i4 + 1] = win_hi(xa[4 * i4 + 1], xb[4 * i4 + 1], w01);
                z[4 * i4 + 2] = win_lo(xa[4 * i4 + 2], xb[4 * i4 + 2], w23); z[4 * i4 + 3] = win_hi(xa[4 * i4 + 3], xb[4 * i4 + 3], w23);
            }
            float ref_a = 0.f, ref_b = 0.f;                                         // |2 X|^2 of an average bin of the two channels (p.unsure only)
            if (POWER || p.unsure) {
                v2f pz = {0.f, 0.f};
#pragma unroll
                for (int i = 0; i < 16; ++i) { const v2f zv = to_v2f(z[i]); asm("v_pk_fma_f32 %0, %1, %1, %0" : "+v"(pz) : "v"(zv)); }
                if (POWER) { ptime.x += pz.x; ptime.y += pz.y; }
                if (p.unsure) { ref_a = 4.f * wave_sum64(pz.x); ref_b = 4.f * wave_sum64(pz.y); }
            }
            // exact zeros?  levels far apart?  (as k_stft_phat_wave; pair_balance.h)
            float ma = max3abs(z[0].x, z[1].x, z[2].x), mb = max3abs(z[0].y, z[1].y, z[2].y);
#pragma unroll
            for (int i = 3; i < 15; i += 2) { ma = max3abs(ma, z[i].x, z[i + 1].x); mb = max3abs(mb, z[i].y, z[i + 1].y); }
            ma = max2abs(ma, z[15].x); mb = max2abs(mb, z[15].y);
            const PairBalance pb = pair_balance(ma, mb, !p.no_balance);
            const bool alive_a = pb.alive_a, alive_b = pb.alive_b;
            float dc_a = ref_a, dc_b = ref_b;
            if (pb.scaled()) {
                const float sa = pb.sa(), sb = pb.sb();
#pragma unroll
                for (int i = 0; i < 16; ++i) z[i] = make_float2(z[i].x * sa, z[i].y * sb);
                dc_a *= sa * sa; dc_b *= sb * sb;                                     // (the DC bin below is the scaled channel's; the Nyquist bin is parked unscaled)
            }
            fft1024c<false, 3>(z, buf, lane, tab, lc, [&]() {
                const bool lastp = pr == NP - 1, last = lastp && fi + 1 >= nfr;
                load_pair(last ? f : (lastp ? frame_of(fi + 1) : f), last ? pr : (lastp ? 0 : pr + 1));
            }, lam);
            // the mirror exchange and the separation of the two channels: as k_stft_phat_wave
            if (lane == 0) {
                const float2 n = z[dr16(8)];
                nyq[(f - f_begin) * NP + pr] = make_float2(n.x * pb.un_a(), n.y * pb.un_b());
                float2 t[16];
#pragma unroll
                for (int j = 0; j < 16; ++j) t[j] = z[dr16(j)];
#pragma unroll
                for (int j = 8; j < 16; ++j) z[dr16(j < 12 ? j + 4 : j - 4)] = t[(j + 1) & 15];
            } else if (self) {
#pragma unroll
                for (int j = 8; j < 12; ++j) { const float2 t = z[dr16(j)]; z[dr16(j)] = z[dr16(j + 4)]; z[dr16(j + 4)] = t; }
            } else {
#pragma unroll
                for (int j = 8; j < 12; ++j) {
                    float2 &u = z[dr16(j)], &w = z[dr16(j + 4)];
                    swap_rows32(u.x, w.x); swap_rows32(w.x, u.x);
                    swap_rows32(u.y, w.y); swap_rows32(w.y, u.y);
                }
            }
            float pw0a = 0.f, pw0b = 0.f;
            const float thr_a = PairBalance::thr(4e-30f, pb.na, alive_a), thr_b = PairBalance::thr(4e-30f, pb.nb, alive_b);
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                const float2 zk = z[dr16(s)], zm = z[dr16(15 - s < 12 ? 15 - s + 4 : 15 - s - 4)];
                const float2 a2 = make_float2(zk.x + zm.x, zk.y - zm.y);                               // 2 X_a
                const float2 b2 = make_float2(zk.y + zm.y, zm.x - zk.x);                               // 2 X_b
                float pwa, pwb;
                const float2 wa = whiten4<false>(a2, pwa, thr_a), wb = whiten4<false>(b2, pwb, thr_b);
                const float2_t va = {wa.x, wa.y}, vb = {wb.x, wb.y};
                Xh[2 * pr][s] = __builtin_convertvector(va, h2);
                Xh[2 * pr + 1][s] = __builtin_convertvector(vb, h2);
                if (s == 0) { pw0a = pwa; pw0b = pwb; }
                if (s == 0 && p.unsure && lane == 0) {                             // lane 0 holds bin 0: against the channel's mean bin power
                    nref[(f - f_begin) * NP + pr] = make_float2(ref_a, ref_b);
                    dc_unsure = dc_unsure || (pw0a < UNSURE * dc_a) || (pw0b < UNSURE * dc_b);
                }
            }
        }
        if (p.unsure && lane == 0) dcbad[f - f_begin] = dc_unsure ? 1.f : 0.f;
        _Float16 *arow = reinterpret_cast<_Float16 *>(p.A) + (row_base + f) * (long long)p.a_row_elems;
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            __builtin_amdgcn_sched_barrier(0);           // (one bin's 30 sums at a time: interleaved, the bins' accumulators spill)
            h2 rj[MT];
#pragma unroll
            for (int j = 1; j < MT; ++j) rj[j] = rot_j_conj(Xh[j][s]);
            float re[MT - 1], im[MT - 1];
#pragma unroll
            for (int g = 0; g < MT - 1; ++g) { re[g] = 0.f; im[g] = 0.f; }
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = i + 1; j < MT; ++j) {
                    re[j - i - 1] = __builtin_amdgcn_fdot2(Xh[i][s], Xh[j][s], re[j - i - 1], false);
                    im[j - i - 1] = __builtin_amdgcn_fdot2(Xh[i][s], rj[j], im[j - i - 1], false);
                }
#pragma unroll
            for (int g = 0; g < MT - 1; ++g) store_a_wave<false>(arow, voff, g * KG + 64 * s, make_float2(re[g], im[g]), p.Kp);
        }
        if (POWER) {
            const float pacc = wave_sum64(ptime.x + ptime.y);
            if (lane == 0) p.power[(long long)a * p.total_frames + p.frame0 + f] = pacc / (float)FFT_N / (float)MT;
        }
    }
    // Nyquist bins of the run: lane = frame (fp32: 120 products once per run)
    wave_lds_fence();
    if (lane < f_end - f_begin) {
        float2 xn[MT], out[MT - 1];
#pragma unroll
        for (int pr = 0; pr < NP; ++pr) {
            const float2 n = nyq[lane * NP + pr];
            float pw;
            xn[2 * pr] = whiten4<false>(make_float2(2.f * n.x, 0.f), pw);
            xn[2 * pr + 1] = whiten4<false>(make_float2(2.f * n.y, 0.f), pw);
        }
        if (p.unsure) {
            bool u = dcbad[lane] != 0.f;
#pragma unroll
            for (int pr = 0; pr < NP; ++pr) {
                const float2 n = nyq[lane * NP + pr], r = nref[lane * NP + pr];       // (n is X, the scale |2 X|^2)
                u = u || (4.f * n.x * n.x < UNSURE * r.x) || (4.f * n.y * n.y < UNSURE * r.y);
            }
            p.unsure[(long long)a * p.total_frames + p.frame0 + f_begin + lane] = u ? 1 : 0;
        }
        _Float16 *arow = reinterpret_cast<_Float16 *>(p.A) + (row_base + f_begin + lane) * (long long)p.a_row_elems;
        pair_products<MT, true>(xn, out);
#pragma unroll
        for (int g = 0; g < MT - 1; ++g) store_a_wave<false>(arow, 0u, g * KG + FFT_H, out[g], p.Kp);
    }
}
template __global__ void k_stft_phat_wave16<false>(StftPhatArgs);
template __global__ void k_stft_phat_wave16<true>(StftPhatArgs);

#define INST_SPW1(MT, ULA, T, PL2, NP) template __global__ void k_stft_phat_wave<MT, ULA, T, PL2, false, NP, false>(StftPhatArgs); \
                                       template __global__ void k_stft_phat_wave<MT, ULA, T, PL2, true, NP, false>(StftPhatArgs);
#define INST_SPW(MT, ULA) INST_SPW1(MT, ULA, _Float16, false, false) INST_SPW1(MT, ULA, _Float16, true, false) INST_SPW1(MT, ULA, float, false, false) \
                          INST_SPW1(MT, ULA, float, false, true)
INST_SPW(8, true) INST_SPW(8, false) INST_SPW(4, true) INST_SPW(4, false)
template __global__ void k_stft_phat_wave<8, true, _Float16, false, false, false, true>(StftPhatArgs);
template __global__ void k_stft_phat_wave<8, true, _Float16, false, true, false, true>(StftPhatArgs);
template __global__ void k_stft_phat_wave<4, true, _Float16, false, false, false, true>(StftPhatArgs);
template __global__ void k_stft_phat_wave<4, true, _Float16, false, true, false, true>(StftPhatArgs);
// list mode of a candidate-column call: two planes, no gate, with the candidate contraction of the unit behind its rows (CAND)
template __global__ void k_stft_phat_wave<8, true, _Float16, true, false, false, false, true>(StftPhatArgs);
template __global__ void k_stft_phat_wave<8, false, _Float16, true, false, false, false, true>(StftPhatArgs);
template __global__ void k_stft_phat_wave<4, true, _Float16, true, false, false, false, true>(StftPhatArgs);
template __global__ void k_stft_phat_wave<4, false, _Float16, true, false, false, false, true>(StftPhatArgs);

}  // namespace mca
