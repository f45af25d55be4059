// kernels_stream.hip -- batched stream kernels of the localisation + separation path (gfx950).
//
//   k_stft_phat      PCM -> windowed 1024-pt FFT per channel (LDS) -> PHAT whitening ->
//                    per-pair cross-spectra summed per delay group -> A operand of the SRP GEMM
//   k_scan_pick      0.8 IIR over frames + selectDOA (SteeringBeamforming.cpp:132-195)
//   k_beamform_ola   PCM -> FFT -> delay-and-sum (Beamformer.cpp:51-71) -> inverse FFT -> overlap-add
//
// Data layout in HBM: PCM fp32 channel-major [array][mic][sample] (coalesced float2 loads along
// time); A operand [frame][Kp] with Kp = roundup(G*1026, 32), flat index (g*513 + k)*2 + {re,im};
// correlation map C [split-K plane][array][frame][Dp] fp32.
#include "fft512.h"
#include "mca_internal.h"
#include "phat_pairs.h"
#include "pair_balance.h"

namespace mca {

// --------------------------------------------------------------------------------------
// k_stft_phat
// --------------------------------------------------------------------------------------
// One 512-thread workgroup (8 waves) walks FPB consecutive frames of one array.  Wave w transforms
// channels w, w+8; the raw second half of a frame stays in registers and becomes the first half
// of the next frame (each PCM sample is loaded once per workgroup) and the next frame's new half is
// loaded one iteration ahead.  After the FFTs the M spectra sit in LDS; thread k whitens bin
// k < 512 of every channel and forms the pair products.  The 513th (Nyquist) bin would make one
// wave run the pair stage twice per frame, so its whitened values are parked in LDS and the
// Nyquist bins of all FPB frames are finished in one extra pass at the end (lane = frame).
//
// MT > 0: compile-time channel count (pair products from registers); MT == 0: runtime M, pair
// operands re-read from LDS.  ULA: pairs with equal (j - i) share one delay table
// (host-verified, bitwise-equal float delays), so their PHAT spectra are summed: G = M - 1
// groups instead of P = M(M-1)/2 -- the contraction depth of the SRP GEMM drops by M/2.
// the per-wave partial sums of a frame's power, in wave order
__device__ __forceinline__ float sum8(const float *s) { return ((((((s[0] + s[1]) + s[2]) + s[3]) + s[4]) + s[5]) + s[6]) + s[7]; }

template <int MT, bool ULA, typename OutT>
// (__launch_bounds__(512, 6) -- three workgroups per CU -- measured 2 % faster, but its spilled registers add 0.1 GB of
// scratch traffic per 32 768 frames: not taken)
__global__ __launch_bounds__(512) void k_stft_phat(StftPhatArgs p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float2 *smem = reinterpret_cast<float2 *>(smem_raw);                // [M][FFT_SCRATCH] spectra
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int M = MT > 0 ? MT : p.M;
    float2 *tab = smem + M * FFT_SCRATCH;                                 // [TW_WORDS] twiddles + window
    float2 *nyq = tab + TW_WORDS;                                         // [fpb][M] whitened Nyquist bins
    float *spow = reinterpret_cast<float *>(nyq + p.fpb * M);             // [fpb][8] per wave: sum_c sum_k w_k |X_c[k]|^2 (power gate)
    constexpr int CPW = (MT > 0 && MT <= 8) ? 1 : 2;   // channels per FFT wave
    fft_table_init(tab, p.window, tid, 512);
    FftTw tw{tab};
    // list mode (repair pass of the adaptive SRP precision): the workgroups walk the listed groups of REPAIR_GROUP frames
    // (a device-side count: the grid is a fixed, moderate size); group number li - list0 of the pass writes A rows
    // (li - list0) * REPAIR_GROUP ...  Otherwise: one run of fpb frames per workgroup.
    const int li_end = p.list ? min(*p.n_list, p.list0 + p.list_cap) : 1, li_step = p.list ? (int)gridDim.x : 1;
    for (int li = p.list ? p.list0 + (int)blockIdx.x : 0; li < li_end; li += li_step) {
    int a = blockIdx.y;
    int f_begin = blockIdx.x * p.fpb;
    long long row_base = (long long)a * p.n_frames;     // A row of frame f = row_base + f
    if (p.list) {
        const int e = p.list[li];
        a = e / p.groups_per_array;
        f_begin = (e - a * p.groups_per_array) * REPAIR_GROUP;
        row_base = (long long)(li - p.list0) * REPAIR_GROUP - f_begin;
    }
    const int f_end = min(f_begin + p.fpb, p.n_frames);
    __syncthreads();

    // raw[cc][0..3]: first half of the current frame; nxt[cc][0..3]: its second half, loaded one
    // iteration ahead so the HBM latency hides behind the FFTs and the pair stage of the previous frame.
    float2 raw[CPW][4], nxt[CPW][4];
    const float *base = p.pcm + (long long)a * p.array_stride;
#pragma unroll
    for (int cc = 0; cc < CPW; ++cc) {
        const int c = wave + 8 * cc;
        if (c < M) {
            const float2 *src = reinterpret_cast<const float2 *>(base + (long long)c * p.mic_stride + (long long)(p.frame0 + f_begin) * FFT_H);
#pragma unroll
            for (int r = 0; r < 4; ++r) raw[cc][r] = src[lane + 64 * r];
#pragma unroll
            for (int r = 0; r < 4; ++r) nxt[cc][r] = src[lane + 64 * (r + 4)];
        }
    }

    for (int f = f_begin; f < f_end; ++f) {
#pragma unroll
        for (int cc = 0; cc < CPW; ++cc) {
            const int c = wave + 8 * cc;
            if (c < M) {
                float2 v[8];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float2 w0 = tw.win(r, lane), w1 = tw.win(r + 4, lane);
                    v[r] = make_float2(raw[cc][r].x * w0.x, raw[cc][r].y * w0.y);
                    v[r + 4] = make_float2(nxt[cc][r].x * w1.x, nxt[cc][r].y * w1.y);
                    raw[cc][r] = nxt[cc][r];
                }
                if (f + 1 < f_end) {
                    const float2 *src = reinterpret_cast<const float2 *>(base + (long long)c * p.mic_stride + (long long)(p.frame0 + f + 1) * FFT_H);
#pragma unroll
                    for (int r = 0; r < 4; ++r) nxt[cc][r] = src[lane + 64 * (r + 4)];
                }
                rfft1024<true>(v, smem + c * FFT_SCRATCH, lane, tw);
            }
        }
        __syncthreads();

        OutT *arow = reinterpret_cast<OutT *>(p.A) + (row_base + f) * (long long)p.a_row_elems;
        if (tid < M) nyq[(f - f_begin) * M + tid] = whiten(smem[tid * FFT_SCRATCH + FFT_H]);
        if (p.power) {
            // dsp::SignalPower::FFTPower [INFERRED, SURVEY A.8]: (1/N^2) sum_k w_k |X[k]|^2, w = 2 except DC and Nyquist
            float acc = 0.f;
            for (int m = 0; m < M; ++m) { const float2 z = smem[m * FFT_SCRATCH + tid]; acc += z.x * z.x + z.y * z.y; }
            acc *= tid == 0 ? 1.f : 2.f;
            if (tid < M) { const float2 z = smem[tid * FFT_SCRATCH + FFT_H]; acc += z.x * z.x + z.y * z.y; }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off);
            if (lane == 0) spow[(f - f_begin) * 8 + wave] = acc;        // (one slot per wave, summed in a fixed order below: the
        }                                                               //  gate's decisions do not depend on the order of atomics)
        {
            const int k = tid;   // bins 0..511
            if constexpr (MT == 0)   // runtime M: whiten the thread's own column in place, pairs re-read it from LDS
                for (int m = 0; m < M; ++m) smem[m * FFT_SCRATCH + k] = whiten(smem[m * FFT_SCRATCH + k]);
            pair_stage<MT, ULA, true, OutT>(smem + k, FFT_SCRATCH, M, arow, p, k);
        }
        __syncthreads();
    }
    if (p.power && tid < f_end - f_begin)
        p.power[(long long)a * p.total_frames + p.frame0 + f_begin + tid] = sum8(spow + tid * 8) / ((float)FFT_N * (float)FFT_N) / (float)M;
    // Nyquist bins of the block's frames: lane = frame
    if (tid < f_end - f_begin) {
        OutT *arow = reinterpret_cast<OutT *>(p.A) + (row_base + f_begin + tid) * (long long)p.a_row_elems;
        pair_stage<MT, ULA, false, OutT>(nyq + tid * M, 1, M, arow, p, FFT_H);
    }
    __syncthreads();                                     // (list mode: the next group reuses nyq / spow)
    }
}

#define INST_STFT(MT, ULA, T) template __global__ void k_stft_phat<MT, ULA, T>(StftPhatArgs);
INST_STFT(0, false, float) INST_STFT(0, true, float)
INST_STFT(4, false, float) INST_STFT(4, true, float)
INST_STFT(8, false, float) INST_STFT(8, true, float)
INST_STFT(16, true, float)
INST_STFT(0, false, _Float16) INST_STFT(0, true, _Float16)
INST_STFT(4, false, _Float16) INST_STFT(4, true, _Float16)
INST_STFT(8, false, _Float16) INST_STFT(8, true, _Float16)
INST_STFT(16, true, _Float16)

// --------------------------------------------------------------------------------------
// k_stft_phat_few: the same stage for few microphones (instantiated for 2; with 4 it measured no faster than
// k_stft_phat).  With one wave per channel only 2 of the 8 waves of k_stft_phat would transform; here a pass takes FP = 8 / MT consecutive frames at once (wave w -> frame slot
// w / MT, channel w % MT), then the 512 threads run the pair stage of each of those frames.  Frames handled by one
// wave are FP apart, so both halves of a frame are loaded (no shared half frame); the next pass is prefetched.
// --------------------------------------------------------------------------------------
template <int MT, bool ULA, typename OutT>
__global__ __launch_bounds__(512) void k_stft_phat_few(StftPhatArgs p)
{
    constexpr int FP = 8 / MT;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float2 *smem = reinterpret_cast<float2 *>(smem_raw);                // [8][FFT_SCRATCH]: wave w's spectrum = (slot, channel)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float2 *tab = smem + 8 * FFT_SCRATCH;                                 // [TW_WORDS]
    float2 *nyq = tab + TW_WORDS;                                         // [fpb][MT] whitened Nyquist bins
    float *spow = reinterpret_cast<float *>(nyq + p.fpb * MT);            // [fpb][8] per wave
    const int a = blockIdx.y;
    const int f_begin = blockIdx.x * p.fpb;
    const int f_end = min(f_begin + p.fpb, p.n_frames);
    const int slot = wave / MT, c = wave % MT;

    fft_table_init(tab, p.window, tid, 512);
    __syncthreads();
    FftTw tw{tab};

    const float *base = p.pcm + (long long)a * p.array_stride + (long long)c * p.mic_stride;
    float2 nxt[8];
    if (f_begin + slot < f_end) {
        const float2 *src = reinterpret_cast<const float2 *>(base + (long long)(p.frame0 + f_begin + slot) * FFT_H);
#pragma unroll
        for (int r = 0; r < 8; ++r) nxt[r] = src[lane + 64 * r];
    }
    for (int f = f_begin; f < f_end; f += FP) {
        if (f + slot < f_end) {
            float2 v[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) { const float2 w = tw.win(r, lane); v[r] = make_float2(nxt[r].x * w.x, nxt[r].y * w.y); }
            if (f + FP + slot < f_end) {
                const float2 *src = reinterpret_cast<const float2 *>(base + (long long)(p.frame0 + f + FP + slot) * FFT_H);
#pragma unroll
                for (int r = 0; r < 8; ++r) nxt[r] = src[lane + 64 * r];
            }
            rfft1024(v, smem + wave * FFT_SCRATCH, lane, tw);
        }
        __syncthreads();
#pragma unroll
        for (int sl = 0; sl < FP; ++sl) {
            const int fr = f + sl;
            if (fr < f_end) {
                const float2 *x = smem + sl * MT * FFT_SCRATCH;
                OutT *arow = reinterpret_cast<OutT *>(p.A) + ((long long)a * p.n_frames + fr) * (long long)p.a_row_elems;
                if (tid < MT) nyq[(fr - f_begin) * MT + tid] = whiten(x[tid * FFT_SCRATCH + FFT_H]);
                if (p.power) {   // FFTPower [INFERRED, SURVEY A.8], as in k_stft_phat
                    float acc = 0.f;
#pragma unroll
                    for (int m = 0; m < MT; ++m) { const float2 z = x[m * FFT_SCRATCH + tid]; acc += z.x * z.x + z.y * z.y; }
                    acc *= tid == 0 ? 1.f : 2.f;
                    if (tid < MT) { const float2 z = x[tid * FFT_SCRATCH + FFT_H]; acc += z.x * z.x + z.y * z.y; }
#pragma unroll
                    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off);
                    if (lane == 0) spow[(fr - f_begin) * 8 + wave] = acc;
                }
                pair_stage<MT, ULA, true, OutT>(x + tid, FFT_SCRATCH, MT, arow, p, tid);
            }
        }
        __syncthreads();
    }
    if (p.power && tid < f_end - f_begin)
        p.power[(long long)a * p.total_frames + p.frame0 + f_begin + tid] = sum8(spow + tid * 8) / ((float)FFT_N * (float)FFT_N) / (float)MT;
    if (tid < f_end - f_begin) {
        OutT *arow = reinterpret_cast<OutT *>(p.A) + ((long long)a * p.n_frames + f_begin + tid) * (long long)p.a_row_elems;
        pair_stage<MT, ULA, false, OutT>(nyq + tid * MT, 1, MT, arow, p, FFT_H);
    }
}

#define INST_FEW(MT, ULA, T) template __global__ void k_stft_phat_few<MT, ULA, T>(StftPhatArgs);
INST_FEW(2, false, float)
INST_FEW(2, false, _Float16)

// --------------------------------------------------------------------------------------
// k_gate -- the power gate of BeamformingSeparationAndLocalisation::processFrameLocalisation
// (BeamformingSeparationAndLocalisation.cpp:55-87) for every frame of the batch
// --------------------------------------------------------------------------------------
// grid (arrays).  The floor estimation accumulates FFTPower * N over the first 3 s of a stream
// (:57-66, sequential by nature, at most ~141 frames: thread 0); every later frame is independent:
// voiced = 10 log10(FFTPower) > floor (:83,:87).  During the estimation setPowerFloor returns the
// running _powerFloor itself, so `power > _powerFloor` is false and those frames never fire.
__global__ __launch_bounds__(256) void k_gate(GateArgs p)
{
    __shared__ int s_first_free;     // first frame after the estimation phase
    __shared__ double s_floor;
    const int a = blockIdx.x, tid = threadIdx.x;
    const float *pl = p.power_lin + (long long)a * p.n_frames;
    double *st = p.state + (long long)a * 4;
    if (tid == 0) {
        double acc = st[0], consumed = st[1], floor_db = st[2];
        bool est = st[3] != 0.0;
        const bool est_at_entry = est;
        int t = 0;
        for (; t < p.n_frames && !est; ++t) {
            acc += (double)pl[t] * (double)p.fft_n + p.eps;              // _powerFloor += FFTPower * (fftCCSLength - 2) :58-59
            consumed += (double)p.fft_n;                                 // :60
            double shown = acc;
            if (consumed >= (double)p.needed_samples) {                  // :62-67
                est = true;
                floor_db = 10.0 * log10(acc / consumed) + (double)p.margin_db;
                acc = floor_db;                                          // _powerFloor now holds the floor in dB
                shown = floor_db;
            }
            p.voiced[(long long)a * p.n_frames + t] = 0;                 // power == _powerFloor, never greater :87
            if (p.power_out) p.power_out[(long long)a * p.n_frames + t] = (float)shown;
        }
        st[0] = acc; st[1] = consumed; st[2] = floor_db; st[3] = est ? 1.0 : 0.0;
        s_first_free = t; s_floor = floor_db;
        if (p.post0) p.post0[a] = est_at_entry ? 0 : (est ? t - 1 : p.n_frames);
    }
    __syncthreads();
    const double floor_db = s_floor;
    for (int t = s_first_free + tid; t < p.n_frames; t += 256) {
        const float pw = 10.f * log10f(pl[t]);                           // FFTLogPower :83
        p.voiced[(long long)a * p.n_frames + t] = (double)pw > floor_db ? 1 : 0;
        if (p.power_out) p.power_out[(long long)a * p.n_frames + t] = pw;
    }
}

// --------------------------------------------------------------------------------------
// k_scan_partial / k_scan_carry / k_scan_pick -- computeEnergyInDOA + selectDOA over all frames
// --------------------------------------------------------------------------------------
// E_t = 0.8f E_{t-1} + (1 - 0.8f) C_t on frames that pass the gate, unchanged otherwise
// (SteeringBeamforming.cpp:132-144 is only reached for voiced frames, BeamformingSeparation...cpp:87-89).
// The recursion over frames is evaluated as an exact chunked scan: (1) k_scan_partial runs it from
// zero inside every chunk (result b, count n of voiced frames), (2) k_scan_carry composes the
// chunks in order, E_start[c+1] = 0.8f^n E_start[c] + b, (3) k_scan_pick re-runs every chunk from its
// true start value and peak-picks.  All chunks of (1) and (3) run in parallel.
__device__ __forceinline__ float median3f(float a, float b, float c)
{
    float lo = fminf(a, b), hi = fmaxf(a, b);
    return fmaxf(lo, fminf(hi, c));
}

// sum of the split-K partial maps at element o (plane 0 first: the same order for every kernel that reads C)
__device__ __forceinline__ float csum(const float *C, long long o, int planes, long long stride)
{
    float v = C[o];
    for (int pl = 1; pl < planes; ++pl) v += C[o + pl * stride];
    return v;
}
// The same for N rows at once.  Through csum a row's sum is a loop of unknown length that waits for every load before the
// next one goes out -- N rows "in flight" were N round trips in a row.  Here all loads of a plane go out together; one and
// two planes (every large batch: the contraction's two K halves) are straight-line code.
template <int N, typename Row>
__device__ __forceinline__ void csum_rows(float (&v)[N], const float *C, Row row_offset, int planes, long long stride)
{
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = C[row_offset(i)];
    if (planes >= 2) {
        float w[N];
#pragma unroll
        for (int i = 0; i < N; ++i) w[i] = C[row_offset(i) + stride];
#pragma unroll
        for (int i = 0; i < N; ++i) v[i] += w[i];
    }
    for (int pl = 2; pl < planes; ++pl) {
        float w[N];
#pragma unroll
        for (int i = 0; i < N; ++i) w[i] = C[row_offset(i) + pl * stride];
#pragma unroll
        for (int i = 0; i < N; ++i) v[i] += w[i];
    }
}

// Folds the partial maps of a deep split-K contraction (small batches: up to 16 maps) into map 0, plane 0
// first, four elements per thread; the scan kernels then read one map.  n4 = elements / 4.
__global__ __launch_bounds__(256) void k_sum_planes(float *C, long long n4, int planes, long long stride)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    float4 *c4 = reinterpret_cast<float4 *>(C);
    const long long s4 = stride >> 2;
    float4 acc = c4[i];
    int pl = 1;
    for (; pl + 3 < planes; pl += 4) {
        const float4 a = c4[i + pl * s4], b = c4[i + (pl + 1) * s4], c = c4[i + (pl + 2) * s4], d = c4[i + (pl + 3) * s4];
        acc.x = (((acc.x + a.x) + b.x) + c.x) + d.x; acc.y = (((acc.y + a.y) + b.y) + c.y) + d.y;
        acc.z = (((acc.z + a.z) + b.z) + c.z) + d.z; acc.w = (((acc.w + a.w) + b.w) + c.w) + d.w;
    }
    for (; pl < planes; ++pl) { const float4 a = c4[i + pl * s4]; acc.x += a.x; acc.y += a.y; acc.z += a.z; acc.w += a.w; }
    c4[i] = acc;
}

// one step of E = 0.8f E + 0.2f C (:134-140), the same two instructions in every kernel that runs the recursion
__device__ __forceinline__ float iir_step(float mu, float E, float omu, float c) { return fmaf(mu, E, omu * c); }

// grid (chunks, arrays), Dp / 4 threads (rounded up to waves): a thread owns four neighbouring delays and reads whole
// 16-byte pieces of the map rows, SCAN_CHUNK / 2 rows in flight.
__global__ __launch_bounds__(256) void k_scan_partial(ScanPickArgs p)
{
    const int q = threadIdx.x, a = blockIdx.y, c = blockIdx.x, d0 = 4 * q;
    const int t_start = c * p.chunk, t_end = min(t_start + p.chunk, p.n_frames);
    const float *C = p.C + (long long)a * p.n_frames * p.Dp;
    const unsigned char *vc = p.voiced ? p.voiced + (long long)a * p.n_frames : nullptr;
    float4 b = make_float4(0.f, 0.f, 0.f, 0.f);
    int nv = 0;
    constexpr int LD = SCAN_CHUNK / 2;
    if (d0 < p.Dp) {
        for (int t0 = t_start; t0 < t_end; t0 += LD) {
            float4 r[LD];
#pragma unroll
            for (int i = 0; i < LD; ++i) r[i] = *reinterpret_cast<const float4 *>(C + (long long)min(t0 + i, t_end - 1) * p.Dp + d0);
            for (int pl = 1; pl < p.c_planes; ++pl) {              // the partial maps of a split-K contraction, plane 0 first; a plane's loads together
                float4 w[LD];
#pragma unroll
                for (int i = 0; i < LD; ++i) w[i] = *reinterpret_cast<const float4 *>(C + (long long)min(t0 + i, t_end - 1) * p.Dp + d0 + pl * p.c_plane_stride);
#pragma unroll
                for (int i = 0; i < LD; ++i) { r[i].x += w[i].x; r[i].y += w[i].y; r[i].z += w[i].z; r[i].w += w[i].w; }
            }
            // (the gate flags of the batch as a mask, one byte per lane: a load of vc[t] inside the recursion is a round trip per step)
            const unsigned vm = (unsigned)__ballot((int)(threadIdx.x & 63) < LD && t0 + (int)(threadIdx.x & 63) < t_end && (!vc || vc[t0 + (threadIdx.x & 63)] != 0));
#pragma unroll
            for (int i = 0; i < LD; ++i)
                if ((vm >> i) & 1u) {
                    b.x = iir_step(p.mu, b.x, p.one_minus_mu, r[i].x); b.y = iir_step(p.mu, b.y, p.one_minus_mu, r[i].y);
                    b.z = iir_step(p.mu, b.z, p.one_minus_mu, r[i].z); b.w = iir_step(p.mu, b.w, p.one_minus_mu, r[i].w);
                    ++nv;
                }
        }
        float *out = p.part + ((long long)a * p.n_chunks + c) * p.D;
        if (d0 < p.D) out[d0] = b.x;
        if (d0 + 1 < p.D) out[d0 + 1] = b.y;
        if (d0 + 2 < p.D) out[d0 + 2] = b.z;
        if (d0 + 3 < p.D) out[d0 + 3] = b.w;
    }
    if (q == 0) p.nvoiced[(long long)a * p.n_chunks + c] = nv;
}

constexpr int CARRY_TILE = 64;                                       // chunks whose partial results (x planes) a thread holds in flight
__global__ __launch_bounds__(512) void k_scan_carry(ScanPickArgs p)
{
    __shared__ float spow[SCAN_CHUNK + 1];                          // 0.8f^n
    __shared__ float sg[CARRY_TILE];                                // 0.8f^(voiced frames of the chunk), the tile's chunks
    __shared__ int s_lv;
    // grid (arrays, slices of 64 delays): one wave per slice spreads the strided loads over many CUs (17.7 -> 16 us; four waves
    // per slice fetching side by side into LDS with one of them composing measured 19.6 us: the composition wants registers)
    const int tl = threadIdx.x, d = blockIdx.y * blockDim.x + tl, a = blockIdx.x;
    if (p.mode == 1 && a == 0 && d == 0) { *p.n_list = 0; *p.n_clist = 0; if (p.n_list_full) *p.n_list_full = 0; }   // adaptive SRP precision: the repair lists start empty
    if (tl == 0) s_lv = -1;
    for (int n = tl; n <= SCAN_CHUNK; n += blockDim.x) {
        float g = 1.f;
        for (int i = 0; i < n; ++i) g *= p.mu;
        spow[n] = g;
    }
    const int *nvp = p.nvoiced + (long long)a * p.n_chunks;
    const bool act = d < p.D;
    float E = act ? p.state_in[(long long)a * p.D + d] : 0.f;
    const float *pp = p.part + (long long)a * p.n_chunks * p.D + d;
    float *ep = p.e_start + (long long)a * p.n_chunks * p.D + d;
    for (int c0 = 0; c0 < p.n_chunks; c0 += CARRY_TILE) {            // a tile's loads all in flight, then the serial composition
        float b[CARRY_TILE];
        if (act) {
#pragma unroll
            for (int i = 0; i < CARRY_TILE; ++i) b[i] = pp[(long long)min(c0 + i, p.n_chunks - 1) * p.D];
            if (p.part_planes == 2) {                                 // the contraction's two K halves, half 0 first
                float b1[CARRY_TILE];
#pragma unroll
                for (int i = 0; i < CARRY_TILE; ++i) b1[i] = pp[(long long)min(c0 + i, p.n_chunks - 1) * p.D + p.part_plane_stride];
#pragma unroll
                for (int i = 0; i < CARRY_TILE; ++i) b[i] += b1[i];
            }
        }
        int nvr[2];                                                  // (at least 64 threads: at most two chunks of the tile per thread)
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int i = tl + k * (int)blockDim.x;
            nvr[k] = i < CARRY_TILE && c0 + i < p.n_chunks ? nvp[c0 + i] : 0;
        }
        __syncthreads();                                             // spow written; the previous tile's sg consumed
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int i = tl + k * (int)blockDim.x, n = nvr[k];
            if (i >= CARRY_TILE) break;
            sg[i] = spow[n];
            // the chunk that holds the array's last frame whose energy advanced (= its last frame without the gate): that frame is
            // always repaired, so that the state handed to the next call is exact
            if (p.mode == 1 && n > 0) atomicMax(&s_lv, c0 + i);
        }
        __syncthreads();
        if (act) {
#pragma unroll
            for (int i = 0; i < CARRY_TILE; ++i)
                if (c0 + i < p.n_chunks) { ep[(long long)(c0 + i) * p.D] = E; E = sg[i] * E + b[i]; }
        }
    }
    if (act) p.state_out[(long long)a * p.D + d] = E;                // _prevEnergyInDOA (:143)
    __syncthreads();
    if (p.mode == 1 && d == 0) p.last_vchunk[a] = s_lv;              // (every slice finds the same chunk; slice 0 writes it)
}

// selectDOA of one frame by one wave (SteeringBeamforming.cpp:146-195): En = the frame's normalised energies (LDS), lane
// evaluates the sign / median-3 / second-derivative chain at PL consecutive positions, the S maxima are found by a DPP
// reduction of the lanes' maxima and a ballot (first index on ties) and written by lane 0.  SENS (coarse pass of the adaptive SRP precision): also
// decides whether the picks could come out differently on the exact map; returns that decision (uniform over the wave).
//
// Adaptive SRP precision: the correlation map comes from ONE fp16 MFMA product per k-step (error sigma ~3e-6 of the largest
// possible map value instead of ~4e-7 for the three-product split).  With |error| of a normalised energy difference
// bounded by tau,
//   * a position dd is UNCERTAIN if any of the four first differences its second derivative reads (the sign / median-3
//     chain of selectDOA) lies within +-tau of zero -- its sd could then be -En, 0 or +En[dd+1];
//   * the frame is sensitive if an uncertain position could reach the S-th picked value (|En[dd+1]| >= v_S - tau; any
//     uncertain position at all when fewer than S positive peaks exist), or if two of the S+1 largest candidates lie within
//     tau of each other (their order, or which one is the last pick, is open), or if the last pick is a zero entry (fewer
//     than S positive candidates) while some candidate's energy lies within tau of zero (it could be on either side).
// Every other frame's picks are the exact map's picks.
// Maximum over the 64 lanes on v_max_f32 with a DPP source operand (quad swaps, row rotations, the two cross-row broadcasts
// of gfx9); taken from lane 63, uniform.  (A DPP read needs two wait states after the vector write of its source.)
__device__ __forceinline__ float wave_max64(float v)
{
    asm("s_nop 1\n\tv_max_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_ror:4 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
        "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
        "s_nop 1"
        : "+v"(v));
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ float wave_min64(float v) { return -wave_max64(-v); }

// (E - mn) / d, d = -2 mn = 30 P (:155-156), as the correctly rounded quotient from the correctly rounded reciprocal r of d:
// q0 = RN(x r), e = x - q0 d (exact in an fma), q = RN(q0 + e r) -- the IEEE quotient for every x in range (Markstein), in
// three instructions instead of the ten of the scaled division sequence.
__device__ __forceinline__ float normalised_energy(float E, float mn, float d, float r)
{
    const float x = E - mn;
    const float q0 = x * r;
    const float e = fmaf(-q0, d, x);
    return fmaf(e, r, q0);
}

// selectDOA of one frame by one wave; the S (SENS: S + 1) picks go to obin / oval (LDS) from lane 0.
template <bool SENS, int PL>
__device__ __forceinline__ bool wave_pick_pl(const float *En, int D, int S, float tau, int *obin, float *oval, int lane)
{
    // lane owns the PL consecutive positions dd = PL lane + i: every first difference, sign and median is formed once.
    // Everything that depends on the lane only (load offsets with the edges replicated as median_filter does: j = -1 reads
    // j = 0, j > D - 2 reads j = D - 2; the penalty that removes the positions past D - 3) is a per-lane VALUE, not a lane mask.
    const int b = PL * lane;
    float df[PL + 3];                                           // En[j + 1] - En[j], j = clamp(b - 1 + c, 0, D - 2)
#pragma unroll
    for (int c = 0; c < PL + 3; ++c) {
        const int j = min(max(b - 1 + c, 0), max(D - 2, 0));
        df[c] = En[min(j + 1, D - 1)] - En[j];
    }
    float fd[PL + 3], md[PL + 1];
#pragma unroll
    for (int c = 0; c < PL + 3; ++c) fd[c] = df[c] < 0.f ? 1.f : 0.f;       // fd(j) = 1 if En[j+1] - En[j] < 0 else 0
#pragma unroll
    for (int i = 0; i < PL + 1; ++i) md[i] = __builtin_amdgcn_fmed3f(fd[i], fd[i + 1], fd[i + 2]);       // median filter  :164
    float sdv[PL];
    float umax = -INFINITY;                                     // SENS: largest |En[dd+1]| over the uncertain positions
    float zmin = INFINITY;                                      // SENS: smallest |sd| over the positions with a non-zero second derivative
#pragma unroll
    for (int i = 0; i < PL; ++i) {
        const float pen = b + i < D - 2 ? -0.f : -INFINITY;      // x + (-0) = x for every x
        const float ed1 = En[min(b + i + 1, D - 1)];            // En[dd + 1]
        sdv[i] = (md[i + 1] - md[i]) * ed1 + pen;                // :170-173
        if (SENS) {
            const float dmin = fminf(fminf(fabsf(df[i]), fabsf(df[i + 1])), fminf(fabsf(df[i + 2]), fabsf(df[i + 3])));
            umax = fmaxf(umax, dmin <= tau ? fabsf(ed1) + pen : -INFINITY);        // (a trough of negative energy is a candidate too: sd = -En > 0)
            zmin = fminf(zmin, md[i + 1] != md[i] ? fabsf(ed1) - pen : INFINITY);  // a candidate whose value could be on either side of zero
        }
    }
    bool sens = false;
    float prev_bv = 0.f, v_last = 0.f;
    for (int s = 0; s < S; ++s) {                                  // :185-194
        float lv = sdv[0]; int li = 0;                            // this lane's first maximum
#pragma unroll
        for (int i = 1; i < PL; ++i)
            if (sdv[i] > lv) { lv = sdv[i]; li = i; }
        // the wave's maximum, first index on ties: lanes are in position order
        const float g = wave_max64(lv);
        const unsigned long long mk = __ballot(lv == g);
        const int fl = mk ? __ffsll((long long)mk) - 1 : 0;
        const int bi = PL * fl + __builtin_amdgcn_readlane(li, fl);
        const float bv = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(lv), fl));
#pragma unroll
        for (int i = 0; i < PL; ++i)
            if (bi == b + i) sdv[i] = 0.f;                         // _secondDerivative[maxIdx] = 0
        if (SENS && s > 0 && prev_bv > 0.f && prev_bv - bv <= tau) sens = true;
        prev_bv = bv;
        v_last = bv;
        if (lane == 0) { obin[s] = bi + 1; oval[s] = bv; }         // doaIdx2angle(maxIdx+1) and the outputs: by the caller
    }
    if (SENS) {
        // The three wave-wide questions are "does ANY position reach a bound", so they are ballots of per-lane answers, not reductions
        // (round 5: the runner-up's pick round and two DPP reductions per frame were 9 of this kernel's 48 us):
        // the runner-up of the last pick within tau of it <=> a remaining second-derivative value >= that pick - tau,
        float lv = sdv[0];
#pragma unroll
        for (int i = 1; i < PL; ++i) lv = fmaxf(lv, sdv[i]);
        if (prev_bv > 0.f && __ballot(prev_bv - lv <= tau) != 0ull) sens = true;
        // an uncertain position whose energy reaches the last pick,
        if (__ballot(v_last > 0.f ? umax >= v_last - tau : umax > -INFINITY) != 0ull) sens = true;
        // the last pick is (within tau of) one of the zero entries: a candidate whose normalised energy is within tau of zero
        // could be a positive peak -- picked before every zero -- or a negative one
        if (v_last <= tau && __ballot(zmin <= tau) != 0ull) sens = true;
    }
    return sens;
}

// The candidate columns of a flagged frame (CandArgs in mca_internal.h has the argument), by one wave, into cm[words] (LDS, one bit per
// column).  full: every column (a frame flagged for the state's sake or on an unsure row).  Returns whether every column was taken.
// The lower bound v of the exact S-th pick comes from GUARANTEED peaks: with every first difference within tau of zero taken as of
// either sign, the median-filtered sign chain (:161-165) has a lowest and a highest possible value at every position; where it is
// pinned to "rising" at i and pinned to "falling" at i + g (g <= 3, nothing pinned in between), the exact chain steps up somewhere in
// i .. i + g - 1, i.e. the exact map has a peak there whose value is at least the smallest coarse energy of those positions (- tau / 2).
// A flat top -- two or three neighbouring delays within tau, the usual reason for a flag -- is such a window.  (Windows are disjoint, so
// the S-th largest of their bounds would bound the S-th pick; the adaptive precision runs with one source, api.hip.)
// (One source: the bound is the largest window's.  Written for few live registers -- every energy is read from LDS where it is used --:
// the call sits in k_scan_pick, whose recursion phase sets the kernel's register count.)
template <int PL>
__device__ __forceinline__ bool wave_candidates(const float *En, int D, float tau, bool full, unsigned *cm, int words, int lane)
{
    constexpr int G = 3;
    const int b = PL * lane;
    float thr = 0.f;
    if (!full) {
        unsigned neg = 0u, unc = 0u;                                        // bit c: first difference j = clamp(b - 1 + c, 0, D - 2) (as wave_pick_pl)
#pragma unroll 4
        for (int c = 0; c < PL + G + 3; ++c) {
            const int j = min(max(b - 1 + c, 0), max(D - 2, 0));
            const float df = En[min(j + 1, D - 1)] - En[j];
            neg |= (df < 0.f ? 1u : 0u) << c;
            unc |= (fabsf(df) <= tau ? 1u : 0u) << c;
        }
        const unsigned lo = neg & ~unc, hi = neg | unc;                     // the sign bit fd at its lowest / highest
        const unsigned mlo = (lo & (lo >> 1)) | (lo & (lo >> 2)) | ((lo >> 1) & (lo >> 2));     // median of three, bit i: position b + i
        const unsigned mhi = (hi & (hi >> 1)) | (hi & (hi >> 2)) | ((hi >> 1) & (hi >> 2));
        const unsigned up = ~mhi, down = mlo, pinned = up | down;           // pinned to 0 (rising) / to 1 (falling)
        float lv = -INFINITY;
#pragma nounroll
        for (int i = 0; i < PL; ++i) {
            float mn = INFINITY;
            bool open = ((up >> i) & 1u) != 0u;
#pragma unroll
            for (int g = 1; g <= G; ++g) {
                mn = fminf(mn, En[min(b + i + g, D - 1)]);                  // En[(b + i + g - 1) + 1]: the value of a step up at position b + i + g - 1
                if (open && ((down >> (i + g)) & 1u) && b + i + g - 1 < D - 2) lv = fmaxf(lv, mn);
                if ((pinned >> (i + g)) & 1u) open = false;                 // (a pinned position ends the window either way)
            }
        }
        const float vcert = wave_max64(lv);
        // no guaranteed positive peak, or one within reach of the zero entries (:188 picks the first of those): every column
        if (!(vcert > 2.f * tau)) full = true;
        thr = vcert - tau;
    }
    for (int w = lane; w < words; w += 64) {
        const int c0 = 32 * w;
        cm[w] = !full || c0 >= D ? 0u : (D - c0 >= 32 ? 0xffffffffu : (1u << (D - c0)) - 1u);
    }
    wave_lds_fence();
    if (!full) {
#pragma unroll 3
        for (int i = 0; i < PL; ++i) {
            if (b + i < D - 2 && fabsf(En[min(b + i + 1, D - 1)]) >= thr) {
                // position b + i reads En[b + i - 1 .. b + i + 3] (two first differences either side of its own two, :159-173)
                const int lo = max(b + i - 1, 0), hi = min(b + i + 3, D - 1);
                const unsigned long long m = ((1ull << (hi - lo + 1)) - 1ull) << (lo & 31);
                atomicOr(&cm[lo >> 5], (unsigned)m);
                if (m >> 32) atomicOr(&cm[(lo >> 5) + 1], (unsigned)(m >> 32));
            }
        }
    }
    wave_lds_fence();
    return full;
}

// k_scan_pick<PL, MODE>: grid (chunks, arrays), 512 threads.  Per batch of SCAN_SUB frames: (1) thread d runs the recursion of
// delay d and leaves the normalised energies in LDS, (2) one wave per frame picks the peaks into LDS, (3) the batch's outputs
// are stored together.  MODE 1 (coarse pass of the adaptive SRP precision): a frame whose picks are sensitive to the fp16
// error -- and the last frame of the call, so that the state handed to the next call is exact too -- is flagged, and a wave
// plans its repair in step (3): the groups of REPAIR_GROUP frames that hold its own row and the REPAIR_WARM rows before it
// (its energy is 0.2 sum_k 0.8^k C_{t-k} over the frames that advanced the recursion: with the power gate only the voiced ones,
// BeamformingSeparationAndLocalisation.cpp:87) go onto the repair list (once: `need` is a test-and-set per group), and the chunk
// learns which chunk its second pick has to restart from.  need / chunk_from are cleaned by the kernels that consume them
// (k_repair_patch, k_scan_repick); the list's length is reset by k_scan_carry.
template <int PL, int MODE>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4))) void k_scan_pick(ScanPickArgs p)   // (two workgroups per CU: <= 128 VGPRs; the plan's last round spills two)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float *sEn = reinterpret_cast<float *>(smem_raw);                   // [SCAN_SUB][Dl]
    __shared__ int s_bin[SCAN_SUB * MCA_MAX_SOURCES];
    __shared__ float s_val[SCAN_SUB * MCA_MAX_SOURCES];
    __shared__ unsigned s_flagmask;                                     // MODE 1: the flags of a batch of SCAN_SUB (<= 32) frames
    __shared__ unsigned s_cm[MODE >= 1 ? 10 : 1][CAND_WORDS_MAX];       // MODE 1: the candidate columns of the flagged frame a wave is planning
    // MODE 1, candidate columns without the gate: the plan of the whole batch is put together in LDS -- the 12 repair units a chunk's frames
    // can need (its own 8 and the 4 before it), their column masks, the earliest row -- and goes to the device-side lists in ONE round
    constexpr int PLAN_UNITS = (SCAN_CHUNK + REPAIR_WARM) / REPAIR_GROUP;
    __shared__ unsigned s_um[MODE >= 1 ? PLAN_UNITS : 1][CAND_WORDS_MAX];
    __shared__ unsigned s_uneed, s_ncol, s_nall;
    __shared__ unsigned s_uneed_full;                                   // MODE 2: the units of the frames that want whole rows (ScanPickArgs::list_full)
    __shared__ int s_umin, s_ue[MODE >= 1 ? PLAN_UNITS : 1];
    static_assert(REPAIR_WARM % REPAIR_GROUP == 0 && SCAN_CHUNK % REPAIR_GROUP == 0, "the units of a chunk's plan are whole");
    static_assert(SCAN_SUB <= 32, "one 32-bit mask per batch");
    static_assert(SCAN_CHUNK <= 64, "one ballot per chunk");
    const int d = threadIdx.x, lane = d & 63, wave = d >> 6, nwaves = blockDim.x >> 6;
    const int a = blockIdx.y, D = p.D, Dl = p.Dp + 8, S = p.S;
    const int t_start = blockIdx.x * p.chunk;
    const int t_end = min(t_start + p.chunk, p.n_frames);
    const bool act = d < D;
    const float mu = p.mu, omu = p.one_minus_mu;
    const float *C = p.C + (long long)a * p.n_frames * p.Dp;
    const unsigned char *vc = p.voiced ? p.voiced + (long long)a * p.n_frames : nullptr;
    const float mn = -15.f * (float)p.P, nd = -2.f * mn, nr = p.inv_norm;
    float E = 0.f;
    if (p.lookback > 0) {
        // E at the start of this chunk from the chunk-local results b_j (the recursion run from zero inside chunk j, k_scan_partial or
        // the contraction's epilogue) of the chunks before it: E_c = g E_{c-1} + b_{c-1}, g = 0.8f^32 -- the composition k_scan_carry
        // runs over all chunks of an array in order, cut off `lookback` chunks back (or at the array's state at entry)
        constexpr int LB = 4;
        if (act) {
            const int c = blockIdx.x, j0 = max(0, c - LB);
            float g = 1.f;
            for (int i = 0; i < p.chunk; ++i) g *= mu;                                   // (as k_scan_carry's table)
            const float *pp = p.part + (long long)a * p.n_chunks * D + d;
            float b[LB];
#pragma unroll
            for (int i = 0; i < LB; ++i) {
                const int j = min(j0 + i, p.n_chunks - 1);
                b[i] = pp[(long long)j * D];
                if (p.part_planes == 2) b[i] += pp[(long long)j * D + p.part_plane_stride];   // the contraction's two K halves, half 0 first
            }
            E = j0 == 0 ? p.state_in[(long long)a * D + d] : 0.f;
#pragma unroll
            for (int i = 0; i < LB; ++i)
                if (j0 + i < c) E = g * E + b[i];
            p.e_start[((long long)a * p.n_chunks + c) * D + d] = E;                       // (k_scan_repick restarts from it)
        }
    } else if (act) {
        E = p.e_start[((long long)a * p.n_chunks + blockIdx.x) * D + d];
    }
    const int last_vchunk = p.lookback > 0 ? p.n_chunks - 1 : (MODE >= 1 ? p.last_vchunk[a] : -1);
    if (MODE >= 1 && p.lookback > 0 && (int)blockIdx.x == p.n_chunks - 1 && d == 0) p.last_vchunk[a] = p.n_chunks - 1;
    // MODE 1: the array's last frame that advances the recursion (this chunk holds it) is always repaired
    int t_force = -1;
    if (MODE >= 1 && (int)blockIdx.x == last_vchunk && !p.lazy) {        // (lazy tails: the next call repairs this call's last rows if it needs them)
        const int u = t_start + lane;
        const unsigned long long m = __ballot(u < t_end && (!vc || vc[u] != 0));
        if (m) t_force = t_start + 63 - __clzll((long long)m);
    }
    for (int ts = t_start; ts < t_end; ts += SCAN_SUB) {
        const int te = min(ts + SCAN_SUB, t_end);
        if (d == 0) s_flagmask = 0u;
        if (MODE >= 1 && p.umask && !vc) {
            if (d < PLAN_UNITS * CAND_WORDS_MAX) (&s_um[0][0])[d] = 0u;
            if (d == 0) { s_uneed = 0u; s_ncol = 0u; s_nall = 0u; s_umin = 0x7fffffff; if (MODE == 2) s_uneed_full = 0u; }
        }
        // the batch's frames that passed the power gate (all without it), one bit each: every wave works the mask out for itself
        // from one byte per lane (a load of vc[t] inside the recursion made every step of it a round trip to memory)
        const unsigned vmask = (unsigned)__ballot(lane < 32 && ts + (lane & 31) < te && (!vc || vc[ts + (lane & 31)] != 0));
        // MODE 1: frames the coarse analysis marked unsure (StftPhatArgs::unsure) are repaired with their six successors (0.8^7 of their
        // error is left after those): bit i of um = frame ts - 6 + i
        unsigned long long um = 0;
        if (MODE >= 1 && p.unsure) {
            const int u = ts - 6 + lane;
            um = __ballot(lane < SCAN_SUB + 6 && u >= 0 && u < te && p.unsure[(long long)a * p.n_frames + u] != 0);
        }
        if (act) {
            auto load_rows = [&](float (&c8)[SCAN_LD], int t0) {
                if (p.c_planes == 1) {
#pragma unroll
                    for (int i = 0; i < SCAN_LD; ++i) c8[i] = C[(long long)min(t0 + i, te - 1) * p.Dp + d];
                } else {
                    csum_rows(c8, C, [&](int i) { return (long long)min(t0 + i, te - 1) * p.Dp + d; }, p.c_planes, p.c_plane_stride);
                }
            };
            for (int t0 = ts; t0 < te; t0 += SCAN_LD) {       // (requesting the next rows before these are used measured the same: 42 us)
                float c8[SCAN_LD];
                load_rows(c8, t0);
#pragma unroll
                for (int i = 0; i < SCAN_LD; ++i) {
                    const int t = t0 + i;
                    if (t < te) {
                        if ((vmask >> (t - ts)) & 1u) E = iir_step(mu, E, omu, c8[i]);   // :134-140
                        if (p.energy) p.energy[((long long)a * p.n_frames + t) * D + d] = E;
                        sEn[(t - ts) * Dl + d] = normalised_energy(E, mn, nd, nr);   // :155-156
                        if (MODE >= 1 && p.lazy) {
                            // lazy tails: the coarse rows of the call's last HIST_FRAMES frames and the energies in front of them
                            const int j = t - (p.n_frames - HIST_FRAMES);
                            if (j >= 0) p.hist_C_out[((long long)a * HIST_FRAMES + j) * p.Dp + d] = c8[i];
                            else if (j == -1) p.e_hist_out[(long long)a * D + d] = E;
                        }
                    }
                }
            }
        }
        __syncthreads();
        for (int tl = wave; tl < te - ts; tl += nwaves) {
            const int t = ts + tl;
            if (!((vmask >> tl) & 1u)) {                                // gated out: selectDOA is not reached (:87)
                if (lane < S) s_bin[tl * MCA_MAX_SOURCES + lane] = -1;
                continue;
            }
#ifdef MCA_ABL_PICK      /* measurement only: wrong results */
            const bool sens = false; if (lane < S) { s_bin[tl * MCA_MAX_SOURCES + lane] = 5; s_val[tl * MCA_MAX_SOURCES + lane] = sEn[tl * Dl + lane]; }
#else
            const bool sens = wave_pick_pl<MODE >= 1, PL>(sEn + tl * Dl, D, S, p.tau, s_bin + tl * MCA_MAX_SOURCES, s_val + tl * MCA_MAX_SOURCES, lane);
#endif
            if (MODE >= 1 && lane == 0 && (sens || t == t_force || ((um >> tl) & 0x7full) != 0)) atomicOr(&s_flagmask, 1u << tl);
        }
        __syncthreads();
        // the batch's outputs, one thread per (frame, source)
        if (d < (te - ts) * S) {
            const int tl = d / S, sidx = d - tl * S;
            const int bin = s_bin[tl * MCA_MAX_SOURCES + sidx];
            const long long o = ((long long)a * p.n_frames + ts + tl) * S + sidx;
            p.doa_bin[o] = bin;
            if (bin >= 0) {
                if (p.doa_rad) p.doa_rad[o] = p.grid[bin];
                if (p.prob) p.prob[o] = s_val[tl * MCA_MAX_SOURCES + sidx];
            }
        }
        if (MODE >= 1) {
            const unsigned fm = s_flagmask;
            if (d < te - ts) p.flags[(long long)a * p.n_frames + ts + d] = (fm >> d) & 1u;
            // plan the repair of the flagged frames, one wave per frame
            int k = 0;
            const bool plan_lds = p.umask != nullptr && !vc;            // (candidate-column calls are ungated: a frame needs its own row and the 16 before it)
            if (plan_lds) {
                const int r_first = t_start - REPAIR_WARM;                // first row of local unit 0
                const int u_lo = p.hist_valid ? -HIST_FRAMES : 0;         // lazy tails: the rows before the call are the previous call's last ones
                for (unsigned rest = fm; rest; rest &= rest - 1, ++k) {
                    if (k % nwaves != wave) continue;
                    const int t = ts + __ffs((int)rest) - 1;
                    const bool want_all = t == t_force || ((um >> (t - ts)) & 0x7full) != 0;
                    // MODE 2 (two work lists: contexts that flag whole rows by construction -- eager tails, unsure rows): a frame that takes every
                    // column goes onto the WHOLE-ROW list (k_srp_gemm_repair + k_repair_patch), the others onto the candidate list as in MODE 1
                    bool all = MODE == 2 && want_all;
                    if (!all) all = wave_candidates<PL>(sEn + (t - ts) * Dl, D, p.tau, want_all, s_cm[wave], p.umask_words, lane);
                    const bool to_full = MODE == 2 && all;
                    const int r_lo = max(t - REPAIR_WARM, u_lo);
                    // a unit is needed if one of the frame's rows in it is not a frame of exact zeros (those are zero in both maps)
                    const int r = r_lo + lane;
                    if (r <= t && !(p.dead && r >= 0 && p.dead[(long long)a * p.n_frames + r] != 0)) atomicOr(to_full ? &s_uneed_full : &s_uneed, 1u << ((r - r_first) >> 2));
                    if (!to_full && lane < p.umask_words) {
                        const unsigned m = s_cm[wave][lane];
                        if (m)
#pragma nounroll
                            for (int lu = (r_lo - r_first) >> 2; lu <= (t - r_first) >> 2; ++lu) atomicOr(&s_um[lu][lane], m);
                    }
                    if (lane == 0) {
                        int nc = 0;
                        if (!to_full) {
#pragma nounroll
                            for (int w = 0; w < p.umask_words; ++w) nc += __popc(s_cm[wave][w]);
                        }
                        atomicAdd(&s_ncol, (unsigned)nc);
                        if (all) atomicAdd(&s_nall, 1u);
                        atomicMin(&s_umin, r_lo);
                    }
                }
                __syncthreads();
                if (fm) {
                    const unsigned needed = s_uneed;
                    if (d < PLAN_UNITS && ((needed >> d) & 1u)) {
                        const int r0 = r_first + REPAIR_GROUP * d;
                        const int e = r0 >= 0 ? a * p.groups_per_array + r0 / REPAIR_GROUP : p.hist_base + a * HIST_UNITS + (HIST_FRAMES + r0) / REPAIR_GROUP;
                        s_ue[d] = e;
                        if (atomicExch(&p.need[e], 1) == 0) {
                            p.list[atomicAdd(p.n_list, 1)] = e;
                            atomicAdd(&p.stats[1], 1ull);
                        }
                    }
                    if (MODE == 2 && d >= 128 && d < 128 + PLAN_UNITS && ((s_uneed_full >> (d - 128)) & 1u)) {      // (a third wave)
                        const int r0 = r_first + REPAIR_GROUP * (d - 128);
                        const int e = r0 >= 0 ? a * p.groups_per_array + r0 / REPAIR_GROUP : p.hist_base + a * HIST_UNITS + (HIST_FRAMES + r0) / REPAIR_GROUP;
                        if (atomicExch(&p.need_full[e], 1) == 0) {
                            p.list_full[atomicAdd(p.n_list_full, 1)] = e;
                            atomicAdd(&p.stats[1], 1ull);
                        }
                    }
                    if (d == 64) {                                        // (another wave than the units')
                        const int ci = a * p.n_chunks + (int)blockIdx.x, u_min = s_umin;
                        if (atomicMin(&p.chunk_from[ci], u_min < 0 ? -1 : u_min / SCAN_CHUNK) >= p.n_chunks) p.clist[atomicAdd(p.n_clist, 1)] = ci;   // (p.chunk == SCAN_CHUNK)
                        atomicAdd(&p.stats[0], (unsigned long long)__popc(fm));
                        atomicAdd(&p.stats[2], (unsigned long long)s_ncol);
                        if (s_nall) atomicAdd(&p.stats[3], (unsigned long long)s_nall);
                    }
                    __syncthreads();
                    if (d < PLAN_UNITS * 32) {                            // (thread = (unit, word): 32 words per unit are more than any grid has)
                        const int j = d >> 5, w = d & 31;
                        static_assert(CAND_WORDS_MAX <= 32, "a word per thread");
                        if (w < p.umask_words && ((needed >> j) & 1u)) {
                            const unsigned m = s_um[j][w];
                            if (m) atomicOr(&p.umask[s_ue[j] * p.umask_words + w], m);
                        }
                    }
                }
            }
            for (unsigned rest = plan_lds ? 0u : fm; rest; rest &= rest - 1, ++k) {      // whole-row calls (and the gate): round 4's plan, a frame at a time
                if (k % nwaves != wave) continue;
                const int t = ts + __ffs((int)rest) - 1;
                // the rows this frame's energy depends on: its own and those of the REPAIR_WARM frames before it that advanced
                // the recursion (all of them without the gate, the voiced ones with it), 64 frames per step backwards
                int remaining = REPAIR_WARM + 1, u_min = t;
                const int u_lo = (p.hist_valid && !vc) ? -HIST_FRAMES : 0;         // lazy tails: the rows before the call are the previous call's last ones
                for (int hi = t; hi >= u_lo && remaining > 0; hi -= 64) {
                    const int u = hi - 63 + lane;                        // lane 63 = frame hi
                    const bool on = u >= u_lo && (u < 0 || !vc || vc[u] != 0);
                    const unsigned long long m = __ballot(on);
                    const int above = lane < 63 ? __popcll(m >> (lane + 1)) : 0;   // advancing frames after u in this window
                    const bool take = on && above < remaining;
                    if (take) {
                        const int e = u >= 0 ? a * p.groups_per_array + u / REPAIR_GROUP : p.hist_base + a * HIST_UNITS + (HIST_FRAMES + u) / REPAIR_GROUP;
                        if (atomicExch(&p.need[e], 1) == 0) {
                            p.list[atomicAdd(p.n_list, 1)] = e;
                            atomicAdd(&p.stats[1], 1ull);
                        }
                    }
                    const unsigned long long mt = __ballot(take);
                    if (mt) u_min = hi - 63 + (__ffsll((long long)mt) - 1);
                    remaining -= __popcll(m);
                }
                if (lane == 0) {
                    // the second pick of this chunk restarts from the (coarse) start value of the chunk that holds the earliest of those rows
                    const int ci = a * p.n_chunks + (int)blockIdx.x;
                    if (atomicMin(&p.chunk_from[ci], u_min < 0 ? -1 : u_min / p.chunk) >= p.n_chunks) p.clist[atomicAdd(p.n_clist, 1)] = ci;   // the chunk's first flagged frame (-1: from the history)
                    atomicAdd(&p.stats[0], 1ull);
                }
            }
        }
        __syncthreads();
    }
    if (p.lookback > 0 && act && (int)blockIdx.x == p.n_chunks - 1) p.state_out[(long long)a * D + d] = E;   // _prevEnergyInDOA (:143)
}
template __global__ void k_scan_pick<2, 0>(ScanPickArgs);
template __global__ void k_scan_pick<6, 0>(ScanPickArgs);
template __global__ void k_scan_pick<8, 0>(ScanPickArgs);
template __global__ void k_scan_pick<2, 1>(ScanPickArgs);
template __global__ void k_scan_pick<6, 1>(ScanPickArgs);
template __global__ void k_scan_pick<8, 1>(ScanPickArgs);
template __global__ void k_scan_pick<2, 2>(ScanPickArgs);
template __global__ void k_scan_pick<6, 2>(ScanPickArgs);
template __global__ void k_scan_pick<8, 2>(ScanPickArgs);

// --------------------------------------------------------------------------------------
// k_scan_repick / k_repair_patch -- adaptive SRP precision: the second pick of the flagged frames on the exact rows
// --------------------------------------------------------------------------------------
// k_scan_repick: a fixed, moderate grid walks the list of the chunks that hold a flagged frame (a grid of all chunks spends
// ~25 ns per empty workgroup: 29 us for 1 024 chunks of which 40 are listed, 7 us for 128).  After the
// flagged frames' rows and the REPAIR_WARM advancing rows before them were recomputed with the three-product split and patched
// into C, the recursion restarts from the (coarse) start value of the chunk that holds the earliest of those rows -- this
// chunk or the one before it without the gate, possibly further back across a silence with it -- and walks the patched map,
// so that at a flagged frame at most 0.8^(REPAIR_WARM+1) of the coarse error is left; the flagged frames are picked again
// (every other frame's coarse pick is safe).  32 rows in flight per thread, the picks of a batch of 32 frames after it.
// The chunk with the array's last advancing frame leaves the exact state.
constexpr int REPICK_B = 32;

template <int PL>
__global__ __launch_bounds__(512) void k_scan_repick(ScanPickArgs p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float *sEn = reinterpret_cast<float *>(smem_raw);                   // [REPICK_B][Dl]
    __shared__ int s_bin[REPICK_B * MCA_MAX_SOURCES];
    __shared__ float s_val[REPICK_B * MCA_MAX_SOURCES];
    const int d = threadIdx.x, lane = d & 63, wave = d >> 6, nwaves = blockDim.x >> 6;
    const int D = p.D, Dl = p.Dp + 8, S = p.S;
    const bool act = d < D;
    const float mu = p.mu, omu = p.one_minus_mu;
    const float mn = -15.f * (float)p.P, nd = -2.f * mn, nr = p.inv_norm;
    const int n_listed = *p.n_clist;
    if (p.probe && blockIdx.x == 0 && d == 0) {
        // how much this call had to repair, for the host's choice between this mode and plain FP16X3 for later calls: page-locked
        // host memory, the sequence number last
        p.probe[0] = p.stats[0]; p.probe[1] = p.stats[1];
        __threadfence_system();
        __hip_atomic_store(&p.probe[2], p.probe_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    for (int li = blockIdx.x; li < n_listed; li += gridDim.x) {
    const int ci = p.clist[li];
    const int a = ci / p.n_chunks, chunk = ci - a * p.n_chunks;
    const int c_from = p.chunk_from[ci];
    const unsigned char *fl = p.flags + (long long)a * p.n_frames;
    const unsigned char *vc = p.voiced ? p.voiced + (long long)a * p.n_frames : nullptr;
    const int t_start = chunk * p.chunk;
    const int t_end = min(t_start + p.chunk, p.n_frames);
    const float *C = p.C + (long long)a * p.n_frames * p.Dp;
    float E = 0.f;
    // one batch of <= 32 frames from tb on: the recursion over the rows c32 (advancing frames: mask vm), then the second pick of the
    // flagged frames of this chunk (mask bm), one wave per frame
    auto batch = [&](int tb, int te, unsigned vm, unsigned bm, const float (&c32)[REPICK_B]) {
        if (act) {
#pragma unroll
            for (int i = 0; i < REPICK_B; ++i) {
                const int t = tb + i;
                if (t < te) {
                    if ((vm >> i) & 1u) E = iir_step(mu, E, omu, c32[i]);   // :134-140
                    if (t >= t_start && p.energy) p.energy[((long long)a * p.n_frames + t) * D + d] = E;
                    if ((bm >> i) & 1u) sEn[i * Dl + d] = normalised_energy(E, mn, nd, nr);   // :155-156
                }
            }
        }
        if (bm) {
            __syncthreads();
            for (int i = wave; i < te - tb; i += nwaves)
                if ((bm >> i) & 1u) wave_pick_pl<false, PL>(sEn + i * Dl, D, S, 0.f, s_bin + i * MCA_MAX_SOURCES, s_val + i * MCA_MAX_SOURCES, lane);
            __syncthreads();
            if (d < (te - tb) * S) {
                const int i = d / S, sidx = d - i * S;
                if ((bm >> i) & 1u) {
                    const int bin = s_bin[i * MCA_MAX_SOURCES + sidx];
                    const long long o = ((long long)a * p.n_frames + tb + i) * S + sidx;
                    p.doa_bin[o] = bin;
                    if (p.doa_rad) p.doa_rad[o] = p.grid[bin];
                    if (p.prob) p.prob[o] = s_val[i * MCA_MAX_SOURCES + sidx];
                }
            }
            __syncthreads();
        }
    };
    if (!vc && p.chunk == REPICK_B) {
        // Without the gate the walk starts in this chunk or in the one before it (REPAIR_WARM < chunk): both start values and the
        // rows of both chunks are requested before c_from is looked at -- two rounds of loads less on the critical path of a kernel
        // that is nothing but dependent loads (list -> chunk_from -> start value -> rows -> rows -> pick)
        const int cprev = max(chunk - 1, 0), te = t_end;
        const int u = t_start + (lane & 31);
        const unsigned bm = (unsigned)__ballot(lane < 32 && u < te && fl[u] != 0);
        float r0[REPICK_B], r1[REPICK_B];
        if (act) {
            const float e1 = p.e_start[((long long)a * p.n_chunks + chunk) * D + d], e0 = p.e_start[((long long)a * p.n_chunks + cprev) * D + d];
            csum_rows(r0, C, [&](int i) { return (long long)(cprev * REPICK_B + i) * p.Dp + d; }, p.c_planes, p.c_plane_stride);
            csum_rows(r1, C, [&](int i) { return (long long)min(t_start + i, te - 1) * p.Dp + d; }, p.c_planes, p.c_plane_stride);
            E = e1;
            if (c_from < 0) {
                // lazy tails: a flagged frame among the call's first REPAIR_WARM ones -- the walk starts in front of the PREVIOUS call's last
                // HIST_FRAMES rows (patched where this call's repair pass recomputed them) from the energies that call left there
                // (into the registers of the rows requested above: this is chunk 0, they are its own rows a second time)
                E = p.e_hist_in[(long long)a * D + d];
                static_assert(HIST_FRAMES <= REPICK_B, "the history rows reuse the previous chunk's registers");
#pragma unroll
                for (int j = 0; j < HIST_FRAMES; ++j) r0[j] = p.hist_C_in[((long long)a * HIST_FRAMES + j) * p.Dp + d];
#pragma unroll
                for (int j = 0; j < HIST_FRAMES; ++j) E = iir_step(mu, E, omu, r0[j]);
            } else if (c_from < chunk) {
                E = e0;
#pragma unroll
                for (int i = 0; i < REPICK_B; ++i) E = iir_step(mu, E, omu, r0[i]);     // (a whole chunk: it is not the array's last)
            }
        }
        batch(t_start, te, 0xffffffffu, bm, r1);
    } else {
        if (act) E = p.e_start[((long long)a * p.n_chunks + c_from) * D + d];
        for (int tb = c_from * p.chunk; tb < t_end; tb += REPICK_B) {
            const int te = min(tb + REPICK_B, t_end);
            // per batch: which frames advance the recursion (all without the gate), which are picked again (flagged, this chunk only);
            // every wave works the two masks out for itself from one byte per lane
            const int u = tb + (lane & 31);
            const unsigned vm = (unsigned)__ballot(lane < 32 && u < te && (!vc || vc[u] != 0));
            const unsigned bm = tb >= t_start ? (unsigned)__ballot(lane < 32 && u < te && fl[u] != 0) : 0u;
            float c32[REPICK_B];
            if (act) {
                csum_rows(c32, C, [&](int i) { return (long long)min(tb + i, te - 1) * p.Dp + d; }, p.c_planes, p.c_plane_stride);
            }
            batch(tb, te, vm, bm, c32);
        }
    }
    if (act && chunk == p.last_vchunk[a]) p.state_out[(long long)a * D + d] = E;   // _prevEnergyInDOA (:143), exact
    __syncthreads();                                                        // (every thread has read chunk_from[ci])
    if (d == 0) p.chunk_from[ci] = 0x7f7f7f7f;                              // consumed: no flagged frame
    }
    // the last workgroup to get here leaves both lists of the repair pass empty for the next call (every reader of n_list ran
    // before this kernel, every reader of n_clist is a workgroup of it that has counted itself in)
    // (only the workgroups that had a chunk to work on count themselves in -- min(grid, listed chunks) of them, a number every workgroup
    // knows; with nothing listed workgroup 0 does the reset: 256 fences and atomics on one word cost the kernel 3 of its 15 us)
    const int n_part = min((int)gridDim.x, n_listed);
    if (d == 0 && ((int)blockIdx.x < n_part || (n_part == 0 && blockIdx.x == 0))) {
        __threadfence();
        if (atomicAdd(p.n_clist + 1, 1) >= max(n_part, 1) - 1) { *p.n_list = 0; *p.n_clist = 0; p.n_clist[1] = 0; if (p.n_list_full) *p.n_list_full = 0; }
    }
}
template __global__ void k_scan_repick<2>(ScanPickArgs);
template __global__ void k_scan_repick<6>(ScanPickArgs);
template __global__ void k_scan_repick<8>(ScanPickArgs);

// k_repair_patch: a fixed grid walks the rows of this pass, 128 threads per row: the exact row (sum of the repair contraction's
// split-K partial maps, plane 0 first) replaces plane 0 of the map, the other planes of that row become zero; the group's
// test-and-set word is released.
__global__ __launch_bounds__(128) void k_repair_patch(RepairPatchArgs p)
{
    const int n_here = min(*p.n_list - p.list0, p.pass_rows / REPAIR_GROUP);
    const int n_rows = n_here * REPAIR_GROUP;
    const int planes_x = repair_ksplit_eff(p.ksplit, n_rows, p.Dp, p.items);
    const long long plane_x_stride = repair_plane_stride(n_rows, p.Dp);
    const int n4 = p.Dp >> 2;                                            // Dp is a multiple of 64
    for (int r = blockIdx.x; r < n_rows; r += gridDim.x) {
        const int g = r / REPAIR_GROUP, j = r - g * REPAIR_GROUP;
        const int e = p.list[p.list0 + g];
        if (j == 0 && threadIdx.x == 0) p.need[e] = 0;
        // lazy tails: a row of the previous call's last frames goes into the history's own map (one plane), where k_scan_repick walks it
        const bool hist = p.hist_C != nullptr && e >= p.hist_base;
        const int eu = hist ? e - p.hist_base : e, upa = hist ? HIST_UNITS : p.groups_per_array;
        const int a = eu / upa, f = (eu - a * upa) * REPAIR_GROUP + j;
        if (f >= (hist ? HIST_FRAMES : p.n_frames)) continue;
        float *drow = hist ? p.hist_C + ((long long)a * HIST_FRAMES + f) * p.Dp : p.C + ((long long)a * p.n_frames + f) * p.Dp;
        const int zero_planes = hist ? 1 : p.c_planes;
        for (int c4 = threadIdx.x; c4 < n4; c4 += 128) {
        const float *src = p.Cx + (long long)r * p.Dp + c4 * 4;
        float4 v = *reinterpret_cast<const float4 *>(src);
        int z = 1;
        for (; z + 3 < planes_x; z += 4) {                                // four partial maps in flight, summed in order
            const float4 w0 = *reinterpret_cast<const float4 *>(src + z * plane_x_stride);
            const float4 w1 = *reinterpret_cast<const float4 *>(src + (z + 1) * plane_x_stride);
            const float4 w2 = *reinterpret_cast<const float4 *>(src + (z + 2) * plane_x_stride);
            const float4 w3 = *reinterpret_cast<const float4 *>(src + (z + 3) * plane_x_stride);
            v.x = (((v.x + w0.x) + w1.x) + w2.x) + w3.x; v.y = (((v.y + w0.y) + w1.y) + w2.y) + w3.y;
            v.z = (((v.z + w0.z) + w1.z) + w2.z) + w3.z; v.w = (((v.w + w0.w) + w1.w) + w2.w) + w3.w;
        }
        for (; z < planes_x; ++z) {
            const float4 w = *reinterpret_cast<const float4 *>(src + z * plane_x_stride);
            v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w;
        }
        float *dst = drow + c4 * 4;
        *reinterpret_cast<float4 *>(dst) = v;
        for (int pl = 1; pl < zero_planes; ++pl) *reinterpret_cast<float4 *>(dst + pl * p.c_plane_stride) = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
}

// --------------------------------------------------------------------------------------
// Lazy tails, settling the debt (api.hip, settle_history): a consumer of the state that is not the lazy path itself -- an FP16X3 call of
// the same context, state_save, a recorded graph -- needs the EXACT energies the eager form would have left.  k_hist_list puts every
// history unit of the arrays on the repair list; the list-mode analysis, the repair contraction and k_repair_patch then recompute the
// previous call's last HIST_FRAMES rows into the history map; k_hist_settle walks them from the energies in front of them.
// --------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_hist_list(int *list, int *n_list, int *need, int n_units)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n_units) { list[i] = i; need[i] = 1; }
    if (i == 0) *n_list = n_units;
}
// grid (arrays), >= D threads: E = 0.8f E + 0.2f C over the HIST_FRAMES exact rows (:134-140), the state a call's repaired last frame leaves
__global__ __launch_bounds__(512) void k_hist_settle(const float *e_hist, const float *hist_C, float *state, int *n_list, int D, int Dp, float mu, float omu)
{
    const int a = blockIdx.x, d = threadIdx.x;
    if (a == 0 && d == 0) *n_list = 0;                                  // (no second pick ran to empty the list)
    if (d >= D) return;
    float E = e_hist[(long long)a * D + d];
    float h[HIST_FRAMES];
#pragma unroll
    for (int j = 0; j < HIST_FRAMES; ++j) h[j] = hist_C[((long long)a * HIST_FRAMES + j) * Dp + d];
#pragma unroll
    for (int j = 0; j < HIST_FRAMES; ++j) E = iir_step(mu, E, omu, h[j]);
    state[(long long)a * D + d] = E;
}

// --------------------------------------------------------------------------------------
// k_doa_fill -- gated-out frames keep the module's previous _currentDOA / _prob
// (BeamformingSeparationAndLocalisation.cpp:87: processFrame is simply not called)
// --------------------------------------------------------------------------------------
// grid (arrays), 1024 threads.  last[t] = index of the last voiced frame <= t (inclusive prefix max
// over the frames, Hillis-Steele in LDS over per-thread segments); unvoiced frames copy from it, or
// from the state carried over from the previous call when no frame has fired yet in this one.
__global__ __launch_bounds__(1024) void k_doa_fill(DoaFillArgs p)
{
    __shared__ int sLast[1024];
    const int a = blockIdx.x, tid = threadIdx.x, F = p.n_frames, S = p.S;
    const unsigned char *vc = p.voiced + (long long)a * F;
    const int per = (F + 1023) / 1024;
    const int t0 = tid * per, t1 = min(t0 + per, F);
    int last = -1;
    for (int t = t0; t < t1; ++t) if (vc[t]) last = t;
    sLast[tid] = last;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        const int other = tid >= off ? sLast[tid - off] : -1;
        __syncthreads();
        sLast[tid] = max(sLast[tid], other);
        __syncthreads();
    }
    int run = tid > 0 ? sLast[tid - 1] : -1;            // last voiced frame before this thread's segment
    const long long base = (long long)a * F * S;
    for (int t = t0; t < t1; ++t) {
        if (vc[t]) { run = t; continue; }
        for (int s = 0; s < S; ++s) {
            const long long o = base + (long long)t * S + s;
            if (run >= 0) {
                p.doa_bin[o] = p.doa_bin[base + (long long)run * S + s];
                if (p.doa_rad) p.doa_rad[o] = p.doa_rad[base + (long long)run * S + s];
                if (p.prob) p.prob[o] = p.prob[base + (long long)run * S + s];
            } else {
                p.doa_bin[o] = p.last_bin[a * S + s];
                if (p.doa_rad) p.doa_rad[o] = p.last_rad[a * S + s];
                if (p.prob) p.prob[o] = p.last_prob[a * S + s];
            }
        }
    }
    __syncthreads();
    // state for the next call = values of the last frame (written by the thread that owns it, after its own fill)
    if (F - 1 >= t0 && F - 1 < t1)
        for (int s = 0; s < S; ++s) {
            const long long o = base + (long long)(F - 1) * S + s;
            p.last_bin[a * S + s] = p.doa_bin[o];
            if (p.doa_rad) p.last_rad[a * S + s] = p.doa_rad[o];
            if (p.prob) p.last_prob[a * S + s] = p.prob[o];
        }
}

// --------------------------------------------------------------------------------------
// k_beamform_ola
// --------------------------------------------------------------------------------------
// grid (frame runs, arrays), 512 threads.  A run is FT frames plus the frame before it (whose
// second half is the overlap-add carry).  Frames are handled in batches of p.nb (<= BF_NB): per frame the 8
// waves transform the channels and thread k applies the delay-and-sum to bin k < 512 with the
// steering phasors factored as exp(j k s) = hi[k >> 5] * lo[k & 31] (49 sincos per channel and
// frame instead of 513; the reference regenerates the whole ramp per frame, Beamformer.cpp:59-60);
// the Nyquist bins of the batch are finished in one pass (lane = slot); then wave w inverse
// transforms the beamformed spectrum of batch slot w; then the 512 threads emit hop samples per
// frame with the carry in a register.  LDS per workgroup ~69 KB (M = 8, S = 1): two per CU.
template <int CPW, int OCC>
__global__ __launch_bounds__(512, OCC) void k_beamform_ola(BeamformArgs p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int M = p.M, S = p.S, a = blockIdx.y;
    float2 *xs = reinterpret_cast<float2 *>(smem_raw);                 // [M][FFT_SCRATCH] channel spectra
    const int NB = p.nb;
    float2 *ys = xs + M * FFT_SCRATCH;                                  // [NB*S][FFT_SCRATCH] beamformed slots
    float2 *steer = ys + NB * S * FFT_SCRATCH;                       // [S][M][49] steering phasors (lo 0..31, hi 32..48)
    float2 *tab = steer + S * M * 49;                                   // [TW_WORDS]
    float2 *xn = tab + TW_WORDS;                                        // [NB][M] Nyquist bins of the batch
    float2 *pn = xn + NB * M;                                           // [NB][S][M] their steering phasors
    double *cdoa = reinterpret_cast<double *>(pn + NB * S * M);      // [ft+1][S] cos(DOA + pi/2) of the run
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int t0 = blockIdx.x * p.ft;
    const int t1 = min(t0 + p.ft, p.n_frames);
    const int tfirst = t0 > 0 ? t0 - 1 : 0;

    fft_table_init(tab, p.window, tid, 512);
    for (int e = tid; e < (t1 - tfirst) * S; e += 512) {
        const double doa = (double)p.doa_rad[((long long)a * p.n_frames + tfirst) * S + e];
        cdoa[e] = cos(doa + 1.57079632679489661923);                    // cos(DOA + M_PI/2), Beamformer.cpp:59
    }
    __syncthreads();
    FftTw tw{tab};

    const float *base = p.pcm + (long long)a * p.array_stride;
    float2 raw[CPW][4], nxt[CPW][4];
#pragma unroll
    for (int cc = 0; cc < CPW; ++cc) {
        const int c = wave + 8 * cc;
        if (c < M) {
            const float2 *src = reinterpret_cast<const float2 *>(base + (long long)c * p.mic_stride + (long long)tfirst * FFT_H);
#pragma unroll
            for (int r = 0; r < 4; ++r) raw[cc][r] = src[lane + 64 * r];
#pragma unroll
            for (int r = 0; r < 4; ++r) nxt[cc][r] = src[lane + 64 * (r + 4)];
        }
    }
    float carry[MCA_MAX_SOURCES];
#pragma unroll
    for (int s = 0; s < MCA_MAX_SOURCES; ++s) carry[s] = 0.f;
    if (t0 == 0) {
#pragma unroll
        for (int s = 0; s < MCA_MAX_SOURCES; ++s)
            if (s < S) carry[s] = p.tail_in[((long long)a * S + s) * FFT_H + tid];
    }

    const double unit = (double)p.fs / (double)FFT_N / 346.1;          // Beamformer.cpp:59 without 2 pi
    const float inv = 1.0f / (float)M;
    const bool one_source = CPW == 1 && S == 1 && M <= 8;
    float2 phc[8];                                                      // one_source: this bin's steering phasor per channel
#pragma unroll
    for (int c = 0; c < 8; ++c) phc[c] = make_float2(1.f, 0.f);

    for (int tb = tfirst; tb < t1; tb += NB) {
        const int nb = min(NB, t1 - tb);
        for (int j = 0; j < nb; ++j) {
            const int t = tb + j;
            // the table only changes when a source moved to another steering angle: sources are slow
            // compared with the 10.7 ms hop, so most frames reuse the previous frame's phasors
            bool same = t > tfirst;
            for (int s = 0; s < S && same; ++s) same = cdoa[(t - tfirst) * S + s] == cdoa[(t - tfirst - 1) * S + s];
            if (!same) {
                for (int e = tid; e < S * M * 49; e += 512) {
                    const int s = e / (M * 49), rem = e - s * (M * 49), c = rem / 49, q = rem - c * 49;
                    const int kk = q < 32 ? q : (q - 32) * 32;
                    double turns = (double)kk * (unit * p.mic_x[c] * cdoa[(t - tfirst) * S + s]);
                    turns -= rint(turns);
                    float sn, cs;
                    sincospif(2.0f * (float)turns, &sn, &cs);
                    steer[e] = make_float2(cs, sn);
                }
            }
            // analysis of frame t
#pragma unroll
            for (int cc = 0; cc < CPW; ++cc) {
                const int c = wave + 8 * cc;
                if (c < M) {
                    float2 v[8];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float2 w0 = tw.win(r, lane), w1 = tw.win(r + 4, lane);
                        v[r] = make_float2(raw[cc][r].x * w0.x, raw[cc][r].y * w0.y);
                        v[r + 4] = make_float2(nxt[cc][r].x * w1.x, nxt[cc][r].y * w1.y);
                        raw[cc][r] = nxt[cc][r];
                    }
                    if (t + 1 < t1) {
                        const float2 *src = reinterpret_cast<const float2 *>(base + (long long)c * p.mic_stride + (long long)(t + 1) * FFT_H);
#pragma unroll
                        for (int r = 0; r < 4; ++r) nxt[cc][r] = src[lane + 64 * (r + 4)];
                    }
                    rfft1024(v, xs + c * FFT_SCRATCH, lane, tw);
                }
            }
            __syncthreads();
            // delay-and-sum: Y[k] = (1/M) sum_c X_c[k] exp(j k s_c), s_c = 2 pi unit x_c cos(DOA+pi/2)
            {
                const int k = tid;   // bins 0..511
                if (one_source) {
                    // one source, <= 8 channels: this thread's bin never changes, so its phasors stay in registers
                    // until the source moves (8 LDS reads per frame instead of 24).  M == 8 gets a branch-free copy:
                    // with per-channel guards every LDS read sits behind a scalar branch and its latency is exposed.
                    if (!same) {
#pragma unroll
                        for (int c = 0; c < 8; ++c)
                            if (c < M) phc[c] = cmul(steer[c * 49 + 32 + (k >> 5)], steer[c * 49 + (k & 31)]);
                    }
                    float2 acc = make_float2(0.f, 0.f);
                    if (M == 8) {
                        float2 x[8];
#pragma unroll
                        for (int c = 0; c < 8; ++c) x[c] = xs[c * FFT_SCRATCH + k];
#pragma unroll
                        for (int c = 0; c < 8; ++c) acc = cmac(acc, x[c], phc[c]);
                    } else {
#pragma unroll
                        for (int c = 0; c < 8; ++c)
                            if (c < M) acc = cmac(acc, xs[c * FFT_SCRATCH + k], phc[c]);
                    }
                    ys[j * FFT_SCRATCH + k] = make_float2(acc.x * inv, acc.y * inv);             // divC :70
                } else
                for (int s = 0; s < S; ++s) {
                    float2 acc = make_float2(0.f, 0.f);
                    for (int c = 0; c < M; ++c) {
                        const float2 *tb2 = steer + (s * M + c) * 49;
                        const float2 ph = cmul(tb2[32 + (k >> 5)], tb2[k & 31]);
                        acc = cmac(acc, xs[c * FFT_SCRATCH + k], ph);
                    }
                    ys[(j * S + s) * FFT_SCRATCH + k] = make_float2(acc.x * inv, acc.y * inv);   // divC :70
                }
                if (tid < M) xn[j * M + tid] = xs[tid * FFT_SCRATCH + FFT_H];
                if (tid < S * M) pn[j * S * M + tid] = steer[tid * 49 + 48];     // k = 512 = 32 * 16 + 0
            }
            __syncthreads();
        }
        // Nyquist bins of the batch: lane = slot
        if (tid < nb * S) {
            const int j = tid / S, s = tid - j * S;
            float2 acc = make_float2(0.f, 0.f);
            for (int c = 0; c < M; ++c) acc = cmac(acc, xn[j * M + c], pn[(j * S + s) * M + c]);
            ys[tid * FFT_SCRATCH + FFT_H] = make_float2(acc.x * inv, acc.y * inv);
        }
        __syncthreads();
        // synthesis: wave w inverse-transforms slots w, w+8, ...
        for (int q = wave; q < nb * S; q += 8) irfft1024(ys + q * FFT_SCRATCH, lane, tw);
        __syncthreads();
        // overlap-add, hop samples per frame
        for (int j = 0; j < nb; ++j) {
            const int t = tb + j;
#pragma unroll
            for (int s = 0; s < MCA_MAX_SOURCES; ++s) {
                if (s < S) {
                    const float *y = reinterpret_cast<const float *>(ys + (j * S + s) * FFT_SCRATCH);
                    if (t >= t0) p.out[((long long)a * S + s) * (long long)p.n_frames * FFT_H + (long long)t * FFT_H + tid] = carry[s] + y[tid];
                    carry[s] = y[tid + FFT_H];
                }
            }
        }
        __syncthreads();
    }
    if (t1 == p.n_frames) {
#pragma unroll
        for (int s = 0; s < MCA_MAX_SOURCES; ++s)
            if (s < S) p.tail_out[((long long)a * S + s) * FFT_H + tid] = carry[s];
    }
}

template __global__ void k_beamform_ola<1, 2>(BeamformArgs);
template __global__ void k_beamform_ola<1, 4>(BeamformArgs);
template __global__ void k_beamform_ola<2, 2>(BeamformArgs);

// --------------------------------------------------------------------------------------
// 512-sample frames (16 kHz at the reference's 0.025 s frame rate) on the wave-level FFT
// --------------------------------------------------------------------------------------
// A 512-point complex transform takes TWO real 512-sample sequences at once: z = a + j b,
// A[k] = (Z[k] + conj Z[512-k]) / 2, B[k] = -j (Z[k] - conj Z[512-k]) / 2, k = 0..256.  So the transform of
// fft512.h serves two channels (analysis) or two frames (synthesis) per wave pass, and a workgroup of 8 waves
// analyses two frames of 8 channels at a time.  Spectra rows have N512_ROW float2 words.
// (rfft512_pair, irfft512_pair, load_pair_512 and the N512_* constants: fft512.h)

// k_stft_phat_512: grid (ceil(frames / fpb), arrays), 512 threads, M <= 8.  Wave w analyses channels (2 (w & 3), + 1) of
// frame slot w >> 2; then thread (slot = tid >> 8, bin = tid & 255) whitens and forms the pair products.
// LDS: [2][8][N512_ROW] spectra + 8 wave scratches + twiddles + [fpb][M] Nyquist bins + [fpb] powers.
template <int MT, bool ULA, typename OutT>
__global__ __launch_bounds__(512) void k_stft_phat_512(StftPhatArgs p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float2 *spec = reinterpret_cast<float2 *>(smem_raw);                   // [2][8][N512_ROW]
    float2 *scr = spec + 2 * 8 * N512_ROW;                                  // [8][FFT_SCRATCH]
    float2 *tab = scr + 8 * FFT_SCRATCH;                                    // [TW_WIN]
    float2 *nyq = tab + TW_WIN;                                             // [fpb][M]
    const int M = MT > 0 ? MT : p.M;
    float *spow = reinterpret_cast<float *>(nyq + p.fpb * M);               // [fpb][8] per wave (4 waves per frame: the other 4 slots stay zero)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    fft_table_init(tab, nullptr, tid, 512);
    for (int e = tid; e < p.fpb * 8; e += 512) spow[e] = 0.f;
    float wreg[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) wreg[r] = 0.5f * p.window[lane + 64 * r];
    FftTw tw{tab};
    const int slot = wave >> 2, c0 = 2 * (wave & 3);
    // list mode (the repair pass of the adaptive SRP precision, as in k_stft_phat): the workgroups walk the listed units of REPAIR_GROUP
    // frames; unit number li - list0 of the pass writes the A rows (li - list0) * REPAIR_GROUP ...  Otherwise one run of fpb frames.
    const int li_end = p.list ? min(*p.n_list, p.list0 + p.list_cap) : 1, li_step = p.list ? (int)gridDim.x : 1;
    for (int li = p.list ? p.list0 + (int)blockIdx.x : 0; li < li_end; li += li_step) {
    int a = blockIdx.y;
    int f_begin = blockIdx.x * p.fpb;
    long long row_base = (long long)a * p.n_frames;                         // A row of frame f = row_base + f
    if (p.list) {
        const int e = p.list[li];
        a = e / p.groups_per_array;
        f_begin = (e - a * p.groups_per_array) * REPAIR_GROUP;
        row_base = (long long)(li - p.list0) * REPAIR_GROUP - f_begin;
    }
    const int f_end = min(f_begin + p.fpb, p.n_frames);
    __syncthreads();
    const float *base = p.pcm + (long long)a * p.array_stride;

    for (int f = f_begin; f < f_end; f += 2) {
        const int nfr = min(2, f_end - f);
        if (slot < nfr && c0 < M) {
            float2 v[8];
            load_pair_512(v, base, p.mic_stride, c0, M, (long long)(p.frame0 + f + slot) * N512_H, wreg, lane);
            const PairBalance pb = pair_balance_512(v);        // (PHAT keeps the phase only: a channel far below its transform partner, pair_balance.h)
            rfft512_pair(v, scr + wave * FFT_SCRATCH, spec + (slot * 8 + c0) * N512_ROW, spec + (slot * 8 + c0 + 1) * N512_ROW, lane, tw);
            pair_restore_512(pb, spec + (slot * 8 + c0) * N512_ROW, spec + (slot * 8 + c0 + 1) * N512_ROW, lane);
        }
        __syncthreads();
        const int fi = tid >> 8, k = tid & 255;
        if (fi < nfr) {
            float2 *xs = spec + fi * 8 * N512_ROW;
            OutT *arow = reinterpret_cast<OutT *>(p.A) + (row_base + f + fi) * (long long)p.a_row_elems;
            if (k < M) nyq[(f + fi - f_begin) * M + k] = whiten(xs[k * N512_ROW + N512_H]);
            if (p.power) {
                // dsp::SignalPower::FFTPower [INFERRED, SURVEY A.8]: (1/N^2) sum_k w_k |X[k]|^2, w = 2 except DC and Nyquist
                float acc = 0.f;
                for (int m = 0; m < M; ++m) { const float2 z = xs[m * N512_ROW + k]; acc += z.x * z.x + z.y * z.y; }
                acc *= k == 0 ? 1.f : 2.f;
                if (k < M) { const float2 z = xs[k * N512_ROW + N512_H]; acc += z.x * z.x + z.y * z.y; }
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off);
                if (lane == 0) spow[(f + fi - f_begin) * 8 + (wave & 3)] = acc;
            }
            if constexpr (MT == 0)
                for (int m = 0; m < M; ++m) xs[m * N512_ROW + k] = whiten(xs[m * N512_ROW + k]);
            pair_stage<MT, ULA, true, OutT>(xs + k, N512_ROW, M, arow, p, k, N512_K);
        }
        __syncthreads();
    }
    if (p.power && tid < f_end - f_begin)
        p.power[(long long)a * p.total_frames + p.frame0 + f_begin + tid] = sum8(spow + tid * 8) / (512.f * 512.f) / (float)M;
    if (tid < f_end - f_begin) {
        OutT *arow = reinterpret_cast<OutT *>(p.A) + (row_base + f_begin + tid) * (long long)p.a_row_elems;
        pair_stage<MT, ULA, false, OutT>(nyq + tid * M, 1, M, arow, p, N512_H, N512_K);
    }
    }                                                                       // (list mode: the barrier at the top of the next unit keeps nyq until every thread is through)
}

#define INST_512(MT, ULA, T) template __global__ void k_stft_phat_512<MT, ULA, T>(StftPhatArgs);
INST_512(0, false, float) INST_512(0, true, float) INST_512(4, false, float) INST_512(4, true, float) INST_512(8, false, float) INST_512(8, true, float)
INST_512(0, false, _Float16) INST_512(0, true, _Float16) INST_512(4, false, _Float16) INST_512(4, true, _Float16) INST_512(8, false, _Float16) INST_512(8, true, _Float16)

// k_stft_phat_sub2<R>: the analysis + PHAT stage for TWO microphones at frames of N = 512 R samples, R = 4 or 8 --
// FreqGCCBinauralLocalisation's 0.075 s frames (BinauralLocalisation.h:196) at 32 kHz (2048) and 44.1 / 48 kHz (4096; the
// configuration of the reference's own test, test_mcarray.cpp:283).  A frame is R 512-sample real sub-sequences x[R n + r]
// per channel: wave w = (frame slot, channel, pair of sub-sequences) transforms two of them in one 512-point complex
// transform (rfft512_pair), one radix-R butterfly per m = k mod 512 recombines them (fft512.h), thread = bin forms the PHAT
// cross-spectrum of the one pair.  8 / R frames per pass.  grid (ceil(frames / fpb), arrays), 512 threads;
// LDS: 16 sub-spectra + 8 wave scratches (reused for the spectra) + twiddles.
template <int R, typename OutT>
__global__ __launch_bounds__(512) void k_stft_phat_sub2(StftPhatArgs p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    constexpr int N = 512 * R, H = N / 2, K = H + 1, XR = H + 2, FPP = 8 / R;      // FPP: frames per pass
    float2 *sub = reinterpret_cast<float2 *>(smem_raw);                    // [FPP][2][R][N512_ROW]
    float2 *scr = sub + 16 * N512_ROW;                                      // [8][FFT_SCRATCH]
    float2 *X = scr;                                                        // [FPP][2][XR] (after the transforms)
    float2 *tab = scr + 8 * FFT_SCRATCH;                                    // [TW_WIN]
    float *spow = reinterpret_cast<float *>(tab + TW_WIN);                  // [FPP][8] per wave
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int a = blockIdx.y;
    const int f_begin = blockIdx.x * p.fpb, f_end = min(f_begin + p.fpb, p.n_frames);
    fft_table_init(tab, nullptr, tid, 512);
    // wave -> (frame slot, channel, pair): samples x[R n + 2 pr], x[R n + 2 pr + 1] are one float2
    const int slot = wave / R, cw = (wave % R) / (R / 2), pr = wave % (R / 2);
    float2 wreg[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const float2 w = reinterpret_cast<const float2 *>(p.window)[(R / 2) * (lane + 64 * r) + pr];
        wreg[r] = make_float2(0.5f * w.x, 0.5f * w.y);
    }
    __syncthreads();
    FftTw tw{tab};
    const float *base = p.pcm + (long long)a * p.array_stride + (long long)cw * p.mic_stride;
    for (int f = f_begin; f < f_end; f += FPP) {
        const int nfr = min(FPP, f_end - f);
        if (slot < nfr) {
            const float2 *src = reinterpret_cast<const float2 *>(base + (long long)(p.frame0 + f + slot) * H);
            float2 v[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const float2 x = src[(R / 2) * (lane + 64 * r) + pr], w = wreg[r];
                v[r] = make_float2(x.x * w.x, x.y * w.y);
            }
            float2 *S = sub + ((slot * 2 + cw) * R + 2 * pr) * N512_ROW;
            rfft512_pair(v, scr + wave * FFT_SCRATCH, S, S + N512_ROW, lane, tw);
        }
        __syncthreads();
        for (int e = tid; e < nfr * 2 * 512; e += 512) {                    // thread = (frame, channel, m): one radix-R butterfly
            const int jc = e >> 9, m = e & 511;
            if constexpr (R == 8) combine4096_m(sub + jc * R * N512_ROW, m, p.tw, X + jc * XR);
            else combine2048_m(sub + jc * R * N512_ROW, m, p.tw, X + jc * XR);
        }
        __syncthreads();
        for (int j = 0; j < nfr; ++j) {
            OutT *arow = reinterpret_cast<OutT *>(p.A) + ((long long)a * p.n_frames + f + j) * (long long)p.a_row_elems;
            const float2 *X0 = X + (j * 2) * XR, *X1 = X0 + XR;
            float acc = 0.f;
            for (int k = tid; k < K; k += 512) {
                const float2 x0 = X0[k], x1 = X1[k];
                if (p.power) {
                    // dsp::SignalPower::FFTPower [INFERRED, SURVEY A.8]: (1/N^2) sum_k w_k |X[k]|^2, w = 2 except DC and Nyquist
                    const float pw = x0.x * x0.x + x0.y * x0.y + x1.x * x1.x + x1.y * x1.y;
                    acc += (k == 0 || k == H) ? pw : 2.f * pw;
                }
                store_a(arow, p, k, cmulc(whiten(x0), whiten(x1)));
            }
            if (p.power) {
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off);
                if (lane == 0) spow[j * 8 + wave] = acc;
            }
        }
        if (p.power) {
            __syncthreads();
            if (tid < nfr) p.power[(long long)a * p.total_frames + p.frame0 + f + tid] = sum8(spow + tid * 8) / ((float)N * (float)N) / 2.f;
        }
        __syncthreads();
    }
}

template __global__ void k_stft_phat_sub2<8, float>(StftPhatArgs);
template __global__ void k_stft_phat_sub2<8, _Float16>(StftPhatArgs);
template __global__ void k_stft_phat_sub2<4, float>(StftPhatArgs);
template __global__ void k_stft_phat_sub2<4, _Float16>(StftPhatArgs);

// k_mvdr_analyse_512: the analysis stage of the MVDR path (kernels_mvdr.hip: k_mvdr_analyse_1024) for 512-sample frames: wave w
// transforms channels 2 w, 2 w + 1 of the frame in one pass (up to 16 channels), then the spectra go out transposed,
// [bin][mic].  grid (ceil(frames / fpb), streams), 512 threads; LDS = 16 spectra + 8 wave scratches + twiddles.
__global__ __launch_bounds__(512) void k_mvdr_analyse_512(MvdrAnalyseArgs p, int fpb)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float2 *spec = reinterpret_cast<float2 *>(smem_raw);                   // [16][N512_ROW]
    float2 *scr = spec + 16 * N512_ROW;                                     // [8][FFT_SCRATCH]
    float2 *tab = scr + 8 * FFT_SCRATCH;                                    // [TW_WIN]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int a = blockIdx.y, M = p.M;
    const int f_begin = blockIdx.x * fpb, f_end = min(f_begin + fpb, p.n_frames);
    fft_table_init(tab, nullptr, tid, 512);
    float wreg[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) wreg[r] = 0.5f * p.window[lane + 64 * r];
    // factored steering phasors of the block's frames (MvdrAnalyseArgs::T)
    {
        const int nhi = (512 >> 6) + 1, nph = nhi + 32;
        for (int e = tid; e < (f_end - f_begin) * M * nph; e += 512) {
            const int f = f_begin + e / (M * nph), rem = e % (M * nph), m = rem / nph, i = rem - m * nph;
            const long long o = (long long)a * p.n_frames + f;
            const double cd = cos((double)p.doa_rad[o] + 1.57079632679489661923);   // cos(DOA + M_PI/2), Beamformer.cpp:59
            const int kk = i < nhi ? (i << 5) : i - nhi;
            double turns = (double)kk * (p.unit * p.mic_x[m] * cd);
            turns -= rint(turns);
            float sn, cs;
            sincospif(2.0f * (float)turns, &sn, &cs);
            p.T[(o * M + m) * nph + i] = make_float2(cs, -sn);
        }
    }
    __syncthreads();
    FftTw tw{tab};
    const float *base = p.pcm + (long long)a * p.stream_stride;
    for (int f = f_begin; f < f_end; ++f) {
        if (2 * wave < M) {
            float2 v[8];
            load_pair_512(v, base, p.mic_stride, 2 * wave, M, (long long)f * N512_H, wreg, lane);
            rfft512_pair(v, scr + wave * FFT_SCRATCH, spec + (2 * wave) * N512_ROW, spec + (2 * wave + 1) * N512_ROW, lane, tw);
        }
        __syncthreads();
        float2 *xo = p.X + ((long long)a * p.n_frames + f) * (long long)N512_K * M;
        for (int e = tid; e < N512_K * M; e += 512) {
            const int k = e / M, m = e - k * M;
            xo[e] = spec[m * N512_ROW + k];
        }
        __syncthreads();
    }
}

// k_beamform_512: grid (runs of ft frames, arrays), 512 threads, M <= 8.  Frames are taken two at a time: the 8 waves analyse
// them as in k_stft_phat_512 and thread (slot, bin) applies the delay-and-sum with phasors factored hi[k >> 5] * lo[k & 31]
// (Beamformer.cpp:51-71; a slot's table is rebuilt only when its DOA differs from the one it was built for).  Then wave s
// inverse-transforms the two beamformed frames of source s in ONE complex transform, and threads 0..255 overlap-add the
// frames in order with the carry in a register.  A run starts one frame early (carry).
__global__ __launch_bounds__(512) void k_beamform_512(BeamformArgs p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int M = p.M, S = p.S, a = blockIdx.y;
    constexpr int NPH = 41;                                                 // 9 hi (k >> 5 = 0..8) + 32 lo phasors
    // frame pairs per synthesis round.  One: the beamformed spectra sit in scratches that the next pair's analysis reuses, and
    // with two workgroups per CU the other workgroup fills the SIMDs while S waves run the inverse transforms.
    constexpr int NPB = 1;
    float2 *spec = reinterpret_cast<float2 *>(smem_raw);                   // [2][8][N512_ROW]
    float2 *scr = spec + 2 * 8 * N512_ROW;                                  // [8][FFT_SCRATCH]
    float2 *tab = scr + 8 * FFT_SCRATCH;                                    // [TW_WIN]
    float2 *steer = tab + TW_WIN;                                           // [2][S][M][NPH]
    double *built = reinterpret_cast<double *>(steer + 2 * S * M * NPH);    // [2][S] cos(DOA + pi/2) each slot's table was built for
    // 80 KiB with one source and 8 channels (two workgroups per CU): the beamformed spectra live in the scratches of
    // waves 4..7 (written after the analysis; the inverse transforms use those of waves 0..3) and the time-domain
    // frames take the place of the channel spectra, which are dead by then
    float2 *ys = scr + 4 * FFT_SCRATCH;                                     // [NPB][S][2][N512_ROW] beamformed spectra
    float *yt = reinterpret_cast<float *>(spec);                            // [NPB][S][2][512] time-domain frames
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int t0 = blockIdx.x * p.ft, t1 = min(t0 + p.ft, p.n_frames);
    const int tfirst = t0 > 0 ? t0 - 1 : 0;
    fft_table_init(tab, nullptr, tid, 512);
    if (tid < 2 * S) built[tid] = 2.0;                                      // no cosine: every table is built on first use
    float wreg[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) wreg[r] = 0.5f * p.window[lane + 64 * r];
    float carry[MCA_MAX_SOURCES];
#pragma unroll
    for (int s = 0; s < MCA_MAX_SOURCES; ++s) carry[s] = (s < S && t0 == 0 && tid < N512_H) ? p.tail_in[((long long)a * S + s) * N512_H + tid] : 0.f;
    __syncthreads();
    FftTw tw{tab};
    const float *base = p.pcm + (long long)a * p.array_stride;
    const double unit = (double)p.fs / 512.0 / 346.1;                       // Beamformer.cpp:59 without 2 pi
    const float inv = 1.0f / (float)M;
    const int slot = wave >> 2, c0 = 2 * (wave & 3);

    for (int tb = tfirst; tb < t1; tb += 2 * NPB) {
        const int npair = min(NPB, (t1 - tb + 1) / 2);
        for (int pr = 0; pr < npair; ++pr) {
            const int t = tb + 2 * pr, nfr = min(2, t1 - t);
            // steering phasors of the two frames (slots), rebuilt when the source moved
            for (int fs_ = 0; fs_ < nfr * S; ++fs_) {
                const int fi = fs_ / S, s_ = fs_ - fi * S;
                const double cd = cos((double)p.doa_rad[((long long)a * p.n_frames + t + fi) * S + s_] + 1.57079632679489661923);
                if (cd != built[fi * S + s_]) {                             // uniform over the workgroup
                    for (int e = tid; e < M * NPH; e += 512) {
                        const int c = e / NPH, q = e - c * NPH;
                        const int kk = q < 9 ? (q << 5) : q - 9;
                        double turns = (double)kk * (unit * p.mic_x[c] * cd);
                        turns -= rint(turns);
                        float sn, cs;
                        sincospif(2.0f * (float)turns, &sn, &cs);
                        steer[(fi * S + s_) * M * NPH + e] = make_float2(cs, sn);
                    }
                    __syncthreads();
                    if (tid == 0) built[fi * S + s_] = cd;
                }
            }
            if (slot < nfr && c0 < M) {
                float2 v[8];
                load_pair_512(v, base, p.mic_stride, c0, M, (long long)(t + slot) * N512_H, wreg, lane);
                rfft512_pair(v, scr + wave * FFT_SCRATCH, spec + (slot * 8 + c0) * N512_ROW, spec + (slot * 8 + c0 + 1) * N512_ROW, lane, tw);
            }
            __syncthreads();
            // delay-and-sum: Y[k] = (1/M) sum_c X_c[k] exp(j k s_c)
            {
                const int fi = tid >> 8, k = tid & 255;
                const float2 *xs = spec + fi * 8 * N512_ROW;
                for (int s = 0; s < S; ++s) {
                    float2 *yo = ys + ((pr * S + s) * 2 + fi) * N512_ROW;
                    if (fi < nfr) {
                        const float2 *st = steer + (fi * S + s) * M * NPH;
                        float2 acc = make_float2(0.f, 0.f), accn = make_float2(0.f, 0.f);
                        for (int c = 0; c < M; ++c) {
                            acc = cmac(acc, xs[c * N512_ROW + k], cmul(st[c * NPH + (k >> 5)], st[c * NPH + 9 + (k & 31)]));
                            if (k == 0) accn = cmac(accn, xs[c * N512_ROW + N512_H], st[c * NPH + 8]);  // k = 256 = 32 * 8 + 0
                        }
                        yo[k] = make_float2(acc.x * inv, acc.y * inv);                                  // divC :70
                        if (k == 0) yo[N512_H] = make_float2(accn.x * inv, accn.y * inv);
                    } else {                                                                            // no second frame: zeros
                        yo[k] = make_float2(0.f, 0.f);
                        if (k == 0) yo[N512_H] = make_float2(0.f, 0.f);
                    }
                }
            }
            __syncthreads();
        }
        // synthesis: one complex inverse transform per (pair, source) gives both frames of the pair
        if (wave < npair * S) {
            float2 v[8];
            irfft512_pair(ys + (wave * 2) * N512_ROW, ys + (wave * 2 + 1) * N512_ROW, scr + wave * FFT_SCRATCH, v, lane, tw);
            float *ya = yt + (wave * 2) * 512, *yb = ya + 512;
#pragma unroll
            for (int i = 0; i < 8; ++i) { const int n = lane + 64 * br3(i); ya[n] = v[i].x; yb[n] = v[i].y; }
        }
        __syncthreads();
        // overlap-add, hop samples per frame, frames in order
        if (tid < N512_H) {
            for (int pr = 0; pr < npair; ++pr)
                for (int fi = 0; fi < 2; ++fi) {
                    const int tt = tb + 2 * pr + fi;
                    if (tt >= t1) break;
#pragma unroll
                    for (int s = 0; s < MCA_MAX_SOURCES; ++s)
                        if (s < S) {
                            const float *y = yt + ((pr * S + s) * 2 + fi) * 512;
                            if (tt >= t0) p.out[((long long)a * S + s) * (long long)p.n_frames * N512_H + (long long)tt * N512_H + tid] = carry[s] + y[tid];
                            carry[s] = y[tid + N512_H];
                        }
                }
        }
        __syncthreads();
    }
    if (t1 == p.n_frames && tid < N512_H) {
#pragma unroll
        for (int s = 0; s < MCA_MAX_SOURCES; ++s)
            if (s < S) p.tail_out[((long long)a * S + s) * N512_H + tid] = carry[s];
    }
}

// --------------------------------------------------------------------------------------
// k_gcc2_scan -- FreqGCCBinauralLocalisation, deterministic part (BinauralLocalisation.cpp:438-523)
// --------------------------------------------------------------------------------------
// grid (chunks, arrays), thread d = steering delay.  Phase 1: corr_t = (1-mu) R_t + mu corr_{t-1}
// (:445-448; mu = 0 on the stream's very first frame, 0.8f afterwards, :323,:523) with the same
// 128-frame warm-up as k_scan_pick; the smoothed correlation of the last GCC2_DOAWARM warm-up
// frames and of the chunk is kept in LDS.  Phase 2: one wave per frame: first-max argmax (:502),
// min and sum (setProbability :584-588).  Phase 3: one thread runs the scalar DOA recursion
// DOA = m DOA + (1-m) angle (:504, m = 0.6f; 0.6^64 = 6e-15 so 64 warm-up frames suffice) and the
// interpolated probability of the PREVIOUS DOA (:454, :590-630) in frame order.
__global__ __launch_bounds__(256) void k_gcc2_scan(Gcc2ScanArgs p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int D = p.D, Dl = (D + 3) / 4 * 4 + 4;                        // LDS row stride (the rows are only read along d)
    float *sC = reinterpret_cast<float *>(smem_raw);                    // [GCC2_DOAWARM + chunk][Dl]
    const int nslot = GCC2_DOAWARM + p.chunk;
    int *sIdx = reinterpret_cast<int *>(sC + nslot * Dl);               // [nslot]
    float *sMin = reinterpret_cast<float *>(sIdx + nslot);              // [nslot]
    float *sSum = sMin + nslot;                                         // [nslot]
    const int d = threadIdx.x, lane = d & 63, wave = d >> 6, nwaves = blockDim.x >> 6;
    const int a = blockIdx.y;
    // with the power gate the recursions only see the frames that passed it (:434): the scan runs over that list
    const int nf = p.nv ? p.nv[a] : p.n_frames;
    const long long done = p.vdone_in[a];
    const int *vi = p.vidx ? p.vidx + (long long)a * p.n_frames : nullptr;
    const unsigned char *vr = p.vreset ? p.vreset + (long long)a * p.n_frames : nullptr;
    if (nf == 0) {                                                      // nothing fired: the state carries over unchanged
        if (blockIdx.x == 0) {
            if (d < D) p.corr_out[(long long)a * D + d] = p.corr_in[(long long)a * D + d];
            if (d == 0) { p.doa_out[a] = p.doa_in[a]; p.vdone_out[a] = done; }
        }
        return;
    }
    const int t_start = blockIdx.x * p.chunk, t_end = min(t_start + p.chunk, nf);
    if (t_start >= nf) return;
    const int keep_start = max(0, t_start - GCC2_DOAWARM);             // first frame whose corr is kept
    const int warm_start = max(0, keep_start - SCAN_WARM);
    const float *C = p.C + (long long)a * p.n_frames * p.Dp;
    if (d < D) {
        float c = warm_start == 0 ? p.corr_in[(long long)a * D + d] : 0.f;
        for (int t0 = warm_start; t0 < t_end; t0 += 8) {                  // 8 independent loads in flight, then the serial recursion
            float r8[8];
            csum_rows(r8, C, [&](int i) { const int j = min(t0 + i, t_end - 1); return (long long)(vi ? vi[j] : j) * p.Dp + d; }, p.c_planes, p.c_plane_stride);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int t = t0 + i;
                if (t < t_end) {
                    // _corrMemoryFactor = 0: on the stream's first frame (:323), or -- with the gate -- on the first
                    // frame that fires after more than windowsToDecay gated-out frames (:530-560)
                    const bool first = vr ? vr[t] != 0 : (done + t) == 0;
                    c = first ? r8[i] : (p.one_minus_mu * r8[i] + p.mu * c);   // :445-447
                    if (t >= keep_start) sC[(t - keep_start) * Dl + d] = c;
                    if (t >= t_start && p.corr) p.corr[((long long)a * p.n_frames + (vi ? vi[t] : t)) * D + d] = c;
                }
            }
        }
        if (t_end == nf) p.corr_out[(long long)a * D + d] = c;          // _prevCorrelationsReal :448
    }
    __syncthreads();
    for (int tl = wave; tl < t_end - keep_start; tl += nwaves) {
        const float *cr = sC + tl * Dl;
        float bv = -INFINITY, mn = INFINITY, sm = 0.f; int bi = 0x7fffffff;
        for (int dd = lane; dd < D; dd += 64) {
            const float v = cr[dd];
            if (v > bv) { bv = v; bi = dd; }
            mn = fminf(mn, v); sm += v;
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const float ov = __shfl_xor(bv, off); const int oi = __shfl_xor(bi, off);
            if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
            mn = fminf(mn, __shfl_xor(mn, off));
            sm += __shfl_xor(sm, off);
        }
        if (lane == 0) { sIdx[tl] = bi; sMin[tl] = mn; sSum[tl] = sm; }
    }
    __syncthreads();
    // Phase 3a: the angles of the argmaxes (all threads), then ONE thread runs the scalar recursion DOA = m DOA + (1-m) angle
    // over the kept frames (:504) and leaves the DOA BEFORE every frame in LDS; phase 3b: thread = frame evaluates
    // setProbability of that previous DOA (:454, :590-630) and writes the outputs.  (The serial thread used to do all of it:
    // 192 frames x ~800 cycles of dependent LDS reads and divisions per workgroup.)
    float *sAng = sSum + nslot;                                        // [nslot] doaIdx2angle(argmax)
    float *sPrev = sAng + nslot;                                       // [nslot + 1] DOA before the frame (entry tl + 1: after it)
    const int nkept = t_end - keep_start;
    for (int tl = d; tl < nkept; tl += blockDim.x) sAng[tl] = p.grid[sIdx[tl]];       // :503
    __syncthreads();
    if (d == 0) {
        float doa = keep_start == 0 ? p.doa_in[a] : 0.f;
        for (int tl = 0; tl < nkept; ++tl) {
            sPrev[tl] = doa;
            const bool first = vr ? vr[keep_start + tl] != 0 : (done + keep_start + tl) == 0;   // _doaMemoryFactor = 0, as above
            doa = first ? sAng[tl] : (p.doa_mem * doa + p.one_minus_doa_mem * sAng[tl]);   // :504
        }
        sPrev[nkept] = doa;
        if (t_end == nf) { p.doa_out[a] = doa; p.vdone_out[a] = done + nf; }
    }
    __syncthreads();
    const float halfpi = 1.57079632679489661923f;
    for (int t = t_start + d; t < t_end; t += blockDim.x) {
        const int tl = t - keep_start;
        const float *cr = sC + tl * Dl;
        const float doa = sPrev[tl];
        // setProbability(_currentDOA, _prob, 1) with the DOA of the previous frame (:454)
        const float mn = sMin[tl];
        const float sum = sSum[tl] - mn * (float)D;                     // :588
        float ang = fminf(fmaxf(doa, -halfpi), halfpi);                 // angle2DOAidx :110-115
        int idx = (int)((ang + halfpi) / p.step);
        idx = min(max(idx, 0), D - 1);
        const float angle = p.grid[idx];
        float pr;
        if (0 < idx && idx < D - 1) {
            float pc, nc, pd, nd;
            if (angle > doa) { pc = cr[idx - 1]; pd = p.grid[idx - 1]; nc = cr[idx]; nd = angle; }
            else { pc = cr[idx]; pd = angle; nc = cr[idx + 1]; nd = p.grid[idx + 1]; }
            pr = (nc - pc) / (nd - pd) * (doa - pd) + pc;
        } else pr = cr[idx];
        float pb = sum > 0.f ? (pr - mn) / sum : 0.f;
        pb = pb < 0.01f ? 0.f : pb;
        const float doa_after = sPrev[tl + 1];
        const long long o = (long long)a * p.n_frames + (vi ? vi[t] : t);
        if (p.prob) p.prob[o] = pb;
        p.argmax[o] = sIdx[tl];
        if (p.doa_rad) p.doa_rad[o] = doa_after;
    }
}

// --------------------------------------------------------------------------------------
// k_gcc2_compact -- the frames of every array that passed the gate, in order (grid (arrays), 256 threads), and for each
// of them whether the memory factors are zero when it fires -- the silence rule of BinauralLocalisation.cpp:530-560:
// every gated-out frame after the floor estimate exists sets the factors to their maxima while _silenceFramesCounter <
// windowsToDecay and to zero from then on, then increments the counter (:536-559); a frame that fires resets it (:525).
// So a frame that fires after a run of r such frames restarts the recursions iff r >= windowsToDecay + 1.  The run
// before the call's first fired frame continues the counter carried in `silence` (per array, updated here).
// --------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_gcc2_compact(const unsigned char *voiced, int n_frames, int *vidx, int *nv, const int *post0,
                                                      int *silence, int windows_to_decay, unsigned char *vreset)
{
    __shared__ int s_w[4], s_last[4];
    const int a = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const unsigned char *vc = voiced + (long long)a * n_frames;
    int *out = vidx + (long long)a * n_frames;
    unsigned char *rs = vreset + (long long)a * n_frames;
    const int sil_in = silence[a], p0 = post0[a];
    int base = 0, lastf = -1;                       // fired frames so far, the last of them
    for (int t0 = 0; t0 < n_frames; t0 += 256) {
        const int t = t0 + tid;
        const bool v = t < n_frames && vc[t] != 0;
        const unsigned long long m = __ballot(v);
        if (lane == 0) { s_w[wave] = __popcll(m); s_last[wave] = m ? t0 + wave * 64 + 63 - __clzll(m) : -1; }
        __syncthreads();
        int off = base;
        for (int w = 0; w < wave; ++w) off += s_w[w];
        if (v) {
            const unsigned long long below = m & ((1ull << lane) - 1ull);
            int prev = lastf;
            for (int w = 0; w < wave; ++w) if (s_last[w] >= 0) prev = s_last[w];
            if (below) prev = t0 + wave * 64 + 63 - __clzll(below);
            const int run = prev >= 0 ? t - prev - 1 : sil_in + (t - p0);
            const int o = off + __popcll(below);
            out[o] = t;
            rs[o] = run >= windows_to_decay + 1 ? 1 : 0;
        }
        base += s_w[0] + s_w[1] + s_w[2] + s_w[3];
        for (int w = 0; w < 4; ++w) if (s_last[w] >= 0) lastf = s_last[w];
        __syncthreads();
    }
    if (tid == 0) {
        nv[a] = base;
        silence[a] = lastf >= 0 ? n_frames - 1 - lastf : sil_in + (n_frames - p0);   // _silenceFramesCounter at the end of the call
    }
}

// --------------------------------------------------------------------------------------
// k_gcc2_fill -- gated-out frames keep _currentDOA / _prob / the smoothed correlation of the last frame that fired
// (BinauralLocalisation.cpp:434: the block is skipped).  grid (arrays), 1024 threads; prefix max as k_doa_fill.
// --------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void k_gcc2_fill(Gcc2FillArgs p)
{
    __shared__ int sLast[1024];
    const int a = blockIdx.x, tid = threadIdx.x, F = p.n_frames, D = p.D;
    const unsigned char *vc = p.voiced + (long long)a * F;
    const int per = (F + 1023) / 1024;
    const int t0 = tid * per, t1 = min(t0 + per, F);
    int last = -1;
    for (int t = t0; t < t1; ++t) if (vc[t]) last = t;
    sLast[tid] = last;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        const int other = tid >= off ? sLast[tid - off] : -1;
        __syncthreads();
        sLast[tid] = max(sLast[tid], other);
        __syncthreads();
    }
    int run = tid > 0 ? sLast[tid - 1] : -1;
    const long long base = (long long)a * F;
    for (int t = t0; t < t1; ++t) {
        if (vc[t]) { run = t; continue; }
        p.argmax[base + t] = run >= 0 ? p.argmax[base + run] : p.last_idx[a];
        p.doa_rad[base + t] = run >= 0 ? p.doa_rad[base + run] : p.last_rad[a];
        p.prob[base + t] = run >= 0 ? p.prob[base + run] : p.last_prob[a];
        if (p.corr) {
            const float *src = run >= 0 ? p.corr + (base + run) * D : p.corr_state + (long long)a * D;
            float *dst = p.corr + (base + t) * D;
            for (int d = 0; d < D; ++d) dst[d] = src[d];
        }
    }
    __syncthreads();
    if (F - 1 >= t0 && F - 1 < t1) {
        p.last_idx[a] = p.argmax[base + F - 1];
        p.last_rad[a] = p.doa_rad[base + F - 1];
        p.last_prob[a] = p.prob[base + F - 1];
    }
}

}  // namespace mca
