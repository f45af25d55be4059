// kernels_multiband.hip -- MultibandBinarualLocalisation (src/mcarray/MultibandBinarualLocalisation.cpp:145-258)
// for batches of frames (gfx950).
//
//   k_mb_analyse   PCM (2 ch) -> windowed FFT -> PHAT cross-spectrum -> per band b: un-smoothed GCC-PHAT at the D
//                  steering delays restricted to the band's bins (processOneSubband :175-177: the band filter only
//                  selects bins, PHAT discards its magnitude), band energies (:188), frame powers for the gate
//   k_mb_scan      per (band, delay): corr = (1-m) corr + m prev (:180-183), first-max per band (:184), the
//                  energy-weighted DOA histogram (:190) and its argmax / share (:227-233)
//   k_mb_summary   the power gate (:125-143, :214-225) and _currentDOA / _prob (:237-255), in frame order
#include "fft_block.h"
#include "mca_internal.h"
#include "pair_balance.h"

namespace mca {

constexpr int MB_WARM = 24;     // frames of recursion warm-up per scan chunk (0.4^24 = 2.8e-10 << fp32 epsilon)

// grid (frames, arrays), 256 threads.  LDS: xs [2][H+1] float2, G [K] float2, pw [K] float, red [8] float
__global__ __launch_bounds__(256) void k_mb_analyse(MbAnalyseArgs p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int logH = p.logH, H = 1 << logH, zs = H + 1, K = H + 1;
    float2 *xs = reinterpret_cast<float2 *>(smem_raw);
    float2 *G = xs + 2 * zs;
    float *pw = reinterpret_cast<float *>(G + K);
    float *red = pw + K;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int a = blockIdx.y, f = blockIdx.x;
    const long long row = (long long)a * p.n_frames + f;
    const float *base = p.pcm + (long long)a * p.array_stride;

    load_frames(xs, zs, 2, logH, base, p.ch_stride, (long long)f, p.window, tid, 256);
    block_fft_dit(xs, zs, 2, logH, p.tw, p.N, tid, 256);
    split_forward(xs, zs, 2, logH, p.tw, tid, 256);

    // per-bin power and PHAT cross-spectrum G = L conj(R) / |L conj(R)|
    const int Kh = K / 2;                                      // bins FFTPower sees when handed (N+2)/2 doubles (:216)
    float full = 0.f, half = 0.f;
    for (int k = tid; k < K; k += 256) {
        const float2 l = xs[k], r = xs[zs + k];
        const float pk = l.x * l.x + l.y * l.y + r.x * r.x + r.y * r.y;
        pw[k] = pk;
        full += (k == 0 || k == K - 1) ? pk : 2.f * pk;
        if (k < Kh) half += (k == 0 || k == Kh - 1) ? pk : 2.f * pk;
        G[k] = whiten_g(cmulc(l, r));
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { full += __shfl_down(full, off); half += __shfl_down(half, off); }
    if (lane == 0) { red[wave] = full; red[4 + wave] = half; }
    __syncthreads();
    if (tid == 0) {
        const float n2 = (float)p.N * (float)p.N, h2 = (float)(K - 2) * (float)(K - 2);
        p.p_full[row] = (red[0] + red[1] + red[2] + red[3]) / n2 * 0.5f;     // mean over the 2 channels
        p.p_half[row] = (red[4] + red[5] + red[6] + red[7]) / h2 * 0.5f;
    }
    // band energies: FFTPower of the sub-band frames = (1/N^2) sum_k w_k H_b[k]^2 (|L|^2 + |R|^2) / 2
    for (int b = wave; b < p.nbins; b += 4) {
        float s = 0.f;
        for (int k = p.lo[b] + lane; k <= p.hi[b]; k += 64) {
            const float h = p.coef[(long long)b * K + k];
            const float w = (k == 0 || k == K - 1) ? 1.f : 2.f;
            s += w * h * h * pw[k];
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off);
        if (lane == 0) p.band_energy[row * p.nbins + b] = s / ((float)p.N * (float)p.N) * 0.5f;
    }
    // band correlations at the steering delays
    const int BD = p.nbins * p.D;
    for (int e = tid; e < BD; e += 256) {
        const int b = e / p.D, d = e - b * p.D;
        float s = 0.f;
        for (int k = p.lo[b]; k <= p.hi[b]; ++k) {
            const float2 g = G[k], t = p.T[(long long)k * p.D + d];
            s += g.x * t.x - g.y * t.y;
        }
        p.raw[row * BD + e] = s;
    }
}

// The same stage for N = 1024 (44.1 / 48 kHz) with the wave-level FFT of fft512.h: grid (frames / fpb, arrays),
// 512 threads; a pass takes 4 frames x 2 channels (wave w -> frame slot w >> 1, channel w & 1), then the 512 threads
// work on the 4 spectra pairs.  LDS: spec [8][FFT_SCRATCH] float2, G [4][520] float2, pw [4][520] float, table, red.
__global__ __launch_bounds__(512) void k_mb_analyse_1024(MbAnalyseArgs p, int fpb)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    constexpr int K = FFT_K, Kh = FFT_K / 2;
    float2 *spec = reinterpret_cast<float2 *>(smem_raw);
    float2 *G = spec + 8 * FFT_SCRATCH;                         // [4][520]
    float *pw = reinterpret_cast<float *>(G + 4 * 520);         // [4][520]
    float2 *tab = reinterpret_cast<float2 *>(pw + 4 * 520);     // [TW_WORDS]
    float *red = reinterpret_cast<float *>(tab + TW_WORDS);     // [4][2][8]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int a = blockIdx.y;
    const int f_begin = blockIdx.x * fpb, f_end = min(f_begin + fpb, p.n_frames);
    const int slot = wave >> 1, ch = wave & 1;
    const float *base = p.pcm + (long long)a * p.array_stride + (long long)ch * p.ch_stride;
    const int BD = p.nbins * p.D;

    fft_table_init(tab, p.window, tid, 512);
    __syncthreads();
    FftTw tw{tab};
    for (int f = f_begin; f < f_end; f += 4) {
        const int nb = min(4, f_end - f);
        if (slot < nb) {
            const float2 *src = reinterpret_cast<const float2 *>(base + (long long)(f + slot) * FFT_H);
            float2 v[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) { const float2 x = src[lane + 64 * r], w = tw.win(r, lane); v[r] = make_float2(x.x * w.x, x.y * w.y); }
            rfft1024(v, spec + wave * FFT_SCRATCH, lane, tw);
        }
        __syncthreads();
        // per-bin power and PHAT cross-spectrum of the nb frames: wave pair (2j, 2j+1) -> frame j, 128 threads x 4 bins (+ Nyquist)
        {
            const int j = tid >> 7, t = tid & 127;
            float full = 0.f, half = 0.f;
            if (j < nb) {
                const float2 *L = spec + (2 * j) * FFT_SCRATCH, *R = L + FFT_SCRATCH;
                for (int k = t; k < K; k += 128) {
                    const float2 l = L[k], r = R[k];
                    const float pk = l.x * l.x + l.y * l.y + r.x * r.x + r.y * r.y;
                    pw[j * 520 + k] = pk;
                    full += (k == 0 || k == K - 1) ? pk : 2.f * pk;
                    if (k < Kh) half += (k == 0 || k == Kh - 1) ? pk : 2.f * pk;
                    G[j * 520 + k] = whiten_g(cmulc(l, r));
                }
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) { full += __shfl_down(full, off); half += __shfl_down(half, off); }
            if (lane == 0) { red[wave] = full; red[8 + wave] = half; }          // waves 2j, 2j+1 belong to frame j
        }
        __syncthreads();
        if (tid < nb) {
            const long long row = (long long)a * p.n_frames + f + tid;
            const float n2 = (float)FFT_N * (float)FFT_N, h2 = (float)(K - 2) * (float)(K - 2);
            p.p_full[row] = (red[2 * tid] + red[2 * tid + 1]) / n2 * 0.5f;      // mean over the 2 channels
            p.p_half[row] = (red[8 + 2 * tid] + red[8 + 2 * tid + 1]) / h2 * 0.5f;
        }
        // band energies: 8 lanes per (frame, band)
        for (int q = tid >> 3; q < nb * p.nbins; q += 64) {
            const int j = q / p.nbins, b = q - j * p.nbins;
            float s = 0.f;
            for (int k = p.lo[b] + (tid & 7); k <= p.hi[b]; k += 8) {
                const float h = p.coef[(long long)b * K + k];
                const float w = (k == 0 || k == K - 1) ? 1.f : 2.f;
                s += w * h * h * pw[j * 520 + k];
            }
            s += __shfl_xor(s, 4); s += __shfl_xor(s, 2); s += __shfl_xor(s, 1);
            if ((tid & 7) == 0) p.band_energy[((long long)a * p.n_frames + f + j) * p.nbins + b] = s / ((float)FFT_N * (float)FFT_N) * 0.5f;
        }
        // band correlations at the steering delays
        for (int e = tid; e < nb * BD; e += 512) {
            const int j = e / BD, r = e - j * BD, b = r / p.D, d = r - b * p.D;
            float s = 0.f;
            for (int k = p.lo[b]; k <= p.hi[b]; ++k) {
                const float2 g = G[j * 520 + k], t = p.T[(long long)k * p.D + d];
                s += g.x * t.x - g.y * t.y;
            }
            p.raw[((long long)a * p.n_frames + f + j) * BD + r] = s;
        }
        __syncthreads();
    }
}

// The same stage for N = 512 (16 kHz at the module's 0.025 s frame rate): the two channels of a frame are ONE 512-point
// complex transform (fft512.h: rfft512_pair), so a pass takes 8 frames (wave = frame slot).  LDS: spec [8][2][N512_ROW],
// 8 wave scratches (reused for G [8][260] float2 and pw [8][260] float once the transforms are done), table, red.
__global__ __launch_bounds__(512) void k_mb_analyse_512(MbAnalyseArgs p, int fpb)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    constexpr int K = N512_K, Kh = N512_K / 2, GS = 260;
    float2 *spec = reinterpret_cast<float2 *>(smem_raw);        // [8][2][N512_ROW]
    float2 *scr = spec + 16 * N512_ROW;                          // [8][FFT_SCRATCH]
    float2 *G = scr;                                             // [8][GS]   (after the transforms)
    float *pw = reinterpret_cast<float *>(G + 8 * GS);           // [8][GS]
    float2 *tab = scr + 8 * FFT_SCRATCH;                         // [TW_WIN]
    float *red = reinterpret_cast<float *>(tab + TW_WIN);        // [2][8]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int a = blockIdx.y;
    const int f_begin = blockIdx.x * fpb, f_end = min(f_begin + fpb, p.n_frames);
    const float *base = p.pcm + (long long)a * p.array_stride;
    const int BD = p.nbins * p.D;

    fft_table_init(tab, nullptr, tid, 512);
    float wreg[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) wreg[r] = 0.5f * p.window[lane + 64 * r];
    __syncthreads();
    FftTw tw{tab};
    for (int f = f_begin; f < f_end; f += 8) {
        const int nb = min(8, f_end - f);
        if (wave < nb) {
            float2 v[8];
            load_pair_512(v, base, p.ch_stride, 0, 2, (long long)(f + wave) * N512_H, wreg, lane);
            const PairBalance pb = pair_balance_512(v);        // (the PHAT cross-spectrum below keeps the phase only: pair_balance.h)
            rfft512_pair(v, scr + wave * FFT_SCRATCH, spec + (2 * wave) * N512_ROW, spec + (2 * wave + 1) * N512_ROW, lane, tw);
            pair_restore_512(pb, spec + (2 * wave) * N512_ROW, spec + (2 * wave + 1) * N512_ROW, lane);
        }
        __syncthreads();
        // per-bin power and PHAT cross-spectrum: wave j -> frame j
        {
            const int j = wave;
            float full = 0.f, half = 0.f;
            if (j < nb) {
                const float2 *L = spec + (2 * j) * N512_ROW, *R = L + N512_ROW;
                for (int k = lane; k < K; k += 64) {
                    const float2 l = L[k], r = R[k];
                    const float pk = l.x * l.x + l.y * l.y + r.x * r.x + r.y * r.y;
                    pw[j * GS + k] = pk;
                    full += (k == 0 || k == K - 1) ? pk : 2.f * pk;
                    if (k < Kh) half += (k == 0 || k == Kh - 1) ? pk : 2.f * pk;
                    G[j * GS + k] = whiten_g(cmulc(l, r));
                }
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) { full += __shfl_down(full, off); half += __shfl_down(half, off); }
            if (lane == 0) { red[wave] = full; red[8 + wave] = half; }
        }
        __syncthreads();
        if (tid < nb) {
            const long long row = (long long)a * p.n_frames + f + tid;
            const float n2 = 512.f * 512.f, h2 = (float)(K - 2) * (float)(K - 2);
            p.p_full[row] = red[tid] / n2 * 0.5f;                               // mean over the 2 channels
            p.p_half[row] = red[8 + tid] / h2 * 0.5f;
        }
        // band energies: 8 lanes per (frame, band)
        for (int q = tid >> 3; q < nb * p.nbins; q += 64) {
            const int j = q / p.nbins, b = q - j * p.nbins;
            float s = 0.f;
            for (int k = p.lo[b] + (tid & 7); k <= p.hi[b]; k += 8) {
                const float h = p.coef[(long long)b * K + k];
                const float w = (k == 0 || k == K - 1) ? 1.f : 2.f;
                s += w * h * h * pw[j * GS + k];
            }
            s += __shfl_xor(s, 4); s += __shfl_xor(s, 2); s += __shfl_xor(s, 1);
            if ((tid & 7) == 0) p.band_energy[((long long)a * p.n_frames + f + j) * p.nbins + b] = s / (512.f * 512.f) * 0.5f;
        }
        // band correlations at the steering delays
        for (int e = tid; e < nb * BD; e += 512) {
            const int j = e / BD, r = e - j * BD, b = r / p.D, d = r - b * p.D;
            float s = 0.f;
            for (int k = p.lo[b]; k <= p.hi[b]; ++k) {
                const float2 g = G[j * GS + k], t = p.T[(long long)k * p.D + d];
                s += g.x * t.x - g.y * t.y;
            }
            p.raw[((long long)a * p.n_frames + f + j) * BD + r] = s;
        }
        __syncthreads();
    }
}

// grid (chunks, arrays), blockDim = roundup(nbins * D, 64).  LDS: sC [chunk][BD] float, sE [chunk][D] float, sIdx [chunk][nbins] int
__global__ __launch_bounds__(1024) void k_mb_scan(MbScanArgs p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int D = p.D, nb = p.nbins, BD = nb * D;
    float *sC = reinterpret_cast<float *>(smem_raw);
    float *sE = sC + p.chunk * BD;
    int *sIdx = reinterpret_cast<int *>(sE + p.chunk * D);
    const int e = threadIdx.x, lane = e & 63, wave = e >> 6, nwaves = blockDim.x >> 6;
    const int a = blockIdx.y;
    const int t_start = blockIdx.x * p.chunk, t_end = min(t_start + p.chunk, p.n_frames), nt = t_end - t_start;
    const int warm_start = max(0, t_start - MB_WARM);
    const float *raw = p.raw + (long long)a * p.n_frames * BD;
    if (e < BD) {
        float c = warm_start == 0 ? p.corr_in[(long long)a * BD + e] : 0.f;
        for (int t0 = warm_start; t0 < t_end; t0 += 8) {              // 8 independent loads in flight, then the serial recursion
            float r8[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) r8[i] = raw[(long long)min(t0 + i, t_end - 1) * BD + e];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int t = t0 + i;
                if (t < t_end) {
                    c = p.one_minus_mem * r8[i] + p.mem * c;                              // :180-183
                    if (t >= t_start) {
                        sC[(t - t_start) * BD + e] = c;
                        if (p.band_corr) p.band_corr[((long long)a * p.n_frames + t) * BD + e] = c;
                    }
                }
            }
        }
        if (t_end == p.n_frames) p.corr_out[(long long)a * BD + e] = c;
    }
    __syncthreads();
    for (int q = wave; q < nt * nb; q += nwaves) {                                        // maxidx per (frame, band) :184
        const float *cr = sC + q * D;                                                     // q = tl * nb + b
        float bv = -INFINITY; int bi = 0x7fffffff;
        for (int dd = lane; dd < D; dd += 64) { const float v = cr[dd]; if (v > bv) { bv = v; bi = dd; } }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const float ov = __shfl_xor(bv, off); const int oi = __shfl_xor(bi, off);
            if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
        }
        if (lane == 0) sIdx[q] = bi < D ? bi : 0;        // all-NaN row: stay in range
    }
    __syncthreads();
    if (e < nt) {
        const int t = t_start + e;
        const long long row = (long long)a * p.n_frames + t;
        float *E = sE + e * D;
        for (int d = 0; d < D; ++d) E[d] = 0.f;                                           // processSetup :147
        for (int b = 0; b < nb; ++b) {
            const int idx = sIdx[e * nb + b];
            E[idx] += p.band_energy[row * nb + b];                                        // :190
            if (p.band_idx) p.band_idx[row * nb + b] = idx;
        }
        float sum = 0.f, mx = E[0]; int idx = 0;
        for (int d = 0; d < D; ++d) { sum += E[d]; if (E[d] > mx) { mx = E[d]; idx = d; } }   // :227-228
        p.hist_idx[row] = idx;
        p.hist_prob[row] = sum != 0.f ? E[idx] / sum : 0.f;                               // :230-233
        if (p.energy_in_doa) for (int d = 0; d < D; ++d) p.energy_in_doa[row * D + d] = E[d];
    }
}

// grid (arrays), 256 threads.  Only the floor estimation over the first 3 s is sequential (thread 0, <= 141
// frames); after it every frame decides for itself, and "a gated-out frame keeps the previous _currentDOA"
// (_doaMemoryFactorSilence = 1, :254) is an inclusive prefix max of the last fired frame over per-thread segments.
__global__ __launch_bounds__(256) void k_mb_summary(MbSummaryArgs p)
{
    __shared__ int sLast[256];
    __shared__ int s_first_free;
    __shared__ double s_floor;
    const int a = blockIdx.x, tid = threadIdx.x, F = p.n_frames;
    const long long base = (long long)a * F;
    double *g = p.gate + (long long)a * 4;
    const float cur_in = p.cur[a * 2];
    if (tid == 0) {
        double acc = g[0], consumed = g[1], floor_db = g[2];
        bool est = g[3] != 0.0;
        const double per_frame = (double)(2 * p.K - 2);
        int t = 0;
        for (; t < F && !est; ++t) {                                                      // setPowerFloor :125-143
            acc += (double)p.p_half[base + t] * per_frame;
            consumed += per_frame;
            if (consumed >= (double)p.needed_samples) {
                est = true;
                floor_db = 10.0 * log10(acc / consumed) + (double)p.margin_db;
                acc = floor_db;
            }
            // power == _powerFloor here (the function returns it), so only an ungated module fires (:225)
            if (p.voiced) p.voiced[base + t] = p.use_floor ? 0 : 1;
            if (p.power) p.power[base + t] = (float)acc;
        }
        g[0] = acc; g[1] = consumed; g[2] = floor_db; g[3] = est ? 1.0 : 0.0;
        s_first_free = t; s_floor = floor_db;
    }
    __syncthreads();
    const int first_free = s_first_free;
    const double floor_db = s_floor;
    const int per = (F + 255) / 256;
    const int t0 = tid * per, t1 = min(t0 + per, F);
    // does frame t fire the callback?  During the estimation the function returns _powerFloor itself, so
    // `power > _powerFloor` is false (:216,:225); afterwards the LINEAR frame power meets the dB floor (:221)
    auto fires = [&](int t) -> bool {
        if (!p.use_floor) return true;
        return t >= first_free && (double)p.p_full[base + t] > floor_db;
    };
    int last = -1;
    for (int t = t0; t < t1; ++t) {
        const bool fire = fires(t);
        if (t >= first_free) {
            if (p.voiced) p.voiced[base + t] = fire ? 1 : 0;
            if (p.power) p.power[base + t] = p.p_full[base + t];
        }
        p.prob[base + t] = fire ? p.hist_prob[base + t] : -100000.f;                      // :233 / :255
        if (fire) last = t;
    }
    sLast[tid] = last;
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {
        const int other = tid >= off ? sLast[tid - off] : -1;
        __syncthreads();
        sLast[tid] = max(sLast[tid], other);
        __syncthreads();
    }
    int run = tid > 0 ? sLast[tid - 1] : -1;                    // last fired frame before this thread's segment
    for (int t = t0; t < t1; ++t) {
        if (fires(t)) run = t;
        p.doa_rad[base + t] = run >= 0 ? p.grid[p.hist_idx[base + run]] : cur_in;        // :237-239, _doaMemoryFactor = 0
    }
    if (F - 1 >= t0 && F - 1 < t1) { p.cur[a * 2] = p.doa_rad[base + F - 1]; p.cur[a * 2 + 1] = p.prob[base + F - 1]; }
}

}  // namespace mca
