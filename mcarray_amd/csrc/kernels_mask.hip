// kernels_mask.hip -- 2-channel spatial + temporal time-frequency masking
// (FastBinauralMasking::processParametrisation, FastBinauralMasking.cpp:126-210) on gfx950.
//
// The reference loops over 45 mel bands and, per band, multiplies both spectra by the band's
// filter, takes four reductions over the bins, decides, scales and re-sums.  The filters are
// triangles, so every bin belongs to at most two (adjacent) bands: the per-bin products
// |L|^2, |R|^2, Re(conj(L) R), |L+R|^2/4 are formed once and each band sums them over its own
// support with weight H_b[k]^2; the output is X[k] * sum_b g_b H_b[k] (g_b = the band's gain;
// the reference scales only the first N doubles, so the Nyquist bin keeps gain 1).
//
//   k_mask_stream   batched streams: STFT -> masking -> ISTFT -> overlap-add, fp32
//   k_mask_frame    one frame of double CCS spectra in place (the DSPONE hook), double
#include "fft_block.h"
#include "mca_internal.h"

namespace mca {

__device__ __forceinline__ void mask_decide(const MaskParams &mp, int b, float S_mix, float S_LL, float S_RR, float S_LR,
                                            float S_LL512, float S_RR512, float &Q, float noise_b, long long gframe,
                                            int &dec, float &gL, float &gR, int H = FFT_H, int K = FFT_K)
{
    // temportalMasking :477-493 ; getFramePower/getPower :496-538 (an RMS over the first N/2 bins)
    const float P = sqrtf(S_mix / (float)H);
    Q = Q * mp.lambda + mp.one_minus_lambda * P;
    bool temp = P < mp.reject * Q;
    bool spat = false;
    if (mp.alg == 0 || mp.alg == 1) {                       // BOTH or SPATIAL :159-166
        const float num = S_LR / (float)K;             // normaliseFFTCorrelation :410-460
        float nc;
        if (num == 0.f) nc = 0.f;
        else {
            const float den = sqrtf((S_LL / (float)K) * (S_RR / (float)K));
            nc = den == 0.f ? 1.f : num / den;
        }
        spat = nc < mp.thr[b];
        if (mp.alg == 1) temp = false;
    }
    dec = spat ? 2 : (temp ? 1 : 0);
    gL = 1.f; gR = 1.f;                                     // enhanceFactor = 1 (FastBinauralMasking.h:120)
    if (dec != 0) {
        switch (mp.method) {                                // maskFrame :294-313
        case 3: gL = gR = 1.f / 1000.f; break;              // FULL: zeroFrame :214-217
        case 0: gL = gR = 1.f / (dec == 2 ? 10.f : 3.f); break;   // FACTOR :289-292 with .h:116-117
        case 1: {                                           // RELATIVE: maskFrameByScaling :245-287
            float fl = (S_LL / (float)K) * mp.rho, fr = (S_RR / (float)K) * mp.rho;
            if (Q < 1e-10f) { fl = mp.rho; fr = mp.rho; } else { fl /= Q; fr /= Q; }
            gL = sqrtf(fl); gR = sqrtf(fr);
            break;
        }
        case 4: {                                           // NOISY: noisyFrame :219-243 (inactive during the first two frames)
            if (gframe >= 2) {
                const float pl = sqrtf(S_LL512 / (float)H), pr = sqrtf(S_RR512 / (float)H);
                gL = pl > 0.f ? noise_b / pl : 1.f;
                gR = pr > 0.f ? noise_b / pr : 1.f;
            }
            break;
        }
        default: break;
        }
    }
}


__global__ __launch_bounds__(512) void k_mask_stream(MaskArgs p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float2 *spec = reinterpret_cast<float2 *>(smem_raw);                  // [MK_NB][2][FFT_SCRATCH]
    float *binq = reinterpret_cast<float *>(spec + MK_NB * 2 * FFT_SCRATCH);     // [MK_NB][3][520]: |L|^2, |R|^2, Re(conj(L) R)
    float2 *tab = reinterpret_cast<float2 *>(binq + MK_NB * 3 * 520);      // [TW_WIN] twiddles (the window stays in registers)
    float *sums = reinterpret_cast<float *>(tab + TW_WIN);                 // [MK_NB][48][6]
    float *gains = sums + MK_NB * 48 * 6;                                  // [MK_NB][48][2]
    float2 *wq = reinterpret_cast<float2 *>(gains + MK_NB * 48 * 2);       // [520] (H_kb[k], H_{kb+1}[k]): the two bands covering bin k
    signed char *kbs = reinterpret_cast<signed char *>(wq + 520);          // [520] first band covering bin k (-1: none)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int s = blockIdx.y;
    const MaskParams &mp = *p.mp;
    const int t0 = blockIdx.x * p.ft, t1 = min(t0 + p.ft, p.n_frames);
    const int tfull = t0 > 0 ? t0 - 1 : 0;                 // first frame that is processed completely (OLA carry)
    const int tbeg = t0 > 0 ? max(0, tfull - MK_WARM) : 0; // first frame of the Q warm-up
    const bool passthrough = mp.method == 5;               // NOTHING :130-134

    fft_table_init(tab, nullptr, tid, 512);
    for (int k = tid; k < FFT_K; k += 512) { wq[k] = make_float2(mp.kw0[k], mp.kw1[k]); kbs[k] = (signed char)mp.kb[k]; }
    __syncthreads();
    FftTw tw{tab};

    float Q = 0.f, noise_b = 0.f;
    if (tid < 45) {
        if (tbeg == 0) Q = p.Q_in[s * 45 + tid];
        noise_b = p.noise[s * 45 + tid];
    }
    float carry[2] = {0.f, 0.f};
    if (t0 == 0) { carry[0] = p.tail_in[(s * 2 + 0) * FFT_H + tid]; carry[1] = p.tail_in[(s * 2 + 1) * FFT_H + tid]; }
    const int jw = wave >> 1, cw = wave & 1;
    const float *base = p.pcm + (long long)s * p.stream_stride + (long long)cw * p.ch_stride;
    float2 wreg[8];                                        // this lane's window samples, halved (the 1/2 of the split step)
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const float2 w = reinterpret_cast<const float2 *>(p.window)[lane + 64 * r];
        wreg[r] = make_float2(0.5f * w.x, 0.5f * w.y);
    }

    for (int tb = tbeg; tb < t1; tb += MK_NB) {
        const int nb = min(MK_NB, t1 - tb);
        // (1) analysis: wave -> (frame jw, channel cw)
        if (jw < nb) {
            const float2 *src = reinterpret_cast<const float2 *>(base + (long long)(tb + jw) * FFT_H);
            float2 v[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const float2 x = src[lane + 64 * r], w = wreg[r];
                v[r] = make_float2(x.x * w.x, x.y * w.y);
            }
            rfft1024(v, spec + (jw * 2 + cw) * FFT_SCRATCH, lane, tw);
        }
        __syncthreads();
        // (2) per-bin products
        for (int e = tid; e < nb * FFT_K; e += 512) {
            const int j = e / FFT_K, k = e - j * FFT_K;
            const float2 L = spec[(j * 2) * FFT_SCRATCH + k], R = spec[(j * 2 + 1) * FFT_SCRATCH + k];
            // |L/2 + R/2|^2 (divC(2) + add :510-512) = (|L|^2 + |R|^2 + 2 Re(conj(L) R)) / 4: formed from the band sums
            float *bq = binq + j * 3 * 520 + k;
            bq[0] = L.x * L.x + L.y * L.y; bq[520] = R.x * R.x + R.y * R.y; bq[1040] = L.x * R.x + L.y * R.y;
        }
        __syncthreads();
        // (3) band sums over the band's support, weight H_b[k]^2: 8 lanes per (frame, band) pair, bins strided over
        // the lanes (the widest mel band has ~100 bins), 3-level shuffle reduction.  One thread per pair walking
        // its whole support serially took 27 % of the kernel.
        for (int q = tid >> 3; q < nb * 45; q += 64) {
            const int j = q / 45, b = q - j * 45;
            float a0 = 0, a1 = 0, a2 = 0, a3 = 0, a4 = 0, a5 = 0;           // a3: Re(conj(L) R) over the first N/2 bins
            for (int k = mp.lo[b] + (tid & 7); k <= mp.hi[b]; k += 8) {
                const float2 hw = wq[k];
                const float h = kbs[k] == b ? hw.x : hw.y;
                const float w = h * h;
                const float *bq = binq + j * 3 * 520 + k;
                const float ll = bq[0], rr = bq[520], lr = bq[1040];
                a0 += w * ll; a1 += w * rr; a2 += w * lr;
                if (k < FFT_H) { a3 += w * lr; a4 += w * ll; a5 += w * rr; }
            }
#pragma unroll
            for (int off = 4; off > 0; off >>= 1) {
                a0 += __shfl_xor(a0, off); a1 += __shfl_xor(a1, off); a2 += __shfl_xor(a2, off);
                a3 += __shfl_xor(a3, off); a4 += __shfl_xor(a4, off); a5 += __shfl_xor(a5, off);
            }
            if ((tid & 7) == 0) {
                float *o = sums + (j * 48 + b) * 6;
                o[0] = a0; o[1] = a1; o[2] = a2; o[3] = 0.25f * (a4 + a5 + 2.f * a3); o[4] = a4; o[5] = a5;
            }
        }
        __syncthreads();
        // (4) the recursion over frames (the only sequential part): thread = band
        if (tid < 45) {
            for (int j = 0; j < nb; ++j) {
                const float *o = sums + (j * 48 + tid) * 6;
                int dec; float gL, gR;
                const long long gframe = p.frames_done + tb + j;
                mask_decide(mp, tid, o[3], o[0], o[1], o[2], o[4], o[5], Q, noise_b, gframe, dec, gL, gR);
                if (gframe == 0) noise_b = Q;                               // noise <- Q after the first call :193-197
                gains[(j * 48 + tid) * 2] = gL; gains[(j * 48 + tid) * 2 + 1] = gR;
                if (p.decisions && tb + j >= t0) p.decisions[((long long)s * p.n_frames + tb + j) * 45 + tid] = dec;
            }
        }
        __syncthreads();
        if (tb + nb > tfull) {
            // (5) out[k] = X[k] * sum_b g_b H_b[k]  (Nyquist bin: gain 1)
            for (int e = tid; e < nb * 2 * FFT_K; e += 512) {
                const int jc = e / FFT_K, k = e - jc * FFT_K, j = jc >> 1, ch = jc & 1;
                float m = 1.f;
                if (!passthrough) {
                    const int b0 = kbs[k];
                    m = 0.f;
                    if (b0 >= 0) {
                        const float g0 = k < FFT_H ? gains[(j * 48 + b0) * 2 + ch] : 1.f;
                        const float g1 = (k < FFT_H && b0 + 1 < 45) ? gains[(j * 48 + b0 + 1) * 2 + ch] : 1.f;
                        const float2 hw = wq[k];
                        m = g0 * hw.x + g1 * hw.y;
                    }
                }
                float2 x = spec[jc * FFT_SCRATCH + k];
                spec[jc * FFT_SCRATCH + k] = make_float2(x.x * m, x.y * m);
            }
            __syncthreads();
            // (6) synthesis
            if (jw < nb) irfft1024(spec + (jw * 2 + cw) * FFT_SCRATCH, lane, tw);
            __syncthreads();
            // (7) overlap-add
            for (int j = 0; j < nb; ++j) {
                const int t = tb + j;
                if (t < tfull) continue;
#pragma unroll
                for (int ch = 0; ch < 2; ++ch) {
                    const float *y = reinterpret_cast<const float *>(spec + (j * 2 + ch) * FFT_SCRATCH);
                    if (t >= t0) p.out[((long long)s * 2 + ch) * (long long)p.n_frames * FFT_H + (long long)t * FFT_H + tid] = carry[ch] + y[tid];
                    carry[ch] = y[tid + FFT_H];
                }
            }
        }
        __syncthreads();
    }
    if (t1 == p.n_frames) {
        p.tail_out[(s * 2 + 0) * FFT_H + tid] = carry[0];
        p.tail_out[(s * 2 + 1) * FFT_H + tid] = carry[1];
        if (tid < 45) { p.Q_out[s * 45 + tid] = Q; }
    }
    if (tid < 45 && p.frames_done == 0 && tbeg == 0) p.noise[s * 45 + tid] = noise_b;   // only one block per stream has tbeg == 0
}

// ---------------------------------------------------------------------------------------
// k_mask_stream_gen: the same module for any power-of-two frame length (the reference derives N from the sample rate:
// 2^round(log2(0.050 fs)), FastBinauralMasking.h:112 -- 1024 at 16 kHz, 2048 at 44.1/48 kHz, 512 at 8 kHz).
// grid (runs of ft frames, streams), 256 ... 1024 threads, one frame at a time with the block-cooperative FFT of
// fft_block.h; LDS = 2 (H + 1) float2 + 3 K + 2 H + 48 * 8 floats.  A run starts MK_WARM + 1 frames early: the Q
// recursion warms up (0.04^8) and the frame before the run rebuilds the overlap-add carry.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void k_mask_stream_gen(MaskGenArgs pg)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const MaskArgs &p = pg.a;
    const int logH = pg.logH, H = 1 << logH, K = H + 1, zs = H + 1;
    float2 *spec = reinterpret_cast<float2 *>(smem_raw);                  // [2][H + 1]
    float *binq = reinterpret_cast<float *>(spec + 2 * zs);               // [3][K]
    float *carry = binq + 3 * K;                                          // [2][H]
    float *sums = carry + 2 * H;                                          // [48][6]
    float *gains = sums + 48 * 6;                                         // [48][2]
    const int tid = threadIdx.x, NT = blockDim.x;
    const int s = blockIdx.y;
    const MaskParams &mp = *p.mp;
    const int t0 = blockIdx.x * p.ft, t1 = min(t0 + p.ft, p.n_frames);
    const int tfull = t0 > 0 ? t0 - 1 : 0;
    const int tbeg = t0 > 0 ? max(0, tfull - MK_WARM) : 0;
    const bool passthrough = mp.method == 5;                              // NOTHING :130-134

    float Q = 0.f, noise_b = 0.f;
    if (tid < 45) {
        if (tbeg == 0) Q = p.Q_in[s * 45 + tid];
        noise_b = p.noise[s * 45 + tid];
    }
    for (int e = tid; e < 2 * H; e += NT) carry[e] = t0 == 0 ? p.tail_in[(long long)s * 2 * H + e] : 0.f;
    const float *base = p.pcm + (long long)s * p.stream_stride;
    const float sc = 1.0f / (float)H;

    for (int t = tbeg; t < t1; ++t) {
        // (1) analysis of both channels
        load_frames(spec, zs, 2, logH, base, p.ch_stride, (long long)t, p.window, tid, NT);
        block_fft_dit(spec, zs, 2, logH, pg.tw, pg.N, tid, NT);
        split_forward(spec, zs, 2, logH, pg.tw, tid, NT);
        // (2) per-bin products
        for (int k = tid; k < K; k += NT) {
            const float2 L = spec[k], R = spec[zs + k];
            binq[k] = L.x * L.x + L.y * L.y; binq[K + k] = R.x * R.x + R.y * R.y; binq[2 * K + k] = L.x * R.x + L.y * R.y;
        }
        __syncthreads();
        // (3) band sums: 8 lanes per band
        for (int b = tid >> 3; b < 45; b += NT >> 3) {
            float a0 = 0, a1 = 0, a2 = 0, a3 = 0, a4 = 0, a5 = 0;
            for (int k = mp.lo[b] + (tid & 7); k <= mp.hi[b]; k += 8) {
                const float2 hw = pg.kw[k];
                const float h = pg.kb[k] == b ? hw.x : hw.y;
                const float w = h * h;
                const float ll = binq[k], rr = binq[K + k], lr = binq[2 * K + k];
                a0 += w * ll; a1 += w * rr; a2 += w * lr;
                if (k < H) { a3 += w * lr; a4 += w * ll; a5 += w * rr; }
            }
#pragma unroll
            for (int off = 4; off > 0; off >>= 1) {
                a0 += __shfl_xor(a0, off); a1 += __shfl_xor(a1, off); a2 += __shfl_xor(a2, off);
                a3 += __shfl_xor(a3, off); a4 += __shfl_xor(a4, off); a5 += __shfl_xor(a5, off);
            }
            if ((tid & 7) == 0) {
                float *o = sums + b * 6;
                o[0] = a0; o[1] = a1; o[2] = a2; o[3] = 0.25f * (a4 + a5 + 2.f * a3); o[4] = a4; o[5] = a5;
            }
        }
        __syncthreads();
        // (4) decisions: thread = band
        if (tid < 45) {
            const float *o = sums + tid * 6;
            int dec; float gL, gR;
            const long long gframe = p.frames_done + t;
            mask_decide(mp, tid, o[3], o[0], o[1], o[2], o[4], o[5], Q, noise_b, gframe, dec, gL, gR, H, K);
            if (gframe == 0) noise_b = Q;                                   // noise <- Q after the first call :193-197
            gains[tid * 2] = gL; gains[tid * 2 + 1] = gR;
            if (p.decisions && t >= t0) p.decisions[((long long)s * p.n_frames + t) * 45 + tid] = dec;
        }
        __syncthreads();
        if (t >= tfull) {
            // (5) out[k] = X[k] * sum_b g_b H_b[k]  (Nyquist bin: gain 1)
            for (int e = tid; e < 2 * K; e += NT) {
                const int ch = e / K, k = e - ch * K;
                float m = 1.f;
                if (!passthrough) {
                    const int b0 = pg.kb[k];
                    m = 0.f;
                    if (b0 >= 0) {
                        const float g0 = k < H ? gains[b0 * 2 + ch] : 1.f;
                        const float g1 = (k < H && b0 + 1 < 45) ? gains[(b0 + 1) * 2 + ch] : 1.f;
                        const float2 hw = pg.kw[k];
                        m = g0 * hw.x + g1 * hw.y;
                    }
                }
                const float2 x = spec[ch * zs + k];
                spec[ch * zs + k] = make_float2(x.x * m, x.y * m);
            }
            __syncthreads();
            // (6) one-sided spectra -> packed Z (imaginary parts of DC and Nyquist ignored), inverse transform
            for (int e = tid; e < 2 * (H / 2 + 1); e += NT) {
                const int ch = e / (H / 2 + 1), k = e - ch * (H / 2 + 1);
                float2 *yy = spec + ch * zs;
                float2 xk = yy[k], xp = yy[H - k];
                if (k == 0) { xk.y = 0.f; xp.y = 0.f; }
                const float2 ev = make_float2(0.5f * (xk.x + xp.x), 0.5f * (xk.y - xp.y));
                const float2 df = make_float2(0.5f * (xk.x - xp.x), 0.5f * (xk.y + xp.y));
                const float2 od = cmulc(df, pg.tw[k]);
                yy[k] = make_float2(ev.x - od.y, ev.y + od.x);
                if (k != 0 && k != H - k) yy[H - k] = make_float2(ev.x + od.y, -ev.y + od.x);
            }
            __syncthreads();
            block_ifft_dif(spec, zs, 2, logH, pg.tw, pg.N, tid, NT);
            // (7) overlap-add
            for (int e = tid; e < 2 * (H / 2); e += NT) {
                const int ch = e / (H / 2), n = e - ch * (H / 2);
                const float2 lo = spec[ch * zs + (int)(__brev((unsigned)n) >> (32 - logH))];
                const float2 hi = spec[ch * zs + (int)(__brev((unsigned)(n + H / 2)) >> (32 - logH))];
                float *cr = carry + ch * H + 2 * n;
                if (t >= t0) {
                    float *o = p.out + ((long long)s * 2 + ch) * (long long)p.n_frames * H + (long long)t * H + 2 * n;
                    o[0] = cr[0] + lo.x * sc; o[1] = cr[1] + lo.y * sc;
                }
                cr[0] = hi.x * sc; cr[1] = hi.y * sc;
            }
        }
        __syncthreads();
    }
    if (t1 == p.n_frames) {
        for (int e = tid; e < 2 * H; e += NT) p.tail_out[(long long)s * 2 * H + e] = carry[e];
        if (tid < 45) p.Q_out[s * 45 + tid] = Q;
    }
    if (tid < 45 && p.frames_done == 0 && tbeg == 0) p.noise[s * 45 + tid] = noise_b;
}

// ---------------------------------------------------------------------------------------
// k_mask_stream_2048: the module at 2048-sample frames (44.1 / 48 kHz: N = 2^round(log2(0.050 fs)), FastBinauralMasking.h:112)
// on the wave-level transform.  A 2048-sample real frame is four 512-sample real sub-sequences x[4 n + r] (decimation in
// time, fft512.h): wave w = (frame slot, channel, pair of sub-sequences) transforms two of them in one 512-point complex
// transform, thread = bin recombines them (radix 4); the inverse is the mirror image.  Two frames x two channels per pass.
// grid (runs of ft frames, streams), 512 threads, 80 KiB of LDS (two workgroups per CU): the spectra take the place of
// the wave scratches between the transforms, the per-bin products that of the sub-spectra.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(512) void k_mask_stream_2048(MaskGenArgs pg)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const MaskArgs &p = pg.a;
    constexpr int H = 1024, K = 1025, XR = 1026, BQ = 1028;
    float2 *sub = reinterpret_cast<float2 *>(smem_raw);                    // [2 frames][2 ch][4][N512_ROW] sub-spectra
    float *binq = reinterpret_cast<float *>(sub);                          // [2][3][BQ] per-bin products (sub-spectra are dead)
    float *yt = reinterpret_cast<float *>(sub);                            // [2][2][2064] time-domain frames (after the inverse)
    float2 *scr = sub + 16 * N512_ROW;                                      // [8][FFT_SCRATCH] wave scratches
    float2 *X = scr;                                                        // [2][2][XR] spectra (between the transforms)
    float2 *tab = scr + 8 * FFT_SCRATCH;                                    // [TW_WIN]
    float *sums = reinterpret_cast<float *>(tab + TW_WIN);                  // [2][48][6]
    float *gains = sums + 2 * 48 * 6;                                       // [2][48][2]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int s = blockIdx.y;
    const MaskParams &mp = *p.mp;
    const int t0 = blockIdx.x * p.ft, t1 = min(t0 + p.ft, p.n_frames);
    const int tfull = t0 > 0 ? t0 - 1 : 0;
    const int tbeg = t0 > 0 ? max(0, tfull - MK_WARM) : 0;
    const bool passthrough = mp.method == 5;                                // NOTHING :130-134

    fft_table_init(tab, nullptr, tid, 512);
    float Q = 0.f, noise_b = 0.f;
    if (tid < 45) {
        if (tbeg == 0) Q = p.Q_in[s * 45 + tid];
        noise_b = p.noise[s * 45 + tid];
    }
    float carry[2][2] = {{0.f, 0.f}, {0.f, 0.f}};                           // samples tid, tid + 512 of the hop, per channel
    if (t0 == 0) {
#pragma unroll
        for (int ch = 0; ch < 2; ++ch) { carry[ch][0] = p.tail_in[((long long)s * 2 + ch) * H + tid]; carry[ch][1] = p.tail_in[((long long)s * 2 + ch) * H + tid + 512]; }
    }
    // wave -> (frame slot, channel, sub-sequence pair); the pair's samples x[4 n + 2 pr], x[4 n + 2 pr + 1] are one float2
    const int jw = wave >> 2, cw = (wave >> 1) & 1, pr = wave & 1;
    const float *base = p.pcm + (long long)s * p.stream_stride + (long long)cw * p.ch_stride;
    float2 wreg[8];                                                         // window at those samples, halved (rfft512_pair)
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const float2 w = reinterpret_cast<const float2 *>(p.window)[2 * (lane + 64 * r) + pr];
        wreg[r] = make_float2(0.5f * w.x, 0.5f * w.y);
    }
    __syncthreads();
    FftTw tw{tab};

    for (int tb = tbeg; tb < t1; tb += 2) {
        const int nb = min(2, t1 - tb);
        // (1) analysis
        if (jw < nb) {
            const float2 *src = reinterpret_cast<const float2 *>(base + (long long)(tb + jw) * H);
            float2 v[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const float2 x = src[2 * (lane + 64 * r) + pr], w = wreg[r];
                v[r] = make_float2(x.x * w.x, x.y * w.y);
            }
            float2 *S = sub + ((jw * 2 + cw) * 4 + 2 * pr) * N512_ROW;
            rfft512_pair(v, scr + wave * FFT_SCRATCH, S, S + N512_ROW, lane, tw);
        }
        __syncthreads();
        // (2) radix-4 recombination: X[k], k = 0..1024
        for (int e = tid; e < nb * 2 * 512; e += 512) {                     // thread = (frame, channel, m): one radix-4 butterfly
            const int jc = e >> 9, m = e & 511;
            combine2048_m(sub + jc * 4 * N512_ROW, m, pg.tw, X + jc * XR);
        }
        __syncthreads();
        // (3) per-bin products
        for (int e = tid; e < nb * K; e += 512) {
            const int j = e / K, k = e - j * K;
            const float2 L = X[(j * 2) * XR + k], R = X[(j * 2 + 1) * XR + k];
            float *bq = binq + j * 3 * BQ + k;
            bq[0] = L.x * L.x + L.y * L.y; bq[BQ] = R.x * R.x + R.y * R.y; bq[2 * BQ] = L.x * R.x + L.y * R.y;
        }
        __syncthreads();
        // (4) band sums: 8 lanes per (frame, band)
        for (int q = tid >> 3; q < nb * 45; q += 64) {
            const int j = q / 45, b = q - j * 45;
            float a0 = 0, a1 = 0, a2 = 0, a3 = 0, a4 = 0, a5 = 0;
            for (int k = mp.lo[b] + (tid & 7); k <= mp.hi[b]; k += 8) {
                const float2 hw = pg.kw[k];
                const float h = pg.kb[k] == b ? hw.x : hw.y;
                const float w = h * h;
                const float *bq = binq + j * 3 * BQ + k;
                const float ll = bq[0], rr = bq[BQ], lr = bq[2 * BQ];
                a0 += w * ll; a1 += w * rr; a2 += w * lr;
                if (k < H) { a3 += w * lr; a4 += w * ll; a5 += w * rr; }
            }
#pragma unroll
            for (int off = 4; off > 0; off >>= 1) {
                a0 += __shfl_xor(a0, off); a1 += __shfl_xor(a1, off); a2 += __shfl_xor(a2, off);
                a3 += __shfl_xor(a3, off); a4 += __shfl_xor(a4, off); a5 += __shfl_xor(a5, off);
            }
            if ((tid & 7) == 0) {
                float *o = sums + (j * 48 + b) * 6;
                o[0] = a0; o[1] = a1; o[2] = a2; o[3] = 0.25f * (a4 + a5 + 2.f * a3); o[4] = a4; o[5] = a5;
            }
        }
        __syncthreads();
        // (5) the recursion over frames: thread = band
        if (tid < 45) {
            for (int j = 0; j < nb; ++j) {
                const float *o = sums + (j * 48 + tid) * 6;
                int dec; float gL, gR;
                const long long gframe = p.frames_done + tb + j;
                mask_decide(mp, tid, o[3], o[0], o[1], o[2], o[4], o[5], Q, noise_b, gframe, dec, gL, gR, H, K);
                if (gframe == 0) noise_b = Q;                               // noise <- Q after the first call :193-197
                gains[(j * 48 + tid) * 2] = gL; gains[(j * 48 + tid) * 2 + 1] = gR;
                if (p.decisions && tb + j >= t0) p.decisions[((long long)s * p.n_frames + tb + j) * 45 + tid] = dec;
            }
        }
        __syncthreads();
        if (tb + nb > tfull) {
            // (6) out[k] = X[k] * sum_b g_b H_b[k]  (Nyquist bin: gain 1)
            for (int e = tid; e < nb * 2 * K; e += 512) {
                const int jc = e / K, k = e - jc * K, j = jc >> 1, ch = jc & 1;
                float m = 1.f;
                if (!passthrough) {
                    const int b0 = pg.kb[k];
                    m = 0.f;
                    if (b0 >= 0) {
                        const float g0 = k < H ? gains[(j * 48 + b0) * 2 + ch] : 1.f;
                        const float g1 = (k < H && b0 + 1 < 45) ? gains[(j * 48 + b0 + 1) * 2 + ch] : 1.f;
                        const float2 hw = pg.kw[k];
                        m = g0 * hw.x + g1 * hw.y;
                    }
                }
                const float2 x = X[jc * XR + k];
                X[jc * XR + k] = make_float2(x.x * m, x.y * m);
            }
            __syncthreads();
            // (7) the four sub-spectra of every masked spectrum (radix 4, inverse)
            for (int e = tid; e < nb * 2 * 257; e += 512) {
                const int jc = e / 257, m = e - jc * 257;
                float2 o4[4];
                split2048_inv(X + jc * XR, m, pg.tw, o4);
#pragma unroll
                for (int r = 0; r < 4; ++r) sub[(jc * 4 + r) * N512_ROW + m] = o4[r];
            }
            __syncthreads();
            // (8) synthesis: two sub-sequences per inverse transform; the spectra are dead, the scratches are scratches again
            float2 v[8];
            if (jw < nb) {
                const float2 *Ya = sub + ((jw * 2 + cw) * 4 + 2 * pr) * N512_ROW;
                irfft512_pair(Ya, Ya + N512_ROW, scr + wave * FFT_SCRATCH, v, lane, tw);
            }
            __syncthreads();                                                // every wave has read its sub-spectra: yt may overwrite them
            if (jw < nb) {
                float *y = yt + (jw * 2 + cw) * 2064;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int n = lane + 64 * br3(i);
                    *reinterpret_cast<float2 *>(y + 4 * n + 2 * pr) = v[i];      // y[4 n + 2 pr], y[4 n + 2 pr + 1]
                }
            }
            __syncthreads();
            // (9) overlap-add, frames in order
            for (int j = 0; j < nb; ++j) {
                const int t = tb + j;
                if (t < tfull) continue;
#pragma unroll
                for (int ch = 0; ch < 2; ++ch) {
                    const float *y = yt + (j * 2 + ch) * 2064;
                    if (t >= t0) {
                        float *o = p.out + ((long long)s * 2 + ch) * (long long)p.n_frames * H + (long long)t * H;
                        o[tid] = carry[ch][0] + y[tid]; o[tid + 512] = carry[ch][1] + y[tid + 512];
                    }
                    carry[ch][0] = y[H + tid]; carry[ch][1] = y[H + tid + 512];
                }
            }
        }
        __syncthreads();
    }
    if (t1 == p.n_frames) {
#pragma unroll
        for (int ch = 0; ch < 2; ++ch) {
            p.tail_out[((long long)s * 2 + ch) * H + tid] = carry[ch][0];
            p.tail_out[((long long)s * 2 + ch) * H + tid + 512] = carry[ch][1];
        }
        if (tid < 45) p.Q_out[s * 45 + tid] = Q;
    }
    if (tid < 45 && p.frames_done == 0 && tbeg == 0) p.noise[s * 45 + tid] = noise_b;
}

// ---------------------------------------------------------------------------------------
// one frame in double, the reference's own loop order (band by band), dense filter table
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ double block_sum_d(double v, double *sred)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    __syncthreads();
    if (lane == 0) sred[wave] = v;
    __syncthreads();
    double r = sred[0];
    for (int w = 1; w < nw; ++w) r += sred[w];
    return r;
}

__global__ __launch_bounds__(256) void k_mask_frame(MaskFrameArgs p)
{
    __shared__ double sred[4];
    __shared__ double sg[2];
    const int tid = threadIdx.x, K = p.K, Kh = K - 1;    // Kh = N/2 complex bins = "first N doubles"
    const double dK = (double)K, dKh = (double)Kh;
    if (p.method == 5) return;                            // NOTHING :130-134
    for (int b = 0; b < 45; ++b) {
        const double *H = p.H + (long long)b * K;
        double s_mix = 0, s_ll = 0, s_rr = 0, s_lr = 0, s_ll_h = 0, s_rr_h = 0;
        for (int k = tid; k < K; k += 256) {
            const double h = H[k];
            const double lr = p.L[2 * k] * h, li = p.L[2 * k + 1] * h, rr = p.R[2 * k] * h, ri = p.R[2 * k + 1] * h;
            const double ll = lr * lr + li * li, rrr = rr * rr + ri * ri;
            s_ll += ll; s_rr += rrr; s_lr += lr * rr + li * ri;
            if (k < Kh) {
                const double mr = lr / 2 + rr / 2, mi = li / 2 + ri / 2;
                s_mix += mr * mr + mi * mi; s_ll_h += ll; s_rr_h += rrr;
            }
        }
        s_mix = block_sum_d(s_mix, sred); s_ll = block_sum_d(s_ll, sred); s_rr = block_sum_d(s_rr, sred);
        s_lr = block_sum_d(s_lr, sred); s_ll_h = block_sum_d(s_ll_h, sred); s_rr_h = block_sum_d(s_rr_h, sred);
        if (tid == 0) {
            const double P = sqrt(s_mix / dKh);
            double Q = p.Q[b] * p.lambda + p.one_minus_lambda * P;
            p.Q[b] = Q;
            bool temp = P < p.reject * Q, spat = false;
            if (p.alg == 0 || p.alg == 1) {
                const double num = s_lr / dK;
                double nc;
                if (num == 0) nc = 0;
                else { const double den = sqrt((s_ll / dK) * (s_rr / dK)); nc = den == 0 ? 1 : num / den; }
                spat = nc < p.thr[b];
                if (p.alg == 1) temp = false;
            }
            const int dec = spat ? 2 : (temp ? 1 : 0);
            double gL = 1, gR = 1;
            if (dec != 0) {
                if (p.method == 3) gL = gR = 1.0 / 1000;
                else if (p.method == 0) gL = gR = 1.0 / (double)(dec == 2 ? 10.f : 3.f);
                else if (p.method == 1) {
                    double fl = (s_ll / dK) * p.rho, fr = (s_rr / dK) * p.rho;
                    if (Q < 1e-10) { fl = p.rho; fr = p.rho; } else { fl /= Q; fr /= Q; }
                    gL = sqrt(fl); gR = sqrt(fr);
                } else if (p.method == 4 && p.first_call >= 2) {
                    const double pl = sqrt(s_ll_h / dKh), pr = sqrt(s_rr_h / dKh);
                    gL = pl > 0 ? p.noise[b] / pl : 1; gR = pr > 0 ? p.noise[b] / pr : 1;
                }
            }
            sg[0] = gL; sg[1] = gR;
            if (p.decisions) p.decisions[b] = dec;
        }
        __syncthreads();
        const double gL = sg[0], gR = sg[1];
        for (int k = tid; k < K; k += 256) {
            const double h = H[k];
            const double ml = k < Kh ? gL : 1.0, mr = k < Kh ? gR : 1.0;
            p.outL[2 * k] += p.L[2 * k] * h * ml; p.outL[2 * k + 1] += p.L[2 * k + 1] * h * ml;
            p.outR[2 * k] += p.R[2 * k] * h * mr; p.outR[2 * k + 1] += p.R[2 * k + 1] * h * mr;
        }
        __syncthreads();
    }
    if (tid < 45 && p.first_call == 0) p.noise[tid] = p.Q[tid];      // :193-197
}

}  // namespace mca
