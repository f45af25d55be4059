// kernels_generic.hip -- any-power-of-two frame length for the stream API (gfx950).
//
// The tuned kernels of kernels_stream.hip are built around the 1024-sample frame the reference's
// 44.1/48 kHz configuration uses (calculateOrderFromSampleRate).  Other sample rates give
// N = 256 ... 4096 (8 kHz -> 256, 16 kHz -> 512, 96 kHz -> 2048).  These two kernels cover them with
// the same data layout and the same arithmetic conventions: one workgroup per frame (analysis) or
// per run of frames (synthesis), all channels transformed together by a block-cooperative radix-2
// FFT in LDS with a twiddle table in global memory (L2 resident).  They feed the same SRP
// contraction, scan and gate kernels; only the transform is slower (one barrier per radix-2 stage).
//
//   k_stft_phat_gen   PCM -> windowed N-pt real FFT per channel -> PHAT -> pair / delay-group sums -> A
//   k_beamform_gen    PCM -> FFT -> delay-and-sum (Beamformer.cpp:51-71) -> inverse FFT -> overlap-add
#include "fft_block.h"
#include "mca_internal.h"

namespace mca {

// threads per workgroup of the any-length kernels: blockDim.x (256 ... 1024, chosen by the host per frame length)

// --------------------------------------------------------------------------------------
// k_stft_phat_gen: grid (frames of the chunk, arrays), 256 ... 1024 threads, LDS = M * (H + 1) float2 (+ 4 floats)
// --------------------------------------------------------------------------------------
template <typename OutT>
__global__ __launch_bounds__(1024) void k_stft_phat_gen(StftPhatArgs p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int M = p.M, logH = p.logH, H = 1 << logH, zs = H + 1;
    float2 *xs = reinterpret_cast<float2 *>(smem_raw);                  // [M][H + 1]
    float *spow = reinterpret_cast<float *>(xs + M * zs);               // [1]
    const int tid = threadIdx.x, lane = tid & 63, NT = blockDim.x;
    const int a = blockIdx.y, f = blockIdx.x;
    const float *base = p.pcm + (long long)a * p.array_stride;
    if (tid == 0) spow[0] = 0.f;

    load_frames(xs, zs, M, logH, base, p.mic_stride, (long long)p.frame0 + f, p.window, tid, NT);
    block_fft_dit(xs, zs, M, logH, p.tw, p.N, tid, NT);
    split_forward(xs, zs, M, logH, p.tw, tid, NT);

    if (p.power) {
        // dsp::SignalPower::FFTPower [INFERRED, SURVEY A.8]: (1/N^2) sum_k w_k |X[k]|^2, w = 2 except DC and Nyquist
        float acc = 0.f;
        for (int k = tid; k <= H; k += NT) {
            float s = 0.f;
            for (int m = 0; m < M; ++m) { const float2 z = xs[m * zs + k]; s += z.x * z.x + z.y * z.y; }
            acc += (k == 0 || k == H) ? s : 2.f * s;
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off);
        if (lane == 0) atomicAdd(spow, acc);
    }
    // PHAT whitening in place (each thread its own bins), then the pair products of those bins
    OutT *arow = reinterpret_cast<OutT *>(p.A) + ((long long)a * p.n_frames + f) * (long long)p.a_row_elems;
    const int kg = p.kg;
    for (int k = tid; k <= H; k += NT) {
        for (int m = 0; m < M; ++m) xs[m * zs + k] = whiten_g(xs[m * zs + k]);
        if (p.ula) {
            for (int g = 0; g < M - 1; ++g) {
                float2 acc = make_float2(0.f, 0.f);
                for (int i = 0; i + g + 1 < M; ++i) acc = cadd(acc, cmulc(xs[i * zs + k], xs[(i + g + 1) * zs + k]));
                store_a(arow, p, g * kg + k, acc);
            }
        } else {
            int pi = 0;
            for (int i = 0; i < M; ++i)
                for (int j = i + 1; j < M; ++j) { store_a(arow, p, pi * kg + k, cmulc(xs[i * zs + k], xs[j * zs + k])); ++pi; }
        }
    }
    if (p.power) {
        __syncthreads();
        if (tid == 0) p.power[(long long)a * p.total_frames + p.frame0 + f] = spow[0] / ((float)p.N * (float)p.N) / (float)M;
    }
}

template __global__ void k_stft_phat_gen<float>(StftPhatArgs);
template __global__ void k_stft_phat_gen<_Float16>(StftPhatArgs);

// --------------------------------------------------------------------------------------
// k_beamform_gen: grid (runs of ft frames, arrays), 256 ... 1024 threads,
// LDS = (M + S) * (H + 1) float2 + S * H floats (overlap-add carry) + (ft + 1) * S doubles
// --------------------------------------------------------------------------------------
// A run starts one frame early (tfirst = t0 - 1) to rebuild the overlap-add carry the previous run
// leaves; the very first run of a call takes it from tail_in and the last one leaves it in tail_out.
__global__ __launch_bounds__(1024) void k_beamform_gen(BeamformArgs p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int M = p.M, S = p.S, a = blockIdx.y, logH = p.logH, H = 1 << logH, zs = H + 1;
    float2 *xs = reinterpret_cast<float2 *>(smem_raw);                 // [M][H + 1]
    float2 *ys = xs + M * zs;                                           // [S][H + 1]
    float *carry = reinterpret_cast<float *>(ys + S * zs);             // [S][H]
    double *cdoa = reinterpret_cast<double *>(carry + S * H + ((S * H) & 1));   // [ft + 1][S]
    const int tid = threadIdx.x, NT = blockDim.x;
    const int t0 = blockIdx.x * p.ft;
    const int t1 = min(t0 + p.ft, p.n_frames);
    const int tfirst = t0 > 0 ? t0 - 1 : 0;

    // (S sources of this launch: numbers s0 ... s0 + S - 1 of the context's S_all -- one launch while M + S_all spectra fit the LDS)
    const int SA = p.S_all, s0 = p.s0;
    for (int e = tid; e < (t1 - tfirst) * S; e += NT) {
        const int r = e / S, s = e - r * S;
        const double doa = (double)p.doa_rad[((long long)a * p.n_frames + tfirst + r) * SA + s0 + s];
        cdoa[e] = cos(doa + 1.57079632679489661923);                    // cos(DOA + M_PI/2), Beamformer.cpp:59
    }
    for (int e = tid; e < S * H; e += NT) carry[e] = t0 == 0 ? p.tail_in[((long long)a * SA + s0) * H + e] : 0.f;
    __syncthreads();

    const float *base = p.pcm + (long long)a * p.array_stride;
    const double unit = (double)p.fs / (double)p.N / 346.1;            // Beamformer.cpp:59 without 2 pi
    const float inv = 1.0f / (float)M;
    const float sc = 1.0f / (float)H;

    for (int t = tfirst; t < t1; ++t) {
        load_frames(xs, zs, M, logH, base, p.mic_stride, (long long)t, p.window, tid, NT);
        block_fft_dit(xs, zs, M, logH, p.tw, p.N, tid, NT);
        split_forward(xs, zs, M, logH, p.tw, tid, NT);
        // delay-and-sum: Y[k] = (1/M) sum_c X_c[k] exp(j 2 pi k unit x_c cos(DOA + pi/2))
        for (int e = tid; e < S * (H + 1); e += NT) {
            const int s = e / (H + 1), k = e - s * (H + 1);
            const double cd = cdoa[(t - tfirst) * S + s];
            float2 acc = make_float2(0.f, 0.f);
            for (int c = 0; c < M; ++c) {
                double turns = (double)k * (unit * p.mic_x[c] * cd);
                turns -= rint(turns);
                float sn, cs;
                sincospif(2.0f * (float)turns, &sn, &cs);
                acc = cadd(acc, cmul(xs[c * zs + k], make_float2(cs, sn)));
            }
            ys[s * zs + k] = make_float2(acc.x * inv, acc.y * inv);     // divC :70
        }
        __syncthreads();
        // one-sided spectrum -> packed Z (imaginary parts of DC and Nyquist ignored, like a CCS inverse)
        for (int e = tid; e < S * (H / 2 + 1); e += NT) {
            const int s = e / (H / 2 + 1), k = e - s * (H / 2 + 1);
            float2 *yy = ys + s * zs;
            float2 xk = yy[k], xp = yy[H - k];
            if (k == 0) { xk.y = 0.f; xp.y = 0.f; }
            const float2 ev = make_float2(0.5f * (xk.x + xp.x), 0.5f * (xk.y - xp.y));
            const float2 df = make_float2(0.5f * (xk.x - xp.x), 0.5f * (xk.y + xp.y));
            const float2 od = cmulc(df, p.tw[k]);                        // conj(W_N^k) (X[k] - conj X[H-k]) / 2
            yy[k] = make_float2(ev.x - od.y, ev.y + od.x);               // E + j O
            if (k != 0 && k != H - k) yy[H - k] = make_float2(ev.x + od.y, -ev.y + od.x);   // conj(E) + j conj(O)
        }
        __syncthreads();
        block_ifft_dif(ys, zs, S, logH, p.tw, p.N, tid, NT);
        // overlap-add: y[2n], y[2n+1] = Re, Im of z[n] / H, z[n] stored at bitrev(n)
        for (int e = tid; e < S * (H / 2); e += NT) {
            const int s = e / (H / 2), n = e - s * (H / 2);              // sample pair (2n, 2n+1) of the hop
            const float2 lo = ys[s * zs + (int)(__brev((unsigned)n) >> (32 - logH))];
            const float2 hi = ys[s * zs + (int)(__brev((unsigned)(n + H / 2)) >> (32 - logH))];
            float *cr = carry + s * H + 2 * n;
            if (t >= t0) {
                float *o = p.out + ((long long)a * SA + s0 + s) * (long long)p.n_frames * H + (long long)t * H + 2 * n;
                o[0] = cr[0] + lo.x * sc; o[1] = cr[1] + lo.y * sc;
            }
            cr[0] = hi.x * sc; cr[1] = hi.y * sc;
        }
        __syncthreads();
    }
    if (t1 == p.n_frames)
        for (int e = tid; e < S * H; e += NT) p.tail_out[((long long)a * SA + s0) * H + e] = carry[e];
}

}  // namespace mca
