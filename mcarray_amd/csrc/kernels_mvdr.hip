// kernels_mvdr.hip -- MVDR-style frequency-domain beamformer with a per-bin spatial covariance (gfx950).
//
// BASELINE.json configs[3] (16 microphones, 256 concurrent streams).  [BUILD-DEFINES -- NO REFERENCE COUNTERPART]:
// the reference has delay-and-sum only (Beamformer.cpp:51-71); the spec is SURVEY A.9, with the steering vector and
// sign conventions of Beamformer.cpp:59 so that w = d/M is the reference's delay-and-sum.  Per stream and bin k:
//     Phi_t = alpha Phi_{t-1} + (1 - alpha) x x^H          (M x M Hermitian, Phi_{-1} = 0)
//     PhiL  = Phi_t + loading tr(Phi_t)/M I
//     Y[k]  = w^H x,  w = PhiL^-1 d / (d^H PhiL^-1 d),  d_m = exp(+j 2 pi k fs x_m sin(DOA) / (N c))
// evaluated through the Cholesky factor PhiL = L L^H: u = L^-1 d, v = L^-1 x, Y = (u^H v) / (u^H u) -- two
// forward substitutions, no back substitution, and only cond(L) = sqrt(cond(PhiL)) enters the fp32 error.
//
//   k_mvdr_analyse   PCM -> windowed N-pt real FFT of the M channels -> X [stream][frame][bin][mic]
//   k_mvdr_solve     the recursion + factorisation above; LP lanes per (stream, bin) problem, lane = matrix row
//   k_mvdr_synth     Y -> inverse FFT -> overlap-add
//
// The covariance (M(M+1)/2 complex per bin: 1.1 KB at M = 16, 0.56 MB per stream) never leaves the registers of
// its LP lanes between frames of a call; HBM sees it once per call ("must stay in LDS/L2 ... or the
// path becomes state-traffic-bound").
#include "fft_block.h"
#include "mca_internal.h"

namespace mca {

// --------------------------------------------------------------------------------------
// k_mvdr_analyse: grid (frames, streams), 256 ... 1024 threads, LDS = M * (H + 1) float2
// --------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void k_mvdr_analyse(MvdrAnalyseArgs p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int M = p.M, logH = p.logH, H = 1 << logH, zs = H + 1, K = H + 1;
    float2 *xs = reinterpret_cast<float2 *>(smem_raw);                  // [M][H + 1]
    const int tid = threadIdx.x, NT = blockDim.x;
    const int a = blockIdx.y, f = blockIdx.x;
    const float *base = p.pcm + (long long)a * p.stream_stride;
    if (tid == 0) {
        const long long o = (long long)a * p.n_frames + f;
        p.cdoa[o] = cos((double)p.doa_rad[o] + 1.57079632679489661923);     // cos(DOA + M_PI/2), Beamformer.cpp:59
    }
    load_frames(xs, zs, M, logH, base, p.mic_stride, (long long)f, p.window, tid, NT);
    block_fft_dit(xs, zs, M, logH, p.tw, p.N, tid, NT);
    split_forward(xs, zs, M, logH, p.tw, tid, NT);
    float2 *xo = p.X + ((long long)a * p.n_frames + f) * (long long)K * M;
    for (int e = tid; e < K * M; e += NT) {
        const int k = e / M, m = e - k * M;
        xo[e] = xs[m * zs + k];
    }
}

// --------------------------------------------------------------------------------------
// k_mvdr_solve<LP>: grid (ceil(K / (4 * 64/LP)), streams), 256 threads.  One (stream, bin) problem per group of LP
// lanes (M <= LP), lane i = row i of the lower triangle.  Frames are walked in order inside the kernel.
//
// Step j of the factorisation (Cholesky-Banachiewicz by columns): lane j owns the finished row j of L; it computes the
// pivot from its own registers, publishes row j, 1/L_jj, u_j and v_j in the group's LDS words, and every lane i > j
// reads them (broadcast reads) to form L_ij and to take L_ij u_j / L_ij v_j off its residuals.  u^H v and u^H u are
// accumulated by all lanes from the published values.  The groups of a wave run in lock step; the only
// synchronisation is the wave-level LDS fence between the write and the reads of a step.
// --------------------------------------------------------------------------------------
template <int LP>
__global__ __launch_bounds__(256) void k_mvdr_solve(MvdrSolveArgs p)
{
    constexpr int GPW = 64 / LP;                 // problems per wave
    constexpr int GS = 2 * LP + 4;               // float2 words per group: x[LP] | row[LP] | {1/L_jj, u_j, v_j, pad}
    __shared__ __attribute__((aligned(16))) float2 sm[4 * GPW * GS];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, grp = lane / LP, i = lane % LP;
    const int M = p.M, K = p.K, F = p.n_frames, a = blockIdx.y;
    const int k = (blockIdx.x * 4 + wave) * GPW + grp;
    const bool kv = k < K;
    const int kk = kv ? k : K - 1;               // surplus groups shadow the last bin and store nothing
    const bool rv = i < M;
    const int ii = rv ? i : 0;
    float2 *g = sm + (wave * GPW + grp) * GS;
    float2 *gx = g, *gr = g + LP, *ge = g + 2 * LP;

    const int tri = M * (M + 1) / 2;
    float2 *st = p.phi + ((long long)a * K + kk) * tri + ii * (ii + 1) / 2;
    float2 P[LP], Lr[LP];
#pragma unroll
    for (int m = 0; m < LP; ++m) P[m] = (rv && m <= i) ? st[m] : make_float2(0.f, 0.f);
    float tr = p.trace[(long long)a * K + kk];
    const double geo = p.unit * p.mic_x[ii];
    const double *cd = p.cdoa + (long long)a * F;
    const long long fstride = (long long)K * M;
    const float2 *X = p.X + (long long)a * F * fstride + (long long)kk * M + ii;
    const float al = p.alpha, oma = p.one_minus_alpha;
    float2 *yo = p.Y + (long long)a * F * K + kk;

    float2 xn = rv ? X[0] : make_float2(0.f, 0.f);
    for (int t = 0; t < F; ++t) {
        const float2 x = xn;
        if (t + 1 < F && rv) xn = X[(long long)(t + 1) * fstride];
        // steering d_i = exp(-j k s_i), s_i = 2 pi fs/N/c x_i cos(DOA + pi/2); the phase is reduced in double
        double turns = (double)kk * (geo * cd[t]);
        turns -= rint(turns);
        float sn, cs;
        sincospif(2.0f * (float)turns, &sn, &cs);
        const float2 d = make_float2(cs, -sn);

        gx[i] = x;
        wave_lds_fence();
        // Phi <- alpha Phi + (1 - alpha) x x^H (row i), tr <- alpha tr + (1 - alpha) |x|^2
        float e = 0.f;
        const float2 xs = make_float2(oma * x.x, oma * x.y);
#pragma unroll
        for (int m = 0; m < LP; ++m)
            if (m < M) {
                const float2 xm = gx[m];
                e = fmaf(xm.x, xm.x, fmaf(xm.y, xm.y, e));
                P[m] = cmacc(make_float2(al * P[m].x, al * P[m].y), xs, xm);
            }
        tr = fmaf(al, tr, oma * e);
        const float delta = p.loading_over_m * tr;

        float2 rd = d, rx = x, num = make_float2(0.f, 0.f);
        float den = 0.f;
#pragma unroll
        for (int j = 0; j < LP; ++j)
            if (j < M) {
                if (i == j) {
                    float s = P[j].x + delta;
#pragma unroll
                    for (int m = 0; m < j; ++m) s -= Lr[m].x * Lr[m].x + Lr[m].y * Lr[m].y;
                    const float inv = rsqrtf(s);
#pragma unroll
                    for (int m = 0; m < j; ++m) gr[m] = Lr[m];
                    ge[0] = make_float2(inv, 0.f);
                    ge[1] = make_float2(rd.x * inv, rd.y * inv);
                    ge[2] = make_float2(rx.x * inv, rx.y * inv);
                }
                wave_lds_fence();
                const float inv = ge[0].x;
                const float2 uj = ge[1], vj = ge[2];
                float2 s = P[j];
#pragma unroll
                for (int m = 0; m < j; ++m) s = cmacc(s, make_float2(-Lr[m].x, -Lr[m].y), gr[m]);
                const float2 l = make_float2(s.x * inv, s.y * inv);
                Lr[j] = l;
                rd = cmac(rd, make_float2(-l.x, -l.y), uj);
                rx = cmac(rx, make_float2(-l.x, -l.y), vj);
                num = cmacc(num, vj, uj);                               // conj(u_j) v_j
                den = fmaf(uj.x, uj.x, fmaf(uj.y, uj.y, den));
            }
        float2 y = make_float2(num.x / den, num.y / den);
        if (!(tr > 1e-30f)) {
            // digital silence so far: w = d/M, the reference's delay-and-sum (Beamformer.cpp:51-71)
            float2 q = rv ? cmulc(x, d) : make_float2(0.f, 0.f);       // conj(d_i) x_i
#pragma unroll
            for (int off = LP / 2; off > 0; off >>= 1) { q.x += __shfl_xor(q.x, off, LP); q.y += __shfl_xor(q.y, off, LP); }
            y = make_float2(q.x / (float)M, q.y / (float)M);
        }
        if (i == 0 && kv) yo[(long long)t * K] = y;
        wave_lds_fence();                                               // gx is rewritten by the next frame
    }
    if (kv && rv) {
#pragma unroll
        for (int m = 0; m < LP; ++m) if (m <= i) st[m] = P[m];
        if (i == 0) p.trace[(long long)a * K + kk] = tr;
    }
}

template __global__ void k_mvdr_solve<4>(MvdrSolveArgs);
template __global__ void k_mvdr_solve<8>(MvdrSolveArgs);
template __global__ void k_mvdr_solve<16>(MvdrSolveArgs);

// --------------------------------------------------------------------------------------
// k_mvdr_synth: grid (runs of ft frames, streams), 256 threads, LDS = (H + 1) float2 + H floats.
// A run starts one frame early to rebuild the overlap-add carry the previous run leaves (as k_beamform_gen).
// --------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void k_mvdr_synth(MvdrSynthArgs p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int a = blockIdx.y, logH = p.logH, H = 1 << logH, zs = H + 1, K = H + 1;
    float2 *ys = reinterpret_cast<float2 *>(smem_raw);                 // [H + 1]
    float *carry = reinterpret_cast<float *>(ys + zs);                 // [H]
    const int tid = threadIdx.x, NT = blockDim.x;
    const int t0 = blockIdx.x * p.ft;
    const int t1 = min(t0 + p.ft, p.n_frames);
    const int tfirst = t0 > 0 ? t0 - 1 : 0;
    for (int e = tid; e < H; e += NT) carry[e] = t0 == 0 ? p.tail_in[(long long)a * H + e] : 0.f;
    const float sc = 1.0f / (float)H;
    for (int t = tfirst; t < t1; ++t) {
        const float2 *Y = p.Y + ((long long)a * p.n_frames + t) * K;
        __syncthreads();
        // one-sided spectrum -> packed Z (imaginary parts of DC and Nyquist ignored, like a CCS inverse)
        for (int k = tid; k <= H / 2; k += NT) {
            float2 xk = Y[k], xp = Y[H - k];
            if (k == 0) { xk.y = 0.f; xp.y = 0.f; }
            const float2 ev = make_float2(0.5f * (xk.x + xp.x), 0.5f * (xk.y - xp.y));
            const float2 df = make_float2(0.5f * (xk.x - xp.x), 0.5f * (xk.y + xp.y));
            const float2 od = cmulc(df, p.tw[k]);
            ys[k] = make_float2(ev.x - od.y, ev.y + od.x);
            if (k != 0 && k != H - k) ys[H - k] = make_float2(ev.x + od.y, -ev.y + od.x);
        }
        __syncthreads();
        block_ifft_dif(ys, zs, 1, logH, p.tw, p.N, tid, NT);
        for (int n = tid; n < H / 2; n += NT) {
            const float2 lo = ys[(int)(__brev((unsigned)n) >> (32 - logH))];
            const float2 hi = ys[(int)(__brev((unsigned)(n + H / 2)) >> (32 - logH))];
            float *cr = carry + 2 * n;
            if (t >= t0) {
                float *o = p.out + (long long)a * p.n_frames * H + (long long)t * H + 2 * n;
                o[0] = cr[0] + lo.x * sc; o[1] = cr[1] + lo.y * sc;
            }
            cr[0] = hi.x * sc; cr[1] = hi.y * sc;
        }
    }
    __syncthreads();
    if (t1 == p.n_frames)
        for (int e = tid; e < H; e += NT) p.tail_out[(long long)a * H + e] = carry[e];
}

}  // namespace mca
