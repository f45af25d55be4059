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
//   k_mvdr_analyse(_1024)   PCM -> windowed N-pt real FFT of the M channels -> X [stream][frame][bin][mic]
//   k_mvdr_solve     the recursion + factorisation above; four lanes per (stream, bin) problem, rows dealt cyclically
//   k_mvdr_synth     Y -> inverse FFT -> overlap-add
//
// The covariance (M(M+1)/2 complex per bin: 1.1 KB at M = 16, 0.56 MB per stream) never leaves the registers of
// its four lanes between frames of a call; HBM sees it once per call ("must stay in LDS/L2 ... or the
// path becomes state-traffic-bound").
#include "fft_block.h"
#include "mca_internal.h"

namespace mca {

// --------------------------------------------------------------------------------------
// k_mvdr_analyse: grid (frames, streams), 256 ... 1024 threads, LDS = M * (H + 1) float2
// --------------------------------------------------------------------------------------
// factored steering phasors of frame f of stream a (see MvdrAnalyseArgs::T), by the threads tid, tid + nthr, ...
__device__ __forceinline__ void mvdr_steering_tables(const MvdrAnalyseArgs &p, int a, int f, int tid, int nthr)
{
    const int nhi = (p.N >> 6) + 1, nph = nhi + 32;
    const long long o = (long long)a * p.n_frames + f;
    const double cd = cos((double)p.doa_rad[o] + 1.57079632679489661923);   // cos(DOA + M_PI/2), Beamformer.cpp:59
    float2 *T = p.T + o * p.M * nph;
    for (int e = tid; e < p.M * nph; e += nthr) {
        const int m = e / nph, i = e - m * nph;
        const int kk = i < nhi ? (i << 5) : i - nhi;
        double turns = (double)kk * (p.unit * p.mic_x[m] * cd);
        turns -= rint(turns);
        float sn, cs;
        sincospif(2.0f * (float)turns, &sn, &cs);
        T[e] = make_float2(cs, -sn);
    }
}

__global__ __launch_bounds__(1024) void k_mvdr_analyse(MvdrAnalyseArgs p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int M = p.M, logH = p.logH, H = 1 << logH, zs = H + 1, K = H + 1;
    float2 *xs = reinterpret_cast<float2 *>(smem_raw);                  // [M][H + 1]
    const int tid = threadIdx.x, NT = blockDim.x;
    const int a = blockIdx.y, f = blockIdx.x;
    const float *base = p.pcm + (long long)a * p.stream_stride;
    mvdr_steering_tables(p, a, f, tid, NT);
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
// k_mvdr_analyse_1024: the same for 1024-sample frames with the wave-level FFT of fft512.h (one wave per channel,
// eight channels per pass).  grid (ceil(frames / fpb), streams), 512 threads,
// LDS = 8 * MV_SP float2 (spectra of a pass) + TW_WORDS float2 (twiddles + half window).
// After a pass the eight spectra go out transposed: 64-byte runs (8 microphones of one bin) per 8 lanes.
// --------------------------------------------------------------------------------------
constexpr int MV_SP = 580;      // float2 words per spectrum: 8 words of bank offset between the channels of a pass

__global__ __launch_bounds__(512) void k_mvdr_analyse_1024(MvdrAnalyseArgs p, int fpb)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    constexpr int K = FFT_K;
    float2 *spec = reinterpret_cast<float2 *>(smem_raw);        // [8][MV_SP]
    float2 *tab = spec + 8 * MV_SP;                              // [TW_WORDS]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int a = blockIdx.y, M = p.M;
    const int f_begin = blockIdx.x * fpb, f_end = min(f_begin + fpb, p.n_frames);
    const float *base = p.pcm + (long long)a * p.stream_stride;

    fft_table_init(tab, p.window, tid, 512);
    for (int f = f_begin; f < f_end; ++f) mvdr_steering_tables(p, a, f, tid, 512);
    __syncthreads();
    FftTw tw{tab};
    for (int f = f_begin; f < f_end; ++f) {
        float2 *xo = p.X + ((long long)a * p.n_frames + f) * (long long)K * M;
        for (int c0 = 0; c0 < M; c0 += 8) {
            const int nc = min(8, M - c0);
            if (wave < nc) {
                const float2 *src = reinterpret_cast<const float2 *>(base + (long long)(c0 + wave) * p.mic_stride + (long long)f * FFT_H);
                float2 v[8];
#pragma unroll
                for (int r = 0; r < 8; ++r) { const float2 x = src[lane + 64 * r], w = tw.win(r, lane); v[r] = make_float2(x.x * w.x, x.y * w.y); }
                rfft1024(v, spec + wave * MV_SP, lane, tw);
            }
            __syncthreads();
            if (nc == 8) {
                for (int e = tid; e < K * 8; e += 512) {
                    const int k = e >> 3, m = e & 7;
                    xo[(long long)k * M + c0 + m] = spec[m * MV_SP + k];
                }
            } else {
                for (int e = tid; e < K * nc; e += 512) {
                    const int k = e / nc, m = e - k * nc;
                    xo[(long long)k * M + c0 + m] = spec[m * MV_SP + k];
                }
            }
            __syncthreads();
        }
    }
}

// --------------------------------------------------------------------------------------
// k_mvdr_solve<Q>: grid (ceil(streams * K / 64)), 256 threads.  FOUR lanes per (stream, bin) problem: lane l of a quad
// owns the rows l, l+4, l+8, ... of the lower triangles of Phi and L (Q = ceil(M/4) row slots, cyclic so that the
// lanes stay balanced as the factorisation shrinks), all in registers.  Frames are walked in order inside the kernel.
//
// Factorisation by columns (Cholesky-Banachiewicz).  In step j the row j of L, the pivot and the residuals of the
// two forward substitutions live in lane j%4; the other lanes of the quad get them with DPP quad-broadcast moves
// (full-rate VALU, no LDS, no synchronisation), every lane then updates the rows it owns below j.  A problem costs
// ~1/4 of the lane-instructions of a lane-per-row layout: the O(M^2) exchange is shared by up to four rows per lane
// and no lane idles while the active part of the matrix shrinks.
// --------------------------------------------------------------------------------------
template <int B>
__device__ __forceinline__ float quad_bcast1(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), B * 0x55, 0xf, 0xf, true));   // quad_perm:[B,B,B,B]
}
__device__ __forceinline__ float quad_bcast(float v, int b)
{
    switch (b) {
    case 0: return quad_bcast1<0>(v);
    case 1: return quad_bcast1<1>(v);
    case 2: return quad_bcast1<2>(v);
    default: return quad_bcast1<3>(v);
    }
}
__device__ __forceinline__ float2 quad_bcast(float2 v, int b) { return make_float2(quad_bcast(v.x, b), quad_bcast(v.y, b)); }

template <int Q, bool FULL>      // FULL: M == 4 Q, no partly empty row slot
__global__ __launch_bounds__(256, 2) void k_mvdr_solve(MvdrSolveArgs p)
{
    constexpr int NE = 2 * Q * (Q + 1);          // row slot q holds 4 (q + 1) entries, starting at 2 q (q + 1)
    const int tid = threadIdx.x, l = tid & 3;
    const int M = p.M, K = p.K, F = p.n_frames;
    const int piece = (int)(blockIdx.x % (unsigned)p.pieces);
    const int t_first = (int)((long long)piece * F / p.pieces), t_last = (int)((long long)(piece + 1) * F / p.pieces);   // frames this workgroup solves
    const long long total = p.pid0 + p.n_prob;
    const long long pid = p.pid0 + (long long)(blockIdx.x / (unsigned)p.pieces) * 64 + (tid >> 2);
    const bool pv = pid < total;
    const long long pc = pv ? pid : total - 1;   // surplus quads shadow the last problem and store nothing
    const int a = (int)(pc / K), k = (int)(pc - (long long)a * K);

    const int tri = M * (M + 1) / 2;
    float2 *st = p.phi + pc * tri;
    float2 P[NE], L[NE];
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        const int i = 4 * q + l;
#pragma unroll
        for (int m = 0; m < 4 * (q + 1); ++m)
            P[2 * q * (q + 1) + m] = (i < M && m <= i) ? st[i * (i + 1) / 2 + m] : make_float2(0.f, 0.f);
    }
    float tr = p.trace[pc];
    const int nhi = ((K - 1) >> 5) + 1, nph = nhi + 32;
    const float2 *T = p.T + (long long)a * F * M * nph + (k >> 5);         // + (t M + m) nph: hi factor; + nhi - (k >> 5) + (k & 31): lo
    const int lo_off = nhi - (k >> 5) + (k & 31);
    const long long fstride = (long long)K * M;
    const float2 *X = p.X + (long long)a * F * fstride + (long long)k * M + l;
    const float al = p.alpha, oma = p.one_minus_alpha;
    float2 *yo = p.Y + (long long)a * F * K + k;

    float2 xn[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) xn[q] = (FULL || 4 * q + l < M) ? X[4 * q] : make_float2(0.f, 0.f);
    for (int t = 0; t < t_last; ++t) {
        float2 x[Q], d[Q], rd[Q], rx[Q];
        float dsum[Q];
        // All loads of the frame go out together and without branches (the next frame's spectra: the last frame reloads its own), THEN
        // the steering products are formed.  With the load, the product and the conditional prefetch of one row slot after the
        // other, every slot waited for its own round trip to memory -- and for the "prefetch" issued just before it: four round
        // trips in a row at the top of every frame of every problem.
        float2 th[Q], tl[Q];
        const long long tn = (long long)min(t + 1, t_last - 1) * fstride;
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            const bool rv = FULL || 4 * q + l < M;
            // steering d_i = exp(-j k s_i), s_i = 2 pi fs/N/c x_i cos(DOA + pi/2), from the factored tables of the analysis
            const float2 *tq = T + ((long long)t * M + (rv ? 4 * q + l : 0)) * nph;
            th[q] = tq[0]; tl[q] = tq[lo_off];
            x[q] = xn[q];
        }
#pragma unroll
        for (int q = 0; q < Q; ++q) xn[q] = (FULL || 4 * q + l < M) ? X[tn + 4 * q] : make_float2(0.f, 0.f);
#pragma unroll
        for (int q = 0; q < Q; ++q) d[q] = (FULL || 4 * q + l < M) ? cmul(th[q], tl[q]) : make_float2(0.f, 0.f);
        // Phi <- alpha Phi + (1 - alpha) x x^H (the rows of this lane), tr <- alpha tr + (1 - alpha) |x|^2
        float e = 0.f;
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            const float2 xs = make_float2(oma * x[q].x, oma * x[q].y);
#pragma unroll
            for (int m = 0; m < 4 * (q + 1); ++m)
                if (FULL || m < M) {
                    const float2 xm = quad_bcast(x[m >> 2], m & 3);
                    float2 &e_ = P[2 * q * (q + 1) + m];
                    e_ = cmacc(make_float2(al * e_.x, al * e_.y), xs, xm);
                    if (q == Q - 1) e = fmaf(xm.x, xm.x, fmaf(xm.y, xm.y, e));
                }
        }
        tr = fmaf(al, tr, oma * e);
        if (t < t_first) continue;               // (an earlier piece solves this frame)
        const float delta = p.loading_over_m * tr;

        float2 num = make_float2(0.f, 0.f);
        float den = 0.f;
#pragma unroll
        for (int q = 0; q < Q; ++q) { rd[q] = d[q]; rx[q] = x[q]; dsum[q] = 0.f; }
#pragma unroll
        for (int j = 0; j < 4 * Q; ++j)
            if (FULL || j < M) {
                const int jq = j >> 2, jl = j & 3, jo = 2 * jq * (jq + 1);
                // pivot and the two substitution values of row j, from its owner
                const float inv = __builtin_amdgcn_rsqf(quad_bcast(P[jo + j].x + delta - dsum[jq], jl));
                float2 uj = quad_bcast(rd[jq], jl), vj = quad_bcast(rx[jq], jl);
                uj = make_float2(uj.x * inv, uj.y * inv);
                vj = make_float2(vj.x * inv, vj.y * inv);
                num = cmacc(num, vj, uj);                               // conj(u_j) v_j
                den = fmaf(uj.x, uj.x, fmaf(uj.y, uj.y, den));
                // L_ij = (Phi_ij - sum_{m<j} L_im conj(L_jm)) / L_jj for the rows below j (rows <= j compute dead values)
                float2 s[Q];
#pragma unroll
                for (int q = jq; q < Q; ++q) s[q] = P[2 * q * (q + 1) + j];
#pragma unroll
                for (int m = 0; m < j; ++m) {
                    const float2 r = quad_bcast(L[jo + m], jl);
#pragma unroll
                    for (int q = jq; q < Q; ++q) s[q] = cnmacc(s[q], L[2 * q * (q + 1) + m], r);
                }
#pragma unroll
                for (int q = jq; q < Q; ++q) {
                    const float2 lq = make_float2(s[q].x * inv, s[q].y * inv);
                    L[2 * q * (q + 1) + j] = lq;
                    dsum[q] = fmaf(lq.x, lq.x, fmaf(lq.y, lq.y, dsum[q]));
                    rd[q] = cnmac(rd[q], lq, uj);
                    rx[q] = cnmac(rx[q], lq, vj);
                }
            }
        const float rden = __builtin_amdgcn_rcpf(den);
        float2 y = make_float2(num.x * rden, num.y * rden);
        if (!(tr > 1e-30f)) {
            // digital silence so far: w = d/M, the reference's delay-and-sum (Beamformer.cpp:51-71)
            float2 acc = make_float2(0.f, 0.f);
#pragma unroll
            for (int q = 0; q < Q; ++q) acc = cmacc(acc, x[q], d[q]);   // conj(d_i) x_i
            acc.x += __shfl_xor(acc.x, 1, 4); acc.y += __shfl_xor(acc.y, 1, 4);
            acc.x += __shfl_xor(acc.x, 2, 4); acc.y += __shfl_xor(acc.y, 2, 4);
            y = make_float2(acc.x / (float)M, acc.y / (float)M);
        }
        if (l == 0 && pv) yo[(long long)t * K] = y;
    }
    if (pv && t_last == F) {
        float2 *so = p.phi_out + (pc - p.out_base) * tri;
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            const int i = 4 * q + l;
#pragma unroll
            for (int m = 0; m < 4 * (q + 1); ++m)
                if (i < M && m <= i) so[i * (i + 1) / 2 + m] = P[2 * q * (q + 1) + m];
        }
        if (l == 0) p.trace_out[pc - p.out_base] = tr;
    }
}

template __global__ void k_mvdr_solve<1, false>(MvdrSolveArgs);
template __global__ void k_mvdr_solve<2, false>(MvdrSolveArgs);
template __global__ void k_mvdr_solve<3, false>(MvdrSolveArgs);
template __global__ void k_mvdr_solve<4, false>(MvdrSolveArgs);
template __global__ void k_mvdr_solve<1, true>(MvdrSolveArgs);
template __global__ void k_mvdr_solve<2, true>(MvdrSolveArgs);
template __global__ void k_mvdr_solve<3, true>(MvdrSolveArgs);
template __global__ void k_mvdr_solve<4, true>(MvdrSolveArgs);

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
