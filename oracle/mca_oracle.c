/*
 * mca_oracle.c -- CPU restatement of the mcarray hot path (see mca_oracle.h).
 * TEST INFRASTRUCTURE ONLY.  "parity unpinned" against an executed reference
 * (the reference cannot be built here); pinned by the reference's own test
 * properties, see header.
 *
 * Style: double precision, scalar loops, same loop nests as the reference
 * (one WIPP-call-equivalent loop per reference line).  File:line citations
 * point into /root/reference at the time of writing.
 */
#include "mca_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif
#ifndef M_PI_2
#define M_PI_2 1.57079632679489661923
#endif

/* ======================================================================= */
/* helpers: src/mcarray/microhponeArrayHelpers.cpp                          */
/* ======================================================================= */

double mca_or_speed_of_sound(void) { return 346.1; } /* :38-43 */

/* :46-67. `doa` and `microDist` are float parameters; the product
 * microDist*sin(doa) is evaluated in double ([BUILD-DEFINES]: with the
 * reference's CI toolchain, gcc-4.8 + <math.h>, ::sin is the C double
 * function), divided by the double speed of sound and narrowed to float. */
float mca_or_doa_to_delay_far_field(float doa, float microDist)
{
    float delay = (float)(((double)microDist * sin((double)doa)) / mca_or_speed_of_sound());
    return delay;
}

/* :69-72  float * int -> float multiply */
float mca_or_doa_to_delay_samples(float doa, float microDist, int fs)
{
    return mca_or_doa_to_delay_far_field(doa, microDist) * (float)fs;
}

/* :110-115 */
float mca_or_angle2doaidx(float angle, float step)
{
    double a = (double)angle;
    if (a < -M_PI_2) a = -M_PI_2;
    a = (double)(float)a;            /* `angle = std::max(...)` stores into a float */
    if (a > M_PI_2) a = M_PI_2;
    a = (double)(float)a;
    return (float)(int)((a + M_PI_2) / (double)step);
}

/* :117-120  float(idx)*step is a float multiply; the subtraction is double;
 * the return narrows to float. */
float mca_or_doaidx2angle(int idx, float step)
{
    float prod = (float)idx * step;
    return (float)((double)prod - M_PI_2);
}

float mca_or_doa_step(double step_deg) { return (float)(step_deg * M_PI / 180.0); } /* SteeringBeamforming.cpp:39 */
int   mca_or_num_steps(float step) { return (int)(round(M_PI / (double)step) + 1); } /* :40 */

/* ======================================================================= */
/* geometry: src/mcarray/ArrayDescription.cpp                               */
/* ======================================================================= */

double mca_or_distance(const double *xyz, int i, int j) /* :57-64 */
{
    return sqrt(pow(xyz[3 * j + 0] - xyz[3 * i + 0], 2) +
                pow(xyz[3 * j + 1] - xyz[3 * i + 1], 2) +
                pow(xyz[3 * j + 2] - xyz[3 * i + 2], 2));
}

double mca_or_max_distance(const double *xyz, int M) /* :77-91 */
{
    double maxd = 0;
    for (int i = 0; i < M; ++i)
        for (int j = 0; j < M; ++j) {
            if (i == j) continue;
            double d = mca_or_distance(xyz, i, j);
            if (d > maxd) maxd = d;
        }
    return maxd;
}

double mca_or_min_distance(const double *xyz, int M) /* :93-107: starts at 0, so always 0 */
{
    double mind = 0;
    for (int i = 0; i < M; ++i)
        for (int j = 0; j < M; ++j) {
            if (i == j) continue;
            double d = mca_or_distance(xyz, i, j);
            if (d < mind) mind = d;
        }
    return mind;
}

double mca_or_bandwidth(const double *xyz, int M) /* :294-301 */
{
    double maxD = mca_or_max_distance(xyz, M);
    if (maxD <= 0) return 0;
    return mca_or_speed_of_sound() / (2 * maxD);
}

/* ======================================================================= */
/* STFT engine [BUILD-DEFINES] -- SURVEY A.1                                */
/* ======================================================================= */

int mca_or_order_from_sample_rate(int fs, double frame_seconds)
{
    int order = (int)round(log2((double)fs * frame_seconds));
    if (order < 8) order = 8;
    if (order > 14) order = 14;
    return order;
}

void mca_or_hann_periodic(double *w, int N)
{
    for (int n = 0; n < N; ++n) w[n] = 0.5 - 0.5 * cos(2.0 * M_PI * (double)n / (double)N);
}

/* in-place iterative radix-2 complex FFT, sign = -1 forward, +1 inverse (unscaled) */
static void cfft(double *re, double *im, int n, int sign)
{
    for (int i = 1, j = 0; i < n; ++i) {
        int bit = n >> 1;
        for (; j & bit; bit >>= 1) j ^= bit;
        j ^= bit;
        if (i < j) {
            double t = re[i]; re[i] = re[j]; re[j] = t;
            t = im[i]; im[i] = im[j]; im[j] = t;
        }
    }
    for (int len = 2; len <= n; len <<= 1) {
        int half = len >> 1;
        for (int k = 0; k < half; ++k) {
            double ang = sign * 2.0 * M_PI * (double)k / (double)len;
            double wr = cos(ang), wi = sin(ang);
            for (int i = k; i < n; i += len) {
                int j = i + half;
                double xr = re[j] * wr - im[j] * wi;
                double xi = re[j] * wi + im[j] * wr;
                re[j] = re[i] - xr; im[j] = im[i] - xi;
                re[i] += xr;        im[i] += xi;
            }
        }
    }
}

void mca_or_rfft_ccs(const double *x, int N, double *ccs)
{
    double *re = (double *)malloc(sizeof(double) * 2 * (size_t)N);
    double *im = re + N;
    memcpy(re, x, sizeof(double) * (size_t)N);
    memset(im, 0, sizeof(double) * (size_t)N);
    cfft(re, im, N, -1);
    for (int k = 0; k <= N / 2; ++k) { ccs[2 * k] = re[k]; ccs[2 * k + 1] = im[k]; }
    free(re);
}

void mca_or_irfft_ccs(const double *ccs, int N, double *x)
{
    double *re = (double *)malloc(sizeof(double) * 2 * (size_t)N);
    double *im = re + N;
    for (int k = 0; k <= N / 2; ++k) { re[k] = ccs[2 * k]; im[k] = ccs[2 * k + 1]; }
    im[0] = 0; im[N / 2] = 0;           /* a real signal has real DC and Nyquist bins */
    for (int k = N / 2 + 1; k < N; ++k) { re[k] = re[N - k]; im[k] = -im[N - k]; }
    cfft(re, im, N, +1);
    for (int n = 0; n < N; ++n) x[n] = re[n] / (double)N;
    free(re);
}

void mca_or_stft_frame(const double *x, const double *win, int N, double *ccs)
{
    double *tmp = (double *)malloc(sizeof(double) * (size_t)N);
    for (int n = 0; n < N; ++n) tmp[n] = x[n] * win[n];
    mca_or_rfft_ccs(tmp, N, ccs);
    free(tmp);
}

/* ======================================================================= */
/* power: dsp::SignalPower [INFERRED] -- SURVEY A.8                         */
/* ======================================================================= */

/* mean over channels of (1/N^2) sum_k w_k |X[k]|^2, w=2 except DC/Nyquist
 * (Parseval: equals the time-domain mean square of the analysed frame).
 * Call sites BeamformingSeparationAndLocalisation.cpp:58,83. */
double mca_or_fft_power(const double *const *frames, int M, int ccs_len)
{
    int N = ccs_len - 2, K = ccs_len / 2;
    double acc = 0;
    for (int c = 0; c < M; ++c) {
        double s = 0;
        for (int k = 0; k < K; ++k) {
            double re = frames[c][2 * k], im = frames[c][2 * k + 1];
            double w = (k == 0 || k == K - 1) ? 1.0 : 2.0;
            s += w * (re * re + im * im);
        }
        acc += s / ((double)N * (double)N);
    }
    return acc / (double)M;
}

double mca_or_fft_log_power(const double *const *frames, int M, int ccs_len)
{
    return 10.0 * log10(mca_or_fft_power(frames, M, ccs_len));
}

/* 10*log10(mean x^2): pinned by the 70 dB expectation for a 5000-amplitude
 * tone at test/test_mcarray.cpp:943 */
double mca_or_log_power(const double *x, int n)
{
    double s = 0;
    for (int i = 0; i < n; ++i) s += x[i] * x[i];
    return 10.0 * log10(s / (double)n);
}

/* ======================================================================= */
/* GCC-PHAT at steering delays [INFERRED] -- SURVEY A.3                     */
/* ======================================================================= */

/* precomputeTauMatrix(tau, D, K, ONESIDEDFFT): T[d][k] = exp(+j 2 pi k tau_d / N),
 * N = 2 (K-1).  Call sites SteeringBeamforming.cpp:87-88, BinauralLocalisation.cpp:371. */
void mca_or_precompute_tau_matrix(const double *tau, int D, int K, double *T)
{
    double N = 2.0 * (double)(K - 1);
    for (int d = 0; d < D; ++d)
        for (int k = 0; k < K; ++k) {
            double ph = 2.0 * M_PI * (double)k * tau[d] / N;
            T[2 * ((size_t)d * K + k) + 0] = cos(ph);
            T[2 * ((size_t)d * K + k) + 1] = sin(ph);
        }
}

/* calculateCorrelationsForPrecomputedTauMatrix(A, B, out, K, D, ONESIDEDFFT):
 * G = A conj(B); Ghat = G / max(|G|, eps) (eps = 1e-30 [BUILD-DEFINES], a bin
 * with |G| = 0 contributes 0); out[d] = sum_k Ghat[k] T[d][k] (complex).
 * All K bins, unit weight.  Call sites SteeringBeamforming.cpp:115-119,
 * BinauralLocalisation.cpp:438-442. */
void mca_or_gcc_phat_tau_matrix(const double *A, const double *B, const double *T,
                                int K, int D, double *out)
{
    mca_or_gcc_tau_matrix(A, B, T, K, D, out, MCA_OR_GCC_PHAT);
}

/* The same steered sum with the weighting as a parameter: MCA_OR_GCC_PHAT (the build's [INFERRED] reading of
 * dsp::GeneralisedCrossCorrelation, SURVEY A.3, and what north_star names) or MCA_OR_GCC_NONE (the plain cross-spectrum
 * X_a conj X_b: the reading under which the reference's own sine test, test_mcarray.cpp:384-423, would pass -- DESIGN.md
 * section 2).  The deciding code is inside DSPONE; tools/pin_against_dspone.cpp settles it on a machine that has it. */
void mca_or_gcc_tau_matrix(const double *A, const double *B, const double *T,
                           int K, int D, double *out, int weighting)
{
    double *g = (double *)malloc(sizeof(double) * 2 * (size_t)K);
    for (int k = 0; k < K; ++k) {
        double ar = A[2 * k], ai = A[2 * k + 1], br = B[2 * k], bi = B[2 * k + 1];
        double gr = ar * br + ai * bi;
        double gi = ai * br - ar * bi;
        double mag = 1.0;
        if (weighting == MCA_OR_GCC_PHAT) {
            mag = sqrt(gr * gr + gi * gi);
            if (mag < 1e-30) mag = 1e-30;
        }
        g[2 * k] = gr / mag; g[2 * k + 1] = gi / mag;
    }
    for (int d = 0; d < D; ++d) {
        const double *t = T + 2 * (size_t)d * K;
        double sr = 0, si = 0;
        for (int k = 0; k < K; ++k) {
            double gr = g[2 * k], gi = g[2 * k + 1], tr = t[2 * k], ti = t[2 * k + 1];
            sr += gr * tr - gi * ti;
            si += gr * ti + gi * tr;
        }
        out[2 * d] = sr; out[2 * d + 1] = si;
    }
    free(g);
}

/* ======================================================================= */
/* SteeringBeamforming: src/mcarray/SteeringBeamforming.cpp                 */
/* ======================================================================= */

struct mca_or_steering {
    int fs, ccs_len, K, M, P, D;
    float step;
    int *pair;        /* P x 2  (_microPairIdx, :91) */
    double *delays;   /* P x D  (delaysForMicroPair, :69-73) */
    double *T;        /* P x D x K complex (one precomputed tau matrix per pair, :87-88) */
    double *corr;     /* P x D  (_correlations) */
    double *ccorr;    /* D complex (_complexCorrelation) */
    double *E, *Eprev;/* D (_energyInDOA, _prevEnergyInDOA) */
    int weighting;    /* MCA_OR_GCC_PHAT (default) / MCA_OR_GCC_NONE */
};

void mca_or_steering_set_weighting(mca_or_steering *s, int weighting) { s->weighting = weighting; }

mca_or_steering *mca_or_steering_create(int fs, const double *xyz, int M, int ccs_len, double step_deg)
{
    mca_or_steering *s = (mca_or_steering *)calloc(1, sizeof(*s));
    s->fs = fs; s->ccs_len = ccs_len; s->K = ccs_len / 2; s->M = M;   /* :34-41 */
    s->step = mca_or_doa_step(step_deg);
    s->D = mca_or_num_steps(s->step);
    s->P = M * (M - 1) / 2;
    s->pair = (int *)malloc(sizeof(int) * 2 * (size_t)s->P);
    s->delays = (double *)malloc(sizeof(double) * (size_t)s->P * s->D);
    s->T = (double *)malloc(sizeof(double) * 2 * (size_t)s->P * s->D * s->K);
    s->corr = (double *)malloc(sizeof(double) * (size_t)s->P * s->D);
    s->ccorr = (double *)malloc(sizeof(double) * 2 * (size_t)s->D);
    s->E = (double *)calloc((size_t)s->D, sizeof(double));
    s->Eprev = (double *)calloc((size_t)s->D, sizeof(double));   /* setZeros :51 */
    /* generateLookupTable :58-94 */
    int p = 0;
    for (int i = 0; i < M; ++i)
        for (int j = i + 1; j < M; ++j, ++p) {
            double distance = mca_or_distance(xyz, i, j);                       /* :67 */
            for (int d = 0; d < s->D; ++d)                                     /* :71-73 */
                s->delays[(size_t)p * s->D + d] =
                    (double)mca_or_doa_to_delay_samples(mca_or_doaidx2angle(d, s->step), (float)distance, fs);
            mca_or_precompute_tau_matrix(s->delays + (size_t)p * s->D, s->D, s->K,
                                         s->T + 2 * (size_t)p * s->D * s->K);   /* :87-88 */
            s->pair[2 * p] = i; s->pair[2 * p + 1] = j;                         /* :91 */
        }
    return s;
}

void mca_or_steering_destroy(mca_or_steering *s)
{
    if (!s) return;
    free(s->pair); free(s->delays); free(s->T); free(s->corr); free(s->ccorr); free(s->E); free(s->Eprev);
    free(s);
}

int mca_or_steering_num_steps(const mca_or_steering *s) { return s->D; }
int mca_or_steering_num_pairs(const mca_or_steering *s) { return s->P; }
const double *mca_or_steering_delays(const mca_or_steering *s, int pair) { return s->delays + (size_t)pair * s->D; }
void mca_or_steering_reset(mca_or_steering *s) { memset(s->Eprev, 0, sizeof(double) * (size_t)s->D); }

/* median_filter(src,dst,n,3) [BUILD-DEFINES edge rule: replicate the end samples] */
static void median3(const double *src, double *dst, int n)
{
    for (int i = 0; i < n; ++i) {
        double a = src[i > 0 ? i - 1 : 0], b = src[i], c = src[i < n - 1 ? i + 1 : n - 1];
        double lo = a < b ? a : b, hi = a < b ? b : a;
        dst[i] = c < lo ? lo : (c > hi ? hi : c);
    }
}

/* selectDOA :146-195 on an un-normalised energy vector E[D] */
void mca_or_select_doa(const double *Ein, int D, int numPairs, float step, int numOfSources,
                       double *DOA, double *prob, int *doa_bin)
{
    double *E = (double *)malloc(sizeof(double) * (size_t)D);
    double *fd = (double *)malloc(sizeof(double) * (size_t)D);
    double *ff = (double *)malloc(sizeof(double) * (size_t)D);
    double *sd = (double *)malloc(sizeof(double) * (size_t)D);
    const double minEnergyInDOA = -15 * numPairs;                      /* :151 */
    for (int i = 0; i < D; ++i) E[i] = Ein[i] - minEnergyInDOA;        /* subC :155 */
    for (int i = 0; i < D; ++i) E[i] = E[i] / (-2 * minEnergyInDOA);   /* divC :156 */
    for (int i = 0; i < D - 1; ++i) fd[i] = E[i + 1] - E[i];           /* sub(a,b,dst)=b-a :159 */
    for (int i = 0; i < D - 1; ++i)                                     /* threshold_lt_gt :161 */
        fd[i] = (fd[i] < 0.0) ? 1.0 : ((fd[i] > 0.0) ? 0.0 : fd[i]);
    median3(fd, ff, D - 1);                                             /* :164 */
    memcpy(fd, ff, sizeof(double) * (size_t)(D - 1));                   /* :165 */
    for (int i = 0; i < D - 2; ++i) sd[i] = fd[i + 1] - fd[i];          /* :170 */
    for (int i = 0; i < D - 2; ++i) sd[i] *= E[i + 1];                  /* :173 */
    for (int s = 0; s < numOfSources; ++s) {                            /* :185-194 */
        double max = sd[0]; int maxIdx = 0;                             /* maxidx: first max [BUILD-DEFINES] */
        for (int i = 1; i < D - 2; ++i) if (sd[i] > max) { max = sd[i]; maxIdx = i; }
        sd[maxIdx] = 0;
        if (DOA) DOA[s] = (double)mca_or_doaidx2angle(maxIdx + 1, step);
        if (prob) prob[s] = max;
        if (doa_bin) doa_bin[s] = maxIdx + 1;
    }
    free(E); free(fd); free(ff); free(sd);
}

/* Is the outcome of selectDOA on E pinned at all?  Returns 1 if moving any normalised energy by up to eps/2 (so any first
 * difference by up to eps) could change one of the numOfSources picks -- the cases a single-precision implementation may
 * legitimately resolve differently from this double-precision one:
 *   (a) two of the numOfSources + 1 largest second-derivative values lie within eps of each other (a tie between peaks);
 *   (b) a position whose second derivative reads a first difference within eps of zero (its sign, :161, could flip and with it
 *       the median filter's output, :164) holds an energy that reaches the last pick: a peak could appear or vanish there --
 *       at the map's edge too, far from the winning bin;
 *   (c) the last pick is a zero entry (fewer positive peaks than sources) while some candidate's energy is within eps of zero.
 * Test infrastructure (the classifier of GPU / oracle bin differences in tests/parity_helpers.py and tools/fuzz_parity.py);
 * it mirrors the sensitivity test that the adaptive SRP precision runs on the device (k_scan_pick, kernels_stream.hip). */
int mca_or_select_doa_fragile(const double *Ein, int D, int numPairs, int numOfSources, double eps)
{
    if (D < 3) return 0;
    double *E = (double *)malloc(sizeof(double) * (size_t)D);
    double *df = (double *)malloc(sizeof(double) * (size_t)D);
    double *fd = (double *)malloc(sizeof(double) * (size_t)D);
    double *ff = (double *)malloc(sizeof(double) * (size_t)D);
    double *sd = (double *)malloc(sizeof(double) * (size_t)D);
    const double minEnergyInDOA = -15 * numPairs;
    for (int i = 0; i < D; ++i) E[i] = (Ein[i] - minEnergyInDOA) / (-2 * minEnergyInDOA);
    for (int i = 0; i < D - 1; ++i) { df[i] = E[i + 1] - E[i]; fd[i] = (df[i] < 0.0) ? 1.0 : 0.0; }
    median3(fd, ff, D - 1);
    double umax = -1.0, zmin = 1e300;        /* largest |E| over the uncertain positions; smallest |E| over the candidates */
    int any_uncertain = 0;
    for (int i = 0; i < D - 2; ++i) {
        sd[i] = (ff[i + 1] - ff[i]) * E[i + 1];
        double dmin = 1e300;
        for (int c = -1; c <= 2; ++c) {      /* ff[i], ff[i + 1] read fd[i - 1 .. i + 2], edges replicated */
            int j = i + c; if (j < 0) j = 0; if (j > D - 2) j = D - 2;
            const double a = fabs(df[j]); if (a < dmin) dmin = a;
        }
        const double ae = fabs(E[i + 1]);
        if (dmin <= eps) { any_uncertain = 1; if (ae > umax) umax = ae; }
        if (ff[i + 1] != ff[i] && ae < zmin) zmin = ae;
    }
    int fragile = 0;
    double prev = 0.0, v_last = 0.0;
    for (int s = 0; s < numOfSources + 1; ++s) {
        double max = sd[0]; int maxIdx = 0;
        for (int i = 1; i < D - 2; ++i) if (sd[i] > max) { max = sd[i]; maxIdx = i; }
        sd[maxIdx] = 0;
        if (s > 0 && prev > 0.0 && prev - max <= eps) fragile = 1;                      /* (a) */
        prev = max;
        if (s < numOfSources) v_last = max;
    }
    if (v_last > 0.0 ? (any_uncertain && umax >= v_last - eps) : any_uncertain) fragile = 1;   /* (b) */
    if (v_last <= eps && zmin <= eps) fragile = 1;                                      /* (c) */
    free(E); free(df); free(fd); free(ff); free(sd);
    return fragile;
}

/* The same classifier with eps scaled by the LOCAL magnitude of what is compared (round 5): a first difference
 * En[j+1] - En[j] is "within eps of zero" if |df| <= eps_rel * max(1, |En[j]|, |En[j+1]|); two candidate values tie if they are
 * within eps_rel * max(1, |v1|, |v2|); a zero pick is open if a candidate lies within eps_rel (absolute: the values are near zero).
 * For rows with max |En| <= 1 this IS the absolute bar; where En is large (it reaches K / 30) an fp32 pipeline resolves
 * |En| * 6e-8, and only the comparisons that involve such values get the wider bar -- not every comparison of the row. */
int mca_or_select_doa_fragile_local(const double *Ein, int D, int numPairs, int numOfSources, double eps_rel)
{
    if (D < 3) return 0;
    double *E = (double *)malloc(sizeof(double) * (size_t)D);
    double *df = (double *)malloc(sizeof(double) * (size_t)D);
    double *de = (double *)malloc(sizeof(double) * (size_t)D);      /* the bar of df[j] */
    double *fd = (double *)malloc(sizeof(double) * (size_t)D);
    double *ff = (double *)malloc(sizeof(double) * (size_t)D);
    double *sd = (double *)malloc(sizeof(double) * (size_t)D);
    const double minEnergyInDOA = -15 * numPairs;
    for (int i = 0; i < D; ++i) E[i] = (Ein[i] - minEnergyInDOA) / (-2 * minEnergyInDOA);
    for (int i = 0; i < D - 1; ++i) {
        df[i] = E[i + 1] - E[i]; fd[i] = (df[i] < 0.0) ? 1.0 : 0.0;
        double m = fabs(E[i]) > fabs(E[i + 1]) ? fabs(E[i]) : fabs(E[i + 1]);
        de[i] = eps_rel * (m > 1.0 ? m : 1.0);
    }
    median3(fd, ff, D - 1);
    double zmin = 1e300;
    int n_unc = 0;
    double *uval = (double *)malloc(sizeof(double) * (size_t)D);    /* |E[i+1]| of the uncertain positions */
    for (int i = 0; i < D - 2; ++i) {
        sd[i] = (ff[i + 1] - ff[i]) * E[i + 1];
        int unc = 0;
        for (int c = -1; c <= 2; ++c) {
            int j = i + c; if (j < 0) j = 0; if (j > D - 2) j = D - 2;
            if (fabs(df[j]) <= de[j]) unc = 1;
        }
        const double ae = fabs(E[i + 1]);
        if (unc) uval[n_unc++] = ae;
        if (ff[i + 1] != ff[i] && ae < zmin) zmin = ae;
    }
    int fragile = 0;
    double prev = 0.0, v_last = 0.0;
    for (int s = 0; s < numOfSources + 1; ++s) {
        double max = sd[0]; int maxIdx = 0;
        for (int i = 1; i < D - 2; ++i) if (sd[i] > max) { max = sd[i]; maxIdx = i; }
        sd[maxIdx] = 0;
        double m = fabs(prev) > fabs(max) ? fabs(prev) : fabs(max);
        if (s > 0 && prev > 0.0 && prev - max <= eps_rel * (m > 1.0 ? m : 1.0)) fragile = 1;           /* (a) */
        prev = max;
        if (s < numOfSources) v_last = max;
    }
    for (int u = 0; u < n_unc; ++u) {                                                                   /* (b) */
        if (v_last > 0.0) {
            double m = uval[u] > v_last ? uval[u] : v_last;
            if (uval[u] >= v_last - eps_rel * (m > 1.0 ? m : 1.0)) fragile = 1;
        } else fragile = 1;
    }
    if (v_last <= eps_rel && zmin <= eps_rel) fragile = 1;                                              /* (c) */
    free(E); free(df); free(de); free(fd); free(ff); free(sd); free(uval);
    return fragile;
}

void mca_or_steering_process_frame(mca_or_steering *s, const double *const *frames,
                                   double *DOA, double *prob, int *doa_bin, int numOfSources,
                                   double *energy_out, double *corr_out)
{
    const int D = s->D, K = s->K, P = s->P;
    /* computeCorrelations :104-130 */
    for (int p = 0; p < P; ++p) {
        const double *A = frames[s->pair[2 * p]];        /* :110 */
        const double *B = frames[s->pair[2 * p + 1]];    /* :111 */
        mca_or_gcc_tau_matrix(A, B, s->T + 2 * (size_t)p * D * K, K, D, s->ccorr, s->weighting);   /* :115-119 */
        for (int d = 0; d < D; ++d) s->corr[(size_t)p * D + d] = s->ccorr[2 * d];           /* wipp::real :122 */
    }
    if (corr_out) {
        for (int d = 0; d < D; ++d) {
            double c = 0;
            for (int p = 0; p < P; ++p) c += s->corr[(size_t)p * D + d];
            corr_out[d] = c;
        }
    }
    /* computeEnergyInDOA :132-144.  _energyMemoryFactor is a float constant
     * (SteeringBeamforming.h:70); 1-_energyMemoryFactor is float arithmetic. */
    const float memf = 0.8f;
    const double mu = (double)memf;
    const double one_minus_mu = (double)(1 - memf);
    for (int d = 0; d < D; ++d) s->E[d] = mu * s->Eprev[d];                                 /* :134 */
    for (int p = 0; p < P; ++p) {
        for (int d = 0; d < D; ++d) s->corr[(size_t)p * D + d] *= one_minus_mu;             /* :139 */
        for (int d = 0; d < D; ++d) s->E[d] += s->corr[(size_t)p * D + d];                  /* :140 */
    }
    memcpy(s->Eprev, s->E, sizeof(double) * (size_t)D);                                     /* :143 */
    if (energy_out) memcpy(energy_out, s->E, sizeof(double) * (size_t)D);
    /* selectDOA :146-195 (normalises _energyInDOA in place; _prevEnergyInDOA keeps the raw values) */
    mca_or_select_doa(s->E, D, P, s->step, numOfSources, DOA, prob, doa_bin);
}

/* ======================================================================= */
/* Beamformer: src/mcarray/Beamformer.cpp:51-71                             */
/* ======================================================================= */

void mca_or_beamformer_process_frame(int fs, const double *xyz, int M, int ccs_len,
                                     const double *const *frames, double *out, double DOA)
{
    const int Kc = ccs_len / 2;
    memset(out, 0, sizeof(double) * (size_t)ccs_len);                                       /* :53 */
    for (int c = 0; c < M; ++c) {                                                            /* :56 */
        /* ramp(phase, n, offset 0, slope) :59 -- uses the x coordinate only */
        double slope = 2 * M_PI * fs / (ccs_len - 2) / mca_or_speed_of_sound() * xyz[3 * c] * cos(DOA + M_PI / 2);
        for (int k = 0; k < Kc; ++k) {
            double ph = 0.0 + slope * (double)k;
            double rr = 1.0 * cos(ph), ri = 1.0 * sin(ph);                                   /* polar2cart :60 */
            double xr = frames[c][2 * k], xi = frames[c][2 * k + 1];
            double yr = xr * rr - xi * ri, yi = xr * ri + xi * rr;                           /* mult :61-64 */
            out[2 * k] += yr; out[2 * k + 1] += yi;                                          /* add :65-67 */
        }
    }
    for (int i = 0; i < ccs_len; ++i) out[i] /= (double)M;                                   /* divC :70 */
}

/* The delay-and-sum stream of mcabeamf (src/programs/mcabeamf.cpp:77-122 feeds chunks to process(); the STFT engine calls the
 * hook once per frame): Beamformer::processFrame at the caller's angle for that frame, then the engine's synthesis. */
void mca_or_das_stream(int fs, int N, const double *xyz, int M, const double *pcm, long stride, int F,
                       const double *doa_rad, double *tail_io, double *out_pcm)
{
    const int hop = N / 2, ccs = N + 2;
    double *win = (double *)malloc(sizeof(double) * (size_t)N);
    mca_or_hann_periodic(win, N);
    double **fr = (double **)malloc(sizeof(double *) * (size_t)M);
    for (int c = 0; c < M; ++c) fr[c] = (double *)malloc(sizeof(double) * (size_t)ccs);
    double *Y = (double *)malloc(sizeof(double) * (size_t)ccs);
    double *y = (double *)malloc(sizeof(double) * (size_t)N);
    for (int t = 0; t < F; ++t) {
        for (int c = 0; c < M; ++c) mca_or_stft_frame(pcm + (size_t)c * stride + (size_t)t * hop, win, N, fr[c]);
        mca_or_beamformer_process_frame(fs, xyz, M, ccs, (const double *const *)fr, Y, doa_rad[t]);   /* Beamformer.cpp:51-71 */
        mca_or_irfft_ccs(Y, N, y);
        double *o = out_pcm + (size_t)t * hop;
        for (int n = 0; n < hop; ++n) { o[n] = tail_io[n] + y[n]; tail_io[n] = y[n + hop]; }
    }
    for (int c = 0; c < M; ++c) free(fr[c]);
    free(fr); free(win); free(Y); free(y);
}

/* ======================================================================= */
/* BeamformingSeparationAndLocalisation                                     */
/* ======================================================================= */

struct mca_or_bsl {
    int fs, ccs_len, M, S, use_floor;
    double *xyz;
    mca_or_steering *st;
    double *curDOA, *prob; int *curBin;
    double powerFloor; int noiseEstimated; int samplesConsumed;
    double **inFrames;
};

mca_or_bsl *mca_or_bsl_create(int fs, int ccs_len, const double *xyz, int M, int S, int use_floor, double step_deg)
{
    mca_or_bsl *b = (mca_or_bsl *)calloc(1, sizeof(*b));
    b->fs = fs; b->ccs_len = ccs_len; b->M = M; b->S = S; b->use_floor = use_floor;
    b->xyz = (double *)malloc(sizeof(double) * 3 * (size_t)M);
    memcpy(b->xyz, xyz, sizeof(double) * 3 * (size_t)M);
    b->st = mca_or_steering_create(fs, xyz, M, ccs_len, step_deg);
    b->curDOA = (double *)calloc((size_t)S, sizeof(double));            /* setZeros :51 */
    b->prob = (double *)malloc(sizeof(double) * (size_t)S);
    b->curBin = (int *)calloc((size_t)S, sizeof(int));
    for (int s = 0; s < S; ++s) b->prob[s] = -1.0;                      /* set(-1.0) :52 */
    b->inFrames = (double **)malloc(sizeof(double *) * (size_t)M);
    for (int c = 0; c < M; ++c) b->inFrames[c] = (double *)malloc(sizeof(double) * (size_t)ccs_len);
    return b;
}

void mca_or_bsl_destroy(mca_or_bsl *b)
{
    if (!b) return;
    for (int c = 0; c < b->M; ++c) free(b->inFrames[c]);
    free(b->inFrames); free(b->xyz); free(b->curDOA); free(b->prob); free(b->curBin);
    mca_or_steering_destroy(b->st); free(b);
}

void mca_or_bsl_set_weighting(mca_or_bsl *b, int weighting) { mca_or_steering_set_weighting(b->st, weighting); }
const double *mca_or_bsl_current_doa(const mca_or_bsl *b) { return b->curDOA; }
const int *mca_or_bsl_current_bin(const mca_or_bsl *b) { return b->curBin; }

/* setPowerFloor :55-72 */
static double bsl_set_power_floor(mca_or_bsl *b, const double *const *frames)
{
    const double durationToEstimatePowerFloor = 3;       /* SoundLocalisationImpl.h:77 */
    const double noiseMarginDB = 3;                      /* BeamformingSeparationAndLocalistaion.h:52 */
    int neededSamples = (int)(durationToEstimatePowerFloor * b->fs);
    double power = mca_or_fft_power(frames, b->M, b->ccs_len) * (b->ccs_len - 2);
    b->powerFloor += power;
    b->samplesConsumed += (b->ccs_len - 2);
    if (b->samplesConsumed >= neededSamples) {
        b->noiseEstimated = 1;
        b->powerFloor /= b->samplesConsumed;
        b->powerFloor = 10 * log10(b->powerFloor) + noiseMarginDB;
    }
    return b->powerFloor;
}

/* processFrameLocalisation :74-101 */
int mca_or_bsl_localise(mca_or_bsl *b, const double *const *frames, double *doa_deg, double *prob, double *power_out)
{
    double power;
    if (!b->noiseEstimated && b->use_floor) power = bsl_set_power_floor(b, frames);   /* :80-81 */
    else power = mca_or_fft_log_power(frames, b->M, b->ccs_len);                      /* :83 */
    if (power_out) *power_out = power;
    if ((power > b->powerFloor) || !b->use_floor) {                                   /* :87 */
        mca_or_steering_process_frame(b->st, frames, b->curDOA, b->prob, b->curBin, b->S, NULL, NULL); /* :89 */
        for (int s = 0; s < b->S; ++s) {
            if (doa_deg) doa_deg[s] = (180 / M_PI) * b->curDOA[s];                    /* toDegrees :93, helpers :91-98 */
            if (prob) prob[s] = b->prob[s];
        }
        return 1;
    }
    return 0;
}

/* processFrameSeparation :103-119 (in == out, as at SourceSeparationAndLocalisation.cpp:92) */
void mca_or_bsl_separate(mca_or_bsl *b, double *const *frames)
{
    int c;
    for (c = 0; c < b->M; ++c) memcpy(b->inFrames[c], frames[c], sizeof(double) * (size_t)b->ccs_len); /* :109-110 */
    int nout = b->M < b->S ? b->M : b->S;
    for (c = 0; c < nout; ++c)                                                                            /* :113-114 */
        mca_or_beamformer_process_frame(b->fs, b->xyz, b->M, b->ccs_len,
                                        (const double *const *)b->inFrames, frames[c], b->curDOA[c]);
    for (; c < b->M; ++c) memset(frames[c], 0, sizeof(double) * (size_t)b->ccs_len);                      /* :117-118 */
}

/* whole stream: SourceSeparationAndLocalisation::processParametrisation
 * (SourceSeparationAndLocalisation.cpp:79-94) once per STFT frame, inside the
 * [BUILD-DEFINES] STFT engine; usePowerFloor=false (as mcabeamf.cpp:194). */
void mca_or_ssl_stream(int fs, int N, const double *xyz, int M, int S, double step_deg,
                       const double *pcm, long stride, int F,
                       int *doa_bin, double *doa_rad, double *prob, double *out_pcm, double *energy_map)
{
    mca_or_ssl_stream_w(fs, N, xyz, M, S, step_deg, MCA_OR_GCC_PHAT, pcm, stride, F, doa_bin, doa_rad, prob, out_pcm, energy_map);
}

void mca_or_ssl_stream_w(int fs, int N, const double *xyz, int M, int S, double step_deg, int weighting,
                         const double *pcm, long stride, int F,
                         int *doa_bin, double *doa_rad, double *prob, double *out_pcm, double *energy_map)
{
    const int hop = N / 2, ccs = N + 2;
    mca_or_bsl *b = mca_or_bsl_create(fs, ccs, xyz, M, S, 0, step_deg);
    mca_or_bsl_set_weighting(b, weighting);
    const int D = b->st->D;
    double *win = (double *)malloc(sizeof(double) * (size_t)N);
    mca_or_hann_periodic(win, N);
    double **fr = (double **)malloc(sizeof(double *) * (size_t)M);
    for (int c = 0; c < M; ++c) fr[c] = (double *)malloc(sizeof(double) * (size_t)ccs);
    int nout = M < S ? M : S;
    double *tail = (double *)calloc((size_t)nout * hop, sizeof(double));
    double *y = (double *)malloc(sizeof(double) * (size_t)N);
    for (int t = 0; t < F; ++t) {
        for (int c = 0; c < M; ++c) mca_or_stft_frame(pcm + (size_t)c * stride + (size_t)t * hop, win, N, fr[c]);
        /* localisation (power gate disabled), keeping the map */
        mca_or_steering_process_frame(b->st, (const double *const *)fr, b->curDOA, b->prob, b->curBin, S,
                                      energy_map ? energy_map + (size_t)t * D : NULL, NULL);
        for (int s = 0; s < S; ++s) {
            if (doa_bin) doa_bin[(size_t)t * S + s] = b->curBin[s];
            if (doa_rad) doa_rad[(size_t)t * S + s] = b->curDOA[s];
            if (prob) prob[(size_t)t * S + s] = b->prob[s];
        }
        if (out_pcm) {
            mca_or_bsl_separate(b, fr);
            for (int s = 0; s < nout; ++s) {
                mca_or_irfft_ccs(fr[s], N, y);
                double *o = out_pcm + (size_t)s * F * hop + (size_t)t * hop;
                double *tl = tail + (size_t)s * hop;
                for (int n = 0; n < hop; ++n) { o[n] = tl[n] + y[n]; tl[n] = y[n + hop]; }
            }
        }
    }
    for (int c = 0; c < M; ++c) free(fr[c]);
    free(fr); free(win); free(tail); free(y);
    mca_or_bsl_destroy(b);
}

void mca_or_ssl_stream_gated(int fs, int N, const double *xyz, int M, int S, double step_deg, int use_floor,
                             const double *pcm, long stride, int F, int *doa_bin, double *doa_rad, double *prob,
                             double *out_pcm, double *energy_map, int *fired, double *power)
{
    const int hop = N / 2, ccs = N + 2;
    mca_or_bsl *b = mca_or_bsl_create(fs, ccs, xyz, M, S, use_floor, step_deg);
    const int D = b->st->D;
    double *win = (double *)malloc(sizeof(double) * (size_t)N);
    mca_or_hann_periodic(win, N);
    double **fr = (double **)malloc(sizeof(double *) * (size_t)M);
    for (int c = 0; c < M; ++c) fr[c] = (double *)malloc(sizeof(double) * (size_t)ccs);
    int nout = M < S ? M : S;
    double *tail = (double *)calloc((size_t)nout * hop, sizeof(double));
    double *y = (double *)malloc(sizeof(double) * (size_t)N);
    for (int s = 0; s < S; ++s) b->curBin[s] = -1;
    for (int t = 0; t < F; ++t) {
        for (int c = 0; c < M; ++c) mca_or_stft_frame(pcm + (size_t)c * stride + (size_t)t * hop, win, N, fr[c]);
        double pw = 0;
        int f = mca_or_bsl_localise(b, (const double *const *)fr, NULL, NULL, &pw);      /* :74-101 */
        if (fired) fired[t] = f;
        if (power) power[t] = pw;
        if (energy_map) memcpy(energy_map + (size_t)t * D, b->st->Eprev, sizeof(double) * (size_t)D);
        for (int s = 0; s < S; ++s) {
            if (doa_bin) doa_bin[(size_t)t * S + s] = b->curBin[s];
            if (doa_rad) doa_rad[(size_t)t * S + s] = b->curDOA[s];
            if (prob) prob[(size_t)t * S + s] = b->prob[s];
        }
        if (out_pcm) {
            mca_or_bsl_separate(b, fr);                                                  /* :103-119 */
            for (int s = 0; s < nout; ++s) {
                mca_or_irfft_ccs(fr[s], N, y);
                double *o = out_pcm + (size_t)s * F * hop + (size_t)t * hop;
                double *tl = tail + (size_t)s * hop;
                for (int n = 0; n < hop; ++n) { o[n] = tl[n] + y[n]; tl[n] = y[n + hop]; }
            }
        }
    }
    for (int c = 0; c < M; ++c) free(fr[c]);
    free(fr); free(win); free(tail); free(y);
    mca_or_bsl_destroy(b);
}

/* ======================================================================= */
/* FreqGCCBinauralLocalisation (deterministic part) -- SURVEY A.7           */
/* ======================================================================= */

struct mca_or_freqgcc {
    int fs, ccs_len, K, D, use_floor;
    float step;
    double micDist;
    double *delays, *T, *ccorr, *corr, *prev;
    float corrMem, doaMem;
    double curDOA, prob;
    double powerFloor; int noiseEstimated; int samplesConsumed;
    int silenceFramesCounter;                              /* :326 */
};

mca_or_freqgcc *mca_or_freqgcc_create(int fs, const double *xyz, int M, int ccs_len, int use_floor, double step_deg)
{
    (void)M;
    mca_or_freqgcc *g = (mca_or_freqgcc *)calloc(1, sizeof(*g));
    g->fs = fs; g->ccs_len = ccs_len; g->K = ccs_len / 2; g->use_floor = use_floor;
    g->step = mca_or_doa_step(step_deg);                   /* BinauralLocalisation.cpp:328 (3 deg default) */
    g->D = mca_or_num_steps(g->step);                      /* :329 */
    g->micDist = mca_or_distance(xyz, 0, 1);               /* :325 */
    g->corrMem = 0; g->doaMem = 0;                         /* :323-324 */
    g->silenceFramesCounter = 0;                           /* :326 */
    g->curDOA = 0; g->prob = -1;                           /* :339-340 */
    g->delays = (double *)malloc(sizeof(double) * (size_t)g->D);
    g->T = (double *)malloc(sizeof(double) * 2 * (size_t)g->D * g->K);
    g->ccorr = (double *)malloc(sizeof(double) * 2 * (size_t)g->D);
    g->corr = (double *)calloc((size_t)g->D, sizeof(double));
    g->prev = (double *)calloc((size_t)g->D, sizeof(double)); /* :353 */
    for (int i = 0; i < g->D; ++i)                         /* :363-367 */
        g->delays[i] = (double)mca_or_doa_to_delay_samples(mca_or_doaidx2angle(i, g->step), (float)g->micDist, fs);
    mca_or_precompute_tau_matrix(g->delays, g->D, g->K, g->T); /* :371 */
    return g;
}

void mca_or_freqgcc_destroy(mca_or_freqgcc *g)
{
    if (!g) return;
    free(g->delays); free(g->T); free(g->ccorr); free(g->corr); free(g->prev); free(g);
}

int mca_or_freqgcc_num_steps(const mca_or_freqgcc *g) { return g->D; }

/* processParametrisation :406-567, deterministic (#else) branch :502-504 */
int mca_or_freqgcc_process(mca_or_freqgcc *g, const double *left, const double *right,
                           double *corr_out, int *argmax_idx, double *doa_rad, double *power_out)
{
    const float maxCorrMem = 0.8f, maxDoaMem = 0.6f;       /* BinauralLocalisation.h:198-199 */
    const float noiseMarginDB = 6.0f;                      /* :197 */
    const double *fr[2] = { left, right };
    double power;
    if (!g->noiseEstimated) {                              /* setPowerFloor :387-404 */
        int needed = (int)(3.0 * g->fs);
        /* dsp::SignalPower::power [INFERRED] = mean over channels and samples of x^2 = FFTPower (Parseval) */
        double p = mca_or_fft_power(fr, 2, g->ccs_len) * (g->ccs_len - 2);
        g->powerFloor += p + 1e-10;
        g->samplesConsumed += (g->ccs_len - 2);
        if (g->samplesConsumed >= needed) {
            g->noiseEstimated = 1;
            if (g->samplesConsumed > 0) g->powerFloor /= g->samplesConsumed;
            g->powerFloor = 10 * log10(g->powerFloor) + (double)noiseMarginDB;
        }
        power = g->powerFloor;
    } else {
        power = mca_or_fft_log_power(fr, 2, g->ccs_len);   /* :432 */
    }
    if (power_out) *power_out = power;
    if (power > g->powerFloor || !g->use_floor) {          /* :434 */
        mca_or_gcc_phat_tau_matrix(left, right, g->T, g->K, g->D, g->ccorr);       /* :438-442 */
        for (int d = 0; d < g->D; ++d) g->corr[d] = g->ccorr[2 * d];                /* :444 */
        double a = (double)(1 - g->corrMem), bq = (double)g->corrMem;
        for (int d = 0; d < g->D; ++d) g->corr[d] *= a;                             /* :445 */
        for (int d = 0; d < g->D; ++d) g->prev[d] *= bq;                            /* :446 */
        for (int d = 0; d < g->D; ++d) g->corr[d] += g->prev[d];                    /* :447 */
        memcpy(g->prev, g->corr, sizeof(double) * (size_t)g->D);                    /* :448 */
        double max = g->corr[0]; int idx = 0;                                       /* :502 first max */
        for (int d = 1; d < g->D; ++d) if (g->corr[d] > max) { max = g->corr[d]; idx = d; }
        double DOA = (double)mca_or_doaidx2angle(idx, g->step);                     /* :503 */
        g->curDOA = (double)g->doaMem * g->curDOA + (double)(1 - g->doaMem) * DOA;  /* :504 */
        if (corr_out) memcpy(corr_out, g->corr, sizeof(double) * (size_t)g->D);
        if (argmax_idx) *argmax_idx = idx;
        if (doa_rad) *doa_rad = g->curDOA;
        g->corrMem = maxCorrMem; g->doaMem = maxDoaMem;                              /* :523-524 */
        g->silenceFramesCounter = 0;                                                 /* :525 */
        return 1;
    }
    if (g->noiseEstimated) {                                                         /* :530-560 */
        /* the memory factors stay at their maxima for 3 s of gated-out frames, then drop to zero: the first frame
           that fires after a longer silence restarts both recursions (corr = R_t, DOA = raw angle).  Note that the
           frame which completes the floor estimation already lands here, so with usePowerFloor the factors are at
           their maxima before any frame can fire (the stream's first fired frame is smoothed against the zero state). */
        const int secondsToDecay = 3;
        const int windowsToDecay = secondsToDecay * g->fs / (g->ccs_len / 2 - 1);    /* :533-534, int arithmetic */
        if (g->silenceFramesCounter < windowsToDecay) {                              /* :536 */
            g->corrMem = maxCorrMem; g->doaMem = maxDoaMem;                          /* :541-542 (the decay is commented out) */
        } else {
            g->corrMem = 0; g->doaMem = 0;                                           /* :556-557 */
        }
        ++g->silenceFramesCounter;                                                   /* :559 */
    }
    return 0;
}

/* setProbability :569-631 */
void mca_or_freqgcc_set_probability(const mca_or_freqgcc *g, const double *doas, double *probs, int size)
{
    const int D = g->D;
    double min = g->corr[0], sum = 0;
    for (int d = 0; d < D; ++d) { if (g->corr[d] < min) min = g->corr[d]; sum += g->corr[d]; }
    sum -= min * D;                                                                  /* :588 */
    double prevcorr = 0, nextcorr = 0, prevdoa = 0, nextdoa = 0;
    for (int i = 0; i < size; ++i) {
        int idx = (int)mca_or_angle2doaidx((float)doas[i], g->step);                 /* :598 */
        double angle = (double)mca_or_doaidx2angle(idx, g->step);
        double p;
        if (0 < idx && idx < (D - 1)) {
            if (angle > doas[i] && idx > 0) {
                prevcorr = g->corr[idx - 1]; prevdoa = (double)mca_or_doaidx2angle(idx - 1, g->step);
                nextcorr = g->corr[idx]; nextdoa = angle;
            } else if (angle <= doas[i] && idx < (D - 1)) {
                prevcorr = g->corr[idx]; prevdoa = angle;
                nextcorr = g->corr[idx + 1]; nextdoa = (double)mca_or_doaidx2angle(idx + 1, g->step);
            }
            double slope = (nextcorr - prevcorr) / (nextdoa - prevdoa);
            p = slope * (doas[i] - prevdoa) + prevcorr;
        } else {
            p = g->corr[idx];
        }
        probs[i] = 0;
        if (sum > 0) probs[i] = (p - min) / sum;
        probs[i] = (probs[i] < 0.01) ? 0 : probs[i];
    }
}

/* ======================================================================= */
/* FastBinauralMasking -- SURVEY A.6                                        */
/* ======================================================================= */

#define NBINS 45    /* FastBinauralMasking.h:111 */

struct mca_or_masking {
    int fs, N, K, method, alg, firstCall;
    double micDist;
    double *coefs;       /* NBINS x K real magnitude responses (the complex table has im = 0, :108) */
    double center[NBINS];
    double thr[NBINS];
    double Q[NBINS], noise[NBINS];
    double *L, *R, *outL, *outR;
};

static double hz2mel(double f) { return 2595.0 * log10(1.0 + f / 700.0); }
static double mel2hz(double m) { return 700.0 * (pow(10.0, m / 2595.0) - 1.0); }

/* [BUILD-DEFINES] stand-in for dsp::FilterBankFFTWMelScale(order, 45, fs, fmin, fmax):
 * nbins triangular filters, HTK-mel spaced edges between fmin and fmax, unit peak,
 * sampled on the K = N/2+1 FFT bin frequencies; getBinCenterFrequency(b) = centre / fs. */
void mca_or_mel_filterbank(int N, int nbins, int fs, double fmin, double fmax, double *coefs, double *center_cyc)
{
    const int K = N / 2 + 1;
    double mlo = hz2mel(fmin), mhi = hz2mel(fmax);
    double *edge = (double *)malloc(sizeof(double) * (size_t)(nbins + 2));
    for (int i = 0; i < nbins + 2; ++i) edge[i] = mel2hz(mlo + (mhi - mlo) * (double)i / (double)(nbins + 1));
    for (int b = 0; b < nbins; ++b) {
        double f0 = edge[b], f1 = edge[b + 1], f2 = edge[b + 2];
        center_cyc[b] = f1 / (double)fs;
        for (int k = 0; k < K; ++k) {
            double f = (double)k * (double)fs / (double)N, h = 0;
            if (f > f0 && f <= f1) h = (f - f0) / (f1 - f0);
            else if (f > f1 && f < f2) h = (f2 - f) / (f2 - f1);
            coefs[(size_t)b * K + k] = h;
        }
    }
    free(edge);
}

mca_or_masking *mca_or_masking_create(int fs, int N, double micDist, float lowFreq, float highFreq, int method, int alg)
{
    mca_or_masking *m = (mca_or_masking *)calloc(1, sizeof(*m));
    m->fs = fs; m->N = N; m->K = N / 2 + 1; m->method = method; m->alg = alg; m->micDist = micDist;
    m->coefs = (double *)malloc(sizeof(double) * NBINS * (size_t)m->K);
    mca_or_mel_filterbank(N, NBINS, fs, (double)lowFreq, (double)highFreq, m->coefs, m->center);  /* :95-98 */
    const double phi = 10 * M_PI / 180;                                            /* FastBinauralMasking.h:113 */
    for (int b = 0; b < NBINS; ++b) {                                              /* calculateThresholds :342-366 */
        double wfreq = m->center[b] * fs * 2 * M_PI;
        m->thr[b] = cos(wfreq * micDist * sin(phi) / mca_or_speed_of_sound());
    }
    m->L = (double *)malloc(sizeof(double) * (size_t)(N + 2));
    m->R = (double *)malloc(sizeof(double) * (size_t)(N + 2));
    m->outL = (double *)malloc(sizeof(double) * (size_t)(N + 2));
    m->outR = (double *)malloc(sizeof(double) * (size_t)(N + 2));
    return m;
}

void mca_or_masking_destroy(mca_or_masking *m)
{
    if (!m) return;
    free(m->coefs); free(m->L); free(m->R); free(m->outL); free(m->outR); free(m);
}

int mca_or_masking_nbins(void) { return NBINS; }
const double *mca_or_masking_thresholds(const mca_or_masking *m) { return m->thr; }
const double *mca_or_masking_filters(const mca_or_masking *m) { return m->coefs; }
const double *mca_or_masking_center_freqs(const mca_or_masking *m) { return m->center; }
const double *mca_or_masking_short_time_power(const mca_or_masking *m) { return m->Q; }

/* getPower :521-538: sqrt(mean_k |F[k]|^2) over length/2 complex bins */
static double mk_get_power(const double *frame, int length)
{
    int cl = length / 2;
    double s = 0;
    for (int k = 0; k < cl; ++k) {
        double mag = sqrt(frame[2 * k] * frame[2 * k] + frame[2 * k + 1] * frame[2 * k + 1]); /* magnitude */
        s += mag * mag;                                                                        /* sqr */
    }
    return sqrt(s / (double)cl);
}

/* getFramePower :496-517 */
static double mk_frame_power(const double *left, const double *right, int length, double *mixed)
{
    for (int i = 0; i < length; ++i) mixed[i] = left[i] / 2 + right[i] / 2;
    return mk_get_power(mixed, length);
}

/* normaliseFFTCorrelation :410-460 */
static double mk_norm_fft_corr(const double *left, const double *right, int K)
{
    double sr = 0;
    for (int k = 0; k < K; ++k) {        /* conj(left) * right, real part of the complex mean */
        double lr = left[2 * k], li = -left[2 * k + 1], rr = right[2 * k], ri = right[2 * k + 1];
        sr += rr * lr - ri * li;
    }
    double numer = sr / (double)K;
    if (numer == 0) return 0;
    double dl = 0, dr = 0;
    for (int k = 0; k < K; ++k) {
        double ml = sqrt(left[2 * k] * left[2 * k] + left[2 * k + 1] * left[2 * k + 1]);
        double mr = sqrt(right[2 * k] * right[2 * k] + right[2 * k + 1] * right[2 * k + 1]);
        dl += ml * ml; dr += mr * mr;
    }
    double denom = (dl / (double)K) * (dr / (double)K);
    denom = sqrt(denom);
    if (denom == 0) return 1;
    return numer / denom;
}

/* maskFrame :294-313 and the four methods :214-292 */
static void mk_mask_frame(mca_or_masking *m, double *frame, int length, float factor, int bin)
{
    const float scalingFactor = 0.01f;       /* FastBinauralMasking.h:124 */
    switch (m->method) {
    case MCA_OR_FULL:                        /* zeroFrame :214-217 */
        for (int i = 0; i < length; ++i) frame[i] /= 1000;
        break;
    case MCA_OR_RELATIVE: {                  /* maskFrameByScaling :245-287 */
        double f = 0;
        for (int k = 0; k < m->K; ++k) {
            double mag = sqrt(frame[2 * k] * frame[2 * k] + frame[2 * k + 1] * frame[2 * k + 1]);
            f += mag * mag;
        }
        f /= (double)m->K;
        f *= (double)scalingFactor;
        if (m->Q[bin] < 1e-10) f = (double)scalingFactor;
        else f /= m->Q[bin];
        f = sqrt(f);
        for (int i = 0; i < length; ++i) frame[i] *= f;
        break;
    }
    case MCA_OR_FACTOR:                      /* maskFrameByFactor :289-292 */
        for (int i = 0; i < length; ++i) frame[i] /= (double)factor;
        break;
    case MCA_OR_NOISY: {                     /* noisyFrame :219-243 */
        double powerBin = mk_get_power(frame, length);
        double f = 1;
        if (powerBin > 0) f = m->noise[bin] / powerBin;
        if (m->firstCall < 2) return;
        for (int i = 0; i < length; ++i) frame[i] *= f;
        break;
    }
    default: break;
    }
}

/* processParametrisation :126-210 */
void mca_or_masking_process(mca_or_masking *m, double *left, double *right, int *decisions)
{
    const int K = m->K, N = m->N, AL = N + 2;
    const float forgetingFactor = 0.04f, rejectTemporalFactor = 0.999f;    /* .h:114,128 */
    const float temporalMaskingFactor = 3, spatialMaskingFactor = 10, enhanceFactor = 1; /* .h:116-120 */
    if (m->method == MCA_OR_NOTHING) return;                                 /* :130-134 */
    memset(m->outL, 0, sizeof(double) * (size_t)AL);
    memset(m->outR, 0, sizeof(double) * (size_t)AL);
    double *mixed = (double *)malloc(sizeof(double) * (size_t)N);
    for (int bin = 0; bin < NBINS; ++bin) {                                  /* :146 */
        const double *H = m->coefs + (size_t)bin * K;
        for (int k = 0; k < K; ++k) {                                        /* complex mult by (H,0) :148-153 */
            m->L[2 * k] = left[2 * k] * H[k] - left[2 * k + 1] * 0.0;
            m->L[2 * k + 1] = left[2 * k] * 0.0 + left[2 * k + 1] * H[k];
            m->R[2 * k] = right[2 * k] * H[k] - right[2 * k + 1] * 0.0;
            m->R[2 * k + 1] = right[2 * k] * 0.0 + right[2 * k + 1] * H[k];
        }
        /* temportalMasking :477-493 (length = _windowSize = N) */
        double power = mk_frame_power(m->L, m->R, N, mixed);
        m->Q[bin] = m->Q[bin] * (double)forgetingFactor + (double)(1 - forgetingFactor) * power;
        int tempMask = power < (double)rejectTemporalFactor * m->Q[bin];
        int spatMask = 0;
        if (m->alg == MCA_OR_BOTH || m->alg == MCA_OR_SPATIAL) {             /* :159-166 */
            double ncorr = mk_norm_fft_corr(m->L, m->R, K);                  /* spatialMasking :369-376 */
            spatMask = ncorr < m->thr[bin];
            if (m->alg == MCA_OR_SPATIAL) tempMask = 0;
        }
        int dec;
        if (spatMask) {                                                      /* :168-174 */
            mk_mask_frame(m, m->L, N, spatialMaskingFactor, bin);
            mk_mask_frame(m, m->R, N, spatialMaskingFactor, bin);
            dec = 2;
        } else if (tempMask) {                                               /* :175-181 */
            mk_mask_frame(m, m->L, N, temporalMaskingFactor, bin);
            mk_mask_frame(m, m->R, N, temporalMaskingFactor, bin);
            dec = 1;
        } else {                                                             /* enhanceFrame :183-187, :315-318 */
            for (int i = 0; i < N; ++i) { m->L[i] *= (double)enhanceFactor; m->R[i] *= (double)enhanceFactor; }
            dec = 0;
        }
        if (decisions) decisions[bin] = dec;
        for (int i = 0; i < AL; ++i) { m->outL[i] += m->L[i]; m->outR[i] += m->R[i]; }   /* :189-190 */
    }
    ++m->firstCall;                                                           /* :192 */
    if (m->firstCall < 2) memcpy(m->noise, m->Q, sizeof(m->Q));               /* :193-197 */
    memcpy(left, m->outL, sizeof(double) * (size_t)AL);                       /* :199-200 */
    memcpy(right, m->outR, sizeof(double) * (size_t)AL);
    free(mixed);
}

void mca_or_masking_stream(mca_or_masking *m, const double *pl, const double *pr, int F, double *ol, double *orr)
{
    const int N = m->N, hop = N / 2;
    double *win = (double *)malloc(sizeof(double) * (size_t)N);
    mca_or_hann_periodic(win, N);
    double *L = (double *)malloc(sizeof(double) * (size_t)(N + 2));
    double *R = (double *)malloc(sizeof(double) * (size_t)(N + 2));
    double *y = (double *)malloc(sizeof(double) * (size_t)N);
    double *tl = (double *)calloc((size_t)hop * 2, sizeof(double));
    for (int t = 0; t < F; ++t) {
        mca_or_stft_frame(pl + (size_t)t * hop, win, N, L);
        mca_or_stft_frame(pr + (size_t)t * hop, win, N, R);
        mca_or_masking_process(m, L, R, NULL);
        mca_or_irfft_ccs(L, N, y);
        for (int n = 0; n < hop; ++n) { ol[(size_t)t * hop + n] = tl[n] + y[n]; tl[n] = y[n + hop]; }
        mca_or_irfft_ccs(R, N, y);
        for (int n = 0; n < hop; ++n) { orr[(size_t)t * hop + n] = tl[hop + n] + y[n]; tl[hop + n] = y[n + hop]; }
    }
    free(win); free(L); free(R); free(y); free(tl);
}

/* ======================================================================= */
/* MultibandBinarualLocalisation -- src/mcarray/MultibandBinarualLocalisation.cpp */
/* ======================================================================= */

/* [BUILD-DEFINES] stand-in for the LINEAR filter bank of dsp::SubBandSTFTAnalysis: nbins unit-peak triangles,
 * edges linearly spaced between fmin and fmax, sampled on the K = N/2+1 bin frequencies. */
void mca_or_linear_filterbank(int N, int nbins, int fs, double fmin, double fmax, double *coefs)
{
    const int K = N / 2 + 1;
    for (int b = 0; b < nbins; ++b) {
        double f0 = fmin + (fmax - fmin) * (double)b / (double)(nbins + 1);
        double f1 = fmin + (fmax - fmin) * (double)(b + 1) / (double)(nbins + 1);
        double f2 = fmin + (fmax - fmin) * (double)(b + 2) / (double)(nbins + 1);
        for (int k = 0; k < K; ++k) {
            double f = (double)k * (double)fs / (double)N, h = 0;
            if (f > f0 && f <= f1) h = (f - f0) / (f1 - f0);
            else if (f > f1 && f < f2) h = (f2 - f) / (f2 - f1);
            coefs[(size_t)b * K + k] = h;
        }
    }
}

struct mca_or_multiband {
    int fs, ccs_len, K, N, D, nbins, use_floor;
    float doaStep;
    double micDist;
    double *delays;            /* _samplesDelay[D] */
    double *coefs;             /* [nbins][K] */
    double *prev;              /* _prevCorrelationsReal [nbins][D] */
    double *bandL, *bandR;     /* sub-band frames */
    double curDOA, prob;       /* _currentDOA[0], _prob[0] */
    double powerFloor; int noiseEstimated, samplesConsumed;
};

/* the class's own doaIdx2angle (MultibandBinarualLocalisation.h:96-100): float result */
static float mb_idx2angle(int idx, float step) { return (float)((double)((float)idx * step) - M_PI_2); }

mca_or_multiband *mca_or_multiband_create(int fs, const double *xyz, int M, int ccs_len, int nbins, int use_floor)
{
    (void)M;
    mca_or_multiband *m = (mca_or_multiband *)calloc(1, sizeof(*m));
    m->fs = fs; m->ccs_len = ccs_len; m->K = ccs_len / 2; m->N = ccs_len - 2; m->nbins = nbins; m->use_floor = use_floor;
    m->micDist = mca_or_distance(xyz, 0, 1);                                   /* :61 */
    m->doaStep = (float)(5 * M_PI / 180);                                      /* :62 */
    m->D = (int)(floor(M_PI / (double)m->doaStep) + 1);                        /* :63 */
    /* maxFreqForSpatialAliasing takes and returns float (microhponeArrayHelpers.cpp:85-89) */
    float fmax = (float)(mca_or_speed_of_sound() / (double)(2 * (float)m->micDist));
    m->coefs = (double *)malloc(sizeof(double) * (size_t)nbins * (size_t)m->K);
    mca_or_linear_filterbank(m->N, nbins, fs, 100.0, (double)fmax, m->coefs);  /* :54-60 */
    m->delays = (double *)malloc(sizeof(double) * (size_t)m->D);
    for (int i = 0; i < m->D; ++i)                                             /* :101-104 */
        m->delays[i] = (double)mca_or_doa_to_delay_samples(mb_idx2angle(i, m->doaStep), (float)m->micDist, fs);
    m->prev = (double *)calloc((size_t)nbins * (size_t)m->D, sizeof(double)); /* :106-110 */
    m->bandL = (double *)malloc(sizeof(double) * (size_t)ccs_len);
    m->bandR = (double *)malloc(sizeof(double) * (size_t)ccs_len);
    m->curDOA = 0; m->prob = -1;                                               /* :81-82 */
    return m;
}

void mca_or_multiband_destroy(mca_or_multiband *m)
{
    if (!m) return;
    free(m->coefs); free(m->delays); free(m->prev); free(m->bandL); free(m->bandR); free(m);
}

int mca_or_multiband_num_steps(const mca_or_multiband *m) { return m->D; }
const double *mca_or_multiband_filters(const mca_or_multiband *m) { return m->coefs; }

/* setPowerFloor :125-143 -- called with analysisLength/2 (:216), so FFTPower sees the first half of the CCS
 * buffer and the sample count per frame is 2*(analysisLength/2) - 2 */
static double mb_set_power_floor(mca_or_multiband *m, const double *const *frames, int length)
{
    const double durationToEstimatePowerFloor = 3;       /* SoundLocalisationImpl.h:77 */
    const double noiseMarginDB = 3.0;                    /* MultibandBinarualLocalisation.h:47 */
    int neededSamples = (int)(durationToEstimatePowerFloor * m->fs);
    double power = mca_or_fft_power(frames, 2, length) * (2 * length - 2);
    m->powerFloor += power;
    m->samplesConsumed += (2 * length - 2);
    if (m->samplesConsumed >= neededSamples) {
        m->noiseEstimated = 1;
        m->powerFloor /= m->samplesConsumed;
        m->powerFloor = 10 * log10(m->powerFloor) + noiseMarginDB;
    }
    return m->powerFloor;
}

int mca_or_multiband_process(mca_or_multiband *m, const double *left, const double *right, int *band_idx,
                             double *band_energy, double *band_corr, double *energy_in_doa, double *doa_rad,
                             double *prob, double *power_out)
{
    const int D = m->D, K = m->K;
    const float corrMem = 0.4f;                                                /* _corrMemoryFactor .h:46 */
    const float doaMem = 0.f, doaMemSilence = 1.f;                             /* .h:44-45 */
    double *E = (double *)calloc((size_t)D, sizeof(double));                   /* processSetup :147 */
    for (int b = 0; b < m->nbins; ++b) {
        const double *h = m->coefs + (size_t)b * K;
        for (int k = 0; k < K; ++k) {
            m->bandL[2 * k] = left[2 * k] * h[k]; m->bandL[2 * k + 1] = left[2 * k + 1] * h[k];
            m->bandR[2 * k] = right[2 * k] * h[k]; m->bandR[2 * k + 1] = right[2 * k + 1] * h[k];
        }
        /* calculateCorrelationsForTauVector(left, right, out, K, delays, D, ONESIDEDFFT) :175-176 [INFERRED, A.3] */
        double *prev = m->prev + (size_t)b * D;
        int idx = 0; double mx = 0;
        for (int d = 0; d < D; ++d) {
            double re = 0;
            for (int k = 0; k < K; ++k) {
                double gr = m->bandL[2 * k] * m->bandR[2 * k] + m->bandL[2 * k + 1] * m->bandR[2 * k + 1];
                double gi = m->bandL[2 * k + 1] * m->bandR[2 * k] - m->bandL[2 * k] * m->bandR[2 * k + 1];
                double mag = sqrt(gr * gr + gi * gi);
                if (mag <= 1e-30) continue;
                double ph = 2.0 * M_PI * (double)k * m->delays[d] / (double)m->N;
                re += (gr * cos(ph) - gi * sin(ph)) / mag;
            }
            double c = re * (double)(1 - corrMem);                             /* :180 (float arithmetic on the constant) */
            prev[d] *= (double)corrMem;                                        /* :181 */
            c += prev[d];                                                      /* :182 */
            prev[d] = c;                                                       /* :183 */
            if (band_corr) band_corr[(size_t)b * D + d] = c;
            if (d == 0 || c > mx) { mx = c; idx = d; }                         /* maxidx :184, first maximum */
        }
        const double *fr[2] = {m->bandL, m->bandR};
        double e = mca_or_fft_power(fr, 2, m->ccs_len);                        /* :188 */
        E[idx] += e;                                                           /* :190 */
        if (band_idx) band_idx[b] = idx;
        if (band_energy) band_energy[b] = e;
    }
    /* processSumamry :198-258 */
    const double *frames[2] = {left, right};
    double power;
    if (!m->noiseEstimated) power = mb_set_power_floor(m, frames, m->ccs_len / 2);   /* :214-217 */
    else power = mca_or_fft_power(frames, 2, m->ccs_len);                     /* :221 (linear, compared with a dB floor) */
    int fired = 0;
    if (power > m->powerFloor || !m->use_floor) {                              /* :225 */
        double sum = 0, mx = 0; int idx = 0;
        for (int d = 0; d < D; ++d) sum += E[d];                               /* :227 */
        for (int d = 0; d < D; ++d) if (d == 0 || E[d] > mx) { mx = E[d]; idx = d; }   /* :228 */
        m->prob = sum;
        if (m->prob != 0) m->prob = E[idx] / m->prob;                          /* :230-233 */
        double DOA = (double)mb_idx2angle(idx, m->doaStep);                    /* :237 */
        m->curDOA = (double)doaMem * m->curDOA + (double)(1 - doaMem) * DOA;   /* :239 */
        fired = 1;
    } else {
        m->curDOA = m->curDOA * (double)doaMemSilence + (double)(1 - doaMemSilence) * 0;   /* :254 */
        m->prob = -100000;                                                     /* :255 */
    }
    if (energy_in_doa) memcpy(energy_in_doa, E, sizeof(double) * (size_t)D);
    if (doa_rad) *doa_rad = m->curDOA;
    if (prob) *prob = m->prob;
    if (power_out) *power_out = power;
    free(E);
    return fired;
}

/* ======================================================================= */
/* MVDR-style frequency-domain beamformer with a per-bin spatial covariance  */
/* (BASELINE.json configs[3]).  [BUILD-DEFINES -- NO REFERENCE COUNTERPART]: */
/* the reference has delay-and-sum only (Beamformer.cpp:51-71); the spec is  */
/* SURVEY A.9.  Steering vector and sign conventions follow Beamformer.cpp:59 */
/* so that w = d/M is exactly the reference's delay-and-sum.                  */
/* ======================================================================= */

struct mca_or_mvdr {
    int fs, N, K, M;
    double alpha, loading;
    double *x;          /* [M] microphone x coordinates (Beamformer.cpp:59 uses x only) */
    double *Phi;        /* [K][M][M] complex, row-major, re/im interleaved */
};

mca_or_mvdr *mca_or_mvdr_create(int fs, int N, const double *xyz, int M, double alpha, double loading)
{
    mca_or_mvdr *v = (mca_or_mvdr *)calloc(1, sizeof(*v));
    v->fs = fs; v->N = N; v->K = N / 2 + 1; v->M = M; v->alpha = alpha; v->loading = loading;
    v->x = (double *)malloc(sizeof(double) * (size_t)M);
    for (int m = 0; m < M; ++m) v->x[m] = xyz[3 * m];
    v->Phi = (double *)calloc((size_t)v->K * M * M * 2, sizeof(double));
    return v;
}

void mca_or_mvdr_destroy(mca_or_mvdr *v)
{
    if (!v) return;
    free(v->x); free(v->Phi); free(v);
}

void mca_or_mvdr_reset(mca_or_mvdr *v) { memset(v->Phi, 0, sizeof(double) * (size_t)v->K * v->M * v->M * 2); }

const double *mca_or_mvdr_covariance(const mca_or_mvdr *v) { return v->Phi; }

/* One frame.  frames[M] -> ccs double[N+2]; out ccs double[N+2]; DOA radians.
 * Per bin k:  Phi <- alpha Phi + (1-alpha) x x^H ;  PhiL = Phi + loading tr(Phi)/M I ;
 *             w = PhiL^-1 d / (d^H PhiL^-1 d) ;  Y[k] = w^H x
 * with d_m = exp(-j k s_m), s_m = 2 pi fs/N/c x_m cos(DOA + pi/2) (the conjugate of the phasor Beamformer.cpp:59-64
 * multiplies X_m by), i.e. d_m = exp(+j 2 pi k fs x_m sin(DOA) / (N c)).  Evaluated through the Cholesky factor
 * PhiL = L L^H:  u = L^-1 d, v = L^-1 x, Y = (u^H v) / (u^H u).  A bin whose covariance trace is <= 1e-30 (digital
 * silence so far) falls back to w = d/M, the reference's delay-and-sum. */
void mca_or_mvdr_process_frame(mca_or_mvdr *v, const double *const *frames, double *out, double DOA)
{
    const int M = v->M, K = v->K;
    double Lr[16 * 16], Li[16 * 16], dr[16], di[16], xr[16], xi[16], ur[16], ui[16], vr[16], vi[16];
    const double cd = cos(DOA + M_PI / 2);
    for (int k = 0; k < K; ++k) {
        double *P = v->Phi + (size_t)k * M * M * 2;
        for (int m = 0; m < M; ++m) {
            xr[m] = frames[m][2 * k]; xi[m] = frames[m][2 * k + 1];
            const double slope = 2 * M_PI * v->fs / v->N / mca_or_speed_of_sound() * v->x[m] * cd;   /* Beamformer.cpp:59 */
            const double ph = slope * (double)k;
            dr[m] = cos(ph); di[m] = -sin(ph);
        }
        double tr = 0;
        for (int i = 0; i < M; ++i)
            for (int j = 0; j < M; ++j) {
                /* x_i conj(x_j) */
                const double pr = xr[i] * xr[j] + xi[i] * xi[j], pi = xi[i] * xr[j] - xr[i] * xi[j];
                double *e = P + ((size_t)i * M + j) * 2;
                e[0] = v->alpha * e[0] + (1 - v->alpha) * pr;
                e[1] = v->alpha * e[1] + (1 - v->alpha) * pi;
                if (i == j) tr += e[0];
            }
        double yr, yi;
        if (!(tr > 1e-30)) {
            yr = yi = 0;
            for (int m = 0; m < M; ++m) { yr += dr[m] * xr[m] + di[m] * xi[m]; yi += dr[m] * xi[m] - di[m] * xr[m]; }
            yr /= M; yi /= M;
        } else {
            const double delta = v->loading * tr / M;
            /* Cholesky-Banachiewicz, lower triangle */
            for (int i = 0; i < M; ++i)
                for (int j = 0; j <= i; ++j) {
                    double sr = P[((size_t)i * M + j) * 2] + (i == j ? delta : 0.0), si = (i == j) ? 0.0 : P[((size_t)i * M + j) * 2 + 1];
                    for (int m = 0; m < j; ++m) {
                        /* L[i][m] conj(L[j][m]) */
                        sr -= Lr[i * 16 + m] * Lr[j * 16 + m] + Li[i * 16 + m] * Li[j * 16 + m];
                        si -= Li[i * 16 + m] * Lr[j * 16 + m] - Lr[i * 16 + m] * Li[j * 16 + m];
                    }
                    if (i == j) { Lr[i * 16 + j] = sqrt(sr); Li[i * 16 + j] = 0; }
                    else { Lr[i * 16 + j] = sr / Lr[j * 16 + j]; Li[i * 16 + j] = si / Lr[j * 16 + j]; }
                }
            /* forward substitutions u = L^-1 d, v = L^-1 x */
            for (int i = 0; i < M; ++i) {
                double ar = dr[i], ai = di[i], br = xr[i], bi = xi[i];
                for (int m = 0; m < i; ++m) {
                    ar -= Lr[i * 16 + m] * ur[m] - Li[i * 16 + m] * ui[m]; ai -= Lr[i * 16 + m] * ui[m] + Li[i * 16 + m] * ur[m];
                    br -= Lr[i * 16 + m] * vr[m] - Li[i * 16 + m] * vi[m]; bi -= Lr[i * 16 + m] * vi[m] + Li[i * 16 + m] * vr[m];
                }
                ur[i] = ar / Lr[i * 16 + i]; ui[i] = ai / Lr[i * 16 + i];
                vr[i] = br / Lr[i * 16 + i]; vi[i] = bi / Lr[i * 16 + i];
            }
            double nr = 0, ni = 0, den = 0;
            for (int m = 0; m < M; ++m) {
                nr += ur[m] * vr[m] + ui[m] * vi[m]; ni += ur[m] * vi[m] - ui[m] * vr[m];   /* conj(u) v */
                den += ur[m] * ur[m] + ui[m] * ui[m];
            }
            yr = nr / den; yi = ni / den;
        }
        out[2 * k] = yr; out[2 * k + 1] = yi;
    }
}

/* whole stream through the [BUILD-DEFINES] STFT engine: pcm M channels at pcm + c*stride,
 * (F+1)*hop samples; doa_rad[F] look direction per frame; out_pcm[F*hop]; out_spec (may be NULL) [F][N+2]. */
void mca_or_mvdr_stream(mca_or_mvdr *v, const double *pcm, long stride, int F, const double *doa_rad,
                        double *out_pcm, double *out_spec)
{
    const int N = v->N, hop = N / 2, ccs = N + 2, M = v->M;
    double *win = (double *)malloc(sizeof(double) * (size_t)N);
    mca_or_hann_periodic(win, N);
    double **fr = (double **)malloc(sizeof(double *) * (size_t)M);
    for (int c = 0; c < M; ++c) fr[c] = (double *)malloc(sizeof(double) * (size_t)ccs);
    double *Y = (double *)malloc(sizeof(double) * (size_t)ccs);
    double *y = (double *)malloc(sizeof(double) * (size_t)N);
    double *tail = (double *)calloc((size_t)hop, sizeof(double));
    for (int t = 0; t < F; ++t) {
        for (int c = 0; c < M; ++c) mca_or_stft_frame(pcm + (size_t)c * stride + (size_t)t * hop, win, N, fr[c]);
        mca_or_mvdr_process_frame(v, (const double *const *)fr, Y, doa_rad[t]);
        if (out_spec) memcpy(out_spec + (size_t)t * ccs, Y, sizeof(double) * (size_t)ccs);
        if (out_pcm) {
            mca_or_irfft_ccs(Y, N, y);
            for (int n = 0; n < hop; ++n) { out_pcm[(size_t)t * hop + n] = tail[n] + y[n]; tail[n] = y[n + hop]; }
        }
    }
    for (int c = 0; c < M; ++c) free(fr[c]);
    free(fr); free(win); free(Y); free(y); free(tail);
}
