// api_mask.hip -- C ABI of the binaural masking module (include/mcarray_hip.h, mca_hip_mask_*).
// Host side only: builds the mel filter bank and thresholds the reference builds in its
// constructor, owns the per-stream state, enqueues the kernels.  No CPU fallback.
#include "../../include/mcarray_hip.h"
#include "fft512.h"
#include "kernels.h"
#include "knobs.h"
#include "stage.h"
#include "state_blob.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

using namespace mca;

struct mca_hip_mask_ctx {
    mca_hip_mask_config cfg{};
    int N = 0, K = 0;
    std::vector<double> H, center, thr;       // [45][K], [45], [45]
    MaskParams hp{};
    MaskParams *d_mp = nullptr;
    float *d_window = nullptr;
    float2 *d_tw = nullptr, *d_kw = nullptr; int *d_kb = nullptr;   // any-length stream kernel
    int hop = 0, logH = 0;
    float *d_Q[2] = {nullptr, nullptr}, *d_noise = nullptr, *d_tail[2] = {nullptr, nullptr};
    int q_cur = 0, tail_cur = 0;
    long long frames_done = 0;
    int streams_in_use = 0;           // n_streams of the first stream call after create / reset: the context counts frames once for all streams
    // frame API (double)
    double *d_H = nullptr, *d_thr = nullptr, *d_Q64 = nullptr, *d_noise64 = nullptr, *d_io = nullptr;   // d_io: L, R, outL, outR
    int *d_dec = nullptr;
    int first_call = 0;
    StagePool stage;
    std::string err;
};

namespace {

std::string g_mask_create_error;

int mfail(mca_hip_mask_ctx *c, int code, const std::string &msg)
{
    if (c) c->err = msg; else g_mask_create_error = msg;
    return code;
}

#define MHIP_TRY(ctx, expr)                                                                             \
    do {                                                                                                \
        hipError_t _e = (expr);                                                                         \
        if (_e != hipSuccess)                                                                           \
            return mfail(ctx, _e == hipErrorOutOfMemory ? MCA_HIP_ERR_OUT_OF_MEMORY : MCA_HIP_ERR_HIP,  \
                         std::string(#expr) + ": " + hipGetErrorString(_e));                           \
    } while (0)

double hz2mel(double f) { return 2595.0 * std::log10(1.0 + f / 700.0); }
double mel2hz(double m) { return 700.0 * (std::pow(10.0, m / 2595.0) - 1.0); }

// [BUILD-DEFINES] stand-in for dsp::FilterBankFFTWMelScale(order, 45, fs, fmin, fmax) (FastBinauralMasking.cpp:95):
// 45 unit-peak triangles with HTK-mel spaced edges, sampled on the K = N/2+1 bin frequencies.
void mel_filterbank(int N, int nb, int fs, double fmin, double fmax, std::vector<double> &H, std::vector<double> &center)
{
    const int K = N / 2 + 1;
    std::vector<double> edge(nb + 2);
    const double mlo = hz2mel(fmin), mhi = hz2mel(fmax);
    for (int i = 0; i < nb + 2; ++i) edge[i] = mel2hz(mlo + (mhi - mlo) * (double)i / (double)(nb + 1));
    H.assign((size_t)nb * K, 0.0);
    center.resize(nb);
    for (int b = 0; b < nb; ++b) {
        const double f0 = edge[b], f1 = edge[b + 1], f2 = edge[b + 2];
        center[b] = f1 / (double)fs;
        for (int k = 0; k < K; ++k) {
            const double f = (double)k * (double)fs / (double)N;
            double h = 0;
            if (f > f0 && f <= f1) h = (f - f0) / (f1 - f0);
            else if (f > f1 && f < f2) h = (f2 - f) / (f2 - f1);
            H[(size_t)b * K + k] = h;
        }
    }
}

void free_mask(mca_hip_mask_ctx *c)
{
    if (!c) return;
    auto F = [](void *p) { if (p) (void)hipFree(p); };
    F(c->d_mp); F(c->d_window); F(c->d_tw); F(c->d_kw); F(c->d_kb); F(c->d_Q[0]); F(c->d_Q[1]); F(c->d_noise); F(c->d_tail[0]); F(c->d_tail[1]);
    F(c->d_H); F(c->d_thr); F(c->d_Q64); F(c->d_noise64); F(c->d_io); F(c->d_dec);
    c->stage.release();
    delete c;
}

}  // namespace

extern "C" {

const char *mca_hip_mask_last_error(const mca_hip_mask_ctx *ctx) { return ctx ? ctx->err.c_str() : g_mask_create_error.c_str(); }

int mca_hip_mask_create(const mca_hip_mask_config *cfg, mca_hip_mask_ctx **out)
{
    if (!cfg || !out) return mfail(nullptr, MCA_HIP_ERR_INVALID_ARGUMENT, "cfg/out is NULL");
    *out = nullptr;
    if (cfg->struct_size != (int)sizeof(mca_hip_mask_config)) return mfail(nullptr, MCA_HIP_ERR_INVALID_ARGUMENT, "struct_size mismatch");
    if (cfg->fft_size < 64 || (cfg->fft_size & (cfg->fft_size - 1)) || cfg->fft_size > 16384) return mfail(nullptr, MCA_HIP_ERR_INVALID_ARGUMENT, "fft_size must be a power of two in [64,16384]");
    if (cfg->sample_rate <= 0 || !(cfg->micro_distance > 0)) return mfail(nullptr, MCA_HIP_ERR_INVALID_ARGUMENT, "sample_rate / micro_distance must be positive");
    if (!(cfg->low_freq >= 0) || !(cfg->high_freq > cfg->low_freq) || cfg->high_freq > 0.5f * cfg->sample_rate) return mfail(nullptr, MCA_HIP_ERR_INVALID_ARGUMENT, "need 0 <= low_freq < high_freq <= fs/2");
    const int m = cfg->method;
    if (!(m == 0 || m == 1 || m == 3 || m == 4 || m == 5)) return mfail(nullptr, MCA_HIP_ERR_INVALID_ARGUMENT, "bad masking method");
    if (cfg->algorithm < 0 || cfg->algorithm > 2) return mfail(nullptr, MCA_HIP_ERR_INVALID_ARGUMENT, "bad masking algorithm");
    if (cfg->max_streams < 1) return mfail(nullptr, MCA_HIP_ERR_INVALID_ARGUMENT, "max_streams < 1");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return mfail(nullptr, MCA_HIP_ERR_NO_DEVICE, "no HIP device visible; libmcarray_hip has no CPU fallback");
    if (cfg->device < 0 || cfg->device >= ndev) return mfail(nullptr, MCA_HIP_ERR_INVALID_ARGUMENT, "device ordinal out of range");
    if (hipSetDevice(cfg->device) != hipSuccess) return mfail(nullptr, MCA_HIP_ERR_HIP, "hipSetDevice failed");

    mca_hip_mask_ctx *c = new mca_hip_mask_ctx();
    c->cfg = *cfg; c->N = cfg->fft_size; c->K = c->N / 2 + 1; c->hop = c->N / 2;
    while ((1 << c->logH) < c->hop) ++c->logH;
    mel_filterbank(c->N, 45, cfg->sample_rate, (double)cfg->low_freq, (double)cfg->high_freq, c->H, c->center);
    c->thr.resize(45);
    const double phi = 10 * M_PI / 180;                                           // FastBinauralMasking.h:113
    for (int b = 0; b < 45; ++b) {                                                // calculateThresholds :342-366
        const double wfreq = c->center[b] * cfg->sample_rate * 2 * M_PI;
        c->thr[b] = std::cos(wfreq * cfg->micro_distance * std::sin(phi) / 346.1);
    }
    const float lam = 0.04f, rej = 0.999f, rho = 0.01f;                           // FastBinauralMasking.h:114,128,124
    MaskParams &hp = c->hp;
    hp.lambda = lam; hp.one_minus_lambda = 1 - lam; hp.reject = rej; hp.rho = rho; hp.method = cfg->method; hp.alg = cfg->algorithm;
    // every bin is covered by at most two adjacent triangles: (first band, its weight, the next band's weight) per bin
    bool compact_ok = true;
    std::vector<int> kb(c->K, -1);
    std::vector<float2> kw(c->K, make_float2(0.f, 0.f));
    for (int b = 0; b < 45; ++b) { hp.thr[b] = (float)c->thr[b]; hp.lo[b] = 1; hp.hi[b] = 0; }
    for (int k = 0; k < c->K; ++k) {
        int nfound = 0;
        for (int b = 0; b < 45; ++b) {
            const double h = c->H[(size_t)b * c->K + k];
            if (h > 0) {
                if (nfound == 0) { kb[k] = b; kw[k].x = (float)h; }
                else if (nfound == 1 && b == kb[k] + 1) kw[k].y = (float)h;
                else compact_ok = false;
                ++nfound;
                if (hp.lo[b] > hp.hi[b]) hp.lo[b] = k;
                hp.hi[b] = k;
            }
        }
    }
    if (!compact_ok) { free_mask(c); return mfail(nullptr, MCA_HIP_ERR_UNSUPPORTED, "a bin is covered by more than two adjacent bands"); }
    if (c->N == FFT_N)
        for (int k = 0; k < c->K; ++k) { hp.kb[k] = kb[k]; hp.kw0[k] = kw[k].x; hp.kw1[k] = kw[k].y; }
    const size_t ns = (size_t)cfg->max_streams;
    std::vector<float> win(c->N);
    for (int n = 0; n < c->N; ++n) win[n] = (float)(0.5 - 0.5 * std::cos(2.0 * M_PI * n / c->N));
    std::vector<float2> tw(c->N / 2);
    for (int i = 0; i < c->N / 2; ++i) tw[i] = make_float2((float)std::cos(2.0 * M_PI * i / c->N), (float)(-std::sin(2.0 * M_PI * i / c->N)));
#define MUP(dst, src, bytes) do { MHIP_TRY(c, hipMalloc((void **)&(dst), (bytes))); MHIP_TRY(c, hipMemcpy((dst), (src), (bytes), hipMemcpyHostToDevice)); } while (0)
#define MZ(dst, bytes) do { MHIP_TRY(c, hipMalloc((void **)&(dst), (bytes))); MHIP_TRY(c, hipMemset((dst), 0, (bytes))); } while (0)
    auto body = [&]() -> int {
        MUP(c->d_mp, &c->hp, sizeof(MaskParams));
        MUP(c->d_window, win.data(), win.size() * 4);
        MUP(c->d_tw, tw.data(), tw.size() * 8); MUP(c->d_kw, kw.data(), kw.size() * 8); MUP(c->d_kb, kb.data(), kb.size() * 4);
        MZ(c->d_Q[0], ns * 45 * 4); MZ(c->d_Q[1], ns * 45 * 4); MZ(c->d_noise, ns * 45 * 4);
        MZ(c->d_tail[0], ns * 2 * c->hop * 4); MZ(c->d_tail[1], ns * 2 * c->hop * 4);
        MUP(c->d_H, c->H.data(), c->H.size() * 8);
        MUP(c->d_thr, c->thr.data(), 45 * 8);
        MZ(c->d_Q64, 45 * 8); MZ(c->d_noise64, 45 * 8);
        MZ(c->d_io, (size_t)4 * (c->N + 2) * 8);
        MZ(c->d_dec, 45 * 4);
        return MCA_HIP_OK;
    };
    int rc = body();
#undef MUP
#undef MZ
    if (rc) { g_mask_create_error = c->err; free_mask(c); return rc; }
    *out = c;
    return MCA_HIP_OK;
}

void mca_hip_mask_destroy(mca_hip_mask_ctx *c)
{
    if (!c) return;
    (void)hipSetDevice(c->cfg.device);
    (void)hipDeviceSynchronize();
    free_mask(c);
}

int mca_hip_mask_reset(mca_hip_mask_ctx *c)
{
    if (!c) return MCA_HIP_ERR_INVALID_ARGUMENT;
    MHIP_TRY(c, hipSetDevice(c->cfg.device));
    const size_t ns = (size_t)c->cfg.max_streams;
    for (int i = 0; i < 2; ++i) { MHIP_TRY(c, hipMemset(c->d_Q[i], 0, ns * 45 * 4)); MHIP_TRY(c, hipMemset(c->d_tail[i], 0, ns * 2 * c->hop * 4)); }
    MHIP_TRY(c, hipMemset(c->d_noise, 0, ns * 45 * 4));
    MHIP_TRY(c, hipMemset(c->d_Q64, 0, 45 * 8)); MHIP_TRY(c, hipMemset(c->d_noise64, 0, 45 * 8));
    c->frames_done = 0; c->first_call = 0; c->streams_in_use = 0;
    return MCA_HIP_OK;
}

int mca_hip_mask_get_thresholds(const mca_hip_mask_ctx *c, double *thresholds, double *center_freqs)
{
    if (!c) return MCA_HIP_ERR_INVALID_ARGUMENT;
    if (thresholds) std::memcpy(thresholds, c->thr.data(), 45 * 8);
    if (center_freqs) std::memcpy(center_freqs, c->center.data(), 45 * 8);
    return MCA_HIP_OK;
}

int mca_hip_mask_frames_dev(mca_hip_mask_ctx *c, const float *pcm, long long stream_stride, long long ch_stride,
                            int n_streams, int n_frames, float *out_pcm, int *decisions, void *stream)
{
    if (!c) return MCA_HIP_ERR_INVALID_ARGUMENT;
    if (c->N > 8192) return mfail(c, MCA_HIP_ERR_UNSUPPORTED, "the stream API takes frame lengths up to 8192 (the frame hook takes any size)");
    if (!pcm || !out_pcm) return mfail(c, MCA_HIP_ERR_INVALID_ARGUMENT, "pcm_dev/out_pcm_dev is NULL");
    if (n_streams < 1 || n_streams > c->cfg.max_streams) return mfail(c, MCA_HIP_ERR_INVALID_ARGUMENT, "n_streams outside [1, max_streams]");
    if (n_frames < 1) return mfail(c, MCA_HIP_ERR_INVALID_ARGUMENT, "n_frames < 1");
    // frames_done (first-frame behaviour of the Q recursion, NOISY's noise capture: FastBinauralMasking.cpp:186-197) is kept once
    // per context, so all streams of a context start together and advance together: a call with another number of streams
    // would give the late or missing slots the not-first-frame path
    if (c->streams_in_use && n_streams != c->streams_in_use)
        return mfail(c, MCA_HIP_ERR_INVALID_ARGUMENT, "n_streams differs from the first call after create / reset: the streams of a masking context advance together (mca_hip_mask_reset starts over)");
    const long long need = (long long)(n_frames + 1) * c->hop;
    if (ch_stride < need || (n_streams > 1 && stream_stride < ch_stride + need)) return mfail(c, MCA_HIP_ERR_INVALID_ARGUMENT, "strides shorter than (n_frames+1)*hop samples");
    if ((ch_stride & 1) || (stream_stride & 1) || (reinterpret_cast<uintptr_t>(pcm) & 7)) return mfail(c, MCA_HIP_ERR_INVALID_ARGUMENT, "pcm_dev must be 8-byte aligned with even strides");
    hipStream_t st = (hipStream_t)stream;
    MaskArgs a{};
    a.pcm = pcm; a.stream_stride = stream_stride; a.ch_stride = ch_stride; a.n_frames = n_frames;
    // NOISY takes its noise estimate from the stream's very first frame (:193-197): that one call runs as a single chunk
    // otherwise runs of up to 256 frames (every run re-analyses 9 frames for the Q warm-up and the overlap-add carry),
    // shorter ones for small batches so that two workgroups per CU exist
    int ft = 256;
    while (ft > 16 && (long long)n_streams * ((n_frames + ft - 1) / ft) < 512) ft >>= 1;
    a.ft = (c->cfg.method == MCA_HIP_MASK_NOISY && c->frames_done == 0) ? n_frames : ft;
    a.frames_done = c->frames_done; a.window = c->d_window; a.mp = c->d_mp;
    a.Q_in = c->d_Q[c->q_cur]; a.Q_out = c->d_Q[c->q_cur ^ 1]; a.noise = c->d_noise;
    a.tail_in = c->d_tail[c->tail_cur]; a.tail_out = c->d_tail[c->tail_cur ^ 1];
    a.out = out_pcm; a.decisions = decisions;
    static const bool no_tuned = mca::measure_env("MCA_HIP_MASK_GENERIC") != nullptr;      // A/B switch for measurements
    if (c->N == FFT_N && !no_tuned) {
        // 79 KiB: two workgroups per CU
        const size_t smem = (size_t)MK_NB * 2 * FFT_SCRATCH * sizeof(float2) + (size_t)MK_NB * 3 * 520 * sizeof(float) +
                            TW_WIN * sizeof(float2) + (size_t)MK_NB * 48 * 8 * sizeof(float) + 520 * sizeof(float2) + 528;
        if (smem > 64 * 1024)
            MHIP_TRY(c, hipFuncSetAttribute(reinterpret_cast<const void *>(k_mask_stream), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        dim3 g((n_frames + a.ft - 1) / a.ft, n_streams);
        hipLaunchKernelGGL(k_mask_stream, g, dim3(512), smem, st, a);
    } else if (c->N == 2048 && !no_tuned) {
        // 2048-sample frames (44.1 / 48 kHz): four 512-sample sub-sequences per frame on the wave-level transform
        MaskGenArgs ga{};
        ga.a = a; ga.N = c->N; ga.logH = c->logH; ga.tw = c->d_tw; ga.kw = c->d_kw; ga.kb = c->d_kb;
        if (!(c->cfg.method == MCA_HIP_MASK_NOISY && c->frames_done == 0)) {
            ga.a.ft = 256;
            while (ga.a.ft > 16 && (long long)n_streams * ((n_frames + ga.a.ft - 1) / ga.a.ft) < 512) ga.a.ft >>= 1;
        }
        const size_t smem = (size_t)(16 * 258 + 8 * FFT_SCRATCH + TW_WIN) * sizeof(float2) + (size_t)2 * 48 * 8 * sizeof(float);
        MHIP_TRY(c, hipFuncSetAttribute(reinterpret_cast<const void *>(k_mask_stream_2048), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        dim3 g((n_frames + ga.a.ft - 1) / ga.a.ft, n_streams);
        hipLaunchKernelGGL(k_mask_stream_2048, g, dim3(512), smem, st, ga);
    } else {
        // any other power of two: one frame at a time, block-cooperative FFT (runs re-analyse MK_WARM + 1 frames)
        MaskGenArgs ga{};
        ga.a = a; ga.N = c->N; ga.logH = c->logH; ga.tw = c->d_tw; ga.kw = c->d_kw; ga.kb = c->d_kb;
        const size_t smem = (size_t)2 * (c->hop + 1) * sizeof(float2) + ((size_t)3 * c->K + 2 * c->hop + 48 * 8) * sizeof(float);
        const int nthr = c->N >= 2048 ? 512 : 256;
        if (!(c->cfg.method == MCA_HIP_MASK_NOISY && c->frames_done == 0)) {
            // latency bound (one frame at a time behind ~20 barriers): as many workgroups per CU as LDS and wave slots allow
            const long long per_cu = std::max<long long>(1, std::min<long long>((160 * 1024) / (long long)smem, 2048 / nthr));
            ga.a.ft = 256;
            while (ga.a.ft > 16 && (long long)n_streams * ((n_frames + ga.a.ft - 1) / ga.a.ft) < 256 * per_cu) ga.a.ft >>= 1;
        }
        if (smem > 64 * 1024)
            MHIP_TRY(c, hipFuncSetAttribute(reinterpret_cast<const void *>(k_mask_stream_gen), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        dim3 g((n_frames + ga.a.ft - 1) / ga.a.ft, n_streams);
        hipLaunchKernelGGL(k_mask_stream_gen, g, dim3(nthr), smem, st, ga);
    }
    MHIP_TRY(c, hipGetLastError());
    c->q_cur ^= 1; c->tail_cur ^= 1;
    c->frames_done += n_frames; c->streams_in_use = n_streams;
    return MCA_HIP_OK;
}

int mca_hip_mask_frames_host(mca_hip_mask_ctx *c, const float *pcm, int n_streams, int n_frames, float *out_pcm, int *decisions)
{
    if (!c || !pcm || !out_pcm) return mfail(c, MCA_HIP_ERR_INVALID_ARGUMENT, "NULL argument");
    if (n_streams < 1 || n_frames < 1) return mfail(c, MCA_HIP_ERR_INVALID_ARGUMENT, "n_streams/n_frames < 1");
    MHIP_TRY(c, hipSetDevice(c->cfg.device));
    const long long cs = (long long)(n_frames + 1) * c->hop, ss = 2 * cs;
    const size_t n_out = (size_t)n_streams * 2 * n_frames * c->hop, n_dec = decisions ? (size_t)n_streams * n_frames * 45 : 0;
    float *d_pcm = (float *)c->stage.get(0, (size_t)ss * n_streams * 4), *d_out = (float *)c->stage.get(1, n_out * 4);
    int *d_dec = (int *)c->stage.get(2, n_dec * 4);
    if (!d_pcm || !d_out || (decisions && !d_dec)) return mfail(c, MCA_HIP_ERR_OUT_OF_MEMORY, "device staging buffers for the host-pointer call");
    MHIP_TRY(c, hipMemcpy(d_pcm, pcm, (size_t)ss * n_streams * 4, hipMemcpyHostToDevice));
    const int rc = mca_hip_mask_frames_dev(c, d_pcm, ss, cs, n_streams, n_frames, d_out, d_dec, nullptr);
    if (rc) return rc;
    MHIP_TRY(c, hipDeviceSynchronize());
    MHIP_TRY(c, hipMemcpy(out_pcm, d_out, n_out * 4, hipMemcpyDeviceToHost));
    if (decisions) MHIP_TRY(c, hipMemcpy(decisions, d_dec, n_dec * 4, hipMemcpyDeviceToHost));
    return MCA_HIP_OK;
}

extern "C++" {
namespace {
constexpr unsigned MASK_MAGIC = 0x4d434d4bu;   // "MCMK"
std::vector<BlobPart> mask_parts(mca_hip_mask_ctx *c)
{
    const size_t ns = (size_t)c->cfg.max_streams;
    return {{c->d_Q[c->q_cur], ns * 45 * 4}, {c->d_noise, ns * 45 * 4}, {c->d_tail[c->tail_cur], ns * 2 * c->hop * 4},
            {c->d_Q64, 45 * 8}, {c->d_noise64, 45 * 8}};
}
unsigned mask_cfg_hash(const mca_hip_mask_ctx *c)
{
    const int v[5] = {c->N, c->cfg.sample_rate, c->cfg.method, c->cfg.algorithm, c->cfg.max_streams};
    unsigned h = blob_fnv(v, sizeof(v));
    h = blob_fnv(&c->cfg.micro_distance, sizeof(double), h);
    h = blob_fnv(&c->cfg.low_freq, sizeof(float), h);
    return blob_fnv(&c->cfg.high_freq, sizeof(float), h);
}
}  // namespace
}  // extern "C++"

long long mca_hip_mask_state_size(const mca_hip_mask_ctx *c)
{
    return c ? blob_size(mask_parts(const_cast<mca_hip_mask_ctx *>(c))) : (long long)MCA_HIP_ERR_INVALID_ARGUMENT;
}

int mca_hip_mask_state_save(mca_hip_mask_ctx *c, void *blob, long long bytes)
{
    if (!c) return MCA_HIP_ERR_INVALID_ARGUMENT;
    MHIP_TRY(c, hipSetDevice(c->cfg.device));
    BlobHeader h{MASK_MAGIC, 1, mask_cfg_hash(c), 0, {c->frames_done, (long long)c->first_call, (long long)c->streams_in_use, 0}};
    const int rc = blob_save(mask_parts(c), h, blob, bytes);
    return rc ? mfail(c, rc == 2 ? MCA_HIP_ERR_HIP : MCA_HIP_ERR_INVALID_ARGUMENT, blob_error(rc)) : MCA_HIP_OK;
}

int mca_hip_mask_state_load(mca_hip_mask_ctx *c, const void *blob, long long bytes)
{
    if (!c) return MCA_HIP_ERR_INVALID_ARGUMENT;
    MHIP_TRY(c, hipSetDevice(c->cfg.device));
    BlobHeader h;
    const int rc = blob_load(mask_parts(c), MASK_MAGIC, mask_cfg_hash(c), blob, bytes, &h);
    if (rc) return mfail(c, rc == 2 ? MCA_HIP_ERR_HIP : MCA_HIP_ERR_INVALID_ARGUMENT, blob_error(rc));
    c->frames_done = h.host[0]; c->first_call = (int)h.host[1]; c->streams_in_use = (int)h.host[2];
    return MCA_HIP_OK;
}

int mca_hip_mask_process_frame(mca_hip_mask_ctx *c, double *left, double *right, int ccs_len, int *decisions)
{
    if (!c) return MCA_HIP_ERR_INVALID_ARGUMENT;
    if (!left || !right) return mfail(c, MCA_HIP_ERR_INVALID_ARGUMENT, "left/right is NULL");
    if (ccs_len != c->N + 2) return mfail(c, MCA_HIP_ERR_INVALID_ARGUMENT, "ccs_len != fft_size + 2");
    if (c->cfg.method == MCA_HIP_MASK_NOTHING) return MCA_HIP_OK;              // :130-134
    MHIP_TRY(c, hipSetDevice(c->cfg.device));
    const size_t nb = (size_t)ccs_len * 8;
    MHIP_TRY(c, hipMemcpy(c->d_io, left, nb, hipMemcpyHostToDevice));
    MHIP_TRY(c, hipMemcpy(c->d_io + ccs_len, right, nb, hipMemcpyHostToDevice));
    MHIP_TRY(c, hipMemset(c->d_io + 2 * ccs_len, 0, 2 * nb));                   // setZeros :142-143
    MaskFrameArgs a{};
    a.L = c->d_io; a.R = c->d_io + ccs_len; a.outL = c->d_io + 2 * ccs_len; a.outR = c->d_io + 3 * ccs_len;
    a.H = c->d_H; a.thr = c->d_thr; a.Q = c->d_Q64; a.noise = c->d_noise64;
    a.K = c->K; a.method = c->cfg.method; a.alg = c->cfg.algorithm; a.first_call = c->first_call;
    const float lam = 0.04f, rej = 0.999f, rho = 0.01f;
    a.lambda = (double)lam; a.one_minus_lambda = (double)(1 - lam); a.reject = (double)rej; a.rho = (double)rho;
    a.decisions = decisions ? c->d_dec : nullptr;
    hipLaunchKernelGGL(k_mask_frame, dim3(1), dim3(256), 0, 0, a);
    MHIP_TRY(c, hipGetLastError());
    ++c->first_call;                                                             // :192
    MHIP_TRY(c, hipMemcpy(left, a.outL, nb, hipMemcpyDeviceToHost));            // :199-200
    MHIP_TRY(c, hipMemcpy(right, a.outR, nb, hipMemcpyDeviceToHost));
    if (decisions) MHIP_TRY(c, hipMemcpy(decisions, c->d_dec, 45 * 4, hipMemcpyDeviceToHost));
    return MCA_HIP_OK;
}

}  // extern "C"
