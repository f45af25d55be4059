// api_multiband.hip -- C ABI of the multiband 2-microphone localiser (include/mcarray_hip.h, mca_hip_mb_*).
// Host side only: builds the tables the reference builds in its constructor
// (MultibandBinarualLocalisation.cpp:52-123), owns the per-array state, enqueues the kernels.  No CPU fallback.
#include "../../include/mcarray_hip.h"
#include "fft512.h"
#include "kernels.h"
#include "knobs.h"
#include "stage.h"
#include "state_blob.h"

#include <cmath>
#include <cstring>
#include <string>
#include <vector>

using namespace mca;

struct mca_hip_mb_ctx {
    mca_hip_mb_config cfg{};
    int N = 0, K = 0, H = 0, logH = 0, D = 0, nb = 0;
    float step = 0.f;
    std::vector<double> coef;                 // [nbins][K]
    std::vector<float> grid, delays;          // [D]
    float *d_window = nullptr, *d_coef = nullptr, *d_grid = nullptr;
    float2 *d_tw = nullptr, *d_T = nullptr;
    int *d_lo = nullptr, *d_hi = nullptr;
    float *d_corr[2] = {nullptr, nullptr}; int corr_cur = 0;
    double *d_gate = nullptr; float *d_cur = nullptr;
    // workspace
    float *d_raw = nullptr, *d_be = nullptr, *d_pf = nullptr, *d_ph = nullptr, *d_hprob = nullptr; int *d_hidx = nullptr;
    size_t ws_rows = 0;
    StagePool stage;
    std::string err;
};

namespace {

std::string g_mb_create_error;

int bfail(mca_hip_mb_ctx *c, int code, const std::string &msg)
{
    if (c) c->err = msg; else g_mb_create_error = msg;
    return code;
}

#define BHIP_TRY(ctx, expr)                                                                             \
    do {                                                                                                \
        hipError_t _e = (expr);                                                                         \
        if (_e != hipSuccess)                                                                           \
            return bfail(ctx, _e == hipErrorOutOfMemory ? MCA_HIP_ERR_OUT_OF_MEMORY : MCA_HIP_ERR_HIP,  \
                         std::string(#expr) + ": " + hipGetErrorString(_e));                           \
    } while (0)

void free_mb(mca_hip_mb_ctx *c)
{
    if (!c) return;
    auto F = [](void *p) { if (p) (void)hipFree(p); };
    F(c->d_window); F(c->d_coef); F(c->d_grid); F(c->d_tw); F(c->d_T); F(c->d_lo); F(c->d_hi);
    F(c->d_corr[0]); F(c->d_corr[1]); F(c->d_gate); F(c->d_cur);
    F(c->d_raw); F(c->d_be); F(c->d_pf); F(c->d_ph); F(c->d_hprob); F(c->d_hidx);
    c->stage.release();
    delete c;
}

int init_state(mca_hip_mb_ctx *c, hipStream_t st)
{
    const size_t na = (size_t)c->cfg.max_arrays;
    for (int i = 0; i < 2; ++i) BHIP_TRY(c, hipMemsetAsync(c->d_corr[i], 0, na * c->nb * c->D * 4, st));   // :106-110
    BHIP_TRY(c, hipMemsetAsync(c->d_gate, 0, na * 4 * 8, st));
    std::vector<float> cur(na * 2);
    for (size_t a = 0; a < na; ++a) { cur[2 * a] = 0.f; cur[2 * a + 1] = -1.f; }                            // :81-82
    BHIP_TRY(c, hipMemcpyAsync(c->d_cur, cur.data(), cur.size() * 4, hipMemcpyHostToDevice, st));
    BHIP_TRY(c, hipStreamSynchronize(st));
    return MCA_HIP_OK;
}

int ensure_ws(mca_hip_mb_ctx *c, size_t rows)
{
    if (rows <= c->ws_rows) return MCA_HIP_OK;
    auto F = [](void *p) { if (p) (void)hipFree(p); };
    F(c->d_raw); F(c->d_be); F(c->d_pf); F(c->d_ph); F(c->d_hprob); F(c->d_hidx);
    c->d_raw = c->d_be = c->d_pf = c->d_ph = c->d_hprob = nullptr; c->d_hidx = nullptr; c->ws_rows = 0;
    BHIP_TRY(c, hipMalloc((void **)&c->d_raw, rows * c->nb * c->D * 4));
    BHIP_TRY(c, hipMalloc((void **)&c->d_be, rows * c->nb * 4));
    BHIP_TRY(c, hipMalloc((void **)&c->d_pf, rows * 4));
    BHIP_TRY(c, hipMalloc((void **)&c->d_ph, rows * 4));
    BHIP_TRY(c, hipMalloc((void **)&c->d_hprob, rows * 4));
    BHIP_TRY(c, hipMalloc((void **)&c->d_hidx, rows * 4));
    c->ws_rows = rows;
    return MCA_HIP_OK;
}

}  // namespace

extern "C" {

const char *mca_hip_mb_last_error(const mca_hip_mb_ctx *ctx) { return ctx ? ctx->err.c_str() : g_mb_create_error.c_str(); }

int mca_hip_mb_create(const mca_hip_mb_config *cfg, mca_hip_mb_ctx **out)
{
    if (!cfg || !out) return bfail(nullptr, MCA_HIP_ERR_INVALID_ARGUMENT, "cfg/out is NULL");
    *out = nullptr;
    if (cfg->struct_size != (int)sizeof(mca_hip_mb_config)) return bfail(nullptr, MCA_HIP_ERR_INVALID_ARGUMENT, "struct_size mismatch");
    if (cfg->fft_size < 64 || (cfg->fft_size & (cfg->fft_size - 1)) || cfg->fft_size > 8192) return bfail(nullptr, MCA_HIP_ERR_INVALID_ARGUMENT, "fft_size must be a power of two in [64,8192]");
    if (cfg->sample_rate <= 0) return bfail(nullptr, MCA_HIP_ERR_INVALID_ARGUMENT, "sample_rate <= 0");
    if (!cfg->mic_xyz) return bfail(nullptr, MCA_HIP_ERR_INVALID_ARGUMENT, "mic_xyz is NULL");
    if (cfg->nbins < 1 || cfg->nbins > 27) return bfail(nullptr, MCA_HIP_ERR_INVALID_ARGUMENT, "nbins must be in [1,27] (nbins x 37 delays <= 1024 threads)");
    if (cfg->max_arrays < 1) return bfail(nullptr, MCA_HIP_ERR_INVALID_ARGUMENT, "max_arrays < 1");
    const double *x = cfg->mic_xyz;
    const double dist = std::sqrt(std::pow(x[3] - x[0], 2) + std::pow(x[4] - x[1], 2) + std::pow(x[5] - x[2], 2));   // distance(0,1) :61
    if (!(dist > 0)) return bfail(nullptr, MCA_HIP_ERR_INVALID_ARGUMENT, "the two microphones coincide");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return bfail(nullptr, MCA_HIP_ERR_NO_DEVICE, "no HIP device visible; libmcarray_hip has no CPU fallback");
    if (cfg->device < 0 || cfg->device >= ndev) return bfail(nullptr, MCA_HIP_ERR_INVALID_ARGUMENT, "device ordinal out of range");
    if (hipSetDevice(cfg->device) != hipSuccess) return bfail(nullptr, MCA_HIP_ERR_HIP, "hipSetDevice failed");

    mca_hip_mb_ctx *c = new mca_hip_mb_ctx();
    c->cfg = *cfg; c->cfg.mic_xyz = nullptr;
    c->N = cfg->fft_size; c->H = c->N / 2; c->K = c->H + 1; c->nb = cfg->nbins;
    while ((1 << c->logH) < c->H) ++c->logH;
    c->step = (float)(5 * M_PI / 180);                                          // :62
    c->D = (int)(std::floor(M_PI / (double)c->step) + 1);                       // :63
    const int K = c->K, D = c->D, nb = c->nb;
    // [BUILD-DEFINES] LINEAR filter bank of dsp::SubBandSTFTAnalysis(nbins, fs, order, 2, 100, fmax, LINEAR) (:54-60):
    // unit-peak triangles, edges linearly spaced between 100 Hz and maxFreqForSpatialAliasing (float in, float out)
    const double fmin = 100.0;
    const double fmax = (double)(float)(346.1 / (double)(2 * (float)dist));     // microhponeArrayHelpers.cpp:85-89
    c->coef.assign((size_t)nb * K, 0.0);
    std::vector<int> lo(nb), hi(nb);
    std::vector<float> coef_f((size_t)nb * K);
    for (int b = 0; b < nb; ++b) {
        const double f0 = fmin + (fmax - fmin) * (double)b / (double)(nb + 1);
        const double f1 = fmin + (fmax - fmin) * (double)(b + 1) / (double)(nb + 1);
        const double f2 = fmin + (fmax - fmin) * (double)(b + 2) / (double)(nb + 1);
        lo[b] = K; hi[b] = -1;
        for (int k = 0; k < K; ++k) {
            const double f = (double)k * (double)cfg->sample_rate / (double)c->N;
            double h = 0;
            if (f > f0 && f <= f1) h = (f - f0) / (f1 - f0);
            else if (f > f1 && f < f2) h = (f2 - f) / (f2 - f1);
            c->coef[(size_t)b * K + k] = h;
            coef_f[(size_t)b * K + k] = (float)h;
            if (h > 0) { lo[b] = std::min(lo[b], k); hi[b] = std::max(hi[b], k); }
        }
    }
    // delay grid (:101-104) with the class's own float doaIdx2angle (MultibandBinarualLocalisation.h:96-100)
    c->grid.resize(D); c->delays.resize(D);
    for (int i = 0; i < D; ++i) {
        const float ang = (float)((double)((float)i * c->step) - M_PI_2);
        c->grid[i] = ang;
        const float tsec = (float)(((double)(float)dist * std::sin((double)ang)) / 346.1);   // doaToDelayFarField :46-67
        c->delays[i] = tsec * (float)cfg->sample_rate;                                       // :69-72
    }
    std::vector<float2> T((size_t)K * D);
    for (int k = 0; k < K; ++k)
        for (int d = 0; d < D; ++d) {
            const double ph = 2.0 * M_PI * (double)k * (double)c->delays[d] / (double)c->N;
            T[(size_t)k * D + d] = make_float2((float)std::cos(ph), (float)std::sin(ph));
        }
    std::vector<float> win(c->N);
    for (int n = 0; n < c->N; ++n) win[n] = (float)(0.5 - 0.5 * std::cos(2.0 * M_PI * n / c->N));
    std::vector<float2> tw(c->N / 2);
    for (int i = 0; i < c->N / 2; ++i) tw[i] = make_float2((float)std::cos(2.0 * M_PI * i / c->N), (float)(-std::sin(2.0 * M_PI * i / c->N)));

    int rc = MCA_HIP_OK;
    auto up = [&](void **dst, const void *src, size_t bytes) -> int {
        BHIP_TRY(c, hipMalloc(dst, bytes));
        BHIP_TRY(c, hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice));
        return MCA_HIP_OK;
    };
    auto zalloc = [&](void **dst, size_t bytes) -> int {
        BHIP_TRY(c, hipMalloc(dst, bytes));
        BHIP_TRY(c, hipMemset(*dst, 0, bytes));
        return MCA_HIP_OK;
    };
    const size_t na = (size_t)cfg->max_arrays;
    if ((rc = up((void **)&c->d_window, win.data(), win.size() * 4)) || (rc = up((void **)&c->d_tw, tw.data(), tw.size() * 8)) ||
        (rc = up((void **)&c->d_coef, coef_f.data(), coef_f.size() * 4)) || (rc = up((void **)&c->d_grid, c->grid.data(), D * 4)) ||
        (rc = up((void **)&c->d_T, T.data(), T.size() * 8)) || (rc = up((void **)&c->d_lo, lo.data(), nb * 4)) || (rc = up((void **)&c->d_hi, hi.data(), nb * 4)) ||
        (rc = zalloc((void **)&c->d_corr[0], na * nb * D * 4)) || (rc = zalloc((void **)&c->d_corr[1], na * nb * D * 4)) ||
        (rc = zalloc((void **)&c->d_gate, na * 4 * 8)) || (rc = zalloc((void **)&c->d_cur, na * 2 * 4)) || (rc = init_state(c, nullptr))) {
        g_mb_create_error = c->err; free_mb(c); return rc;
    }
    *out = c;
    return MCA_HIP_OK;
}

void mca_hip_mb_destroy(mca_hip_mb_ctx *c)
{
    if (!c) return;
    (void)hipSetDevice(c->cfg.device);
    (void)hipDeviceSynchronize();
    free_mb(c);
}

int mca_hip_mb_num_steps(const mca_hip_mb_ctx *c) { return c ? c->D : MCA_HIP_ERR_INVALID_ARGUMENT; }

int mca_hip_mb_get_filters(const mca_hip_mb_ctx *c, double *out)
{
    if (!c || !out) return MCA_HIP_ERR_INVALID_ARGUMENT;
    std::memcpy(out, c->coef.data(), c->coef.size() * sizeof(double));
    return MCA_HIP_OK;
}

int mca_hip_mb_reset(mca_hip_mb_ctx *c, void *stream)
{
    if (!c) return MCA_HIP_ERR_INVALID_ARGUMENT;
    BHIP_TRY(c, hipSetDevice(c->cfg.device));
    return init_state(c, (hipStream_t)stream);
}

int mca_hip_mb_frames_dev(mca_hip_mb_ctx *c, const float *pcm, long long array_stride, long long ch_stride, int n_arrays,
                          int n_frames, float *doa_rad, float *prob, unsigned char *voiced, float *power, int *band_idx,
                          float *energy_in_doa, float *band_corr, void *stream)
{
    if (!c) return MCA_HIP_ERR_INVALID_ARGUMENT;
    if (!pcm || !doa_rad || !prob) return bfail(c, MCA_HIP_ERR_INVALID_ARGUMENT, "pcm_dev / doa_rad_dev / prob_dev is NULL");
    if (n_arrays < 1 || n_arrays > c->cfg.max_arrays) return bfail(c, MCA_HIP_ERR_INVALID_ARGUMENT, "n_arrays outside [1, max_arrays]");
    if (n_frames < 1) return bfail(c, MCA_HIP_ERR_INVALID_ARGUMENT, "n_frames < 1");
    const long long need = (long long)(n_frames + 1) * c->H;
    if (ch_stride < need) return bfail(c, MCA_HIP_ERR_INVALID_ARGUMENT, "ch_stride shorter than (n_frames+1)*hop samples");
    if (n_arrays > 1 && array_stride < ch_stride + need) return bfail(c, MCA_HIP_ERR_INVALID_ARGUMENT, "array_stride too short");
    if ((ch_stride & 1) || (array_stride & 1) || (reinterpret_cast<uintptr_t>(pcm) & 7))
        return bfail(c, MCA_HIP_ERR_INVALID_ARGUMENT, "pcm_dev must be 8-byte aligned with even strides (float2 loads)");
    hipStream_t st = (hipStream_t)stream;
    int rc = ensure_ws(c, (size_t)n_arrays * n_frames);
    if (rc) return rc;

    MbAnalyseArgs aa{};
    aa.pcm = pcm; aa.array_stride = array_stride; aa.ch_stride = ch_stride; aa.n_frames = n_frames;
    aa.N = c->N; aa.logH = c->logH; aa.nbins = c->nb; aa.D = c->D;
    aa.window = c->d_window; aa.tw = c->d_tw; aa.coef = c->d_coef; aa.lo = c->d_lo; aa.hi = c->d_hi; aa.T = c->d_T;
    aa.raw = c->d_raw; aa.band_energy = c->d_be; aa.p_full = c->d_pf; aa.p_half = c->d_ph;
    static const bool no_tuned = mca::measure_env("MCA_HIP_MB_GENERIC") != nullptr;     // A/B switch for measurements
    if (c->N == FFT_N && !no_tuned) {
        // 1024-sample frames: wave-level FFT, 4 frames x 2 channels per pass
        int fpb = 16;
        while (fpb > 4 && (long long)n_arrays * ((n_frames + fpb - 1) / fpb) < 512) fpb >>= 1;
        const size_t smem1 = (size_t)8 * FFT_SCRATCH * 8 + (size_t)4 * 520 * (8 + 4) + (size_t)TW_WORDS * 8 + 16 * 4;
        BHIP_TRY(c, hipFuncSetAttribute(reinterpret_cast<const void *>(k_mb_analyse_1024), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem1));
        hipLaunchKernelGGL(k_mb_analyse_1024, dim3((n_frames + fpb - 1) / fpb, n_arrays), dim3(512), smem1, st, aa, fpb);
    } else if (c->N == 512 && !no_tuned) {
        // 512-sample frames: both channels of a frame in one 512-point complex transform, 8 frames per pass
        int fpb = 32;
        while (fpb > 8 && (long long)n_arrays * ((n_frames + fpb - 1) / fpb) < 512) fpb >>= 1;
        const size_t smem1 = (size_t)(16 * 258 + 8 * FFT_SCRATCH + TW_WIN) * 8 + 16 * 4;
        BHIP_TRY(c, hipFuncSetAttribute(reinterpret_cast<const void *>(k_mb_analyse_512), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem1));
        hipLaunchKernelGGL(k_mb_analyse_512, dim3((n_frames + fpb - 1) / fpb, n_arrays), dim3(512), smem1, st, aa, fpb);
    } else {
        const size_t smem1 = (size_t)2 * (c->H + 1) * 8 + (size_t)c->K * 8 + (size_t)c->K * 4 + 8 * 4;
        if (smem1 > 64 * 1024)
            BHIP_TRY(c, hipFuncSetAttribute(reinterpret_cast<const void *>(k_mb_analyse), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem1));
        hipLaunchKernelGGL(k_mb_analyse, dim3(n_frames, n_arrays), dim3(256), smem1, st, aa);
    }

    MbScanArgs sa{};
    sa.raw = c->d_raw; sa.band_energy = c->d_be; sa.n_frames = n_frames; sa.nbins = c->nb; sa.D = c->D;
    // frames per chunk: every chunk re-reads 24 warm-up frames; its smoothed correlations stay in LDS (chunk x nbins x D floats)
    sa.chunk = (size_t)32 * (c->nb * c->D + c->D + c->nb) * 4 <= 80 * 1024 ? 32 : 16;
    const float mem = 0.4f;                                                     // _corrMemoryFactor (MultibandBinarualLocalisation.h:46)
    sa.mem = mem; sa.one_minus_mem = 1 - mem;
    sa.corr_in = c->d_corr[c->corr_cur]; sa.corr_out = c->d_corr[c->corr_cur ^ 1];
    sa.hist_idx = c->d_hidx; sa.hist_prob = c->d_hprob;
    sa.band_idx = band_idx; sa.energy_in_doa = energy_in_doa; sa.band_corr = band_corr;
    const int BD = c->nb * c->D;
    const size_t smem2 = (size_t)sa.chunk * (BD + c->D) * 4 + (size_t)sa.chunk * c->nb * 4;
    if (smem2 > 64 * 1024)
        BHIP_TRY(c, hipFuncSetAttribute(reinterpret_cast<const void *>(k_mb_scan), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem2));
    hipLaunchKernelGGL(k_mb_scan, dim3((n_frames + sa.chunk - 1) / sa.chunk, n_arrays), dim3((BD + 63) / 64 * 64), smem2, st, sa);

    MbSummaryArgs ma{};
    ma.p_full = c->d_pf; ma.p_half = c->d_ph; ma.hist_idx = c->d_hidx; ma.hist_prob = c->d_hprob;
    ma.n_frames = n_frames; ma.K = c->K; ma.needed_samples = (int)(3.0 * c->cfg.sample_rate);   // SoundLocalisationImpl.h:77
    ma.use_floor = c->cfg.use_power_floor; ma.margin_db = 3.0f;                 // _noiseMarginDB (.h:47)
    ma.grid = c->d_grid; ma.gate = c->d_gate; ma.cur = c->d_cur;
    ma.doa_rad = doa_rad; ma.prob = prob; ma.power = power; ma.voiced = voiced;
    hipLaunchKernelGGL(k_mb_summary, dim3(n_arrays), dim3(256), 0, st, ma);
    BHIP_TRY(c, hipGetLastError());
    c->corr_cur ^= 1;
    return MCA_HIP_OK;
}

extern "C++" {
namespace {
constexpr unsigned MB_MAGIC = 0x4d434d42u;     // "MCMB"
std::vector<BlobPart> mb_parts(mca_hip_mb_ctx *c)
{
    const size_t na = (size_t)c->cfg.max_arrays;
    return {{c->d_corr[c->corr_cur], na * c->nb * c->D * 4}, {c->d_gate, na * 4 * 8}, {c->d_cur, na * 2 * 4}};
}
unsigned mb_cfg_hash(const mca_hip_mb_ctx *c)
{
    const int v[6] = {c->N, c->nb, c->D, c->cfg.max_arrays, c->cfg.sample_rate, c->cfg.use_power_floor};
    return blob_fnv(c->delays.data(), c->delays.size() * sizeof(float), blob_fnv(v, sizeof(v)));
}
}  // namespace
}  // extern "C++"

long long mca_hip_mb_state_size(const mca_hip_mb_ctx *c)
{
    return c ? blob_size(mb_parts(const_cast<mca_hip_mb_ctx *>(c))) : (long long)MCA_HIP_ERR_INVALID_ARGUMENT;
}

int mca_hip_mb_state_save(mca_hip_mb_ctx *c, void *blob, long long bytes)
{
    if (!c) return MCA_HIP_ERR_INVALID_ARGUMENT;
    BHIP_TRY(c, hipSetDevice(c->cfg.device));
    BlobHeader h{MB_MAGIC, 1, mb_cfg_hash(c), 0, {0, 0, 0, 0}};
    const int rc = blob_save(mb_parts(c), h, blob, bytes);
    return rc ? bfail(c, rc == 2 ? MCA_HIP_ERR_HIP : MCA_HIP_ERR_INVALID_ARGUMENT, blob_error(rc)) : MCA_HIP_OK;
}

int mca_hip_mb_state_load(mca_hip_mb_ctx *c, const void *blob, long long bytes)
{
    if (!c) return MCA_HIP_ERR_INVALID_ARGUMENT;
    BHIP_TRY(c, hipSetDevice(c->cfg.device));
    BlobHeader h;
    const int rc = blob_load(mb_parts(c), MB_MAGIC, mb_cfg_hash(c), blob, bytes, &h);
    return rc ? bfail(c, rc == 2 ? MCA_HIP_ERR_HIP : MCA_HIP_ERR_INVALID_ARGUMENT, blob_error(rc)) : MCA_HIP_OK;
}

int mca_hip_mb_frames_host(mca_hip_mb_ctx *c, const float *pcm, int n_arrays, int n_frames, float *doa_rad, float *prob,
                           unsigned char *voiced, float *power, int *band_idx, float *energy_in_doa, float *band_corr)
{
    if (!c || !pcm || !doa_rad || !prob) return bfail(c, MCA_HIP_ERR_INVALID_ARGUMENT, "NULL argument");
    if (n_arrays < 1 || n_frames < 1) return bfail(c, MCA_HIP_ERR_INVALID_ARGUMENT, "n_arrays/n_frames < 1");
    BHIP_TRY(c, hipSetDevice(c->cfg.device));
    const long long cs = (long long)(n_frames + 1) * c->H, as = 2 * cs;
    const size_t nf = (size_t)n_arrays * n_frames;
    const size_t n_bi = band_idx ? nf * c->nb : 0, n_eid = energy_in_doa ? nf * c->D : 0, n_bc = band_corr ? nf * c->nb * c->D : 0;
    float *d_pcm = (float *)c->stage.get(0, (size_t)as * n_arrays * 4), *d_rad = (float *)c->stage.get(1, nf * 4);
    float *d_prob = (float *)c->stage.get(2, nf * 4), *d_pow = (float *)c->stage.get(3, nf * 4);
    unsigned char *d_v = (unsigned char *)c->stage.get(4, nf);
    int *d_bi = (int *)c->stage.get(5, n_bi * 4);
    float *d_eid = (float *)c->stage.get(6, n_eid * 4), *d_bc = (float *)c->stage.get(7, n_bc * 4);
    if (!d_pcm || !d_rad || !d_prob || !d_pow || !d_v || (band_idx && !d_bi) || (energy_in_doa && !d_eid) || (band_corr && !d_bc))
        return bfail(c, MCA_HIP_ERR_OUT_OF_MEMORY, "device staging buffers for the host-pointer call");
    BHIP_TRY(c, hipMemcpy(d_pcm, pcm, (size_t)as * n_arrays * 4, hipMemcpyHostToDevice));
    const int rc = mca_hip_mb_frames_dev(c, d_pcm, as, cs, n_arrays, n_frames, d_rad, d_prob, d_v, d_pow, d_bi, d_eid, d_bc, nullptr);
    if (rc) return rc;
    BHIP_TRY(c, hipDeviceSynchronize());
    BHIP_TRY(c, hipMemcpy(doa_rad, d_rad, nf * 4, hipMemcpyDeviceToHost));
    BHIP_TRY(c, hipMemcpy(prob, d_prob, nf * 4, hipMemcpyDeviceToHost));
    if (voiced) BHIP_TRY(c, hipMemcpy(voiced, d_v, nf, hipMemcpyDeviceToHost));
    if (power) BHIP_TRY(c, hipMemcpy(power, d_pow, nf * 4, hipMemcpyDeviceToHost));
    if (band_idx) BHIP_TRY(c, hipMemcpy(band_idx, d_bi, n_bi * 4, hipMemcpyDeviceToHost));
    if (energy_in_doa) BHIP_TRY(c, hipMemcpy(energy_in_doa, d_eid, n_eid * 4, hipMemcpyDeviceToHost));
    if (band_corr) BHIP_TRY(c, hipMemcpy(band_corr, d_bc, n_bc * 4, hipMemcpyDeviceToHost));
    return MCA_HIP_OK;
}

}  // extern "C"
