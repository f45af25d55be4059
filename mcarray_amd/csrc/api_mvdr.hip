// api_mvdr.hip -- C ABI of the MVDR-style beamformer with a per-bin spatial covariance (include/mcarray_hip.h,
// mca_hip_mvdr_*; BASELINE.json configs[3]; SURVEY A.9 -- no reference counterpart, conventions of Beamformer.cpp:59).
// Host side only: owns the per-stream state (covariances, their traces, overlap-add tails) and the spectra
// workspace, enqueues the three kernels of kernels_mvdr.hip.  No CPU fallback.
#include "../../include/mcarray_hip.h"
#include "fft512.h"
#include "kernels.h"
#include "knobs.h"
#include "stage.h"
#include "state_blob.h"

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

using namespace mca;

struct mca_hip_mvdr_ctx {
    mca_hip_mvdr_config cfg{};
    int N = 0, K = 0, H = 0, logH = 0, M = 0, tri = 0;
    float *d_window = nullptr;
    float2 *d_tw = nullptr;
    double *d_micx = nullptr;
    float2 *d_phi = nullptr;      // [max_streams][K][tri]
    float *d_trace = nullptr;     // [max_streams][K]
    float2 *d_phi_tail = nullptr; float *d_trace_tail = nullptr;   // exit state of the pieced tail launch (<= 128 workgroups x 64 problems), copied back behind it
    float *d_tail[2] = {nullptr, nullptr}; int tail_cur = 0;   // [max_streams][H]
    // workspace
    float2 *d_X = nullptr; size_t x_rows = 0;      // [rows][K][M]
    float2 *d_Y = nullptr; float2 *d_T = nullptr; size_t y_rows = 0;   // d_T: factored steering phasors [rows][M][N/64 + 33]
    StagePool stage;
    bool timing = false;
    struct Ev { int id; hipEvent_t a, b; };
    std::vector<Ev> events;
    int t_launches[3] = {};
    double t_ms[3] = {};
    std::string err;
};

namespace {

std::string g_mvdr_create_error;

int vfail(mca_hip_mvdr_ctx *c, int code, const std::string &msg)
{
    if (c) c->err = msg; else g_mvdr_create_error = msg;
    return code;
}

#define VHIP_TRY(ctx, expr)                                                                             \
    do {                                                                                                \
        hipError_t _e = (expr);                                                                         \
        if (_e != hipSuccess)                                                                           \
            return vfail(ctx, _e == hipErrorOutOfMemory ? MCA_HIP_ERR_OUT_OF_MEMORY : MCA_HIP_ERR_HIP,  \
                         std::string(#expr) + ": " + hipGetErrorString(_e));                           \
    } while (0)

void free_mvdr(mca_hip_mvdr_ctx *c)
{
    if (!c) return;
    auto F = [](void *p) { if (p) (void)hipFree(p); };
    F(c->d_window); F(c->d_tw); F(c->d_micx); F(c->d_phi); F(c->d_trace); F(c->d_phi_tail); F(c->d_trace_tail); F(c->d_tail[0]); F(c->d_tail[1]);
    F(c->d_X); F(c->d_Y); F(c->d_T);
    for (auto &e : c->events) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
    c->stage.release();
    delete c;
}

int init_state(mca_hip_mvdr_ctx *c, hipStream_t st)
{
    const size_t ns = (size_t)c->cfg.max_streams;
    VHIP_TRY(c, hipMemsetAsync(c->d_phi, 0, ns * c->K * c->tri * sizeof(float2), st));
    VHIP_TRY(c, hipMemsetAsync(c->d_trace, 0, ns * c->K * 4, st));
    for (int i = 0; i < 2; ++i) VHIP_TRY(c, hipMemsetAsync(c->d_tail[i], 0, ns * c->H * 4, st));
    VHIP_TRY(c, hipStreamSynchronize(st));
    return MCA_HIP_OK;
}

int ensure_ws(mca_hip_mvdr_ctx *c, size_t rows)
{
    auto F = [](void *p) { if (p) (void)hipFree(p); };
    if (rows > c->x_rows) {
        F(c->d_X); c->d_X = nullptr; c->x_rows = 0;
        VHIP_TRY(c, hipMalloc((void **)&c->d_X, rows * c->K * c->M * sizeof(float2)));
        c->x_rows = rows;
    }
    if (rows > c->y_rows) {
        F(c->d_Y); F(c->d_T); c->d_Y = nullptr; c->d_T = nullptr; c->y_rows = 0;
        VHIP_TRY(c, hipMalloc((void **)&c->d_Y, rows * c->K * sizeof(float2)));
        VHIP_TRY(c, hipMalloc((void **)&c->d_T, rows * c->M * (c->N / 64 + 33) * sizeof(float2)));
        c->y_rows = rows;
    }
    return MCA_HIP_OK;
}

void t_begin(mca_hip_mvdr_ctx *c, int id, hipStream_t st)
{
    if (!c->timing) return;
    mca_hip_mvdr_ctx::Ev ev; ev.id = id;
    (void)hipEventCreate(&ev.a); (void)hipEventCreate(&ev.b);
    (void)hipEventRecord(ev.a, st);
    c->events.push_back(ev);
}
void t_end(mca_hip_mvdr_ctx *c, hipStream_t st)
{
    if (!c->timing) return;
    (void)hipEventRecord(c->events.back().b, st);
}

// threads per workgroup of the block-cooperative FFT: a radix-4 pass has nch * N/8 work items, two per thread
int fft_threads(int N, int nch)
{
    const int items = nch * (N / 8);
    int t = 256;
    while (t < 1024 && t * 2 <= items) t <<= 1;
    return t;
}

}  // namespace

extern "C" {

const char *mca_hip_mvdr_last_error(const mca_hip_mvdr_ctx *ctx) { return ctx ? ctx->err.c_str() : g_mvdr_create_error.c_str(); }

int mca_hip_mvdr_create(const mca_hip_mvdr_config *cfg, mca_hip_mvdr_ctx **out)
{
    if (!cfg || !out) return vfail(nullptr, MCA_HIP_ERR_INVALID_ARGUMENT, "cfg/out is NULL");
    *out = nullptr;
    if (cfg->struct_size != (int)sizeof(mca_hip_mvdr_config)) return vfail(nullptr, MCA_HIP_ERR_INVALID_ARGUMENT, "struct_size mismatch");
    if (cfg->fft_size < 64 || (cfg->fft_size & (cfg->fft_size - 1)) || cfg->fft_size > 8192)
        return vfail(nullptr, MCA_HIP_ERR_INVALID_ARGUMENT, "fft_size must be a power of two in [64,8192]");
    if (cfg->sample_rate <= 0) return vfail(nullptr, MCA_HIP_ERR_INVALID_ARGUMENT, "sample_rate <= 0");
    if (cfg->n_mics < 2 || cfg->n_mics > 16) return vfail(nullptr, MCA_HIP_ERR_INVALID_ARGUMENT, "n_mics must be in [2,16]");
    if (!cfg->mic_xyz) return vfail(nullptr, MCA_HIP_ERR_INVALID_ARGUMENT, "mic_xyz is NULL");
    if (!(cfg->alpha >= 0.0 && cfg->alpha < 1.0)) return vfail(nullptr, MCA_HIP_ERR_INVALID_ARGUMENT, "alpha must be in [0,1)");
    if (!(cfg->loading > 0.0)) return vfail(nullptr, MCA_HIP_ERR_INVALID_ARGUMENT, "loading must be > 0 (the first M-1 covariances of a stream are rank deficient)");
    if (cfg->max_streams < 1) return vfail(nullptr, MCA_HIP_ERR_INVALID_ARGUMENT, "max_streams < 1");
    if ((size_t)cfg->n_mics * (cfg->fft_size / 2 + 1) * 8 > 160 * 1024)
        return vfail(nullptr, MCA_HIP_ERR_UNSUPPORTED, "n_mics spectra of N/2+1 bins exceed the 160 KiB LDS of a CU");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return vfail(nullptr, MCA_HIP_ERR_NO_DEVICE, "no HIP device visible; libmcarray_hip has no CPU fallback");
    if (cfg->device < 0 || cfg->device >= ndev) return vfail(nullptr, MCA_HIP_ERR_INVALID_ARGUMENT, "device ordinal out of range");
    if (hipSetDevice(cfg->device) != hipSuccess) return vfail(nullptr, MCA_HIP_ERR_HIP, "hipSetDevice failed");

    mca_hip_mvdr_ctx *c = new mca_hip_mvdr_ctx();
    c->cfg = *cfg; c->cfg.mic_xyz = nullptr;
    c->N = cfg->fft_size; c->H = c->N / 2; c->K = c->H + 1; c->M = cfg->n_mics; c->tri = c->M * (c->M + 1) / 2;
    while ((1 << c->logH) < c->H) ++c->logH;
    std::vector<float> win(c->N);
    for (int n = 0; n < c->N; ++n) win[n] = (float)(0.5 - 0.5 * std::cos(2.0 * M_PI * n / c->N));
    std::vector<float2> tw(c->N / 2);
    for (int i = 0; i < c->N / 2; ++i) tw[i] = make_float2((float)std::cos(2.0 * M_PI * i / c->N), (float)(-std::sin(2.0 * M_PI * i / c->N)));
    std::vector<double> mx(c->M);
    for (int m = 0; m < c->M; ++m) mx[m] = cfg->mic_xyz[3 * m];                                                // Beamformer.cpp:59: x only

    int rc = MCA_HIP_OK;
    auto up = [&](void **dst, const void *src, size_t bytes) -> int {
        VHIP_TRY(c, hipMalloc(dst, bytes));
        VHIP_TRY(c, hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice));
        return MCA_HIP_OK;
    };
    auto alloc = [&](void **dst, size_t bytes) -> int { VHIP_TRY(c, hipMalloc(dst, bytes)); return MCA_HIP_OK; };
    const size_t ns = (size_t)cfg->max_streams;
    if ((rc = up((void **)&c->d_window, win.data(), win.size() * 4)) || (rc = up((void **)&c->d_tw, tw.data(), tw.size() * 8)) ||
        (rc = up((void **)&c->d_micx, mx.data(), mx.size() * 8)) ||
        (rc = alloc((void **)&c->d_phi, ns * c->K * c->tri * sizeof(float2))) || (rc = alloc((void **)&c->d_trace, ns * c->K * 4)) ||
        (rc = alloc((void **)&c->d_phi_tail, (size_t)128 * 64 * c->tri * sizeof(float2))) || (rc = alloc((void **)&c->d_trace_tail, (size_t)128 * 64 * 4)) ||
        (rc = alloc((void **)&c->d_tail[0], ns * c->H * 4)) || (rc = alloc((void **)&c->d_tail[1], ns * c->H * 4)) ||
        (rc = init_state(c, nullptr))) {
        g_mvdr_create_error = c->err; free_mvdr(c); return rc;
    }
    *out = c;
    return MCA_HIP_OK;
}

void mca_hip_mvdr_destroy(mca_hip_mvdr_ctx *c)
{
    if (!c) return;
    (void)hipSetDevice(c->cfg.device);
    (void)hipDeviceSynchronize();
    free_mvdr(c);
}

int mca_hip_mvdr_reset(mca_hip_mvdr_ctx *c, void *stream)
{
    if (!c) return MCA_HIP_ERR_INVALID_ARGUMENT;
    VHIP_TRY(c, hipSetDevice(c->cfg.device));
    return init_state(c, (hipStream_t)stream);
}

int mca_hip_mvdr_frames_dev(mca_hip_mvdr_ctx *c, const float *pcm, long long stream_stride, long long mic_stride, int n_streams,
                            int n_frames, const float *doa_rad, float *out_pcm, float *out_spec, void *stream)
{
    if (!c) return MCA_HIP_ERR_INVALID_ARGUMENT;
    if (!pcm || !doa_rad) return vfail(c, MCA_HIP_ERR_INVALID_ARGUMENT, "pcm_dev / doa_rad_dev is NULL");
    if (!out_pcm && !out_spec) return vfail(c, MCA_HIP_ERR_INVALID_ARGUMENT, "out_pcm_dev and out_spec_dev are both NULL");
    if (n_streams < 1 || n_streams > c->cfg.max_streams) return vfail(c, MCA_HIP_ERR_INVALID_ARGUMENT, "n_streams outside [1, max_streams]");
    if (n_frames < 1) return vfail(c, MCA_HIP_ERR_INVALID_ARGUMENT, "n_frames < 1");
    const long long need = (long long)(n_frames + 1) * c->H;
    if (mic_stride < need) return vfail(c, MCA_HIP_ERR_INVALID_ARGUMENT, "mic_stride shorter than (n_frames+1)*hop samples");
    if (n_streams > 1 && stream_stride < (long long)(c->M - 1) * mic_stride + need) return vfail(c, MCA_HIP_ERR_INVALID_ARGUMENT, "stream_stride too short");
    if ((mic_stride & 1) || (stream_stride & 1) || (reinterpret_cast<uintptr_t>(pcm) & 7))
        return vfail(c, MCA_HIP_ERR_INVALID_ARGUMENT, "pcm_dev must be 8-byte aligned with even strides (float2 loads)");
    if (out_spec && (reinterpret_cast<uintptr_t>(out_spec) & 7)) return vfail(c, MCA_HIP_ERR_INVALID_ARGUMENT, "out_spec_dev must be 8-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    VHIP_TRY(c, hipSetDevice(c->cfg.device));
    const size_t rows = (size_t)n_streams * n_frames;
    // the beamformed spectra go straight to the caller's buffer when one is given
    float2 *Y = out_spec ? reinterpret_cast<float2 *>(out_spec) : nullptr;
    int rc = ensure_ws(c, rows);
    if (rc) return rc;
    if (!Y) Y = c->d_Y;

    MvdrAnalyseArgs aa{};
    aa.pcm = pcm; aa.stream_stride = stream_stride; aa.mic_stride = mic_stride; aa.n_frames = n_frames;
    aa.N = c->N; aa.logH = c->logH; aa.M = c->M; aa.window = c->d_window; aa.tw = c->d_tw; aa.doa_rad = doa_rad;
    aa.X = c->d_X; aa.T = c->d_T; aa.mic_x = c->d_micx;
    aa.unit = (double)c->cfg.sample_rate / (double)c->N / 346.1;                      // Beamformer.cpp:59 without 2 pi
    static const bool no_tuned = mca::measure_env("MCA_HIP_MVDR_GENERIC") != nullptr;     // A/B switch for measurements
    t_begin(c, 0, st);
    if (c->N == FFT_N && !no_tuned) {
        // 1024-sample frames: wave-level FFT, one wave per channel, eight channels per pass
        int fpb = 8;
        while (fpb > 1 && (long long)n_streams * ((n_frames + fpb - 1) / fpb) < 1024) fpb >>= 1;
        const size_t smem1 = (size_t)(8 * 580 + TW_WORDS) * sizeof(float2);
        VHIP_TRY(c, hipFuncSetAttribute(reinterpret_cast<const void *>(k_mvdr_analyse_1024), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem1));
        hipLaunchKernelGGL(k_mvdr_analyse_1024, dim3((n_frames + fpb - 1) / fpb, n_streams), dim3(512), smem1, st, aa, fpb);
    } else if (c->N == 512 && !no_tuned) {
        // 512-sample frames: two channels per wave pass (kernels_stream.hip)
        int fpb = 8;
        while (fpb > 1 && (long long)n_streams * ((n_frames + fpb - 1) / fpb) < 1024) fpb >>= 1;
        const size_t smem1 = (size_t)(16 * 258 + 8 * FFT_SCRATCH + TW_WIN) * sizeof(float2);
        VHIP_TRY(c, hipFuncSetAttribute(reinterpret_cast<const void *>(k_mvdr_analyse_512), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem1));
        hipLaunchKernelGGL(k_mvdr_analyse_512, dim3((n_frames + fpb - 1) / fpb, n_streams), dim3(512), smem1, st, aa, fpb);
    } else {
        const size_t smem1 = (size_t)c->M * (c->H + 1) * sizeof(float2);
        if (smem1 > 64 * 1024)
            VHIP_TRY(c, hipFuncSetAttribute(reinterpret_cast<const void *>(k_mvdr_analyse), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem1));
        hipLaunchKernelGGL(k_mvdr_analyse, dim3(n_frames, n_streams), dim3(fft_threads(c->N, c->M)), smem1, st, aa);
    }
    t_end(c, st);

    MvdrSolveArgs sa{};
    sa.X = c->d_X; sa.T = c->d_T;
    sa.n_frames = n_frames; sa.K = c->K; sa.M = c->M;
    sa.alpha = (float)c->cfg.alpha; sa.one_minus_alpha = (float)(1.0 - c->cfg.alpha);
    sa.loading_over_m = (float)(c->cfg.loading / c->M);
    sa.phi = c->d_phi; sa.trace = c->d_trace; sa.Y = Y;
    sa.n_streams = n_streams;
    const int Q = (c->M + 3) / 4;                                                     // row slots per lane
    auto launch_solve = [&](long long pid0, long long n_prob, int pieces) {
        sa.pid0 = pid0; sa.n_prob = n_prob; sa.pieces = pieces;
        // an unsplit launch updates the state in place; the pieces of a split one all read the entry state, so the last piece
        // writes the exit state to a scratch copy that is moved over behind the launch
        if (pieces > 1) { sa.phi_out = c->d_phi_tail; sa.trace_out = c->d_trace_tail; sa.out_base = pid0; }
        else { sa.phi_out = c->d_phi; sa.trace_out = c->d_trace; sa.out_base = 0; }
        const dim3 sgrid((unsigned)((n_prob + 63) / 64 * pieces));
#define SOLVE(QQ)                                                                                          \
    do {                                                                                                   \
        if (c->M == 4 * (QQ)) hipLaunchKernelGGL((k_mvdr_solve<QQ, true>), sgrid, dim3(256), 0, st, sa);   \
        else hipLaunchKernelGGL((k_mvdr_solve<QQ, false>), sgrid, dim3(256), 0, st, sa);                   \
    } while (0)
        if (Q == 1) SOLVE(1);
        else if (Q == 2) SOLVE(2);
        else if (Q == 3) SOLVE(3);
        else SOLVE(4);
#undef SOLVE
    };
    // 512 workgroups are resident (two per CU at 253 VGPRs) and all take the same time: the workgroups behind the last whole
    // round (256 streams x 513 bins: 4 of 2052) would hold the GPU for a round of their own.  They go in a second launch,
    // cut along the FRAMES into pieces that each repeat the (cheap) covariance recursion of the frames before their own.
    static const int env_pieces = mca::measure_env("MCA_HIP_MVDR_PIECES") ? std::atoi(mca::measure_env("MCA_HIP_MVDR_PIECES")) : -1;   // A/B switch
    const long long n_prob = (long long)n_streams * c->K, n_wg = (n_prob + 63) / 64;
    const long long rem_wg = n_wg % 512;
    int pieces = 1;
    if (n_wg > 512 && rem_wg > 0 && rem_wg <= 128) {
        while (pieces < 8 && rem_wg * pieces * 2 <= 512 && n_frames / (pieces * 2) >= 4) pieces *= 2;
    }
    if (env_pieces >= 0 && (env_pieces <= 1 || (n_wg > 512 && rem_wg > 0 && rem_wg <= 128))) pieces = env_pieces == 0 ? 1 : std::min(std::max(env_pieces, 1), n_frames);
    t_begin(c, 1, st);
    if (pieces > 1 && n_wg > 512) {
        const long long main_prob = (n_wg - rem_wg) * 64, tail_prob = n_prob - main_prob;       // tail_prob <= 128 x 64: the scratch copy's size
        launch_solve(0, main_prob, 1);
        launch_solve(main_prob, tail_prob, pieces);
        VHIP_TRY(c, hipMemcpyAsync(c->d_phi + main_prob * c->tri, c->d_phi_tail, (size_t)tail_prob * c->tri * sizeof(float2), hipMemcpyDeviceToDevice, st));
        VHIP_TRY(c, hipMemcpyAsync(c->d_trace + main_prob, c->d_trace_tail, (size_t)tail_prob * 4, hipMemcpyDeviceToDevice, st));
    } else {
        launch_solve(0, n_prob, 1);
    }
    t_end(c, st);

    if (out_pcm) {
        MvdrSynthArgs ya{};
        ya.Y = Y; ya.n_frames = n_frames; ya.N = c->N; ya.logH = c->logH; ya.tw = c->d_tw;
        ya.ft = 16;
        while (ya.ft > 2 && (long long)n_streams * ((n_frames + ya.ft - 1) / ya.ft) < 1024) ya.ft >>= 1;
        ya.tail_in = c->d_tail[c->tail_cur]; ya.tail_out = c->d_tail[c->tail_cur ^ 1]; ya.out = out_pcm;
        const size_t smem3 = (size_t)(c->H + 1) * sizeof(float2) + (size_t)c->H * 4;
        if (smem3 > 64 * 1024)
            VHIP_TRY(c, hipFuncSetAttribute(reinterpret_cast<const void *>(k_mvdr_synth), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem3));
        t_begin(c, 2, st);
        hipLaunchKernelGGL(k_mvdr_synth, dim3((n_frames + ya.ft - 1) / ya.ft, n_streams), dim3(c->H >= 1024 ? 512 : 256), smem3, st, ya);
        t_end(c, st);
        c->tail_cur ^= 1;
    }
    VHIP_TRY(c, hipGetLastError());
    return MCA_HIP_OK;
}

int mca_hip_mvdr_frames_host(mca_hip_mvdr_ctx *c, const float *pcm, int n_streams, int n_frames, const float *doa_rad,
                             float *out_pcm, float *out_spec)
{
    if (!c || !pcm || !doa_rad) return vfail(c, MCA_HIP_ERR_INVALID_ARGUMENT, "NULL argument");
    if (n_streams < 1 || n_frames < 1) return vfail(c, MCA_HIP_ERR_INVALID_ARGUMENT, "n_streams/n_frames < 1");
    VHIP_TRY(c, hipSetDevice(c->cfg.device));
    const long long ms = (long long)(n_frames + 1) * c->H, ss = ms * c->M;
    const size_t nf = (size_t)n_streams * n_frames;
    float *d_pcm = (float *)c->stage.get(0, (size_t)ss * n_streams * 4), *d_doa = (float *)c->stage.get(1, nf * 4);
    float *d_out = out_pcm ? (float *)c->stage.get(2, nf * c->H * 4) : nullptr;
    float *d_spec = out_spec ? (float *)c->stage.get(3, nf * c->K * 8) : nullptr;
    if (!d_pcm || !d_doa || (out_pcm && !d_out) || (out_spec && !d_spec))
        return vfail(c, MCA_HIP_ERR_OUT_OF_MEMORY, "device staging buffers for the host-pointer call");
    VHIP_TRY(c, hipMemcpy(d_pcm, pcm, (size_t)ss * n_streams * 4, hipMemcpyHostToDevice));
    VHIP_TRY(c, hipMemcpy(d_doa, doa_rad, nf * 4, hipMemcpyHostToDevice));
    const int rc = mca_hip_mvdr_frames_dev(c, d_pcm, ss, ms, n_streams, n_frames, d_doa, d_out, d_spec, nullptr);
    if (rc) return rc;
    VHIP_TRY(c, hipDeviceSynchronize());
    if (out_pcm) VHIP_TRY(c, hipMemcpy(out_pcm, d_out, nf * c->H * 4, hipMemcpyDeviceToHost));
    if (out_spec) VHIP_TRY(c, hipMemcpy(out_spec, d_spec, nf * c->K * 8, hipMemcpyDeviceToHost));
    return MCA_HIP_OK;
}

int mca_hip_mvdr_get_covariance(mca_hip_mvdr_ctx *c, int s, double *out)
{
    if (!c || !out) return MCA_HIP_ERR_INVALID_ARGUMENT;
    if (s < 0 || s >= c->cfg.max_streams) return vfail(c, MCA_HIP_ERR_INVALID_ARGUMENT, "stream_index out of range");
    VHIP_TRY(c, hipSetDevice(c->cfg.device));
    VHIP_TRY(c, hipDeviceSynchronize());
    std::vector<float2> h((size_t)c->K * c->tri);
    VHIP_TRY(c, hipMemcpy(h.data(), c->d_phi + (size_t)s * c->K * c->tri, h.size() * sizeof(float2), hipMemcpyDeviceToHost));
    const int M = c->M;
    for (int k = 0; k < c->K; ++k)
        for (int i = 0; i < M; ++i)
            for (int j = 0; j <= i; ++j) {
                const float2 v = h[(size_t)k * c->tri + i * (i + 1) / 2 + j];
                double *lo = out + (((size_t)k * M + i) * M + j) * 2, *up = out + (((size_t)k * M + j) * M + i) * 2;
                lo[0] = v.x; lo[1] = i == j ? 0.0 : v.y;
                up[0] = v.x; up[1] = i == j ? 0.0 : -(double)v.y;
            }
    return MCA_HIP_OK;
}

extern "C++" {
namespace {
constexpr unsigned MVDR_MAGIC = 0x4d435644u;   // "MCVD"
std::vector<BlobPart> mvdr_parts(mca_hip_mvdr_ctx *c)
{
    const size_t ns = (size_t)c->cfg.max_streams;
    return {{c->d_phi, ns * c->K * c->tri * sizeof(float2)}, {c->d_trace, ns * c->K * 4}, {c->d_tail[c->tail_cur], ns * c->H * 4}};
}
unsigned mvdr_cfg_hash(const mca_hip_mvdr_ctx *c)
{
    const int v[5] = {c->N, c->M, c->cfg.max_streams, c->cfg.sample_rate, 0};
    unsigned h = blob_fnv(v, sizeof(v));
    h = blob_fnv(&c->cfg.alpha, sizeof(double), h);
    return blob_fnv(&c->cfg.loading, sizeof(double), h);
}
}  // namespace
}  // extern "C++"

long long mca_hip_mvdr_state_size(const mca_hip_mvdr_ctx *c)
{
    return c ? blob_size(mvdr_parts(const_cast<mca_hip_mvdr_ctx *>(c))) : (long long)MCA_HIP_ERR_INVALID_ARGUMENT;
}

int mca_hip_mvdr_state_save(mca_hip_mvdr_ctx *c, void *blob, long long bytes)
{
    if (!c) return MCA_HIP_ERR_INVALID_ARGUMENT;
    VHIP_TRY(c, hipSetDevice(c->cfg.device));
    BlobHeader h{MVDR_MAGIC, 1, mvdr_cfg_hash(c), 0, {0, 0, 0, 0}};
    const int rc = blob_save(mvdr_parts(c), h, blob, bytes);
    return rc ? vfail(c, rc == 2 ? MCA_HIP_ERR_HIP : MCA_HIP_ERR_INVALID_ARGUMENT, blob_error(rc)) : MCA_HIP_OK;
}

int mca_hip_mvdr_state_load(mca_hip_mvdr_ctx *c, const void *blob, long long bytes)
{
    if (!c) return MCA_HIP_ERR_INVALID_ARGUMENT;
    VHIP_TRY(c, hipSetDevice(c->cfg.device));
    BlobHeader h;
    const int rc = blob_load(mvdr_parts(c), MVDR_MAGIC, mvdr_cfg_hash(c), blob, bytes, &h);
    return rc ? vfail(c, rc == 2 ? MCA_HIP_ERR_HIP : MCA_HIP_ERR_INVALID_ARGUMENT, blob_error(rc)) : MCA_HIP_OK;
}

int mca_hip_mvdr_set_timing(mca_hip_mvdr_ctx *c, int enable)
{
    if (!c) return MCA_HIP_ERR_INVALID_ARGUMENT;
    c->timing = enable != 0;
    return MCA_HIP_OK;
}

int mca_hip_mvdr_get_timing(mca_hip_mvdr_ctx *c, int kernel_id, int *launches, double *total_ms)
{
    if (!c || kernel_id < 0 || kernel_id >= 3) return MCA_HIP_ERR_INVALID_ARGUMENT;
    for (auto &e : c->events) {
        VHIP_TRY(c, hipEventSynchronize(e.b));
        float ms = 0.f;
        VHIP_TRY(c, hipEventElapsedTime(&ms, e.a, e.b));
        c->t_ms[e.id] += ms; c->t_launches[e.id] += 1;
        (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b);
    }
    c->events.clear();
    if (launches) *launches = c->t_launches[kernel_id];
    if (total_ms) *total_ms = c->t_ms[kernel_id];
    return MCA_HIP_OK;
}

}  // extern "C"
