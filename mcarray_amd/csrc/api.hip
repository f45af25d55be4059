// api.hip -- the extern "C" layer of libmcarray_hip.so (see include/mcarray_hip.h).
// Host side only: validates arguments (every shape a kernel or its grid assumes is checked here
// before a launch), builds the tables the reference builds in its constructors, owns the
// per-context state and workspace, and enqueues the kernels.  No compute happens on the host
// and there is no CPU fallback: without a gfx950 device mca_hip_create fails.
#include "../../include/mcarray_hip.h"
#include "fft1024c.h"
#include "kernels.h"
#include "knobs.h"
#include "stage.h"

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

using namespace mca;

namespace {

std::string g_create_error;

struct TimedEvent { int id; hipEvent_t a, b; };

}  // namespace

struct mca_hip_graph;

#define MCA_MAX_LANES 1

// ---- switches --------------------------------------------------------------------------------------------------------------
// Read ONCE, by mca_hip_create, into the context: nothing on a call path reads the environment.  The product switches are fields of
// mca_hip_config (adaptive_fallback, adaptive_min_rows, adaptive_max_sources, scan_carry) with an environment override for tests and
// tools that reach a context only through a wrapper.  The A/B switches that exist so that a measured claim of DESIGN.md can be
// repeated -- one of them (MCA_HIP_BFW_ABL) returns deliberately wrong audio -- are compiled in only with -DMCA_MEASURE
// (make MEASURE=1, tools/*.py say when they need it); in the default build they are the constants below.
// The back-off policy reads the report of an adaptive call at a FIXED lag: the eligible call FB_LAG calls after the one that enqueued
// it (and blocks on the page-locked word if it has not arrived: at a lag of two the work of the call in between is still queued
// behind it, so the device does not idle).  What a call does therefore depends on the sequence of calls and their content only, not
// on when a report happens to land: two runs of one stream return the same bits.
constexpr int FB_LAG = 2, FB_RING = 64, FB_WAIT_MS = 4000;
struct Knobs {
    // product
    bool fb_enabled = true;            // MCA_HIP_ADAPT_FALLBACK=0 / cfg.adaptive_fallback = OFF
    long long adapt_min_rows = 4096;   // MCA_HIP_ADAPT_MIN_ROWS / cfg.adaptive_min_rows
    int repair_items = 768;            // (measurement) MCA_HIP_REPAIR_ITEMS: work items above which the repair contraction halves its K split (0: never)
    int adapt_max_sources = 1;         // MCA_HIP_ADAPT_MAX_SOURCES / cfg.adaptive_max_sources
    double tau_scale = 1.0;            // MCA_HIP_ADAPT_TAU_SCALE (tools/adaptive_check.py: 1e9 turns the repair off to measure the coarse error)
    long long ws_max_bytes = 4LL << 30;   // MCA_HIP_WS_MAX_MB: A-operand workspace budget per slice of frames (tests force the sliced path)
    bool force_generic = false;        // MCA_HIP_FORCE_GENERIC: the any-length kernels at N = 1024 too (parity test of both)
    bool scan_carry = false;           // MCA_HIP_SCAN_CARRY / cfg.scan_carry
    bool lazy_ks_shape = false;        // (measurement) MCA_HIP_LAZY_KS_SHAPE: the repair contraction's K segments by the call's shape also with lazy tails
    int cand = -1;                     // MCA_HIP_ADAPT_CAND: -1 / unset = by the back-off policy's reports (cand_call), 1 = wherever the call's shape allows, 0 = never (whole-row repair kernels)
    bool cand_fuse = true;             // (measurement) MCA_HIP_CAND_FUSE=0: k_srp_cand as a launch of its own
    int cand_grid = 512;               // (measurement) MCA_HIP_CAND_GRID: workgroups of k_srp_cand
    bool lazy_tails = true;            // MCA_HIP_ADAPT_LAZY=0: every adaptive call repairs its own last rows for the state it hands over (round 4)
    // measurement only (-DMCA_MEASURE)
    bool gemm_ks2 = false;             // MCA_HIP_GEMM_KS2: two K halves for 16 384 ... 32 767 rows (round 4) instead of four quarters
    bool no_merge = false, stft_wg = false, bf_ola = false, bf_occ2 = false, no_fused_partial = false, gemm_v1 = false, gemm_v2 = false,
         v1_nosplit = false, no_n512 = false, no_sub2 = false, no_n2048 = false, no_adapt_other_n = false;   // (no_n2048: MCA_HIP_NO_N2048, the any-length kernels at 2048-sample frames: A/B and parity of both)
    int spw_fpw = 0, bfw_ft = 0, bfw_abl = 0, bfw_var = 15, repair_ksplit = 0, v2_min_rows = 0, repick_grid = 256, list_grid = 512;
    bool dyn = false;                  // MCA_HIP_DYN: k_stft_phat_wave takes its runs off a device-side queue (round 5: measured slower, profiles/r05_run_queue_negative.log)
    int spw_waves = 4;                 // MCA_HIP_SPW_WAVES: 8 = the regular launches of k_stft_phat_wave as one workgroup of eight waves per CU
    int bfw_skew = -1;                 // MCA_HIP_BFW_SKEW: BeamformWaveArgs::skew (-1: the shipped rule, 0: off)
    int spw_skew = -1;                 // MCA_HIP_SPW_SKEW: StftPhatArgs::skew (-1: the shipped rule, 0: off)
    int spw_lds_pad = 0;               // MCA_HIP_SPW_LDS_PAD: KiB of unused LDS added to every k_stft_phat_wave launch (fewer workgroups per CU: occupancy A/B, tools/third_wave.sh)
    bool spw_xcd = false;              // MCA_HIP_SPW_XCD: StftPhatArgs::xcd_map
    bool no_balance = false;           // MCA_HIP_NO_BALANCE: StftPhatArgs::no_balance (the cost / the effect of pair_balance.h)
    bool dyn_flat = false;             // MCA_HIP_DYN_FLAT: every run of the queue has the first runs' length
    const char *wave_clock = nullptr;  // MCA_HIP_WAVE_CLOCK=<file>: entry / exit clocks of every wave of the last k_stft_phat_wave launch, written at destruction
    int dyn_len0 = 0;                  // MCA_HIP_DYN_LEN0: length of the first (longest) runs of the queue (0: half a wave's share, at most 16)
};

// Per-call workspace (everything a stream call allocates besides the per-array state, which lives in the context).  A call
// can be worked off in pieces over a sub-range of its arrays -- the chunks of the host-pointer path -- with the per-array
// state addressed at c->a0.  (Round 2 also split large device-pointer calls over 2-4 internal streams, "lanes"; measured 7 %
// slower with two and 57 % slower with three on the bench shape, HISTORY.md, so that code is gone.)
struct Workspace {
    void *d_A = nullptr; size_t a_bytes = 0;          // A operand: [rows][a_row_elems]
    void *d_Ax = nullptr; size_t ax_bytes = 0;        // ADAPTIVE: the two-plane A rows (repair pass, FP16X3 calls)
    float *d_C = nullptr; size_t c_bytes = 0;         // correlation map
    int c_planes = 1;                                 // partial maps (split-K) the last contraction left in d_C
    bool partial_done = false;                        // the last contraction also left the scan's chunk-local results (d_part, d_nv)
    int part_planes = 1;                              // ... one set per K half of a split-K launch
    long long c_plane = 0;
    float *d_Cx = nullptr; size_t cx_bytes = 0;       // ADAPTIVE: partial maps of the repair contraction
    // exact chunked scan + power gate
    float *d_part = nullptr, *d_estart = nullptr; int *d_nv = nullptr; size_t scan_ws_chunks = 0;
    float *d_power = nullptr; unsigned char *d_voiced = nullptr; float *d_power_out = nullptr; size_t gate_frames = 0;
    // ADAPTIVE: flags, repair list
    int *d_list_full = nullptr, *d_need_full = nullptr;   // two work lists (ScanPickArgs::list_full): the units of the frames that take whole rows
    unsigned char *d_flags = nullptr; int *d_chunk_from = nullptr, *d_list = nullptr, *d_need = nullptr; unsigned *d_umask = nullptr; size_t adapt_frames = 0, adapt_chunks = 0, adapt_groups = 0;
    unsigned char *d_unsure = nullptr;                // 16 microphones: frames whose DC / Nyquist bin the coarse analysis could not vouch for (StftPhatArgs::unsure)
    int *d_nlist = nullptr, *d_last_vchunk = nullptr; size_t lastv_arrays = 0;
    int last_a0 = 0, last_arrays = 0;                 // the arrays this lane ran in the last call (mca_hip_copy_gate)
    void release()
    {
        auto F = [](void *q) { if (q) (void)hipFree(q); };
        F(d_A); F(d_Ax); F(d_C); F(d_Cx); F(d_part); F(d_estart); F(d_nv); F(d_power); F(d_voiced); F(d_power_out);
        F(d_flags); F(d_chunk_from); F(d_list); F(d_need); F(d_umask); F(d_nlist); F(d_last_vchunk); F(d_unsure); F(d_list_full); F(d_need_full);
        *this = Workspace();
    }
};

struct mca_hip_ctx {
    mca_hip_config cfg{};
    std::vector<double> xyz;
    int M = 0, P = 0, G = 0, D = 0, Dp = 0, K = 0, N = 0, H = 0, logH = 0, Kp = 0, S = 1, prec = 0;
    bool ula = false, stream_ok = false, generic = false;
    bool n512 = false;             // 512-sample frames with <= 8 microphones: k_stft_phat_512 / k_beamform_512 instead of the any-length kernels
    bool n2048 = false;            // 2048-sample frames on the wave-level 1024-point transform (kernels_2048.hip): the beamformer for any M, the analysis for M <= 8 and > 2
    std::string stream_why;       // why the stream API is unavailable for this configuration
    int v2_min_rows = 16384;       // one operand plane; twice that with two (plan_gemm)
    float step = 0.f;
    std::vector<float> delays, grid;
    std::vector<int2> pairs;
    // device tables
    float *d_window = nullptr, *d_grid = nullptr, *d_delays = nullptr;
    float2 *d_tw = nullptr;       // [N/2] exp(-j 2 pi i / N), any-N kernels
    double *d_micx = nullptr;
    int2 *d_pairs = nullptr;
    void *d_B = nullptr, *d_Bt = nullptr;
    // ULA, one fp16 operand plane (the ADAPTIVE coarse pass, plain FP16), k_stft_phat_wave: the contraction index is the PRODUCT
    // m = k * (j - i) -- pairs of spacing g at bin k steer with exp(j 2 pi k g tau_1 / N), so all (k, g) of equal product share one
    // steering column and their PHAT sums are merged before the contraction: 1 962 instead of 3 591 complex terms per row for 8
    // microphones (build_merged_tables)
    bool merged = false; int n_merged = 0, Kp_m = 0;
    void *d_Bm = nullptr, *d_Btm = nullptr; unsigned short *d_mrank = nullptr;
    float2 *d_bftab = nullptr; int bf_pairs = 0;   // k_beamform_wave: steering rows per grid angle, [D + 1][bf_pairs][1024] (built on first use)
    // stream state (double buffered: kernels read [cur], write [cur^1])
    float *d_E[2] = {nullptr, nullptr};
    float *d_tail[2] = {nullptr, nullptr};
    int e_cur = 0, tail_cur = 0;
    Workspace lanes[MCA_MAX_LANES];
    int cur_lane = 0, a0 = 0;              // the workspace in use and the first array of the piece being enqueued (host side, sequential)
    long long plan_rows = 0;               // > 0: rows of the whole call while it is worked off in pieces (plan_gemm)
    int plan_arrays = 0;                   // ... and its arrays (repair_ksplit_for)
    int n_lanes_last = 1;
    hipStream_t io_stream[3] = {}; hipEvent_t io_ev[8] = {};   // host-pointer entry points with page-locked buffers: copy in / run / copy out
    Workspace &ws() { return lanes[cur_lane]; }
    double *d_gate_state = nullptr;
    int *d_last_bin = nullptr; float *d_last_rad = nullptr, *d_last_prob = nullptr;
    int last_arrays = 0, last_frames = 0;
    float *d_doa[2] = {nullptr, nullptr};   // 2-mic path: smoothed _currentDOA per array
    int doa_cur = 0;
    long long gcc2_frames_done = 0;         // (kept in the state header for blob compatibility; the counters below are what runs)
    long long *d_vdone[2] = {nullptr, nullptr};   // 2-mic path: frames that fired so far, per array (double-buffered with d_doa)
    int *d_g2_vidx = nullptr, *d_g2_nv = nullptr; float *d_g2_rad = nullptr, *d_g2_prob = nullptr; size_t g2_rows = 0;   // gated 2-mic path
    unsigned char *d_g2_reset = nullptr;    // [rows] per fired frame: the memory factors are zero (silence rule)
    int *d_g2_post0 = nullptr;              // [max_arrays] first frame of the call at which the floor estimate exists
    int *d_silence = nullptr;               // [max_arrays] _silenceFramesCounter (BinauralLocalisation.cpp:326), stream state
    // workspace
    unsigned long long ws_gen = 0;          // bumped whenever a workspace buffer is reallocated: recorded graphs hold the old pointers
    std::vector<mca_hip_graph *> graphs;   // live graphs of this context (orphaned by mca_hip_destroy)
    // adaptive SRP precision: fp16 coarse scan (one plane) + exact repair (hi + lo planes)
    int tab_planes = 1;            // planes of the steering tables (2: FP16X3 and ADAPTIVE)
    unsigned long long *d_rstats = nullptr;
    unsigned long long *d_wave_clock = nullptr; int wave_clock_n = 0;   // (measurement) StftPhatArgs::wave_clock of the last regular launch
    // lazy tails (mca_internal.h, HIST_FRAMES): what the last lazy adaptive call left for the next one, double buffered like d_E
    float *d_hist_pcm[2] = {nullptr, nullptr};     // [max_arrays][M][HIST_SAMPLES]
    float *d_hist_C[2] = {nullptr, nullptr};       // [max_arrays][HIST_FRAMES][Dp] coarse rows (patched where a repair pass recomputed them)
    float *d_ehist[2] = {nullptr, nullptr};        // [max_arrays][D] the energies in front of them
    int hist_cur = 0, hist_n = 0;                  // the buffers that hold the pending history, the arrays it covers
    bool hist_pending = false;                     // the state in d_E is coarse: exact = settle_history, or the next lazy call's own repair pass
    bool lazy_now = false;                         // (localise_impl) this call leaves its tails to the next: the coarse analysis keeps the PCM
    bool host_call = false;                        // inside a host-pointer entry point: its calls keep the eager form (the pageable and the page-locked path return the same bits)
    bool lazy_entry = false;                       // this API call may leave its tails to the next (device-pointer stream calls outside captures)
    unsigned *d_queue = nullptr;   // [16] run-queue words of the wave-per-run kernels (StftPhatArgs::queue), zero between launches
    int n_cu = 256;
    unsigned long long adapt_frames_total = 0;
    float tau_en = 0.f;            // normalised energies closer than this cannot be ordered from the coarse map
    // ADAPTIVE backs off to plain FP16X3 (the arithmetic its repair pass reproduces) while most rows need the repair -- noise
    // only, silence: every pick is a near tie and coarse + repair of everything costs twice the direct exact pass.  The last
    // kernel of an adaptive call leaves its running totals in page-locked memory; the next calls read them when they have arrived
    // (no synchronisation) and suspend the mode for fb_backoff eligible calls, then probe again with one adaptive call.
    Knobs kn;
    bool adapt_suspended = false, capturing = false;
    int fb_left = 0, fb_backoff = 8, fb_state = 0;      // (adapt_policy_begin)
    bool cand_heavy = false;                            // the last report had more than a fifth of the rows recomputed: whole-row repair kernels (adapt_policy_begin)
    unsigned long long fb_probe_seq = 0;
    unsigned long long *h_probe = nullptr;              // [FB_RING][4] one slot per adaptive call (seq - 1) % FB_RING: flagged frames, listed repair units, sequence number of the call
    unsigned long long fb_calls = 0, fb_seq_seen = 0, fb_groups_prev = 0, fb_frames_prev = 0, fb_flagged_prev = 0;
    unsigned long long fb_frames_ring[FB_RING] = {};    // adapt_frames_total after adaptive call number i + 1
    unsigned long long fb_m = 0;                        // eligible stream calls seen by adapt_policy_begin
    unsigned long long fb_launch_m[FB_RING] = {};       // ... and the one during which adaptive call number i + 1 was enqueued
    bool fb_lost = false;                               // a report did not arrive within FB_WAIT_MS: the policy stops consuming reports (until mca_hip_reset)
    int a_row_elems = 0, a_planes = 1, a_elem = 4;
    // frame API (double)
    double *d_fr = nullptr; size_t fr_elems = 0;
    double *d_E64[2] = {nullptr, nullptr}; int e64_cur = 0;
    double *d_res = nullptr;      // [S] doa, [S] prob, power
    int *d_bins = nullptr;
    double *d_out64 = nullptr; size_t out64_elems = 0;
    std::vector<double> h_stage;
    StagePool stage;              // device staging of the host-pointer entry points
    // timing
    unsigned timing = 0;                 // bit k: launches of kernel id k are bracketed with events
    bool timing_open = false;
    std::vector<TimedEvent> events;
    std::vector<hipEvent_t> pool;
    int t_launches[MCA_HIP_K_COUNT] = {};
    double t_ms[MCA_HIP_K_COUNT] = {};
    std::string err;
};

namespace {

int fail(mca_hip_ctx *c, int code, const std::string &msg)
{
    if (c) c->err = msg; else g_create_error = msg;
    return code;
}

#define HIP_TRY(ctx, expr)                                                                              \
    do {                                                                                                \
        hipError_t _e = (expr);                                                                         \
        if (_e != hipSuccess)                                                                           \
            return fail(ctx, _e == hipErrorOutOfMemory ? MCA_HIP_ERR_OUT_OF_MEMORY : MCA_HIP_ERR_HIP,   \
                        std::string(#expr) + ": " + hipGetErrorString(_e));                            \
    } while (0)

// ---- the reference's helper chain, host side (exact float/double sequence of
//      src/mcarray/microhponeArrayHelpers.cpp:38-120; [BUILD-DEFINES] double sin) ----
double speed_of_sound() { return 346.1; }                                           // :38-43
float doa_idx2angle(int idx, float step) { float pr = (float)idx * step; return (float)((double)pr - M_PI_2); }   // :117-120
float doa_to_delay_far_field(float doa, float microDist)                            // :46-67
{
    return (float)(((double)microDist * std::sin((double)doa)) / speed_of_sound());
}
float doa_to_delay_samples(float doa, float microDist, int fs) { return doa_to_delay_far_field(doa, microDist) * (float)fs; }   // :69-72
double distance(const std::vector<double> &xyz, int i, int j)                        // ArrayDescription.cpp:57-64
{
    return std::sqrt(std::pow(xyz[3 * j] - xyz[3 * i], 2) + std::pow(xyz[3 * j + 1] - xyz[3 * i + 1], 2) +
                     std::pow(xyz[3 * j + 2] - xyz[3 * i + 2], 2));
}

void free_ctx(mca_hip_ctx *c)
{
    if (!c) return;
    if (c->d_wave_clock && c->kn.wave_clock) {                      // (measurement builds only)
        std::vector<unsigned long long> h(3 * (size_t)c->wave_clock_n);
        if (hipDeviceSynchronize() == hipSuccess && hipMemcpy(h.data(), c->d_wave_clock, h.size() * 8, hipMemcpyDeviceToHost) == hipSuccess)
            if (FILE *f = std::fopen(c->kn.wave_clock, "w")) {
                for (int i = 0; i < c->wave_clock_n; ++i) std::fprintf(f, "%d %llu %llu %llu\n", i, h[3 * i], h[3 * i + 1], h[3 * i + 2]);
                std::fclose(f);
            }
        (void)hipFree(c->d_wave_clock);
    }
    auto F = [](void *p) { if (p) (void)hipFree(p); };
    F(c->d_window); F(c->d_tw); F(c->d_grid); F(c->d_delays); F(c->d_micx); F(c->d_pairs); F(c->d_B); F(c->d_Bt); F(c->d_bftab); F(c->d_Bm); F(c->d_Btm); F(c->d_mrank);
    F(c->d_E[0]); F(c->d_E[1]); F(c->d_tail[0]); F(c->d_tail[1]); F(c->d_doa[0]); F(c->d_doa[1]); F(c->d_vdone[0]); F(c->d_vdone[1]); F(c->d_g2_vidx); F(c->d_g2_nv); F(c->d_g2_rad); F(c->d_g2_prob);
    F(c->d_g2_reset); F(c->d_g2_post0); F(c->d_silence);
    F(c->d_rstats); F(c->d_gate_state); F(c->d_queue);
    for (int i = 0; i < 2; ++i) { F(c->d_hist_pcm[i]); F(c->d_hist_C[i]); F(c->d_ehist[i]); }
    if (c->h_probe) (void)hipHostFree(c->h_probe);
    for (Workspace &w : c->lanes) w.release();
    for (auto &e : c->io_ev) if (e) (void)hipEventDestroy(e);
    for (auto &q : c->io_stream) if (q) (void)hipStreamDestroy(q);
    F(c->d_last_bin); F(c->d_last_rad); F(c->d_last_prob);
    F(c->d_fr); F(c->d_E64[0]); F(c->d_E64[1]); F(c->d_res); F(c->d_bins); F(c->d_out64);
    c->stage.release();
    for (auto &e : c->events) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
    for (auto &e : c->pool) (void)hipEventDestroy(e);
    delete c;
}

int round_up(int v, int m) { return (v + m - 1) / m * m; }

// (the environment is read through knobs.h, by mca_hip_create only)
Knobs read_knobs(const mca_hip_config &cfg)
{
    Knobs k;
    auto geti = [](const char *v, long long dflt) { return v ? std::atoll(v) : dflt; };
    k.fb_enabled = cfg.adaptive_fallback != MCA_HIP_ADAPT_FALLBACK_OFF;
    if (const char *v = env_str("MCA_HIP_ADAPT_FALLBACK")) k.fb_enabled = std::atoi(v) != 0;
    // (4096 rows -- round 2: 8192; with the merged index the coarse contraction of 4096 rows takes 26 us against 81 + 12 us of the
    // three-product one: the literal BASELINE configs[2] call, 1 array x 4096 frames, is inside)
    k.adapt_min_rows = geti(env_str("MCA_HIP_ADAPT_MIN_ROWS"), cfg.adaptive_min_rows > 0 ? cfg.adaptive_min_rows : 4096);
    k.adapt_max_sources = (int)geti(env_str("MCA_HIP_ADAPT_MAX_SOURCES"), cfg.adaptive_max_sources > 0 ? cfg.adaptive_max_sources : 1);
    if (const char *v = env_str("MCA_HIP_ADAPT_TAU_SCALE")) k.tau_scale = std::atof(v);
    if (const char *v = env_str("MCA_HIP_WS_MAX_MB")) k.ws_max_bytes = (long long)std::atoll(v) << 20;
    k.force_generic = env_str("MCA_HIP_FORCE_GENERIC") != nullptr;
    k.scan_carry = cfg.scan_carry != 0 || env_str("MCA_HIP_SCAN_CARRY") != nullptr;
    if (const char *v = env_str("MCA_HIP_ADAPT_LAZY")) k.lazy_tails = std::atoi(v) != 0;
    // measurement only: constants unless the library was built with -DMCA_MEASURE
    k.no_merge = measure_env("MCA_HIP_NO_MERGE") != nullptr;
    k.repair_items = (int)geti(measure_env("MCA_HIP_REPAIR_ITEMS"), 768);
    if (const char *v = env_str("MCA_HIP_ADAPT_CAND")) k.cand = std::atoi(v) != 0 ? 1 : 0;
    k.cand_grid = (int)geti(measure_env("MCA_HIP_CAND_GRID"), 512);
    k.cand_fuse = geti(measure_env("MCA_HIP_CAND_FUSE"), 1) != 0;
    k.stft_wg = measure_env("MCA_HIP_STFT_WG") != nullptr;
    k.bf_ola = measure_env("MCA_HIP_BF_OLA") != nullptr;
    k.bf_occ2 = measure_env("MCA_HIP_BF_OCC2") != nullptr;
    k.no_fused_partial = measure_env("MCA_HIP_NO_FUSED_PARTIAL") != nullptr;
    k.gemm_v1 = measure_env("MCA_HIP_GEMM_V1") != nullptr;
    k.gemm_v2 = measure_env("MCA_HIP_GEMM_V2") != nullptr;
    k.gemm_ks2 = measure_env("MCA_HIP_GEMM_KS2") != nullptr;
    k.v1_nosplit = measure_env("MCA_HIP_V1_NOSPLIT") != nullptr;
    k.no_n512 = measure_env("MCA_HIP_NO_N512") != nullptr;
    k.no_n2048 = env_str("MCA_HIP_NO_N2048") != nullptr;
    k.no_sub2 = measure_env("MCA_HIP_NO_SUB2") != nullptr;
    k.no_adapt_other_n = measure_env("MCA_HIP_NO_ADAPT_OTHER_N") != nullptr;     // (A/B: 512- and 2048-sample frames as round 6 began, ADAPTIVE = FP16X3 there)
    k.spw_fpw = (int)geti(measure_env("MCA_HIP_SPW_FPW"), 0);
    k.bfw_ft = (int)geti(measure_env("MCA_HIP_BFW_FT"), 0);
    k.bfw_abl = (int)geti(measure_env("MCA_HIP_BFW_ABL"), 0);
    k.bfw_var = (int)geti(measure_env("MCA_HIP_BFW_VAR"), 15);
    k.repair_ksplit = (int)geti(measure_env("MCA_HIP_REPAIR_KSPLIT"), 0);
    k.v2_min_rows = (int)geti(measure_env("MCA_HIP_V2_MIN_ROWS"), 0);
    k.repick_grid = (int)geti(measure_env("MCA_HIP_REPICK_GRID"), 256);
    k.list_grid = (int)geti(measure_env("MCA_HIP_LIST_GRID"), 512);
    k.dyn = measure_env("MCA_HIP_DYN") != nullptr;
    k.dyn_len0 = (int)geti(measure_env("MCA_HIP_DYN_LEN0"), 0);
    k.dyn_flat = measure_env("MCA_HIP_DYN_FLAT") != nullptr;
    k.spw_xcd = measure_env("MCA_HIP_SPW_XCD") != nullptr;
    k.no_balance = measure_env("MCA_HIP_NO_BALANCE") != nullptr;
    k.spw_lds_pad = (int)geti(measure_env("MCA_HIP_SPW_LDS_PAD"), 0);
    k.lazy_ks_shape = measure_env("MCA_HIP_LAZY_KS_SHAPE") != nullptr;
    k.spw_skew = (int)geti(measure_env("MCA_HIP_SPW_SKEW"), -1);
    k.bfw_skew = (int)geti(measure_env("MCA_HIP_BFW_SKEW"), -1);
    k.spw_waves = (int)geti(measure_env("MCA_HIP_SPW_WAVES"), 4);
    k.wave_clock = measure_env("MCA_HIP_WAVE_CLOCK");
    return k;
}

// _currentDOA = 0, _prob = -1 (BeamformingSeparationAndLocalisation.cpp:51-52); no frame has fired yet -> bin -1
int init_last_state(mca_hip_ctx *c, hipStream_t st)
{
    const size_t n = (size_t)c->cfg.max_arrays * MCA_MAX_SOURCES;
    std::vector<int> b(n, -1); std::vector<float> pr(n, -1.f);
    HIP_TRY(c, hipMemcpyAsync(c->d_last_bin, b.data(), n * 4, hipMemcpyHostToDevice, st));
    HIP_TRY(c, hipMemsetAsync(c->d_last_rad, 0, n * 4, st));
    HIP_TRY(c, hipMemcpyAsync(c->d_last_prob, pr.data(), n * 4, hipMemcpyHostToDevice, st));
    HIP_TRY(c, hipMemsetAsync(c->d_gate_state, 0, (size_t)c->cfg.max_arrays * 4 * 8, st));
    HIP_TRY(c, hipStreamSynchronize(st));       // (the host vectors above go out of scope; mca_hip_reset returns with nothing queued)
    return MCA_HIP_OK;
}

// steering table B of the SRP contraction = precomputeTauMatrix (SteeringBeamforming.cpp:87-88)
// for the first pair of every delay group, laid out for the MFMA kernels.
int build_steering_table(mca_hip_ctx *c)
{
    const int K = c->K, D = c->D, Dp = c->Dp, Kp = c->Kp;
    std::vector<int> first_pair(c->G);
    if (c->ula) { for (int g = 0; g < c->G; ++g) first_pair[g] = g; /* pair (0, g+1) has index g */ }
    else for (int g = 0; g < c->G; ++g) first_pair[g] = g;
    const double N = 2.0 * (K - 1);
    if (c->prec == MCA_HIP_SRP_FP32) {
        std::vector<float> B((size_t)Kp * Dp, 0.f);
        for (int g = 0; g < c->G; ++g)
            for (int k = 0; k < K; ++k)
                for (int d = 0; d < D; ++d) {
                    double ph = 2.0 * M_PI * (double)k * (double)c->delays[(size_t)first_pair[g] * D + d] / N;
                    size_t r = ((size_t)g * K + k) * 2;
                    B[r * Dp + d] = (float)std::cos(ph);
                    B[(r + 1) * Dp + d] = (float)(-std::sin(ph));
                }
        HIP_TRY(c, hipMalloc(&c->d_B, B.size() * sizeof(float)));
        HIP_TRY(c, hipMemcpy(c->d_B, B.data(), B.size() * sizeof(float), hipMemcpyHostToDevice));
    } else {
        const int planes = c->tab_planes;
        std::vector<_Float16> B((size_t)planes * Dp * Kp, (_Float16)0.f);
        for (int g = 0; g < c->G; ++g)
            for (int d = 0; d < D; ++d)
                for (int k = 0; k < K; ++k) {
                    double ph = 2.0 * M_PI * (double)k * (double)c->delays[(size_t)first_pair[g] * D + d] / N;
                    float v[2] = {(float)std::cos(ph), (float)(-std::sin(ph))};
                    for (int q = 0; q < 2; ++q) {
                        size_t kk = ((size_t)g * K + k) * 2 + q;
                        _Float16 hi = (_Float16)v[q];
                        B[(size_t)d * Kp + kk] = hi;
                        if (planes == 2) B[((size_t)Dp + d) * Kp + kk] = (_Float16)(v[q] - (float)hi);
                    }
                }
        HIP_TRY(c, hipMalloc(&c->d_B, B.size() * sizeof(_Float16)));
        HIP_TRY(c, hipMemcpy(c->d_B, B.data(), B.size() * sizeof(_Float16), hipMemcpyHostToDevice));
        if (Dp == 384) {
            // the same table tiled for the 256 x 384 kernel: one 32-deep K stage of all 384 columns is 24 KiB
            // contiguous, so every LDS-DMA instruction of that kernel moves one contiguous KiB
            const int ns32 = Kp / 32;
            std::vector<_Float16> Bt(B.size());
            for (int pl = 0; pl < planes; ++pl)
                for (int sl = 0; sl < ns32; ++sl)
                    for (int d = 0; d < Dp; ++d)
                        std::memcpy(&Bt[(((size_t)pl * ns32 + sl) * Dp + d) * 32], &B[((size_t)pl * Dp + d) * Kp + (size_t)sl * 32], 32 * sizeof(_Float16));
            HIP_TRY(c, hipMalloc(&c->d_Bt, Bt.size() * sizeof(_Float16)));
            HIP_TRY(c, hipMemcpy(c->d_Bt, Bt.data(), Bt.size() * sizeof(_Float16), hipMemcpyHostToDevice));
        }
    }
    return MCA_HIP_OK;
}

// the contraction depth (elements per operand plane) of the current call: the merged index with one plane on a merged context
int cur_kp(const mca_hip_ctx *c) { return (c->merged && c->a_planes == 1) ? c->Kp_m : c->Kp; }

// Merged contraction index of the one-plane fp16 path on a uniform linear array (see mca_hip_ctx::merged).  The pairs of
// spacing g have the delay table tau_g(d); on a ULA tau_g = g tau_1 up to the rounding of the reference's float chain
// (doaToDelayFarFieldSamples, SteeringBeamforming.cpp:73), so exp(j 2 pi k tau_g / N) = exp(j 2 pi (k g) tau_1 / N) to ~1e-5
// rad -- a twentieth of the fp16 rounding of the operands this path carries anyway; checked here, else no merging.  The exact
// paths (FP32, FP16X3, the repair pass) keep the per-group index and the reference's own tables.
int build_merged_tables(mca_hip_ctx *c)
{
    c->merged = false;
    // (round 6: also at 2048-sample frames, k_stft_phat_2048<..., MERGE>)
    if (!c->ula || (c->generic && !c->n2048) || c->n512 || (c->M != 4 && c->M != 8) || c->Dp % 64 != 0 || c->kn.no_merge || c->kn.stft_wg) return MCA_HIP_OK;
    if (c->prec != MCA_HIP_SRP_ADAPTIVE && c->prec != MCA_HIP_SRP_FP16) return MCA_HIP_OK;
    const int K = c->K, D = c->D, Dp = c->Dp, G = c->G;
    const double N = 2.0 * (K - 1);
    for (int g = 1; g < G; ++g)
        for (int d = 0; d < D; ++d)
            if (std::fabs((double)c->delays[(size_t)g * D + d] - (g + 1) * (double)c->delays[d]) > 1e-4) return MCA_HIP_OK;   // (pair (0, g + 1) has index g)
    std::vector<unsigned short> rank((size_t)G * (K - 1) + 1, 0xffff);
    std::vector<int> ms;
    for (int g = 1; g <= G; ++g)
        for (int k = 0; k < K; ++k) rank[(size_t)k * g] = 0;
    for (size_t m = 0; m < rank.size(); ++m)
        if (rank[m] == 0) { rank[m] = (unsigned short)ms.size(); ms.push_back((int)m); }
    c->n_merged = (int)ms.size();
    c->Kp_m = round_up(2 * ((c->n_merged + 63) & ~63), 32);       // (whole wave rows: the analysis kernel stores its LDS region as it stands, zeros behind n_merged)
    const int Kp = c->Kp_m;
    std::vector<_Float16> B((size_t)Dp * Kp, (_Float16)0.f);
    for (int d = 0; d < D; ++d)
        for (int r = 0; r < c->n_merged; ++r) {
            const double ph = 2.0 * M_PI * (double)ms[(size_t)r] * (double)c->delays[d] / N;
            B[(size_t)d * Kp + 2 * r] = (_Float16)(float)std::cos(ph);
            B[(size_t)d * Kp + 2 * r + 1] = (_Float16)(float)(-std::sin(ph));
        }
    HIP_TRY(c, hipMalloc(&c->d_Bm, B.size() * sizeof(_Float16)));
    HIP_TRY(c, hipMemcpy(c->d_Bm, B.data(), B.size() * sizeof(_Float16), hipMemcpyHostToDevice));
    if (Dp == 384) {
        const int ns32 = Kp / 32;
        std::vector<_Float16> Bt(B.size());
        for (int sl = 0; sl < ns32; ++sl)
            for (int d = 0; d < Dp; ++d)
                std::memcpy(&Bt[((size_t)sl * Dp + d) * 32], &B[(size_t)d * Kp + (size_t)sl * 32], 32 * sizeof(_Float16));
        HIP_TRY(c, hipMalloc(&c->d_Btm, Bt.size() * sizeof(_Float16)));
        HIP_TRY(c, hipMemcpy(c->d_Btm, Bt.data(), Bt.size() * sizeof(_Float16), hipMemcpyHostToDevice));
    }
    rank.resize((rank.size() + 3) / 4 * 4, 0xffff);
    HIP_TRY(c, hipMalloc((void **)&c->d_mrank, rank.size() * sizeof(unsigned short)));
    HIP_TRY(c, hipMemcpy(c->d_mrank, rank.data(), rank.size() * sizeof(unsigned short), hipMemcpyHostToDevice));
    c->merged = true;
    c->a_row_elems = cur_kp(c) * c->a_planes;
    return MCA_HIP_OK;
}

void time_begin(mca_hip_ctx *c, int id, hipStream_t st)
{
    c->timing_open = (c->timing >> id) & 1u;
    if (!c->timing_open) return;
    TimedEvent ev; ev.id = id;
    auto get = [&]() { hipEvent_t e; if (!c->pool.empty()) { e = c->pool.back(); c->pool.pop_back(); } else (void)hipEventCreate(&e); return e; };
    ev.a = get(); ev.b = get();
    (void)hipEventRecord(ev.a, st);
    c->events.push_back(ev);
}
void time_end(mca_hip_ctx *c, hipStream_t st)
{
    if (!c->timing_open) return;
    c->timing_open = false;
    (void)hipEventRecord(c->events.back().b, st);
}

// threads per workgroup of the any-length kernels, measured per frame length (a workgroup's spectra fill half a
// CU's LDS from N = 2048 on, so more waves per workgroup are the only way to more waves per CU): analysis
// 256 / 1024 / 1024, synthesis 256 / 512 / 1024 threads for N <= 1024 / 2048 / 4096+
int gen_threads(const mca_hip_ctx *c, bool synthesis)
{
    if (c->N >= 4096) return 1024;
    if (c->N >= 2048) return synthesis ? 512 : 1024;
    return 256;
}

// which contraction kernel a chunk of `rows` frames runs on, and over how many workgroups its K range is split
struct GemmPlan { bool v2; int ksplit; };
GemmPlan plan_gemm(const mca_hip_ctx *c, long long rows)
{
    GemmPlan g;
    // a call that is worked off in pieces (the chunks of the host-pointer path) plans every piece as the whole call would
    // be planned: a row's result then does not depend on how the call was cut (same kernel, same K segments, same order)
    if (c->plan_rows > 0) rows = c->plan_rows;
    // 256 x 384 tiles from 16 384 rows with one operand plane (128 workgroups of k_srp_gemm_f16_v3 beat the 128 x 192 kernel
    // there: 134 vs 156 us, and leave the scan's chunk results), from 32 768 rows with the hi + lo planes (343 vs 299 us at 16 384)
    // (round 6: a contraction twice as deep -- 2048-sample frames, 16 microphones: 14 350 / 15 390 elements per plane -- gives a 256 x 384
    // workgroup of a K QUARTER the work a K half has at 7 182: two planes take the big tiles from 16 384 rows there, profiles/r06_n2048.log)
    const long long deep = cur_kp(c) >= 12288 ? 2 : 1;
    g.v2 = c->prec != MCA_HIP_SRP_FP32 && c->Dp == 384 && rows * (c->a_planes == 2 ? deep : 1) >= (long long)c->v2_min_rows * (c->a_planes == 2 ? 2 : 1) && !c->kn.gemm_v1;
    if (g.v2) {
        // 256 x 384 tiles need >= ~256 workgroups to fill the chip: one K range from 65 536 rows, two halves from 32 768, and below that four
        // quarters where the contraction is deep (round 5: 16 microphones, 15 392 terms per row -- 16 384 rows were 128 workgroups on half
        // the CUs: 0.280 -> 0.183 ms + 0.021 for k_sum_planes, which folds the four partial maps before the scan; with the 3 968 merged terms
        // of 8 microphones the fold costs what the quarters save: 75 -> 54 + 22 us, profiles/r05_gemm_ksplit4.log)
        g.ksplit = rows >= 65536 ? 1 : (rows >= 32768 || c->kn.gemm_ks2 || cur_kp(c) < 6144 ? 2 : 4);      // (6144: the merged rows of 2048-sample frames, 7 936 deep, take quarters)
    } else {
        // 128 x 192 tiles: small batches (a single stream) would leave most CUs idle and walk the whole K range
        // in a handful of workgroups (0.29 ms however few frames); split K until ~512 workgroups exist, keeping
        // at least 8 K steps per workgroup
        const long long wgs = (rows + 127) / 128 * (c->Dp == 64 ? 1 : c->Dp / 192);
        const int nk = cur_kp(c) / (c->prec == MCA_HIP_SRP_FP32 ? 16 : 32);
        long long ks = 512 / (wgs > 0 ? wgs : 1);
        if (ks > nk / 8) ks = nk / 8;
        if (ks > 16) ks = 16;
        if (ks < 1) ks = 1;
        g.ksplit = c->kn.v1_nosplit ? 1 : (int)ks;
    }
    return g;
}

// the A buffer of the current call: an ADAPTIVE context keeps its two-plane rows (repair pass, FP16X3 calls) apart from the
// one-plane rows of the coarse pass, so that a buffer only ever sees one row layout (its Kp padding columns stay zero)
void *&a_buf(mca_hip_ctx *c) { return (c->prec == MCA_HIP_SRP_ADAPTIVE && c->a_planes == 2) ? c->ws().d_Ax : c->ws().d_A; }
size_t &a_buf_bytes(mca_hip_ctx *c) { return (c->prec == MCA_HIP_SRP_ADAPTIVE && c->a_planes == 2) ? c->ws().ax_bytes : c->ws().a_bytes; }

int ensure_a(mca_hip_ctx *c, long long rows)
{
    // rows rounded up to the 256-row tile of the split-K MFMA kernel, which loads whole tiles unclamped
    size_t need_a = (size_t)((rows + 255) / 256 * 256) * c->a_row_elems * c->a_elem;
    void *&buf = a_buf(c); size_t &bytes = a_buf_bytes(c);
    if (need_a > bytes) {
        if (buf) (void)hipFree(buf);
        buf = nullptr; bytes = 0;
        HIP_TRY(c, hipMalloc(&buf, need_a));
        HIP_TRY(c, hipMemset(buf, 0, need_a));          // the Kp padding columns stay zero forever
        HIP_TRY(c, hipDeviceSynchronize());             // (the caller's stream need not wait for the null stream's memset by itself)
        bytes = need_a; ++c->ws_gen;
    }
    return MCA_HIP_OK;
}

int ensure_workspace(mca_hip_ctx *c, long long rows_chunk, long long rows_total)
{
    int rc_a = ensure_a(c, rows_chunk);
    if (rc_a) return rc_a;
    const int planes = std::max(2, plan_gemm(c, rows_chunk).ksplit);
    size_t need_c = (size_t)rows_total * c->Dp * sizeof(float) * planes;   // room for the partial maps of a split-K contraction
    if (need_c > c->ws().c_bytes) {
        if (c->ws().d_C) (void)hipFree(c->ws().d_C);
        c->ws().d_C = nullptr; c->ws().c_bytes = 0;
        HIP_TRY(c, hipMalloc((void **)&c->ws().d_C, need_c));
        c->ws().c_bytes = need_c; ++c->ws_gen;
    }
    return MCA_HIP_OK;
}

int ensure_scan_workspace(mca_hip_ctx *c, int n_arrays, int n_frames, int n_chunks)
{
    const size_t need = (size_t)n_arrays * n_chunks;
    if (need > c->ws().scan_ws_chunks) {
        if (c->ws().d_part) (void)hipFree(c->ws().d_part);
        if (c->ws().d_estart) (void)hipFree(c->ws().d_estart);
        if (c->ws().d_nv) (void)hipFree(c->ws().d_nv);
        c->ws().d_part = c->ws().d_estart = nullptr; c->ws().d_nv = nullptr; c->ws().scan_ws_chunks = 0;
        HIP_TRY(c, hipMalloc((void **)&c->ws().d_part, need * c->D * 4 * 2));      // (two sets when a split-K contraction leaves them)
        HIP_TRY(c, hipMalloc((void **)&c->ws().d_estart, need * c->D * 4));
        HIP_TRY(c, hipMalloc((void **)&c->ws().d_nv, need * 4));
        c->ws().scan_ws_chunks = need; ++c->ws_gen;
    }
    const size_t nf = (size_t)n_arrays * n_frames;
    if (c->cfg.use_power_floor && nf > c->ws().gate_frames) {
        if (c->ws().d_power) (void)hipFree(c->ws().d_power);
        if (c->ws().d_voiced) (void)hipFree(c->ws().d_voiced);
        if (c->ws().d_power_out) (void)hipFree(c->ws().d_power_out);
        c->ws().d_power = c->ws().d_power_out = nullptr; c->ws().d_voiced = nullptr; c->ws().gate_frames = 0;
        HIP_TRY(c, hipMalloc((void **)&c->ws().d_power, nf * 4));
        HIP_TRY(c, hipMalloc((void **)&c->ws().d_power_out, nf * 4));
        HIP_TRY(c, hipMalloc((void **)&c->ws().d_voiced, nf));
        c->ws().gate_frames = nf; ++c->ws_gen;
    }
    return MCA_HIP_OK;
}

// A-operand workspace budget per slice of frames (MCA_HIP_WS_MAX_MB lets the tests force the sliced path)
long long ws_max_bytes(const mca_hip_ctx *c) { return c->kn.ws_max_bytes; }

long long chunk_frames_for(const mca_hip_ctx *c, int n_arrays, int n_frames)
{
    long long row_bytes = (long long)c->a_row_elems * c->a_elem;
    long long rows_cap = ws_max_bytes(c) / row_bytes;
    long long fc = rows_cap / n_arrays;
    if (fc >= n_frames) return n_frames;
    fc = fc / 8 * 8;
    if (fc < 8) fc = 8;
    return fc < n_frames ? fc : n_frames;
}

// ---- adaptive SRP precision (MCA_HIP_SRP_ADAPTIVE) ------------------------------------------------------------
void set_call_planes(mca_hip_ctx *c, int planes) { c->a_planes = planes; c->a_row_elems = cur_kp(c) * planes; }

// Does a call of this shape run coarse + repair?  The repair pass needs the list mode of k_stft_phat (1024-sample frames,
// more than two microphones) and has a fixed cost of a few small launches, so small batches -- which are latency bound
// whatever the precision -- run as plain FP16X3, which is what the repair pass reproduces.  (Gated streams do take the
// adaptive path: the planning wave lists the last 25 VOICED rows behind a flagged frame)
bool adaptive_shape(const mca_hip_ctx *c, int n_arrays, int n_frames)
{
    const long long rows = c->plan_rows > 0 ? c->plan_rows : (long long)n_arrays * n_frames;     // (see plan_gemm)
    // (more than one source: the S-th pick is a weak peak more often than not -- a second source, or noise when fewer than S
    // are active -- and a third to all of the frames are flagged: 8 x 4096 frames with S = 2 / 3 / 4 real sources spend 0.89 / 1.60 /
    // 1.76 ms in the repair pass, more than the 0.3 ms the coarse contraction saves; MCA_HIP_ADAPT_MAX_SOURCES lifts the limit)
    return c->prec == MCA_HIP_SRP_ADAPTIVE && ((!c->generic && !c->n512 && c->N == FFT_N) || ((c->n2048 || c->n512) && c->M <= 8 && !c->kn.no_adapt_other_n)) && c->M > 2 && c->S <= c->kn.adapt_max_sources &&
           rows >= c->kn.adapt_min_rows && n_frames >= 2 * SCAN_CHUNK;
}
bool adaptive_applies(const mca_hip_ctx *c, int n_arrays, int n_frames) { return adaptive_shape(c, n_arrays, n_frames) && !c->adapt_suspended; }
// lazy tails: contexts whose adaptive calls may leave the repair of their last rows to the next call -- no power gate (the rows a frame
// depends on are then simply the 16 before it), the wave-per-run analysis (4 / 8 microphones at 1024 samples: it keeps the PCM and reads it back)
bool lazy_context(const mca_hip_ctx *c)
{
    return c->prec == MCA_HIP_SRP_ADAPTIVE && c->kn.lazy_tails && !c->cfg.use_power_floor && (c->M == 8 || c->M == 4) && !c->generic && !c->n512 &&
           !c->kn.stft_wg && c->cfg.gcc_weighting == MCA_HIP_GCC_PHAT && c->stream_ok;
}

// Candidate columns (k_srp_cand) for this call?  Lazy calls (no frame is repeated for the state's sake) of contexts whose coarse analysis
// marks no unsure rows, one source (wave_candidates bounds the first pick) -- a unit that wants every column costs k_srp_cand the whole
// steering table.  Only under the AUTO back-off policy, and not while it is probing or the last report was heavy (noise only: 45 % of the
// frames flagged, ten columns each -- 4.3 ms per call against 2.0 for the whole-row kernels): a context pinned to the mode
// (adaptive_fallback OFF: bit-reproducible runs) keeps the whole-row kernels whatever the content.  MCA_HIP_ADAPT_CAND=1 / 0 overrides the
// policy (always where the shape allows / never).
static bool wave16_applies(const mca_hip_ctx *c);
static bool cand_call(const mca_hip_ctx *c, bool lazy)
{
    if (c->kn.cand == 0 || !lazy || wave16_applies(c) || c->S != 1 || c->Dp / 32 > CAND_WORDS_MAX) return false;
    if (c->kn.cand == 1) return true;                                 // MCA_HIP_ADAPT_CAND=1: whatever the content (tests, sweeps, pinned contexts that want it)
    return c->kn.fb_enabled && c->h_probe && c->fb_state == 0 && !c->cand_heavy;
}

// Two work lists (round 6; k_scan_pick<PL, 2>): contexts whose flagged frames include whole rows BY CONSTRUCTION -- the 16-microphone ULA: eager
// tails (its coarse kernel keeps no PCM) and unsure rows -- send those frames' units to the whole-row kernels and everything else to the
// candidate kernel (k_srp_cand as a launch of its own: the exact rows of such an array come from k_stft_phat<16>).  Ungated calls only (the plan
// in LDS assumes every frame advances the recursion).  MEASURED NEGATIVE (profiles/r06_m16_two_lists_negative.log: repair 0.186 -> 0.193 ms at
// 8 x 2 048 frames -- the exact analysis, 69 us of k_stft_phat<16> at one workgroup per CU, runs once per list and the second chain's launches
// cost what the narrower contraction saves): only with MCA_HIP_ADAPT_CAND=1 ("candidate columns wherever the call's shape allows"), never by
// the policy; the parity test of the mode runs it that way.
static bool cand_mixed(const mca_hip_ctx *c, bool lazy, bool gate)
{
    return c->kn.cand == 1 && !lazy && !gate && wave16_applies(c) && c->S == 1 && c->Dp / 32 <= CAND_WORDS_MAX;
}

// Called once at the top of an eager stream call (not per piece of a call, not while a graph is recorded): consumes the reports of
// the adaptive calls that were enqueued FB_LAG eligible calls ago (or earlier) and decides whether this call runs coarse + repair or
// plain FP16X3.
//   NORMAL     adaptive; a report of more than 30 % of the rows recomputed (or of the frames flagged) since the last one suspends the mode for fb_backoff
//              eligible calls (8, doubling up to 256 while the probes keep reporting that)
//   SUSPENDED  plain FP16X3, counting down; then ONE adaptive call probes
//   WAITING    plain FP16X3 until the probe's report is consumed (FB_LAG calls after the probe): heavy again -> SUSPENDED, else NORMAL
// Round 6: every adaptive call has its own report slot and a report is consumed by the call FB_LAG calls after its
// own, which WAITS for it if need be -- before, a call took "whatever has arrived by now" and the switch (and with it the choice between
// candidate columns and whole rows, cand_call) could move by a call from run to run.  A report that does not arrive within FB_WAIT_MS (a
// stream held up by something only this thread would release) ends the policy for the stream: the mode stays what it is.
void adapt_policy_begin(mca_hip_ctx *c, int n_arrays, int n_frames)
{
    if (!c->kn.fb_enabled || c->capturing || !c->h_probe || !adaptive_shape(c, n_arrays, n_frames)) return;
    bool fresh = false, heavy = false;
    const unsigned long long m = c->fb_m++;
    unsigned long long seq = c->fb_seq_seen;                                   // the last report due at this call
    while (seq < c->fb_calls && c->fb_launch_m[seq % FB_RING] + FB_LAG <= m) ++seq;      // (call number seq + 1 lives at index seq % FB_RING)
    if (seq > c->fb_seq_seen && !c->fb_lost && c->fb_calls - seq < FB_RING - 8) {
        const unsigned long long *slot = c->h_probe + 4 * ((seq - 1) % FB_RING);
        const auto t0 = std::chrono::steady_clock::now();
        while (__atomic_load_n(&slot[2], __ATOMIC_ACQUIRE) != seq) {
            if (std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t0).count() > FB_WAIT_MS) { c->fb_lost = true; break; }
            std::this_thread::yield();
        }
        if (!c->fb_lost) {
        const unsigned long long groups = __atomic_load_n(&slot[1], __ATOMIC_RELAXED), frames = c->fb_frames_ring[(seq - 1) % FB_RING];
        const unsigned long long flagged = __atomic_load_n(&slot[0], __ATOMIC_RELAXED);
        const bool fwd = groups >= c->fb_groups_prev && frames > c->fb_frames_prev && flagged >= c->fb_flagged_prev;      // (the totals restart with mca_hip_reset_timing)
        const unsigned long long dg = fwd ? groups - c->fb_groups_prev : 0, df = fwd ? frames - c->fb_frames_prev : 0, dfl = fwd ? flagged - c->fb_flagged_prev : 0;
        c->fb_seq_seen = seq; c->fb_groups_prev = groups; c->fb_frames_prev = frames; c->fb_flagged_prev = flagged;
        fresh = df > 0;
        // (... or 30 % of the frames flagged: digital silence lists no rows -- they are zero in both maps -- but every frame is planned and picked twice)
        heavy = fresh && (dg * REPAIR_GROUP * 100 > df * 30 || dfl * 100 > df * 30);
        // candidate columns pay while the list is short (one workgroup per unit); a call that recomputes most rows is a dense contraction again
        if (fresh) c->cand_heavy = dg * REPAIR_GROUP * 100 > df * 20;
        }
    } else if (seq > c->fb_seq_seen) {
        c->fb_seq_seen = seq;                                                  // (more calls in flight than slots, or the policy has ended: dropped)
    }
    auto suspend = [&]() {
        c->fb_state = 1; c->adapt_suspended = true;
        c->fb_left = c->fb_backoff - 1;                                                   // (this call is the first of them)
        c->fb_backoff = std::min(c->fb_backoff * 2, 256);
    };
    switch (c->fb_state) {
    case 0:
        if (heavy) suspend(); else if (fresh) c->fb_backoff = 8;
        break;
    case 1:
        if (--c->fb_left < 0) { c->fb_state = 2; c->adapt_suspended = false; c->fb_probe_seq = c->fb_calls + 1; }   // this call probes
        break;
    default:
        c->adapt_suspended = true;
        if (fresh && seq >= c->fb_probe_seq) {
            if (heavy) suspend(); else { c->fb_state = 0; c->adapt_suspended = false; c->fb_backoff = 8; }
        }
        break;
    }
}

// K segments of the repair contraction AT MOST: by the shape of the (whole) call (a long list lowers the number on the device,
// repair_ksplit_eff).  The tails of every array -- REPAIR_WARM + 1 rows
// rounded up to repair units -- are always recomputed: a few arrays leave a few hundred rows, where 32 segments fill the chip;
// 128 arrays leave thousands, where 32 partial maps per row cost more (k_repair_patch reads them all) than they buy.
int repair_ksplit_for(const mca_hip_ctx *c, int n_arrays)
{
    if (c->kn.repair_ksplit > 0) return std::min(c->kn.repair_ksplit, REPAIR_KSPLIT_MAX);
    // lazy tails: no array lists its last rows as a matter of course -- the list is as long as the content makes it, and the device-side
    // rule (repair_ksplit_eff) halves the segments when it is long
    if (lazy_context(c) && !c->kn.lazy_ks_shape) return REPAIR_KSPLIT_MAX;
    const long long arrays = c->plan_arrays > 0 ? c->plan_arrays : n_arrays;
    const long long tail_rows = arrays * (REPAIR_WARM + 1 + REPAIR_GROUP);
    return tail_rows <= 1024 ? 32 : (tail_rows <= 2560 ? 16 : 8);
}

// rows of one repair pass (the two-plane A rows of all listed groups may not fit the workspace budget at once)
long long repair_pass_rows(const mca_hip_ctx *c, int n_arrays, int n_frames)
{
    const long long gpa = (n_frames + REPAIR_GROUP - 1) / REPAIR_GROUP + HIST_UNITS;     // (+ the history units of the lazy tails)
    const long long all = (long long)n_arrays * gpa * REPAIR_GROUP;
    long long cap = ws_max_bytes(c) / ((long long)2 * c->Kp * 2) / 128 * 128;
    // (the repair contraction always leaves REPAIR_KSPLIT partial maps: Cx is sized for the worst case, all rows listed --
    // 1.6 GB next to the 1.9 GB of two-plane A rows on the bench shape; one pass in the common case, so no empty launches)
    if (cap < 128) cap = 128;
    return std::min((all + 127) / 128 * 128, cap);
}

int ensure_adapt_workspace(mca_hip_ctx *c, int n_arrays, int n_frames, int n_chunks)
{
    const size_t nf = (size_t)n_arrays * n_frames, nc = (size_t)n_arrays * n_chunks;
    const size_t ng = (size_t)n_arrays * ((n_frames + REPAIR_GROUP - 1) / REPAIR_GROUP + HIST_UNITS);     // (+ the history units of the lazy tails)
    if (nf > c->ws().adapt_frames) {
        if (c->ws().d_flags) (void)hipFree(c->ws().d_flags);
        c->ws().d_flags = nullptr; c->ws().adapt_frames = 0;
        HIP_TRY(c, hipMalloc((void **)&c->ws().d_flags, nf));
        if (c->ws().d_unsure) (void)hipFree(c->ws().d_unsure);
        c->ws().d_unsure = nullptr;
        HIP_TRY(c, hipMalloc((void **)&c->ws().d_unsure, nf));
        c->ws().adapt_frames = nf; ++c->ws_gen;
    }
    if (nc > c->ws().adapt_chunks) {
        if (c->ws().d_chunk_from) (void)hipFree(c->ws().d_chunk_from);
        c->ws().d_chunk_from = nullptr; c->ws().adapt_chunks = 0;
        HIP_TRY(c, hipMalloc((void **)&c->ws().d_chunk_from, 2 * nc * 4));   // + the list of the chunks that hold a flagged frame
        HIP_TRY(c, hipMemset(c->ws().d_chunk_from, 0x7f, nc * 4));       // "no flagged frame"; kept so by k_scan_repick
        HIP_TRY(c, hipDeviceSynchronize());
        c->ws().adapt_chunks = nc; ++c->ws_gen;
    }
    if (ng > c->ws().adapt_groups) {
        if (c->ws().d_list) (void)hipFree(c->ws().d_list);
        if (c->ws().d_need) (void)hipFree(c->ws().d_need);
        c->ws().d_list = c->ws().d_need = nullptr; c->ws().adapt_groups = 0;
        HIP_TRY(c, hipMalloc((void **)&c->ws().d_list, ng * 4));
        HIP_TRY(c, hipMalloc((void **)&c->ws().d_need, ng * 4));
        HIP_TRY(c, hipMemset(c->ws().d_need, 0, ng * 4));                 // test-and-set words; released by k_repair_patch / k_scan_repick
        if (c->ws().d_list_full) (void)hipFree(c->ws().d_list_full);
        if (c->ws().d_need_full) (void)hipFree(c->ws().d_need_full);
        c->ws().d_list_full = c->ws().d_need_full = nullptr;
        HIP_TRY(c, hipMalloc((void **)&c->ws().d_list_full, ng * 4));
        HIP_TRY(c, hipMalloc((void **)&c->ws().d_need_full, ng * 4));
        HIP_TRY(c, hipMemset(c->ws().d_need_full, 0, ng * 4));
        if (c->ws().d_umask) (void)hipFree(c->ws().d_umask);
        c->ws().d_umask = nullptr;
        HIP_TRY(c, hipMalloc((void **)&c->ws().d_umask, ng * (c->Dp / 32) * 4));
        HIP_TRY(c, hipMemset(c->ws().d_umask, 0, ng * (c->Dp / 32) * 4));   // candidate columns per unit; cleared by k_scan_repick
        HIP_TRY(c, hipDeviceSynchronize());
        c->ws().adapt_groups = ng; ++c->ws_gen;
    }
    if ((size_t)n_arrays > c->ws().lastv_arrays) {
        if (c->ws().d_last_vchunk) (void)hipFree(c->ws().d_last_vchunk);
        c->ws().d_last_vchunk = nullptr; c->ws().lastv_arrays = 0;
        HIP_TRY(c, hipMalloc((void **)&c->ws().d_last_vchunk, (size_t)n_arrays * 4));
        c->ws().lastv_arrays = n_arrays; ++c->ws_gen;
    }
    if (!c->ws().d_nlist) {
        HIP_TRY(c, hipMalloc((void **)&c->ws().d_nlist, 16));             // listed repair units | listed chunks, workgroups of k_scan_repick done
        HIP_TRY(c, hipMemset(c->ws().d_nlist, 0, 16));
        HIP_TRY(c, hipDeviceSynchronize());
        ++c->ws_gen;
    }
    const long long rows = repair_pass_rows(c, n_arrays, n_frames);
    const int planes = c->a_planes;
    set_call_planes(c, 2);
    const int rc = ensure_a(c, rows);
    set_call_planes(c, planes);
    if (rc) return rc;
    const size_t need_cx = (size_t)repair_cx_rows(rows, REPAIR_KSPLIT_MAX) * c->Dp * sizeof(float);
    if (need_cx > c->ws().cx_bytes) {
        if (c->ws().d_Cx) (void)hipFree(c->ws().d_Cx);
        c->ws().d_Cx = nullptr; c->ws().cx_bytes = 0;
        HIP_TRY(c, hipMalloc((void **)&c->ws().d_Cx, need_cx));
        c->ws().cx_bytes = need_cx; ++c->ws_gen;
    }
    return MCA_HIP_OK;
}

// The correctly rounded float reciprocal of d (a small positive integer): the candidate nearest 1/d in exact arithmetic
// (r d is exact in a double: 24 + <= 24 bits).  The scan kernels divide by d with it (normalised_energy).
static float exact_reciprocal(float d)
{
    const float r0 = (float)(1.0 / (double)d);
    float best = r0;
    double err = std::fabs(1.0 - (double)r0 * (double)d);
    const float cand[2] = {std::nextafterf(r0, 0.f), std::nextafterf(r0, 1.f)};
    for (float r : cand) {
        const double e = std::fabs(1.0 - (double)r * (double)d);
        if (e < err) { err = e; best = r; }
    }
    return best;
}

// the coarse (one fp16 plane) analysis of a 16-microphone uniform linear array runs on k_stft_phat_wave16
static bool wave16_applies(const mca_hip_ctx *c)
{
    return c->M == 16 && c->ula && !c->generic && c->cfg.gcc_weighting != MCA_HIP_GCC_NONE && !c->kn.stft_wg;
}

// 512-sample frames, up to 8 microphones: two frames of up to 8 channels per pass on the wave-level transform (k_stft_phat_512).
// grid.x = 0: one run of frames per workgroup, sized here; list mode (a.list): grid as given, REPAIR_GROUP frames per unit.
template <typename OutT>
int launch_stft_512(mca_hip_ctx *c, const StftPhatArgs &a, dim3 grid, hipStream_t st)
{
    StftPhatArgs sa = a;
    if (!sa.list) {
        sa.fpb = 8;
        while (sa.fpb > 2 && (long long)grid.y * ((sa.n_frames + sa.fpb - 1) / sa.fpb) < 512) sa.fpb >>= 1;
        grid.x = (sa.n_frames + sa.fpb - 1) / sa.fpb;
    }
    const size_t smem5 = ((size_t)2 * 8 * 258 + 8 * FFT_SCRATCH + TW_WIN + (size_t)sa.fpb * c->M) * sizeof(float2) + (size_t)sa.fpb * 8 * sizeof(float);
#define L512(MT, U)                                                                                                       \
    do {                                                                                                                  \
        HIP_TRY(c, hipFuncSetAttribute(reinterpret_cast<const void *>(&k_stft_phat_512<MT, U, OutT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem5)); \
        hipLaunchKernelGGL((k_stft_phat_512<MT, U, OutT>), grid, dim3(512), smem5, st, sa);                               \
    } while (0)
    if (c->M == 8 && c->ula) L512(8, true); else if (c->M == 8) L512(8, false);
    else if (c->M == 4 && c->ula) L512(4, true); else if (c->M == 4) L512(4, false);
    else if (c->ula) L512(0, true); else L512(0, false);
#undef L512
    HIP_TRY(c, hipGetLastError());
    return MCA_HIP_OK;
}

// 2048-sample frames, 3 .. 8 microphones: a wave per channel on the 1024-point complex transform + split, spectra in LDS, thread = two bins
// (k_stft_phat_2048).  grid.x = 0: one run of frames per workgroup, sized here; list mode (a.list): grid as given, REPAIR_GROUP frames per unit.
static bool n2048_analysis(const mca_hip_ctx *c) { return c->n2048 && c->M > 2 && c->M <= 8; }
template <typename OutT>
int launch_stft_2048(mca_hip_ctx *c, const StftPhatArgs &a, dim3 grid, hipStream_t st)
{
    StftPhatArgs sa = a;
    const int fp = c->M == 4 ? 2 : 1, mr = c->M == 4 ? 4 : 8;
    if (!sa.list) {
        sa.fpb = 16;
        while (sa.fpb > 2 && (long long)grid.y * ((sa.n_frames + sa.fpb - 1) / sa.fpb) < 512) sa.fpb >>= 1;
        grid.x = (sa.n_frames + sa.fpb - 1) / sa.fpb;
    }
    const size_t smem6 = ((size_t)fp * mr * 1026 + F1K_TWORDS + 8 * F1K_SCRATCH + (size_t)sa.fpb * c->M) * sizeof(float2) + (size_t)sa.fpb * 8 * sizeof(float);
#define L2048(MT, U)                                                                                                      \
    do {                                                                                                                  \
        HIP_TRY(c, hipFuncSetAttribute(reinterpret_cast<const void *>(&k_stft_phat_2048<MT, U, OutT, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem6)); \
        hipLaunchKernelGGL((k_stft_phat_2048<MT, U, OutT, false>), grid, dim3(512), smem6, st, sa);                              \
    } while (0)
    bool done = false;
    if constexpr (sizeof(OutT) == 2) {
        if (sa.mrank && sa.a_planes == 1 && !sa.list && c->ula && (c->M == 8 || c->M == 4)) {      // merged contraction index: ULA, one plane
            if (c->M == 8) {
                HIP_TRY(c, hipFuncSetAttribute(reinterpret_cast<const void *>(&k_stft_phat_2048<8, true, _Float16, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem6));
                hipLaunchKernelGGL((k_stft_phat_2048<8, true, _Float16, true>), grid, dim3(512), smem6, st, sa);
            } else {
                HIP_TRY(c, hipFuncSetAttribute(reinterpret_cast<const void *>(&k_stft_phat_2048<4, true, _Float16, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem6));
                hipLaunchKernelGGL((k_stft_phat_2048<4, true, _Float16, true>), grid, dim3(512), smem6, st, sa);
            }
            done = true;
        }
    }
    if (done) { }
    else if (c->M == 8 && c->ula) L2048(8, true); else if (c->M == 8) L2048(8, false);
    else if (c->M == 4 && c->ula) L2048(4, true); else if (c->M == 4) L2048(4, false);
    else if (c->ula) L2048(0, true); else L2048(0, false);
#undef L2048
    HIP_TRY(c, hipGetLastError());
    return MCA_HIP_OK;
}

template <typename OutT>
int launch_stft(mca_hip_ctx *c, const StftPhatArgs &a, dim3 grid, size_t smem, hipStream_t st)
{
    const int M = c->M; const bool ula = c->ula;
    if (n2048_analysis(c)) return launch_stft_2048<OutT>(c, a, grid, st);      // (the list-mode launches of the repair pass come through here)
    if (c->n512) return launch_stft_512<OutT>(c, a, grid, st);
    // a 16-microphone uniform linear array, one fp16 operand plane (the ADAPTIVE coarse pass, plain FP16): the wave-per-run kernel
    // with its whitened spectra packed to fp16 (k_stft_phat_wave16); the exact rows of such an array stay on k_stft_phat<16>
    if constexpr (sizeof(OutT) == 2) {
        if (wave16_applies(c) && a.a_planes == 1 && !a.list) {
            StftPhatArgs w = a;
            w.no_balance = c->kn.no_balance ? 1 : 0;
            w.fpb = 16;
            while (w.fpb > 1 && (long long)grid.y * ((a.n_frames + w.fpb - 1) / w.fpb) < 2048) w.fpb >>= 1;
            dim3 gw(((a.n_frames + w.fpb - 1) / w.fpb + 3) / 4, grid.y);
            // (one resident round of two workgroups per CU: the older workgroup's waves take more frames, StftPhatArgs::skew)
            if (c->kn.spw_skew != 0 && (gw.x & 1) == 0 && (long long)gw.x * gw.y == 2LL * c->n_cu && (long long)gw.x * 4 * w.fpb == a.n_frames) {
                const int sk = c->kn.spw_skew > 0 ? c->kn.spw_skew : (3 * w.fpb + 8) / 16;
                if (sk > 0 && sk < w.fpb) { w.skew = sk; gw = dim3(gw.y, gw.x); }
            }
            const int fpa = w.fpb + w.skew;
            const size_t smw = (size_t)(F1K_TWORDS + 4 * F1K_SCRATCH + 2 * 4 * fpa * 8) * sizeof(float2) + (1024 + 4 * fpa) * sizeof(float);   // (+ the scales and DC marks of unsure frames)
            if (a.power) {
                HIP_TRY(c, hipFuncSetAttribute(reinterpret_cast<const void *>(&k_stft_phat_wave16<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smw));
                hipLaunchKernelGGL(k_stft_phat_wave16<true>, gw, dim3(256), smw, st, w);
            } else {
                HIP_TRY(c, hipFuncSetAttribute(reinterpret_cast<const void *>(&k_stft_phat_wave16<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smw));
                hipLaunchKernelGGL(k_stft_phat_wave16<false>, gw, dim3(256), smw, st, w);
            }
            HIP_TRY(c, hipGetLastError());
            return MCA_HIP_OK;
        }
    }
    // 4 or 8 microphones: one wave per run of frames on the 1024-point transform of channel pairs (k_stft_phat_wave)
    if ((M == 8 || M == 4) && (a.no_phat || !c->kn.stft_wg)) {
        StftPhatArgs w = a;
        w.no_balance = c->kn.no_balance ? 1 : 0;
        dim3 gw;
        if (a.list) { w.fpb = 1; gw = dim3(grid.x, 1); }      // a listed group of REPAIR_GROUP = 4 frames per workgroup, a frame per wave
        else {
            const int env = c->kn.spw_fpw;
            w.fpb = env > 0 ? env : 16;       // frames per wave: two waves per SIMD want 2048 runs
            while (!env && w.fpb > 1 && (long long)grid.y * ((a.n_frames + w.fpb - 1) / w.fpb) < 2048) w.fpb >>= 1;
            gw = dim3(((a.n_frames + w.fpb - 1) / w.fpb + 3) / 4, grid.y);
        }
        // Dynamic runs (round 5, measurement builds only: MCA_HIP_DYN).  With one static run per wave the kernel is exactly one round of
        // resident waves and the average wave lives 85 % of the kernel (SQ counters), which looked like SIMDs waiting for the slowest wave.
        // The waves' exit clocks (MCA_HIP_WAVE_CLOCK) say otherwise: the spread is not between waves of equal standing -- the FIRST workgroup
        // of every CU (arrays 0..3 of the bench shape) is done at 180 us, the SECOND (arrays 4..7) at 244 us, 4 us apart inside a workgroup:
        // the CU issues oldest-first, the younger workgroup gets what is left and then runs alone at 83 % of the two-wave rate
        // (HISTORY.md section 8: one workgroup per CU).  Perfect balance is worth 11 of 290 us.  A queue of runs per WAVE (tickets in
        // arrival order) costs more than that: waves of a workgroup no longer stream neighbouring frames, and the same 16-frame runs take
        // 357 instead of 287 us; runs of 8 / 4 / 2 / 1 frames 357 / 343 / 375 / 515 us (profiles/r05_run_queue_negative.log).
        const int wg_per_cu = M == 4 ? 3 : 2;                                        // (by registers: 164 ... 175 / 233 ... 253)
        const long long share = a.list ? 0 : (long long)grid.y * a.n_frames / ((long long)wg_per_cu * c->n_cu * 4);   // frames per wave
        if (!a.list && c->kn.dyn && !c->kn.spw_fpw && share >= 8 && a.n_frames >= 64) {
            int len0 = c->kn.dyn_len0 > 0 ? c->kn.dyn_len0 : 16;
            while (!c->kn.dyn_len0 && len0 > 1 && len0 > share / 2) len0 >>= 1;
            w.queue = c->d_queue;
            w.q_sh0 = 0;
            while ((1 << (w.q_sh0 + 1)) <= len0) ++w.q_sh0;
            w.q_arrays = (int)grid.y;
            int t0, t1, t2, t3, t4;
            w.q_flat = c->kn.dyn_flat ? 1 : 0;
            w.q_total = dyn_run(0x7fffffff, a.n_frames, w.q_arrays, w.q_sh0, t0, t1, t2, t3, t4, w.q_flat);
            w.fpb = 1 << w.q_sh0;                                                          // (sizes the Nyquist slots of the unmerged kernels)
            gw = dim3(wg_per_cu * c->n_cu, 1);
        }
        w.xcd_map = c->kn.spw_xcd ? 1 : 0;
        // One resident round of two workgroups per CU (the bench shapes: 8 x 4096 and 128 x 256 frames of 8 microphones): the CU issues
        // oldest-first, so the workgroup it got first finishes early and the other then runs alone at 83 % of the two-wave rate.  The first
        // half of every array's run groups -- dispatched first -- takes more frames per wave (StftPhatArgs::skew).
        if (!a.list && !w.queue && c->kn.spw_skew != 0 && M == 8 && (gw.x & 1) == 0 && (long long)gw.x * gw.y == 2LL * c->n_cu &&
            (long long)gw.x * 4 * w.fpb == a.n_frames) {
            const int sk = c->kn.spw_skew > 0 ? c->kn.spw_skew : (3 * w.fpb + 8) / 16;      // 16 frames per wave: 19 / 13 (measured 1 ... 4: 3 is best)
            if (sk > 0 && sk < w.fpb) {
                w.skew = sk;
                gw = dim3(gw.y, gw.x);             // (arrays, run groups): the first half of the run groups of EVERY array is dispatched first
            }
        }
        // (measurement) MCA_HIP_SPW_WAVES=8: the same runs as ONE workgroup of eight waves per CU -- waves of one age, no skew needed
        int nwv = 4;
        if (!a.list && !w.queue && c->kn.spw_waves == 8 && (gw.x & 1) == 0 && !w.skew) { nwv = 8; gw = dim3(gw.x / 2, gw.y); }
        else if (!a.list && !w.queue && c->kn.spw_waves == 8 && w.skew) { w.skew = 0; gw = dim3(gw.y / 2, gw.x); nwv = 8; }   // (the skew rule had swapped the grid)
        if (c->kn.wave_clock && !a.list) {
            if (!c->d_wave_clock) HIP_TRY(c, hipMalloc((void **)&c->d_wave_clock, 3 * 8 * 16384));
            HIP_TRY(c, hipMemsetAsync(c->d_wave_clock, 0, 3 * 8 * 16384, st));
            c->wave_clock_n = std::min<int>(16384, (int)(gw.x * gw.y * 4));
            if ((int)(gw.x * gw.y * 4) <= 16384) w.wave_clock = c->d_wave_clock;
        }
        const bool mg = w.mrank != nullptr && !a.list;
        const int nrank = 2 * (M - 1) * 64 * 4 + 8, regw = mg ? std::max(F1K_SCRATCH, (w.n_merged + 63) & ~63) : F1K_SCRATCH;
        // (the merged kernel keeps its Nyquist bins in registers: with its 15.5 KiB regions two workgroups just fit the 160 KiB of a CU)
        const size_t smw = (size_t)(F1K_TWORDS + (mg ? nrank / 4 : 0) + nwv * regw + (mg ? 0 : nwv * (w.fpb + w.skew) * (M / 2))) * sizeof(float2) + (size_t)c->kn.spw_lds_pad * 1024;
        const bool pl2 = a.a_planes == 2, pw = a.power != nullptr;
#define LAUNCH_K(K)                                                                                                 \
        do {                                                                                                         \
            if (smw > 64 * 1024) HIP_TRY(c, hipFuncSetAttribute(reinterpret_cast<const void *>(&K), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smw)); \
            hipLaunchKernelGGL(K, gw, dim3(64 * nwv), smw, st, w);                                                   \
        } while (0)
#define LAUNCH_W2(MT, U, PL2, NP)                                                                                  \
        do {                                                                                                         \
            if (pw) LAUNCH_K((k_stft_phat_wave<MT, U, OutT, PL2, true, NP, false>));                                 \
            else LAUNCH_K((k_stft_phat_wave<MT, U, OutT, PL2, false, NP, false>));                                   \
        } while (0)
#define LAUNCH_W(MT, U)                                                                                              \
        do {                                                                                                         \
            if constexpr (sizeof(OutT) == 2) {                                                                       \
                if (pl2 && w.cand_on && a.list && !pw) LAUNCH_K((k_stft_phat_wave<MT, U, OutT, true, false, false, false, true>)); \
                else if (pl2) LAUNCH_W2(MT, U, true, false); else LAUNCH_W2(MT, U, false, false);                    \
            }                                                                                                        \
            else { if (a.no_phat) LAUNCH_W2(MT, U, false, true); else LAUNCH_W2(MT, U, false, false); }              \
        } while (0)
        if constexpr (sizeof(OutT) == 2) {
            if (mg) {          // merged contraction index: ULA, one plane
                if (M == 8) { if (pw) LAUNCH_K((k_stft_phat_wave<8, true, OutT, false, true, false, true>)); else LAUNCH_K((k_stft_phat_wave<8, true, OutT, false, false, false, true>)); }
                else { if (pw) LAUNCH_K((k_stft_phat_wave<4, true, OutT, false, true, false, true>)); else LAUNCH_K((k_stft_phat_wave<4, true, OutT, false, false, false, true>)); }
                HIP_TRY(c, hipGetLastError());
                return MCA_HIP_OK;
            }
        }
        if (M == 8 && ula) LAUNCH_W(8, true); else if (M == 8) LAUNCH_W(8, false);
        else if (ula) LAUNCH_W(4, true); else LAUNCH_W(4, false);
#undef LAUNCH_W2
#undef LAUNCH_K
#undef LAUNCH_W
        HIP_TRY(c, hipGetLastError());
        return MCA_HIP_OK;
    }
#define LAUNCH(MT, U)                                                                                        \
    do {                                                                                                     \
        if (smem > 64 * 1024)                                                                                \
            HIP_TRY(c, hipFuncSetAttribute(reinterpret_cast<const void *>(&k_stft_phat<MT, U, OutT>),        \
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));          \
        hipLaunchKernelGGL((k_stft_phat<MT, U, OutT>), grid, dim3(512), smem, st, a);                        \
    } while (0)
    // 2 microphones: four frames per pass so that all eight waves transform (k_stft_phat_few)
#define LAUNCH_FEW(MT, U)                                                                                    \
    do {                                                                                                     \
        const size_t smf = ((size_t)8 * FFT_SCRATCH + TW_WORDS + (size_t)a.fpb * MT) * sizeof(float2) + (size_t)a.fpb * 8 * sizeof(float); \
        if (smf > 64 * 1024)                                                                                 \
            HIP_TRY(c, hipFuncSetAttribute(reinterpret_cast<const void *>(&k_stft_phat_few<MT, U, OutT>),    \
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)smf));           \
        hipLaunchKernelGGL((k_stft_phat_few<MT, U, OutT>), grid, dim3(512), smf, st, a);                     \
    } while (0)
    // (measured: 0.51 -> 0.33 ms for 131 072 two-channel frames; no gain with 4 channels, which stay on k_stft_phat)
    if (M == 2 && !ula) LAUNCH_FEW(2, false);
    else if (M == 4 && ula) LAUNCH(4, true);
    else if (M == 4) LAUNCH(4, false);
    else if (M == 8 && ula) LAUNCH(8, true);
    else if (M == 8) LAUNCH(8, false);
    else if (M == 16 && ula) LAUNCH(16, true);
    else if (ula) LAUNCH(0, true);
    else LAUNCH(0, false);
#undef LAUNCH
#undef LAUNCH_FEW
    HIP_TRY(c, hipGetLastError());
    return MCA_HIP_OK;
}

int check_stream_args(mca_hip_ctx *c, const float *pcm, long long array_stride, long long mic_stride, int n_arrays, int n_frames)
{
    if (!c) return MCA_HIP_ERR_INVALID_ARGUMENT;
    if (!c->stream_ok) return fail(c, MCA_HIP_ERR_UNSUPPORTED, c->stream_why);
    if (!pcm) return fail(c, MCA_HIP_ERR_INVALID_ARGUMENT, "pcm_dev is NULL");
    if (n_arrays < 1 || n_arrays > c->cfg.max_arrays) return fail(c, MCA_HIP_ERR_INVALID_ARGUMENT, "n_arrays outside [1, max_arrays]");
    if (n_frames < 1) return fail(c, MCA_HIP_ERR_INVALID_ARGUMENT, "n_frames < 1");
    const long long need = (long long)(n_frames + 1) * c->H;
    if (mic_stride < need) return fail(c, MCA_HIP_ERR_INVALID_ARGUMENT, "mic_stride shorter than (n_frames+1)*hop samples");
    if (n_arrays > 1 && array_stride < (long long)(c->M - 1) * mic_stride + need) return fail(c, MCA_HIP_ERR_INVALID_ARGUMENT, "array_stride too short");
    if ((mic_stride & 1) || (array_stride & 1) || (reinterpret_cast<uintptr_t>(pcm) & 7))
        return fail(c, MCA_HIP_ERR_INVALID_ARGUMENT, "pcm_dev must be 8-byte aligned with even strides (float2 loads)");
    return MCA_HIP_OK;
}

}  // namespace

extern "C" {

static int settle_history(mca_hip_ctx *c, hipStream_t st);    // lazy tails: exact state from the kept history (defined with localise_impl)

const char *mca_hip_version(void) { return "mcarray-hip 0.1.0 (gfx950)"; }

const char *mca_hip_last_error(const mca_hip_ctx *ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

int mca_hip_create(const mca_hip_config *cfg, mca_hip_ctx **out)
{
    if (!cfg || !out) return fail(nullptr, MCA_HIP_ERR_INVALID_ARGUMENT, "cfg/out is NULL");
    *out = nullptr;
    // Callers built against an earlier header pass its size: round 2 ended before gcc_weighting (PHAT), round 3 behind it -- that
    // struct was 60 bytes of fields padded to 64, and the padding is not read -- and the fields appended since default to zero.
    constexpr int size_r2 = (int)offsetof(mca_hip_config, gcc_weighting), size_r3 = (size_r2 + 4 + 7) / 8 * 8;
    static_assert(size_r3 <= (int)offsetof(mca_hip_config, adaptive_min_rows), "round-3 struct size");
    if (cfg->struct_size != (int)sizeof(mca_hip_config) && cfg->struct_size != size_r2 && cfg->struct_size != size_r3)
        return fail(nullptr, MCA_HIP_ERR_INVALID_ARGUMENT, "struct_size mismatch");
    mca_hip_config cfg_full{};
    std::memcpy(&cfg_full, cfg, cfg->struct_size == size_r3 ? (size_t)offsetof(mca_hip_config, adaptive_fallback) : (size_t)cfg->struct_size);
    cfg_full.struct_size = (int)sizeof(mca_hip_config);
    cfg = &cfg_full;
    if (cfg->gcc_weighting != MCA_HIP_GCC_PHAT && cfg->gcc_weighting != MCA_HIP_GCC_NONE)
        return fail(nullptr, MCA_HIP_ERR_INVALID_ARGUMENT, "gcc_weighting must be MCA_HIP_GCC_PHAT or MCA_HIP_GCC_NONE");
    if (cfg->gcc_weighting == MCA_HIP_GCC_NONE && cfg->srp_precision != MCA_HIP_SRP_FP32)
        return fail(nullptr, MCA_HIP_ERR_UNSUPPORTED, "gcc_weighting NONE needs srp_precision MCA_HIP_SRP_FP32 (un-normalised spectra do not fit fp16 operands)");
    if (cfg->n_mics < 2 || cfg->n_mics > MCA_MAX_MICS) return fail(nullptr, MCA_HIP_ERR_INVALID_ARGUMENT, "n_mics must be in [2,16]");
    if (!cfg->mic_xyz) return fail(nullptr, MCA_HIP_ERR_INVALID_ARGUMENT, "mic_xyz is NULL");
    if (cfg->n_sources < 1 || cfg->n_sources > MCA_MAX_SOURCES) return fail(nullptr, MCA_HIP_ERR_INVALID_ARGUMENT, "n_sources must be in [1,4]");
    if (cfg->fft_size < 16 || (cfg->fft_size & 1) || cfg->fft_size > 8192) return fail(nullptr, MCA_HIP_ERR_INVALID_ARGUMENT, "fft_size must be even, 16..8192");
    if (cfg->sample_rate <= 0) return fail(nullptr, MCA_HIP_ERR_INVALID_ARGUMENT, "sample_rate <= 0");
    if (!(cfg->doa_step_deg > 0) || cfg->doa_step_deg > 45) return fail(nullptr, MCA_HIP_ERR_INVALID_ARGUMENT, "doa_step_deg must be in (0,45]");
    if (cfg->srp_precision < 0 || cfg->srp_precision > 3) return fail(nullptr, MCA_HIP_ERR_INVALID_ARGUMENT, "bad srp_precision");
    if (cfg->max_arrays < 1) return fail(nullptr, MCA_HIP_ERR_INVALID_ARGUMENT, "max_arrays < 1");
    if (cfg->adaptive_fallback != MCA_HIP_ADAPT_FALLBACK_AUTO && cfg->adaptive_fallback != MCA_HIP_ADAPT_FALLBACK_OFF)
        return fail(nullptr, MCA_HIP_ERR_INVALID_ARGUMENT, "adaptive_fallback must be MCA_HIP_ADAPT_FALLBACK_AUTO or _OFF");
    if (cfg->adaptive_min_rows < 0 || cfg->adaptive_max_sources < 0) return fail(nullptr, MCA_HIP_ERR_INVALID_ARGUMENT, "adaptive_min_rows / adaptive_max_sources < 0");

    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(nullptr, MCA_HIP_ERR_NO_DEVICE, "no HIP device visible; libmcarray_hip has no CPU fallback");
    if (cfg->device < 0 || cfg->device >= ndev) return fail(nullptr, MCA_HIP_ERR_INVALID_ARGUMENT, "device ordinal out of range");
    if (hipSetDevice(cfg->device) != hipSuccess) return fail(nullptr, MCA_HIP_ERR_HIP, "hipSetDevice failed");
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, cfg->device) != hipSuccess) return fail(nullptr, MCA_HIP_ERR_HIP, "hipGetDeviceProperties failed");
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(nullptr, MCA_HIP_ERR_NO_DEVICE, std::string("device is ") + prop.gcnArchName + ", this library is built for gfx950 only");

    mca_hip_ctx *c = new mca_hip_ctx();
    c->cfg = *cfg;
    c->kn = read_knobs(*cfg);
    c->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    c->M = cfg->n_mics; c->S = cfg->n_sources; c->N = cfg->fft_size; c->K = c->N / 2 + 1; c->prec = cfg->srp_precision;
    c->xyz.assign(cfg->mic_xyz, cfg->mic_xyz + 3 * c->M);
    c->cfg.mic_xyz = nullptr;
    c->step = (float)(cfg->doa_step_deg * M_PI / 180.0);                    // SteeringBeamforming.cpp:39
    c->D = (int)(std::round(M_PI / (double)c->step) + 1);                   // :40
    if (c->D < 3 || c->D > 512) { free_ctx(c); return fail(nullptr, MCA_HIP_ERR_UNSUPPORTED, "number of steering angles must be in [3,512]"); }
    c->P = c->M * (c->M - 1) / 2;
    // column padding of the correlation map = tile width of the contraction kernels: 192, or 64 for the fp16 kernels on
    // grids of up to 64 angles (the reference's own 37, the 2-microphone module's 61)
    c->Dp = (c->D <= 64 && c->prec != MCA_HIP_SRP_FP32) ? 64 : round_up(c->D, 192);
    c->H = c->N / 2;
    // stream API: the tuned 1024-sample kernels, or the any-power-of-two kernels (kernels_generic.hip)
    // whose channel spectra of one frame must fit the 160 KiB LDS of a CU
    c->generic = c->N != FFT_N || c->kn.force_generic;
    c->stream_ok = true;
    if (c->generic) {
        while ((1 << c->logH) < c->H) ++c->logH;
        if ((1 << c->logH) != c->H || c->N < 64) { c->stream_ok = false; c->stream_why = "the stream API needs a power-of-two fft_size >= 64 (the frame API takes any even size)"; }
        else if ((size_t)(c->M + 1) * (c->H + 1) * 8 + (size_t)c->H * 4 + 17 * 8 + 16 > 160 * 1024) {      // (the beamformer takes the sources in passes if need be)
            c->stream_ok = false; c->stream_why = "fft_size x n_mics exceeds the 160 KiB LDS of a CU in the stream API (use the frame API)";
        }
    }
    if (c->stream_ok && c->cfg.gcc_weighting == MCA_HIP_GCC_NONE && (c->generic || (c->M != 4 && c->M != 8))) {
        c->stream_ok = false;
        c->stream_why = "gcc_weighting NONE: the stream API serves it at fft_size 1024 with 4 or 8 microphones (the frame API takes any shape)";
    }
    c->n512 = c->N == 512 && c->M <= 8 && c->stream_ok && !c->kn.no_n512;
    c->n2048 = c->N == 2048 && c->stream_ok && !c->kn.no_n2048;
    if (c->kn.v2_min_rows > 0) c->v2_min_rows = c->kn.v2_min_rows;

    // generateLookupTable (SteeringBeamforming.cpp:58-94): pairs i<j lexicographic, float delays
    c->delays.resize((size_t)c->P * c->D);
    std::vector<float> fdist(c->P);
    {
        int p = 0;
        for (int i = 0; i < c->M; ++i)
            for (int j = i + 1; j < c->M; ++j, ++p) {
                double dist = distance(c->xyz, i, j);                         // :67
                fdist[p] = (float)dist;                                       // `float microDist` parameter
                for (int d = 0; d < c->D; ++d)                                // :71-73
                    c->delays[(size_t)p * c->D + d] = doa_to_delay_samples(doa_idx2angle(d, c->step), (float)dist, cfg->sample_rate);
                c->pairs.push_back(make_int2(i, j));
            }
    }
    c->grid.resize(c->D);
    for (int d = 0; d < c->D; ++d) c->grid[d] = doa_idx2angle(d, c->step);
    // delay-group merging: legal iff all pairs with equal (j - i) have bit-identical float distances
    // (then their D delays are bit-identical too) -- true for uniform linear arrays.
    c->ula = c->M > 2;
    {
        int p = 0;
        for (int i = 0; i < c->M && c->ula; ++i)
            for (int j = i + 1; j < c->M; ++j, ++p)
                if (std::memcmp(&fdist[p], &fdist[j - i - 1], sizeof(float)) != 0) { c->ula = false; break; }
    }
    c->G = c->ula ? c->M - 1 : c->P;
    c->Kp = round_up(c->G * c->K * 2, 32);     // K == KG == 513 on the tuned path
    c->tab_planes = (c->prec == MCA_HIP_SRP_FP16X3 || c->prec == MCA_HIP_SRP_ADAPTIVE) ? 2 : 1;
    c->a_planes = c->prec == MCA_HIP_SRP_FP16X3 ? 2 : 1;          // (ADAPTIVE: set per call, 1 for the coarse pass, 2 otherwise)
    c->a_elem = c->prec == MCA_HIP_SRP_FP32 ? 4 : 2;
    c->a_row_elems = c->Kp * c->a_planes;
    {
        // Error model of the coarse (one fp16 product) map, for the sensitivity test of k_scan_pick: every operand carries
        // a relative rounding error of rms ~2-3e-4 (11-bit significand), so a term a*b of the contraction is off by
        // ~4e-4 rms and C[d] by sigma_C = 4e-4 sqrt(sum_k a_k^2 b_k^2) <= 4e-4 sqrt(K/2 sum_g n_g^2) (n_g = pairs per delay
        // group, |PHAT sum of a group| <= n_g; b = cos / sin).  For a static source the error repeats from frame to frame,
        // so the 0.8 recursion does not average it: sigma_E = sigma_C.  A difference of two energies is decided at
        // 8 sigma plus head room: tau = 8 sqrt(2) 1.25 sigma_C, in units of the normalised energy En = (E + 15 P) / (30 P).
        // Calibration (tools/adaptive_check.py, 60 random configurations, 1e9 map values): the largest |E_fp16 - E_fp16x3|
        // seen is 0.42 tau (16-microphone ULA, static source), typically 0.2 tau.
        double sum_n2 = 0;
        if (c->ula) for (int g = 0; g < c->G; ++g) sum_n2 += (double)(c->M - 1 - g) * (c->M - 1 - g);
        else sum_n2 = c->P;
        const double sigma_c = 5.0e-4 * std::sqrt(0.5 * c->K * sum_n2);
        c->tau_en = (float)(c->kn.tau_scale * 8.0 * std::sqrt(2.0) * sigma_c / (30.0 * c->P));
    }

    int rc = MCA_HIP_OK;
    auto up = [&](void **dst, const void *src, size_t bytes) -> int {
        HIP_TRY(c, hipMalloc(dst, bytes));
        HIP_TRY(c, hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice));
        return MCA_HIP_OK;
    };
    auto zalloc = [&](void **dst, size_t bytes) -> int {
        HIP_TRY(c, hipMalloc(dst, bytes));
        HIP_TRY(c, hipMemset(*dst, 0, bytes));
        return MCA_HIP_OK;
    };
    std::vector<double> micx(c->M);
    for (int m = 0; m < c->M; ++m) micx[m] = c->xyz[3 * m];
    std::vector<float> win(c->N);
    for (int n = 0; n < c->N; ++n) win[n] = (float)(0.5 - 0.5 * std::cos(2.0 * M_PI * n / c->N));   // periodic Hann
    std::vector<float2> twv(c->N / 2);
    for (int i = 0; i < c->N / 2; ++i) twv[i] = make_float2((float)std::cos(2.0 * M_PI * i / c->N), (float)(-std::sin(2.0 * M_PI * i / c->N)));
    const size_t na = (size_t)cfg->max_arrays;
    if ((rc = up((void **)&c->d_window, win.data(), win.size() * 4)) || (rc = up((void **)&c->d_tw, twv.data(), twv.size() * 8)) || (rc = up((void **)&c->d_grid, c->grid.data(), c->grid.size() * 4)) ||
        (rc = up((void **)&c->d_delays, c->delays.data(), c->delays.size() * 4)) || (rc = up((void **)&c->d_micx, micx.data(), micx.size() * 8)) ||
        (rc = up((void **)&c->d_pairs, c->pairs.data(), c->pairs.size() * sizeof(int2))) ||
        (rc = zalloc((void **)&c->d_E[0], na * c->D * 4)) || (rc = zalloc((void **)&c->d_E[1], na * c->D * 4)) ||
        (rc = zalloc((void **)&c->d_tail[0], na * c->S * c->H * 4)) || (rc = zalloc((void **)&c->d_tail[1], na * c->S * c->H * 4)) ||
        (rc = zalloc((void **)&c->d_gate_state, na * 4 * 8)) || (rc = zalloc((void **)&c->d_last_bin, na * MCA_MAX_SOURCES * 4)) ||
        (rc = zalloc((void **)&c->d_last_rad, na * MCA_MAX_SOURCES * 4)) || (rc = zalloc((void **)&c->d_last_prob, na * MCA_MAX_SOURCES * 4)) ||
        (rc = zalloc((void **)&c->d_doa[0], na * 4)) || (rc = zalloc((void **)&c->d_doa[1], na * 4)) ||
        (rc = zalloc((void **)&c->d_vdone[0], na * 8)) || (rc = zalloc((void **)&c->d_vdone[1], na * 8)) ||
        (rc = zalloc((void **)&c->d_silence, na * 4)) || (rc = zalloc((void **)&c->d_g2_post0, na * 4)) ||
        (rc = zalloc((void **)&c->d_rstats, 32)) || (rc = zalloc((void **)&c->d_queue, 64)) ||
        (rc = zalloc((void **)&c->d_E64[0], c->D * 8)) || (rc = zalloc((void **)&c->d_E64[1], c->D * 8)) ||
        (rc = zalloc((void **)&c->d_res, (2 * MCA_MAX_SOURCES + 1) * 8)) || (rc = zalloc((void **)&c->d_bins, MCA_MAX_SOURCES * 4))) {
        g_create_error = c->err; free_ctx(c); return rc;
    }
    if (lazy_context(c)) {
        for (int i = 0; i < 2; ++i)
            if ((rc = zalloc((void **)&c->d_hist_pcm[i], na * c->M * HIST_SAMPLES * 4)) || (rc = zalloc((void **)&c->d_hist_C[i], na * HIST_FRAMES * c->Dp * 4)) ||
                (rc = zalloc((void **)&c->d_ehist[i], na * c->D * 4))) { g_create_error = c->err; free_ctx(c); return rc; }
    }
    if (c->prec == MCA_HIP_SRP_ADAPTIVE && c->kn.fb_enabled) {
        if (hipHostMalloc((void **)&c->h_probe, FB_RING * 32, hipHostMallocDefault) == hipSuccess) std::memset(c->h_probe, 0, FB_RING * 32);
        else { c->h_probe = nullptr; (void)hipGetLastError(); }                    // (no page-locked memory: the mode never backs off)
    }
    if (c->stream_ok && ((rc = build_steering_table(c)) || (rc = build_merged_tables(c)))) { g_create_error = c->err; free_ctx(c); return rc; }
    if ((rc = init_last_state(c, nullptr))) { g_create_error = c->err; free_ctx(c); return rc; }
    *out = c;
    return MCA_HIP_OK;
}

static void graph_drop_recordings(mca_hip_graph *g);
static void graph_orphan(mca_hip_graph *g);

void mca_hip_destroy(mca_hip_ctx *ctx)
{
    if (!ctx) return;
    (void)hipSetDevice(ctx->cfg.device);
    (void)hipDeviceSynchronize();
    // graphs that outlive their context: their recordings point into this context's buffers, so they go now; the handle
    // stays valid for mca_hip_graph_destroy and fails cleanly in mca_hip_graph_launch
    for (mca_hip_graph *g : ctx->graphs) graph_orphan(g);
    ctx->graphs.clear();
    free_ctx(ctx);
}

int mca_hip_num_steps(const mca_hip_ctx *c) { return c ? c->D : MCA_HIP_ERR_INVALID_ARGUMENT; }
int mca_hip_num_pairs(const mca_hip_ctx *c) { return c ? c->P : MCA_HIP_ERR_INVALID_ARGUMENT; }
int mca_hip_num_groups(const mca_hip_ctx *c) { return c ? c->G : MCA_HIP_ERR_INVALID_ARGUMENT; }

int mca_hip_get_pair_delays(const mca_hip_ctx *c, float *out)
{
    if (!c || !out) return MCA_HIP_ERR_INVALID_ARGUMENT;
    std::memcpy(out, c->delays.data(), c->delays.size() * sizeof(float));
    return MCA_HIP_OK;
}

int mca_hip_get_doa_grid(const mca_hip_ctx *c, float *out)
{
    if (!c || !out) return MCA_HIP_ERR_INVALID_ARGUMENT;
    std::memcpy(out, c->grid.data(), c->grid.size() * sizeof(float));
    return MCA_HIP_OK;
}

int mca_hip_reset(mca_hip_ctx *c, void *stream)
{
    if (!c) return MCA_HIP_ERR_INVALID_ARGUMENT;
    hipStream_t st = (hipStream_t)stream;
    const size_t na = (size_t)c->cfg.max_arrays;
    for (int i = 0; i < 2; ++i) {
        HIP_TRY(c, hipMemsetAsync(c->d_E[i], 0, na * c->D * 4, st));
        HIP_TRY(c, hipMemsetAsync(c->d_tail[i], 0, na * c->S * c->H * 4, st));
        HIP_TRY(c, hipMemsetAsync(c->d_E64[i], 0, (size_t)c->D * 8, st));
        HIP_TRY(c, hipMemsetAsync(c->d_doa[i], 0, na * 4, st));
        HIP_TRY(c, hipMemsetAsync(c->d_vdone[i], 0, na * 8, st));
    }
    HIP_TRY(c, hipMemsetAsync(c->d_silence, 0, na * 4, st));
    c->adapt_suspended = false; c->fb_state = 0; c->fb_left = 0; c->fb_backoff = 8; c->cand_heavy = false;     // new streams: the adaptive mode starts afresh
    c->fb_lost = false; c->fb_seq_seen = c->fb_calls;                                                          // (reports still in flight belong to the old streams)
    c->hist_pending = false;                                                              // ... and no call owes the next one its last rows
    // the self-cleaning words of the adaptive path (a call that failed half way may have left some set)
    for (Workspace &w : c->lanes) {
        if (w.d_need) HIP_TRY(c, hipMemsetAsync(w.d_need, 0, w.adapt_groups * 4, st));
        if (w.d_need_full) HIP_TRY(c, hipMemsetAsync(w.d_need_full, 0, w.adapt_groups * 4, st));
        if (w.d_umask) HIP_TRY(c, hipMemsetAsync(w.d_umask, 0, w.adapt_groups * (c->Dp / 32) * 4, st));
        if (w.d_chunk_from) HIP_TRY(c, hipMemsetAsync(w.d_chunk_from, 0x7f, w.adapt_chunks * 4, st));
        if (w.d_nlist) HIP_TRY(c, hipMemsetAsync(w.d_nlist, 0, 16, st));
    }
    c->gcc2_frames_done = 0;
    return init_last_state(c, st);
}

extern "C++" {
namespace {

struct StateHeader {
    unsigned magic; int version, M, D, S, H, max_arrays, use_floor;
    unsigned delays_hash;            // FNV-1a of the float delay tables: geometry + sample rate + grid
    long long gcc2_frames_done;
};
constexpr unsigned STATE_MAGIC = 0x4d434153u;   // "MCAS"
constexpr int STATE_VERSION = 2;                // 2: + _silenceFramesCounter per array

unsigned delays_hash(const mca_hip_ctx *c)
{
    unsigned h = 2166136261u;
    const unsigned char *p = reinterpret_cast<const unsigned char *>(c->delays.data());
    for (size_t i = 0; i < c->delays.size() * sizeof(float); ++i) { h ^= p[i]; h *= 16777619u; }
    return h;
}

// the device buffers that make up the state, in blob order (current halves of the double buffers)
struct StatePart { void *ptr; size_t bytes; };
std::vector<StatePart> state_parts(mca_hip_ctx *c)
{
    const size_t na = (size_t)c->cfg.max_arrays;
    return {
        {c->d_E[c->e_cur], na * c->D * 4}, {c->d_tail[c->tail_cur], na * c->S * c->H * 4}, {c->d_gate_state, na * 4 * 8},
        {c->d_last_bin, na * MCA_MAX_SOURCES * 4}, {c->d_last_rad, na * MCA_MAX_SOURCES * 4}, {c->d_last_prob, na * MCA_MAX_SOURCES * 4},
        {c->d_doa[c->doa_cur], na * 4}, {c->d_E64[c->e64_cur], (size_t)c->D * 8}, {c->d_vdone[c->doa_cur], na * 8},
        {c->d_silence, na * 4},
    };
}

}  // namespace
}  // extern "C++"

long long mca_hip_state_size(const mca_hip_ctx *c)
{
    if (!c) return MCA_HIP_ERR_INVALID_ARGUMENT;
    long long n = sizeof(StateHeader);
    for (const StatePart &p : state_parts(const_cast<mca_hip_ctx *>(c))) n += (long long)p.bytes;
    return n;
}

int mca_hip_state_save(mca_hip_ctx *c, void *blob, long long blob_bytes)
{
    if (!c) return MCA_HIP_ERR_INVALID_ARGUMENT;
    if (!blob || blob_bytes < mca_hip_state_size(c)) return fail(c, MCA_HIP_ERR_INVALID_ARGUMENT, "state blob is NULL or smaller than mca_hip_state_size()");
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    if (c->hist_pending) {                        // lazy tails: the blob holds the exact state, as the eager form's would
        HIP_TRY(c, hipDeviceSynchronize());
        const int rc = settle_history(c, nullptr);
        if (rc) return rc;
    }
    HIP_TRY(c, hipDeviceSynchronize());
    StateHeader h{STATE_MAGIC, STATE_VERSION, c->M, c->D, c->S, c->H, c->cfg.max_arrays, c->cfg.use_power_floor, delays_hash(c), c->gcc2_frames_done};
    unsigned char *out = static_cast<unsigned char *>(blob);
    std::memcpy(out, &h, sizeof(h)); out += sizeof(h);
    for (const StatePart &p : state_parts(c)) { HIP_TRY(c, hipMemcpy(out, p.ptr, p.bytes, hipMemcpyDeviceToHost)); out += p.bytes; }
    return MCA_HIP_OK;
}

int mca_hip_state_load(mca_hip_ctx *c, const void *blob, long long blob_bytes)
{
    if (!c) return MCA_HIP_ERR_INVALID_ARGUMENT;
    if (!blob || blob_bytes < (long long)sizeof(StateHeader)) return fail(c, MCA_HIP_ERR_INVALID_ARGUMENT, "state blob is NULL or truncated");
    StateHeader h;
    std::memcpy(&h, blob, sizeof(h));
    // version 1 blobs end before the _silenceFramesCounter part (the only change of version 2): they load with the counters at zero
    if (h.magic != STATE_MAGIC || (h.version != STATE_VERSION && h.version != 1)) return fail(c, MCA_HIP_ERR_INVALID_ARGUMENT, "not a state blob of this library version");
    if (h.M != c->M || h.D != c->D || h.S != c->S || h.H != c->H || h.max_arrays != c->cfg.max_arrays || h.use_floor != c->cfg.use_power_floor ||
        h.delays_hash != delays_hash(c))
        return fail(c, MCA_HIP_ERR_INVALID_ARGUMENT, "state blob was saved by a context with a different configuration");
    const long long v1_short = h.version == 1 ? (long long)c->cfg.max_arrays * 4 : 0;       // bytes of the part a version-1 blob lacks
    if (blob_bytes < mca_hip_state_size(c) - v1_short) return fail(c, MCA_HIP_ERR_INVALID_ARGUMENT, "state blob is truncated");
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    HIP_TRY(c, hipDeviceSynchronize());
    const unsigned char *in = static_cast<const unsigned char *>(blob) + sizeof(h);
    for (const StatePart &p : state_parts(c)) {
        if (h.version == 1 && p.ptr == c->d_silence) { HIP_TRY(c, hipMemset(p.ptr, 0, p.bytes)); continue; }
        HIP_TRY(c, hipMemcpy(p.ptr, in, p.bytes, hipMemcpyHostToDevice)); in += p.bytes;
    }
    c->gcc2_frames_done = h.gcc2_frames_done;
    c->hist_pending = false;                      // (a blob holds the exact state)
    return MCA_HIP_OK;
}

// workspace of one lane for a call of n_arrays x n_frames
static int reserve_lane(mca_hip_ctx *c, int n_arrays, int n_frames)
{
    const int n_chunks = (n_frames + SCAN_CHUNK - 1) / SCAN_CHUNK;
    int rc;
    if (c->prec == MCA_HIP_SRP_ADAPTIVE && adaptive_shape(c, n_arrays, n_frames)) {
        // by the SHAPE of the call, whatever the back-off is doing at the moment: a graph recording always runs coarse + repair
        // (graph_record clears adapt_suspended), and reserving for plain FP16X3 only would make it allocate inside the capture
        // (ADVICE r3).  While the mode is suspended the eager calls run as FP16X3: their two-plane rows are reserved as well.
        set_call_planes(c, 1);
        if ((rc = ensure_adapt_workspace(c, n_arrays, n_frames, n_chunks))) return rc;
        const long long fc1 = chunk_frames_for(c, n_arrays, n_frames);
        if ((rc = ensure_workspace(c, (long long)n_arrays * fc1, (long long)n_arrays * n_frames))) return rc;
        if (!c->adapt_suspended) return ensure_scan_workspace(c, n_arrays, n_frames, n_chunks);
    }
    if (c->prec == MCA_HIP_SRP_ADAPTIVE) set_call_planes(c, 2);
    long long fc = chunk_frames_for(c, n_arrays, n_frames);
    rc = ensure_workspace(c, (long long)n_arrays * fc, (long long)n_arrays * n_frames);
    if (rc) return rc;
    return ensure_scan_workspace(c, n_arrays, n_frames, n_chunks);
}

static int reserve_impl(mca_hip_ctx *c, int n_arrays, int n_frames)
{
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    c->cur_lane = 0;
    return reserve_lane(c, n_arrays, n_frames);
}

int mca_hip_reserve(mca_hip_ctx *c, int n_arrays, int n_frames)
{
    if (!c || n_arrays < 1 || n_frames < 1) return MCA_HIP_ERR_INVALID_ARGUMENT;
    return reserve_impl(c, n_arrays, n_frames);
}

// STFT + PHAT + steering contraction for every frame: fills c->ws().d_C [arrays][n_frames][Dp]
static int run_correlation_map(mca_hip_ctx *c, const float *pcm, long long array_stride, long long mic_stride,
                               int n_arrays, int n_frames, hipStream_t st)
{
    int rc;
    const long long fc = chunk_frames_for(c, n_arrays, n_frames);
    if ((rc = ensure_workspace(c, (long long)n_arrays * fc, (long long)n_arrays * n_frames))) return rc;
    if ((rc = ensure_scan_workspace(c, n_arrays, n_frames, (n_frames + SCAN_CHUNK - 1) / SCAN_CHUNK))) return rc;
    // The 256 x 384 contraction leaves the chunk-local results of the scan over frames (k_scan_partial's job) when every
    // 32-row block of every launch is one scan chunk of one map: no gate, one map, whole chunks, that kernel for every slice.
    bool fused_partial = SCAN_CHUNK == 32 && c->a_planes == 1 && !c->cfg.use_power_floor && n_frames % SCAN_CHUNK == 0 && fc % SCAN_CHUNK == 0 &&
                         plan_gemm(c, (long long)n_arrays * fc).ksplit <= 2 && !c->kn.no_fused_partial;
    for (int f0 = 0; fused_partial && f0 < n_frames; f0 += (int)fc)
        fused_partial = plan_gemm(c, (long long)n_arrays * std::min<long long>(fc, n_frames - f0)).v2;
    c->ws().partial_done = fused_partial;
    c->ws().part_planes = fused_partial ? plan_gemm(c, (long long)n_arrays * fc).ksplit : 1;
    for (int f0 = 0; f0 < n_frames; f0 += (int)fc) {
        const int nf = (int)std::min<long long>(fc, n_frames - f0);
        StftPhatArgs sa{};
        sa.pcm = pcm; sa.array_stride = array_stride; sa.mic_stride = mic_stride;
        sa.M = c->M; sa.n_frames = nf; sa.frame0 = f0;
        // frames per workgroup: 16 amortise the table set-up and the half frame two neighbouring workgroups both read (17 half
        // frames of input per 16 frames; measured 4 / 8 / 16 / 32: 0.312 / 0.292 / 0.286 / 0.288 ms); small batches take fewer
        // so that a few hundred workgroups exist
        sa.fpb = 16;
        while (sa.fpb > 1 && (long long)n_arrays * ((nf + sa.fpb - 1) / sa.fpb) < 256) sa.fpb >>= 1;
        sa.power = c->cfg.use_power_floor ? c->ws().d_power : nullptr; sa.total_frames = n_frames;
        sa.window = c->d_window; sa.A = a_buf(c); sa.Kp = c->Kp; sa.a_row_elems = c->a_row_elems; sa.a_planes = c->a_planes;
        if (c->merged && c->a_planes == 1) { sa.mrank = c->d_mrank; sa.n_merged = c->n_merged; }
        sa.N = c->N; sa.logH = c->logH; sa.kg = c->K; sa.ula = c->ula ? 1 : 0; sa.tw = c->d_tw;
        sa.no_phat = c->cfg.gcc_weighting == MCA_HIP_GCC_NONE ? 1 : 0;
        if (c->prec == MCA_HIP_SRP_ADAPTIVE && c->a_planes == 1 && wave16_applies(c)) sa.unsure = c->ws().d_unsure;   // (adaptive coarse pass)
        if (c->lazy_now && cand_call(c, true)) sa.dead = c->ws().d_unsure;                                               // (... of a candidate-column call: frames of exact zeros)
        if (c->lazy_now) sa.hist_out = c->d_hist_pcm[c->hist_cur ^ 1];       // lazy tails: the call's last frames of PCM stay behind
        time_begin(c, MCA_HIP_K_STFT_PHAT, st);
        if (c->n512) {
            rc = c->prec == MCA_HIP_SRP_FP32 ? launch_stft_512<float>(c, sa, dim3(0, n_arrays), st) : launch_stft_512<_Float16>(c, sa, dim3(0, n_arrays), st);
        } else if ((c->N == 4096 || c->N == 2048) && c->M == 2 && c->stream_ok && !c->kn.no_sub2) {
            // two microphones at 2048- / 4096-sample frames (FreqGCC at 32 / 44.1 / 48 kHz): 512-sample sub-sequences per channel
            sa.fpb = 8;
            while (sa.fpb > 2 && (long long)n_arrays * ((nf + sa.fpb - 1) / sa.fpb) < 512) sa.fpb >>= 1;
            const size_t smem4 = ((size_t)16 * 258 + 8 * FFT_SCRATCH + TW_WIN) * sizeof(float2) + 4 * 8 * sizeof(float);
            dim3 g4((nf + sa.fpb - 1) / sa.fpb, n_arrays);
#define LSUB(RR, T)                                                                                                       \
            do {                                                                                                          \
                HIP_TRY(c, hipFuncSetAttribute(reinterpret_cast<const void *>(&k_stft_phat_sub2<RR, T>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem4)); \
                hipLaunchKernelGGL((k_stft_phat_sub2<RR, T>), g4, dim3(512), smem4, st, sa);                              \
            } while (0)
            if (c->N == 4096) { if (c->prec == MCA_HIP_SRP_FP32) LSUB(8, float); else LSUB(8, _Float16); }
            else { if (c->prec == MCA_HIP_SRP_FP32) LSUB(4, float); else LSUB(4, _Float16); }
#undef LSUB
            rc = MCA_HIP_OK;
        } else if (n2048_analysis(c)) {
            rc = c->prec == MCA_HIP_SRP_FP32 ? launch_stft_2048<float>(c, sa, dim3(0, n_arrays), st) : launch_stft_2048<_Float16>(c, sa, dim3(0, n_arrays), st);
        } else if (c->generic) {
            const size_t smem1 = (size_t)c->M * (c->H + 1) * sizeof(float2) + 16;
#define GEN_LAUNCH(K)                                                                                                     \
            do {                                                                                                          \
                if (smem1 > 64 * 1024)                                                                                    \
                    HIP_TRY(c, hipFuncSetAttribute(reinterpret_cast<const void *>(&K), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem1)); \
                hipLaunchKernelGGL(K, dim3(nf, n_arrays), dim3(gen_threads(c, false)), smem1, st, sa);                                      \
            } while (0)
            if (c->prec == MCA_HIP_SRP_FP32) GEN_LAUNCH(k_stft_phat_gen<float>);
            else GEN_LAUNCH(k_stft_phat_gen<_Float16>);
#undef GEN_LAUNCH
            rc = MCA_HIP_OK;
        } else {
            dim3 g1((nf + sa.fpb - 1) / sa.fpb, n_arrays);
            const size_t smem1 = ((size_t)c->M * FFT_SCRATCH + TW_WORDS + (size_t)sa.fpb * c->M) * sizeof(float2) + (size_t)sa.fpb * 8 * sizeof(float);
            rc = c->prec == MCA_HIP_SRP_FP32 ? launch_stft<float>(c, sa, g1, smem1, st) : launch_stft<_Float16>(c, sa, g1, smem1, st);
        }
        time_end(c, st);
        if (rc) return rc;

        GemmArgs ga{};
        const bool mg = c->merged && c->a_planes == 1;
        ga.A = a_buf(c); ga.B = mg ? c->d_Bm : c->d_B; ga.C = c->ws().d_C; ga.Bt = mg ? c->d_Btm : c->d_Bt;
        ga.rows = n_arrays * nf; ga.chunk_frames = nf; ga.total_frames = n_frames; ga.frame0 = f0;
        ga.Kp = cur_kp(c); ga.Dp = c->Dp; ga.a_row_elems = c->a_row_elems;
        ga.c_plane_elems = (long long)n_arrays * n_frames * c->Dp;
        if (fused_partial) {
            ga.part = c->ws().d_part; ga.nvoiced = c->ws().d_nv; ga.D = c->D; ga.n_chunks = n_frames / SCAN_CHUNK;
            ga.part_plane_stride = (long long)n_arrays * ga.n_chunks * c->D;
            ga.scan_w[31] = 1 - 0.8f;                                              // as pa.one_minus_mu / pa.mu below
            for (int t = 30; t >= 0; --t) ga.scan_w[t] = ga.scan_w[t + 1] * 0.8f;
        }
        // one split factor per call (the scan sums the same number of partial maps for every frame): the one that
        // suits the full-size chunks; a shorter last chunk may still fall back to the 128 x 192 kernel
        const bool v2 = plan_gemm(c, ga.rows).v2;
        const int ksplit = plan_gemm(c, (long long)n_arrays * fc).ksplit;
        c->ws().c_planes = ksplit; c->ws().c_plane = ga.c_plane_elems;
        time_begin(c, MCA_HIP_K_SRP_GEMM, st);
        if (v2) {
            const int np = c->a_planes;
            const size_t smem = (size_t)(np == 2 ? 2 : 4) * np * (256 + 384) * 64;   // 32-deep stages: two with hi + lo planes, four with one plane (all 160 KiB)
            dim3 gv((ga.rows + 255) / 256, ksplit);
#define V2_LAUNCH(K)                                                                                                      \
            do {                                                                                                          \
                HIP_TRY(c, hipFuncSetAttribute(reinterpret_cast<const void *>(&K), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem)); \
                hipLaunchKernelGGL(K, gv, dim3(512), smem, st, ga);                                                       \
            } while (0)
            const bool use_v3 = !c->kn.gemm_v2;       // (A/B switch: the 32x32x16 one-plane kernel)
            if (c->a_planes == 2) V2_LAUNCH((k_srp_gemm_f16_v2<true>));
            else if (use_v3) V2_LAUNCH(k_srp_gemm_f16_v3);
            else V2_LAUNCH((k_srp_gemm_f16_v2<false>));
#undef V2_LAUNCH
        } else {
            dim3 g2((ga.rows + 127) / 128, c->Dp == 64 ? 1 : c->Dp / 192, ksplit);
            if (c->prec == MCA_HIP_SRP_FP32) hipLaunchKernelGGL(k_srp_gemm_f32, g2, dim3(256), 0, st, ga);
            else if (c->Dp == 64 && c->a_planes == 2) hipLaunchKernelGGL((k_srp_gemm_f16<true, 64>), g2, dim3(256), 0, st, ga);
            else if (c->Dp == 64) hipLaunchKernelGGL((k_srp_gemm_f16<false, 64>), g2, dim3(256), 0, st, ga);
            else if (c->a_planes == 2) hipLaunchKernelGGL((k_srp_gemm_f16<true, 192>), g2, dim3(256), 0, st, ga);
            else hipLaunchKernelGGL((k_srp_gemm_f16<false, 192>), g2, dim3(256), 0, st, ga);
        }
        time_end(c, st);
        HIP_TRY(c, hipGetLastError());
    }
    if (c->ws().c_planes > 2) {      // deep split-K of a small batch: fold the partial maps once, the scans read one map
        const long long n4 = c->ws().c_plane / 4;       // Dp is a multiple of 64
        time_begin(c, MCA_HIP_K_FOLD, st);
        hipLaunchKernelGGL(k_sum_planes, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, c->ws().d_C, n4, c->ws().c_planes, c->ws().c_plane);
        time_end(c, st);
        HIP_TRY(c, hipGetLastError());
        c->ws().c_planes = 1;
    }

    return MCA_HIP_OK;
}

// _energyMemoryFactor and its float complement (SteeringBeamforming.h:70, .cpp:134,139: float arithmetic) -- one place for every launch
constexpr float ENERGY_MU = 0.8f, ENERGY_ONE_MINUS_MU = 1 - 0.8f;

// One pass of the whole-row repair behind the list-mode analysis: the exact rows of the listed units [list0, list0 + pass_rows /
// REPAIR_GROUP) in w.d_Ax -> three-product contraction in K segments (w.d_Cx) -> summed into the map by k_repair_patch (pp: where the
// rows go; list0, pass_rows, col_tiles, ksplit and items are filled in here).  Shared by localise_impl and settle_history.
static void launch_repair_contraction(mca_hip_ctx *c, Workspace &w, long long list0, long long pass_rows, int ksplit_arrays, RepairPatchArgs pp, hipStream_t st,
                                      const int *list = nullptr, const int *n_list = nullptr, int *need = nullptr)
{
    if (!list) { list = w.d_list; n_list = w.d_nlist; need = w.d_need; }
    GemmArgs ga{};
    ga.A = w.d_Ax; ga.B = c->d_B; ga.C = w.d_Cx; ga.Bt = c->d_Bt;
    ga.rows = (int)pass_rows; ga.chunk_frames = (int)pass_rows; ga.total_frames = (int)pass_rows; ga.frame0 = 0;
    ga.Kp = c->Kp; ga.Dp = c->Dp; ga.a_row_elems = c->a_row_elems; ga.c_plane_elems = pass_rows * c->Dp;
    ga.n_list = n_list; ga.list0 = (int)list0;
    ga.repair_ksplit = repair_ksplit_for(c, ksplit_arrays); ga.repair_items = c->kn.repair_items;
    const int col_tiles = c->Dp == 64 ? 1 : c->Dp / 192;
    const long long max_work = (pass_rows + 127) / 128 * col_tiles * ga.repair_ksplit;
    dim3 gg((unsigned)std::min<long long>(max_work, 768));
    if (c->Dp == 64) hipLaunchKernelGGL((k_srp_gemm_repair<64>), gg, dim3(256), 0, st, ga);
    else hipLaunchKernelGGL((k_srp_gemm_repair<192>), gg, dim3(256), 0, st, ga);
    pp.Cx = w.d_Cx; pp.pass_rows = (int)pass_rows; pp.col_tiles = col_tiles; pp.ksplit = ga.repair_ksplit; pp.items = ga.repair_items;
    pp.list = list; pp.n_list = n_list; pp.list0 = (int)list0; pp.need = need; pp.Dp = c->Dp;
    hipLaunchKernelGGL(k_repair_patch, dim3((unsigned)std::min<long long>(pass_rows, 2048)), dim3(128), 0, st, pp);
}

// Lazy tails (mca_internal.h, HIST_FRAMES): the repair of a call's last rows is left to the call that needs them.  Settling the debt --
// for every consumer of the state that is not the next lazy call itself (an FP16X3 call of the same context, another array count,
// state_save, a recorded graph): the previous call's last HIST_FRAMES rows of EVERY array are recomputed exactly from the kept PCM (the
// repair pass's own kernels on a list of all history units) and the state becomes what the eager form would have left.
static int settle_history(mca_hip_ctx *c, hipStream_t st)
{
    if (!c->hist_pending) return MCA_HIP_OK;
    const int n = c->hist_n, units = n * HIST_UNITS, rows = units * REPAIR_GROUP, cur = c->hist_cur;
    Workspace &w = c->lanes[0];
    if (!w.d_list || !w.d_need || !w.d_nlist || !w.d_Ax || !w.d_Cx) return fail(c, MCA_HIP_ERR_HIP, "lazy tails: the adaptive workspace is gone");
    // The workspace was sized by the lazy call that left the debt (ensure_adapt_workspace: repair_pass_rows of its shape, capped by the
    // budget, MCA_HIP_WS_MAX_MB); the history's rows go through it in passes of at most that many rows, as localise_impl's do (ADVICE r5:
    // one pass of all of them wrote past d_Ax / d_Cx when the budget was small and the arrays many).
    const int ksplit = repair_ksplit_for(c, n);
    long long pass_rows = std::min<long long>((rows + 127) / 128 * 128, repair_pass_rows(c, n, HIST_FRAMES));
    const long long fit_a = (long long)(w.ax_bytes / ((size_t)2 * c->Kp * 2)) / 128 * 128, fit_c = (long long)(w.cx_bytes / ((size_t)ksplit * c->Dp * 4)) / 128 * 128;
    pass_rows = std::min(pass_rows, std::min(fit_a, fit_c));
    if (pass_rows < 128) return fail(c, MCA_HIP_ERR_HIP, "lazy tails: the adaptive workspace is smaller than one repair pass");
    const int pass_groups = (int)(pass_rows / REPAIR_GROUP);
    const int planes_before = c->a_planes;
    set_call_planes(c, 2);
    hipLaunchKernelGGL(k_hist_list, dim3((units + 255) / 256), dim3(256), 0, st, w.d_list, w.d_nlist, w.d_need, units);
    int rc = MCA_HIP_OK;
    for (long long g0 = 0; g0 < units && !rc; g0 += pass_groups) {
        StftPhatArgs sa{};
        sa.pcm = c->d_hist_pcm[cur]; sa.array_stride = (long long)c->M * HIST_SAMPLES; sa.mic_stride = HIST_SAMPLES;
        sa.M = c->M; sa.n_frames = HIST_FRAMES; sa.frame0 = 0; sa.fpb = REPAIR_GROUP; sa.total_frames = HIST_FRAMES;
        sa.window = c->d_window; sa.A = w.d_Ax; sa.Kp = c->Kp; sa.a_row_elems = c->a_row_elems; sa.a_planes = 2;
        sa.N = c->N; sa.logH = c->logH; sa.kg = c->K; sa.ula = c->ula ? 1 : 0; sa.tw = c->d_tw;
        sa.list = w.d_list; sa.n_list = w.d_nlist; sa.list0 = (int)g0; sa.list_cap = pass_groups; sa.groups_per_array = HIST_UNITS;
        sa.hist_in = c->d_hist_pcm[cur]; sa.hist_base = 0;
        const size_t smem1 = ((size_t)c->M * FFT_SCRATCH + TW_WORDS + (size_t)sa.fpb * c->M) * sizeof(float2) + (size_t)sa.fpb * 8 * sizeof(float);
        rc = launch_stft<_Float16>(c, sa, dim3(std::min(pass_groups, 512), 1), smem1, st);
        if (rc) break;
        RepairPatchArgs pp{};
        pp.groups_per_array = HIST_UNITS;
        pp.C = w.d_C; pp.c_planes = 1; pp.c_plane_stride = 0; pp.n_frames = HIST_FRAMES;
        pp.hist_C = c->d_hist_C[cur]; pp.hist_base = 0;
        launch_repair_contraction(c, w, g0, pass_rows, n, pp, st);
    }
    if (!rc) {
        hipLaunchKernelGGL(k_hist_settle, dim3(n), dim3(std::max(round_up(c->D, 64), 64)), 0, st, c->d_ehist[cur], c->d_hist_C[cur], c->d_E[c->e_cur], w.d_nlist, c->D, c->Dp,
                           ENERGY_MU, ENERGY_ONE_MINUS_MU);
        if (hipGetLastError() != hipSuccess) rc = fail(c, MCA_HIP_ERR_HIP, "lazy tails: a launch of the settling pass failed");
    }
    set_call_planes(c, planes_before);
    if (!rc) c->hist_pending = false;           // (only now: a settling pass that failed leaves the debt standing and is tried again by the next consumer -- ADVICE r5)
    return rc;
}

// the localisation stage of the arrays [c->a0, c->a0 + n_arrays) on the workspace of lane c->cur_lane; all pointers already
// point at the lane's first array
static int localise_impl(mca_hip_ctx *c, const float *pcm, long long array_stride, long long mic_stride,
                         int n_arrays, int n_frames, int *doa_bin, float *doa_rad, float *prob, float *energy, hipStream_t st)
{
    int rc;
    const size_t a0 = (size_t)c->a0;
    const bool adaptive = adaptive_applies(c, n_arrays, n_frames);
    // Lazy tails: this call leaves the exact repair of its last rows to the next one (lazy), and repairs the previous call's last rows
    // itself where one of its own first frames needs them (hist_valid).  Any other consumer of the state settles the debt first.
    const bool lazy = adaptive && c->lazy_entry && !c->capturing && lazy_context(c) && c->a0 == 0 && n_frames >= 2 * SCAN_CHUNK && SCAN_CHUNK == 32;
    const bool hist_valid = lazy && c->hist_pending && c->hist_n == n_arrays;
    if (c->hist_pending && !hist_valid) {
        if (c->capturing) return fail(c, MCA_HIP_ERR_HIP, "lazy tails: a recording found an unsettled state (mca_hip_graph_launch settles it first)");
        if ((rc = settle_history(c, st))) return rc;
    }
    c->lazy_now = lazy;
    if (c->prec == MCA_HIP_SRP_ADAPTIVE) set_call_planes(c, adaptive ? 1 : 2);
    const int n_chunks = (n_frames + SCAN_CHUNK - 1) / SCAN_CHUNK;
    if (adaptive && (rc = ensure_adapt_workspace(c, n_arrays, n_frames, n_chunks))) { c->lazy_now = false; return rc; }
    rc = run_correlation_map(c, pcm, array_stride, mic_stride, n_arrays, n_frames, st);
    c->lazy_now = false;
    if (rc) return rc;

    const bool gate = c->cfg.use_power_floor != 0;
    time_begin(c, MCA_HIP_K_SCAN_PICK, st);
    if (gate) {
        GateArgs gg{};
        gg.power_lin = c->ws().d_power; gg.n_frames = n_frames; gg.fft_n = c->N;
        gg.needed_samples = (int)(3.0 * c->cfg.sample_rate);              // _durationToEstimatePowerFloor (SoundLocalisationImpl.h:77)
        gg.margin_db = 3.f;                                                // _noiseMarginDB (BeamformingSeparationAndLocalistaion.h:52)
        gg.state = c->d_gate_state + a0 * 4; gg.voiced = c->ws().d_voiced; gg.power_out = c->ws().d_power_out;
        hipLaunchKernelGGL(k_gate, dim3(n_arrays), dim3(256), 0, st, gg);
    }
    ScanPickArgs pa{};
    pa.C = c->ws().d_C; pa.c_planes = c->ws().c_planes; pa.c_plane_stride = c->ws().c_plane; pa.n_frames = n_frames; pa.Dp = c->Dp; pa.D = c->D; pa.P = c->P; pa.S = c->S;
    pa.chunk = SCAN_CHUNK; pa.n_chunks = n_chunks;
    pa.mu = ENERGY_MU; pa.one_minus_mu = ENERGY_ONE_MINUS_MU;                 // SteeringBeamforming.h:70, .cpp:134,139 (float arithmetic)
    pa.inv_norm = exact_reciprocal(30.f * (float)c->P);
    pa.state_in = c->d_E[c->e_cur] + a0 * c->D; pa.state_out = c->d_E[c->e_cur ^ 1] + a0 * c->D;
    pa.part_planes = c->ws().partial_done ? c->ws().part_planes : 1; pa.part_plane_stride = (long long)n_arrays * n_chunks * c->D;
    pa.part = c->ws().d_part; pa.nvoiced = c->ws().d_nv; pa.e_start = c->ws().d_estart; pa.voiced = gate ? c->ws().d_voiced : nullptr;
    pa.grid = c->d_grid; pa.doa_bin = doa_bin; pa.doa_rad = doa_rad; pa.prob = prob; pa.energy = energy;
    const int gpa = (n_frames + REPAIR_GROUP - 1) / REPAIR_GROUP;
    const bool mixed = adaptive && cand_mixed(c, lazy, gate);
    if (adaptive) {
        pa.mode = 1; pa.tau = c->tau_en; pa.flags = c->ws().d_flags; pa.groups_per_array = gpa;
        pa.need = c->ws().d_need; pa.list = c->ws().d_list; pa.n_list = c->ws().d_nlist; pa.chunk_from = c->ws().d_chunk_from; pa.last_vchunk = c->ws().d_last_vchunk; pa.stats = c->d_rstats;
        pa.unsure = wave16_applies(c) ? c->ws().d_unsure : nullptr;
        pa.hist_base = n_arrays * gpa;
        if (lazy) { pa.lazy = 1; pa.hist_C_out = c->d_hist_C[c->hist_cur ^ 1]; pa.e_hist_out = c->d_ehist[c->hist_cur ^ 1]; }
        if (hist_valid) { pa.hist_valid = 1; pa.hist_C_in = c->d_hist_C[c->hist_cur]; pa.e_hist_in = c->d_ehist[c->hist_cur]; }
        if (cand_call(c, lazy)) { pa.umask = c->ws().d_umask; pa.umask_words = c->Dp / 32; pa.dead = c->ws().d_unsure; }   // (dead: the unsure bytes, unused by these contexts)
        else if (mixed) {
            pa.umask = c->ws().d_umask; pa.umask_words = c->Dp / 32;
            pa.need_full = c->ws().d_need_full; pa.list_full = c->ws().d_list_full; pa.n_list_full = c->ws().d_nlist + 3;
        }
        pa.clist = c->ws().d_chunk_from + c->ws().adapt_chunks; pa.n_clist = c->ws().d_nlist + 1;   // (+ 2: see ScanPickArgs)
        if (c->h_probe && !c->capturing) {
            c->fb_frames_ring[c->fb_calls % FB_RING] = c->adapt_frames_total + (unsigned long long)n_arrays * n_frames;
            c->fb_launch_m[c->fb_calls % FB_RING] = c->fb_m > 0 ? c->fb_m - 1 : 0;       // (the eligible call this piece belongs to)
            pa.probe = c->h_probe + 4 * (c->fb_calls % FB_RING); pa.probe_seq = ++c->fb_calls;
        }
    }
    const int nthr = round_up(c->D, 64);
    dim3 g3(pa.n_chunks, n_arrays);
    if (!c->ws().partial_done) hipLaunchKernelGGL(k_scan_partial, g3, dim3(round_up(c->Dp / 4, 64)), 0, st, pa);   // a thread per four delays
    // ungated calls: k_scan_pick composes its chunk's start value itself (the chunks further back than four have decayed below the
    // last bit); with the gate a chunk may hold no advancing frame at all and the composition runs over all of them, in order
    // The look-back drops g^4 E_{c-4} (g = 0.8^32: 3.9e-13 of E), which is below the last bit only while the map is bounded: under PHAT
    // |C| <= P whatever the level.  With gcc_weighting NONE the map scales with the amplitude squared, and after a loud passage that
    // falls to digital silence the exact recursion still carries the old peak for ~180 frames when the look-back has zeroed it
    // (ADVICE r3): NONE always runs the serial carry pass.
    pa.lookback = (!gate && !c->kn.scan_carry && c->cfg.gcc_weighting == MCA_HIP_GCC_PHAT) ? 4 : 0;
    if (!pa.lookback) hipLaunchKernelGGL(k_scan_carry, dim3(n_arrays, nthr / 64), dim3(64), 0, st, pa);   // one wave per 64 delays
    const size_t smem3 = (size_t)SCAN_SUB * (c->Dp + 8) * sizeof(float);
    const int ppl = c->D - 2 <= 128 ? 2 : c->D - 2 <= 384 ? 6 : 8;            // positions per lane of the peak pick
#define LAUNCH_PICK(PL, MODE)                                                                                                       \
    do {                                                                                                                            \
        if (smem3 > 64 * 1024)                                                                                                      \
            HIP_TRY(c, hipFuncSetAttribute(reinterpret_cast<const void *>(k_scan_pick<PL, MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem3)); \
        hipLaunchKernelGGL((k_scan_pick<PL, MODE>), g3, dim3(std::max(nthr, 512)), smem3, st, pa);   /* 8 waves: the per-frame pick is one wave per frame */ \
    } while (0)
    if (mixed) { if (ppl == 2) LAUNCH_PICK(2, 2); else if (ppl == 6) LAUNCH_PICK(6, 2); else LAUNCH_PICK(8, 2); }
    else if (adaptive) { if (ppl == 2) LAUNCH_PICK(2, 1); else if (ppl == 6) LAUNCH_PICK(6, 1); else LAUNCH_PICK(8, 1); }
    else { if (ppl == 2) LAUNCH_PICK(2, 0); else if (ppl == 6) LAUNCH_PICK(6, 0); else LAUNCH_PICK(8, 0); }
#undef LAUNCH_PICK
    DoaFillArgs fa{};
    fa.voiced = c->ws().d_voiced; fa.n_frames = n_frames; fa.S = c->S;
    fa.doa_bin = doa_bin; fa.doa_rad = doa_rad; fa.prob = prob;
    fa.last_bin = c->d_last_bin + a0 * c->S; fa.last_rad = c->d_last_rad + a0 * c->S; fa.last_prob = c->d_last_prob + a0 * c->S;
    if (gate && !adaptive) hipLaunchKernelGGL(k_doa_fill, dim3(n_arrays), dim3(1024), 0, st, fa);   // (ADAPTIVE: after the second pick)
    time_end(c, st);
    HIP_TRY(c, hipGetLastError());
    if (adaptive) {
        // exact repair: list the row groups the flagged frames depend on, recompute them with the hi + lo operand planes and
        // the three-product contraction, patch them into the map, pick the flagged chunks again (k_scan_pick, mode 2).
        // Every launch is sized for the worst case (all rows) and exits at the device-side count.
        time_begin(c, MCA_HIP_K_REPAIR, st);
        const long long pass_rows = repair_pass_rows(c, n_arrays, n_frames);
        const int pass_groups = (int)(pass_rows / REPAIR_GROUP);
        const long long all_groups = (long long)n_arrays * (gpa + (hist_valid ? HIST_UNITS : 0));     // (the list cannot be longer)
        set_call_planes(c, 2);
        for (long long g0 = 0; g0 < all_groups; g0 += pass_groups) {
            StftPhatArgs sa{};
            sa.pcm = pcm; sa.array_stride = array_stride; sa.mic_stride = mic_stride;
            sa.M = c->M; sa.n_frames = n_frames; sa.frame0 = 0; sa.fpb = REPAIR_GROUP; sa.total_frames = n_frames;
            sa.window = c->d_window; sa.A = c->ws().d_Ax; sa.Kp = c->Kp; sa.a_row_elems = c->a_row_elems; sa.a_planes = 2;
            sa.N = c->N; sa.logH = c->logH; sa.kg = c->K; sa.ula = c->ula ? 1 : 0; sa.tw = c->d_tw;
            sa.list = c->ws().d_list; sa.n_list = c->ws().d_nlist; sa.list0 = (int)g0; sa.list_cap = pass_groups; sa.groups_per_array = gpa;
            if (hist_valid) { sa.hist_in = c->d_hist_pcm[c->hist_cur]; sa.hist_base = n_arrays * gpa; }
            CandArgs ca{};
            if (pa.umask) {
                // candidate columns: the exact values where the flagged frames need them, straight into the map (no partial maps, no patch) --
                // by the workgroup that writes a unit's rows, inside the list-mode launch (4 / 8 microphones: k_stft_phat_wave)
                ca.A = c->ws().d_Ax; ca.B = c->d_B; ca.Kp = c->Kp; ca.Dp = c->Dp; ca.a_row_elems = c->a_row_elems;
                ca.list = c->ws().d_list; ca.n_list = c->ws().d_nlist; ca.list0 = (int)g0; ca.pass_rows = (int)pass_rows;
                ca.umask = c->ws().d_umask; ca.umask_words = pa.umask_words; ca.need = c->ws().d_need; ca.groups_per_array = gpa; ca.n_frames = n_frames;
                ca.C = c->ws().d_C; ca.c_planes = c->ws().c_planes; ca.c_plane_stride = c->ws().c_plane;
                if (hist_valid) { ca.hist_C = c->d_hist_C[c->hist_cur]; ca.hist_base = n_arrays * gpa; }
                if (c->kn.cand_fuse && (c->M == 8 || c->M == 4) && !c->kn.stft_wg) { sa.cand_on = 1; sa.cand = ca; }
            }
            const size_t smem1 = ((size_t)c->M * FFT_SCRATCH + TW_WORDS + (size_t)sa.fpb * c->M) * sizeof(float2) + (size_t)sa.fpb * 8 * sizeof(float);
            // fixed, moderate grids: the kernels of the repair pass walk their device-side work lists
            if ((rc = launch_stft<_Float16>(c, sa, dim3(std::min(pass_groups, std::max(1, c->kn.list_grid)), 1), smem1, st))) { set_call_planes(c, 1); return rc; }
            if (pa.umask && sa.cand_on) continue;
            if (pa.umask) {
                const long long max_items = pass_rows / REPAIR_GROUP;
                hipLaunchKernelGGL(k_srp_cand, dim3((unsigned)std::min<long long>(max_items, std::max(1, c->kn.cand_grid))), dim3(1024), 0, st, ca);
                continue;
            }
            RepairPatchArgs pp{};
            pp.groups_per_array = gpa;
            pp.C = c->ws().d_C; pp.c_planes = c->ws().c_planes; pp.c_plane_stride = c->ws().c_plane; pp.n_frames = n_frames;
            if (hist_valid) { pp.hist_C = c->d_hist_C[c->hist_cur]; pp.hist_base = n_arrays * gpa; }
            launch_repair_contraction(c, c->ws(), g0, pass_rows, n_arrays, pp, st);
        }
        // two work lists: the units of the frames that take whole rows (tails, unsure rows) through the whole-row kernels
        for (long long g0 = 0; mixed && g0 < all_groups; g0 += pass_groups) {
            StftPhatArgs sa{};
            sa.pcm = pcm; sa.array_stride = array_stride; sa.mic_stride = mic_stride;
            sa.M = c->M; sa.n_frames = n_frames; sa.frame0 = 0; sa.fpb = REPAIR_GROUP; sa.total_frames = n_frames;
            sa.window = c->d_window; sa.A = c->ws().d_Ax; sa.Kp = c->Kp; sa.a_row_elems = c->a_row_elems; sa.a_planes = 2;
            sa.N = c->N; sa.logH = c->logH; sa.kg = c->K; sa.ula = c->ula ? 1 : 0; sa.tw = c->d_tw;
            sa.list = pa.list_full; sa.n_list = pa.n_list_full; sa.list0 = (int)g0; sa.list_cap = pass_groups; sa.groups_per_array = gpa;
            const size_t smem1 = ((size_t)c->M * FFT_SCRATCH + TW_WORDS + (size_t)sa.fpb * c->M) * sizeof(float2) + (size_t)sa.fpb * 8 * sizeof(float);
            if ((rc = launch_stft<_Float16>(c, sa, dim3(std::min(pass_groups, std::max(1, c->kn.list_grid)), 1), smem1, st))) { set_call_planes(c, 1); return rc; }
            RepairPatchArgs pp{};
            pp.groups_per_array = gpa;
            pp.C = c->ws().d_C; pp.c_planes = c->ws().c_planes; pp.c_plane_stride = c->ws().c_plane; pp.n_frames = n_frames;
            launch_repair_contraction(c, c->ws(), g0, pass_rows, n_arrays, pp, st, pa.list_full, pa.n_list_full, pa.need_full);
        }
        set_call_planes(c, 1);
        const size_t smem4 = (size_t)32 * (c->Dp + 8) * sizeof(float);
        const int repick_grid = std::max(1, c->kn.repick_grid);
#define LAUNCH_REPICK(PL)                                                                                                           \
        do {                                                                                                                        \
            if (smem4 > 64 * 1024)      /* grids finer than 0.45 degrees: Dp >= 512 */                                              \
                HIP_TRY(c, hipFuncSetAttribute(reinterpret_cast<const void *>(&k_scan_repick<PL>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem4)); \
            hipLaunchKernelGGL((k_scan_repick<PL>), dim3((unsigned)std::min<long long>((long long)pa.n_chunks * n_arrays, repick_grid)), dim3(std::max(nthr, 512)), smem4, st, pa); \
        } while (0)
        if (ppl == 2) LAUNCH_REPICK(2); else if (ppl == 6) LAUNCH_REPICK(6); else LAUNCH_REPICK(8);
#undef LAUNCH_REPICK
        if (gate) hipLaunchKernelGGL(k_doa_fill, dim3(n_arrays), dim3(1024), 0, st, fa);   // gated-out frames repeat the last FINAL pick
        time_end(c, st);
        HIP_TRY(c, hipGetLastError());
        c->adapt_frames_total += (unsigned long long)n_arrays * n_frames;
    }
    if (lazy) { c->hist_cur ^= 1; c->hist_n = n_arrays; c->hist_pending = true; }
    else c->hist_pending = false;                // (settled above, or consumed by an eager adaptive call: its own last frame was repaired)
    c->ws().last_a0 = c->a0; c->ws().last_arrays = n_arrays;
    return MCA_HIP_OK;
}

int mca_hip_get_repair_stats(mca_hip_ctx *c, unsigned long long *frames, unsigned long long *flagged_frames,
                             unsigned long long *recomputed_frames)
{
    if (!c) return MCA_HIP_ERR_INVALID_ARGUMENT;
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    HIP_TRY(c, hipDeviceSynchronize());
    unsigned long long st[2] = {0, 0};
    HIP_TRY(c, hipMemcpy(st, c->d_rstats, sizeof(st), hipMemcpyDeviceToHost));
    if (frames) *frames = c->adapt_frames_total;
    if (flagged_frames) *flagged_frames = st[0];
    if (recomputed_frames) *recomputed_frames = st[1] * REPAIR_GROUP;
    return MCA_HIP_OK;
}

int mca_hip_get_repair_columns(mca_hip_ctx *c, unsigned long long *candidate_columns, unsigned long long *whole_row_frames)
{
    if (!c) return MCA_HIP_ERR_INVALID_ARGUMENT;
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    HIP_TRY(c, hipDeviceSynchronize());
    unsigned long long st[4] = {0, 0, 0, 0};
    HIP_TRY(c, hipMemcpy(st, c->d_rstats, sizeof(st), hipMemcpyDeviceToHost));
    if (candidate_columns) *candidate_columns = st[2];
    if (whole_row_frames) *whole_row_frames = st[3];
    return MCA_HIP_OK;
}

// the wave-per-run beamformer (k_beamform_wave) serves the 1024-sample path (one grid row of workgroups per source) when the caller's DOAs are grid
// bins (the localiser's own picks); everything else stays on k_beamform_ola / _512 / _gen
static bool wave_beamformer_applies(const mca_hip_ctx *c)
{
    // several sources: up to 8 microphones k_beamform_wave_ms (forward transforms shared, the pair spectra in registers); more
    // microphones with two sources a grid row of k_beamform_wave per source, with three or four k_beamform_ola
    return !c->kn.bf_ola && !c->generic && !c->n512 && (c->S <= 2 || c->M <= 8) && c->M >= 2 && c->M <= MCA_MAX_MICS;
}

// steering rows of every grid angle (+ the initial DOA): allocated and built once, outside any capture
static int ensure_bf_table(mca_hip_ctx *c)
{
    if (!c->d_bftab && c->n2048) {
        // 2048-sample frames: a row per (angle, CHANNEL), [D + 1][M][1032] (k_beamform_wave_2048)
        HIP_TRY(c, hipMalloc((void **)&c->d_bftab, (size_t)(c->D + 1) * c->M * 1032 * sizeof(float2)));
        c->bf_pairs = c->M;
        const double unit = (double)c->cfg.sample_rate / (double)c->N / 346.1;     // Beamformer.cpp:59 without 2 pi
        hipLaunchKernelGGL(k_bf_table_2048, dim3(c->D + 1, c->M), dim3(256), 0, nullptr, c->d_bftab, c->d_grid, c->d_micx, c->M, unit);
        HIP_TRY(c, hipGetLastError());
        HIP_TRY(c, hipDeviceSynchronize());
        return MCA_HIP_OK;
    }
    if (c->d_bftab || !wave_beamformer_applies(c)) return MCA_HIP_OK;
    const int np = (c->M + 1) / 2;
    HIP_TRY(c, hipMalloc((void **)&c->d_bftab, (size_t)(c->D + 1) * np * 1024 * sizeof(float2)));
    c->bf_pairs = np;
    const double unit = (double)c->cfg.sample_rate / (double)FFT_N / 346.1;     // Beamformer.cpp:59 without 2 pi
    hipLaunchKernelGGL(k_bf_table, dim3(c->D + 1, np), dim3(256), 0, nullptr, c->d_bftab, c->d_grid, c->d_micx, c->M, np, unit);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipDeviceSynchronize());
    return MCA_HIP_OK;
}

static int separate_impl(mca_hip_ctx *c, const float *pcm, long long array_stride, long long mic_stride,
                         int n_arrays, int n_frames, const float *doa_rad, float *out_pcm, hipStream_t st, const int *doa_bin = nullptr)
{
    const size_t a0 = (size_t)c->a0;
    if (doa_bin && c->d_bftab && c->n2048 && (reinterpret_cast<uintptr_t>(out_pcm) & 7) == 0) {
        // 2048-sample frames, the localiser's own picks: one wave per run of frames, a transform per channel (k_beamform_wave_2048)
        BeamformWaveArgs wa{};
        wa.pcm = pcm; wa.array_stride = array_stride; wa.mic_stride = mic_stride;
        wa.M = c->M; wa.n_pairs = c->M; wa.n_frames = n_frames; wa.S = c->S;
        // frames per wave: a workgroup covers 4 ft - 1 frames (one analysed twice); the smallest ft whose workgroups fit ONE resident round of
        // two per CU (255 registers: two waves per SIMD), 16 when no such ft exists (many rounds anyway)
        wa.ft = 16;
        for (int ft = 2; ft <= 64; ++ft)
            if ((long long)n_arrays * c->S * ((n_frames + 4 * ft - 2) / (4 * ft - 1)) <= 2LL * c->n_cu) { wa.ft = ft; break; }
        wa.window = c->d_window; wa.doa_bin = doa_bin; wa.table = c->d_bftab; wa.out = out_pcm;
        wa.tail_in = c->d_tail[c->tail_cur] + a0 * c->S * c->H; wa.tail_out = c->d_tail[c->tail_cur ^ 1] + a0 * c->S * c->H;
        const int wgs = (n_frames + 4 * wa.ft - 2) / (4 * wa.ft - 1);
        const size_t smem_w = (size_t)(F1K_TWORDS + 4 * F1K_SCRATCH + 4 * 512) * sizeof(float2);
        time_begin(c, MCA_HIP_K_BEAMFORM, st);
        hipLaunchKernelGGL(k_beamform_wave_2048, dim3(wgs, n_arrays, c->S), dim3(256), smem_w, st, wa);
        time_end(c, st);
        HIP_TRY(c, hipGetLastError());
        return MCA_HIP_OK;
    }
    if (doa_bin && c->d_bftab && wave_beamformer_applies(c)) {
        BeamformWaveArgs wa{};
        wa.pcm = pcm; wa.array_stride = array_stride; wa.mic_stride = mic_stride;
        wa.M = c->M; wa.n_pairs = c->bf_pairs; wa.n_frames = n_frames; wa.S = c->S;
        // frames per run (one wave each; every run re-analyses one extra frame for its overlap-add carry): long runs are
        // cheaper per frame, short ones fill the chip -- two waves per SIMD want 2048 runs
        const int ft_env = c->kn.bfw_ft;
        wa.ft = ft_env > 0 ? ft_env : 16;
        while (!ft_env && wa.ft > 2 && (long long)n_arrays * c->S * ((n_frames + wa.ft - 1) / wa.ft) < 2048) wa.ft >>= 1;
        wa.window = c->d_window; wa.doa_bin = doa_bin; wa.table = c->d_bftab; wa.out = out_pcm;
        wa.tail_in = c->d_tail[c->tail_cur] + a0 * c->S * c->H; wa.tail_out = c->d_tail[c->tail_cur ^ 1] + a0 * c->S * c->H;
        if (c->S >= 2 && c->M <= 8) {
            // the forward transforms of a frame shared by its sources
            wa.ft = 16;
            while (wa.ft > 2 && (long long)n_arrays * ((n_frames + wa.ft - 1) / wa.ft) < 2048) wa.ft >>= 1;
            const int runs = (n_frames + wa.ft - 1) / wa.ft;
            const size_t smem_ms = (size_t)(F1K_TWORDS + 4 * F1K_SCRATCH) * sizeof(float2) + (size_t)4 * c->S * FFT_H * sizeof(float);
            const dim3 gms((runs + 3) / 4, n_arrays);
            time_begin(c, MCA_HIP_K_BEAMFORM, st);
#define BFMS(NPT)                                                                                                                   \
            do {                                                                                                                    \
                if (smem_ms > 64 * 1024) {                                                                                          \
                    HIP_TRY(c, hipFuncSetAttribute(reinterpret_cast<const void *>(&k_beamform_wave_ms<NPT, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_ms)); \
                    HIP_TRY(c, hipFuncSetAttribute(reinterpret_cast<const void *>(&k_beamform_wave_ms<NPT, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_ms));  \
                }                                                                                                                   \
                if (c->M & 1) hipLaunchKernelGGL((k_beamform_wave_ms<NPT, true>), gms, dim3(256), smem_ms, st, wa);                 \
                else hipLaunchKernelGGL((k_beamform_wave_ms<NPT, false>), gms, dim3(256), smem_ms, st, wa);                         \
            } while (0)
            if (c->bf_pairs == 1) {          // two microphones: one pair (never odd)
                if (smem_ms > 64 * 1024) HIP_TRY(c, hipFuncSetAttribute(reinterpret_cast<const void *>(&k_beamform_wave_ms<1, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_ms));
                hipLaunchKernelGGL((k_beamform_wave_ms<1, false>), gms, dim3(256), smem_ms, st, wa);
            }
            else if (c->bf_pairs == 2) BFMS(2);
            else if (c->bf_pairs == 3) BFMS(3);
            else BFMS(4);
#undef BFMS
            time_end(c, st);
            HIP_TRY(c, hipGetLastError());
            return MCA_HIP_OK;
        }
        const int abl = c->kn.bfw_abl & 3;     // (-DMCA_MEASURE only: ablations with wrong results)
        int var = abl ? 14 : (c->kn.bfw_var & 31);
        // (MEASURE builds, MCA_HIP_BFW_VAR=31) the staged variant needs 16-byte aligned rows: otherwise the shipped one
        if ((var & 16) && (var != 31 || (mic_stride & 3) || (array_stride & 3) || (reinterpret_cast<uintptr_t>(pcm) & 15))) var = 15;
        // workgroups per array: 4 runs of ft frames each, or (hand-off of the overlap-add carries inside the workgroup, VAR bit 1)
        // 4 ft - 1 frames.  With the hand-off ft is the smallest run length whose workgroups are all resident at once (two per
        // CU: 512) -- one workgroup more than that costs a whole extra round.
        if ((var & 2) && !ft_env) {
            wa.ft = 2;
            while (wa.ft < 256 && (long long)n_arrays * c->S * ((n_frames + 4 * wa.ft - 2) / (4 * wa.ft - 1)) > 512) ++wa.ft;
        }
        const int wgs = (var & 2) ? (n_frames + 4 * wa.ft - 2) / (4 * wa.ft - 1) : ((n_frames + wa.ft - 1) / wa.ft + 3) / 4;
        const size_t smem = (size_t)(F1K_TWORDS + 4 * F1K_SCRATCH) * sizeof(float2) + ((var & 2) ? 4 * FFT_H * sizeof(float) : 0) + ((var & 16) ? 3 * 8192 : 0);
        // one resident round of (nearly) two workgroups per CU: the older workgroup of every CU takes longer runs (BeamformWaveArgs::skew)
        dim3 gb(wgs, n_arrays, c->S);
        if ((var & 2) && !abl && c->kn.bfw_skew != 0 && c->S == 1 && (wgs & 1) == 0 && wa.ft >= 8 &&
            (long long)wgs * n_arrays > (long long)c->n_cu * 3 / 2 && (long long)wgs * n_arrays <= 2LL * c->n_cu) {
            wa.skew = c->kn.bfw_skew > 0 ? c->kn.bfw_skew : (3 * wa.ft + 8) / 16;
            if (wa.skew >= wa.ft) wa.skew = 0;
            else gb = dim3(n_arrays, wgs, c->S);
        }
        time_begin(c, MCA_HIP_K_BEAMFORM, st);
#define BFW_CASE(V) case V: if (c->M & 1) hipLaunchKernelGGL((k_beamform_wave<true, V, 0>), gb, dim3(256), smem, st, wa); \
                            else hipLaunchKernelGGL((k_beamform_wave<false, V, 0>), gb, dim3(256), smem, st, wa); break;
#ifdef MCA_MEASURE
        if (abl == 1) hipLaunchKernelGGL((k_beamform_wave<false, 14, 1>), dim3(wgs, n_arrays, c->S), dim3(256), smem, st, wa);
        else if (abl == 2) hipLaunchKernelGGL((k_beamform_wave<false, 14, 2>), dim3(wgs, n_arrays, c->S), dim3(256), smem, st, wa);
        else if (abl == 3) hipLaunchKernelGGL((k_beamform_wave<false, 14, 3>), dim3(wgs, n_arrays, c->S), dim3(256), smem, st, wa);
        else
        switch (var) {
            BFW_CASE(0) BFW_CASE(1) BFW_CASE(2) BFW_CASE(3) BFW_CASE(4) BFW_CASE(5) BFW_CASE(6) BFW_CASE(7)
            BFW_CASE(8) BFW_CASE(9) BFW_CASE(10) BFW_CASE(11) BFW_CASE(12) BFW_CASE(13) BFW_CASE(14) BFW_CASE(15)
            case 31:
                HIP_TRY(c, hipFuncSetAttribute(reinterpret_cast<const void *>(&k_beamform_wave<false, 31, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
                HIP_TRY(c, hipFuncSetAttribute(reinterpret_cast<const void *>(&k_beamform_wave<true, 31, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
                if (c->M & 1) hipLaunchKernelGGL((k_beamform_wave<true, 31, 0>), gb, dim3(256), smem, st, wa);
                else hipLaunchKernelGGL((k_beamform_wave<false, 31, 0>), gb, dim3(256), smem, st, wa);
                break;
        }
#else
        switch (var) { BFW_CASE(15) }
#endif
#undef BFW_CASE
        time_end(c, st);
        HIP_TRY(c, hipGetLastError());
        return MCA_HIP_OK;
    }
    BeamformArgs ba{};
    ba.pcm = pcm; ba.array_stride = array_stride; ba.mic_stride = mic_stride;
    ba.M = c->M; ba.Mpad = c->M; ba.S = c->S; ba.n_frames = n_frames; ba.fs = c->cfg.sample_rate;
    // frames per run: every run re-analyses one extra frame for its overlap-add carry, so long runs are cheaper per
    // frame (16 -> 64 frames: 0.344 -> 0.323 ms on the bench shape); smaller batches take shorter runs so that two
    // workgroups per CU exist (multiples of the 4-frame batch)
    ba.ft = 64;
    // (the any-length kernel walks its frames one at a time behind ~15 barriers each and is latency bound: it wants as
    // many workgroups per CU as its LDS allows, up to the 8 that fill the wave slots with 256 threads)
    long long want_wgs = 512;
    if (c->generic) {
        const size_t smem_g = (size_t)(c->M + c->S) * (c->H + 1) * sizeof(float2) + (size_t)c->S * c->H * sizeof(float);
        const long long per_cu = std::max<long long>(1, std::min<long long>((160 * 1024) / (long long)smem_g, 2048 / gen_threads(c, true)));
        want_wgs = 256 * per_cu;
    }
    while (ba.ft > BF_NB && (long long)n_arrays * ((n_frames + ba.ft - 1) / ba.ft) < want_wgs) ba.ft >>= 1;
    ba.window = c->d_window; ba.mic_x = c->d_micx; ba.doa_rad = doa_rad; ba.out = out_pcm;
    ba.tail_in = c->d_tail[c->tail_cur] + a0 * c->S * c->H; ba.tail_out = c->d_tail[c->tail_cur ^ 1] + a0 * c->S * c->H;
    ba.N = c->N; ba.logH = c->logH; ba.tw = c->d_tw;
    if (c->n512) {
        const size_t smem = ((size_t)2 * 8 * 258 + 8 * FFT_SCRATCH + TW_WIN + (size_t)2 * c->S * c->M * 41) * sizeof(float2) + (size_t)2 * c->S * sizeof(double);
        ba.ft = 64;
        const long long per_cu = smem <= 80 * 1024 ? 2 : 1;
        while (ba.ft > 4 && (long long)n_arrays * ((n_frames + ba.ft - 1) / ba.ft) < 256 * per_cu) ba.ft >>= 1;
        HIP_TRY(c, hipFuncSetAttribute(reinterpret_cast<const void *>(k_beamform_512), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        time_begin(c, MCA_HIP_K_BEAMFORM, st);
        hipLaunchKernelGGL(k_beamform_512, dim3((n_frames + ba.ft - 1) / ba.ft, n_arrays), dim3(512), smem, st, ba);
        time_end(c, st);
        HIP_TRY(c, hipGetLastError());
        return MCA_HIP_OK;
    }
    if (c->generic) {
        // all sources in one launch while M + S spectra (and the S carries) fit the 160 KiB of a CU; else as many per launch as do --
        // 16 microphones at 2048-sample frames hold two -- each pass transforming the channels again
        auto smem_for = [&](int s_pass) {
            return (size_t)(c->M + s_pass) * (c->H + 1) * sizeof(float2) + (size_t)s_pass * c->H * sizeof(float) + (size_t)(ba.ft + 1) * s_pass * sizeof(double) + 8;
        };
        int s_pass = c->S;
        while (s_pass > 1 && smem_for(s_pass) > 160 * 1024) --s_pass;
        const size_t smem = smem_for(s_pass);
        if (smem > 160 * 1024) return fail(c, MCA_HIP_ERR_UNSUPPORTED, "fft_size x n_mics exceeds the 160 KiB LDS of a CU");
        if (smem > 64 * 1024)
            HIP_TRY(c, hipFuncSetAttribute(reinterpret_cast<const void *>(k_beamform_gen), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        time_begin(c, MCA_HIP_K_BEAMFORM, st);
        ba.S_all = c->S;
        for (int s0 = 0; s0 < c->S; s0 += s_pass) {
            ba.s0 = s0; ba.S = std::min(s_pass, c->S - s0);
            hipLaunchKernelGGL(k_beamform_gen, dim3((n_frames + ba.ft - 1) / ba.ft, n_arrays), dim3(gen_threads(c, true)), smem, st, ba);
        }
        time_end(c, st);
        HIP_TRY(c, hipGetLastError());
        return MCA_HIP_OK;
    }
    ba.nb = std::max(1, BF_NB / c->S);      // 4 beamformed slots in LDS whatever the number of sources
    const size_t smem = ((size_t)ba.M + ba.nb * c->S) * FFT_SCRATCH * sizeof(float2) + (size_t)c->S * c->M * 49 * sizeof(float2) +
                        TW_WORDS * sizeof(float2) + (size_t)ba.nb * c->M * (1 + c->S) * sizeof(float2) +
                        (size_t)(ba.ft + 1) * c->S * sizeof(double);
    if (smem > 160 * 1024) return fail(c, MCA_HIP_ERR_UNSUPPORTED, "n_mics/n_sources combination exceeds the 160 KiB LDS of a CU");
    dim3 g((n_frames + ba.ft - 1) / ba.ft, n_arrays);
    time_begin(c, MCA_HIP_K_BEAMFORM, st);
    const bool occ4 = !c->kn.bf_occ2;
#define BF_LAUNCH(K)                                                                                                     \
    do {                                                                                                                 \
        if (smem > 64 * 1024)                                                                                            \
            HIP_TRY(c, hipFuncSetAttribute(reinterpret_cast<const void *>(&K), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem)); \
        hipLaunchKernelGGL(K, g, dim3(512), smem, st, ba);                                                               \
    } while (0)
    if (c->M <= 8 && occ4) BF_LAUNCH((k_beamform_ola<1, 4>));
    else if (c->M <= 8) BF_LAUNCH((k_beamform_ola<1, 2>));
    else BF_LAUNCH((k_beamform_ola<2, 2>));
#undef BF_LAUNCH
    time_end(c, st);
    HIP_TRY(c, hipGetLastError());
    return MCA_HIP_OK;
}

int mca_hip_localise_frames_dev(mca_hip_ctx *c, const float *pcm, long long array_stride, long long mic_stride,
                                int n_arrays, int n_frames, int *doa_bin, float *doa_rad, float *prob,
                                float *energy, void *stream)
{
    int rc = check_stream_args(c, pcm, array_stride, mic_stride, n_arrays, n_frames);
    if (rc) return rc;
    if (!doa_bin) return fail(c, MCA_HIP_ERR_INVALID_ARGUMENT, "doa_bin_dev is NULL");
    c->n_lanes_last = 1; c->cur_lane = 0; c->a0 = 0;
    adapt_policy_begin(c, n_arrays, n_frames);
    c->lazy_entry = !c->host_call;
    rc = localise_impl(c, pcm, array_stride, mic_stride, n_arrays, n_frames, doa_bin, doa_rad, prob, energy, (hipStream_t)stream);
    c->lazy_entry = false;
    if (rc) return rc;
    c->last_arrays = n_arrays; c->last_frames = n_frames;
    c->e_cur ^= 1;
    return MCA_HIP_OK;
}

// doa_bin: the grid bins behind doa_rad when they are the localiser's own picks (NULL: arbitrary angles)
static int separate_frames_dev_bins(mca_hip_ctx *c, const float *pcm, long long array_stride, long long mic_stride,
                                    int n_arrays, int n_frames, const float *doa_rad, float *out_pcm, void *stream, const int *doa_bin)
{
    int rc = check_stream_args(c, pcm, array_stride, mic_stride, n_arrays, n_frames);
    if (rc) return rc;
    if (!doa_rad || !out_pcm) return fail(c, MCA_HIP_ERR_INVALID_ARGUMENT, "doa_rad_dev/out_pcm_dev is NULL");
    c->n_lanes_last = 1; c->cur_lane = 0; c->a0 = 0;
    rc = separate_impl(c, pcm, array_stride, mic_stride, n_arrays, n_frames, doa_rad, out_pcm, (hipStream_t)stream, doa_bin);
    if (rc) return rc;
    c->tail_cur ^= 1;
    return MCA_HIP_OK;
}

int mca_hip_separate_frames_dev(mca_hip_ctx *c, const float *pcm, long long array_stride, long long mic_stride,
                                int n_arrays, int n_frames, const float *doa_rad, float *out_pcm, void *stream)
{
    return separate_frames_dev_bins(c, pcm, array_stride, mic_stride, n_arrays, n_frames, doa_rad, out_pcm, stream, nullptr);
}

int mca_hip_separate_frames_bins_dev(mca_hip_ctx *c, const float *pcm, long long array_stride, long long mic_stride,
                                     int n_arrays, int n_frames, const int *doa_bin, const float *doa_rad, float *out_pcm, void *stream)
{
    if (!c) return MCA_HIP_ERR_INVALID_ARGUMENT;
    if (!doa_bin) return fail(c, MCA_HIP_ERR_INVALID_ARGUMENT, "doa_bin_dev is NULL");
    int rc = ensure_bf_table(c);
    if (rc) return rc;
    return separate_frames_dev_bins(c, pcm, array_stride, mic_stride, n_arrays, n_frames, doa_rad, out_pcm, stream, doa_bin);
}

int mca_hip_process_frames_dev(mca_hip_ctx *c, const float *pcm, long long array_stride, long long mic_stride,
                               int n_arrays, int n_frames, int *doa_bin, float *doa_rad, float *prob,
                               float *energy, float *out_pcm, void *stream)
{
    if (!c) return MCA_HIP_ERR_INVALID_ARGUMENT;
    if (!doa_rad) return fail(c, MCA_HIP_ERR_INVALID_ARGUMENT, "doa_rad_dev is NULL (the separation stage steers with it)");
    int rc = check_stream_args(c, pcm, array_stride, mic_stride, n_arrays, n_frames);
    if (rc) return rc;
    if (!doa_bin) return fail(c, MCA_HIP_ERR_INVALID_ARGUMENT, "doa_bin_dev is NULL");
    if (!out_pcm) return fail(c, MCA_HIP_ERR_INVALID_ARGUMENT, "doa_rad_dev/out_pcm_dev is NULL");
    if ((rc = ensure_bf_table(c))) return rc;
    c->n_lanes_last = 1; c->cur_lane = 0; c->a0 = 0;
    adapt_policy_begin(c, n_arrays, n_frames);
    c->lazy_entry = !c->host_call;
    rc = localise_impl(c, pcm, array_stride, mic_stride, n_arrays, n_frames, doa_bin, doa_rad, prob, energy, (hipStream_t)stream);
    c->lazy_entry = false;
    if (!rc) rc = separate_impl(c, pcm, array_stride, mic_stride, n_arrays, n_frames, doa_rad, out_pcm, (hipStream_t)stream, doa_bin);
    if (rc) return rc;
    c->last_arrays = n_arrays; c->last_frames = n_frames;
    c->e_cur ^= 1; c->tail_cur ^= 1;
    return MCA_HIP_OK;
}

// ---- real-time mode: the stream call as a HIP graph ------------------------------------------
struct mca_hip_graph {
    mca_hip_ctx *c = nullptr;
    const float *pcm = nullptr; long long array_stride = 0, mic_stride = 0;
    int n_arrays = 0, n_frames = 0;
    int *doa_bin = nullptr; float *doa_rad = nullptr, *prob = nullptr, *energy = nullptr, *out_pcm = nullptr;
    hipStream_t cap = nullptr;                 // recording stream
    hipGraph_t graph[4] = {};                  // one per (e_cur, tail_cur): the state buffers a call reads / writes
    hipGraphExec_t exec[4] = {};
    unsigned long long ws_gen = 0;             // c->ws_gen the recordings were made under
};

static void graph_orphan(mca_hip_graph *g);
// a recording bakes in the workspace pointers (d_A, d_C, scan / gate buffers): drop every recording of the graph
static void graph_drop_recordings(mca_hip_graph *g)
{
    for (int i = 0; i < 4; ++i) {
        if (g->exec[i]) (void)hipGraphExecDestroy(g->exec[i]);
        if (g->graph[i]) (void)hipGraphDestroy(g->graph[i]);
        g->exec[i] = nullptr; g->graph[i] = nullptr;
    }
}

static void graph_orphan(mca_hip_graph *g) { graph_drop_recordings(g); g->c = nullptr; }

int mca_hip_graph_create(mca_hip_ctx *c, const float *pcm, long long array_stride, long long mic_stride, int n_arrays,
                         int n_frames, int *doa_bin, float *doa_rad, float *prob, float *energy, float *out_pcm,
                         mca_hip_graph **out)
{
    if (!out) return fail(c, MCA_HIP_ERR_INVALID_ARGUMENT, "out is NULL");
    *out = nullptr;
    int rc = check_stream_args(c, pcm, array_stride, mic_stride, n_arrays, n_frames);
    if (rc) return rc;
    // doa_bin_dev NULL: the separation stage alone, steered by the caller's angles in doa_rad_dev (mca_hip_separate_frames_dev: the
    // delay-and-sum stream of mcabeamf.cpp:77-122, BASELINE configs[1])
    if (!doa_bin && !out_pcm) return fail(c, MCA_HIP_ERR_INVALID_ARGUMENT, "doa_bin_dev and out_pcm_dev are both NULL: nothing to record");
    if (out_pcm && !doa_rad) return fail(c, MCA_HIP_ERR_INVALID_ARGUMENT, "doa_rad_dev is NULL (the separation stage steers with it)");
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    if (doa_bin && (rc = reserve_impl(c, n_arrays, n_frames))) return rc;   // recording must not allocate (a recording runs as one lane)
    if (out_pcm && (rc = ensure_bf_table(c))) return rc;
    mca_hip_graph *g = new mca_hip_graph();
    g->c = c; g->pcm = pcm; g->array_stride = array_stride; g->mic_stride = mic_stride; g->n_arrays = n_arrays; g->n_frames = n_frames;
    g->doa_bin = doa_bin; g->doa_rad = doa_rad; g->prob = prob; g->energy = energy; g->out_pcm = out_pcm;
    if (hipStreamCreateWithFlags(&g->cap, hipStreamNonBlocking) != hipSuccess) { delete g; return fail(c, MCA_HIP_ERR_HIP, "hipStreamCreateWithFlags failed"); }
    g->ws_gen = c->ws_gen;
    c->graphs.push_back(g);
    *out = g;
    return MCA_HIP_OK;
}

static int graph_record(mca_hip_graph *g, int idx)
{
    mca_hip_ctx *c = g->c;
    const int e_cur = c->e_cur, tail_cur = c->tail_cur;
    const unsigned timing = c->timing;
    const unsigned long long adapt_frames = c->adapt_frames_total;
    c->timing = 0;                              // event pairs belong to eager calls
    const bool suspended = c->adapt_suspended;
    c->adapt_suspended = false; c->capturing = true;   // a recording is the mode's own kernels, whatever the eager calls do at the moment
    struct Restore { mca_hip_ctx *c; bool s; ~Restore() { c->adapt_suspended = s; c->capturing = false; } } restore{c, suspended};
    HIP_TRY(c, hipStreamBeginCapture(g->cap, hipStreamCaptureModeRelaxed));
    int rc = MCA_HIP_OK;
    if (g->doa_bin)
        rc = mca_hip_localise_frames_dev(c, g->pcm, g->array_stride, g->mic_stride, g->n_arrays, g->n_frames, g->doa_bin, g->doa_rad,
                                         g->prob, g->energy, g->cap);
    if (!rc && g->out_pcm)
        rc = separate_frames_dev_bins(c, g->pcm, g->array_stride, g->mic_stride, g->n_arrays, g->n_frames, g->doa_rad, g->out_pcm, g->cap, g->doa_bin);
    hipGraph_t graph = nullptr;
    const hipError_t e = hipStreamEndCapture(g->cap, &graph);
    c->timing = timing;
    c->e_cur = e_cur; c->tail_cur = tail_cur;   // nothing ran: the state has not moved
    c->adapt_frames_total = adapt_frames;       // ... and no frame was processed (mca_hip_graph_launch counts the replays)
    if (rc) { if (graph) (void)hipGraphDestroy(graph); return rc; }
    if (e != hipSuccess) return fail(c, MCA_HIP_ERR_HIP, std::string("hipStreamEndCapture: ") + hipGetErrorString(e));
    g->graph[idx] = graph;
    HIP_TRY(c, hipGraphInstantiate(&g->exec[idx], graph, nullptr, nullptr, 0));
    return MCA_HIP_OK;
}

int mca_hip_graph_launch(mca_hip_graph *g, void *stream)
{
    if (!g) return MCA_HIP_ERR_INVALID_ARGUMENT;
    mca_hip_ctx *c = g->c;
    if (!c) { g_create_error = "mca_hip_graph_launch: the context of this graph has been destroyed"; return MCA_HIP_ERR_INVALID_ARGUMENT; }
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    if (g->doa_bin && c->hist_pending) {          // lazy tails: a recording takes the state as exact
        const int rc = settle_history(c, (hipStream_t)stream);
        if (rc) return rc;
    }
    if (g->ws_gen != c->ws_gen) {
        // an eager call with more arrays / frames reallocated the workspace since the recording: the recorded kernels
        // would run on freed memory.  Make sure the workspace (still) fits this graph's shape, then record afresh.
        HIP_TRY(c, hipDeviceSynchronize());      // launches of the old recordings may still be in flight
        graph_drop_recordings(g);
        const int rc = g->doa_bin ? reserve_impl(c, g->n_arrays, g->n_frames) : MCA_HIP_OK;
        if (rc) return rc;
        g->ws_gen = c->ws_gen;
    }
    const int idx = c->e_cur | (c->tail_cur << 1);
    if (!g->exec[idx]) {
        const int rc = graph_record(g, idx);
        if (rc) return rc;
    }
    HIP_TRY(c, hipGraphLaunch(g->exec[idx], (hipStream_t)stream));
    if (g->doa_bin && adaptive_shape(c, g->n_arrays, g->n_frames)) c->adapt_frames_total += (unsigned long long)g->n_arrays * g->n_frames;   // (a recording counts nothing)
    if (g->doa_bin) c->e_cur ^= 1;              // as the eager calls do
    if (g->out_pcm) c->tail_cur ^= 1;
    if (g->doa_bin) {
        c->last_arrays = g->n_arrays; c->last_frames = g->n_frames;
        c->n_lanes_last = 1; c->lanes[0].last_a0 = 0; c->lanes[0].last_arrays = g->n_arrays;
    }
    return MCA_HIP_OK;
}

void mca_hip_graph_destroy(mca_hip_graph *g)
{
    if (!g) return;
    if (g->c) {
        (void)hipSetDevice(g->c->cfg.device);
        (void)hipDeviceSynchronize();
        auto &v = g->c->graphs;
        for (size_t i = 0; i < v.size(); ++i) if (v[i] == g) { v.erase(v.begin() + i); break; }
    }
    graph_drop_recordings(g);
    if (g->cap) (void)hipStreamDestroy(g->cap);
    delete g;
}

// 16-bit PCM: upload the shorts (half the PCIe bytes), widen on the GPU
__global__ void k_i16_to_f32(const short *src, float *dst, long long n4)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n4) {
        const short4 v = reinterpret_cast<const short4 *>(src)[i];
        reinterpret_cast<float4 *>(dst)[i] = make_float4((float)v.x, (float)v.y, (float)v.z, (float)v.w);
    }
}

extern "C++" {
namespace {

// is p page-locked host memory (hipHostMalloc / hipHostRegister, e.g. through mca_hip_host_alloc / _register)?
bool is_pinned(const void *p)
{
    if (!p) return false;
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, p) != hipSuccess) { (void)hipGetLastError(); return false; }
    return at.type == hipMemoryTypeHost;
}

// The host-pointer stream call.  Pageable buffers: one synchronous copy in, the kernels, synchronous copies out (the
// driver stages pageable memory through its own bounce buffers at about half the PCIe rate).  Page-locked input: the
// arrays go up in up to four chunks on a copy stream, each chunk's kernels start when its samples have arrived, and its
// results leave on a third stream while the next chunk computes -- input, kernels and output of a call overlap, and the
// rate is that of the PCIe link.  Same kernels on the same per-array state either way: the outputs are bit-identical.
template <typename SampleT>
int process_frames_host_impl(mca_hip_ctx *c, const SampleT *pcm, int n_arrays, int n_frames, int *doa_bin, float *doa_rad, float *prob,
                             float *energy, float *out_pcm)
{
    constexpr bool I16 = sizeof(SampleT) == 2;
    if (!c || !pcm || !doa_bin) return fail(c, MCA_HIP_ERR_INVALID_ARGUMENT, "NULL argument");
    if (n_arrays < 1 || n_frames < 1) return fail(c, MCA_HIP_ERR_INVALID_ARGUMENT, "n_arrays/n_frames < 1");
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    const long long ms = (long long)(n_frames + 1) * c->H, as = ms * c->M;
    const size_t n_pcm = (size_t)as * n_arrays, fs_ = (size_t)n_frames * c->S, fd_ = (size_t)n_frames * c->D, fo_ = (size_t)c->S * n_frames * c->H;
    const size_t n_fs = n_arrays * fs_, n_en = energy ? n_arrays * fd_ : 0, n_out = out_pcm ? n_arrays * fo_ : 0;
    float *d_pcm = (float *)c->stage.get(0, n_pcm * 4), *d_rad = (float *)c->stage.get(2, n_fs * 4), *d_prob = (float *)c->stage.get(3, n_fs * 4);
    float *d_en = (float *)c->stage.get(4, n_en * 4), *d_out = (float *)c->stage.get(5, n_out * 4);
    int *d_bin = (int *)c->stage.get(1, n_fs * 4);
    short *d_i16 = I16 ? (short *)c->stage.get(6, n_pcm * 2) : nullptr;
    if (!d_pcm || !d_bin || !d_rad || !d_prob || (energy && !d_en) || (out_pcm && !d_out) || (I16 && !d_i16))
        return fail(c, MCA_HIP_ERR_OUT_OF_MEMORY, "device staging buffers for the host-pointer call");
    int rc = check_stream_args(c, d_pcm, as, ms, n_arrays, n_frames);
    if (!rc && out_pcm) rc = ensure_bf_table(c);
    if (rc) return rc;

    if (!is_pinned(pcm)) {
        if (I16) {
            HIP_TRY(c, hipMemcpy(d_i16, pcm, n_pcm * 2, hipMemcpyHostToDevice));
            const long long n4 = (long long)(n_pcm / 4);                    // (F+1)*hop is a multiple of 32
            hipLaunchKernelGGL(k_i16_to_f32, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, nullptr, d_i16, d_pcm, n4);
        } else {
            HIP_TRY(c, hipMemcpy(d_pcm, pcm, n_pcm * 4, hipMemcpyHostToDevice));
        }
        c->host_call = true;
        rc = mca_hip_localise_frames_dev(c, d_pcm, as, ms, n_arrays, n_frames, d_bin, d_rad, d_prob, d_en, nullptr);
        c->host_call = false;
        if (!rc && out_pcm) rc = separate_frames_dev_bins(c, d_pcm, as, ms, n_arrays, n_frames, d_rad, d_out, nullptr, d_bin);
        if (rc) return rc;
        HIP_TRY(c, hipDeviceSynchronize());
        HIP_TRY(c, hipMemcpy(doa_bin, d_bin, n_fs * 4, hipMemcpyDeviceToHost));
        if (doa_rad) HIP_TRY(c, hipMemcpy(doa_rad, d_rad, n_fs * 4, hipMemcpyDeviceToHost));
        if (prob) HIP_TRY(c, hipMemcpy(prob, d_prob, n_fs * 4, hipMemcpyDeviceToHost));
        if (energy) HIP_TRY(c, hipMemcpy(energy, d_en, n_en * 4, hipMemcpyDeviceToHost));
        if (out_pcm) HIP_TRY(c, hipMemcpy(out_pcm, d_out, n_out * 4, hipMemcpyDeviceToHost));
        return MCA_HIP_OK;
    }

    // page-locked input: chunks of arrays through three streams
    for (int i = 0; i < 3; ++i)
        if (!c->io_stream[i]) HIP_TRY(c, hipStreamCreateWithFlags(&c->io_stream[i], hipStreamNonBlocking));
    constexpr int MAXC = 4;
    for (int i = 0; i < 2 * MAXC; ++i)
        if (!c->io_ev[i]) HIP_TRY(c, hipEventCreateWithFlags(&c->io_ev[i], hipEventDisableTiming));
    hipStream_t s_in = c->io_stream[0], s_run = c->io_stream[1], s_out = c->io_stream[2];
    // the three internal streams are non-blocking: nothing orders them after earlier work on the null stream or on a caller's
    // stream (a *_dev call or a graph launch that still writes d_E / d_tail, the memsets of mca_hip_reset).  This entry point
    // is synchronous anyway, so it starts from an idle device.
    HIP_TRY(c, hipDeviceSynchronize());
    // on every exit -- also the early ones of HIP_TRY -- the planning override is dropped and the streams are drained
    struct Guard {
        mca_hip_ctx *c; hipStream_t s[3];
        ~Guard() { c->plan_rows = 0; c->plan_arrays = 0; c->a0 = 0; for (hipStream_t q : s) (void)hipStreamSynchronize(q); }
    } guard{c, {s_in, s_run, s_out}};
    // (with the power gate the call stays one chunk: mca_hip_copy_gate reads the flags of the whole call from one workspace)
    const int nchunk = c->cfg.use_power_floor ? 1 : std::min(n_arrays, MAXC);
    const bool out_pinned[5] = {is_pinned(doa_bin), is_pinned(doa_rad), is_pinned(prob), is_pinned(energy), is_pinned(out_pcm)};
    int a0 = 0;
    for (int k = 0; k < nchunk; ++k) {                       // all uploads are queued first: the link never waits for a kernel
        const int na = n_arrays / nchunk + (k < n_arrays % nchunk ? 1 : 0);
        if (I16) HIP_TRY(c, hipMemcpyAsync(d_i16 + (size_t)a0 * as, pcm + (size_t)a0 * as, (size_t)na * as * 2, hipMemcpyHostToDevice, s_in));
        else HIP_TRY(c, hipMemcpyAsync(d_pcm + (size_t)a0 * as, pcm + (size_t)a0 * as, (size_t)na * as * 4, hipMemcpyHostToDevice, s_in));
        HIP_TRY(c, hipEventRecord(c->io_ev[k], s_in));
        a0 += na;
    }
    a0 = 0;
    adapt_policy_begin(c, n_arrays, n_frames);
    c->plan_rows = (long long)n_arrays * n_frames;           // every chunk is planned as the whole call (bit-identical results)
    c->plan_arrays = n_arrays;
    for (int k = 0; k < nchunk && !rc; ++k) {
        const int na = n_arrays / nchunk + (k < n_arrays % nchunk ? 1 : 0);
        HIP_TRY(c, hipStreamWaitEvent(s_run, c->io_ev[k], 0));
        if (I16) {
            const long long n4 = (long long)na * as / 4;
            hipLaunchKernelGGL(k_i16_to_f32, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s_run, d_i16 + (size_t)a0 * as, d_pcm + (size_t)a0 * as, n4);
        }
        c->cur_lane = 0; c->a0 = a0;                          // the chunk's arrays on lane 0's workspace, per-array state at its offset
        rc = localise_impl(c, d_pcm + (size_t)a0 * as, as, ms, na, n_frames, d_bin + a0 * fs_, d_rad + a0 * fs_, d_prob + a0 * fs_,
                           energy ? d_en + a0 * fd_ : nullptr, s_run);
        if (!rc && out_pcm) rc = separate_impl(c, d_pcm + (size_t)a0 * as, as, ms, na, n_frames, d_rad + a0 * fs_, d_out + a0 * fo_, s_run, d_bin + a0 * fs_);
        c->a0 = 0;
        if (rc) break;
        HIP_TRY(c, hipEventRecord(c->io_ev[MAXC + k], s_run));
        HIP_TRY(c, hipStreamWaitEvent(s_out, c->io_ev[MAXC + k], 0));
        if (out_pinned[0]) HIP_TRY(c, hipMemcpyAsync(doa_bin + a0 * fs_, d_bin + a0 * fs_, na * fs_ * 4, hipMemcpyDeviceToHost, s_out));
        if (doa_rad && out_pinned[1]) HIP_TRY(c, hipMemcpyAsync(doa_rad + a0 * fs_, d_rad + a0 * fs_, na * fs_ * 4, hipMemcpyDeviceToHost, s_out));
        if (prob && out_pinned[2]) HIP_TRY(c, hipMemcpyAsync(prob + a0 * fs_, d_prob + a0 * fs_, na * fs_ * 4, hipMemcpyDeviceToHost, s_out));
        if (energy && out_pinned[3]) HIP_TRY(c, hipMemcpyAsync(energy + a0 * fd_, d_en + a0 * fd_, na * fd_ * 4, hipMemcpyDeviceToHost, s_out));
        if (out_pcm && out_pinned[4]) HIP_TRY(c, hipMemcpyAsync(out_pcm + a0 * fo_, d_out + a0 * fo_, na * fo_ * 4, hipMemcpyDeviceToHost, s_out));
        a0 += na;
    }
    c->plan_rows = 0; c->plan_arrays = 0;
    (void)hipStreamSynchronize(s_in); (void)hipStreamSynchronize(s_run); (void)hipStreamSynchronize(s_out);
    if (rc) return rc;
    c->last_arrays = n_arrays; c->last_frames = n_frames;
    c->n_lanes_last = 1; c->lanes[0].last_a0 = 0; c->lanes[0].last_arrays = n_arrays;
    c->e_cur ^= 1;
    if (out_pcm) c->tail_cur ^= 1;
    // pageable result buffers: plain copies now that everything has finished
    if (!out_pinned[0]) HIP_TRY(c, hipMemcpy(doa_bin, d_bin, n_fs * 4, hipMemcpyDeviceToHost));
    if (doa_rad && !out_pinned[1]) HIP_TRY(c, hipMemcpy(doa_rad, d_rad, n_fs * 4, hipMemcpyDeviceToHost));
    if (prob && !out_pinned[2]) HIP_TRY(c, hipMemcpy(prob, d_prob, n_fs * 4, hipMemcpyDeviceToHost));
    if (energy && !out_pinned[3]) HIP_TRY(c, hipMemcpy(energy, d_en, n_en * 4, hipMemcpyDeviceToHost));
    if (out_pcm && !out_pinned[4]) HIP_TRY(c, hipMemcpy(out_pcm, d_out, n_out * 4, hipMemcpyDeviceToHost));
    return MCA_HIP_OK;
}

}  // namespace
}  // extern "C++"

int mca_hip_process_frames_host(mca_hip_ctx *c, const float *pcm, int n_arrays, int n_frames, int *doa_bin,
                                float *doa_rad, float *prob, float *energy, float *out_pcm)
{
    return process_frames_host_impl<float>(c, pcm, n_arrays, n_frames, doa_bin, doa_rad, prob, energy, out_pcm);
}

int mca_hip_process_frames_host_i16(mca_hip_ctx *c, const short *pcm, int n_arrays, int n_frames, int *doa_bin,
                                    float *doa_rad, float *prob, float *energy, float *out_pcm)
{
    return process_frames_host_impl<short>(c, pcm, n_arrays, n_frames, doa_bin, doa_rad, prob, energy, out_pcm);
}

// page-locked host memory for the host-pointer entry points (thin wrappers, so that a caller need not link HIP)
void *mca_hip_host_alloc(long long bytes)
{
    void *p = nullptr;
    if (bytes <= 0 || hipHostMalloc(&p, (size_t)bytes, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    return p;
}
void mca_hip_host_free(void *p) { if (p) (void)hipHostFree(p); }
int mca_hip_host_register(void *p, long long bytes)
{
    if (!p || bytes <= 0) return MCA_HIP_ERR_INVALID_ARGUMENT;
    if (hipHostRegister(p, (size_t)bytes, hipHostRegisterDefault) != hipSuccess) { (void)hipGetLastError(); g_create_error = "hipHostRegister failed"; return MCA_HIP_ERR_HIP; }
    return MCA_HIP_OK;
}
int mca_hip_host_unregister(void *p)
{
    if (!p) return MCA_HIP_ERR_INVALID_ARGUMENT;
    if (hipHostUnregister(p) != hipSuccess) { (void)hipGetLastError(); g_create_error = "hipHostUnregister failed"; return MCA_HIP_ERR_HIP; }
    return MCA_HIP_OK;
}

int mca_hip_gcc2_frames_dev(mca_hip_ctx *c, const float *pcm, long long array_stride, long long mic_stride,
                            int n_arrays, int n_frames, int *argmax, float *doa_rad, float *prob, float *corr, void *stream)
{
    int rc = check_stream_args(c, pcm, array_stride, mic_stride, n_arrays, n_frames);
    if (rc) return rc;
    if (c->M != 2) return fail(c, MCA_HIP_ERR_INVALID_ARGUMENT, "the 2-microphone GCC path needs a context with n_mics == 2");
    if (c->Dp > 192) return fail(c, MCA_HIP_ERR_UNSUPPORTED, "the 2-microphone GCC path supports up to 192 steering delays");
    if (!argmax) return fail(c, MCA_HIP_ERR_INVALID_ARGUMENT, "argmax_dev is NULL");
    hipStream_t st = (hipStream_t)stream;
    if (c->prec == MCA_HIP_SRP_ADAPTIVE) set_call_planes(c, 2);      // the 2-microphone path always runs the exact split
    if ((rc = run_correlation_map(c, pcm, array_stride, mic_stride, n_arrays, n_frames, st))) return rc;
    Gcc2ScanArgs ga{};
    ga.C = c->ws().d_C; ga.c_planes = c->ws().c_planes; ga.c_plane_stride = c->ws().c_plane; ga.n_frames = n_frames; ga.Dp = c->Dp; ga.D = c->D;
    // frames per chunk: every chunk re-reads 160 warm-up frames (recursion + DOA smoothing), so long chunks are cheaper;
    // shorter ones for small batches so that a few hundred workgroups exist
    ga.chunk = 128;
    while (ga.chunk > 32 && (long long)n_arrays * ((n_frames + ga.chunk - 1) / ga.chunk) < 512) ga.chunk >>= 1;
    ga.vdone_in = c->d_vdone[c->doa_cur]; ga.vdone_out = c->d_vdone[c->doa_cur ^ 1];
    ga.mu = 0.8f; ga.one_minus_mu = 1 - 0.8f;                      // _maxCorrMemoryFactor (BinauralLocalisation.h:198)
    ga.doa_mem = 0.6f; ga.one_minus_doa_mem = 1 - 0.6f;            // _maxDoaMemoryFactor (:199)
    ga.step = c->step;
    ga.corr_in = c->d_E[c->e_cur]; ga.corr_out = c->d_E[c->e_cur ^ 1];
    ga.doa_in = c->d_doa[c->doa_cur]; ga.doa_out = c->d_doa[c->doa_cur ^ 1];
    ga.grid = c->d_grid; ga.argmax = argmax; ga.doa_rad = doa_rad; ga.prob = prob; ga.corr = corr;
    const bool gate = c->cfg.use_power_floor != 0;
    time_begin(c, MCA_HIP_K_GCC2_SCAN, st);
    if (gate) {
        // setPowerFloor + the gate of processParametrisation (BinauralLocalisation.cpp:387-404, :425-434): 3 s of floor
        // estimation (power + 1e-10 per frame), then voiced = FFTLogPower > floor + 6 dB; the recursions below only see
        // the frames that fired, gated-out frames repeat the previous outputs
        const size_t rows = (size_t)n_arrays * n_frames;
        if (rows > c->g2_rows) {
            auto Fr = [](void *q) { if (q) (void)hipFree(q); };
            Fr(c->d_g2_vidx); Fr(c->d_g2_nv); Fr(c->d_g2_rad); Fr(c->d_g2_prob); Fr(c->d_g2_reset);
            c->d_g2_vidx = c->d_g2_nv = nullptr; c->d_g2_rad = c->d_g2_prob = nullptr; c->d_g2_reset = nullptr; c->g2_rows = 0;
            HIP_TRY(c, hipMalloc((void **)&c->d_g2_vidx, rows * 4));
            HIP_TRY(c, hipMalloc((void **)&c->d_g2_nv, (size_t)c->cfg.max_arrays * 4));
            HIP_TRY(c, hipMalloc((void **)&c->d_g2_rad, rows * 4));
            HIP_TRY(c, hipMalloc((void **)&c->d_g2_prob, rows * 4));
            HIP_TRY(c, hipMalloc((void **)&c->d_g2_reset, rows));
            c->g2_rows = rows;
        }
        GateArgs gg{};
        gg.power_lin = c->ws().d_power; gg.n_frames = n_frames; gg.fft_n = c->N;
        gg.needed_samples = (int)(3.0 * c->cfg.sample_rate);              // _durationToEstimatePowerFloor (SoundLocalisationImpl.h:77)
        gg.margin_db = 6.f;                                                // _noiseMarginDB (BinauralLocalisation.h:197)
        gg.eps = 1e-10;                                                    // BinauralLocalisation.cpp:391
        gg.state = c->d_gate_state; gg.voiced = c->ws().d_voiced; gg.power_out = c->ws().d_power_out; gg.post0 = c->d_g2_post0;
        hipLaunchKernelGGL(k_gate, dim3(n_arrays), dim3(256), 0, st, gg);
        // the silence rule (:530-560): windowsToDecay = 3 * fs / (analysisLength / 2 - 1), int arithmetic, analysisLength = N + 2
        const int windows_to_decay = 3 * c->cfg.sample_rate / c->H;
        hipLaunchKernelGGL(k_gcc2_compact, dim3(n_arrays), dim3(256), 0, st, c->ws().d_voiced, n_frames, c->d_g2_vidx, c->d_g2_nv,
                           c->d_g2_post0, c->d_silence, windows_to_decay, c->d_g2_reset);
        ga.vidx = c->d_g2_vidx; ga.nv = c->d_g2_nv; ga.vreset = c->d_g2_reset;
        if (!ga.doa_rad) ga.doa_rad = c->d_g2_rad;                         // the hold-over needs them whatever the caller asked for
        if (!ga.prob) ga.prob = c->d_g2_prob;
    }
    const int nslot = GCC2_DOAWARM + ga.chunk;
    const size_t smem = (size_t)nslot * ((c->D + 3) / 4 * 4 + 4) * sizeof(float) + ((size_t)nslot * 5 + 1) * sizeof(float);
    if (smem > 64 * 1024)
        HIP_TRY(c, hipFuncSetAttribute(reinterpret_cast<const void *>(k_gcc2_scan), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    dim3 g((n_frames + ga.chunk - 1) / ga.chunk, n_arrays);
    hipLaunchKernelGGL(k_gcc2_scan, g, dim3(std::max(256, round_up(c->D, 64))), smem, st, ga);   // >= 4 waves: the per-frame argmax / min / sum is one wave per frame
    if (gate) {
        Gcc2FillArgs fa{};
        fa.voiced = c->ws().d_voiced; fa.n_frames = n_frames; fa.D = c->D;
        fa.argmax = argmax; fa.doa_rad = ga.doa_rad; fa.prob = ga.prob; fa.corr = corr; fa.corr_state = ga.corr_in;
        fa.last_idx = c->d_last_bin; fa.last_rad = c->d_last_rad; fa.last_prob = c->d_last_prob;
        hipLaunchKernelGGL(k_gcc2_fill, dim3(n_arrays), dim3(1024), 0, st, fa);
    }
    time_end(c, st);
    HIP_TRY(c, hipGetLastError());
    c->e_cur ^= 1; c->doa_cur ^= 1;
    c->last_arrays = n_arrays; c->last_frames = n_frames;
    c->n_lanes_last = 1; c->lanes[0].last_a0 = 0; c->lanes[0].last_arrays = n_arrays;
    return MCA_HIP_OK;
}

int mca_hip_gcc2_frames_host(mca_hip_ctx *c, const float *pcm, int n_arrays, int n_frames, int *argmax,
                             float *doa_rad, float *prob, float *corr)
{
    if (!c || !pcm || !argmax) return fail(c, MCA_HIP_ERR_INVALID_ARGUMENT, "NULL argument");
    if (n_arrays < 1 || n_frames < 1) return fail(c, MCA_HIP_ERR_INVALID_ARGUMENT, "n_arrays/n_frames < 1");
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    const long long ms = (long long)(n_frames + 1) * c->H, as = ms * c->M;
    const size_t n_pcm = (size_t)as * n_arrays, n_f = (size_t)n_arrays * n_frames, n_corr = corr ? n_f * c->D : 0;
    float *d_pcm = (float *)c->stage.get(0, n_pcm * 4), *d_rad = (float *)c->stage.get(2, n_f * 4), *d_prob = (float *)c->stage.get(3, n_f * 4);
    float *d_corr = (float *)c->stage.get(4, n_corr * 4);
    int *d_idx = (int *)c->stage.get(1, n_f * 4);
    if (!d_pcm || !d_idx || !d_rad || !d_prob || (corr && !d_corr))
        return fail(c, MCA_HIP_ERR_OUT_OF_MEMORY, "device staging buffers for the host-pointer call");
    HIP_TRY(c, hipMemcpy(d_pcm, pcm, n_pcm * 4, hipMemcpyHostToDevice));
    const int rc = mca_hip_gcc2_frames_dev(c, d_pcm, as, ms, n_arrays, n_frames, d_idx, d_rad, d_prob, d_corr, nullptr);
    if (rc) return rc;
    HIP_TRY(c, hipDeviceSynchronize());
    HIP_TRY(c, hipMemcpy(argmax, d_idx, n_f * 4, hipMemcpyDeviceToHost));
    if (doa_rad) HIP_TRY(c, hipMemcpy(doa_rad, d_rad, n_f * 4, hipMemcpyDeviceToHost));
    if (prob) HIP_TRY(c, hipMemcpy(prob, d_prob, n_f * 4, hipMemcpyDeviceToHost));
    if (corr) HIP_TRY(c, hipMemcpy(corr, d_corr, n_corr * 4, hipMemcpyDeviceToHost));
    return MCA_HIP_OK;
}

int mca_hip_copy_gate(mca_hip_ctx *c, unsigned char *voiced, float *power)
{
    if (!c) return MCA_HIP_ERR_INVALID_ARGUMENT;
    if (!c->cfg.use_power_floor) return fail(c, MCA_HIP_ERR_INVALID_ARGUMENT, "the context was created with use_power_floor = 0");
    if (c->last_frames == 0) return fail(c, MCA_HIP_ERR_INVALID_ARGUMENT, "no stream call has run yet");
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    HIP_TRY(c, hipDeviceSynchronize());
    for (int i = 0; i < c->n_lanes_last; ++i) {          // every lane holds the flags of its own block of arrays
        const Workspace &w = c->lanes[i];
        const size_t n = (size_t)w.last_arrays * c->last_frames, off = (size_t)w.last_a0 * c->last_frames;
        if (voiced) HIP_TRY(c, hipMemcpy(voiced + off, w.d_voiced, n, hipMemcpyDeviceToHost));
        if (power) HIP_TRY(c, hipMemcpy(power + off, w.d_power_out, n * 4, hipMemcpyDeviceToHost));
    }
    return MCA_HIP_OK;
}

// ---- frame API ---------------------------------------------------------------------------
static int upload_frames(mca_hip_ctx *c, const double *const *frames, int ccs_len)
{
    if (!frames) return fail(c, MCA_HIP_ERR_INVALID_ARGUMENT, "frames is NULL");
    if (ccs_len != c->N + 2) return fail(c, MCA_HIP_ERR_INVALID_ARGUMENT, "ccs_len != fft_size + 2");
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    const size_t n = (size_t)c->M * ccs_len;
    if (c->fr_elems < n) {
        if (c->d_fr) (void)hipFree(c->d_fr);
        c->d_fr = nullptr; c->fr_elems = 0;
        HIP_TRY(c, hipMalloc((void **)&c->d_fr, n * 8));
        c->fr_elems = n;
    }
    // SignalVector = one separate allocation per channel (mcadefs.h:86-88): gather, then one copy
    c->h_stage.resize(n);
    for (int m = 0; m < c->M; ++m) {
        if (!frames[m]) return fail(c, MCA_HIP_ERR_INVALID_ARGUMENT, "frames[c] is NULL");
        std::memcpy(c->h_stage.data() + (size_t)m * ccs_len, frames[m], (size_t)ccs_len * 8);
    }
    HIP_TRY(c, hipMemcpy(c->d_fr, c->h_stage.data(), n * 8, hipMemcpyHostToDevice));
    return MCA_HIP_OK;
}

int mca_hip_steering_process_frame(mca_hip_ctx *c, const double *const *frames, int ccs_len, double *DOA,
                                   double *prob, int *doa_bin, int n_sources)
{
    if (!c) return MCA_HIP_ERR_INVALID_ARGUMENT;
    if (n_sources < 1 || n_sources > MCA_MAX_SOURCES) return fail(c, MCA_HIP_ERR_INVALID_ARGUMENT, "n_sources must be in [1,4]");
    if (!DOA || !prob) return fail(c, MCA_HIP_ERR_INVALID_ARGUMENT, "DOA/prob is NULL");
    int rc = upload_frames(c, frames, ccs_len);
    if (rc) return rc;
    const float memf = 0.8f;                                                  // SteeringBeamforming.h:70
    const double mu = (double)memf, omu = (double)(1 - memf);
    double *Ein = c->d_E64[c->e64_cur], *Eout = c->d_E64[c->e64_cur ^ 1];
    hipLaunchKernelGGL(k_frame_srp<double>, dim3(c->D), dim3(256), 0, 0, reinterpret_cast<const C2<double> *>(c->d_fr), c->K, c->D,
                       c->P, c->d_pairs, c->d_delays, Ein, Eout, mu, omu, c->cfg.gcc_weighting == MCA_HIP_GCC_NONE ? 1 : 0);
    hipLaunchKernelGGL(k_frame_pick<double>, dim3(1), dim3(512), 0, 0, Eout, c->D, c->P, n_sources, c->d_grid, c->d_res,
                       c->d_res + MCA_MAX_SOURCES, c->d_bins);
    HIP_TRY(c, hipGetLastError());
    c->e64_cur ^= 1;
    double res[2 * MCA_MAX_SOURCES]; int bins[MCA_MAX_SOURCES];
    HIP_TRY(c, hipMemcpy(res, c->d_res, sizeof(res), hipMemcpyDeviceToHost));
    HIP_TRY(c, hipMemcpy(bins, c->d_bins, sizeof(bins), hipMemcpyDeviceToHost));
    for (int s = 0; s < n_sources; ++s) {
        DOA[s] = res[s]; prob[s] = res[MCA_MAX_SOURCES + s];
        if (doa_bin) doa_bin[s] = bins[s];
    }
    return MCA_HIP_OK;
}

int mca_hip_beamformer_process_frame(mca_hip_ctx *c, const double *const *frames, int ccs_len, double *out, double DOA)
{
    if (!c) return MCA_HIP_ERR_INVALID_ARGUMENT;
    if (!out) return fail(c, MCA_HIP_ERR_INVALID_ARGUMENT, "out is NULL");
    int rc = upload_frames(c, frames, ccs_len);
    if (rc) return rc;
    if (c->out64_elems < (size_t)ccs_len) {
        if (c->d_out64) (void)hipFree(c->d_out64);
        c->d_out64 = nullptr; c->out64_elems = 0;
        HIP_TRY(c, hipMalloc((void **)&c->d_out64, (size_t)ccs_len * 8));
        c->out64_elems = ccs_len;
    }
    hipLaunchKernelGGL(k_frame_beamform<double>, dim3((c->K + 255) / 256), dim3(256), 0, 0,
                       reinterpret_cast<const C2<double> *>(c->d_fr), c->M, c->K, c->cfg.sample_rate, c->d_micx, DOA,
                       reinterpret_cast<C2<double> *>(c->d_out64));
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipMemcpy(out, c->d_out64, (size_t)ccs_len * 8, hipMemcpyDeviceToHost));
    return MCA_HIP_OK;
}

int mca_hip_fft_log_power(mca_hip_ctx *c, const double *const *frames, int ccs_len, double *power_db)
{
    if (!c) return MCA_HIP_ERR_INVALID_ARGUMENT;
    if (!power_db) return fail(c, MCA_HIP_ERR_INVALID_ARGUMENT, "power_db is NULL");
    int rc = upload_frames(c, frames, ccs_len);
    if (rc) return rc;
    double *dp = c->d_res + 2 * MCA_MAX_SOURCES;
    hipLaunchKernelGGL(k_frame_power<double>, dim3(1), dim3(256), 0, 0, reinterpret_cast<const C2<double> *>(c->d_fr), c->M, c->K, dp);
    HIP_TRY(c, hipGetLastError());
    double lin = 0;
    HIP_TRY(c, hipMemcpy(&lin, dp, 8, hipMemcpyDeviceToHost));
    *power_db = 10.0 * std::log10(lin);
    return MCA_HIP_OK;
}

int mca_hip_get_energy(mca_hip_ctx *c, double *out)
{
    if (!c || !out) return MCA_HIP_ERR_INVALID_ARGUMENT;
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    HIP_TRY(c, hipMemcpy(out, c->d_E64[c->e64_cur], (size_t)c->D * 8, hipMemcpyDeviceToHost));
    return MCA_HIP_OK;
}

// ---- timing ------------------------------------------------------------------------------
int mca_hip_set_timing(mca_hip_ctx *c, int enable)
{
    if (!c) return MCA_HIP_ERR_INVALID_ARGUMENT;
    c->timing = enable != 0 ? ~0u : 0u;
    return MCA_HIP_OK;
}

int mca_hip_set_timing_mask(mca_hip_ctx *c, unsigned kernel_mask)
{
    if (!c) return MCA_HIP_ERR_INVALID_ARGUMENT;
    c->timing = kernel_mask;
    return MCA_HIP_OK;
}

static int drain_events(mca_hip_ctx *c)
{
    for (auto &e : c->events) {
        HIP_TRY(c, hipEventSynchronize(e.b));
        float ms = 0.f;
        HIP_TRY(c, hipEventElapsedTime(&ms, e.a, e.b));
        c->t_ms[e.id] += ms; c->t_launches[e.id] += 1;
        c->pool.push_back(e.a); c->pool.push_back(e.b);
    }
    c->events.clear();
    return MCA_HIP_OK;
}

int mca_hip_get_timing(mca_hip_ctx *c, int kernel_id, int *launches, double *total_ms)
{
    if (!c || kernel_id < 0 || kernel_id >= MCA_HIP_K_COUNT) return MCA_HIP_ERR_INVALID_ARGUMENT;
    int rc = drain_events(c);
    if (rc) return rc;
    if (launches) *launches = c->t_launches[kernel_id];
    if (total_ms) *total_ms = c->t_ms[kernel_id];
    return MCA_HIP_OK;
}

int mca_hip_reset_timing(mca_hip_ctx *c)
{
    if (!c) return MCA_HIP_ERR_INVALID_ARGUMENT;
    int rc = drain_events(c);
    for (int i = 0; i < MCA_HIP_K_COUNT; ++i) { c->t_ms[i] = 0; c->t_launches[i] = 0; }
    if (c->d_rstats) { HIP_TRY(c, hipDeviceSynchronize()); HIP_TRY(c, hipMemset(c->d_rstats, 0, 32)); }
    c->adapt_frames_total = 0;
    c->fb_groups_prev = 0; c->fb_frames_prev = 0; c->fb_flagged_prev = 0; c->fb_seq_seen = c->fb_calls;   // (the reports in flight belong to the old totals)
    // ... including a probe's, if the back-off is waiting for one: that report will never count as fresh, so the wait ends here
    // and the mode resumes (the context stayed on FP16X3 until mca_hip_reset)
    if (c->fb_state == 2) { c->fb_state = 0; c->adapt_suspended = false; }
    return rc;
}

}  // extern "C"
