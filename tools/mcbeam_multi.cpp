// mcbeam_multi.cpp -- many independent microphone arrays on the GPUs of one node: one PROCESS per GPU, the hot path through the
// C ABI's device-pointer calls, RCCL over xGMI used only to gather the DOA / output buffers (BASELINE configs[4]: 1024
// independent 8-mic arrays sharded across 8 x MI355X).  The multi-device counterpart of the reference's single-stream driver loop
// (src/programs/mcabeamf.cpp:77-122: read a block, process(), write the output).
//
//   parent   (no MCBEAM_RANK in the environment) never touches a GPU: it starts --ranks N fresh child processes of this very
//            binary (posix_spawn; RANK r -> device r) and waits; the worst exit code is its own
//   rank r   owns the contiguous block mca::localArrays(arrays, r, N) (the blocks of bench.py / mcarray_amd/dist.py), keeps its
//            PCM, state and outputs in its own HBM and runs every step as ONE asynchronous mca_hip_process_frames_dev on its
//            compute stream -- no collective inside the compute.  The call writes DOA bins (int32) and probabilities (fp32)
//            straight into the two halves of a packed [2][n_max][F] buffer; per step ONE ncclAllGather of that buffer (8 bytes
//            per frame) on a second stream, double buffered so that step i + 1 computes while the gather of step i is in flight;
//            with --audio the beamformed audio (2 KB per frame) goes to rank 0 by grouped ncclSend / ncclRecv (a gather over rank
//            0's xGMI links).  Blocks are padded to the largest one (n_max) so that every rank sends the same count.
//   rank 0   unpacks the gathered buffers into global array order, checks every array's DOA against its source and, with
//            --check, runs the UNSHARDED call (all arrays, one context, one device) and requires the gathered bins to equal it.
// The rendezvous is a 128-byte ncclUniqueId that rank 0 leaves in a file the parent names (no network, no MPI).
//
//   make -C tests/cxx mcbeam_multi          (hipcc, -lrccl)
//   ./mcbeam_multi --ranks 8 --arrays 1024 --frames 256 --steps 4 [--audio] [--check] [--precision adaptive|fp16x3|fp32]
// Input: synthetic far-field white sources (one per array, angle and seed from the array's GLOBAL index), like bench.py.
// N > 1 has not run on hardware: every box this build has seen has one GPU.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <spawn.h>
#include <sys/wait.h>
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "mcarray/Partition.h"
#include "mcarray_hip.h"

extern char **environ;

namespace {

struct Options { int ranks = 1, arrays = 8, frames = 256, steps = 2, audio = 0, check = 0, precision = MCA_HIP_SRP_ADAPTIVE; double step_deg = 0.5; };

#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "rank %d: %s: %s\n", g_rank, #x, hipGetErrorString(e_)); return 1; } } while (0)
#define NCCL_OK(x) do { ncclResult_t e_ = (x); if (e_ != ncclSuccess) { std::fprintf(stderr, "rank %d: %s: %s\n", g_rank, #x, ncclGetErrorString(e_)); return 1; } } while (0)
#define MCA_OK(ctx, x) do { int e_ = (x); if (e_ != MCA_HIP_OK) { std::fprintf(stderr, "rank %d: %s: %s\n", g_rank, #x, mca_hip_last_error(ctx)); return 1; } } while (0)
int g_rank = 0;

// splitmix64 -> uniform -> Box-Muller: the input of array g depends on g only, not on the number of ranks
struct Rng {
    unsigned long long s;
    unsigned long long next() { unsigned long long z = (s += 0x9E3779B97F4A7C15ull); z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); }
    double uni() { return ((next() >> 11) + 0.5) * (1.0 / 9007199254740992.0); }
    double gauss() { return std::sqrt(-2.0 * std::log(uni())) * std::cos(6.283185307179586 * uni()); }
};

double theta_of(int g) { return (-80.0 + 160.0 * ((g * 37) % 101) / 100.0) * 3.141592653589793 / 180.0; }

// one array: M channels of n samples, a white source at `theta` (integer-sample delays are enough for a driver example)
void synth_array(float *pcm, int M, long long n, long long pitch, double spacing, double theta, int fs, unsigned long long seed)
{
    Rng r{seed};
    const int guard = 64;
    std::vector<float> src((size_t)n + 2 * guard);
    for (float &v : src) v = (float)(0.1 * r.gauss());
    for (int m = 0; m < M; ++m) {
        const int lag = (int)std::lround(m * spacing * std::sin(theta) / 346.1 * fs);
        for (long long t = 0; t < n; ++t) pcm[m * pitch + t] = src[(size_t)(t + guard + lag)] + (float)(0.01 * r.gauss());
    }
}

constexpr int M = 8, N = 1024, H = N / 2, FS = 48000;

int make_context(const Options &o, int device, int max_arrays, mca_hip_ctx **ctx)
{
    static double xyz[M * 3];
    for (int m = 0; m < M; ++m) { xyz[3 * m] = 0.04 * m; xyz[3 * m + 1] = xyz[3 * m + 2] = 0.0; }
    mca_hip_config cfg = mca_hip_config();
    cfg.struct_size = (int)sizeof(cfg); cfg.device = device; cfg.sample_rate = FS; cfg.fft_size = N; cfg.n_mics = M; cfg.mic_xyz = xyz;
    cfg.doa_step_deg = o.step_deg; cfg.n_sources = 1; cfg.use_power_floor = 0; cfg.srp_precision = o.precision; cfg.max_arrays = max_arrays;
    if (mca_hip_create(&cfg, ctx) != MCA_HIP_OK) { std::fprintf(stderr, "rank %d: %s\n", g_rank, mca_hip_last_error(nullptr)); return 1; }
    return 0;
}

// the PCM of the arrays [first, first + count) in page-locked host memory -> device
int upload_arrays(int first, int count, int F, float **d_pcm)
{
    const long long pitch = (long long)(F + 1) * H, per_array = pitch * M;
    float *h = nullptr;
    HIP_OK(hipHostMalloc((void **)&h, sizeof(float) * per_array * count, hipHostMallocDefault));
    for (int a = 0; a < count; ++a)
        synth_array(h + (size_t)a * per_array, M, pitch, pitch, 0.04, theta_of(first + a), FS, 0x5EED0000ull + (unsigned long long)(first + a));
    HIP_OK(hipMalloc((void **)d_pcm, sizeof(float) * per_array * count));
    HIP_OK(hipMemcpy(*d_pcm, h, sizeof(float) * per_array * count, hipMemcpyHostToDevice));
    HIP_OK(hipHostFree(h));
    return 0;
}

int run_rank(const Options &o, int rank, int world, const char *id_file)
{
    g_rank = rank;
    int n_dev = 0;
    HIP_OK(hipGetDeviceCount(&n_dev));
    if (n_dev < world) { std::fprintf(stderr, "rank %d: %d rank(s) want a GPU each, %d visible\n", rank, world, n_dev); return 1; }
    HIP_OK(hipSetDevice(rank));

    // rendezvous: rank 0's unique id, through the file the parent named (written under another name and renamed: never read half-written)
    ncclUniqueId id;
    if (rank == 0) {
        NCCL_OK(ncclGetUniqueId(&id));
        const std::string tmp = std::string(id_file) + ".tmp";
        FILE *f = std::fopen(tmp.c_str(), "wb");
        if (!f || std::fwrite(&id, sizeof(id), 1, f) != 1) { std::fprintf(stderr, "rank 0: cannot write %s\n", tmp.c_str()); return 1; }
        std::fclose(f);
        if (std::rename(tmp.c_str(), id_file) != 0) { std::perror("rename"); return 1; }
    } else {
        bool got = false;
        for (int i = 0; i < 1200 && !got; ++i) {               // 60 s
            FILE *f = std::fopen(id_file, "rb");
            if (f) { got = std::fread(&id, sizeof(id), 1, f) == 1; std::fclose(f); }
            if (!got) std::this_thread::sleep_for(std::chrono::milliseconds(50));
        }
        if (!got) { std::fprintf(stderr, "rank %d: no unique id from rank 0 within 60 s\n", rank); return 1; }
    }
    ncclComm_t comm;
    NCCL_OK(ncclCommInitRank(&comm, world, id, rank));

    const mca::ArrayBlock blk = mca::localArrays(o.arrays, rank, world);
    const int F = o.frames, n_max = (o.arrays + world - 1) / world;       // the largest block: every rank sends that many
    mca_hip_ctx *ctx = nullptr;
    if (make_context(o, rank, std::max(blk.count, 1), &ctx)) return 1;
    float *d_pcm = nullptr;
    if (blk.count && upload_arrays(blk.first, blk.count, F, &d_pcm)) return 1;
    const long long pitch = (long long)(F + 1) * H, per_array = pitch * M;
    const size_t half = (size_t)n_max * F, pack_words = 2 * half, audio_words = (size_t)n_max * F * H;
    int *d_pack[2] = {nullptr, nullptr}, *d_all[2] = {nullptr, nullptr};
    float *d_rad = nullptr, *d_audio[2] = {nullptr, nullptr}, *d_audio_all[2] = {nullptr, nullptr};
    for (int b = 0; b < 2; ++b) {
        HIP_OK(hipMalloc((void **)&d_pack[b], pack_words * 4));
        HIP_OK(hipMemset(d_pack[b], 0, pack_words * 4));                  // (the padding rows of a short block)
        HIP_OK(hipMalloc((void **)&d_all[b], pack_words * 4 * world));
        if (o.audio) {
            HIP_OK(hipMalloc((void **)&d_audio[b], audio_words * 4));
            HIP_OK(hipMemset(d_audio[b], 0, audio_words * 4));
            if (rank == 0) HIP_OK(hipMalloc((void **)&d_audio_all[b], audio_words * 4 * world));
        }
    }
    HIP_OK(hipMalloc((void **)&d_rad, std::max<size_t>(half, 1) * 4));
    hipStream_t s_run, s_comm;
    HIP_OK(hipStreamCreateWithFlags(&s_run, hipStreamNonBlocking));
    HIP_OK(hipStreamCreateWithFlags(&s_comm, hipStreamNonBlocking));
    hipEvent_t computed[2], gathered[2];
    for (int b = 0; b < 2; ++b) { HIP_OK(hipEventCreateWithFlags(&computed[b], hipEventDisableTiming)); HIP_OK(hipEventCreateWithFlags(&gathered[b], hipEventDisableTiming)); }
    HIP_OK(hipDeviceSynchronize());

    // a first all-gather of the (zero) buffers: the ranks meet, the communicator's channels are set up before the clock starts
    NCCL_OK(ncclAllGather(d_pack[0], d_all[0], pack_words, ncclInt32, comm, s_comm));
    HIP_OK(hipStreamSynchronize(s_comm));
    const auto t0 = std::chrono::steady_clock::now();
    for (int s = 0; s < o.steps; ++s) {
        const int b = s & 1;
        if (s >= 2) HIP_OK(hipStreamWaitEvent(s_run, gathered[b], 0));     // the gather of step s - 2 has read this buffer pair
        // (a live application would refill d_pcm with the next block here, as mcabeamf.cpp:101-112 does per 1024 samples)
        if (blk.count)
            MCA_OK(ctx, mca_hip_process_frames_dev(ctx, d_pcm, per_array, pitch, blk.count, F, d_pack[b], d_rad, reinterpret_cast<float *>(d_pack[b]) + half,
                                                   nullptr, o.audio ? d_audio[b] : nullptr, s_run));
        HIP_OK(hipEventRecord(computed[b], s_run));
        HIP_OK(hipStreamWaitEvent(s_comm, computed[b], 0));
        NCCL_OK(ncclAllGather(d_pack[b], d_all[b], pack_words, ncclInt32, comm, s_comm));
        if (o.audio) {
            NCCL_OK(ncclGroupStart());
            NCCL_OK(ncclSend(d_audio[b], audio_words, ncclFloat, 0, comm, s_comm));
            if (rank == 0)
                for (int r = 0; r < world; ++r) NCCL_OK(ncclRecv(d_audio_all[b] + (size_t)r * audio_words, audio_words, ncclFloat, r, comm, s_comm));
            NCCL_OK(ncclGroupEnd());
        }
        HIP_OK(hipEventRecord(gathered[b], s_comm));
    }
    HIP_OK(hipStreamSynchronize(s_run));
    HIP_OK(hipStreamSynchronize(s_comm));
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();

    int rc = 0;
    if (rank == 0) {
        // unpack: [world][2][n_max][F] of the last step -> global array order
        const int bl = (o.steps - 1) & 1;
        std::vector<int> all(pack_words * world);
        HIP_OK(hipMemcpy(all.data(), d_all[bl], all.size() * 4, hipMemcpyDeviceToHost));
        std::vector<int> bins((size_t)o.arrays * F);
        std::vector<float> prob((size_t)o.arrays * F);
        for (int r = 0; r < world; ++r) {
            const mca::ArrayBlock br = mca::localArrays(o.arrays, r, world);
            const int *src = all.data() + (size_t)r * pack_words;
            std::memcpy(bins.data() + (size_t)br.first * F, src, (size_t)br.count * F * 4);
            std::memcpy(prob.data() + (size_t)br.first * F, src + half, (size_t)br.count * F * 4);
        }
        int bad = 0;
        for (int g = 0; g < o.arrays; ++g) {
            const int bn = bins[(size_t)g * F + F - 1];
            const double deg = bn * o.step_deg - 90.0, want = theta_of(g) * 180.0 / 3.141592653589793;
            if (bn < 0 || std::fabs(deg - want) > 3.0) ++bad;
        }
        double audio_rms = -1.0;
        if (o.audio) {
            std::vector<float> au(audio_words * world);
            HIP_OK(hipMemcpy(au.data(), d_audio_all[bl], au.size() * 4, hipMemcpyDeviceToHost));
            double acc = 0; size_t cnt = 0;
            for (int r = 0; r < world; ++r) {
                const mca::ArrayBlock br = mca::localArrays(o.arrays, r, world);
                for (size_t i = 0; i < (size_t)br.count * F * H; ++i) { const double v = au[(size_t)r * audio_words + i]; acc += v * v; ++cnt; }
            }
            audio_rms = cnt ? std::sqrt(acc / cnt) : 0.0;
            if (!(audio_rms > 1e-3)) { std::fprintf(stderr, "rank 0: the gathered audio is silent (rms %.3g)\n", audio_rms); rc = 1; }
        }
        long long mismatched = -1;
        if (o.check) {
            // the unsharded call: all arrays, one context on this device, the same number of steps from the same initial state
            mca_hip_ctx *whole = nullptr;
            if (make_context(o, 0, o.arrays, &whole)) return 1;
            float *d_all_pcm = nullptr;
            if (upload_arrays(0, o.arrays, F, &d_all_pcm)) return 1;
            int *d_b = nullptr; float *d_r = nullptr, *d_p = nullptr;
            HIP_OK(hipMalloc((void **)&d_b, (size_t)o.arrays * F * 4)); HIP_OK(hipMalloc((void **)&d_r, (size_t)o.arrays * F * 4)); HIP_OK(hipMalloc((void **)&d_p, (size_t)o.arrays * F * 4));
            for (int s = 0; s < o.steps; ++s)
                MCA_OK(whole, mca_hip_localise_frames_dev(whole, d_all_pcm, per_array, pitch, o.arrays, F, d_b, d_r, d_p, nullptr, nullptr));
            HIP_OK(hipDeviceSynchronize());
            std::vector<int> wb((size_t)o.arrays * F);
            std::vector<float> wp((size_t)o.arrays * F);
            HIP_OK(hipMemcpy(wb.data(), d_b, wb.size() * 4, hipMemcpyDeviceToHost));
            HIP_OK(hipMemcpy(wp.data(), d_p, wp.size() * 4, hipMemcpyDeviceToHost));
            mismatched = 0;
            for (size_t i = 0; i < wb.size(); ++i) mismatched += (wb[i] != bins[i]) || std::memcmp(&wp[i], &prob[i], 4) != 0;
            if (mismatched) rc = 1;
            (void)hipFree(d_b); (void)hipFree(d_r); (void)hipFree(d_p); (void)hipFree(d_all_pcm);
            mca_hip_destroy(whole);
        }
        const double nf = (double)o.arrays * F;
        std::printf("%d arrays x %d frames on %d rank(s) (one process per GPU, RCCL all-gather of %zu bytes per rank and step%s), %d steps: %.1f ms wall, "
                    "%.2f M frames/s; %d arrays off their source", o.arrays, F, world, pack_words * 4, o.audio ? ", audio to rank 0" : "", o.steps, ms,
                    nf * o.steps / ms / 1e3, bad);
        if (o.audio) std::printf("; gathered audio rms %.4f", audio_rms);
        if (o.check) std::printf("; %lld of %zu gathered (bin, prob) pairs differ from the unsharded call", mismatched, bins.size());
        std::printf("\n");
        if (bad) rc = 1;
    }
    NCCL_OK(ncclCommDestroy(comm));
    for (int b = 0; b < 2; ++b) { (void)hipFree(d_pack[b]); (void)hipFree(d_all[b]); (void)hipFree(d_audio[b]); (void)hipFree(d_audio_all[b]); }
    (void)hipFree(d_rad); (void)hipFree(d_pcm);
    mca_hip_destroy(ctx);
    return rc;
}

// the parent: N fresh processes of this binary, nothing of HIP before (or after) the spawn
int launch(const Options &o, char **argv)
{
    char id_file[64];
    std::snprintf(id_file, sizeof(id_file), "/tmp/mcbeam_%ld.id", (long)getpid());
    std::remove(id_file);
    setenv("HSA_ENABLE_IPC_MODE_LEGACY", "0", 0);              // (dmabuf IPC: what this pool's driver supports)
    setenv("MCBEAM_WORLD", std::to_string(o.ranks).c_str(), 1);
    setenv("MCBEAM_ID_FILE", id_file, 1);
    std::vector<pid_t> pids;
    for (int r = 0; r < o.ranks; ++r) {
        setenv("MCBEAM_RANK", std::to_string(r).c_str(), 1);
        pid_t pid = 0;
        if (posix_spawn(&pid, "/proc/self/exe", nullptr, nullptr, argv, environ) != 0) { std::perror("posix_spawn"); break; }
        pids.push_back(pid);
    }
    int worst = (int)pids.size() == o.ranks ? 0 : 1;
    // a rank that dies before the rendezvous would leave the others waiting in it: the first failure ends the rest
    size_t left = pids.size();
    while (left) {
        int st = 0;
        const pid_t done = waitpid(-1, &st, 0);
        if (done <= 0) break;
        --left;
        const int code = WIFEXITED(st) ? WEXITSTATUS(st) : 128 + (WIFSIGNALED(st) ? WTERMSIG(st) : 0);
        if (code && !worst) { for (pid_t p : pids) if (p != done) kill(p, SIGTERM); }
        worst = std::max(worst, code);
    }
    std::remove(id_file);
    return worst;
}

}  // namespace

int main(int argc, char **argv)
{
    Options o;
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        auto val = [&](int &dst) { if (i + 1 < argc) dst = std::atoi(argv[++i]); };
        if (a == "--ranks" || a == "--devices") val(o.ranks); else if (a == "--arrays") val(o.arrays); else if (a == "--frames") val(o.frames);
        else if (a == "--steps") val(o.steps); else if (a == "--audio") o.audio = 1; else if (a == "--check") o.check = 1;
        else if (a == "--precision" && i + 1 < argc) {
            const std::string p = argv[++i];
            o.precision = p == "fp32" ? MCA_HIP_SRP_FP32 : p == "fp16x3" ? MCA_HIP_SRP_FP16X3 : p == "fp16" ? MCA_HIP_SRP_FP16 : MCA_HIP_SRP_ADAPTIVE;
        } else if (a == "--partition") {      // print the blocks and exit (no GPU needed): used by tests/test_partition.py
            for (int r = 0; r < o.ranks; ++r) { const mca::ArrayBlock b = mca::localArrays(o.arrays, r, o.ranks); std::printf("%d %d %d\n", r, b.first, b.count); }
            return 0;
        } else { std::fprintf(stderr, "usage: %s [--ranks N] [--arrays A] [--frames F] [--steps K] [--audio] [--check] [--precision P] [--partition]\n", argv[0]); return 2; }
    }
    if (o.ranks < 1 || o.arrays < 1 || o.frames < 64 || o.steps < 1) { std::fprintf(stderr, "bad arguments\n"); return 2; }
    const char *rk = std::getenv("MCBEAM_RANK");
    if (!rk) return launch(o, argv);
    const char *idf = std::getenv("MCBEAM_ID_FILE");
    const char *ws = std::getenv("MCBEAM_WORLD");
    if (!idf || !ws || std::atoi(ws) != o.ranks) { std::fprintf(stderr, "rank environment incomplete\n"); return 2; }
    return run_rank(o, std::atoi(rk), o.ranks, idf);
}
