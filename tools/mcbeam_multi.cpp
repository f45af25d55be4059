// mcbeam_multi.cpp -- many independent microphone arrays on the GPUs of one node through the C ABI (BASELINE configs[4]:
// 1024 independent 8-mic arrays sharded across 8 x MI355X, DOA / audio gathered), the multi-device counterpart of the
// reference's single-stream driver loop (src/programs/mcabeamf.cpp:77-122: read a block, process(), write the output).
//
// One host thread per device: thread r creates a context with cfg.device = r for its contiguous block of arrays
// (mca::localArrays, the same blocks as bench.py / mcarray_amd/dist.py), runs the whole path on them -- no collective inside
// the compute -- and writes its DOA bins / probabilities (and, with --audio, the beamformed audio) into its slice of the
// shared host result arrays: the gather is the threads' writes into one address space.  (One PROCESS per GPU with an RCCL
// all-gather of the same buffers is what bench.py --gpus N runs; this is the in-process form a C++ host application uses.)
// Page-locked buffers (mca_hip_host_alloc): uploads, kernels and downloads of a call overlap inside the library.
//
//   g++ -std=c++11 -O2 -pthread -Iinclude tools/mcbeam_multi.cpp -o mcbeam_multi -Lmcarray_amd -lmcarray_hip -Wl,-rpath,$PWD/mcarray_amd
//   ./mcbeam_multi --devices 8 --arrays 1024 --frames 256 --steps 4
// Input: synthetic far-field white sources (one per array, angle from the array's global index), like bench.py.
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

namespace {

struct Options { int devices = 1, arrays = 8, frames = 256, steps = 2, audio = 0; double step_deg = 0.5; };

// splitmix64 -> uniform -> Box-Muller: the input of array g depends on g only, not on the number of devices
struct Rng {
    unsigned long long s;
    unsigned long long next() { unsigned long long z = (s += 0x9E3779B97F4A7C15ull); z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); }
    double uni() { return ((next() >> 11) + 0.5) * (1.0 / 9007199254740992.0); }
    double gauss() { return std::sqrt(-2.0 * std::log(uni())) * std::cos(6.283185307179586 * uni()); }
};

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

struct Shared { int *bins; float *prob; float *audio; double *thetas; int failed; };

void run_device(const Options &o, int rank, Shared *sh, double *ms_out)
{
    const mca::ArrayBlock blk = mca::localArrays(o.arrays, rank, o.devices);
    if (blk.count == 0) { *ms_out = 0; return; }
    const int M = 8, N = 1024, H = N / 2, fs = 48000, F = o.frames;
    std::vector<double> xyz((size_t)M * 3, 0.0);
    for (int m = 0; m < M; ++m) xyz[3 * m] = 0.04 * m;
    mca_hip_config cfg = mca_hip_config();
    cfg.struct_size = (int)sizeof(cfg); cfg.device = rank; cfg.sample_rate = fs; cfg.fft_size = N; cfg.n_mics = M; cfg.mic_xyz = xyz.data();
    cfg.doa_step_deg = o.step_deg; cfg.n_sources = 1; cfg.use_power_floor = 0; cfg.srp_precision = MCA_HIP_SRP_ADAPTIVE; cfg.max_arrays = blk.count;
    mca_hip_ctx *ctx = nullptr;
    if (mca_hip_create(&cfg, &ctx) != MCA_HIP_OK) { std::fprintf(stderr, "device %d: %s\n", rank, mca_hip_last_error(nullptr)); sh->failed = 1; return; }
    const long long pitch = (long long)(F + 1) * H, per_array = pitch * M;
    float *pcm = static_cast<float *>(mca_hip_host_alloc((long long)sizeof(float) * per_array * blk.count));
    float *rad = static_cast<float *>(mca_hip_host_alloc((long long)sizeof(float) * blk.count * F));
    if (!pcm || !rad) { std::fprintf(stderr, "device %d: page-locked allocation failed\n", rank); sh->failed = 1; mca_hip_destroy(ctx); return; }
    for (int a = 0; a < blk.count; ++a) {
        const int g = blk.first + a;
        sh->thetas[g] = (-80.0 + 160.0 * ((g * 37) % 101) / 100.0) * 3.141592653589793 / 180.0;
        synth_array(pcm + (size_t)a * per_array, M, pitch, pitch, 0.04, sh->thetas[g], fs, 0x5EED0000ull + (unsigned long long)g);
    }
    // this rank's slice of the gathered results: [arrays][frames]
    int *bins = sh->bins + (size_t)blk.first * F;
    float *prob = sh->prob + (size_t)blk.first * F;
    float *audio = o.audio ? sh->audio + (size_t)blk.first * F * H : nullptr;
    const auto t0 = std::chrono::steady_clock::now();
    for (int s = 0; s < o.steps; ++s) {
        // (a live application would refill pcm with the next block here, as mcabeamf.cpp:101-112 does per 1024 samples)
        const int rc = mca_hip_process_frames_host(ctx, pcm, blk.count, F, bins, rad, prob, nullptr, audio);
        if (rc != MCA_HIP_OK) { std::fprintf(stderr, "device %d: %s\n", rank, mca_hip_last_error(ctx)); sh->failed = 1; break; }
    }
    *ms_out = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    mca_hip_host_free(pcm); mca_hip_host_free(rad);
    mca_hip_destroy(ctx);
}

}  // namespace

int main(int argc, char **argv)
{
    Options o;
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        auto val = [&](int &dst) { if (i + 1 < argc) dst = std::atoi(argv[++i]); };
        if (a == "--devices") val(o.devices); else if (a == "--arrays") val(o.arrays); else if (a == "--frames") val(o.frames);
        else if (a == "--steps") val(o.steps); else if (a == "--audio") o.audio = 1;
        else if (a == "--partition") {      // print the blocks and exit (no GPU needed): used by tests/test_partition.py
            for (int r = 0; r < o.devices; ++r) { const mca::ArrayBlock b = mca::localArrays(o.arrays, r, o.devices); std::printf("%d %d %d\n", r, b.first, b.count); }
            return 0;
        } else { std::fprintf(stderr, "usage: %s [--devices N] [--arrays A] [--frames F] [--steps K] [--audio] [--partition]\n", argv[0]); return 2; }
    }
    if (o.devices < 1 || o.arrays < 1 || o.frames < 64 || o.steps < 1) { std::fprintf(stderr, "bad arguments\n"); return 2; }
    const size_t nf = (size_t)o.arrays * o.frames;
    std::vector<int> bins(nf, -2);
    std::vector<float> prob(nf), audio(o.audio ? nf * 512 : 0);
    std::vector<double> thetas((size_t)o.arrays), ms((size_t)o.devices, 0.0);
    Shared sh{bins.data(), prob.data(), audio.empty() ? nullptr : audio.data(), thetas.data(), 0};
    std::vector<std::thread> th;
    for (int r = 0; r < o.devices; ++r) th.emplace_back(run_device, std::cref(o), r, &sh, &ms[(size_t)r]);
    for (std::thread &t : th) t.join();
    if (sh.failed) return 1;
    // every array's DOA after the recursion has settled: within a grid step of its source (integer-sample delays: a few steps)
    int bad = 0;
    double worst_ms = 0;
    for (double m : ms) worst_ms = std::fmax(worst_ms, m);
    for (int g = 0; g < o.arrays; ++g) {
        const int b = bins[(size_t)g * o.frames + o.frames - 1];
        const double deg = b * o.step_deg - 90.0, want = thetas[(size_t)g] * 180.0 / 3.141592653589793;
        if (b < 0 || std::fabs(deg - want) > 3.0) ++bad;
    }
    std::printf("%d arrays x %d frames on %d device(s), %d steps: %.1f ms wall (slowest device), %.2f M frames/s incl. PCIe; %d arrays off their source\n",
                o.arrays, o.frames, o.devices, o.steps, worst_ms, (double)nf * o.steps / worst_ms / 1e3, bad);
    return bad ? 1 : 0;
}
