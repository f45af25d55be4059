// What does a device-side work counter cost on an MI355X?  (round 5: the run queue of k_stft_phat_wave)
// 2048 waves (512 workgroups of 256 threads, as the analysis kernel) each take `n` tickets from (a) one counter, (b) one counter per XCD
// (XCC_ID), (c) one counter per workgroup, with the next ticket asked for only when the previous one has arrived (latency bound) and
// with `work` microseconds of ALU work between tickets.  hipcc --offload-arch=gfx950 -O3 atomic_queue_probe.hip -o atomic_queue_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ __launch_bounds__(256) void k(unsigned *q, int mode, int n, int spin, unsigned long long *cycles, unsigned *sink)
{
    const int lane = threadIdx.x & 63;
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    xcc &= 0xf;
    unsigned *c = mode == 0 ? q : mode == 1 ? q + 64 * xcc : q + 64 * blockIdx.x;
    unsigned acc = 0;
    float x = (float)lane;
    unsigned long long t = 0;
    for (int i = 0; i < n; ++i) {
        const unsigned long long t0 = wall_clock64();
        unsigned v = 0;
        if (lane == 0) v = atomicAdd(c, 1u);
        v = __builtin_amdgcn_readfirstlane(v);
        t += wall_clock64() - t0;
        acc += v;
        for (int s = 0; s < spin; ++s) x = fmaf(x, 1.0001f, 0.5f);
    }
    if (lane == 0) { atomicAdd(&cycles[0], t); sink[blockIdx.x * 4 + (threadIdx.x >> 6)] = acc + (unsigned)x; }
}
int main()
{
    unsigned *q, *sink; unsigned long long *cyc;
    hipMalloc(&q, 64 * 4 * 1024); hipMalloc(&sink, 2048 * 4); hipMalloc(&cyc, 8);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int n = 64;
    for (int spin : {0, 2000, 8000})
        for (int mode = 0; mode < 3; ++mode) {
            float best = 1e9f; unsigned long long cy = 0;
            for (int rep = 0; rep < 3; ++rep) {
                hipMemset(q, 0, 64 * 4 * 1024); hipMemset(cyc, 0, 8);
                hipEventRecord(a);
                hipLaunchKernelGGL(k, dim3(512), dim3(256), 0, 0, q, mode, n, spin, cyc, sink);
                hipEventRecord(b); hipEventSynchronize(b);
                float ms; hipEventElapsedTime(&ms, a, b);
                if (ms < best) { best = ms; hipMemcpy(&cy, cyc, 8, hipMemcpyDeviceToHost); }
            }
            printf("spin %5d  %-22s kernel %8.1f us   %7.1f ns per ticket of the kernel's time (2048 waves x %d)   mean wait per ticket %7.2f us (100 MHz clock)\n",
                   spin, mode == 0 ? "one counter" : mode == 1 ? "one counter per XCD" : "one per workgroup", best * 1e3, best * 1e6 / (2048.0 * n), n,
                   (double)cy / (2048.0 * n) / 100.0);
        }
    return 0;
}
