// Does a CU-masked stream confine its kernels, and can a chain of small dependent kernels run beside a machine-filling kernel on the
// complementary mask without waiting for its workgroups?  (round 5: the repair chain beside the beamformer)
// hipcc --offload-arch=gfx950 -O3 cumask_probe.hip -o cumask_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <set>
#include <chrono>
__global__ void k_where(unsigned *out)
{
    unsigned xcc, hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    if (threadIdx.x == 0) out[blockIdx.x] = ((xcc & 0xf) << 16) | ((hw >> 8) & 0xf) | (((hw >> 13) & 0x7) << 4) | (((hw >> 12) & 1) << 7);   // cu_id, se_id, sh_id
}
__global__ __launch_bounds__(256) void k_hog(float *x, int iters)      // long workgroups, like the beamformer's
{
    float v = x[blockIdx.x * 256 + threadIdx.x];
    for (int i = 0; i < iters; ++i) v = fmaf(v, 1.0001f, 0.5f);
    x[blockIdx.x * 256 + threadIdx.x] = v;
}
__global__ __launch_bounds__(256) void k_small(float *x, int iters)    // a link of the chain: few workgroups, short
{
    float v = x[blockIdx.x * 256 + threadIdx.x];
    for (int i = 0; i < iters; ++i) v = fmaf(v, 1.0001f, 0.5f);
    x[blockIdx.x * 256 + threadIdx.x] = v;
}
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main()
{
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int ncu = p.multiProcessorCount;
    printf("CUs %d\n", ncu);
    float *x; hipMalloc(&x, 4096 * 256 * 4); hipMemset(x, 0, 4096 * 256 * 4);
    unsigned *w; hipMalloc(&w, 4096 * 4);
    auto mk = [&](int lo, int hi) {       // CUs [lo, hi) of the flat numbering
        std::vector<uint32_t> m((ncu + 31) / 32, 0u);
        for (int i = lo; i < hi; ++i) m[i / 32] |= 1u << (i % 32);
        hipStream_t s; hipError_t e = hipExtStreamCreateWithCUMask(&s, (uint32_t)m.size(), m.data());
        if (e != hipSuccess) { printf("hipExtStreamCreateWithCUMask failed: %s\n", hipGetErrorString(e)); exit(1); }
        return s;
    };
    const int small = ncu / 8;
    hipStream_t sa = mk(small, ncu), sb = mk(0, small), s0; hipStreamCreate(&s0);
    for (hipStream_t s : {s0, sa, sb}) {
        hipLaunchKernelGGL(k_where, dim3(4096), dim3(64), 0, s, w); hipStreamSynchronize(s);
        std::vector<unsigned> h(4096); hipMemcpy(h.data(), w, 4096 * 4, hipMemcpyDeviceToHost);
        std::set<unsigned> cus, xccs; for (unsigned v : h) { cus.insert(v); xccs.insert(v >> 16); }
        printf("stream %s: %zu distinct (xcc, se, sh, cu) places, %zu XCCs\n", s == s0 ? "unmasked" : s == sa ? "mask big  " : "mask small", cus.size(), xccs.size());
    }
    auto chain = [&](hipStream_t s, int links) { for (int i = 0; i < links; ++i) hipLaunchKernelGGL(k_small, dim3(48), dim3(256), 0, s, x, 2000); };
    auto timeit = [&](const char *name, auto fn) {
        fn(); hipDeviceSynchronize();
        double best = 1e9;
        for (int r = 0; r < 5; ++r) { const double t = now(); fn(); hipDeviceSynchronize(); best = std::min(best, now() - t); }
        printf("%-70s %8.1f us\n", name, best * 1e6);
    };
    const int hog_wg = 4 * ncu, hog_it = 60000;
    timeit("hog alone, unmasked stream", [&] { hipLaunchKernelGGL(k_hog, dim3(hog_wg), dim3(256), 0, s0, x, hog_it); });
    timeit("hog alone, big mask", [&] { hipLaunchKernelGGL(k_hog, dim3(hog_wg), dim3(256), 0, sa, x, hog_it); });
    timeit("chain of 4 alone, unmasked stream", [&] { chain(s0, 4); });
    timeit("chain of 4 alone, small mask", [&] { chain(sb, 4); });
    timeit("serial: chain then hog, one unmasked stream", [&] { chain(s0, 4); hipLaunchKernelGGL(k_hog, dim3(hog_wg), dim3(256), 0, s0, x, hog_it); });
    hipStream_t s1; hipStreamCreate(&s1);
    timeit("side by side, two UNMASKED streams (hog first)", [&] { hipLaunchKernelGGL(k_hog, dim3(hog_wg), dim3(256), 0, s0, x, hog_it); chain(s1, 4); });
    timeit("side by side, complementary masks (hog first)", [&] { hipLaunchKernelGGL(k_hog, dim3(hog_wg), dim3(256), 0, sa, x, hog_it); chain(sb, 4); });
    // how long does the chain itself take beside the hog?
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int masked = 0; masked < 2; ++masked) {
        hipStream_t hs = masked ? sa : s0, cs = masked ? sb : s1;
        hipDeviceSynchronize();
        hipLaunchKernelGGL(k_hog, dim3(hog_wg), dim3(256), 0, hs, x, hog_it);
        hipEventRecord(e0, cs); chain(cs, 4); hipEventRecord(e1, cs);
        hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("chain of 4 beside the hog, %s: %8.1f us\n", masked ? "complementary masks" : "unmasked streams   ", ms * 1e3);
    }
    return 0;
}
