// Probe of v_mfma_scale_f32_32x32x64_f8f6f4 with e4m3 operands on gfx950: operand lane maps and E8M0 scales,
// checked with exact integer data.  Build: hipcc --offload-arch=gfx950 -O2 mx8_probe.hip -o mx8_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// e4m3 encode of small integers / halves (exact)
__host__ __device__ inline unsigned char e4m3(float v)
{
    if (v == 0.f) return 0;
    unsigned char s = v < 0 ? 0x80 : 0; v = fabsf(v);
    int e; float m = frexpf(v, &e);          // v = m 2^e, m in [0.5,1)
    // normal: 1.mmm 2^(E-7), E in 1..15
    int E = e - 1 + 7; float frac = m * 2.f - 1.f;
    if (E < 1) { int mant = (int)lrintf(v * 512.f); return s | (unsigned char)mant; }   // subnormal: mant/8 * 2^-6
    int mant = (int)lrintf(frac * 8.f);
    if (mant == 8) { mant = 0; ++E; }
    return s | (unsigned char)((E << 3) | mant);
}

__global__ void k(const unsigned char *A, const unsigned char *B, float *C, int sa, int sb)
{
    // assumed map: lane l (r = l & 31, h = l >> 5) holds A[r][32 h + j], B[32 h + j][r], j = 0..31, byte j of the 8 VGPRs
    const int l = threadIdx.x, r = l & 31, h = l >> 5;
    v8i a, b;
    for (int w = 0; w < 8; ++w) {
        unsigned av = 0, bv = 0;
        for (int q = 0; q < 4; ++q) {
            const int kk = 32 * h + 4 * w + q;
            av |= (unsigned)A[r * 64 + kk] << (8 * q);
            bv |= (unsigned)B[kk * 32 + r] << (8 * q);
        }
        a[w] = (int)av; b[w] = (int)bv;
    }
    f32x16 c;
    for (int i = 0; i < 16; ++i) c[i] = 0.f;
    c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, sa, 0, sb);
    for (int i = 0; i < 16; ++i) {
        const int row = (i & 3) + 8 * (i >> 2) + 4 * h;
        C[row * 32 + r] = c[i];
    }
}

int main()
{
    unsigned char hA[32 * 64], hB[64 * 32];
    float fA[32 * 64], fB[64 * 32];
    srand(1);
    for (int i = 0; i < 32 * 64; ++i) { fA[i] = (float)(rand() % 9 - 4) * 0.5f; hA[i] = e4m3(fA[i]); }
    for (int i = 0; i < 64 * 32; ++i) { fB[i] = (float)(rand() % 7 - 3); hB[i] = e4m3(fB[i]); }
    unsigned char *dA, *dB; float *dC;
    hipMalloc(&dA, sizeof(hA)); hipMalloc(&dB, sizeof(hB)); hipMalloc(&dC, 32 * 32 * 4);
    hipMemcpy(dA, hA, sizeof(hA), hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof(hB), hipMemcpyHostToDevice);
    const int tests[3][2] = {{127, 127}, {115, 127}, {120, 130}};
    int bad_total = 0;
    for (int t = 0; t < 3; ++t) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dC, tests[t][0], tests[t][1]);
        float hC[32 * 32];
        hipMemcpy(hC, dC, sizeof(hC), hipMemcpyDeviceToHost);
        const double sc = ldexp(1.0, tests[t][0] - 127 + tests[t][1] - 127);
        int bad = 0;
        for (int i = 0; i < 32; ++i)
            for (int j = 0; j < 32; ++j) {
                double ref = 0;
                for (int kk = 0; kk < 64; ++kk) ref += (double)fA[i * 64 + kk] * fB[kk * 32 + j];
                if (fabs(hC[i * 32 + j] - ref * sc) > 1e-6 * fabs(ref * sc) + 1e-12) { if (bad < 3) printf("  mismatch [%d][%d] got %g want %g\n", i, j, hC[i * 32 + j], ref * sc); ++bad; }
            }
        printf("scales (%d,%d): %d mismatches\n", tests[t][0], tests[t][1], bad);
        bad_total += bad;
    }
    printf(bad_total ? "PROBE FAILED\n" : "PROBE OK\n");
    return bad_total != 0;
}
