// round 5: pins the operand / result layout of v_mfma_f32_4x4x4_16b_f16 that k_srp_cand relies on:
// lane l = 4 b + q: A holds row q, B holds column q of block b (4 halves = k 0..3); D register i of lane l = element (row i, column q) of block b.
// build: hipcc --offload-arch=gfx950 -O2 -o /tmp/mfma4x4_probe tools/probes/mfma4x4_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ void k(const _Float16 *A, const _Float16 *B, float *D)
{
    const int l = threadIdx.x, b = l >> 2, q = l & 3;
    h4 a, bb;
    for (int k = 0; k < 4; ++k) { a[k] = A[(b * 4 + q) * 4 + k]; bb[k] = B[(b * 4 + q) * 4 + k]; }   // A[b][row q][k], B[b][col q][k]
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    acc = __builtin_amdgcn_mfma_f32_4x4x4f16(a, bb, acc, 0, 0, 0);
    for (int i = 0; i < 4; ++i) D[(b * 4 + i) * 4 + q] = acc[i];                                      // D[b][row i][col q]
}
int main()
{
    _Float16 hA[256], hB[256]; float hD[256], ref[256];
    for (int i = 0; i < 256; ++i) { hA[i] = (_Float16)((i * 7 % 13) - 6); hB[i] = (_Float16)((i * 5 % 11) - 5); }
    for (int b = 0; b < 16; ++b) for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) {
        float s = 0; for (int k = 0; k < 4; ++k) s += (float)hA[(b * 4 + i) * 4 + k] * (float)hB[(b * 4 + j) * 4 + k];
        ref[(b * 4 + i) * 4 + j] = s;
    }
    _Float16 *dA, *dB; float *dD;
    hipMalloc(&dA, 512); hipMalloc(&dB, 512); hipMalloc(&dD, 1024);
    hipMemcpy(dA, hA, 512, hipMemcpyHostToDevice); hipMemcpy(dB, hB, 512, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dD);
    hipMemcpy(hD, dD, 1024, hipMemcpyDeviceToHost);
    int bad = 0; for (int i = 0; i < 256; ++i) bad += hD[i] != ref[i];
    printf("mfma 4x4x4 f16 layout (lane = 4 block + row/col, D register = row): %s (%d of 256 differ)\n", bad ? "WRONG" : "as assumed", bad);
    return bad != 0;
}
