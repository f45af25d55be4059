// mca_internal.h -- argument blocks shared by the kernels and the C-ABI layer.
#pragma once
#include <hip/hip_runtime.h>

#define MCA_MAX_SOURCES 4
#define MCA_MAX_MICS 16

namespace mca {

constexpr int BF_NB = 4;           // frames per inverse-FFT batch of k_beamform_ola
constexpr int SCAN_WARM = 128;    // frames of IIR warm-up per scan chunk (0.8^128 = 4e-13)

struct StftPhatArgs {
    const float *pcm;
    long long array_stride, mic_stride;
    int M, n_frames, frame0, fpb;
    const float *window;     // [1024] periodic Hann
    void *A;                 // [arrays][n_frames][a_row_elems]
    int Kp;                  // padded contraction depth (elements per plane)
    int a_row_elems;         // elements per A row = Kp * planes
    int a_planes;            // 1 (fp32 / fp16) or 2 (fp16 hi plane + lo plane)
};

__device__ __forceinline__ void store_a(float *row, const StftPhatArgs &, int cidx, float2 v)
{
    reinterpret_cast<float2 *>(row)[cidx] = v;
}

typedef _Float16 half2_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void store_a(_Float16 *row, const StftPhatArgs &p, int cidx, float2 v)
{
    half2_t hi;
    hi[0] = (_Float16)v.x; hi[1] = (_Float16)v.y;
    reinterpret_cast<half2_t *>(row)[cidx] = hi;
    if (p.a_planes == 2) {
        half2_t lo;
        lo[0] = (_Float16)(v.x - (float)hi[0]); lo[1] = (_Float16)(v.y - (float)hi[1]);
        reinterpret_cast<half2_t *>(row + p.Kp)[cidx] = lo;
    }
}

struct GemmArgs {
    const void *A;           // [rows][a_row_elems]
    const void *B;           // fp32: [Kp][Dp]; fp16: [planes][Dp][Kp] (k contiguous)
    float *C;                // [arrays][total_frames][Dp]
    int rows;                // arrays * chunk_frames
    int chunk_frames, total_frames, frame0;
    int Kp, Dp, a_row_elems;
};

struct ScanPickArgs {
    const float *C;          // [arrays][n_frames][Dp]
    int n_frames, Dp, D, P, S, chunk;
    float mu, one_minus_mu;
    const float *state_in;   // [arrays][D]  E_prev at entry
    float *state_out;        // [arrays][D]  E_prev at exit
    const float *grid;       // [D] doaIdx2angle
    int *doa_bin; float *doa_rad; float *prob; float *energy;
};

struct BeamformArgs {
    const float *pcm;
    long long array_stride, mic_stride;
    int M, Mpad, S, n_frames, ft, fs;
    const float *window;
    const double *mic_x;     // [M] x coordinate (Beamformer.cpp:59 steers with x only)
    const float *doa_rad;    // [arrays][n_frames][S]
    float *out;              // [arrays][S][n_frames*hop]
    const float *tail_in;    // [arrays][S][hop] overlap-add carry at entry
    float *tail_out;         // at exit
};

}  // namespace mca
