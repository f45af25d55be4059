// kernels.h -- declarations of the kernels defined in kernels_*.hip (explicitly instantiated there).
#pragma once
#include "mca_internal.h"

namespace mca {

template <int MT, bool ULA, typename OutT> __global__ void k_stft_phat(StftPhatArgs p);
template <int MT, bool ULA, typename OutT> __global__ void k_stft_phat_few(StftPhatArgs p);
__global__ void k_sum_planes(float *C, long long n4, int planes, long long stride);
__global__ void k_scan_partial(ScanPickArgs p);
__global__ void k_scan_carry(ScanPickArgs p);
template <int PL, int MODE> __global__ void k_scan_pick(ScanPickArgs p);   // PL: positions per lane of the peak pick (2 / 6 / 8, by D); MODE 0 plain, 1 adaptive coarse pass, 2 the same with two work lists
template <int PL> __global__ void k_scan_repick(ScanPickArgs p);
__global__ void k_repair_patch(RepairPatchArgs p);
__global__ void k_hist_list(int *list, int *n_list, int *need, int n_units);      // lazy tails (round 5): settle_history
__global__ void k_hist_settle(const float *e_hist, const float *hist_C, float *state, int *n_list, int D, int Dp, float mu, float omu);
__global__ void k_gate(GateArgs p);
__global__ void k_doa_fill(DoaFillArgs p);
template <int CPW, int OCC> __global__ void k_beamform_ola(BeamformArgs p);
template <typename OutT> __global__ void k_stft_phat_gen(StftPhatArgs p);
template <int MT, bool ULA, typename OutT> __global__ void k_stft_phat_512(StftPhatArgs p);
__global__ void k_beamform_512(BeamformArgs p);
template <int R, typename OutT> __global__ void k_stft_phat_sub2(StftPhatArgs p);
__global__ void k_beamform_gen(BeamformArgs p);
__global__ void k_bf_table(float2 *tab, const float *grid, const double *mic_x, int M, int n_pairs, double unit);
template <bool ODD, int VAR, int ABL> __global__ void k_beamform_wave(BeamformWaveArgs p);
template <bool POWER> __global__ void k_stft_phat_wave16(StftPhatArgs p);   // 16-microphone ULA, one fp16 plane
template <int MT, bool ULA, typename OutT, bool MERGE = false> __global__ void k_stft_phat_2048(StftPhatArgs p);   // 2048-sample frames, M <= 8 (kernels_2048.hip)
__global__ void k_bf_table_2048(float2 *tab, const float *grid, const double *mic_x, int M, double unit);
__global__ void k_beamform_wave_2048(BeamformWaveArgs p);
template <int NPT, bool ODD> __global__ void k_beamform_wave_ms(BeamformWaveArgs p);   // several sources, forward transforms shared (M <= 8)
template <int MT, bool ULA, typename OutT, bool PL2, bool POWER, bool NOPHAT, bool MERGE, bool CAND = false> __global__ void k_stft_phat_wave(StftPhatArgs p);

__global__ void k_srp_gemm_f32(GemmArgs p);
template <bool SPLIT, int BN> __global__ void k_srp_gemm_f16(GemmArgs p);
template <bool SPLIT> __global__ void k_srp_gemm_f16_v2(GemmArgs p);
__global__ void k_srp_gemm_f16_v3(GemmArgs p);      // one plane, v_mfma_f32_16x16x32_f16
template <int BN> __global__ void k_srp_gemm_repair(GemmArgs p);
__global__ void k_srp_cand(CandArgs p);

template <typename T> struct C2;
template <typename T>
__global__ void k_frame_srp(const C2<T> *X, int K, int D, int P, const int2 *pairs, const float *delays,
                            const T *E_in, T *E_out, T mu, T omu, int no_phat);
template <typename T>
__global__ void k_frame_pick(const T *E, int D, int P, int S, const float *grid, T *doa, T *prob, int *bins);
template <typename T>
__global__ void k_frame_beamform(const C2<T> *X, int M, int K, int fs, const double *mic_x, double doa, C2<T> *Y);
template <typename T> __global__ void k_frame_power(const C2<T> *X, int M, int K, T *out);

__global__ void k_gcc2_scan(Gcc2ScanArgs p);
__global__ void k_gcc2_compact(const unsigned char *voiced, int n_frames, int *vidx, int *nv, const int *post0, int *silence,
                               int windows_to_decay, unsigned char *vreset);
__global__ void k_gcc2_fill(Gcc2FillArgs p);
__global__ void k_mask_stream(MaskArgs p);
__global__ void k_mask_stream_gen(MaskGenArgs p);
__global__ void k_mask_stream_2048(MaskGenArgs p);
__global__ void k_mask_frame(MaskFrameArgs p);
__global__ void k_mb_analyse(MbAnalyseArgs p);
__global__ void k_mb_analyse_1024(MbAnalyseArgs p, int fpb);
__global__ void k_mb_analyse_512(MbAnalyseArgs p, int fpb);
__global__ void k_mb_scan(MbScanArgs p);
__global__ void k_mb_summary(MbSummaryArgs p);
__global__ void k_mvdr_analyse(MvdrAnalyseArgs p);
__global__ void k_mvdr_analyse_1024(MvdrAnalyseArgs p, int fpb);
__global__ void k_mvdr_analyse_512(MvdrAnalyseArgs p, int fpb);
template <int Q, bool FULL> __global__ void k_mvdr_solve(MvdrSolveArgs p);
__global__ void k_mvdr_synth(MvdrSynthArgs p);

}  // namespace mca
