// mca_internal.h -- argument blocks shared by the kernels and the C-ABI layer.
#pragma once
#include <hip/hip_runtime.h>

#define MCA_MAX_SOURCES 4
#define MCA_MAX_MICS 16

namespace mca {

constexpr int MK_NB = 4;           // frames per batch of k_mask_stream (8 waves = 4 frames x 2 channels)
constexpr int MK_WARM = 8;         // frames of Q warm-up for mask chunks that do not start a stream (0.04^8 = 6.6e-12)
constexpr int BF_NB = 4;           // frames per inverse-FFT batch of k_beamform_ola
constexpr int GCC2_DOAWARM = 64;   // frames of DOA-recursion warm-up in k_gcc2_scan (0.6^64 = 6e-15)
constexpr int SCAN_WARM = 96;     // frames of IIR warm-up per scan chunk (0.8^96 = 5e-10 << fp32 epsilon)
constexpr int KG = 513;            // complex K-slots per delay group in the A / B contraction index (g * KG + k)
#ifndef MCA_SCAN_CHUNK
#define MCA_SCAN_CHUNK 32
#endif
#ifndef MCA_SCAN_SUB
#define MCA_SCAN_SUB 32
#endif
constexpr int SCAN_CHUNK = MCA_SCAN_CHUNK;    // frames per chunk of the exact chunked scan (32: 0.928 ms per bench step, 64: 0.965; must exceed REPAIR_WARM)
constexpr int SCAN_SUB = MCA_SCAN_SUB;      // frames per LDS sub-batch of k_scan_pick
#ifndef MCA_SCAN_LD
#define MCA_SCAN_LD 8
#endif
constexpr int SCAN_LD = MCA_SCAN_LD;        // map rows (x split-K planes) a thread of the scan kernels keeps in flight (16 / 32 measured slower: 0.939 / 0.955 vs 0.939 ms per step)
#ifndef MCA_REPAIR_WARM
#define MCA_REPAIR_WARM 16
#endif
constexpr int REPAIR_WARM = MCA_REPAIR_WARM;   // exact rows recomputed BEFORE a flagged frame (adaptive SRP precision): 0.8^17 = 2.3e-2 of the coarse
                                  // error (~1.6e-5 of the map's peak) remains, i.e. ~3.6e-7 of the peak -- five times below the error of the
                                  // three-product split itself and below the 1e-6 tie bar of the parity tests.  (Round 2 used 24 rows,
                                  // 0.8^25: the always-recomputed tail of every array and call is what the repair pass mostly works on
                                  // when a call has many arrays and few frames -- 128 x 256: 13.9 % of the rows with 24, 10.7 % with 16.
                                  // Round 4, ADVICE r3: the flips against FP16X3 over 40 configurations / 696 320 frames are the SAME with
                                  // 16 and with 24 rows, every differing frame a tie at the parity bar's level (profiles/r04_adaptive_check.json /
                                  // _warm24.json; the count of the shipped build is the one in include/mcarray_hip.h: 41 picks on 29 frames,
                                  // profiles/r05_adaptive_check.json); 24 rows cost +12 % / +30 % repair time at 8 x 4096 / 128 x 256.)
constexpr int REPAIR_GROUP = 4;   // frames per repair unit = frames per list-mode pass of a k_stft_phat workgroup
// Lazy tails (round 5; ADAPTIVE, ungated, the wave-per-run analysis): a call does NOT recompute its last REPAIR_WARM + 1 rows exactly for the
// state it hands over.  It keeps what the NEXT call needs to do so -- the last HIST_FRAMES frames of PCM (HIST_SAMPLES per channel), their
// coarse map rows and the energies in front of them -- and the next call lists them as repair units only when one of its first
// REPAIR_WARM frames is flagged (a few arrays per call instead of all of them: 2 560 of the 3 492 rows a 128 x 256 call recomputed).
// History units are numbered behind the call's own: hist_base + array * HIST_UNITS + u, frames 4 u .. 4 u + 3 of the history.
constexpr int HIST_FRAMES = REPAIR_WARM, HIST_UNITS = HIST_FRAMES / REPAIR_GROUP, HIST_SAMPLES = (HIST_FRAMES + 1) * 512;
static_assert(HIST_FRAMES % REPAIR_GROUP == 0, "history = whole repair units");

// cand_unit.h / k_srp_cand: the exact values of the listed rows AT THEIR CANDIDATE COLUMNS, written straight into the map.
// Which columns (wave_candidates, kernels_stream.hip): a flagged frame's pick is, on the exact map, among the positions whose coarse |En|
// reaches v - tau, v the best GUARANTEED peak of the coarse row -- a window in which the median-filtered sign chain, with every first
// difference within tau taken as of either sign, is pinned to "rising" at one end and to "falling" at the other: the exact map has a
// peak in there worth at least the window's smallest coarse energy - tau / 2, and a position below v - tau cannot reach that -- plus the
// two columns either side that feed the sign / median chain (:159-173) of such a position.  The second pick then runs on a row that is
// exact wherever it matters and coarse elsewhere.  A frame without such a window (a flat map) takes every column; the contexts that flag
// whole rows by construction (eager tails, unsure rows, several sources) keep the whole-row kernels (api.hip, cand_call).
constexpr int CAND_WORDS_MAX = 20;          // Dp <= 640 (the peak pick handles D <= 514)
struct CandArgs {
    const void *A;           // exact analysis rows of the listed units, fp16 hi + lo planes: [rows][a_row_elems], row = 4 x list position + frame
    const void *B;           // steering table, fp16 hi + lo planes: [2][Dp][Kp]
    int Kp, Dp, a_row_elems;
    const int *list; const int *n_list; int list0, pass_rows;
    unsigned *umask; int umask_words;                        // read and cleared
    int *need;                                               // the units' test-and-set words, released here
    int groups_per_array, n_frames;
    float *C; int c_planes; long long c_plane_stride;       // plane 0 takes the exact value, the others zeros (as k_repair_patch)
    float *hist_C; int hist_base;                            // lazy tails: units >= hist_base are rows of hist_C
};

struct StftPhatArgs {
    const float *pcm;
    long long array_stride, mic_stride;
    int M, n_frames, frame0, fpb;
    const float *window;     // [1024] periodic Hann
    void *A;                 // [arrays][n_frames][a_row_elems]
    int Kp;                  // padded contraction depth (elements per plane)
    int a_row_elems;         // elements per A row = Kp * planes
    int a_planes;            // 1 (fp32 / fp16) or 2 (fp16 hi plane + lo plane)
    float *power;            // [arrays][total_frames] linear FFTPower per frame, or NULL (ungated)
    int total_frames;
    // any-N kernels only (kernels_generic.hip)
    int N, logH, kg, ula;    // frame length, log2(N/2), K-slots per delay group (= N/2 + 1), delay-group merging on/off
    const float2 *tw;        // [N/2] exp(-j 2 pi i / N)
    // list mode (k_stft_phat only; the repair pass of the adaptive SRP precision): workgroup b takes the REPAIR_GROUP frames of
    // list[list0 + b] = array * groups_per_array + group and writes A rows b * REPAIR_GROUP ...; b >= *n_list - list0 exits
    const int *list; const int *n_list; int list0, list_cap, groups_per_array;   // list_cap: groups of this pass at most
    // lazy tails (see HIST_FRAMES): the coarse launch keeps the call's last HIST_FRAMES frames of PCM in hist_out [arrays][M][HIST_SAMPLES];
    // the list-mode launch of the NEXT call reads the units >= hist_base from hist_in (the same layout)
    float *hist_out; const float *hist_in; int hist_base;
    const unsigned short *mrank; int n_merged;   // k_stft_phat_wave, merged index (ULA, one fp16 plane): rank of the product m = k (j - i)
                             // among the n_merged distinct ones, [(M - 1) * 512 + 1]; NULL: per-group index g * 513 + k
    int no_phat;             // 1: gcc_weighting NONE -- the pair products of the spectra themselves (k_stft_phat_wave, fp32 rows only)
    int no_balance;          // (measurement, make MEASURE=1 + MCA_HIP_NO_BALANCE) 1: the two channels of a pair transform are never level-balanced (pair_balance.h): round 5's analysis
    // k_stft_phat_wave, dynamic runs (round 5): queue != NULL -- the grid is one resident wave set and every wave takes runs of frames off
    // a device-side counter until none is left: queue[0] next run, queue[1] waves that have left (the last one zeroes both: the words
    // are clean for the next launch, recorded graphs included).  The runs get shorter towards the end (dyn_run below: 8, 4, 2, 1 frames
    // at the bench shape), so that the waves finish within about a frame of each other whatever their individual speed.
    unsigned *queue;
    int q_sh0, q_total, q_arrays;   // log2 of the first runs' length, number of runs, arrays of the launch
    int q_flat;                     // (measurement) 1: every run has the first runs' length
    int xcd_map;                    // (measurement) 1: workgroups of one XCD (linear id mod 8) take neighbouring runs of frames
    int skew;                       // > 0 (round 5; a launch that is exactly one resident round of two workgroups per CU): grid (arrays, run groups),
                                    // and the run groups of the first half -- dispatched first: the OLDER workgroup of every CU, which the CU's
                                    // oldest-first issue favours (its waves were done at 180 us, the younger one's at 244: profiles/
                                    // r05_run_queue_negative.log) -- take fpb + skew frames per wave, the others fpb - skew, so that both finish
                                    // together (19 / 13 frames: 302 -> 281 us per 32 768 frames, profiles/r05_skew.log).  The rows do not
                                    // depend on which wave forms them: the same bits.
    unsigned long long *wave_clock; // (measurement, make MEASURE=1 + MCA_HIP_WAVE_CLOCK) [waves][3]: wall_clock64 at entry and exit, runs taken
    unsigned char *dead;     // k_stft_phat_wave in the coarse pass of a candidate-column call, else NULL: [arrays][total_frames] 1 = every channel of the frame is
                             // exact zeros (digital silence): its row is zero in the coarse and in the exact map alike, the repair pass need not list it
    unsigned char *unsure;   // k_stft_phat_wave16 in the adaptive coarse pass, else NULL: [arrays][total_frames] 1 = a channel's DC or Nyquist bin of this
                             // frame is at the rounding level of the transform.  PHAT keeps only the SIGN of such a bin, and the exact rows of 16
                             // microphones come from another kernel (k_stft_phat<16>) that need not round it the same way: k_scan_pick repairs the
                             // frame and the six after it whatever the map says ("A limit of PHAT itself")
    // list mode of a candidate-column call: the workgroup that wrote a unit's four rows contracts them at the unit's columns (cand_unit.h) --
    // k_srp_cand's work without its launch
    int cand_on; CandArgs cand;
};

// The run schedule of the dynamic mode, the same arithmetic on the host (the number of runs) and in the kernel (run r -> array, frames):
// per array, half of the frames that are left go in runs of 2^sh frames, then sh drops by one; the last phase takes one frame at a
// time.  Runs are numbered phase by phase, array-major inside a phase.  Shifts and multiplies only (scalar ALU) up to the one division
// by the caller.  Returns the number of runs when r is past the end (a, f_begin, f_end untouched).
__host__ __device__ inline int dyn_run(int r, int n_frames, int n_arrays, int sh0, int &rr, int &rpa, int &f_first, int &len, int &f_last, int flat = 0)
{
    int f0 = 0, first = 0;
    if (flat) {
        const int n = (n_frames + (1 << sh0) - 1) >> sh0;
        if (r < n * n_arrays) { rr = r; rpa = n; f_first = 0; len = 1 << sh0; f_last = n_frames; return -1; }
        return n * n_arrays;
    }
    for (int sh = sh0; ; --sh) {
        const bool lastph = sh <= 0;
        const int nfr = lastph ? n_frames - f0 : (((n_frames - f0) >> 1) >> sh) << sh;
        const int n = lastph ? nfr : nfr >> sh;
        if (r >= first && r < first + n * n_arrays) { rr = r - first; rpa = n; f_first = f0; len = lastph ? 1 : 1 << sh; f_last = f0 + nfr; return -1; }
        first += n * n_arrays; f0 += nfr;
        if (lastph) return first;
    }
}

__device__ __forceinline__ void store_a(float *row, const StftPhatArgs &, int cidx, float2 v)
{
    reinterpret_cast<float2 *>(row)[cidx] = v;
}

typedef _Float16 half2_t __attribute__((ext_vector_type(2)));

typedef float float2_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void store_a(_Float16 *row, const StftPhatArgs &p, int cidx, float2 v)
{
    // hi = fp16(v), lo = fp16(v - hi): vector conversions so that one packed convert serves both the store
    // of hi and the residual
    const float2_t vv = {v.x, v.y};
    const half2_t hi = __builtin_convertvector(vv, half2_t);
    reinterpret_cast<half2_t *>(row)[cidx] = hi;
    if (p.a_planes == 2) {
        const float2_t back = __builtin_convertvector(hi, float2_t);
        const half2_t lo = __builtin_convertvector(vv - back, half2_t);
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
    long long c_plane_elems;  // elements between the partial maps of a split-K launch
    const void *Bt;          // fp16 steering table tiled for the 256 x 384 kernel: [plane][Kp/32][Dp][32] (a K stage of all columns is contiguous)
    // device-side row count (repair pass): rows = min(rows, (*n_list - list0) * REPAIR_GROUP); workgroups beyond it exit
    const int *n_list; int list0;
    int repair_ksplit;       // K segments of k_srp_gemm_repair at most (shape-dependent, see REPAIR_KSPLIT_MAX)
    int repair_items;        // ... halved while the list gives more work items than this (repair_ksplit_eff; 0: never)
    // chunk-local scan result from the contraction's epilogue (256 x 384 kernel, one map, no gate, 32-row blocks = scan chunks):
    // part[arr][chunk][d] = sum_t scan_w[t] C[t][d] = the recursion E = 0.8f E + 0.2f C run from zero over the chunk's frames
    float *part; int *nvoiced; int D, n_chunks;
    long long part_plane_stride;   // a split-K launch leaves one set of results per K half (summed by k_scan_carry, half 0 first)
    float scan_w[32];
};

struct ScanPickArgs {
    const float *C;          // [c_planes][arrays][n_frames][Dp]: partial maps of a split-K contraction, summed here
    int c_planes; long long c_plane_stride;
    int n_frames, Dp, D, P, S, chunk, n_chunks;
    float mu, one_minus_mu;
    float inv_norm;          // correctly rounded 1 / (30 P): the normalisation (:155-156) divides by 30 P
    const float *state_in;   // [arrays][D]  E_prev at entry
    float *state_out;        // [arrays][D]  E_prev at exit
    float *part;             // [part_planes][arrays][n_chunks][D]  chunk-local recursion result (E from 0); 2 planes: the two K halves of the contraction
    int part_planes; long long part_plane_stride;
    int *nvoiced;            // [arrays][n_chunks]     voiced frames per chunk
    float *e_start;          // [arrays][n_chunks][D]  E at the start of every chunk
    const unsigned char *voiced;   // [arrays][n_frames] 1 = frame passed the power gate; NULL = ungated (all frames)
    const float *grid;       // [D] doaIdx2angle
    int *doa_bin; float *doa_rad; float *prob; float *energy;
    // adaptive SRP precision (fp16 coarse scan + exact repair of the frames whose pick is sensitive to the fp16 error)
    int mode;                // 0: plain; 1: coarse pass, flags the sensitive frames and plans their repair
    float tau;               // two normalised energies closer than this cannot be ordered from the coarse map
    unsigned char *flags;    // [arrays][n_frames] 1 = the frame's pick must be repeated on exact rows
    int groups_per_array;    // repair units (REPAIR_GROUP frames) per array
    int *need;               // [arrays * groups_per_array] test-and-set: the group is on the list (cleared by k_repair_patch)
    int *list;               // [arrays * groups_per_array] groups whose rows are recomputed, in order of arrival
    int *n_list;             // [1] their number (reset by k_scan_carry)
    int *chunk_from;         // [arrays][n_chunks] chunk the second pick of this chunk restarts from, >= n_chunks: no flagged frame (reset by k_scan_repick)
    int *clist; int *n_clist;     // [arrays * n_chunks], [2] (the second word counts the workgroups of k_scan_repick that are done: the last one empties both lists): the chunks that hold a flagged frame, in order of arrival (k_scan_repick walks them; reset by k_scan_carry)
    int lookback;                 // > 0 (ungated calls): no k_scan_carry -- k_scan_pick composes the start value of its chunk from the chunk-local
                                  // results of the `lookback` chunks before it (0.8^32 per chunk: the fifth chunk back is 3e-16 of it) and stores it
    unsigned long long *probe;    // host-mapped [3] or NULL: k_scan_repick leaves stats[0], stats[1] and probe_seq there (api.hip: adapt_policy_begin)
    unsigned long long probe_seq;
    int *last_vchunk;        // [arrays] chunk of the array's last frame that advanced the recursion, -1 = none (k_scan_carry)
    unsigned long long *stats;    // [4] running totals: flagged frames, listed groups, candidate columns of the flagged frames, flagged frames that took every column
    const unsigned char *dead;    // [arrays][n_frames] or NULL: frames of exact zeros (StftPhatArgs::dead): their rows are not listed
    const unsigned char *unsure;  // [arrays][n_frames] or NULL: frames the coarse analysis could not vouch for (StftPhatArgs::unsure): flagged with their six successors
    // lazy tails (see HIST_FRAMES).  lazy: the call's last frame is not flagged for the state's sake; k_scan_pick leaves the coarse rows of
    // the last HIST_FRAMES frames in hist_C_out [arrays][HIST_FRAMES][Dp] and the energies in front of them in e_hist_out [arrays][D].
    // hist_valid: the previous call did so (hist_C_in, e_hist_in): a flagged frame t < REPAIR_WARM lists the history units its rows
    // t - REPAIR_WARM .. -1 live in (numbered from hist_base), and k_scan_repick starts chunk 0 from e_hist_in over the history rows
    // (chunk_from = -1).
    int lazy, hist_valid, hist_base;
    float *hist_C_out, *e_hist_out;
    const float *hist_C_in, *e_hist_in;
    // candidate columns (round 5): per repair unit the delays its rows are needed at, one bit per column, umask_words = Dp / 32 words
    // per unit (k_scan_pick ORs a flagged frame's candidate columns into every unit it lists; k_srp_cand -- one workgroup per unit -- takes
    // and clears them and releases the unit's test-and-set word).  NULL: whole rows (k_srp_gemm_repair + k_repair_patch).
    unsigned *umask; int umask_words;
    // two work lists (round 6; k_scan_pick<PL, 2>: contexts whose flagged frames include whole rows by construction -- eager tails, the
    // 16-microphone unsure marks): a frame that takes every column lists its units HERE (test-and-set words, list, length; walked by a
    // second list-mode analysis + k_srp_gemm_repair + k_repair_patch), the others on `list` with their column masks.  NULL: one list.
    int *need_full, *list_full, *n_list_full;
};

// The repair contraction runs on however many rows the coarse pass listed (a device-side count): the K range is what
// parallelises.  It is cut into a number of segments that depends on the SHAPE of the call only (api.hip, repair_ksplit_for:
// 32 for a few arrays, 8 when the always-recomputed tails of many arrays make thousands of rows), never on how many rows were
// listed, so that an exact row's value does not depend on what was listed with it (a call worked off in chunks returns the
// same bits).  Partial maps: [ksplit][repair_plane_stride], summed in order by k_repair_patch.
constexpr int REPAIR_KSPLIT_MAX = 32;
__host__ __device__ inline long long repair_plane_stride(int n_rows, int Dp) { return (long long)((n_rows + 127) / 128 * 128) * Dp; }
__host__ __device__ inline long long repair_cx_rows(long long pass_rows, int ksplit) { return ((pass_rows + 127) / 128 * 128) * ksplit; }
// K segments the repair contraction really uses for n_rows listed rows: the call's shape gives the most (repair_ksplit_for: enough to
// fill the chip when only the tails are listed); a long list has the work items without cutting K that fine, and every halving
// halves the partial maps that are written and read back -- halved until at most items_max items (row tile, column tile, segment)
// are left.  The row count is the device's, so k_srp_gemm_repair and k_repair_patch both work it out from it.  (The exact rows' last
// bits then depend on how many rows a call lists -- as they already do on the call's shape; tie-level)
__host__ __device__ inline int repair_ksplit_eff(int ksplit, int n_rows, int Dp, int items_max)
{
    if (items_max <= 0) return ksplit;
    const int tiles = ((n_rows + 127) / 128) * (Dp == 64 ? 1 : Dp / 192);
    int k = ksplit;
    while (k > 4 && tiles * k > items_max) k >>= 1;
    return k;
}

struct RepairPatchArgs {
    const float *Cx;         // [repair_ksplit][repair_plane_stride] exact rows, split-K partial maps
    int pass_rows, col_tiles, ksplit, items;     // (ksplit, items: as GemmArgs::repair_ksplit, repair_items)
    const int *list; const int *n_list; int list0, groups_per_array;
    int *need;               // the groups' test-and-set words, released here
    float *C;                // [c_planes][arrays][n_frames][Dp]: plane 0 takes the exact row, the others zeros
    int c_planes; long long c_plane_stride;
    int n_frames, Dp;
    float *hist_C; int hist_base;   // lazy tails: the rows of the units >= hist_base go to hist_C [arrays][HIST_FRAMES][Dp] (one plane)
};

struct GateArgs {
    const float *power_lin;  // [arrays][n_frames] FFTPower of every frame
    int n_frames, fft_n, needed_samples;
    float margin_db;
    double eps;              // added to every frame's power before it is accumulated (FreqGCC: 1e-10, BinauralLocalisation.cpp:391)
    // persistent gate state per array: {accumulated power (double), samples consumed (double), floor dB (double), estimated (double 0/1)}
    double *state;           // [arrays][4]
    unsigned char *voiced;   // [arrays][n_frames]
    float *power_out;        // [arrays][n_frames] value handed to setDOA / compared with the floor (may be NULL)
    int *post0;              // [arrays] first frame of this call at which the floor estimate exists (the frame that completes it
                             // counts; 0 if it existed before the call, n_frames if it still does not); may be NULL
};

struct DoaFillArgs {
    const unsigned char *voiced;   // [arrays][n_frames]
    int n_frames, S;
    int *doa_bin; float *doa_rad; float *prob;      // [arrays][n_frames][S], filled forward over gated-out frames
    int *last_bin; float *last_rad; float *last_prob;   // [arrays][S] _currentDOA / _prob carried between calls
};

struct BeamformArgs {
    const float *pcm;
    long long array_stride, mic_stride;
    int M, Mpad, S, n_frames, ft, fs;
    int nb;                  // frames per inverse-FFT batch (<= BF_NB; fewer with several sources so that two workgroups still share a CU)
    const float *window;
    const double *mic_x;     // [M] x coordinate (Beamformer.cpp:59 steers with x only)
    const float *doa_rad;    // [arrays][n_frames][S]
    float *out;              // [arrays][S][n_frames*hop]
    const float *tail_in;    // [arrays][S][hop] overlap-add carry at entry
    float *tail_out;         // at exit
    // any-N kernel only (kernels_generic.hip)
    int N, logH;
    const float2 *tw;        // [N/2] exp(-j 2 pi i / N)
    int S_all, s0;           // ... which takes the sources s0 ... s0 + S - 1 of S_all per launch (all at once while M + S spectra fit the LDS)
};

// wave-per-run delay-and-sum on the 1024-point complex transform (kernels_wave.hip): one source, any M <= 16
struct BeamformWaveArgs {
    const float *pcm;
    long long array_stride, mic_stride;
    int M, n_pairs, n_frames, ft, S;   // S sources: blockIdx.z, each steered by its own column of doa_bin
    const float *window;     // [1024] periodic Hann
    const int *doa_bin;      // [arrays][n_frames][S] grid index of the frame's DOA, -1: the initial _currentDOA = 0
    const float2 *table;     // [D + 1][n_pairs][1024] (P_a - j P_b) / (M N) per steering angle (k_bf_table)
    float *out;              // [arrays][S][n_frames*hop]
    const float *tail_in;    // [arrays][S][hop] overlap-add carry at entry
    float *tail_out;         // at exit
    int skew;                // k_beamform_wave with the hand-off, > 0 (round 5, as StftPhatArgs::skew): grid (arrays, workgroups, sources); the first half
                             // of an array's workgroups -- dispatched first, the older workgroup of every CU -- run ft + skew frames per wave, the others ft - skew
};

struct Gcc2ScanArgs {
    const float *C;          // [c_planes][arrays][n_frames][Dp] un-smoothed GCC-PHAT R_t[d] (split-K partial maps, summed here)
    int c_planes; long long c_plane_stride;
    int n_frames, Dp, D, chunk;
    const long long *vdone_in; long long *vdone_out;   // [arrays] voiced frames processed before / after this call (0 = the stream's first)
    const int *vidx;         // [arrays][n_frames] frames that passed the gate, in order (NULL: ungated, every frame)
    const int *nv;           // [arrays] their number (NULL: n_frames)
    const unsigned char *vreset;   // [arrays][n_frames] per fired frame: 1 = both memory factors are zero at this frame (the
                             // silence rule, BinauralLocalisation.cpp:530-560); NULL (ungated): only the stream's first frame
    float mu, one_minus_mu;  // _maxCorrMemoryFactor 0.8f and 1 - 0.8f (float arithmetic)
    float doa_mem, one_minus_doa_mem;   // _maxDoaMemoryFactor 0.6f
    float step;              // _doaStep
    const float *corr_in; float *corr_out;      // [arrays][D] _prevCorrelationsReal
    const float *doa_in; float *doa_out;        // [arrays] _currentDOA
    const float *grid;       // [D]
    int *argmax; float *doa_rad; float *prob; float *corr;
};

struct Gcc2FillArgs {
    const unsigned char *voiced;   // [arrays][n_frames]
    int n_frames, D;
    int *argmax; float *doa_rad, *prob, *corr;          // outputs of the scan (corr may be NULL); gated-out frames are filled here
    const float *corr_state;       // [arrays][D] smoothed correlation at the start of the call
    int *last_idx; float *last_rad, *last_prob;          // [arrays] values of the last frame, carried to the next call
};

struct MaskParams {
    float thr[45];
    int lo[45], hi[45];       // support of band b (bins with H_b > 0), lo > hi for an empty band
    int kb[513];              // first band covering bin k (the second one is kb+1), -1 if none
    float kw0[513], kw1[513]; // H_kb[k], H_{kb+1}[k]
    float lambda, one_minus_lambda, reject, rho;
    int method, alg;
};

struct MaskArgs {
    const float *pcm;
    long long stream_stride, ch_stride;
    int n_frames, ft;
    long long frames_done;
    const float *window;
    const MaskParams *mp;
    const float *Q_in; float *Q_out; float *noise;      // [streams][45]
    const float *tail_in; float *tail_out;              // [streams][2][512]
    float *out; int *decisions;
};

// any-length variant: the per-bin band tables live in global memory (length K = N/2 + 1)
struct MaskGenArgs {
    MaskArgs a;
    int N, logH;
    const float2 *tw;         // [N/2]
    const float2 *kw;         // [K] (H_kb[k], H_{kb+1}[k])
    const int *kb;            // [K] first band covering bin k, -1 if none
};

struct MaskFrameArgs {
    const double *L, *R;      // [K] complex (CCS)
    double *outL, *outR;      // [K] complex, zeroed by the caller
    const double *H;          // [45][K] filter magnitudes
    const double *thr;        // [45]
    double *Q, *noise;        // [45] state
    int K, method, alg, first_call;
    double lambda, one_minus_lambda, reject, rho;
    int *decisions;
};

// ---- MultibandBinarualLocalisation (kernels_multiband.hip) ----
struct MbAnalyseArgs {
    const float *pcm;
    long long array_stride, ch_stride;
    int n_frames, N, logH, nbins, D;
    const float *window;
    const float2 *tw;         // [N/2]
    const float *coef;        // [nbins][K] filter magnitudes (0 outside the support)
    const int *lo, *hi;       // [nbins] support of band b (bins with H_b > 0, inclusive), lo > hi if empty
    const float2 *T;          // [K][D] exp(+j 2 pi k tau_d / N)
    float *raw;               // [arrays][n_frames][nbins * D] un-smoothed band correlations
    float *band_energy;       // [arrays][n_frames][nbins] FFTPower of the sub-band frames
    float *p_full, *p_half;   // [arrays][n_frames] FFTPower(frames, N+2) and FFTPower(frames, (N+2)/2)
};

struct MbScanArgs {
    const float *raw, *band_energy;
    int n_frames, nbins, D, chunk;
    float mem, one_minus_mem;               // _corrMemoryFactor 0.4f and 1 - 0.4f (float arithmetic)
    const float *corr_in; float *corr_out;  // [arrays][nbins * D] _prevCorrelationsReal
    int *hist_idx; float *hist_prob;        // [arrays][n_frames] argmax of _energyInDOA and its share
    int *band_idx; float *energy_in_doa; float *band_corr;   // optional outputs
};

struct MbSummaryArgs {
    const float *p_full, *p_half;
    const int *hist_idx; const float *hist_prob;
    int n_frames, K, needed_samples, use_floor;
    float margin_db;
    const float *grid;        // [D]
    double *gate;             // [arrays][4] {_powerFloor, _samplesConsumedForNoise, floor dB, _noiseEstimated}
    float *cur;               // [arrays][2] {_currentDOA[0], _prob[0]}
    float *doa_rad, *prob, *power; unsigned char *voiced;   // [arrays][n_frames]
};

// ---- MVDR-style beamformer with a per-bin spatial covariance (BASELINE configs[3]) ----
struct MvdrAnalyseArgs {
    const float *pcm;
    long long stream_stride, mic_stride;
    int n_frames, N, logH, M;
    const float *window;
    const float2 *tw;         // [N/2]
    const float *doa_rad;     // [streams][n_frames]
    float2 *X;                // [streams][n_frames][K][M] one-sided spectra, microphone fastest
    // steering phasors of every (stream, frame, microphone), factored: d_m[k] = T[k >> 5] * T[nhi + (k & 31)] with
    // T[i < nhi] = exp(-j 2 pi 32 i u), T[nhi + i] = exp(-j 2 pi i u), u = fs/N/c x_m cos(DOA + pi/2) (Beamformer.cpp:59);
    // the phase is reduced in double.  N/64 + 33 sincos per microphone and frame instead of N/2 + 1.
    float2 *T;                // [streams][n_frames][M][nhi + 32], nhi = N/64 + 1
    const double *mic_x;      // [M]
    double unit;              // fs / N / 346.1
};

struct MvdrSolveArgs {
    // A launch covers the problems pid0 .. pid0 + n_prob - 1 (problem = stream * K + bin), `pieces` workgroups per 64 problems:
    // piece j solves the frames [j F / pieces, (j + 1) F / pieces) after running the covariance recursion alone -- the same
    // operations in the same order, so the same bits -- over the frames before them; the last piece stores the state.
    long long pid0, n_prob;
    int pieces;
    const float2 *X;          // [streams][n_frames][K][M]
    const float2 *T;          // [streams][n_frames][M][nhi + 32] factored steering phasors (MvdrAnalyseArgs)
    int n_streams, n_frames, K, M;
    float alpha, one_minus_alpha, loading_over_m;
    float2 *phi;              // [streams][K][M(M+1)/2] lower triangle of the covariance, row-major (state at entry)
    float *trace;             // [streams][K] tr(Phi), carried as its own recursion
    // state at exit: problem pid goes to phi_out[(pid - out_base) tri], trace_out[pid - out_base].  The same buffers for an unsplit
    // launch (every problem is read and written by one workgroup); a launch cut into pieces writes to a scratch copy, because
    // every piece reads the entry state when it starts and nothing orders that before the last piece's stores
    float2 *phi_out; float *trace_out; long long out_base;
    float2 *Y;                // [streams][n_frames][K] beamformed spectrum
};

struct MvdrSynthArgs {
    const float2 *Y;          // [streams][n_frames][K]
    int n_frames, N, logH, ft;
    const float2 *tw;
    const float *tail_in; float *tail_out;   // [streams][N/2] overlap-add carry
    float *out;               // [streams][n_frames * N/2]
};

}  // namespace mca
