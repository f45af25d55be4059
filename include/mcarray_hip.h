/*
 * mcarray_hip.h -- C ABI of libmcarray_hip.so: the MI355X (gfx950) implementation of
 * mcarray's per-frame localisation + beamforming hot path.
 *
 * This is the drop-in boundary.  Plain pointers and sizes only; no C++
 * or torch types cross it.  Every entry point names the reference interface it
 * replaces (paths relative to the reference root, jordi-adell/mcarray v0.3.0-alpha).
 * The C++ classes in include/mcarray/ (same names and signatures as the reference's)
 * are thin callers of these functions; INTEGRATION.md shows the binding a reference
 * maintainer would add.
 *
 * Conventions
 *   - return value: 0 = MCA_HIP_OK, < 0 = error (mca_hip_status); the message is
 *     available from mca_hip_last_error().  The C++ wrappers rethrow it as
 *     mca::MCArrayException (include/mcarray/mcarray_exception.h:51 in the reference).
 *   - "_dev" pointers are device (HBM) pointers, work is enqueued on `stream`
 *     (a hipStream_t passed as void*; NULL = the null stream) and the call does
 *     not synchronise.  Functions without "_dev" pointers take host pointers and
 *     return after the result is on the host.
 *   - A context is stateful like the reference's module objects (E_prev of
 *     SteeringBeamforming.h:69, current DOA, overlap-add tails) and is not
 *     thread-safe; use one context per stream of arrays (SURVEY 8b "Ownership").
 *   - PCM layout (stream API): fp32, channel-major: sample n of microphone m of
 *     array a at pcm[a*array_stride + m*mic_stride + n]; a call that processes F
 *     frames reads (F+1)*hop samples per microphone (frame t = samples
 *     [t*hop, t*hop+N), hop = N/2, periodic Hann analysis window).
 *   - Spectrum layout (frame API): the reference's CCS layout, double[N+2], bin k at
 *     [2k],[2k+1], k = 0..N/2 (Beamformer.cpp:59, test_mcarray.cpp:662).
 */
#ifndef MCARRAY_HIP_H
#define MCARRAY_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    MCA_HIP_OK = 0,
    MCA_HIP_ERR_INVALID_ARGUMENT = -1,
    MCA_HIP_ERR_HIP = -2,            /* a HIP runtime call failed */
    MCA_HIP_ERR_OUT_OF_MEMORY = -3,
    MCA_HIP_ERR_UNSUPPORTED = -4,    /* configuration outside what the kernels cover */
    MCA_HIP_ERR_NO_DEVICE = -5       /* no gfx950 device visible: there is NO CPU fallback */
} mca_hip_status;

typedef enum {
    MCA_HIP_SRP_FP32 = 0,    /* v_mfma_f32_32x32x2_f32, exact fp32 (parity anchor) */
    MCA_HIP_SRP_FP16X3 = 1,  /* fp16 hi/lo split operands, 3 MFMAs per k-step, ~fp32 accuracy */
    MCA_HIP_SRP_FP16 = 2,    /* single fp16 MFMA per k-step (fast; energy map error ~1.5e-5 of the peak) */
    MCA_HIP_SRP_ADAPTIVE = 3 /* fp16 coarse scan of every frame + exact repair: the frames whose peak pick is sensitive to the
                                fp16 error (and the rows their energy depends on) are recomputed with the FP16X3 split and
                                picked again, so the DOA bins are those of FP16X3 at about the cost of FP16.  Applies to large
                                batches (>= 4096 frames per call) of 1024-sample frames -- and of 2048- or 512-sample frames
                                with up to 8 microphones -- with more than two microphones and
                                ONE source (with several, the S-th pick is a near tie too often), with or without the power
                                gate; every other call of such a context runs as FP16X3 -- as do its calls while most rows
                                need the repair (noise only, silence: the context backs off by itself and probes again
                                later; mca_hip_config.adaptive_fallback = OFF pins the mode).  The optional energy map
                                keeps fp16 accuracy on unrepaired frames.
                                GUARANTEE: an unflagged frame carries the picks of the exact (FP16X3) map under the error model of
                                the sensitivity test; a flagged frame is picked on energies recomputed exactly for its own row and
                                the 16 rows before it (0.8^17 of the coarse error remains: within ~4e-7 of the map's peak of
                                FP16X3's energies).  Picks that are TIES at that level -- two candidates, or a first difference
                                against zero, closer than 1e-6 of the row's largest normalised energy: the bar of the parity tests
                                -- may resolve differently from FP16X3 and between calls of different shapes.  Measured over 40
                                random configurations / 696 320 frames: 41 picks on 29 frames differ from FP16X3, all 29 such ties
                                (profiles/r05_adaptive_check.json; the warm-row count of 16 against 24: r04_adaptive_check_warm24.json).
                                ROUND 5, device-pointer calls of 4 / 8-microphone contexts without the gate: (a) LAZY TAILS -- the
                                call does not repeat its last frame for the state's sake; it keeps its last 16 frames of PCM, their
                                coarse rows and the energies in front of them, and the next call repairs them only if one of its
                                first 16 frames is flagged.  Every other consumer of the state (host-pointer calls, graph launches,
                                mca_hip_state_save, calls of another batch size or too small for the mode) first makes the carried
                                energies exact, so what they see is what the eager form left.  (b) CANDIDATE COLUMNS (AUTO policy,
                                one source) -- a flagged frame's rows are recomputed exactly at the delays its pick can be among
                                (those whose coarse energy reaches the lowest value the exact pick can have, and two either side),
                                not at all D; the second pick runs on rows that are exact there and coarse elsewhere, and the
                                optional energy map of a flagged frame is exact at those delays only.
                                MCA_HIP_ADAPT_LAZY=0 restores the eager, whole-row form.
                                OUTSIDE THE MODEL, in any precision: a frame in which a channel's DC or Nyquist bin -- the two real
                                bins -- is at the rounding level of an fp32 transform (about one frame in 10^5 per 8 channels).
                                PHAT keeps only the SIGN of such a bin, and no two implementations, the reference's double-precision
                                one included, need agree on it; the frame's normalised energies then differ by up to
                                2 (M - 1) / (30 P) x 0.2 between them, decaying 0.8 per frame.  Inside this mode the coarse and the
                                exact pass agree on such a bin: with 4 and 8 microphones both run the same transform; with 16 they
                                are two kernels, so the coarse one marks the frame and it is repaired with its six successors
                                whatever the map says (profiles/r04_case23_real_bin_at_rounding_level.log). */
} mca_hip_srp_precision;

/* Weighting of the generalised cross-correlation inside dsp::GeneralisedCrossCorrelation::calculateCorrelationsForPrecomputedTauMatrix
 * (call site SteeringBeamforming.cpp:115-119; the code itself is in DSPONE, which the reference does not vendor).
 * PHAT is the build's reading and what BASELINE.json's north_star names.  NONE -- the plain cross-spectrum
 * X_a conj X_b -- is the reading under which the reference's own SRP test stimulus (1 kHz sines, test_mcarray.cpp:384-423)
 * localises within its 7 degrees; it is offered so that a maintainer who has DSPONE can switch once
 * tools/pin_against_dspone.cpp has settled which one DSPONE implements.  NONE needs MCA_HIP_SRP_FP32 (un-normalised spectra
 * do not fit fp16 operands): frame API always, stream API at N = 1024 with 4 or 8 microphones. */
typedef enum { MCA_HIP_GCC_PHAT = 0, MCA_HIP_GCC_NONE = 1 } mca_hip_gcc_weighting;

typedef struct mca_hip_ctx mca_hip_ctx;

/* Configuration = the constructor arguments of the reference's modules
 * (SourceSeparationAndLocalisation.h:47, BeamformingSeparationAndLocalistaion.h:40,
 * SteeringBeamforming.h:43, Beamformer.h:39) plus the constants the reference
 * hard-codes and BASELINE.json needs as parameters (SURVEY section 5 "config"). */
typedef struct {
    int struct_size;           /* sizeof(mca_hip_config), for ABI evolution */
    int device;                /* HIP device ordinal */
    int sample_rate;           /* Hz */
    int fft_size;              /* N; the reference derives it with calculateOrderFromSampleRate
                                  (SourceSeparationAndLocalisation.cpp:52).  Stream API: N = 1024 and N = 512 (up to 8
                                  microphones) run the tuned kernels, any other power of two 64..8192 the any-length kernels as long as
                                  (n_mics + n_sources) spectra of N/2+1 bins fit the 160 KiB LDS; the frame API
                                  takes any even N with N/2+1 <= 4097 */
    int n_mics;                /* M, 2..16 */
    const double *mic_xyz;     /* [M][3] metres, ArrayDescription coordinates (ArrayDescription.h:31-92) */
    double doa_step_deg;       /* SteeringBeamforming.cpp:39 hard-codes 5.0; BASELINE uses 0.5 */
    int n_sources;             /* numOfSources, 1..4 (a limit of this build -- the reference's loops, SteeringBeamforming.cpp:185-194, take any count;
                                  more is refused with MCA_HIP_ERR_INVALID_ARGUMENT, never truncated) */
    int use_power_floor;       /* usePowerFloor (reference default true, SourceSeparationAndLocalisation.h:47): the stream API
                                  then runs the power gate of BeamformingSeparationAndLocalisation.cpp:55-87 on the GPU;
                                  the frame API exposes mca_hip_fft_log_power for the caller's gate */
    int srp_precision;         /* mca_hip_srp_precision */
    int max_arrays;            /* number of independent arrays whose state the context holds (>= 1) */
    int gcc_weighting;         /* mca_hip_gcc_weighting; 0 = PHAT.  (Appended in round 3: a struct_size that ends before this
                                  field is accepted and means PHAT.) */
    /* Appended in round 4 (a struct_size that ends before them is accepted; zero = the default of each): */
    int adaptive_fallback;     /* mca_hip_adaptive_fallback.  MCA_HIP_SRP_ADAPTIVE only.  0 = AUTO: the context backs off to
                                  FP16X3 by itself while most rows need the repair (noise only, silence) and probes again later.
                                  Round 6: a call's report is consumed by the eligible call TWO calls after it, which waits for it
                                  if it has not arrived (the call in between is still queued behind it: the device does not idle),
                                  so which call switches depends on the sequence of calls and their content only -- two runs of the
                                  same calls return the same bits in every output under AUTO as well (tests/test_gpu_adaptive.py).
                                  A report that stays away for 4 s (a stream held up by something only the calling thread would
                                  release) ends the policy until mca_hip_reset.  HIP-graph recordings never switch.
                                  1 = OFF: the mode is pinned -- every eligible call runs coarse + repair. */
    int adaptive_min_rows;     /* ADAPTIVE: calls of fewer rows (arrays x frames) run as FP16X3; 0 = 4096 */
    int adaptive_max_sources;  /* ADAPTIVE: contexts with more sources run as FP16X3; 0 = 1 */
    int scan_carry;            /* 1: the chunk start values of the energy recursion always come from the serial carry pass
                                  (k_scan_carry) instead of the four-chunk look-back of ungated PHAT calls; results agree to the
                                  last bits (tests cross-check the two) */
} mca_hip_config;

typedef enum { MCA_HIP_ADAPT_FALLBACK_AUTO = 0, MCA_HIP_ADAPT_FALLBACK_OFF = 1 } mca_hip_adaptive_fallback;

/* ---- page-locked host memory ---------------------------------------------- */
/* Thin wrappers over hipHostMalloc / hipHostFree / hipHostRegister / hipHostUnregister so that a caller of the host-pointer
 * entry points need not link HIP.  A registered range must stay allocated until it is unregistered.  The reference has no
 * counterpart (its SignalVector buffers are plain new[] arrays, mcadefs.h:86-88). */
void *mca_hip_host_alloc(long long bytes);
void mca_hip_host_free(void *p);
int mca_hip_host_register(void *p, long long bytes);
int mca_hip_host_unregister(void *p);

/* ---- lifetime ------------------------------------------------------------ */
/* Replaces the constructors SteeringBeamforming::SteeringBeamforming + generateLookupTable
 * (SteeringBeamforming.cpp:34-94), Beamformer::Beamformer (Beamformer.cpp:33-49) and
 * BeamformingSeparationAndLocalisation's (BeamformingSeparationAndLocalisation.cpp:29-53). */
int mca_hip_create(const mca_hip_config *cfg, mca_hip_ctx **out);
void mca_hip_destroy(mca_hip_ctx *ctx);
/* message of the last failure on this context (ctx == NULL: last failed create) */
const char *mca_hip_last_error(const mca_hip_ctx *ctx);

/* ---- introspection --------------------------------------------------------- */
int mca_hip_num_steps(const mca_hip_ctx *ctx);   /* D = _numSteps (SteeringBeamforming.cpp:40) */
int mca_hip_num_pairs(const mca_hip_ctx *ctx);   /* P = M(M-1)/2 */
int mca_hip_num_groups(const mca_hip_ctx *ctx);  /* pairs sharing a bit-identical delay table are merged */
/* delaysForMicroPair (SteeringBeamforming.cpp:69-73): out[P][D], the float delays in samples */
int mca_hip_get_pair_delays(const mca_hip_ctx *ctx, float *out);
/* doaIdx2angle for every grid point (microhponeArrayHelpers.cpp:117-120): out[D] radians */
int mca_hip_get_doa_grid(const mca_hip_ctx *ctx, float *out);

/* ---- state ----------------------------------------------------------------- */
/* zero E_prev / DOA / overlap-add tails of every array (a freshly constructed module) */
int mca_hip_reset(mca_hip_ctx *ctx, void *stream);
/* pre-size the internal workspace so later *_dev calls allocate nothing (graph capture) */
int mca_hip_reserve(mca_hip_ctx *ctx, int n_arrays, int n_frames);
/* Checkpoint / resume of everything a module object carries between process() calls, for all max_arrays
 * arrays: E_prev (SteeringBeamforming.h:69), the overlap-add tails, the power-floor estimation
 * (SoundLocalisationImpl.h:84-86), _currentDOA / _prob (BeamformingSeparationAndLocalisation.cpp:51-52), the 2-mic
 * path's smoothed DOA and frame count, the frame API's E_prev.  A blob is only valid for a context created with
 * the same geometry, grid, frame length, n_sources and max_arrays (checked: MCA_HIP_ERR_INVALID_ARGUMENT).
 * mca_hip_state_size returns the number of bytes (or a negative status). */
long long mca_hip_state_size(const mca_hip_ctx *ctx);
int mca_hip_state_save(mca_hip_ctx *ctx, void *blob, long long blob_bytes);
int mca_hip_state_load(mca_hip_ctx *ctx, const void *blob, long long blob_bytes);

/* ---- stream API: batched frames, device pointers ---------------------------- */
/* STFT analysis + SteeringBeamforming::processFrame (SteeringBeamforming.cpp:96-195) for
 * n_frames consecutive frames of n_arrays independent arrays: GCC-PHAT over all pairs,
 * SRP scan, 0.8 IIR over frames (continuing from the context state), selectDOA.
 * Outputs (any may be NULL except doa_bin_dev):
 *   doa_bin_dev [A][F][S] int32   maxIdx+1 of selectDOA (:191) -- the "DOA bin"
 *   doa_rad_dev [A][F][S] float   doaIdx2angle(maxIdx+1) (:191), radians
 *   prob_dev    [A][F][S] float   the max (:193)
 *   energy_dev  [A][F][D] float   un-normalised smoothed _energyInDOA (before :155) */
int mca_hip_localise_frames_dev(mca_hip_ctx *ctx, const float *pcm_dev, long long array_stride,
                                long long mic_stride, int n_arrays, int n_frames,
                                int *doa_bin_dev, float *doa_rad_dev, float *prob_dev,
                                float *energy_dev, void *stream);

/* STFT analysis + BeamformingSeparationAndLocalisation::processFrameSeparation
 * (BeamformingSeparationAndLocalisation.cpp:103-119 -> Beamformer::processFrame,
 * Beamformer.cpp:51-71) + inverse FFT and overlap-add, steering frame t of array a at
 * doa_rad_dev[a][t][s].  out_pcm_dev [A][S][F*hop] fp32. */
int mca_hip_separate_frames_dev(mca_hip_ctx *ctx, const float *pcm_dev, long long array_stride,
                                long long mic_stride, int n_arrays, int n_frames,
                                const float *doa_rad_dev, float *out_pcm_dev, void *stream);

/* The same with the grid bins behind the angles: doa_bin_dev[a][t][s] as written by
 * mca_hip_localise_frames_dev (-1 = no frame has fired yet, the initial _currentDOA = 0,
 * BeamformingSeparationAndLocalisation.cpp:51), doa_rad_dev = the grid angles of those bins.  With
 * one or two sources on the 1024-sample path the steering phasors then come from a per-angle table built
 * once per context (no sincos per frame); other configurations run as mca_hip_separate_frames_dev. */
int mca_hip_separate_frames_bins_dev(mca_hip_ctx *ctx, const float *pcm_dev, long long array_stride,
                                     long long mic_stride, int n_arrays, int n_frames,
                                     const int *doa_bin_dev, const float *doa_rad_dev,
                                     float *out_pcm_dev, void *stream);

/* Both of the above in sequence = SourceSeparationAndLocalisation::processParametrisation
 * (SourceSeparationAndLocalisation.cpp:79-94) for every frame. */
int mca_hip_process_frames_dev(mca_hip_ctx *ctx, const float *pcm_dev, long long array_stride,
                               long long mic_stride, int n_arrays, int n_frames,
                               int *doa_bin_dev, float *doa_rad_dev, float *prob_dev,
                               float *energy_dev, float *out_pcm_dev, void *stream);

/* The same for 16-bit PCM, the sample type of the reference's process(std::vector<int16_t*>&, ...) overloads and of the
 * WAV / raw files its tools read (mcabeamf.cpp:112, test_mcarray.cpp:618).  pcm is [A][M][(F+1)*hop] int16; the samples
 * are taken at face value (+-32768, like a cast to double: PHAT and the DOA do not depend on the scale, the audio out is
 * in the same units).  Half the bytes cross PCIe; the conversion to fp32 runs on the GPU. */
int mca_hip_process_frames_host_i16(mca_hip_ctx *ctx, const short *pcm, int n_arrays, int n_frames,
                                    int *doa_bin, float *doa_rad, float *prob, float *energy, float *out_pcm);

/* After a stream call on a context with use_power_floor = 1: copies, for the frames of that call,
 * voiced[A][F] (1 where processFrameLocalisation passed the gate and the callback fires,
 * BeamformingSeparationAndLocalisation.cpp:87-94) and power[A][F] (the value handed to setDOA).  Either may be NULL.
 * Gated-out frames repeat the previous _currentDOA/_prob in the DOA outputs (initially 0 rad / -1, bin -1). */
int mca_hip_copy_gate(mca_hip_ctx *ctx, unsigned char *voiced, float *power);

/* Host-buffer variant of mca_hip_process_frames_dev (copies in, runs, copies out, synchronises);
 * pcm is [A][M][(F+1)*hop] contiguous; outputs as above, any of doa_rad/prob/energy/out_pcm may be NULL.
 * This is what a drop-in process() caller hits (src/programs/mcabeamf.cpp:101-112, test/test_mcarray.cpp:869,937).
 * If pcm is page-locked (mca_hip_host_alloc / mca_hip_host_register below, or hipHostMalloc / hipHostRegister), the arrays
 * go up in up to four chunks and the upload of chunk i+1, the kernels of chunk i and the download of chunk i-1 (into
 * page-locked result buffers) overlap; pageable buffers take one synchronous copy each way.  Same results either way. */
int mca_hip_process_frames_host(mca_hip_ctx *ctx, const float *pcm, int n_arrays, int n_frames,
                                int *doa_bin, float *doa_rad, float *prob, float *energy, float *out_pcm);

/* ---- frame API: one frame of CCS spectra, host pointers, double precision ------ */
/* SteeringBeamforming::processFrame(const SignalVector&, SignalPtr DOA, SignalPtr prob,
 * int numOfSources, SignalVector& wienerCoefs) (SteeringBeamforming.h:54).  frames[c] ->
 * double[ccs_len]; DOA[S] radians, prob[S]; doa_bin (may be NULL) [S].  State of array 0. */
int mca_hip_steering_process_frame(mca_hip_ctx *ctx, const double *const *frames, int ccs_len,
                                   double *DOA, double *prob, int *doa_bin, int n_sources);
/* Beamformer::processFrame(SignalVector&, SignalPtr outputFrame, double DOA) (Beamformer.h:49) */
int mca_hip_beamformer_process_frame(mca_hip_ctx *ctx, const double *const *frames, int ccs_len,
                                     double *out, double DOA);
/* dsp::SignalPower::FFTLogPower as used by the power gate
 * (BeamformingSeparationAndLocalisation.cpp:83); *power_db = 10 log10(mean-square) */
int mca_hip_fft_log_power(mca_hip_ctx *ctx, const double *const *frames, int ccs_len, double *power_db);
/* copy of the current un-normalised _prevEnergyInDOA of array 0: out[D] */
int mca_hip_get_energy(mca_hip_ctx *ctx, double *out);

/* ---- real-time mode: the stream call as a HIP graph --------------------------------------------
 * A caller that hands over a stream chunk by chunk (SourceSeparationAndLocalisation::process() on a live input,
 * mcabeamf.cpp:112; BASELINE configs[1]) is bound by the launches of a call, not by its kernels: one frame is 18 KB.
 * mca_hip_graph_create fixes the shape and the device buffers of a mca_hip_process_frames_dev call (out_pcm_dev NULL:
 * of a mca_hip_localise_frames_dev call; doa_bin_dev NULL: of a mca_hip_separate_frames_dev call, the delay-and-sum stage
 * alone at the caller's angles in doa_rad_dev); mca_hip_graph_launch replays the kernels of that call as ONE graph launch on
 * whatever the caller has put into pcm_dev since the last launch, continuing the context state exactly like the plain
 * call (results are bit-identical).  The graphs (one per parity of the double-buffered state) are recorded on first
 * use; recording executes nothing.  Plain stream calls and graph launches on the same context may be mixed. */
typedef struct mca_hip_graph mca_hip_graph;
int mca_hip_graph_create(mca_hip_ctx *ctx, const float *pcm_dev, long long array_stride, long long mic_stride,
                         int n_arrays, int n_frames, int *doa_bin_dev, float *doa_rad_dev, float *prob_dev,
                         float *energy_dev, float *out_pcm_dev, mca_hip_graph **out);
int mca_hip_graph_launch(mca_hip_graph *g, void *stream);
void mca_hip_graph_destroy(mca_hip_graph *g);

/* ---- 2-microphone GCC-PHAT localisation (FreqGCCBinauralLocalisation) ----------------- */
/* Deterministic part of FreqGCCBinauralLocalisation::processParametrisation
 * (BinauralLocalisation.cpp:406-567) for n_frames consecutive frames of n_arrays independent
 * 2-microphone arrays, on a context created with n_mics == 2 (the reference grid is
 * doa_step_deg = 3, BinauralLocalisation.cpp:328): GCC-PHAT at the D steering delays (:438-444),
 * correlation smoothing corr = (1-mu) corr + mu prev with mu = 0 on a stream's first frame and
 * 0.8f afterwards (:445-448, :523), first-max argmax and the author's deterministic DOA smoothing
 * DOA = m DOA + (1-m) angle, m = 0 then 0.6f (the #else branch :502-504; the particle filter of
 * :456-473 is a stochastic DSPONE component and stays out of scope), and setProbability of the
 * previous DOA (:454, :569-631).  With use_power_floor = 1 the gate of :387-404 / :425-434 runs on the GPU (3 s of floor
 * estimation, then a frame fires when its FFTLogPower exceeds the floor + 6 dB): the recursions only see the frames that
 * fired, the others repeat the outputs of the last fired frame (argmax -1, DOA 0, prob -1 before the first), and
 * mca_hip_copy_gate returns voiced[A][F] / power[A][F] of the call.
 *   argmax_dev [A][F] int32, doa_rad_dev [A][F] float (smoothed), prob_dev [A][F] float,
 *   corr_dev [A][F][D] float smoothed correlation (any but argmax_dev may be NULL). */
int mca_hip_gcc2_frames_dev(mca_hip_ctx *ctx, const float *pcm_dev, long long array_stride,
                            long long mic_stride, int n_arrays, int n_frames, int *argmax_dev,
                            float *doa_rad_dev, float *prob_dev, float *corr_dev, void *stream);
int mca_hip_gcc2_frames_host(mca_hip_ctx *ctx, const float *pcm, int n_arrays, int n_frames,
                             int *argmax, float *doa_rad, float *prob, float *corr);

/* ---- binaural masking (FastBinauralMasking) --------------------------------------------- */
typedef struct mca_hip_mask_ctx mca_hip_mask_ctx;
/* BinauralMasking::MaskingMethod / MaskingAlg (ArrayModules.h:81,89) */
typedef enum { MCA_HIP_MASK_FACTOR = 0, MCA_HIP_MASK_RELATIVE = 1, MCA_HIP_MASK_FULL = 3, MCA_HIP_MASK_NOISY = 4, MCA_HIP_MASK_NOTHING = 5 } mca_hip_mask_method;
typedef enum { MCA_HIP_MASK_BOTH = 0, MCA_HIP_MASK_SPATIAL = 1, MCA_HIP_MASK_TEMPORAL = 2 } mca_hip_mask_alg;
/* constructor arguments of FastBinauralMasking(int samplerate, double microDistance, float lowFreq,
 * float highFreq, MaskingMethod, MaskingAlg) (FastBinauralMasking.h:71-76) */
typedef struct {
    int struct_size;
    int device;
    int sample_rate;
    int fft_size;            /* N = 2^calculateOrderFromSampleRate(fs, 0.050): 1024 at 16 kHz (tuned kernel), 2048 at 44.1 / 48 kHz;
                                the stream API takes powers of two up to 8192, the frame hook any even N */
    double micro_distance;
    float low_freq, high_freq;
    int method;              /* mca_hip_mask_method */
    int algorithm;           /* mca_hip_mask_alg */
    int max_streams;
} mca_hip_mask_config;
/* FastBinauralMasking::FastBinauralMasking + init + calculateThresholds (FastBinauralMasking.cpp:51-123, :342-366) */
int mca_hip_mask_create(const mca_hip_mask_config *cfg, mca_hip_mask_ctx **out);
void mca_hip_mask_destroy(mca_hip_mask_ctx *ctx);
const char *mca_hip_mask_last_error(const mca_hip_mask_ctx *ctx);
int mca_hip_mask_reset(mca_hip_mask_ctx *ctx);
/* _thresholds (:361-362) and getBinCenterFrequency (cycles/sample): out[45] each */
int mca_hip_mask_get_thresholds(const mca_hip_mask_ctx *ctx, double *thresholds, double *center_freqs);
/* STFT + FastBinauralMasking::processParametrisation (FastBinauralMasking.cpp:126-210) + inverse FFT +
 * overlap-add for n_frames frames of n_streams independent 2-channel streams.
 * pcm_dev: sample n of channel c of stream s at pcm[s*stream_stride + c*ch_stride + n], (F+1)*hop samples;
 * The streams of a context start and advance together (the module's first-frame behaviour, FastBinauralMasking.cpp:186-197,
 * is tracked once per context): every call after create / reset must pass the same n_streams, else INVALID_ARGUMENT.
 * out_pcm_dev [streams][2][F*hop]; decisions_dev (may be NULL) [streams][F][45] int32:
 * 0 enhance, 1 temporal mask, 2 spatial mask. */
int mca_hip_mask_frames_dev(mca_hip_mask_ctx *ctx, const float *pcm_dev, long long stream_stride, long long ch_stride,
                            int n_streams, int n_frames, float *out_pcm_dev, int *decisions_dev, void *stream);
int mca_hip_mask_frames_host(mca_hip_mask_ctx *ctx, const float *pcm, int n_streams, int n_frames, float *out_pcm,
                             int *decisions);
/* the DSPONE hook itself: one frame, left/right CCS double[N+2] modified in place (:199-200), double on the GPU */
int mca_hip_mask_process_frame(mca_hip_mask_ctx *ctx, double *left, double *right, int ccs_len, int *decisions);
/* checkpoint / resume as mca_hip_state_*: the short-time powers Q and noise estimates of every stream, the overlap-add
 * tails, the frame counters (and the state of the frame hook) */
long long mca_hip_mask_state_size(const mca_hip_mask_ctx *ctx);
int mca_hip_mask_state_save(mca_hip_mask_ctx *ctx, void *blob, long long blob_bytes);
int mca_hip_mask_state_load(mca_hip_mask_ctx *ctx, const void *blob, long long blob_bytes);

/* ---- MultibandBinarualLocalisation (2 microphones) ---------------------------
 * Replaces mca::MultibandBinarualLocalisation(int sampleRate, ArrayDescription, int nbins = 15, bool usePowerFloor = 1)
 * (include/mcarray/MultibandBinarualLocalisation.h:38) with its per-frame hooks processSetup / processOneSubband /
 * processSumamry (src/mcarray/MultibandBinarualLocalisation.cpp:145-258) for batches of frames.  The sub-band
 * splitting (dsp::SubBandSTFTAnalysis, DSPONE) is [BUILD-DEFINES]: nbins unit-peak triangular filters, edges
 * linearly spaced between 100 Hz and maxFreqForSpatialAliasing(distance(0,1)) (ctor call :54-60). */
typedef struct mca_hip_mb_ctx mca_hip_mb_ctx;
typedef struct {
    int struct_size;
    int device;
    int sample_rate;
    int fft_size;            /* N = 2^calculateOrderFromSampleRate(fs, 0.025) (MultibandBinarualLocalisation.h:43); power of two 64..8192 */
    const double *mic_xyz;   /* [2][3] metres */
    int nbins;               /* sub-bands, reference default 15; 1..27 */
    int use_power_floor;     /* usePowerFloor (reference default true) */
    int max_arrays;          /* independent module objects (streams) this context holds state for */
} mca_hip_mb_config;
int  mca_hip_mb_create(const mca_hip_mb_config *cfg, mca_hip_mb_ctx **out);
void mca_hip_mb_destroy(mca_hip_mb_ctx *ctx);
const char *mca_hip_mb_last_error(const mca_hip_mb_ctx *ctx);
int  mca_hip_mb_reset(mca_hip_mb_ctx *ctx, void *stream);
int  mca_hip_mb_num_steps(const mca_hip_mb_ctx *ctx);                 /* _numSteps = floor(pi/step)+1 = 37 (:63) */
int  mca_hip_mb_get_filters(const mca_hip_mb_ctx *ctx, double *out);  /* [nbins][N/2+1] filter magnitudes */
/* n_frames frames of n_arrays independent 2-channel streams; pcm_dev as in mca_hip_mask_frames_dev.
 * Per frame (all [arrays][F]): doa_rad = _currentDOA[0] after the frame (:239/:254), prob = _prob[0] (:233/:255),
 * voiced (may be NULL) = 1 where setDOA fires (:225,:248), power (may be NULL) = the value handed to setDOA.
 * Optional: band_idx [arrays][F][nbins] first-max delay index per band (:184), energy_in_doa [arrays][F][D]
 * (_energyInDOA :190), band_corr [arrays][F][nbins][D] smoothed band correlations (:180-183). */
int mca_hip_mb_frames_dev(mca_hip_mb_ctx *ctx, const float *pcm_dev, long long array_stride, long long ch_stride,
                          int n_arrays, int n_frames, float *doa_rad_dev, float *prob_dev, unsigned char *voiced_dev,
                          float *power_dev, int *band_idx_dev, float *energy_in_doa_dev, float *band_corr_dev, void *stream);
int mca_hip_mb_frames_host(mca_hip_mb_ctx *ctx, const float *pcm, int n_arrays, int n_frames, float *doa_rad, float *prob,
                           unsigned char *voiced, float *power, int *band_idx, float *energy_in_doa, float *band_corr);
/* checkpoint / resume as mca_hip_state_*: smoothed band correlations, power-floor estimation, _currentDOA / _prob of every array */
long long mca_hip_mb_state_size(const mca_hip_mb_ctx *ctx);
int mca_hip_mb_state_save(mca_hip_mb_ctx *ctx, void *blob, long long blob_bytes);
int mca_hip_mb_state_load(mca_hip_mb_ctx *ctx, const void *blob, long long blob_bytes);

/* ---- MVDR-style beamformer with a per-bin spatial covariance (BASELINE.json configs[3]) ---------
 * [BUILD-DEFINES -- NO REFERENCE COUNTERPART]: the reference's only beamformer is the delay-and-sum of
 * mca::Beamformer::processFrame (src/mcarray/Beamformer.cpp:51-71); this module keeps its interface shape
 * (frames in, one output channel, a look direction in radians) and its steering convention (Beamformer.cpp:59:
 * x coordinate only, cos(DOA + pi/2)), and replaces the uniform weights 1/M by the minimum-variance
 * distortionless-response weights of SURVEY A.9.  Per stream and bin k:
 *     Phi_t = alpha Phi_{t-1} + (1 - alpha) x x^H,   PhiL = Phi_t + loading tr(Phi_t)/M I,
 *     w = PhiL^-1 d / (d^H PhiL^-1 d),  d_m = exp(+j 2 pi k fs x_m sin(DOA) / (N c)),   Y[k] = w^H x.
 * A bin whose covariance trace is <= 1e-30 (digital silence so far) uses w = d/M, i.e. the reference's
 * delay-and-sum.  With alpha = 0 ... 1 and loading > 0 the response towards DOA is exactly 1 (w^H d = 1). */
typedef struct mca_hip_mvdr_ctx mca_hip_mvdr_ctx;
typedef struct {
    int struct_size;
    int device;
    int sample_rate;
    int fft_size;            /* N, power of two 64..8192 with n_mics spectra of N/2+1 bins within the 160 KiB LDS */
    int n_mics;              /* M, 2..16 */
    const double *mic_xyz;   /* [M][3] metres */
    double alpha;            /* covariance memory, SURVEY A.9: 0.95 */
    double loading;          /* diagonal loading relative to tr(Phi)/M, SURVEY A.9: 1e-3 */
    int max_streams;         /* independent streams whose covariance / overlap-add state the context holds */
} mca_hip_mvdr_config;
int  mca_hip_mvdr_create(const mca_hip_mvdr_config *cfg, mca_hip_mvdr_ctx **out);
void mca_hip_mvdr_destroy(mca_hip_mvdr_ctx *ctx);
const char *mca_hip_mvdr_last_error(const mca_hip_mvdr_ctx *ctx);
int  mca_hip_mvdr_reset(mca_hip_mvdr_ctx *ctx, void *stream);      /* Phi = 0, overlap-add tails = 0 */
/* STFT analysis + the recursion above + inverse FFT + overlap-add for n_frames consecutive frames of n_streams
 * independent M-microphone streams (PCM layout as in mca_hip_process_frames_dev).
 *   doa_rad_dev  [streams][F] float   look direction per frame (e.g. the doa_rad output of mca_hip_localise_frames_dev)
 *   out_pcm_dev  [streams][F*hop] float (may be NULL)
 *   out_spec_dev [streams][F][N/2+1] interleaved re,im float: the beamformed spectra Y (may be NULL) */
int mca_hip_mvdr_frames_dev(mca_hip_mvdr_ctx *ctx, const float *pcm_dev, long long stream_stride, long long mic_stride,
                            int n_streams, int n_frames, const float *doa_rad_dev, float *out_pcm_dev,
                            float *out_spec_dev, void *stream);
int mca_hip_mvdr_frames_host(mca_hip_mvdr_ctx *ctx, const float *pcm, int n_streams, int n_frames, const float *doa_rad,
                             float *out_pcm, float *out_spec);
/* copy of the covariance of one stream: out[N/2+1][M][M] interleaved re,im double (full Hermitian matrices) */
int mca_hip_mvdr_get_covariance(mca_hip_mvdr_ctx *ctx, int stream_index, double *out);
/* checkpoint / resume as mca_hip_state_*: the covariances, their traces and the overlap-add tails of every stream */
long long mca_hip_mvdr_state_size(const mca_hip_mvdr_ctx *ctx);
int mca_hip_mvdr_state_save(mca_hip_mvdr_ctx *ctx, void *blob, long long blob_bytes);
int mca_hip_mvdr_state_load(mca_hip_mvdr_ctx *ctx, const void *blob, long long blob_bytes);
/* per-kernel timing as mca_hip_set_timing / mca_hip_get_timing: kernel_id 0 = analysis, 1 = solve, 2 = synthesis */
int mca_hip_mvdr_set_timing(mca_hip_mvdr_ctx *ctx, int enable);
int mca_hip_mvdr_get_timing(mca_hip_mvdr_ctx *ctx, int kernel_id, int *launches, double *total_ms);

/* ---- measurement ------------------------------------------------------------ */
typedef enum {
    MCA_HIP_K_STFT_PHAT = 0,   /* STFT + PHAT whitening + pair-group sums */
    MCA_HIP_K_SRP_GEMM = 1,    /* steering contraction (MFMA) */
    MCA_HIP_K_SCAN_PICK = 2,   /* IIR over frames + selectDOA */
    MCA_HIP_K_BEAMFORM = 3,    /* STFT + delay-and-sum + inverse FFT + overlap-add */
    MCA_HIP_K_GCC2_SCAN = 4,   /* 2-mic correlation smoothing + argmax + probability */
    MCA_HIP_K_MASK = 5,        /* binaural masking */
    MCA_HIP_K_FOLD = 6,        /* sum of the partial maps of a deep split-K contraction (small batches only) */
    MCA_HIP_K_REPAIR = 7,      /* MCA_HIP_SRP_ADAPTIVE: plan + exact recomputation of the sensitive rows + second pick */
    MCA_HIP_K_COUNT = 8
} mca_hip_kernel_id;
/* enable = 1: bracket every launch of the stream API with hipEvents on its stream */
int mca_hip_set_timing(mca_hip_ctx *ctx, int enable);
/* as mca_hip_set_timing for the kernel ids whose bit (1u << id) is set only: every event pair costs the stream ~1.5 us, a
 * throughput measurement brackets the one kernel it reports on */
int mca_hip_set_timing_mask(mca_hip_ctx *ctx, unsigned kernel_mask);
/* synchronises the recorded events; *launches and *total_ms accumulate since the last reset */
int mca_hip_get_timing(mca_hip_ctx *ctx, int kernel_id, int *launches, double *total_ms);
int mca_hip_reset_timing(mca_hip_ctx *ctx);
/* MCA_HIP_SRP_ADAPTIVE: totals since the last mca_hip_reset_timing (synchronises the device): frames that went through the
 * adaptive path, frames whose pick was flagged as sensitive to the fp16 error (in the eager form this includes the last frame of
 * every array and call, which is repeated so that the carried state is exact; lazy calls flag no frame for that), and frames
 * whose rows went through the exact analysis.  The
 * reference has no counterpart (it computes every pair and delay in double, SteeringBeamforming.cpp:104-130). */
int mca_hip_get_repair_stats(mca_hip_ctx *ctx, unsigned long long *frames, unsigned long long *flagged_frames,
                             unsigned long long *recomputed_frames);
/* ... and how much of those rows was recomputed (round 5, candidate columns): the exact contraction runs at the delays a flagged
 * frame's picks can be among -- the positions whose coarse energy reaches the lowest value the exact picks can have, plus the two
 * delays either side that feed the sign / median chain of SteeringBeamforming.cpp:159-173 -- instead of at all D of them.
 * candidate_columns: their number summed over the flagged frames; whole_row_frames: the flagged frames that took every delay
 * (no lower bound from the coarse row, a frame repeated for the carried state's sake, a row the coarse analysis could not vouch for). */
int mca_hip_get_repair_columns(mca_hip_ctx *ctx, unsigned long long *candidate_columns, unsigned long long *whole_row_frames);

/* library version string */
const char *mca_hip_version(void);

#ifdef __cplusplus
}
#endif
#endif /* MCARRAY_HIP_H */
