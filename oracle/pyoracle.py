"""ctypes binding of the CPU oracle (oracle/libmca_oracle.so).

TEST INFRASTRUCTURE ONLY -- may be imported by tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg, never by the product package (mcarray_amd).
"parity unpinned" against an executed reference, see mca_oracle.h.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

c_dp = C.POINTER(C.c_double)
c_ip = C.POINTER(C.c_int)


def build(force=False):
    so = os.path.join(_HERE, "libmca_oracle.so")
    src = os.path.join(_HERE, "mca_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        # MCA_ORACLE_LIB: another build of the same source (the sanitizer build of tests/test_oracle_asan.py)
        so = os.environ.get("MCA_ORACLE_LIB") or os.path.join(_HERE, "libmca_oracle.so")
        if not os.path.exists(so):
            build()
        L = C.CDLL(so)
        L.mca_or_speed_of_sound.restype = C.c_double
        L.mca_or_doa_to_delay_far_field.restype = C.c_float
        L.mca_or_doa_to_delay_far_field.argtypes = [C.c_float, C.c_float]
        L.mca_or_doa_to_delay_samples.restype = C.c_float
        L.mca_or_doa_to_delay_samples.argtypes = [C.c_float, C.c_float, C.c_int]
        L.mca_or_angle2doaidx.restype = C.c_float
        L.mca_or_angle2doaidx.argtypes = [C.c_float, C.c_float]
        L.mca_or_doaidx2angle.restype = C.c_float
        L.mca_or_doaidx2angle.argtypes = [C.c_int, C.c_float]
        L.mca_or_doa_step.restype = C.c_float
        L.mca_or_doa_step.argtypes = [C.c_double]
        L.mca_or_num_steps.restype = C.c_int
        L.mca_or_num_steps.argtypes = [C.c_float]
        for f in (L.mca_or_distance,):
            f.restype = C.c_double
            f.argtypes = [c_dp, C.c_int, C.c_int]
        for f in (L.mca_or_max_distance, L.mca_or_min_distance, L.mca_or_bandwidth):
            f.restype = C.c_double
            f.argtypes = [c_dp, C.c_int]
        L.mca_or_order_from_sample_rate.restype = C.c_int
        L.mca_or_order_from_sample_rate.argtypes = [C.c_int, C.c_double]
        L.mca_or_hann_periodic.argtypes = [c_dp, C.c_int]
        L.mca_or_rfft_ccs.argtypes = [c_dp, C.c_int, c_dp]
        L.mca_or_irfft_ccs.argtypes = [c_dp, C.c_int, c_dp]
        L.mca_or_log_power.restype = C.c_double
        L.mca_or_log_power.argtypes = [c_dp, C.c_int]
        L.mca_or_fft_power.restype = C.c_double
        L.mca_or_fft_power.argtypes = [C.POINTER(c_dp), C.c_int, C.c_int]
        L.mca_or_fft_log_power.restype = C.c_double
        L.mca_or_fft_log_power.argtypes = [C.POINTER(c_dp), C.c_int, C.c_int]
        L.mca_or_precompute_tau_matrix.argtypes = [c_dp, C.c_int, C.c_int, c_dp]
        L.mca_or_gcc_phat_tau_matrix.argtypes = [c_dp, c_dp, c_dp, C.c_int, C.c_int, c_dp]
        L.mca_or_gcc_tau_matrix.argtypes = [c_dp, c_dp, c_dp, C.c_int, C.c_int, c_dp, C.c_int]
        L.mca_or_steering_create.restype = C.c_void_p
        L.mca_or_steering_create.argtypes = [C.c_int, c_dp, C.c_int, C.c_int, C.c_double]
        L.mca_or_steering_destroy.argtypes = [C.c_void_p]
        L.mca_or_steering_reset.argtypes = [C.c_void_p]
        L.mca_or_steering_num_steps.restype = C.c_int
        L.mca_or_steering_num_steps.argtypes = [C.c_void_p]
        L.mca_or_steering_num_pairs.restype = C.c_int
        L.mca_or_steering_num_pairs.argtypes = [C.c_void_p]
        L.mca_or_steering_delays.restype = c_dp
        L.mca_or_steering_delays.argtypes = [C.c_void_p, C.c_int]
        L.mca_or_steering_process_frame.argtypes = [C.c_void_p, C.POINTER(c_dp), c_dp, c_dp, c_ip, C.c_int, c_dp, c_dp]
        L.mca_or_select_doa.argtypes = [c_dp, C.c_int, C.c_int, C.c_float, C.c_int, c_dp, c_dp, c_ip]
        L.mca_or_select_doa_fragile.restype = C.c_int
        L.mca_or_select_doa_fragile.argtypes = [c_dp, C.c_int, C.c_int, C.c_int, C.c_double]
        L.mca_or_select_doa_fragile_local.restype = C.c_int
        L.mca_or_select_doa_fragile_local.argtypes = [c_dp, C.c_int, C.c_int, C.c_int, C.c_double]
        L.mca_or_beamformer_process_frame.argtypes = [C.c_int, c_dp, C.c_int, C.c_int, C.POINTER(c_dp), c_dp, C.c_double]
        L.mca_or_bsl_create.restype = C.c_void_p
        L.mca_or_bsl_create.argtypes = [C.c_int, C.c_int, c_dp, C.c_int, C.c_int, C.c_int, C.c_double]
        L.mca_or_bsl_destroy.argtypes = [C.c_void_p]
        L.mca_or_bsl_localise.restype = C.c_int
        L.mca_or_bsl_localise.argtypes = [C.c_void_p, C.POINTER(c_dp), c_dp, c_dp, c_dp]
        L.mca_or_bsl_separate.argtypes = [C.c_void_p, C.POINTER(c_dp)]
        L.mca_or_das_stream.argtypes = [C.c_int, C.c_int, c_dp, C.c_int, c_dp, C.c_long, C.c_int, c_dp, c_dp, c_dp]
        L.mca_or_das_stream.restype = None
        L.mca_or_ssl_stream.argtypes = [C.c_int, C.c_int, c_dp, C.c_int, C.c_int, C.c_double, c_dp, C.c_long,
                                        C.c_int, c_ip, c_dp, c_dp, c_dp, c_dp]
        L.mca_or_ssl_stream_w.argtypes = [C.c_int, C.c_int, c_dp, C.c_int, C.c_int, C.c_double, C.c_int, c_dp, C.c_long,
                                          C.c_int, c_ip, c_dp, c_dp, c_dp, c_dp]
        L.mca_or_steering_set_weighting.argtypes = [C.c_void_p, C.c_int]
        L.mca_or_bsl_set_weighting.argtypes = [C.c_void_p, C.c_int]
        L.mca_or_ssl_stream_gated.argtypes = [C.c_int, C.c_int, c_dp, C.c_int, C.c_int, C.c_double, C.c_int, c_dp, C.c_long,
                                              C.c_int, c_ip, c_dp, c_dp, c_dp, c_dp, c_ip, c_dp]
        L.mca_or_freqgcc_create.restype = C.c_void_p
        L.mca_or_freqgcc_create.argtypes = [C.c_int, c_dp, C.c_int, C.c_int, C.c_int, C.c_double]
        L.mca_or_freqgcc_destroy.argtypes = [C.c_void_p]
        L.mca_or_freqgcc_num_steps.restype = C.c_int
        L.mca_or_freqgcc_num_steps.argtypes = [C.c_void_p]
        L.mca_or_freqgcc_process.restype = C.c_int
        L.mca_or_freqgcc_process.argtypes = [C.c_void_p, c_dp, c_dp, c_dp, c_ip, c_dp, c_dp]
        L.mca_or_freqgcc_set_probability.argtypes = [C.c_void_p, c_dp, c_dp, C.c_int]
        L.mca_or_multiband_create.restype = C.c_void_p
        L.mca_or_multiband_create.argtypes = [C.c_int, c_dp, C.c_int, C.c_int, C.c_int, C.c_int]
        L.mca_or_multiband_destroy.argtypes = [C.c_void_p]
        L.mca_or_multiband_num_steps.restype = C.c_int
        L.mca_or_multiband_num_steps.argtypes = [C.c_void_p]
        L.mca_or_multiband_filters.restype = c_dp
        L.mca_or_multiband_filters.argtypes = [C.c_void_p]
        L.mca_or_multiband_process.restype = C.c_int
        L.mca_or_multiband_process.argtypes = [C.c_void_p, c_dp, c_dp, c_ip, c_dp, c_dp, c_dp, c_dp, c_dp, c_dp]
        L.mca_or_masking_create.restype = C.c_void_p
        L.mca_or_masking_create.argtypes = [C.c_int, C.c_int, C.c_double, C.c_float, C.c_float, C.c_int, C.c_int]
        L.mca_or_masking_destroy.argtypes = [C.c_void_p]
        L.mca_or_masking_nbins.restype = C.c_int
        for f in (L.mca_or_masking_thresholds, L.mca_or_masking_filters, L.mca_or_masking_center_freqs,
                  L.mca_or_masking_short_time_power):
            f.restype = c_dp
            f.argtypes = [C.c_void_p]
        L.mca_or_masking_process.argtypes = [C.c_void_p, c_dp, c_dp, c_ip]
        L.mca_or_masking_stream.argtypes = [C.c_void_p, c_dp, c_dp, C.c_int, c_dp, c_dp]
        L.mca_or_mel_filterbank.argtypes = [C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, c_dp, c_dp]
        L.mca_or_mvdr_create.restype = C.c_void_p
        L.mca_or_mvdr_create.argtypes = [C.c_int, C.c_int, c_dp, C.c_int, C.c_double, C.c_double]
        L.mca_or_mvdr_destroy.argtypes = [C.c_void_p]
        L.mca_or_mvdr_reset.argtypes = [C.c_void_p]
        L.mca_or_mvdr_covariance.restype = c_dp
        L.mca_or_mvdr_covariance.argtypes = [C.c_void_p]
        L.mca_or_mvdr_process_frame.argtypes = [C.c_void_p, C.POINTER(c_dp), c_dp, C.c_double]
        L.mca_or_mvdr_stream.argtypes = [C.c_void_p, c_dp, C.c_long, C.c_int, c_dp, c_dp, c_dp]
        _LIB = L
    return _LIB


def _dp(a):
    return a.ctypes.data_as(c_dp)


def _ip(a):
    return a.ctypes.data_as(c_ip)


def _xyz(x):
    a = np.ascontiguousarray(np.asarray(x, dtype=np.float64))
    if a.ndim == 1:  # linear array along x
        a = np.stack([a, np.zeros_like(a), np.zeros_like(a)], axis=1)
    return np.ascontiguousarray(a)


def _ptr_array(rows):
    arr = (c_dp * len(rows))()
    for i, r in enumerate(rows):
        arr[i] = _dp(r)
    return arr


# ---- helpers -------------------------------------------------------------
def doa_step(step_deg):
    return float(lib().mca_or_doa_step(step_deg))


def num_steps(step_deg):
    return int(lib().mca_or_num_steps(lib().mca_or_doa_step(step_deg)))


def doaidx2angle(idx, step_deg):
    return float(lib().mca_or_doaidx2angle(int(idx), lib().mca_or_doa_step(step_deg)))


def delay_samples(doa, dist, fs):
    return float(lib().mca_or_doa_to_delay_samples(doa, dist, fs))


def distance(xyz, i, j):
    return float(lib().mca_or_distance(_dp(_xyz(xyz)), i, j))


def max_distance(xyz):
    a = _xyz(xyz)
    return float(lib().mca_or_max_distance(_dp(a), len(a)))


def rfft_ccs(x):
    x = np.ascontiguousarray(x, dtype=np.float64)
    out = np.empty(len(x) + 2)
    lib().mca_or_rfft_ccs(_dp(x), len(x), _dp(out))
    return out


def irfft_ccs(ccs):
    ccs = np.ascontiguousarray(ccs, dtype=np.float64)
    n = len(ccs) - 2
    out = np.empty(n)
    lib().mca_or_irfft_ccs(_dp(ccs), n, _dp(out))
    return out


def hann(N):
    w = np.empty(N)
    lib().mca_or_hann_periodic(_dp(w), N)
    return w


def log_power(x):
    x = np.ascontiguousarray(x, dtype=np.float64)
    return float(lib().mca_or_log_power(_dp(x), len(x)))


def stft_frames(pcm, N):
    """pcm [M][(F+1)*hop] float64 -> ccs [F][M][N+2] (Hann periodic, hop N/2)."""
    pcm = np.ascontiguousarray(pcm, dtype=np.float64)
    M, L = pcm.shape
    hop = N // 2
    F = L // hop - 1
    w = hann(N)
    out = np.empty((F, M, N + 2))
    for t in range(F):
        for c in range(M):
            out[t, c] = rfft_ccs(pcm[c, t * hop:t * hop + N] * w)
    return out


class Steering:
    """mca::SteeringBeamforming restatement (SteeringBeamforming.cpp:34-195)."""

    def __init__(self, fs, xyz, ccs_len, step_deg=5.0):
        self.xyz = _xyz(xyz)
        self.M = len(self.xyz)
        self.ccs_len = ccs_len
        self.h = lib().mca_or_steering_create(fs, _dp(self.xyz), self.M, ccs_len, step_deg)
        self.D = lib().mca_or_steering_num_steps(self.h)
        self.P = lib().mca_or_steering_num_pairs(self.h)

    def __del__(self):
        if getattr(self, "h", None):
            lib().mca_or_steering_destroy(self.h)
            self.h = None

    def reset(self):
        lib().mca_or_steering_reset(self.h)

    def delays(self, pair):
        p = lib().mca_or_steering_delays(self.h, pair)
        return np.ctypeslib.as_array(p, shape=(self.D,)).copy()

    def process_frame(self, frames, n_sources=1):
        """frames [M][ccs_len] -> dict(doa, prob, bin, energy, corr)"""
        frames = np.ascontiguousarray(frames, dtype=np.float64)
        rows = [frames[c] for c in range(self.M)]
        doa = np.empty(n_sources)
        prob = np.empty(n_sources)
        b = np.empty(n_sources, dtype=np.int32)
        E = np.empty(self.D)
        Cc = np.empty(self.D)
        lib().mca_or_steering_process_frame(self.h, _ptr_array(rows), _dp(doa), _dp(prob), _ip(b), n_sources,
                                            _dp(E), _dp(Cc))
        return dict(doa=doa, prob=prob, bin=b, energy=E, corr=Cc)


def select_doa(E, n_pairs, step_deg, n_sources=1):
    E = np.ascontiguousarray(E, dtype=np.float64)
    doa = np.empty(n_sources)
    prob = np.empty(n_sources)
    b = np.empty(n_sources, dtype=np.int32)
    lib().mca_or_select_doa(_dp(E), len(E), n_pairs, lib().mca_or_doa_step(step_deg), n_sources, _dp(doa), _dp(prob), _ip(b))
    return doa, prob, b


def precompute_tau_matrix(tau, K):
    """dsp::GeneralisedCrossCorrelation::precomputeTauMatrix [INFERRED, SURVEY A.3]: T[d][k] = exp(+j 2 pi k tau_d / N) as
    [D][K][2] doubles"""
    tau = np.ascontiguousarray(tau, dtype=np.float64)
    T = np.empty((len(tau), K, 2))
    lib().mca_or_precompute_tau_matrix(_dp(tau), len(tau), K, _dp(T))
    return T


def gcc_tau_matrix(A_ccs, B_ccs, T, K, D, weighting=0):
    """calculateCorrelationsForPrecomputedTauMatrix on two CCS frames: [D][2] complex correlations (mca_or_gcc_tau_matrix)"""
    A = np.ascontiguousarray(A_ccs, dtype=np.float64)
    B = np.ascontiguousarray(B_ccs, dtype=np.float64)
    out = np.empty((D, 2))
    lib().mca_or_gcc_tau_matrix(_dp(A), _dp(B), _dp(np.ascontiguousarray(T)), K, D, _dp(out), int(weighting))
    return out


def select_doa_fragile(E, n_pairs, n_sources=1, eps=1e-6):
    """True if selectDOA's picks on the energy row E are not pinned against perturbations of size eps of the normalised
    energies (peak ties, sign-chain ties, zero picks): mca_or_select_doa_fragile."""
    E = np.ascontiguousarray(E, dtype=np.float64)
    return bool(lib().mca_or_select_doa_fragile(_dp(E), len(E), int(n_pairs), int(n_sources), float(eps)))


def select_doa_fragile_local(E, n_pairs, n_sources=1, eps_rel=1e-6):
    """mca_or_select_doa_fragile_local: the same classifier, every comparison at eps_rel x max(1, |values compared|)."""
    E = np.ascontiguousarray(E, dtype=np.float64)
    return bool(lib().mca_or_select_doa_fragile_local(_dp(E), len(E), int(n_pairs), int(n_sources), float(eps_rel)))


def beamformer_process_frame(fs, xyz, frames, doa):
    """mca::Beamformer::processFrame (Beamformer.cpp:51-71). frames [M][ccs] -> out [ccs]"""
    a = _xyz(xyz)
    frames = np.ascontiguousarray(frames, dtype=np.float64)
    M, ccs = frames.shape
    out = np.empty(ccs)
    rows = [frames[c] for c in range(M)]
    lib().mca_or_beamformer_process_frame(fs, _dp(a), M, ccs, _ptr_array(rows), _dp(out), float(doa))
    return out


def das_stream(fs, N, xyz, pcm, doa_rad, tail=None):
    """The delay-and-sum stream at caller-given angles (mca_or_das_stream: mcabeamf.cpp:77-122 around Beamformer.cpp:51-71).
    pcm [M][(F+1)*hop]; doa_rad scalar or [F]; tail [hop] is the overlap-add carry, updated in place (None: a fresh stream).
    Returns out [F*hop]."""
    a = _xyz(xyz)
    pcm = np.ascontiguousarray(pcm, dtype=np.float64)
    M, L = pcm.shape
    hop = N // 2
    F = L // hop - 1
    doa = np.ascontiguousarray(np.broadcast_to(np.asarray(doa_rad, dtype=np.float64), (F,)))
    if tail is None:
        tail = np.zeros(hop)
    assert tail.dtype == np.float64 and tail.shape == (hop,) and tail.flags["C_CONTIGUOUS"]
    out = np.empty(F * hop)
    lib().mca_or_das_stream(fs, N, _dp(a), M, _dp(pcm), L, F, _dp(doa), _dp(tail), _dp(out))
    return out


class BSL:
    """mca::BeamformingSeparationAndLocalisation restatement."""

    def __init__(self, fs, ccs_len, xyz, n_sources=1, use_power_floor=False, step_deg=5.0):
        self.xyz = _xyz(xyz)
        self.M = len(self.xyz)
        self.S = n_sources
        self.ccs_len = ccs_len
        self.h = lib().mca_or_bsl_create(fs, ccs_len, _dp(self.xyz), self.M, n_sources, int(use_power_floor), step_deg)

    def __del__(self):
        if getattr(self, "h", None):
            lib().mca_or_bsl_destroy(self.h)
            self.h = None

    def localise(self, frames):
        frames = np.ascontiguousarray(frames, dtype=np.float64)
        rows = [frames[c] for c in range(self.M)]
        doa = np.zeros(self.S)
        prob = np.zeros(self.S)
        power = C.c_double(0)
        fired = lib().mca_or_bsl_localise(self.h, _ptr_array(rows), _dp(doa), _dp(prob), C.byref(power))
        return bool(fired), doa, prob, power.value

    def separate(self, frames):
        frames = np.array(frames, dtype=np.float64, order="C")
        rows = [frames[c] for c in range(self.M)]
        lib().mca_or_bsl_separate(self.h, _ptr_array(rows))
        return frames


GCC_WEIGHTING = {"phat": 0, "none": 1}


def ssl_stream(fs, N, xyz, pcm, n_sources=1, step_deg=5.0, want_map=False, want_audio=True, weighting="phat"):
    """Whole SourceSeparationAndLocalisation stream. pcm [M][(F+1)*hop].  weighting: "phat" (default) or "none" -- the GCC
    weighting of the steered sum (mca_or_gcc_tau_matrix)."""
    a = _xyz(xyz)
    pcm = np.ascontiguousarray(pcm, dtype=np.float64)
    M, L = pcm.shape
    hop = N // 2
    F = L // hop - 1
    D = num_steps(step_deg)
    S = n_sources
    bins = np.empty((F, S), dtype=np.int32)
    doa = np.empty((F, S))
    prob = np.empty((F, S))
    nout = min(M, S)
    out = np.zeros((nout, F * hop)) if want_audio else None
    emap = np.empty((F, D)) if want_map else None
    lib().mca_or_ssl_stream_w(fs, N, _dp(a), M, S, step_deg, GCC_WEIGHTING[weighting], _dp(pcm), L, F, _ip(bins), _dp(doa), _dp(prob),
                              _dp(out) if want_audio else None, _dp(emap) if want_map else None)
    return dict(bin=bins, doa=doa, prob=prob, out=out, energy=emap)


def ssl_stream_gated(fs, N, xyz, pcm, n_sources=1, step_deg=5.0, use_power_floor=True, want_audio=True):
    """SourceSeparationAndLocalisation stream with the power gate (usePowerFloor, the reference's default)."""
    a = _xyz(xyz)
    pcm = np.ascontiguousarray(pcm, dtype=np.float64)
    M, L = pcm.shape
    hop = N // 2
    F = L // hop - 1
    D = num_steps(step_deg)
    S = n_sources
    bins = np.empty((F, S), dtype=np.int32)
    doa = np.empty((F, S))
    prob = np.empty((F, S))
    out = np.zeros((min(M, S), F * hop)) if want_audio else None
    emap = np.empty((F, D))
    fired = np.empty(F, dtype=np.int32)
    power = np.empty(F)
    lib().mca_or_ssl_stream_gated(fs, N, _dp(a), M, S, step_deg, int(use_power_floor), _dp(pcm), L, F, _ip(bins), _dp(doa),
                                  _dp(prob), _dp(out) if want_audio else None, _dp(emap), _ip(fired), _dp(power))
    return dict(bin=bins, doa=doa, prob=prob, out=out, energy=emap, fired=fired, power=power)


class FreqGCC:
    """mca::FreqGCCBinauralLocalisation deterministic part (BinauralLocalisation.cpp:320-631)."""

    def __init__(self, fs, xyz, ccs_len, use_power_floor=False, step_deg=3.0):
        self.xyz = _xyz(xyz)
        self.h = lib().mca_or_freqgcc_create(fs, _dp(self.xyz), len(self.xyz), ccs_len, int(use_power_floor), step_deg)
        self.D = lib().mca_or_freqgcc_num_steps(self.h)

    def __del__(self):
        if getattr(self, "h", None):
            lib().mca_or_freqgcc_destroy(self.h)
            self.h = None

    def process(self, left, right):
        left = np.ascontiguousarray(left, dtype=np.float64)
        right = np.ascontiguousarray(right, dtype=np.float64)
        corr = np.zeros(self.D)
        idx = C.c_int(0)
        doa = C.c_double(0)
        power = C.c_double(0)
        v = lib().mca_or_freqgcc_process(self.h, _dp(left), _dp(right), _dp(corr), C.byref(idx), C.byref(doa), C.byref(power))
        return bool(v), corr, idx.value, doa.value, power.value

    def set_probability(self, doas):
        doas = np.ascontiguousarray(doas, dtype=np.float64)
        probs = np.empty(len(doas))
        lib().mca_or_freqgcc_set_probability(self.h, _dp(doas), _dp(probs), len(doas))
        return probs


class Multiband:
    """mca::MultibandBinarualLocalisation restatement (MultibandBinarualLocalisation.cpp:52-258)."""

    def __init__(self, fs, xyz, ccs_len, nbins=15, use_power_floor=True):
        self.xyz = _xyz(xyz)
        self.nbins = nbins
        self.K = ccs_len // 2
        self.h = lib().mca_or_multiband_create(fs, _dp(self.xyz), len(self.xyz), ccs_len, nbins, int(use_power_floor))
        self.D = lib().mca_or_multiband_num_steps(self.h)

    def __del__(self):
        if getattr(self, "h", None):
            lib().mca_or_multiband_destroy(self.h)
            self.h = None

    def filters(self):
        p = lib().mca_or_multiband_filters(self.h)
        return np.ctypeslib.as_array(p, shape=(self.nbins, self.K)).copy()

    def process(self, left, right):
        """one frame -> dict(fired, band_idx [nbins], band_energy [nbins], band_corr [nbins][D], energy_in_doa [D], doa, prob, power)"""
        left = np.ascontiguousarray(left, dtype=np.float64)
        right = np.ascontiguousarray(right, dtype=np.float64)
        bi = np.zeros(self.nbins, dtype=np.int32)
        be = np.zeros(self.nbins)
        bc = np.zeros((self.nbins, self.D))
        eid = np.zeros(self.D)
        doa, prob, power = C.c_double(0), C.c_double(0), C.c_double(0)
        v = lib().mca_or_multiband_process(self.h, _dp(left), _dp(right), _ip(bi), _dp(be), _dp(bc), _dp(eid),
                                           C.byref(doa), C.byref(prob), C.byref(power))
        return dict(fired=bool(v), band_idx=bi, band_energy=be, band_corr=bc, energy_in_doa=eid, doa=doa.value,
                    prob=prob.value, power=power.value)


class MVDR:
    """MVDR-style beamformer with a per-bin spatial covariance, SURVEY A.9 [BUILD-DEFINES, no reference counterpart]."""

    def __init__(self, fs, N, xyz, alpha=0.95, loading=1e-3):
        self.xyz = _xyz(xyz)
        self.M = len(self.xyz)
        assert self.M <= 16
        self.N, self.K = N, N // 2 + 1
        self.h = lib().mca_or_mvdr_create(fs, N, _dp(self.xyz), self.M, alpha, loading)

    def __del__(self):
        if getattr(self, "h", None):
            lib().mca_or_mvdr_destroy(self.h)
            self.h = None

    def reset(self):
        lib().mca_or_mvdr_reset(self.h)

    def covariance(self):
        p = lib().mca_or_mvdr_covariance(self.h)
        a = np.ctypeslib.as_array(p, shape=(self.K, self.M, self.M, 2)).copy()
        return a[..., 0] + 1j * a[..., 1]

    def process_frame(self, frames, doa):
        """frames [M][N+2] CCS -> out [N+2] CCS"""
        rows = [np.ascontiguousarray(f, dtype=np.float64) for f in frames]
        out = np.zeros(self.N + 2)
        lib().mca_or_mvdr_process_frame(self.h, _ptr_array(rows), _dp(out), float(doa))
        return out

    def stream(self, pcm, doa_rad, want_spec=False):
        """pcm [M][(F+1)*hop] double, doa_rad [F] -> dict(out [F*hop], spec [F][N+2] or None)"""
        pcm = np.ascontiguousarray(pcm, dtype=np.float64)
        hop = self.N // 2
        F = pcm.shape[1] // hop - 1
        doa = np.ascontiguousarray(np.broadcast_to(np.asarray(doa_rad, dtype=np.float64), (F,)))
        out = np.zeros(F * hop)
        spec = np.zeros((F, self.N + 2)) if want_spec else None
        lib().mca_or_mvdr_stream(self.h, _dp(pcm), pcm.shape[1], F, _dp(doa), _dp(out), _dp(spec) if want_spec else None)
        return dict(out=out, spec=spec)


FACTOR, RELATIVE, FULL, NOISY, NOTHING = 0, 1, 3, 4, 5
BOTH, SPATIAL, TEMPORAL = 0, 1, 2


class Masking:
    """mca::FastBinauralMasking restatement (FastBinauralMasking.cpp:51-538)."""

    def __init__(self, fs, N, micro_distance, low_freq, high_freq, method=RELATIVE, algorithm=BOTH):
        self.N = N
        self.K = N // 2 + 1
        self.h = lib().mca_or_masking_create(fs, N, micro_distance, low_freq, high_freq, method, algorithm)

    def __del__(self):
        if getattr(self, "h", None):
            lib().mca_or_masking_destroy(self.h)
            self.h = None

    @property
    def thresholds(self):
        return np.ctypeslib.as_array(lib().mca_or_masking_thresholds(self.h), shape=(45,)).copy()

    @property
    def filters(self):
        return np.ctypeslib.as_array(lib().mca_or_masking_filters(self.h), shape=(45, self.K)).copy()

    @property
    def center_freqs(self):
        return np.ctypeslib.as_array(lib().mca_or_masking_center_freqs(self.h), shape=(45,)).copy()

    @property
    def short_time_power(self):
        return np.ctypeslib.as_array(lib().mca_or_masking_short_time_power(self.h), shape=(45,)).copy()

    def process(self, left, right):
        left = np.array(left, dtype=np.float64, order="C")
        right = np.array(right, dtype=np.float64, order="C")
        dec = np.zeros(45, dtype=np.int32)
        lib().mca_or_masking_process(self.h, _dp(left), _dp(right), _ip(dec))
        return left, right, dec

    def stream(self, pcm_l, pcm_r):
        pcm_l = np.ascontiguousarray(pcm_l, dtype=np.float64)
        pcm_r = np.ascontiguousarray(pcm_r, dtype=np.float64)
        hop = self.N // 2
        F = len(pcm_l) // hop - 1
        ol = np.empty(F * hop)
        orr = np.empty(F * hop)
        lib().mca_or_masking_stream(self.h, _dp(pcm_l), _dp(pcm_r), F, _dp(ol), _dp(orr))
        return ol, orr
