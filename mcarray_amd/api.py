"""Python mirror of the reference's module API on top of the C ABI (libmcarray_hip.so).

Class and method names follow the reference (mca::SteeringBeamforming, mca::Beamformer,
mca::BeamformingSeparationAndLocalisation, mca::SourceSeparationAndLocalisation), so the parity
tests read like the reference's tests.  Everything numerical happens in the HIP library; numpy is
used only to marshal buffers.  torch is optional here and only used by the *_dev helpers
(device tensors in, device tensors out).
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import MCArrayHipError

SRP_FP32, SRP_FP16X3, SRP_FP16, SRP_ADAPTIVE = 0, 1, 2, 3
GCC_PHAT, GCC_NONE = 0, 1            # mca_hip_gcc_weighting
K_STFT_PHAT, K_SRP_GEMM, K_SCAN_PICK, K_BEAMFORM, K_GCC2_SCAN, K_MASK, K_FOLD, K_REPAIR = 0, 1, 2, 3, 4, 5, 6, 7
KERNEL_NAMES = {K_STFT_PHAT: "k_stft_phat", K_SRP_GEMM: "k_srp_gemm", K_SCAN_PICK: "k_scan_pick", K_BEAMFORM: "k_beamform_ola",
                K_FOLD: "k_sum_planes", K_REPAIR: "repair"}


class PinnedBuffer:
    """numpy view of page-locked host memory (mca_hip_host_alloc): buffers of this kind make the host-pointer entry points
    overlap upload, kernels and download at the rate of the PCIe link.  Keep the object alive while `.array` is in use."""

    def __init__(self, shape, dtype):
        self._lib = _lib.load()
        self.nbytes = int(np.prod(shape)) * np.dtype(dtype).itemsize
        self.ptr = self._lib.mca_hip_host_alloc(max(self.nbytes, 1))
        if not self.ptr:
            raise MCArrayHipError("mca_hip_host_alloc(%d) failed" % self.nbytes)
        raw = (C.c_char * max(self.nbytes, 1)).from_address(self.ptr)
        self.array = np.frombuffer(raw, dtype=dtype, count=int(np.prod(shape))).reshape(shape)

    def close(self):
        if getattr(self, "ptr", None):
            self.array = None
            self._lib.mca_hip_host_free(self.ptr)
            self.ptr = None

    def __del__(self):
        self.close()


def _xyz(x):
    a = np.asarray(x, dtype=np.float64)
    if a.ndim == 1:   # ArrayDescription::make_linear_array_description (ArrayDescription.cpp:41-49)
        a = np.stack([a, np.zeros_like(a), np.zeros_like(a)], axis=1)
    return np.ascontiguousarray(a)


class Context:
    """Owns one mca_hip_ctx (the state of max_arrays independent module objects)."""

    def __init__(self, sample_rate, mic_positions, fft_size=1024, doa_step_deg=5.0, n_sources=1, use_power_floor=False,
                 srp_precision=SRP_FP32, max_arrays=1, device=0, gcc_weighting=GCC_PHAT, adaptive_fallback=True,
                 adaptive_min_rows=0, adaptive_max_sources=0, scan_carry=False):
        self._lib = _lib.load()
        self.xyz = _xyz(mic_positions)
        self.M = len(self.xyz)
        self.N = fft_size
        self.hop = fft_size // 2
        self.S = n_sources
        self.fs = sample_rate
        self.use_power_floor = bool(use_power_floor)
        cfg = _lib.Config()
        cfg.struct_size = C.sizeof(_lib.Config)
        cfg.device = device
        cfg.sample_rate = sample_rate
        cfg.fft_size = fft_size
        cfg.n_mics = self.M
        cfg.mic_xyz = self.xyz.ctypes.data_as(_lib.c_dp)
        cfg.doa_step_deg = doa_step_deg
        cfg.n_sources = n_sources
        cfg.use_power_floor = int(use_power_floor)
        cfg.srp_precision = srp_precision
        cfg.max_arrays = max_arrays
        cfg.gcc_weighting = gcc_weighting
        cfg.adaptive_fallback = 0 if adaptive_fallback else 1      # mca_hip_adaptive_fallback: AUTO / OFF (OFF: bit-reproducible runs)
        cfg.adaptive_min_rows = adaptive_min_rows
        cfg.adaptive_max_sources = adaptive_max_sources
        cfg.scan_carry = int(scan_carry)
        h = C.c_void_p()
        rc = self._lib.mca_hip_create(C.byref(cfg), C.byref(h))
        if rc != 0:
            raise MCArrayHipError("mca_hip_create failed (%d): %s" % (rc, self._lib.mca_hip_last_error(None).decode()))
        self.h = h
        self.D = self._lib.mca_hip_num_steps(h)
        self.P = self._lib.mca_hip_num_pairs(h)
        self.G = self._lib.mca_hip_num_groups(h)

    def close(self):
        if getattr(self, "h", None):
            self._lib.mca_hip_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def _check(self, rc):
        if rc != 0:
            raise MCArrayHipError("libmcarray_hip error %d: %s" % (rc, self._lib.mca_hip_last_error(self.h).decode()))

    # ---- introspection ----
    def pair_delays(self):
        out = np.empty((self.P, self.D), dtype=np.float32)
        self._check(self._lib.mca_hip_get_pair_delays(self.h, out.ctypes.data_as(_lib.c_fp)))
        return out

    def doa_grid(self):
        out = np.empty(self.D, dtype=np.float32)
        self._check(self._lib.mca_hip_get_doa_grid(self.h, out.ctypes.data_as(_lib.c_fp)))
        return out

    def reset(self, stream=None):
        self._check(self._lib.mca_hip_reset(self.h, stream))

    def reserve(self, n_arrays, n_frames):
        self._check(self._lib.mca_hip_reserve(self.h, n_arrays, n_frames))

    def state_save(self):
        """checkpoint of the per-array stream state (E_prev, overlap-add tails, gate, last DOA ...) as bytes"""
        n = self._lib.mca_hip_state_size(self.h)
        buf = C.create_string_buffer(n)
        self._check(self._lib.mca_hip_state_save(self.h, buf, n))
        return buf.raw

    def state_load(self, blob):
        self._check(self._lib.mca_hip_state_load(self.h, blob, len(blob)))

    # ---- stream API, host buffers ----
    def process_frames_host(self, pcm, want_energy=False, want_audio=True, into=None):
        """pcm float32 [A][M][(F+1)*hop] -> dict(bin [A][F][S], doa, prob, energy [A][F][D], out [A][S][F*hop]).
        into: optional dict of preallocated result arrays (e.g. PinnedBuffer(...).array) for bin / doa / prob / energy / out."""
        i16 = isinstance(pcm, np.ndarray) and pcm.dtype == np.int16       # 16-bit PCM goes up as it is (half the PCIe bytes)
        pcm = np.ascontiguousarray(pcm, dtype=np.int16 if i16 else np.float32)
        if pcm.ndim == 2:
            pcm = pcm[None]
        A, M, L = pcm.shape
        if M != self.M:
            raise MCArrayHipError("pcm has %d channels, context has %d microphones" % (M, self.M))
        F = L // self.hop - 1
        if F < 1 or (F + 1) * self.hop != L:
            raise MCArrayHipError("pcm length must be (F+1)*hop samples")
        S, D = self.S, self.D
        into = into or {}

        def buf(key, shape, dtype):
            b = into.get(key)
            if b is None:
                return np.empty(shape, dtype=dtype)
            if b.shape != shape or b.dtype != dtype or not b.flags["C_CONTIGUOUS"]:
                raise MCArrayHipError("into[%r] must be a contiguous %s array of shape %s" % (key, np.dtype(dtype).name, shape))
            return b
        bins = buf("bin", (A, F, S), np.int32)
        doa = buf("doa", (A, F, S), np.float32)
        prob = buf("prob", (A, F, S), np.float32)
        energy = buf("energy", (A, F, D), np.float32) if want_energy else None
        out = buf("out", (A, S, F * self.hop), np.float32) if want_audio else None
        fp = _lib.c_fp
        entry = self._lib.mca_hip_process_frames_host_i16 if i16 else self._lib.mca_hip_process_frames_host
        self._check(entry(
            self.h, pcm.ctypes.data_as(C.POINTER(C.c_short) if i16 else fp), A, F, bins.ctypes.data_as(_lib.c_ip), doa.ctypes.data_as(fp),
            prob.ctypes.data_as(fp), energy.ctypes.data_as(fp) if want_energy else None,
            out.ctypes.data_as(fp) if want_audio else None))
        res = dict(bin=bins, doa=doa, prob=prob, energy=energy, out=out)
        if self.use_power_floor:
            voiced = np.empty((A, F), dtype=np.uint8)
            power = np.empty((A, F), dtype=np.float32)
            self._check(self._lib.mca_hip_copy_gate(self.h, voiced.ctypes.data_as(C.c_void_p), power.ctypes.data_as(fp)))
            res["voiced"] = voiced
            res["power"] = power
        return res

    # ---- stream API, device tensors (torch used for memory only) ----
    def process_frames_dev(self, pcm, n_frames, doa_bin, doa_rad, prob, energy=None, out_pcm=None, stream=None,
                           localise=True, separate=True, bins_are_grid=False):
        """pcm: torch float32 cuda tensor [A][M][>= (F+1)*hop]; outputs preallocated cuda tensors.
        bins_are_grid (separation only): doa_rad holds the grid angles of doa_bin (the localiser's own picks)."""
        A, M, L = pcm.shape
        if M != self.M:
            raise MCArrayHipError("pcm has %d channels, context has %d microphones" % (M, self.M))
        if not pcm.is_contiguous():
            raise MCArrayHipError("pcm must be contiguous")
        ptr = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
        if localise and separate and out_pcm is not None and doa_rad is not None:
            self._check(self._lib.mca_hip_process_frames_dev(self.h, ptr(pcm), M * L, L, A, n_frames, ptr(doa_bin), ptr(doa_rad), ptr(prob),
                                                             ptr(energy), ptr(out_pcm), stream))
            return
        if localise:
            self._check(self._lib.mca_hip_localise_frames_dev(self.h, ptr(pcm), M * L, L, A, n_frames, ptr(doa_bin),
                                                              ptr(doa_rad), ptr(prob), ptr(energy), stream))
        if separate and out_pcm is not None:
            if bins_are_grid and doa_bin is not None:
                self._check(self._lib.mca_hip_separate_frames_bins_dev(self.h, ptr(pcm), M * L, L, A, n_frames, ptr(doa_bin), ptr(doa_rad),
                                                                       ptr(out_pcm), stream))
            else:
                self._check(self._lib.mca_hip_separate_frames_dev(self.h, ptr(pcm), M * L, L, A, n_frames, ptr(doa_rad),
                                                                  ptr(out_pcm), stream))

    # ---- real-time mode: the stream call as a HIP graph ----
    def graph_create(self, pcm, n_frames, doa_bin, doa_rad, prob, energy=None, out_pcm=None):
        """Fixes the shape and buffers (torch cuda tensors, as process_frames_dev) of a stream call; returns a StreamGraph
        whose launch() replays the call's kernels as one HIP graph on the current contents of `pcm`."""
        A, M, L = pcm.shape
        if M != self.M or not pcm.is_contiguous():
            raise MCArrayHipError("pcm must be a contiguous [A][M][L] tensor with M = the context's microphones")
        ptr = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
        h = C.c_void_p()
        self._check(self._lib.mca_hip_graph_create(self.h, ptr(pcm), M * L, L, A, n_frames, ptr(doa_bin), ptr(doa_rad), ptr(prob),
                                                   ptr(energy), ptr(out_pcm), C.byref(h)))
        return StreamGraph(self, h, (pcm, doa_bin, doa_rad, prob, energy, out_pcm))

    # ---- 2-microphone GCC-PHAT path ----
    def gcc2_frames_host(self, pcm, want_corr=False):
        """pcm float32 [A][2][(F+1)*hop] -> dict(argmax [A][F], doa [A][F] (smoothed, rad), prob [A][F], corr [A][F][D])"""
        pcm = np.ascontiguousarray(pcm, dtype=np.float32)
        if pcm.ndim == 2:
            pcm = pcm[None]
        A, M, L = pcm.shape
        F = L // self.hop - 1
        if M != 2 or F < 1 or (F + 1) * self.hop != L:
            raise MCArrayHipError("pcm must be [A][2][(F+1)*hop]")
        idx = np.empty((A, F), dtype=np.int32)
        doa = np.empty((A, F), dtype=np.float32)
        prob = np.empty((A, F), dtype=np.float32)
        corr = np.empty((A, F, self.D), dtype=np.float32) if want_corr else None
        fp = _lib.c_fp
        self._check(self._lib.mca_hip_gcc2_frames_host(self.h, pcm.ctypes.data_as(fp), A, F, idx.ctypes.data_as(_lib.c_ip),
                                                       doa.ctypes.data_as(fp), prob.ctypes.data_as(fp),
                                                       corr.ctypes.data_as(fp) if want_corr else None))
        res = dict(argmax=idx, doa=doa, prob=prob, corr=corr)
        if self.use_power_floor:     # the gate of BinauralLocalisation.cpp:425-434: which frames fired, and the power handed to setDOA
            voiced = np.empty((A, F), dtype=np.uint8)
            power = np.empty((A, F), dtype=np.float32)
            self._check(self._lib.mca_hip_copy_gate(self.h, voiced.ctypes.data_as(C.c_void_p), power.ctypes.data_as(fp)))
            res["voiced"] = voiced
            res["power"] = power
        return res

    # ---- frame API ----
    def _rows(self, frames):
        frames = np.ascontiguousarray(frames, dtype=np.float64)
        if frames.shape[0] != self.M:
            raise MCArrayHipError("expected %d channel spectra" % self.M)
        arr = (_lib.c_dp * self.M)()
        for c in range(self.M):
            arr[c] = frames[c].ctypes.data_as(_lib.c_dp)
        return frames, arr

    def steering_process_frame(self, frames, n_sources=1):
        frames, arr = self._rows(frames)
        doa = np.empty(n_sources)
        prob = np.empty(n_sources)
        bins = np.empty(n_sources, dtype=np.int32)
        self._check(self._lib.mca_hip_steering_process_frame(self.h, arr, frames.shape[1], doa.ctypes.data_as(_lib.c_dp),
                                                             prob.ctypes.data_as(_lib.c_dp), bins.ctypes.data_as(_lib.c_ip),
                                                             n_sources))
        return doa, prob, bins

    def beamformer_process_frame(self, frames, doa):
        frames, arr = self._rows(frames)
        out = np.empty(frames.shape[1])
        self._check(self._lib.mca_hip_beamformer_process_frame(self.h, arr, frames.shape[1], out.ctypes.data_as(_lib.c_dp), float(doa)))
        return out

    def fft_log_power(self, frames):
        frames, arr = self._rows(frames)
        p = C.c_double(0)
        self._check(self._lib.mca_hip_fft_log_power(self.h, arr, frames.shape[1], C.byref(p)))
        return p.value

    def energy(self):
        out = np.empty(self.D)
        self._check(self._lib.mca_hip_get_energy(self.h, out.ctypes.data_as(_lib.c_dp)))
        return out

    # ---- measurement ----
    def set_timing(self, enable):
        self._check(self._lib.mca_hip_set_timing(self.h, int(enable)))

    def set_timing_kernels(self, kernel_ids):
        """event pairs around the launches of these kernel ids only (each pair costs the stream ~1.5 us)"""
        mask = 0
        for k in kernel_ids:
            mask |= 1 << int(k)
        self._check(self._lib.mca_hip_set_timing_mask(self.h, mask))

    def reset_timing(self):
        self._check(self._lib.mca_hip_reset_timing(self.h))

    def repair_stats(self):
        """SRP_ADAPTIVE: dict(frames, flagged, recomputed) since the last reset_timing()"""
        a, b, c_ = C.c_ulonglong(0), C.c_ulonglong(0), C.c_ulonglong(0)
        self._check(self._lib.mca_hip_get_repair_stats(self.h, C.byref(a), C.byref(b), C.byref(c_)))
        return {"frames": a.value, "flagged": b.value, "recomputed": c_.value}

    def repair_columns(self):
        """SRP_ADAPTIVE: dict(candidate_columns, whole_row_frames) since the last reset_timing() (mca_hip_get_repair_columns)"""
        a, b = C.c_ulonglong(0), C.c_ulonglong(0)
        self._check(self._lib.mca_hip_get_repair_columns(self.h, C.byref(a), C.byref(b)))
        return {"candidate_columns": a.value, "whole_row_frames": b.value}

    def get_timing(self, kernel_id):
        n = C.c_int(0)
        ms = C.c_double(0)
        self._check(self._lib.mca_hip_get_timing(self.h, kernel_id, C.byref(n), C.byref(ms)))
        return n.value, ms.value


class StreamGraph:
    """Handle of mca_hip_graph (include/mcarray_hip.h, real-time mode); keeps the buffers alive."""

    def __init__(self, ctx, h, keep):
        self.ctx, self.h, self._keep = ctx, h, keep
        self._lib = ctx._lib

    def launch(self, stream=None):
        rc = self.ctx._lib.mca_hip_graph_launch(self.h, stream)
        if rc != 0:     # (the context may be gone: its error string then lives in the library, not in the context)
            raise MCArrayHipError("libmcarray_hip error %d: %s" % (rc, self.ctx._lib.mca_hip_last_error(self.ctx.h).decode()))

    def close(self):
        # safe in either order: a context that is destroyed first orphans its graphs (their recordings go with it)
        if getattr(self, "h", None):
            self._lib.mca_hip_graph_destroy(self.h)
        self.h = None

    def __del__(self):
        self.close()


class SteeringBeamforming:
    """mca::SteeringBeamforming(int sampleRate, ArrayDescription, int fftCCSLength, unsigned nchannels)
    (SteeringBeamforming.h:43); processFrame (:54) returns (DOA[S] rad, prob[S], bin[S])."""

    def __init__(self, sample_rate, mic_positions, fft_ccs_length, nchannels=None, doa_step_deg=5.0, device=0):
        self.ctx = Context(sample_rate, mic_positions, fft_ccs_length - 2, doa_step_deg, n_sources=4, device=device)
        if nchannels is not None and nchannels != self.ctx.M:
            raise MCArrayHipError("nchannels does not match the array description")

    def process_frame(self, analysis_frames, n_sources=1):
        return self.ctx.steering_process_frame(analysis_frames, n_sources)


class Beamformer:
    """mca::Beamformer(int sampleRate, ArrayDescription, int fftCCSLength, unsigned nchannels) (Beamformer.h:39)."""

    def __init__(self, sample_rate, mic_positions, fft_ccs_length, nchannels=None, device=0):
        self.ctx = Context(sample_rate, mic_positions, fft_ccs_length - 2, device=device)

    def process_frame(self, analysis_frames, doa):
        return self.ctx.beamformer_process_frame(analysis_frames, doa)


def calculate_order_from_sample_rate(sample_rate, frame_seconds):
    """[BUILD-DEFINES] stand-in for dsp::STFT::calculateOrderFromSampleRate: the frame length the
    reference's stream modules derive from the sample rate, N = 2^order."""
    order = int(np.floor(np.log2(sample_rate * frame_seconds) + 0.5))
    return min(max(order, 8), 14)


class SourceSeparationAndLocalisation:
    """mca::SourceSeparationAndLocalisation(int sampleRate, ArrayDescription, unsigned numOfSources,
    bool usePowerFloor) (SourceSeparationAndLocalisation.h:47) driven over whole buffers: process()
    takes channel-major PCM, returns the beamformed audio and calls the callback once per frame
    like LocalisationCallback::setDOA(doaDegrees, prob, power, numOfSources) (SoundLocalisationCallback.h:53)."""

    FRAME_SECONDS = 0.025        # _frameRate (SourceSeparationAndLocalisation.h:60)

    def __init__(self, sample_rate, mic_positions, n_sources=1, use_power_floor=False, doa_step_deg=5.0, fft_size=None,
                 srp_precision=SRP_FP32, device=0):
        if fft_size is None:
            fft_size = 1 << calculate_order_from_sample_rate(sample_rate, self.FRAME_SECONDS)   # .cpp:52
        self.ctx = Context(sample_rate, mic_positions, fft_size, doa_step_deg, n_sources, use_power_floor, srp_precision, 1, device)
        self.callback = None

    def set_callback(self, cb):
        self.callback = cb

    def process(self, pcm):
        r = self.ctx.process_frames_host(np.asarray(pcm, dtype=np.float32)[None], want_energy=False, want_audio=True)
        if self.callback is not None:
            deg = r["doa"][0].astype(np.float64) * (180.0 / np.pi)   # toDegrees (microhponeArrayHelpers.cpp:91-98)
            for t in range(deg.shape[0]):
                if "voiced" in r and not r["voiced"][0, t]:
                    continue                                   # gated out: the reference does not call setDOA (:87-94)
                self.callback(deg[t], r["prob"][0, t], r["power"][0, t] if "power" in r else None, self.ctx.S)
        return r["out"][0], r


class SourceLocalisation(SourceSeparationAndLocalisation):
    """mca::SourceLocalisation(int sampleRate, ArrayDescription, unsigned numOfSources, bool usePowerFloor)
    (SourceLocalisation.h:43): the analysis-only sibling -- localisation and callbacks, no audio out
    (SourceLocalisation.cpp:63-79 calls processFrameLocalisation only)."""

    def process(self, pcm):
        r = self.ctx.process_frames_host(np.asarray(pcm, dtype=np.float32)[None], want_energy=False, want_audio=False)
        if self.callback is not None:
            deg = r["doa"][0].astype(np.float64) * (180.0 / np.pi)
            for t in range(deg.shape[0]):
                if "voiced" in r and not r["voiced"][0, t]:
                    continue
                self.callback(deg[t], r["prob"][0, t], r["power"][0, t] if "power" in r else None, self.ctx.S)
        return r


class FreqGCCBinauralLocalisation:
    """mca::FreqGCCBinauralLocalisation(int sampleRate, ArrayDescription, bool usePowerFloor)
    (BinauralLocalisation.h:191), deterministic part: smoothed GCC-PHAT correlation, first-max argmax,
    DOA smoothing and setProbability.  The reference's grid is 3 degrees (BinauralLocalisation.cpp:328)."""

    FRAME_SECONDS = 0.075        # _frameRate (BinauralLocalisation.h:196)

    def __init__(self, sample_rate, mic_positions, use_power_floor=False, doa_step_deg=3.0, fft_size=None,
                 srp_precision=SRP_FP32, max_arrays=1, device=0):
        if fft_size is None:
            fft_size = 1 << calculate_order_from_sample_rate(sample_rate, self.FRAME_SECONDS)
        self.ctx = Context(sample_rate, mic_positions, fft_size, doa_step_deg, 1, use_power_floor, srp_precision, max_arrays, device)
        if self.ctx.M != 2:
            raise MCArrayHipError("FreqGCCBinauralLocalisation needs exactly 2 microphones")
        self.callback = None

    def set_callback(self, cb):
        self.callback = cb

    def process(self, pcm, want_corr=False):
        """-> dict(argmax, doa, prob[, corr][, voiced, power]); the callback fires per frame of array 0 that passed the gate
        as setDOA(degrees, prob, power, 1) (BinauralLocalisation.cpp:521)."""
        r = self.ctx.gcc2_frames_host(pcm, want_corr)
        if self.callback is not None:
            for t in range(r["doa"].shape[1]):
                if "voiced" in r and not r["voiced"][0, t]:
                    continue
                self.callback(np.array([np.rad2deg(float(r["doa"][0, t]))]), np.array([r["prob"][0, t]]),
                              float(r["power"][0, t]) if "power" in r else 0.0, 1)
        return r


class _StateBlob:
    """state_save() / state_load() over mca_hip_<module>_state_* (checkpoint / resume of everything the module carries between calls)."""
    _STATE = None    # module infix of the C entry points

    def state_save(self):
        size = getattr(self._lib, "mca_hip_%s_state_size" % self._STATE)(self.h)
        if size < 0:
            self._check(int(size))
        blob = C.create_string_buffer(int(size))
        self._check(getattr(self._lib, "mca_hip_%s_state_save" % self._STATE)(self.h, blob, size))
        return blob.raw

    def state_load(self, blob):
        self._check(getattr(self._lib, "mca_hip_%s_state_load" % self._STATE)(self.h, blob, len(blob)))


class MultibandBinarualLocalisation(_StateBlob):
    _STATE = "mb"
    """mca::MultibandBinarualLocalisation(int sampleRate, ArrayDescription, int nbins = 15, bool usePowerFloor = 1)
    (MultibandBinarualLocalisation.h:38): per sub-band GCC-PHAT + energy-weighted DOA histogram over whole buffers."""

    FRAME_SECONDS = 0.025        # _frameRate (MultibandBinarualLocalisation.h:43)

    def __init__(self, sample_rate, mic_positions, nbins=15, use_power_floor=True, fft_size=None, max_arrays=1, device=0):
        self._lib = _lib.load()
        xyz = _xyz(mic_positions)
        if len(xyz) != 2:
            raise MCArrayHipError("MultibandBinarualLocalisation needs exactly 2 microphones")
        if fft_size is None:
            fft_size = 1 << calculate_order_from_sample_rate(sample_rate, self.FRAME_SECONDS)
        cfg = _lib.MbConfig()
        cfg.struct_size = C.sizeof(_lib.MbConfig)
        cfg.device = device
        cfg.sample_rate = sample_rate
        cfg.fft_size = fft_size
        cfg.mic_xyz = xyz.ctypes.data_as(_lib.c_dp)
        cfg.nbins = nbins
        cfg.use_power_floor = int(use_power_floor)
        cfg.max_arrays = max_arrays
        h = C.c_void_p()
        rc = self._lib.mca_hip_mb_create(C.byref(cfg), C.byref(h))
        if rc != 0:
            raise MCArrayHipError("mca_hip_mb_create failed (%d): %s" % (rc, self._lib.mca_hip_mb_last_error(None).decode()))
        self.h = h
        self.N, self.hop, self.nbins = fft_size, fft_size // 2, nbins
        self.D = self._lib.mca_hip_mb_num_steps(h)
        self.callback = None

    def close(self):
        if getattr(self, "h", None):
            self._lib.mca_hip_mb_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def _check(self, rc):
        if rc != 0:
            raise MCArrayHipError("libmcarray_hip error %d: %s" % (rc, self._lib.mca_hip_mb_last_error(self.h).decode()))

    def set_callback(self, cb):
        self.callback = cb

    def reset(self):
        self._check(self._lib.mca_hip_mb_reset(self.h, None))

    def filters(self):
        out = np.empty((self.nbins, self.N // 2 + 1))
        self._check(self._lib.mca_hip_mb_get_filters(self.h, out.ctypes.data_as(_lib.c_dp)))
        return out

    def process(self, pcm, want_bands=False):
        """pcm float32 [A][2][(F+1)*hop] -> dict(doa [A][F] rad, prob, voiced, power[, band_idx [A][F][nbins],
        energy_in_doa [A][F][D], band_corr [A][F][nbins][D]]); the callback fires per voiced frame of array 0
        as setDOA(degrees, prob, power, 1) (MultibandBinarualLocalisation.cpp:248)."""
        pcm = np.ascontiguousarray(pcm, dtype=np.float32)
        if pcm.ndim == 2:
            pcm = pcm[None]
        A, ch, L = pcm.shape
        F = L // self.hop - 1
        if ch != 2 or F < 1 or (F + 1) * self.hop != L:
            raise MCArrayHipError("pcm must be [A][2][(F+1)*hop]")
        doa = np.empty((A, F), dtype=np.float32)
        prob = np.empty((A, F), dtype=np.float32)
        voiced = np.empty((A, F), dtype=np.uint8)
        power = np.empty((A, F), dtype=np.float32)
        bi = np.empty((A, F, self.nbins), dtype=np.int32) if want_bands else None
        eid = np.empty((A, F, self.D), dtype=np.float32) if want_bands else None
        bc = np.empty((A, F, self.nbins, self.D), dtype=np.float32) if want_bands else None
        fp = _lib.c_fp
        self._check(self._lib.mca_hip_mb_frames_host(
            self.h, pcm.ctypes.data_as(fp), A, F, doa.ctypes.data_as(fp), prob.ctypes.data_as(fp),
            voiced.ctypes.data_as(C.c_void_p), power.ctypes.data_as(fp), bi.ctypes.data_as(_lib.c_ip) if want_bands else None,
            eid.ctypes.data_as(fp) if want_bands else None, bc.ctypes.data_as(fp) if want_bands else None))
        if self.callback is not None:
            for t in range(F):
                if voiced[0, t]:
                    self.callback(np.array([np.rad2deg(float(doa[0, t]))]), np.array([prob[0, t]]), float(power[0, t]), 1)
        return dict(doa=doa, prob=prob, voiced=voiced, power=power, band_idx=bi, energy_in_doa=eid, band_corr=bc)


class MvdrBeamformer(_StateBlob):
    _STATE = "mvdr"
    """Frequency-domain beamformer with a per-bin spatial covariance (BASELINE.json configs[3]; SURVEY A.9).
    No reference counterpart: the interface follows mca::Beamformer (Beamformer.h:39,49: frames in, one channel out,
    a look direction in radians) with the delay-and-sum weights replaced by MVDR weights."""

    K_ANALYSE, K_SOLVE, K_SYNTH = 0, 1, 2

    def __init__(self, sample_rate, mic_positions, fft_size=1024, alpha=0.95, loading=1e-3, max_streams=1, device=0):
        self._lib = _lib.load()
        xyz = _xyz(mic_positions)
        cfg = _lib.MvdrConfig()
        cfg.struct_size = C.sizeof(_lib.MvdrConfig)
        cfg.device = device
        cfg.sample_rate = sample_rate
        cfg.fft_size = fft_size
        cfg.n_mics = len(xyz)
        cfg.mic_xyz = xyz.ctypes.data_as(_lib.c_dp)
        cfg.alpha = alpha
        cfg.loading = loading
        cfg.max_streams = max_streams
        h = C.c_void_p()
        rc = self._lib.mca_hip_mvdr_create(C.byref(cfg), C.byref(h))
        if rc != 0:
            raise MCArrayHipError("mca_hip_mvdr_create failed (%d): %s" % (rc, self._lib.mca_hip_mvdr_last_error(None).decode()))
        self.h = h
        self.M, self.N, self.hop, self.K = len(xyz), fft_size, fft_size // 2, fft_size // 2 + 1

    def close(self):
        if getattr(self, "h", None):
            self._lib.mca_hip_mvdr_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def _check(self, rc):
        if rc != 0:
            raise MCArrayHipError("libmcarray_hip error %d: %s" % (rc, self._lib.mca_hip_mvdr_last_error(self.h).decode()))

    def reset(self):
        self._check(self._lib.mca_hip_mvdr_reset(self.h, None))

    def process(self, pcm, doa_rad, want_audio=True, want_spec=False):
        """pcm float32 [streams][M][(F+1)*hop], doa_rad [streams][F] (or a scalar) ->
        dict(out [streams][F*hop], spec complex64 [streams][F][K])"""
        pcm = np.ascontiguousarray(pcm, dtype=np.float32)
        if pcm.ndim == 2:
            pcm = pcm[None]
        A, M, L = pcm.shape
        F = L // self.hop - 1
        if M != self.M or F < 1 or (F + 1) * self.hop != L:
            raise MCArrayHipError("pcm must be [streams][M][(F+1)*hop]")
        doa = np.ascontiguousarray(np.broadcast_to(np.asarray(doa_rad, dtype=np.float32), (A, F)))
        out = np.empty((A, F * self.hop), dtype=np.float32) if want_audio else None
        spec = np.empty((A, F, self.K), dtype=np.complex64) if want_spec else None
        fp = _lib.c_fp
        self._check(self._lib.mca_hip_mvdr_frames_host(
            self.h, pcm.ctypes.data_as(fp), A, F, doa.ctypes.data_as(fp), out.ctypes.data_as(fp) if want_audio else None,
            spec.ctypes.data_as(fp) if want_spec else None))
        return dict(out=out, spec=spec)

    def process_dev(self, pcm, n_frames, doa_rad, out_pcm=None, out_spec=None, stream=None):
        """device tensors (torch, contiguous): pcm [streams][M][>= (F+1)*hop] float32, doa_rad [streams][F] float32,
        out_pcm [streams][F*hop], out_spec [streams][F][K][2]; asynchronous on `stream` (a raw hipStream_t or None)."""
        A = pcm.shape[0]
        self._check(self._lib.mca_hip_mvdr_frames_dev(
            self.h, pcm.data_ptr(), pcm.stride(0), pcm.stride(1), A, n_frames, doa_rad.data_ptr(),
            out_pcm.data_ptr() if out_pcm is not None else None, out_spec.data_ptr() if out_spec is not None else None, stream))

    def covariance(self, stream_index=0):
        out = np.empty((self.K, self.M, self.M, 2))
        self._check(self._lib.mca_hip_mvdr_get_covariance(self.h, stream_index, out.ctypes.data_as(_lib.c_dp)))
        return out[..., 0] + 1j * out[..., 1]

    def set_timing(self, enable):
        self._check(self._lib.mca_hip_mvdr_set_timing(self.h, int(enable)))

    def repair_stats(self):
        """SRP_ADAPTIVE: dict(frames, flagged, recomputed) since the last reset_timing()"""
        a, b, c_ = C.c_ulonglong(0), C.c_ulonglong(0), C.c_ulonglong(0)
        self._check(self._lib.mca_hip_get_repair_stats(self.h, C.byref(a), C.byref(b), C.byref(c_)))
        return {"frames": a.value, "flagged": b.value, "recomputed": c_.value}

    def get_timing(self, kernel_id):
        n, ms = C.c_int(0), C.c_double(0)
        self._check(self._lib.mca_hip_mvdr_get_timing(self.h, kernel_id, C.byref(n), C.byref(ms)))
        return n.value, ms.value


FACTOR, RELATIVE, FULL, NOISY, NOTHING = 0, 1, 3, 4, 5      # BinauralMasking::MaskingMethod (ArrayModules.h:81)
BOTH, SPATIAL, TEMPORAL = 0, 1, 2                           # BinauralMasking::MaskingAlg (ArrayModules.h:89)


class FastBinauralMasking(_StateBlob):
    _STATE = "mask"
    """mca::FastBinauralMasking(int samplerate, double microDistance, float lowFreq, float highFreq,
    MaskingMethod = RELATIVE, MaskingAlg = BOTH) (FastBinauralMasking.h:71-76)."""

    def __init__(self, samplerate, micro_distance, low_freq, high_freq, method=RELATIVE, algorithm=BOTH, fft_size=1024,
                 max_streams=1, device=0):
        self._lib = _lib.load()
        cfg = _lib.MaskConfig()
        cfg.struct_size = C.sizeof(_lib.MaskConfig)
        cfg.device = device
        cfg.sample_rate = samplerate
        cfg.fft_size = fft_size
        cfg.micro_distance = micro_distance
        cfg.low_freq = low_freq
        cfg.high_freq = high_freq
        cfg.method = method
        cfg.algorithm = algorithm
        cfg.max_streams = max_streams
        h = C.c_void_p()
        rc = self._lib.mca_hip_mask_create(C.byref(cfg), C.byref(h))
        if rc != 0:
            raise MCArrayHipError("mca_hip_mask_create failed (%d): %s" % (rc, self._lib.mca_hip_mask_last_error(None).decode()))
        self.h = h
        self.N = fft_size
        self.hop = fft_size // 2

    def close(self):
        if getattr(self, "h", None):
            self._lib.mca_hip_mask_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def _check(self, rc):
        if rc != 0:
            raise MCArrayHipError("libmcarray_hip error %d: %s" % (rc, self._lib.mca_hip_mask_last_error(self.h).decode()))

    def reset(self):
        self._check(self._lib.mca_hip_mask_reset(self.h))

    def thresholds(self):
        thr = np.empty(45)
        cen = np.empty(45)
        self._check(self._lib.mca_hip_mask_get_thresholds(self.h, thr.ctypes.data_as(_lib.c_dp), cen.ctypes.data_as(_lib.c_dp)))
        return thr, cen

    def process(self, pcm, want_decisions=True):
        """pcm float32 [streams][2][(F+1)*hop] -> (out [streams][2][F*hop], decisions [streams][F][45])"""
        pcm = np.ascontiguousarray(pcm, dtype=np.float32)
        if pcm.ndim == 2:
            pcm = pcm[None]
        ns, ch, L = pcm.shape
        F = L // self.hop - 1
        if ch != 2 or F < 1 or (F + 1) * self.hop != L:
            raise MCArrayHipError("pcm must be [streams][2][(F+1)*hop]")
        out = np.empty((ns, 2, F * self.hop), dtype=np.float32)
        dec = np.empty((ns, F, 45), dtype=np.int32) if want_decisions else None
        self._check(self._lib.mca_hip_mask_frames_host(self.h, pcm.ctypes.data_as(_lib.c_fp), ns, F, out.ctypes.data_as(_lib.c_fp),
                                                       dec.ctypes.data_as(_lib.c_ip) if want_decisions else None))
        return out, dec

    def process_parametrisation(self, left, right):
        """The DSPONE hook for one frame: CCS double[N+2] spectra, returns the modified copies + decisions."""
        left = np.array(left, dtype=np.float64, order="C")
        right = np.array(right, dtype=np.float64, order="C")
        dec = np.zeros(45, dtype=np.int32)
        self._check(self._lib.mca_hip_mask_process_frame(self.h, left.ctypes.data_as(_lib.c_dp), right.ctypes.data_as(_lib.c_dp),
                                                         len(left), dec.ctypes.data_as(_lib.c_ip)))
        return left, right, dec
