"""ctypes binding of libmcarray_hip.so (the C-ABI of include/mcarray_hip.h).

The product path: if the HIP library is missing or no gfx950 device is visible this
module raises -- there is no CPU or PyTorch fallback anywhere in the package.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MCA_HIP_LIB", os.path.join(_HERE, "libmcarray_hip.so"))   # MCA_HIP_LIB: A/B builds side by side

c_dp = C.POINTER(C.c_double)
c_fp = C.POINTER(C.c_float)
c_ip = C.POINTER(C.c_int)


class MCArrayHipError(RuntimeError):
    """Raised for every non-zero status of the C ABI (the C++ side throws mca::MCArrayException)."""


class Config(C.Structure):
    _fields_ = [
        ("struct_size", C.c_int),
        ("device", C.c_int),
        ("sample_rate", C.c_int),
        ("fft_size", C.c_int),
        ("n_mics", C.c_int),
        ("mic_xyz", c_dp),
        ("doa_step_deg", C.c_double),
        ("n_sources", C.c_int),
        ("use_power_floor", C.c_int),
        ("srp_precision", C.c_int),
        ("max_arrays", C.c_int),
        ("gcc_weighting", C.c_int),
        ("adaptive_fallback", C.c_int),
        ("adaptive_min_rows", C.c_int),
        ("adaptive_max_sources", C.c_int),
        ("scan_carry", C.c_int),
    ]


class MaskConfig(C.Structure):
    _fields_ = [
        ("struct_size", C.c_int),
        ("device", C.c_int),
        ("sample_rate", C.c_int),
        ("fft_size", C.c_int),
        ("micro_distance", C.c_double),
        ("low_freq", C.c_float),
        ("high_freq", C.c_float),
        ("method", C.c_int),
        ("algorithm", C.c_int),
        ("max_streams", C.c_int),
    ]


class MbConfig(C.Structure):
    _fields_ = [
        ("struct_size", C.c_int),
        ("device", C.c_int),
        ("sample_rate", C.c_int),
        ("fft_size", C.c_int),
        ("mic_xyz", c_dp),
        ("nbins", C.c_int),
        ("use_power_floor", C.c_int),
        ("max_arrays", C.c_int),
    ]


class MvdrConfig(C.Structure):
    _fields_ = [
        ("struct_size", C.c_int),
        ("device", C.c_int),
        ("sample_rate", C.c_int),
        ("fft_size", C.c_int),
        ("n_mics", C.c_int),
        ("mic_xyz", c_dp),
        ("alpha", C.c_double),
        ("loading", C.c_double),
        ("max_streams", C.c_int),
    ]


# every symbol include/mcarray_hip.h declares: (name, restype, argtypes)
SYMBOLS = [
    ("mca_hip_create", C.c_int, [C.POINTER(Config), C.POINTER(C.c_void_p)]),
    ("mca_hip_destroy", None, [C.c_void_p]),
    ("mca_hip_last_error", C.c_char_p, [C.c_void_p]),
    ("mca_hip_num_steps", C.c_int, [C.c_void_p]),
    ("mca_hip_num_pairs", C.c_int, [C.c_void_p]),
    ("mca_hip_num_groups", C.c_int, [C.c_void_p]),
    ("mca_hip_get_pair_delays", C.c_int, [C.c_void_p, c_fp]),
    ("mca_hip_get_doa_grid", C.c_int, [C.c_void_p, c_fp]),
    ("mca_hip_reset", C.c_int, [C.c_void_p, C.c_void_p]),
    ("mca_hip_reserve", C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    ("mca_hip_state_size", C.c_longlong, [C.c_void_p]),
    ("mca_hip_state_save", C.c_int, [C.c_void_p, C.c_void_p, C.c_longlong]),
    ("mca_hip_state_load", C.c_int, [C.c_void_p, C.c_void_p, C.c_longlong]),
    ("mca_hip_localise_frames_dev", C.c_int,
     [C.c_void_p, C.c_void_p, C.c_longlong, C.c_longlong, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
      C.c_void_p, C.c_void_p]),
    ("mca_hip_separate_frames_dev", C.c_int,
     [C.c_void_p, C.c_void_p, C.c_longlong, C.c_longlong, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    ("mca_hip_separate_frames_bins_dev", C.c_int,
     [C.c_void_p, C.c_void_p, C.c_longlong, C.c_longlong, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    ("mca_hip_process_frames_dev", C.c_int,
     [C.c_void_p, C.c_void_p, C.c_longlong, C.c_longlong, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
      C.c_void_p, C.c_void_p, C.c_void_p]),
    ("mca_hip_copy_gate", C.c_int, [C.c_void_p, C.c_void_p, c_fp]),
    ("mca_hip_process_frames_host", C.c_int,
     [C.c_void_p, c_fp, C.c_int, C.c_int, c_ip, c_fp, c_fp, c_fp, c_fp]),
    ("mca_hip_steering_process_frame", C.c_int,
     [C.c_void_p, C.POINTER(c_dp), C.c_int, c_dp, c_dp, c_ip, C.c_int]),
    ("mca_hip_beamformer_process_frame", C.c_int, [C.c_void_p, C.POINTER(c_dp), C.c_int, c_dp, C.c_double]),
    ("mca_hip_fft_log_power", C.c_int, [C.c_void_p, C.POINTER(c_dp), C.c_int, c_dp]),
    ("mca_hip_get_energy", C.c_int, [C.c_void_p, c_dp]),
    ("mca_hip_gcc2_frames_dev", C.c_int,
     [C.c_void_p, C.c_void_p, C.c_longlong, C.c_longlong, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
      C.c_void_p, C.c_void_p]),
    ("mca_hip_gcc2_frames_host", C.c_int, [C.c_void_p, c_fp, C.c_int, C.c_int, c_ip, c_fp, c_fp, c_fp]),
    ("mca_hip_mask_create", C.c_int, [C.POINTER(MaskConfig), C.POINTER(C.c_void_p)]),
    ("mca_hip_mask_destroy", None, [C.c_void_p]),
    ("mca_hip_mask_last_error", C.c_char_p, [C.c_void_p]),
    ("mca_hip_mask_reset", C.c_int, [C.c_void_p]),
    ("mca_hip_mask_get_thresholds", C.c_int, [C.c_void_p, c_dp, c_dp]),
    ("mca_hip_mask_frames_dev", C.c_int,
     [C.c_void_p, C.c_void_p, C.c_longlong, C.c_longlong, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    ("mca_hip_mask_frames_host", C.c_int, [C.c_void_p, c_fp, C.c_int, C.c_int, c_fp, c_ip]),
    ("mca_hip_mask_process_frame", C.c_int, [C.c_void_p, c_dp, c_dp, C.c_int, c_ip]),
    ("mca_hip_mb_create", C.c_int, [C.POINTER(MbConfig), C.POINTER(C.c_void_p)]),
    ("mca_hip_mb_destroy", None, [C.c_void_p]),
    ("mca_hip_mb_last_error", C.c_char_p, [C.c_void_p]),
    ("mca_hip_mb_reset", C.c_int, [C.c_void_p, C.c_void_p]),
    ("mca_hip_mb_num_steps", C.c_int, [C.c_void_p]),
    ("mca_hip_mb_get_filters", C.c_int, [C.c_void_p, c_dp]),
    ("mca_hip_mb_frames_dev", C.c_int,
     [C.c_void_p, C.c_void_p, C.c_longlong, C.c_longlong, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
      C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    ("mca_hip_mb_frames_host", C.c_int,
     [C.c_void_p, c_fp, C.c_int, C.c_int, c_fp, c_fp, C.c_void_p, c_fp, c_ip, c_fp, c_fp]),
    ("mca_hip_process_frames_host_i16", C.c_int, [C.c_void_p, C.POINTER(C.c_short), C.c_int, C.c_int, c_ip, c_fp, c_fp, c_fp, c_fp]),
    ("mca_hip_graph_create", C.c_int,
     [C.c_void_p, C.c_void_p, C.c_longlong, C.c_longlong, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
      C.POINTER(C.c_void_p)]),
    ("mca_hip_graph_launch", C.c_int, [C.c_void_p, C.c_void_p]),
    ("mca_hip_graph_destroy", None, [C.c_void_p]),
    ("mca_hip_mvdr_create", C.c_int, [C.POINTER(MvdrConfig), C.POINTER(C.c_void_p)]),
    ("mca_hip_mvdr_destroy", None, [C.c_void_p]),
    ("mca_hip_mvdr_last_error", C.c_char_p, [C.c_void_p]),
    ("mca_hip_mvdr_reset", C.c_int, [C.c_void_p, C.c_void_p]),
    ("mca_hip_mvdr_frames_dev", C.c_int,
     [C.c_void_p, C.c_void_p, C.c_longlong, C.c_longlong, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    ("mca_hip_mvdr_frames_host", C.c_int, [C.c_void_p, c_fp, C.c_int, C.c_int, c_fp, c_fp, c_fp]),
    ("mca_hip_mvdr_get_covariance", C.c_int, [C.c_void_p, C.c_int, c_dp]),
    ("mca_hip_mvdr_set_timing", C.c_int, [C.c_void_p, C.c_int]),
    ("mca_hip_mvdr_get_timing", C.c_int, [C.c_void_p, C.c_int, c_ip, c_dp]),
    ("mca_hip_mask_state_size", C.c_longlong, [C.c_void_p]),
    ("mca_hip_mask_state_save", C.c_int, [C.c_void_p, C.c_void_p, C.c_longlong]),
    ("mca_hip_mask_state_load", C.c_int, [C.c_void_p, C.c_void_p, C.c_longlong]),
    ("mca_hip_mb_state_size", C.c_longlong, [C.c_void_p]),
    ("mca_hip_mb_state_save", C.c_int, [C.c_void_p, C.c_void_p, C.c_longlong]),
    ("mca_hip_mb_state_load", C.c_int, [C.c_void_p, C.c_void_p, C.c_longlong]),
    ("mca_hip_mvdr_state_size", C.c_longlong, [C.c_void_p]),
    ("mca_hip_mvdr_state_save", C.c_int, [C.c_void_p, C.c_void_p, C.c_longlong]),
    ("mca_hip_mvdr_state_load", C.c_int, [C.c_void_p, C.c_void_p, C.c_longlong]),
    ("mca_hip_set_timing", C.c_int, [C.c_void_p, C.c_int]),
    ("mca_hip_set_timing_mask", C.c_int, [C.c_void_p, C.c_uint]),
    ("mca_hip_get_timing", C.c_int, [C.c_void_p, C.c_int, c_ip, c_dp]),
    ("mca_hip_reset_timing", C.c_int, [C.c_void_p]),
    ("mca_hip_host_alloc", C.c_void_p, [C.c_longlong]),
    ("mca_hip_host_free", None, [C.c_void_p]),
    ("mca_hip_host_register", C.c_int, [C.c_void_p, C.c_longlong]),
    ("mca_hip_host_unregister", C.c_int, [C.c_void_p]),
    ("mca_hip_get_repair_stats", C.c_int, [C.c_void_p, C.POINTER(C.c_ulonglong), C.POINTER(C.c_ulonglong), C.POINTER(C.c_ulonglong)]),
    ("mca_hip_get_repair_columns", C.c_int, [C.c_void_p, C.POINTER(C.c_ulonglong), C.POINTER(C.c_ulonglong)]),
    ("mca_hip_version", C.c_char_p, []),
]

_LIB = None


def _pin_single_hip_runtime():
    """A process must hold ONE HIP/HSA runtime.  PyTorch-ROCm wheels bundle their own copy of
    libamdhip64 (same SONAME as /opt/rocm's); if this library pulled in the system copy first and
    torch were imported later, the second runtime would find no device.  So when torch is installed
    its copy is loaded first (without importing torch) and libmcarray_hip.so binds to it by SONAME."""
    import importlib.util
    spec = importlib.util.find_spec("torch")
    if spec is None or not spec.submodule_search_locations:
        return
    cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(cand):
        C.CDLL(cand, mode=C.RTLD_GLOBAL)


def load():
    """Load libmcarray_hip.so; raises if it has not been built (python -c 'import __graft_entry__ as g; g.build()')."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise MCArrayHipError(
                "%s not found: the HIP extension is not built (run __graft_entry__.build()); "
                "mcarray_amd has no CPU fallback" % LIB_PATH)
        _pin_single_hip_runtime()
        lib = C.CDLL(LIB_PATH)
        for name, res, args in SYMBOLS:
            fn = getattr(lib, name)   # AttributeError if the library lacks a declared symbol
            fn.restype = res
            fn.argtypes = args
        _LIB = lib
    return _LIB
