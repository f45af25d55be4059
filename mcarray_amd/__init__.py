"""mcarray_amd -- MI355X-native implementation of mcarray's localisation + beamforming hot path.

Only what the path needs: csrc/ (HIP kernels + the C ABI of include/mcarray_hip.h), the ctypes
binding (_lib), the Python mirror of the reference's module API (api) and the synthetic input
generator (synth).  Importing the package does not need a GPU; creating a Context does.
"""
from . import synth  # noqa: F401
from ._lib import LIB_PATH, MCArrayHipError  # noqa: F401

__all__ = ["synth", "LIB_PATH", "MCArrayHipError"]
