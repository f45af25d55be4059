"""CPU: the C-ABI library loads and exports every symbol include/mcarray_hip.h declares
(no compute calls without a GPU), and fails loudly when no device is present."""
import ctypes as C
import os
import re

import pytest

from mcarray_amd import _lib, api

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "mcarray_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mca_hip_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_agree():
    declared = _declared_symbols()
    bound = sorted(name for name, _, _ in _lib.SYMBOLS)
    assert declared == bound


def test_library_exports_every_declared_symbol():
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    lib = C.CDLL(_lib.LIB_PATH)
    for name in _declared_symbols():
        assert hasattr(lib, name), name
    assert b"gfx950" in _lib.load().mca_hip_version()


def test_create_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(api.MCArrayHipError, match="no CPU fallback"):
        api.Context(48000, [0.0, 0.1, 0.2], 1024)


def test_product_package_does_not_import_oracle():
    # the oracle is test infrastructure: nothing under mcarray_amd/ may reference it
    pkg = os.path.join(ROOT, "mcarray_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in src.lower().replace("test oracle", ""), os.path.join(dirpath, f)


def test_measurement_switches_are_not_in_the_shipped_library():
    """the A/B switches behind DESIGN.md's measurements (one of them, MCA_HIP_BFW_ABL, returns wrong audio on
    purpose) are compiled in only with `make MEASURE=1`; the shipped library knows the ten product switches, reads them once in
    mca_hip_create (csrc/knobs.h is the only place that touches the environment) and nothing else."""
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    blob = open(os.path.join(root, "mcarray_amd", "libmcarray_hip.so"), "rb").read()
    names = set(m.decode() for m in re.findall(rb"MCA_HIP_[A-Z0-9_]{3,}", blob))
    switches = {n for n in names if not re.match(r"MCA_HIP_(OK|ERR_|SRP_|GCC_|K_|ADAPT_FALLBACK_(AUTO|OFF))", n)}
    assert switches == {"MCA_HIP_ADAPT_CAND", "MCA_HIP_ADAPT_FALLBACK", "MCA_HIP_ADAPT_LAZY", "MCA_HIP_ADAPT_MAX_SOURCES", "MCA_HIP_ADAPT_MIN_ROWS", "MCA_HIP_ADAPT_TAU_SCALE",
                        "MCA_HIP_FORCE_GENERIC", "MCA_HIP_NO_N2048", "MCA_HIP_SCAN_CARRY", "MCA_HIP_WS_MAX_MB"}, switches
    src = os.path.join(root, "mcarray_amd", "csrc")
    for f in os.listdir(src):
        if f.endswith((".hip", ".h")) and f != "knobs.h":
            assert "getenv" not in open(os.path.join(src, f)).read(), f


def test_no_packed_fp32_instruction_takes_the_high_half_of_src1_into_the_low_result():
    """fft512.h RULE / DESIGN.md section 7: that operand path is not sound beside an MFMA + LDS neighbour on this pool's MI355X.
    The lint disassembles every device code object of the built library (no GPU needed)."""
    import importlib.util
    import shutil
    if shutil.which("/opt/rocm/lib/llvm/bin/llvm-objdump") is None:
        pytest.skip("no llvm-objdump in this image")
    spec = importlib.util.spec_from_file_location("check_isa", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "check_isa.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    bad = mod.offenders(_lib.LIB_PATH)
    assert not bad, {k: len(v) for k, v in bad.items()}

