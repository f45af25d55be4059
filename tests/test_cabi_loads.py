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
