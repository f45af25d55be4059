"""Runs the C++ module-API test program (tests/cxx/test_mcarray_api.cpp, modelled on the reference's
test/test_mcarray.cpp).  CPU: the known-answer ArrayDescription test; GPU: everything."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "cxx", "test_mcarray_api")


def _build():
    if not os.path.exists(os.path.join(ROOT, "mcarray_amd", "libmcarray_hip.so")):
        import __graft_entry__ as g
        g.build()
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "tests", "cxx"), "-s"])


def test_cxx_array_description_known_answers():
    _build()
    r = subprocess.run([EXE, "--cpu"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "ALL PASSED" in r.stdout


@pytest.mark.gpu
def test_cxx_module_api_on_gpu():
    if not os.path.exists(EXE):
        _build()
    r = subprocess.run([EXE], capture_output=True, text=True, timeout=600)
    print(r.stdout)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "ALL PASSED" in r.stdout
