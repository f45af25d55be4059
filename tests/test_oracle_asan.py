"""The CPU oracle under AddressSanitizer + UndefinedBehaviorSanitizer: `make -C oracle asan` builds libmca_oracle_asan.so from
the same source, and the golden-vector tests of the oracle run against it in a child process (the sanitizer runtime has to be
preloaded into the interpreter).  CPU only -- GPU sanitizers are not available on this pool."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _runtime(name):
    try:
        p = subprocess.check_output(["gcc", "-print-file-name=" + name], text=True).strip()
    except (OSError, subprocess.CalledProcessError):
        return None
    return p if os.path.isabs(p) and os.path.exists(p) else None


def test_oracle_golden_vectors_under_asan_ubsan():
    asan, ubsan = _runtime("libasan.so"), _runtime("libubsan.so")
    if not asan:
        pytest.skip("no libasan.so next to gcc")
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "asan"])
    so = os.path.join(ROOT, "oracle", "libmca_oracle_asan.so")
    assert os.path.exists(so)
    env = dict(os.environ)
    env["LD_PRELOAD"] = asan + ((":" + ubsan) if ubsan else "")
    env["ASAN_OPTIONS"] = "detect_leaks=0:abort_on_error=1"          # (CPython's own allocations are not ours to judge)
    env["UBSAN_OPTIONS"] = "halt_on_error=1:print_stacktrace=1"
    env["MCA_ORACLE_LIB"] = so
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider", os.path.join(ROOT, "tests", "test_oracle_golden.py")],
                       env=env, cwd=ROOT, capture_output=True, text=True, timeout=1500)
    tail = (r.stdout + r.stderr)[-3000:]
    assert r.returncode == 0, tail
    assert "AddressSanitizer" not in tail and "runtime error" not in tail, tail
