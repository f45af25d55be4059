"""tools/mcbeam_multi.cpp -- the C++ multi-device driver behind the C ABI (BASELINE configs[4]): its sharding of the arrays
(include/mcarray/Partition.h) must be the one bench.py / mcarray_amd/dist.py use; on the GPU box the driver runs end to end on
the one visible device."""
import os
import subprocess

import pytest

from mcarray_amd import dist as mdist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "cxx", "mcbeam_multi")


def _build():
    if not os.path.exists(os.path.join(ROOT, "mcarray_amd", "libmcarray_hip.so")):
        import __graft_entry__ as g
        g.build()
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "tests", "cxx"), "-s", "mcbeam_multi"])


@pytest.mark.parametrize("n_arrays,world", [(1024, 8), (10, 3), (7, 8), (128, 1), (1000, 6)])
def test_cxx_partition_equals_python_partition(n_arrays, world):
    _build()
    r = subprocess.run([EXE, "--devices", str(world), "--arrays", str(n_arrays), "--partition"], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, r.stderr
    rows = [tuple(int(v) for v in line.split()) for line in r.stdout.strip().splitlines()]
    assert len(rows) == world
    for rank, first, count in rows:
        want = mdist.local_range(n_arrays, rank, world)
        assert (first, count) == (want.start, len(want))
    assert sum(c for _, _, c in rows) == n_arrays


@pytest.mark.gpu
def test_cxx_multi_device_driver_runs_on_the_visible_device():
    _build()
    r = subprocess.run([EXE, "--devices", "1", "--arrays", "6", "--frames", "128", "--steps", "2", "--audio"], capture_output=True, text=True, timeout=600)
    print(r.stdout, r.stderr)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "0 arrays off their source" in r.stdout
