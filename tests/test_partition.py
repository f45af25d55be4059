"""tools/mcbeam_multi.cpp -- the C++ multi-GPU driver behind the C ABI (BASELINE configs[4]): one process per GPU, device-pointer
calls, ONE ncclAllGather of the packed DOA buffer per step (RCCL), audio to rank 0 by ncclSend / ncclRecv -- the north_star's "RCCL
over xGMI used only to gather DOA/output buffers" in the host language of the reference (its own driver: src/programs/mcabeamf.cpp:77-122).
Its sharding of the arrays (include/mcarray/Partition.h) must be the one bench.py / mcarray_amd/dist.py use; on the GPU box the driver
runs end to end with world size 1 (the box has one GPU) and its gathered bins must equal the unsharded call's.  N > 1: unmeasured on
hardware."""
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
@pytest.mark.parametrize("precision", ["adaptive", "fp32"])
def test_cxx_rccl_driver_gathers_what_the_unsharded_call_returns(precision):
    """world size 1 on the box's one GPU: the parent spawns the rank as a fresh process, the rank runs three steps through
    mca_hip_process_frames_dev, all-gathers the packed [2][n][F] buffer (RCCL), sends its audio to rank 0 (itself), and --check compares
    every gathered (bin, probability) pair with the unsharded call on a second context: all equal"""
    _build()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([EXE, "--ranks", "1", "--arrays", "6", "--frames", "128", "--steps", "3", "--audio", "--check", "--precision", precision],
                       capture_output=True, text=True, timeout=600, env=env)
    print(r.stdout, r.stderr)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "0 arrays off their source" in r.stdout
    assert "0 of 768 gathered (bin, prob) pairs differ from the unsharded call" in r.stdout
    assert "RCCL all-gather" in r.stdout and "gathered audio rms" in r.stdout


def test_cxx_rccl_driver_refuses_more_ranks_than_gpus():
    """(runs here without a GPU too: every rank wants a device of its own -- two ranks on one GPU is not a configuration RCCL accepts)"""
    _build()
    r = subprocess.run([EXE, "--ranks", "2", "--arrays", "4", "--frames", "64", "--steps", "1"], capture_output=True, text=True, timeout=120)
    assert r.returncode != 0
