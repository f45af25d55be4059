"""bench.py's launch contract: `--gpus N` must never report an N-GPU number from fewer ranks, and the RCCL exchange
(mcarray_amd/dist.py StepGather) must have run on hardware at least once (single-rank nccl group on the 1-GPU box)."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _clean_env():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    return env


def _refuses_two_gpus():
    p = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "1", "--warmup", "0"], env=_clean_env(),
                       capture_output=True, text=True, timeout=600)
    assert p.returncode != 0
    assert "GPU(s) visible" in p.stderr and "--gpus 2" in p.stderr, p.stderr
    assert not any(line.startswith("{") for line in p.stdout.splitlines())          # and no JSON line that could be mistaken for a result


def test_bench_refuses_more_gpus_than_visible_cpu():
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("this node has 2+ GPUs")
    _refuses_two_gpus()


@pytest.mark.gpu
def test_bench_refuses_more_gpus_than_visible_on_the_gpu_box():
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("this node has 2+ GPUs")
    _refuses_two_gpus()


def test_bench_rejects_launcher_mismatch():
    env = _clean_env()
    env.update(WORLD_SIZE="2", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999")
    p = subprocess.run([sys.executable, BENCH, "--gpus", "4"], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode != 0 and "WORLD_SIZE=2" in p.stderr


@pytest.mark.gpu
def test_bench_single_rank_rccl_exchange():
    """One step of the real bench loop with a single-rank `nccl` process group in a fresh process: RCCL is initialised on
    the hardware and the packed DOA all_gather + the audio gather of StepGather run on it."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = _clean_env()
    env.update(WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), MCA_BENCH_FORCE_DIST="1")
    p = subprocess.run([sys.executable, BENCH, "--gpus", "1", "--steps", "3", "--warmup", "2", "--arrays", "2", "--frames", "512",
                        "--cpu-frames", "0", "--gather-audio"], env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [line for line in p.stdout.splitlines() if line.startswith("{")]
    assert len(lines) == 1
    r = json.loads(lines[0])
    assert r["n_gpus"] == 1 and r["value"] > 0 and r["exchange"]["backend"] == "nccl" and r["exchange"]["gather_audio"] is True


def test_the_printed_line_is_compact_and_complete():
    """The driver reads the END of bench.py's stdout: the line it prints must stay well under 6 KB and still carry every field
    of the contract, `roofline` (with `traffic`) and `cpu_baseline`, and the three other BASELINE configs measured in the same run.  Checked on
    the committed full record of the driver's command (profiles/r06_bench_driver_cmd_detail.json) -- no GPU needed."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench", BENCH)
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    full = json.load(open(os.path.join(ROOT, "profiles", "r06_bench_driver_cmd_detail.json")))
    line = bench.compact_line(full)
    text = json.dumps(line)
    assert len(text) < 6000, len(text)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"):
        assert k in line, k
    assert line["vs_baseline"] is None and line["value"] == full["value"] and line["ms_per_step"] == full["ms_per_step"]
    assert "model" not in line["config"] and line["config"]["workload"].startswith("BASELINE configs[2]")
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in line["roofline"], k
    assert line["roofline"]["bound"] == "hbm" and abs(line["roofline"]["frac"] - line["roofline"]["achieved"] / line["roofline"]["peak"]) < 1e-3
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in line["cpu_baseline"], k
    for k in ("single_stream_4096", "das_single_stream", "mvdr_256x64"):
        assert line["config"][k]["value" if k != "das_single_stream" else "offline_any_angle"], k
    gemm = [e for e in line["roofline_by_kernel"] if e["kernel"] == "k_srp_gemm"][0]
    assert 0 < gemm["mfma_pipe_frac"] < gemm["frac"] < 1
