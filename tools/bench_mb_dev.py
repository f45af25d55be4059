"""Device-side throughput of the multiband 2-microphone localiser (mca_hip_mb_frames_dev) on buffers resident in HBM.
usage (GPU box): python tools/bench_mb_dev.py"""
import ctypes as C
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mcarray_amd import _lib, api, synth  # noqa: E402

lib = _lib.load()
A, F, fs = 64, 1024, 48000
m = api.MultibandBinarualLocalisation(fs, synth.BINAURAL, 15, False, max_arrays=A)
dev = torch.device("cuda", 0)
L = (F + 1) * m.hop
x = (torch.randn(A, 2, L, device=dev) * 0.1).contiguous()
doa = torch.empty(A, F, dtype=torch.float32, device=dev)
prob = torch.empty(A, F, dtype=torch.float32, device=dev)
st = torch.cuda.current_stream().cuda_stream
p = lambda t: C.c_void_p(t.data_ptr())


def call():
    rc = lib.mca_hip_mb_frames_dev(m.h, p(x), 2 * L, L, A, F, p(doa), p(prob), None, None, None, None, None, st)
    assert rc == 0, lib.mca_hip_mb_last_error(m.h)


for _ in range(3):
    call()
torch.cuda.synchronize()
t0 = time.perf_counter(); n = 10
for _ in range(n):
    call()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
print("multiband localiser: %d arrays x %d frames, N=%d, 15 bands: %.3f ms per call, %.1f M frames/s" % (A, F, m.N, dt * 1e3, A * F / dt / 1e6))
