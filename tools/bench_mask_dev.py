import os, sys, time, ctypes as C
sys.path.insert(0, os.getcwd())
import torch, numpy as np
from mcarray_amd import api, _lib
lib = _lib.load()
A, F = 64, 1024
m = api.FastBinauralMasking(16000, 0.086, 500.0, 5000.0, max_streams=A)
dev = torch.device("cuda", 0)
x = (torch.randn(A, 2, (F + 1) * 512, device=dev) * 0.1).contiguous()
out = torch.empty(A, 2, F * 512, device=dev)
st = torch.cuda.current_stream().cuda_stream
def call():
    rc = lib.mca_hip_mask_frames_dev(m.h, C.c_void_p(x.data_ptr()), 2 * (F + 1) * 512, (F + 1) * 512, A, F, C.c_void_p(out.data_ptr()), None, st)
    assert rc == 0, lib.mca_hip_mask_last_error(m.h)
for _ in range(3): call()
torch.cuda.synchronize()
t0 = time.perf_counter(); n = 10
for _ in range(n): call()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
print("k_mask_stream: %d streams x %d frames: %.3f ms per call, %.2f M frames/s, %.0f GB/s algorithmic (8192 B/frame)" % (A, F, dt * 1e3, A * F / dt / 1e6, A * F * 8192 / dt / 1e9))
