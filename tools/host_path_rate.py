"""PCIe-inclusive rate of the host-pointer entry points on the bench shape (8 arrays x 4096 frames, 8 microphones): fp32 PCM
and 16-bit PCM from pageable host memory.  This is what a drop-in caller of process() with host buffers sees; it is never
bench.py's `value`.  Run on the GPU box: python tools/host_path_rate.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mcarray_amd import api, synth
fs, N, A, F = 48000, 1024, 8, 4096
rng = np.random.default_rng(0)
x = (rng.standard_normal((A, 8, (F + 1) * 512)) * 3000).astype(np.int16)
xf = x.astype(np.float32)
ctx = api.Context(fs, synth.ULA8, N, 0.5, 1, srp_precision=api.SRP_FP16X3, max_arrays=A)
for name, buf in (("fp32 PCM", xf), ("int16 PCM", x)):
    for _ in range(2): ctx.process_frames_host(buf)
    t0 = time.perf_counter(); n = 5
    for _ in range(n): ctx.process_frames_host(buf)
    dt = (time.perf_counter() - t0) / n
    print("%s, host pointers, %d arrays x %d frames: %.1f ms per call = %.2f M frames/s (%.1f GB/s of input over PCIe)"
          % (name, A, F, dt * 1e3, A * F / dt / 1e6, buf.nbytes / dt / 1e9))
