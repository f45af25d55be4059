"""Device-side throughput of the 2-microphone GCC path (mca_hip_gcc2_frames_dev) on buffers resident in HBM.
usage (GPU box): python tools/bench_gcc2_dev.py"""
import ctypes as C
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mcarray_amd import _lib, api, synth  # noqa: E402

lib = _lib.load()
dev = torch.device("cuda", 0)
p = lambda t: C.c_void_p(t.data_ptr())
names = {0: "k_stft_phat", 1: "k_srp_gemm", 4: "k_gcc2_scan", 6: "k_sum_planes"}
# 16 kHz: N = 1024 (tuned kernels); 44.1 kHz: N = 4096, the configuration of the reference's own FreqGCC test (test_mcarray.cpp:283)
for fs, N, A, F in ((16000, 1024, 64, 2048), (44100, 4096, 64, 512)):
    hop = N // 2
    ctx = api.Context(fs, synth.BINAURAL, N, 3.0, 1, srp_precision=api.SRP_FP16X3, max_arrays=A)
    L = (F + 1) * hop
    x = (torch.randn(A, 2, L, device=dev) * 0.1).contiguous()
    idx = torch.empty(A, F, dtype=torch.int32, device=dev)
    doa = torch.empty(A, F, dtype=torch.float32, device=dev)
    prob = torch.empty(A, F, dtype=torch.float32, device=dev)
    st = torch.cuda.current_stream().cuda_stream

    def call():
        rc = lib.mca_hip_gcc2_frames_dev(ctx.h, p(x), 2 * L, L, A, F, p(idx), p(doa), p(prob), None, st)
        assert rc == 0, lib.mca_hip_last_error(ctx.h)

    for _ in range(3):
        call()
    torch.cuda.synchronize()
    ctx.set_timing(True); ctx.reset_timing()
    t0 = time.perf_counter(); n = 10
    for _ in range(n):
        call()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    kt = {v: round(ctx.get_timing(k)[1] / max(ctx.get_timing(k)[0], 1), 3) for k, v in names.items() if ctx.get_timing(k)[0]}
    print("2-mic GCC path, fs %d, N %d: %d arrays x %d frames, 61 delays: %.3f ms per call, %.1f M frames/s  %s" % (fs, N, A, F, dt * 1e3, A * F / dt / 1e6, kt))
    ctx.close()
