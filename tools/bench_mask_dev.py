"""Device-side throughput of the masking stream (FastBinauralMasking) on device buffers: the tuned kernel at 16 kHz (N = 1024)
and the any-length kernel at the frame lengths other sample rates give (N = 2^round(log2(0.050 fs))).
Run on the GPU box: python tools/bench_mask_dev.py"""
import os, sys, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from mcarray_amd import api, _lib
lib = _lib.load()
dev = torch.device("cuda", 0)
for fs, N in ((16000, 1024), (48000, 2048), (8000, 512)):
    A, F, hop = 64, 1024 * 1024 // N, N // 2
    m = api.FastBinauralMasking(fs, 0.086, 300.0, min(5000.0, 0.45 * fs), max_streams=A, fft_size=N)
    x = (torch.randn(A, 2, (F + 1) * hop, device=dev) * 0.1).contiguous()
    out = torch.empty(A, 2, F * hop, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    def call():
        rc = lib.mca_hip_mask_frames_dev(m.h, C.c_void_p(x.data_ptr()), 2 * (F + 1) * hop, (F + 1) * hop, A, F, C.c_void_p(out.data_ptr()), None, st)
        assert rc == 0, lib.mca_hip_mask_last_error(m.h)
    for _ in range(3): call()
    torch.cuda.synchronize()
    t0 = time.perf_counter(); n = 10
    for _ in range(n): call()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    bpf = 2 * hop * 4 * 2
    print("%s, fs %d, N %d: %d streams x %d frames: %.3f ms per call, %.2f M frames/s, %.0f GB/s algorithmic (%d B/frame)"
          % ("k_mask_stream" if N == 1024 else ("k_mask_stream_2048" if N == 2048 else "k_mask_stream_gen"), fs, N, A, F, dt * 1e3, A * F / dt / 1e6, A * F * bpf / dt / 1e9, bpf))
    m.close()
