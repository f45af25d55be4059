"""Device-side throughput of the multiband 2-microphone localiser on device buffers: 48 kHz (N = 1024) and 16 kHz (N = 512).
Run on the GPU box: python tools/bench_mb_dev.py"""
import os, sys, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mcarray_amd import api, synth, _lib
lib = _lib.load()
dev = torch.device("cuda", 0)
for fs in (48000, 16000):
    A = 64
    loc = api.MultibandBinarualLocalisation(fs, synth.BINAURAL, 15, False, max_arrays=A)
    N, hop = loc.N, loc.hop
    F = 1024 * 1024 // N
    x = (torch.randn(A, 2, (F + 1) * hop, device=dev) * 0.1).contiguous()
    doa = torch.empty(A, F, device=dev); prob = torch.empty(A, F, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    def call():
        rc = lib.mca_hip_mb_frames_dev(loc.h, C.c_void_p(x.data_ptr()), 2 * (F + 1) * hop, (F + 1) * hop, A, F, C.c_void_p(doa.data_ptr()),
                                       C.c_void_p(prob.data_ptr()), None, None, None, None, None, st)
        assert rc == 0, lib.mca_hip_mb_last_error(loc.h)
    for _ in range(3): call()
    torch.cuda.synchronize()
    t0 = time.perf_counter(); n = 10
    for _ in range(n): call()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print("multiband localiser: %d arrays x %d frames, fs %d, N=%d, 15 bands: %.3f ms per call, %.1f M frames/s" % (A, F, fs, N, dt * 1e3, A * F / dt / 1e6))
    loc.close()
