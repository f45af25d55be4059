"""Per-call latency of the host-pointer stream entry point for small chunks of one array (the shape a real-time
caller of SourceSeparationAndLocalisation::process() produces).  Run on the GPU box: python tools/stream_latency.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import time, numpy as np
from mcarray_amd import api, synth
fs, N = 48000, 1024
for F in (1, 8, 32, 256):
    ctx = api.Context(fs, synth.ULA8, N, 0.5, 1, srp_precision=api.SRP_FP16X3)
    pcm = synth.noise_source_stream(synth.ULA8, 0.3, fs, (F + 1) * 512, 1)[None].astype(np.float32)
    for _ in range(5): ctx.process_frames_host(pcm)
    t0 = time.perf_counter(); n = 100
    for _ in range(n): ctx.process_frames_host(pcm)
    dt = (time.perf_counter() - t0) / n
    print("host-pointer call, 1 array x %d frames: %.3f ms per call = %.1f us per frame (real time per frame: %.0f us)" % (F, dt * 1e3, dt * 1e6 / F, 512 / fs * 1e6))
    ctx.close()
