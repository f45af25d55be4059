"""Per-call latency of small chunks of one array (the shape a real-time caller of
SourceSeparationAndLocalisation::process() produces; BASELINE configs[1]): host-pointer entry point, eager device-pointer
call, and the same call replayed as one HIP graph (mca_hip_graph_*).  Run on the GPU box: python tools/stream_latency.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import time, numpy as np
import torch
from mcarray_amd import api, synth
fs, N, hop = 48000, 1024, 512
dev = torch.device("cuda:0")
for F in (1, 8, 32, 256):
    ctx = api.Context(fs, synth.ULA8, N, 0.5, 1, srp_precision=api.SRP_FP16X3)
    pcm = synth.noise_source_stream(synth.ULA8, 0.3, fs, (F + 1) * hop, 1)[None].astype(np.float32)
    for _ in range(5): ctx.process_frames_host(pcm)
    t0 = time.perf_counter(); n = 100
    for _ in range(n): ctx.process_frames_host(pcm)
    t_host = (time.perf_counter() - t0) / n
    d = torch.from_numpy(pcm).to(dev)
    b = torch.zeros((1, F, 1), dtype=torch.int32, device=dev); r = torch.zeros((1, F, 1), device=dev); p = torch.zeros((1, F, 1), device=dev)
    o = torch.zeros((1, 1, F * hop), device=dev)
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(5): ctx.process_frames_dev(d, F, b, r, p, None, o, stream=st)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        ctx.process_frames_dev(d, F, b, r, p, None, o, stream=st); torch.cuda.synchronize()
    t_dev = (time.perf_counter() - t0) / n
    g = ctx.graph_create(d, F, b, r, p, None, o)
    for _ in range(6): g.launch(st)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        g.launch(st); torch.cuda.synchronize()
    t_graph = (time.perf_counter() - t0) / n
    print("1 array x %3d frames per call: host pointers %.3f ms, device pointers %.3f ms, HIP graph %.3f ms (%.1f us per frame; real time per frame: %.0f us)"
          % (F, t_host * 1e3, t_dev * 1e3, t_graph * 1e3, t_graph * 1e6 / F, hop / fs * 1e6))
    g.close(); ctx.close()
