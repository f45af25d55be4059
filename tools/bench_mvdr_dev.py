"""Device-side throughput of the MVDR path (BASELINE.json configs[3]: 16 microphones, 256 concurrent streams x 64 frames)
on device buffers, with the per-kernel split from the library's HIP events and a parity spot check of stream 0
against the CPU oracle.  Usage: python tools/bench_mvdr_dev.py [--streams 256] [--frames 64] [--mics 16] [--steps 10]"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from mcarray_amd import api, synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--streams", type=int, default=256)
    ap.add_argument("--frames", type=int, default=64)
    ap.add_argument("--mics", type=int, default=16)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--check", type=int, default=1)
    a = ap.parse_args()
    fs, N = 48000, 1024
    hop, K = N // 2, N // 2 + 1
    xs = [0.32 / a.mics * m for m in range(a.mics)] if a.mics != 16 else synth.ULA16
    dev = torch.device("cuda:0")
    L = (a.frames + 1) * hop
    g = torch.Generator(device=dev); g.manual_seed(1234)
    pcm = (torch.randn((a.streams, a.mics, L), device=dev, generator=g) * 0.1).contiguous()
    if a.check:
        p0 = synth.noise_source_stream(xs, np.deg2rad(20.0), fs, L, 77) + synth.noise_source_stream(xs, np.deg2rad(-50.0), fs, L, 78, snr_db=60)
        pcm[0] = torch.from_numpy(p0.astype(np.float32)).to(dev)
    doa = torch.full((a.streams, a.frames), float(np.deg2rad(20.0)), device=dev, dtype=torch.float32)
    out = torch.empty((a.streams, a.frames * hop), device=dev, dtype=torch.float32)
    bf = api.MvdrBeamformer(fs, xs, N, max_streams=a.streams)
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(a.warmup):
        bf.process_dev(pcm, a.frames, doa, out_pcm=out, stream=st)
    torch.cuda.synchronize()
    res = {}
    if a.check:
        from oracle import pyoracle as po
        bf.reset()
        bf.process_dev(pcm, a.frames, doa, out_pcm=out, stream=st)
        torch.cuda.synchronize()
        o = po.MVDR(fs, N, xs).stream(pcm[0].cpu().numpy().astype(np.float64), np.full(a.frames, float(np.float32(np.deg2rad(20.0)))))
        err = np.abs(out[0].cpu().numpy() - o["out"]).max() / np.abs(o["out"]).max()
        res["audio_err_rel_max"] = float(err)
    bf.set_timing(True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        bf.process_dev(pcm, a.frames, doa, out_pcm=out, stream=st)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.steps
    frames = a.streams * a.frames
    res.update(dict(workload="%d streams x %d frames, %d mics, N=%d" % (a.streams, a.frames, a.mics, N), ms_per_step=dt * 1e3,
                    frames_per_s=frames / dt, algorithmic_GBps=frames / dt * (a.mics * hop * 4 + hop * 4) / 1e9))
    for kid, name in ((0, "k_mvdr_analyse"), (1, "k_mvdr_solve"), (2, "k_mvdr_synth")):
        n, ms = bf.get_timing(kid)
        res[name + "_ms"] = ms / max(n, 1)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
