"""Throughput of the stream API for shapes other than the official bench workload (not a bench line: a survey).
usage (GPU box): python tools/bench_shapes.py"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mcarray_amd import api, synth  # noqa: E402


def run(M, fs, N, step, A, F, prec=api.SRP_FP16X3, S=1, steps=10, gate=False, sources=False):
    xs = {2: synth.BINAURAL, 4: synth.REEM_C, 8: synth.ULA8, 16: synth.ULA16}[M]
    dev = torch.device("cuda", 0)
    hop = N // 2
    g = torch.Generator(device=dev); g.manual_seed(1)
    if sources:         # S far-field white sources per array (bench.py's generator) instead of noise alone
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        import bench
        pcm = sum(bench.synth_batch(xs, [1000 * s_ + a for a in range(A)], F, dev, hop=hop, fs=fs)[0] for s_ in range(S))[:, :, :(F + 1) * hop].contiguous()
    else:
        pcm = (torch.randn(A, M, (F + 1) * hop, device=dev, generator=g) * 0.1).contiguous()
    ctx = api.Context(fs, xs, N, step, S, srp_precision=prec, max_arrays=A, use_power_floor=gate)
    doa_bin = torch.empty(A, F, S, dtype=torch.int32, device=dev)
    doa_rad = torch.empty(A, F, S, dtype=torch.float32, device=dev)
    prob = torch.empty(A, F, S, dtype=torch.float32, device=dev)
    out = torch.empty(A, S, F * hop, dtype=torch.float32, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(3):
        ctx.process_frames_dev(pcm, F, doa_bin, doa_rad, prob, None, out, stream=st)
    torch.cuda.synchronize()
    ctx.set_timing(True); ctx.reset_timing()
    t0 = time.perf_counter()
    for _ in range(steps):
        ctx.process_frames_dev(pcm, F, doa_bin, doa_rad, prob, None, out, stream=st)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    kt = {name: round(ctx.get_timing(kid)[1] / max(ctx.get_timing(kid)[0], 1), 3) for kid, name in api.KERNEL_NAMES.items() if ctx.get_timing(kid)[0]}
    print("M=%2d fs=%6d N=%4d D=%3d S=%d%s  %3d arrays x %5d frames: %6.2f M frames/s  %.3f ms/step  %s%s" %
          (M, fs, N, ctx.D, S, " gate" if gate else "", A, F, A * F / dt / 1e6, dt * 1e3, kt,
           "  repair %s %s" % (ctx.repair_stats(), ctx.repair_columns()) if prec == api.SRP_ADAPTIVE else ""))
    ctx.close()


def run_two_channel():
    """host-pointer calls of the 2-channel modules on batches already in host memory (kernel time by HIP events is not
    exposed for these contexts; the figure includes the PCIe copies, so it is a lower bound of the device rate)"""
    rng = np.random.default_rng(0)
    for name, make, call in (
        ("FastBinauralMasking 16 kHz", lambda A: api.FastBinauralMasking(16000, 0.086, 500.0, 5000.0, max_streams=A), lambda m, x: m.process(x, want_decisions=False)),
        ("MultibandBinarualLocalisation 48 kHz, 15 bands", lambda A: api.MultibandBinarualLocalisation(48000, synth.BINAURAL, 15, False, max_arrays=A), lambda m, x: m.process(x)),
        ("FreqGCCBinauralLocalisation 16 kHz", lambda A: api.FreqGCCBinauralLocalisation(16000, synth.BINAURAL, False, 3.0, max_arrays=A), lambda m, x: m.process(x)),
    ):
        A, F = 64, 1024
        m = make(A)
        hop = m.hop if hasattr(m, "hop") else m.ctx.hop
        x = (rng.standard_normal((A, 2, (F + 1) * hop)) * 0.1).astype(np.float32)
        for _ in range(2):
            call(m, x)
        t0 = time.perf_counter(); n = 5
        for _ in range(n):
            call(m, x)
        dt = (time.perf_counter() - t0) / n
        print("%-50s %3d streams x %4d frames: %6.2f M frames/s incl. host copies (%.2f ms per call)" % (name, A, F, A * F / dt / 1e6, dt * 1e3))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "two":
        run_two_channel()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "n2048":       # 2048-sample frames (kernels_2048.hip; MCA_HIP_NO_N2048=1: the any-length kernels)
        run(8, 96000, 2048, 0.5, 8, 2048, steps=30)
        run(8, 96000, 2048, 0.5, 8, 2048, prec=api.SRP_FP16, steps=30)
        run(4, 96000, 2048, 0.5, 8, 2048, steps=30)
        # one far-field source per array: the exact split and the adaptive mode (round 6: coarse + repair at this frame length too)
        for M in (8, 4):
            run(M, 96000, 2048, 0.5, 8, 2048, sources=True, steps=30)
            run(M, 96000, 2048, 0.5, 8, 2048, prec=api.SRP_ADAPTIVE, sources=True, steps=30)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "n512":        # 512-sample frames at 16 kHz, one far-field source per array: the exact split and the adaptive mode
        for M in (8, 4):
            run(M, 16000, 512, 0.5, 8, 4096, sources=True, steps=30)
            run(M, 16000, 512, 0.5, 8, 4096, prec=api.SRP_ADAPTIVE, sources=True, steps=30)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "n2048a":      # the 8-microphone 2048-sample call alone (under rocprofv3: PMC_CMD of tools/pmc_sq.sh)
        run(8, 96000, 2048, 0.5, 8, 2048, steps=5)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "n2048b":      # ... and its adaptive call on one source per array (under rocprofv3 --kernel-trace --stats)
        run(8, 96000, 2048, 0.5, 8, 2048, prec=api.SRP_ADAPTIVE, sources=True, steps=30)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "n2048c":      # ... of 8 and of 4 microphones, adaptive and plain fp16 (tools/ab_merge_2048.sh)
        for M in (8, 4):
            run(M, 96000, 2048, 0.5, 8, 2048, prec=api.SRP_ADAPTIVE, sources=True, steps=30)
            run(M, 96000, 2048, 0.5, 8, 2048, prec=api.SRP_FP16, sources=True, steps=30)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "gate":          # the reference's default usePowerFloor, in the bench's precision
        for gate in (False, True):
            run(8, 48000, 1024, 0.5, 8, 4096, prec=api.SRP_ADAPTIVE, gate=gate, steps=30)
            run(4, 48000, 1024, 0.5, 8, 4096, prec=api.SRP_ADAPTIVE, gate=gate, steps=30)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "m16":         # 16 microphones, one far-field source per array: the exact split and the shipped default
        run(16, 48000, 1024, 0.5, 8, 2048, prec=api.SRP_FP16X3, sources=True, steps=30)
        run(16, 48000, 1024, 0.5, 8, 2048, prec=api.SRP_FP16, sources=True, steps=30)
        run(16, 48000, 1024, 0.5, 8, 2048, prec=api.SRP_ADAPTIVE, sources=True, steps=30)
        run(16, 48000, 1024, 0.5, 8, 4096, prec=api.SRP_ADAPTIVE, sources=True, steps=30)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "m16a":        # the adaptive 16-microphone call alone (tools/m16_breakdown.sh runs it under rocprofv3)
        run(16, 48000, 1024, 0.5, 8, 2048, prec=api.SRP_ADAPTIVE, sources=True, steps=30)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "s1":          # one source per array, seeds 0..7: the content with clustered near ties (bench.py's second repair_spread input)
        run(8, 48000, 1024, 0.5, 8, 4096, S=1, sources=True, prec=api.SRP_ADAPTIVE, steps=30)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "sources":
        for S in (1, 2, 3, 4):
            run(8, 48000, 1024, 0.5, 8, 4096, S=S, sources=True, prec=api.SRP_ADAPTIVE, steps=30)
        sys.exit(0)
    run(8, 48000, 1024, 0.5, 8, 4096)
    run(8, 48000, 1024, 5.0, 8, 4096)
    run(8, 48000, 1024, 0.5, 8, 4096, S=2)
    run(16, 48000, 1024, 0.5, 8, 2048)
    run(4, 48000, 1024, 0.5, 8, 4096)
    run(2, 16000, 1024, 3.0, 32, 4096)
    run(8, 16000, 512, 0.5, 8, 4096)
    run(8, 96000, 2048, 0.5, 8, 2048)
    run(8, 48000, 1024, 0.5, 128, 256)
