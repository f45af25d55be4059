"""The ADAPTIVE mode on input where nearly every frame needs the repair (independent white noise per channel, no source, no
power gate), with and without its back-off to plain FP16X3 (MCA_HIP_ADAPT_FALLBACK, api.hip: adapt_policy_begin), next to
FP16X3 itself.  The host waits for every call before it enqueues the next one -- a caller that queues many calls ahead gets
the reports late and backs off later.  usage (GPU box): python tools/bench_fallback.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mcarray_amd import api, synth  # noqa: E402


def run(prec, fallback, steps=60, A=8, F=4096):
    os.environ["MCA_HIP_ADAPT_FALLBACK"] = "1" if fallback else "0"
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev); g.manual_seed(1)
    pcm = (torch.randn(A, 8, (F + 1) * 512, device=dev, generator=g) * 0.1).contiguous()
    ctx = api.Context(48000, synth.ULA8, 1024, 0.5, 1, srp_precision=prec, max_arrays=A)
    b = torch.empty(A, F, 1, dtype=torch.int32, device=dev); d = torch.empty(A, F, 1, dtype=torch.float32, device=dev)
    pr = torch.empty(A, F, 1, dtype=torch.float32, device=dev); o = torch.empty(A, 1, F * 512, dtype=torch.float32, device=dev)
    for _ in range(2):
        ctx.process_frames_dev(pcm, F, b, d, pr, None, o)
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        ctx.process_frames_dev(pcm, F, b, d, pr, None, o)
        torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    st = ctx.repair_stats() if prec == api.SRP_ADAPTIVE else None
    print("%-9s back-off %-3s: %.3f ms per call of %d x %d frames (%.1f M frames/s)%s" % (
        {api.SRP_ADAPTIVE: "adaptive", api.SRP_FP16X3: "fp16x3"}[prec], "on" if fallback else "off", dt * 1e3, A, F, A * F / dt / 1e6,
        "  adaptive frames %d of %d" % (st["frames"], (steps + 2) * A * F) if st else ""), flush=True)
    ctx.close()


def onset(kind, fallback, A=8, F=4096):
    """the first calls of a stream whose content turns bad for the adaptive mode: a source for 4 calls, then `kind` (noise only /
    digital silence) -- each call timed on its own (the host waits for it)"""
    os.environ["MCA_HIP_ADAPT_FALLBACK"] = "1" if fallback else "0"
    dev = torch.device("cuda", 0)
    import bench
    good = bench.synth_batch(synth.ULA8, list(range(100, 100 + A)), F, dev)[0][:, :, :(F + 1) * 512].contiguous()
    g = torch.Generator(device=dev); g.manual_seed(2)
    bad = (torch.randn(A, 8, (F + 1) * 512, device=dev, generator=g) * 0.1).contiguous() if kind == "noise" else torch.zeros(A, 8, (F + 1) * 512, device=dev)
    ctx = api.Context(48000, synth.ULA8, 1024, 0.5, 1, srp_precision=api.SRP_ADAPTIVE, max_arrays=A)
    b = torch.empty(A, F, 1, dtype=torch.int32, device=dev); d = torch.empty(A, F, 1, dtype=torch.float32, device=dev)
    pr = torch.empty(A, F, 1, dtype=torch.float32, device=dev); o = torch.empty(A, 1, F * 512, dtype=torch.float32, device=dev)
    ms = []
    for i in range(12):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ctx.process_frames_dev(good if i < 4 else bad, F, b, d, pr, None, o)
        torch.cuda.synchronize()
        ms.append((time.perf_counter() - t0) * 1e3)
    print("onset of %-7s back-off %-3s: ms per call %s" % (kind, "on" if fallback else "off", " ".join("%.2f" % m for m in ms)), flush=True)
    ctx.close()


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "onset":
        for kind in ("noise", "silence"):
            for fb in (True, False):
                onset(kind, fb)
        sys.exit(0)
    run(api.SRP_FP16X3, False)
    run(api.SRP_ADAPTIVE, False)
    run(api.SRP_ADAPTIVE, True)
