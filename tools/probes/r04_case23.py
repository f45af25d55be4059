"""Replays tools/adaptive_check.py (30 cases, seed 1) up to case 23 -- 16-microphone ULA, moving source, two calls -- where the coarse map
was 1.05 tau off the exact split, and shows where the error sits (array, frame, direction), with and without the split into two calls."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
import adaptive_check as ac
from mcarray_amd import api
rng = np.random.default_rng(1)
dev = torch.device("cuda:0")
for case in range(24):
    M = int(rng.choice([3, 4, 5, 8, 8, 8, 16])); ula = bool(rng.integers(0, 2))
    xs = ((0.02 + 0.03 * rng.random()) * np.arange(M) if ula else np.sort(rng.uniform(0, 0.05 * M, M))).tolist()
    step = float(rng.choice([0.5, 0.5, 1.0, 3.0, 5.0])); S = int(rng.choice([1, 1, 2, 3, 4])); A = int(rng.choice([4, 8])); F = int(rng.choice([2048, 2304, 4096]))
    kind = str(rng.choice(["static", "static", "two", "noise", "moving"])); cut = int(rng.integers(200, F - 200)) if rng.integers(0, 2) else 0
    pcm = ac.synth(xs, A, F, kind, rng, dev)
print("case", case, dict(M=M, ula=ula, spacing=xs[1] - xs[0], step=step, S=S, A=A, F=F, kind=kind, cut=cut))
for c in (cut, 0):
    res = {}
    for name, prec in (("x3", api.SRP_FP16X3), ("fp16", api.SRP_FP16), ("fp32", api.SRP_FP32)):
        ctx = api.Context(ac.FS, xs, ac.N, step, S, srp_precision=prec, max_arrays=A)
        res[name] = ac.run(ctx, pcm, F, S, c); P = ctx.P; ctx.close()
    sum_n2 = sum((M - 1 - g) ** 2 for g in range(M - 1))
    tau = 8.0 * np.sqrt(2.0) * 5.0e-4 * np.sqrt(0.5 * 513 * sum_n2) / (30.0 * P)
    for a_name, b_name in (("fp16", "x3"), ("x3", "fp32")):
        err = (res[a_name][2] - res[b_name][2]).abs() / (30.0 * P) / tau
        w = int(err.argmax()); a_, t_, d_ = np.unravel_index(w, tuple(err.shape))
        per_frame = err.amax(dim=2)
        bad = (per_frame > 0.5).nonzero().tolist()
        print("cut %4d  %s vs %s: max %.3f tau at array %d frame %d direction %d; frames above 0.5 tau: %s" % (c, a_name, b_name, float(err.max()), a_, t_, d_, bad[:12]))
        if bad:
            a0, t0 = bad[0]
            print("     error over the frames around it (max over directions):", [round(float(per_frame[a0, t]), 3) for t in range(max(0, t0 - 3), min(F, t0 + 24))])

# does the spike follow the DATA or the POSITION?  array 1 alone, and a window of its frames starting elsewhere
def err_of(pc, Fw):
    out = {}
    for name, prec in (("x3", api.SRP_FP16X3), ("fp16", api.SRP_FP16)):
        ctx = api.Context(ac.FS, xs, ac.N, step, S, srp_precision=prec, max_arrays=pc.shape[0])
        out[name] = ac.run(ctx, pc, Fw, S, 0); ctx.close()
    e = (out["fp16"][2] - out["x3"][2]).abs().amax(dim=2) / (30.0 * P) / tau
    return e, out
e, _ = err_of(pcm[1:2].contiguous(), F)
print("array 1 alone: frames above 0.5 tau:", (e > 0.5).nonzero().tolist()[:8], "max %.3f" % float(e.max()))
for start in (800, 801, 802, 803, 810, 812):
    Fw = 256
    pc = pcm[1:2, :, start * 512:(start + Fw + 1) * 512].contiguous()
    e, out = err_of(pc, Fw)
    print("array 1, frames %d..%d as a call of their own: frames above 0.5 tau (call-relative): %s, max %.3f" % (start, start + Fw - 1, (e > 0.5).nonzero().tolist()[:8], float(e.max())))
# the frame's samples: anything unusual?
fr = pcm[1, :, 812 * 512:812 * 512 + 1024]
print("frame 812 of array 1: max |x| per channel", [round(float(v), 4) for v in fr.abs().amax(dim=1)], "clipped samples", int((fr.abs() >= 1.0).sum()), "exact zeros", int((fr == 0).sum()))

# which channel carries it?  the frames 812..1067 of array 1 as a call, one channel silenced at a time (exact zeros: both paths drop it)
base = pcm[1:2, :, 812 * 512:(812 + 257) * 512].contiguous()
for ch in range(M):
    pc = base.clone(); pc[0, ch] = 0
    e, _ = err_of(pc, 256)
    print("channel %2d silenced: error of the first frame %.3f tau" % (ch, float(e[0, 0])))

# razor's edge or robust?  tiny noise on channel 15 of the bad frame only; then a different gain on channel 15
g = torch.Generator(device=dev); g.manual_seed(3)
for eps in (1e-7, 1e-5, 1e-3):
    pc = base.clone(); pc[0, 15, :1024] += torch.randn(1024, device=dev, generator=g) * eps
    e, _ = err_of(pc, 256)
    print("noise %.0e on channel 15's first 1024 samples: error of the first frame %.3f tau" % (eps, float(e[0, 0])))
for gain in (0.5, 0.999, 2.0):
    pc = base.clone(); pc[0, 15] *= gain
    e, _ = err_of(pc, 256)
    print("channel 15 scaled by %.3f: error of the first frame %.3f tau" % (gain, float(e[0, 0])))
pc = base.clone(); pc[0, 15] = base[0, 14]; pc[0, 14] = base[0, 15]
e, _ = err_of(pc, 256)
print("channels 14 and 15 exchanged: error of the first frame %.3f tau" % float(e[0, 0]))
x = base[0, 15, :1024].double().cpu().numpy(); w = np.hanning(1025)[:1024]
X = np.fft.rfft(x * w)
print("channel 15, frame 812: |X| min %.3e at bin %d, max %.3e; bins below 1e-4 of the max: %d" % (np.abs(X).min(), int(np.abs(X).argmin()), np.abs(X).max(), int((np.abs(X) < 1e-4 * np.abs(X).max()).sum())))

# the adaptive mode on the same case: the coarse analysis marks the frame, k_scan_pick repairs it and its six successors
os.environ["MCA_HIP_ADAPT_FALLBACK"] = "0"
res = {}
for name, prec in (("x3", api.SRP_FP16X3), ("adaptive", api.SRP_ADAPTIVE)):
    ctx = api.Context(ac.FS, xs, ac.N, step, S, srp_precision=prec, max_arrays=A)
    ctx.reset_timing()
    res[name] = ac.run(ctx, pcm, F, S, 0)
    if name == "adaptive":
        print("adaptive repair statistics:", ctx.repair_stats())
    ctx.close()
err = (res["adaptive"][2] - res["x3"][2]).abs().amax(dim=2) / (30.0 * P) / tau
print("adaptive vs x3 around array 1 frame 812 (max over directions, in tau):", [round(float(err[1, t]), 3) for t in range(808, 824)])
print("bins equal:", bool((res["adaptive"][0] == res["x3"][0]).all()))
