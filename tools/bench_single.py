"""Per-call time and per-kernel times (the library's hipEvent brackets) of small stream calls in the adaptive precision: the
literal BASELINE configs[2] call (1 array x 4096 frames), 2 x 4096 and 1 x 8192.  usage (GPU box): python tools/bench_single.py [adaptive|fp16x3|fp16|fp32]"""
import sys, os, torch
sys.path.insert(0, os.getcwd())
from mcarray_amd import api, synth
dev = torch.device("cuda:0")
prec = {"adaptive": api.SRP_ADAPTIVE, "fp16x3": api.SRP_FP16X3, "fp16": api.SRP_FP16, "fp32": api.SRP_FP32}[sys.argv[1] if len(sys.argv) > 1 else "adaptive"]
for A, F in ((1, 4096), (2, 4096), (1, 8192), (4, 4096)):
    ctx = api.Context(48000, synth.ULA8, 1024, 0.5, 1, srp_precision=prec, device=0, max_arrays=A)
    g = torch.Generator(device=dev); g.manual_seed(1)
    import numpy as np
    pcm = torch.from_numpy(np.stack([synth.noise_source_stream(synth.ULA8, np.deg2rad(20.0 + 10 * a), 48000, (F + 1) * 512, 5 + a) for a in range(A)]).astype(np.float32)).to(dev)
    b = torch.empty(A, F, 1, dtype=torch.int32, device=dev); d = torch.empty(A, F, 1, dtype=torch.float32, device=dev)
    pr = torch.empty(A, F, 1, dtype=torch.float32, device=dev); o = torch.empty(A, 1, F * 512, dtype=torch.float32, device=dev)
    run = lambda: ctx.process_frames_dev(pcm, F, b, d, pr, None, o)
    for _ in range(8): run()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): run()
    e1.record(); torch.cuda.synchronize()
    tot = e0.elapsed_time(e1) / 50
    ctx.set_timing(True); ctx.reset_timing()
    for _ in range(20): run()
    torch.cuda.synchronize()
    kt = {name: ctx.get_timing(kid) for kid, name in api.KERNEL_NAMES.items()}
    print(A, F, "ms/call %.4f" % tot, {k: round(ms / max(n, 1), 4) for k, (n, ms) in kt.items() if n}, ctx.repair_stats(), flush=True)
    ctx.close()
