"""A/B of a measurement switch, bit for bit: the same ADAPTIVE stream calls (9 shapes, 3 chunks each -- the exact state goes from call
to call --, eager and as a HIP graph) with and without the environment switch named on the command line must return the same bins,
probabilities, energy maps, audio and repair statistics.
usage (GPU box): tools/ab_build.sh measure "-DMCA_MEASURE" ; MCA_HIP_LIB=abtest/lib_measure.so python tools/ab_bits.py MCA_HIP_REPAIR_PATCH_KERNEL=1"""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASES = [("ULA8", 8, 4096, 1, 0), ("ULA8", 1, 4096, 1, 0), ("ULA8", 128, 256, 1, 0), ("ULA8", 3, 1501, 1, 0), ("ULA8", 2, 2050, 1, 0), ("ULA16", 4, 1024, 1, 0),
         ("REEM_C", 4, 1027, 1, 0), ("ULA8", 4, 1024, 1, 1), ("ULA8", 2, 4096, 2, 0)]

if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT)
    import torch
    from mcarray_amd import api, synth
    os.environ.setdefault("MCA_HIP_ADAPT_FALLBACK", "0")
    os.environ.setdefault("MCA_HIP_ADAPT_MAX_SOURCES", "4")
    dev = torch.device("cuda:0")
    out = {}
    for ci, (arr, A, F, S, gate) in enumerate(CASES):
        xs = getattr(synth, arr)
        chunks = 3
        pcm = np.stack([sum(synth.noise_source_stream(xs, np.deg2rad(-50.0 + 13 * (a % 9) + 35 * s), 48000, (chunks * F + 1) * 512, 31 + a + 77 * s, snr_db=15.0) for s in range(S)) for a in range(A)]).astype(np.float32)
        if gate:
            pcm[:, :, : F * 256] *= 1e-3
        for mode in ("eager", "graph"):
            ctx = api.Context(48000, xs, 1024, 0.5, S, srp_precision=api.SRP_ADAPTIVE, max_arrays=A, use_power_floor=bool(gate))
            buf = torch.zeros(A, len(xs), (F + 1) * 512, device=dev)
            b = torch.zeros(A, F, S, dtype=torch.int32, device=dev); r = torch.zeros(A, F, S, device=dev); q = torch.zeros(A, F, S, device=dev)
            en = torch.zeros(A, F, ctx.D, device=dev); o = torch.zeros(A, S, F * 512, device=dev)
            st = torch.cuda.current_stream().cuda_stream
            g = ctx.graph_create(buf, F, b, r, q, en, o) if mode == "graph" else None
            for k in range(chunks):
                buf.copy_(torch.from_numpy(pcm[:, :, k * F * 512:(k * F + F + 1) * 512]).to(dev))
                if g is not None:
                    g.launch(st)
                else:
                    ctx.process_frames_dev(buf, F, b, r, q, en, o, stream=st)
                torch.cuda.synchronize()
                for name, t in (("bin", b), ("prob", q), ("energy", en), ("out", o)):
                    out["%d/%s/%d/%s" % (ci, mode, k, name)] = t.cpu().numpy().copy()
            out["%d/%s/stats" % (ci, mode)] = np.array(list(ctx.repair_stats().values()), dtype=np.int64)
            if g is not None:
                g.close()
            ctx.close()
    np.savez(sys.argv[2], **out)
    sys.exit(0)

res = {}
switch = dict(kv.split("=", 1) for kv in sys.argv[1:])
assert switch, "name the switch: VAR=value"
for tag, env in (("ahead", {}), ("behind", switch)):
    path = "/tmp/ab_bits_%s.npz" % tag
    subprocess.check_call([sys.executable, os.path.abspath(__file__), "child", path], env=dict(os.environ, **env))
    res[tag] = np.load(path)
bad = 0
for k in sorted(res["ahead"].files):
    same = np.array_equal(res["ahead"][k], res["behind"][k])
    if not same:
        bad += 1
        print("DIFFERENT", k, CASES[int(k.split("/")[0])])
for ci, c in enumerate(CASES):
    print("case %d %s: repair statistics (frames, flagged, recomputed) eager %s graph %s" % (ci, c, res["ahead"]["%d/eager/stats" % ci].tolist(), res["ahead"]["%d/graph/stats" % ci].tolist()))
print("%d arrays compared, %d differ between the default and %s" % (len(res["ahead"].files), bad, " ".join(sys.argv[1:])))
sys.exit(1 if bad else 0)
