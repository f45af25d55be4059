import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
os.environ["MCA_HIP_ADAPT_FALLBACK"] = "0"; os.environ["MCA_HIP_ADAPT_MIN_ROWS"] = "256"
from mcarray_amd import api, synth
fs, N, F = 48000, 1024, 330
xs = synth.ULA8
thetas = (23.0, -61.5, 79.0)
A = len(thetas)
pcm = np.stack([synth.noise_source_stream(xs, np.deg2rad(th), fs, (F + 1) * N // 2, 900 + i, snr_db=20.0 - 8 * i) for i, th in enumerate(thetas)])
def run(ctx):
    ctx.reset()
    ra = ctx.process_frames_host(pcm[:, :, :151 * 512]); rb = ctx.process_frames_host(pcm[:, :, 150 * 512:])
    return np.concatenate([ra["out"], rb["out"]], axis=2), np.concatenate([ra["bin"], rb["bin"]], axis=1)
os.environ["MCA_HIP_NO_OVERLAP"] = "1"
ref_ctx = api.Context(fs, xs, N, 0.5, 1, srp_precision=api.SRP_ADAPTIVE, max_arrays=A)
ref, refb = run(ref_ctx)
os.environ.pop("MCA_HIP_NO_OVERLAP")
ctx = api.Context(fs, xs, N, 0.5, 1, srp_precision=api.SRP_ADAPTIVE, max_arrays=A)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 50
bad = 0
for it in range(n):
    o, b = run(ctx)
    if not (np.array_equal(o, ref) and np.array_equal(b, refb)):
        bad += 1
print("iterations", n, "with a difference to the serial result:", bad)
