"""What do the wrong values look like?  The library's serial path beside the standalone neighbour (coresidency_standalone.hip, mode 1);
prints the structure of the differences from a run without a neighbour.  usage (GPU box): python tools/probes/coresidency_diag.py [precision]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from mcarray_amd import api, synth  # noqa: E402

nb = C.CDLL(os.path.join(ROOT, "abtest", "libneighbour.so"))
nb.neighbour_launch.argtypes = [C.c_int, C.c_int, C.c_longlong, C.c_void_p, C.c_void_p]
dev = torch.device("cuda", 0)
fs, N, F, A = 48000, 1024, 512, 8
xs = synth.ULA8
pcm = torch.from_numpy(np.stack([synth.noise_source_stream(xs, np.deg2rad(-60.0 + 17 * a), fs, (F + 1) * 512, 11 + a) for a in range(A)])).to(dev)
sink = torch.zeros(2 * 1024 * 1024, dtype=torch.float32, device=dev)
side = torch.cuda.Stream(device=dev)
main = torch.cuda.current_stream().cuda_stream
prec = {"fp32": api.SRP_FP32, "fp16": api.SRP_FP16, "fp16x3": api.SRP_FP16X3}[sys.argv[1] if len(sys.argv) > 1 else "fp16x3"]


def run(mode, iters):
    ctx = api.Context(fs, xs, N, 0.5, 1, srp_precision=prec, max_arrays=A)
    b = torch.empty(A, F, 1, dtype=torch.int32, device=dev); r = torch.empty(A, F, 1, dtype=torch.float32, device=dev)
    q = torch.empty(A, F, 1, dtype=torch.float32, device=dev); o = torch.zeros(A, 1, F * 512, dtype=torch.float32, device=dev)
    en = torch.zeros(A, F, ctx.D, dtype=torch.float32, device=dev)
    ctx.process_frames_dev(pcm, F, b, r, q, None, o, stream=main)
    torch.cuda.synchronize(); ctx.reset(); torch.cuda.synchronize()
    if mode >= 0:
        nb.neighbour_launch(mode, 256, iters, C.c_void_p(sink.data_ptr()), C.c_void_p(side.cuda_stream))
    ctx.process_frames_dev(pcm, F, b, r, q, en, o, stream=main)
    torch.cuda.synchronize()
    out = (b.cpu().numpy().copy(), o.cpu().numpy().reshape(A, F, 512).copy(), en.cpu().numpy().copy())
    ctx.close()
    return out


ref = run(-1, 0)
got = run(1, 20000)
de = np.abs(got[2] - ref[2])
rows = np.argwhere(de.max(axis=2) > 0)
print("energy rows that differ: %d of %d; entries per differing row: min %d median %d max %d of %d" % (
    len(rows), A * F, *(np.percentile((de > 0).sum(axis=2)[de.max(axis=2) > 0], [0, 50, 100]).astype(int)), de.shape[2]))
print("per array:", [(a, int((de[a].max(axis=1) > 0).sum())) for a in range(A)])
print("relative size of the differences: quantiles 50/90/99/100 of |d|/max|row|:", np.percentile((de / np.abs(ref[2]).max(axis=2, keepdims=True))[de > 0], [50, 90, 99, 100]))
for a, f in rows[:3]:
    d = got[2][a, f] - ref[2][a, f]
    print("  array %d frame %d: first differing entries" % (a, f), [(int(i), float(ref[2][a, f, i]), float(got[2][a, f, i])) for i in np.nonzero(d)[0][:6]])
# is a wrong row another frame's right row?
a, f = rows[len(rows) // 2]
match = [(aa, ff) for aa in range(A) for ff in range(F) if np.array_equal(got[2][a, f], ref[2][aa, ff])]
print("the wrong row (%d, %d) equals the right row of:" % (a, f), match[:4])
first = {a: int(np.nonzero(de[a].max(axis=1) > 0)[0][0]) if (de[a].max(axis=1) > 0).any() else -1 for a in range(A)}
print("first differing frame per array (the smoothing carries an error to every later frame):", first)
do = np.abs(got[1] - ref[1])
hops = np.argwhere(do.max(axis=2) > 0)
print("audio hops that differ: %d; samples per differing hop: min %d median %d max %d of 512" % (len(hops), *(np.percentile((do > 0).sum(axis=2)[do.max(axis=2) > 0], [0, 50, 100]).astype(int))))
for a, f in hops[:3]:
    i = np.nonzero(do[a, f])[0]
    print("  array %d hop %d: samples %d..%d differ, max |d| %.3e, rms of the right hop %.3e" % (a, f, i[0], i[-1], do[a, f].max(), np.sqrt((ref[1][a, f] ** 2).mean())))
runs = np.diff(np.concatenate([[0], (do.max(axis=2) > 0).any(axis=0).astype(int), [0]]))
print("hop indices with a difference in any array, as runs (start, length):", list(zip(np.nonzero(runs == 1)[0][:12].tolist(), (np.nonzero(runs == -1)[0] - np.nonzero(runs == 1)[0])[:12].tolist())))
# which frequency bins carry the audio error?  (a hop of output is the overlap-add of two frames, so take two hops = one frame's support)
for a, f in hops[:200:40]:
    if f + 1 >= F:
        continue
    d = np.concatenate([got[1][a, f] - ref[1][a, f], got[1][a, f + 1] - ref[1][a, f + 1]])
    D = np.abs(np.fft.rfft(d))
    R = np.abs(np.fft.rfft(np.concatenate([ref[1][a, f], ref[1][a, f + 1]])))
    top = np.argsort(D)[::-1][:8]
    print("  array %d hops %d,%d: error spectrum peaks at bins %s (|error| %s, |signal| there %s); share of the error energy in bins 0..7: %.3f" % (
        a, f, f + 1, top.tolist(), np.round(D[top], 3).tolist(), np.round(R[top], 3).tolist(), (D[:8] ** 2).sum() / (D ** 2).sum()))
