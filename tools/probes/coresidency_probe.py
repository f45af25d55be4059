"""Drives tools/probes/coresidency_probe.hip: the library's own serial path (8-mic, FP16X3: no side streams anywhere) with a hog kernel of
one kind running on ANOTHER stream at the same time; the beamformed audio is compared with a run without a neighbour.
usage (GPU box): hipcc ... -o abtest/libhog.so (see the .hip); python tools/probes/coresidency_probe.py"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mcarray_amd import api, synth  # noqa: E402

hog = C.CDLL(os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "abtest", "libhog.so"))
hog.hog_launch.argtypes = [C.c_int, C.c_int, C.c_longlong, C.c_void_p, C.c_void_p]
dev = torch.device("cuda", 0)
fs, N, F, A = 48000, 1024, 512, 8
xs = synth.ULA8
pcm = torch.from_numpy(np.stack([synth.noise_source_stream(xs, np.deg2rad(-60.0 + 17 * a), fs, (F + 1) * 512, 11 + a) for a in range(A)])).to(dev)
sink = torch.zeros(1024 * 256, dtype=torch.float32, device=dev)
side = torch.cuda.Stream(device=dev)
main = torch.cuda.current_stream().cuda_stream


def run(kind, iters, n_wg=256):
    ctx = api.Context(fs, xs, N, 0.5, 1, srp_precision=api.SRP_FP16X3, max_arrays=A)
    b = torch.empty(A, F, 1, dtype=torch.int32, device=dev); r = torch.empty(A, F, 1, dtype=torch.float32, device=dev)
    q = torch.empty(A, F, 1, dtype=torch.float32, device=dev); o = torch.zeros(A, 1, F * 512, dtype=torch.float32, device=dev)
    ctx.process_frames_dev(pcm, F, b, r, q, None, o, stream=main)          # builds the tables, warms up
    torch.cuda.synchronize()
    ctx.reset()
    torch.cuda.synchronize()
    if kind >= 0:
        hog.hog_launch(kind, n_wg, iters, C.c_void_p(sink.data_ptr()), C.c_void_p(side.cuda_stream))
    ctx.process_frames_dev(pcm, F, b, r, q, None, o, stream=main)
    torch.cuda.synchronize()
    out = (b.cpu().numpy().copy(), o.cpu().numpy().copy())
    ctx.close()
    return out


ref = run(-1, 0)
again = run(-1, 0)
print("no neighbour, twice: bins equal %s, audio equal %s" % (np.array_equal(ref[0], again[0]), np.array_equal(ref[1], again[1])))
for name, kind, iters in (("matrix cores, accumulators in AGPRs, no LDS", 0, 400000), ("50 KiB of static LDS, reads + writes + barriers", 1, 300000), ("plain vector work", 2, 3000000)):
    for rep in range(3):
        got = run(kind, iters)
        bad = np.nonzero(np.abs(got[1] - ref[1]).reshape(A, F, 512).max(axis=2) > 0)
        print("%-50s run %d: bins equal %s, hops that differ from the run without a neighbour: %d" % (name, rep, np.array_equal(got[0], ref[0]), len(bad[0])))
