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
nb = C.CDLL(os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "abtest", "libneighbour.so"))   # coresidency_standalone.hip -DAS_LIB
nb.neighbour_launch.argtypes = [C.c_int, C.c_int, C.c_longlong, C.c_void_p, C.c_void_p]
dev = torch.device("cuda", 0)
fs, N, F, A = 48000, 1024, 512, 8
xs = synth.ULA8
pcm = torch.from_numpy(np.stack([synth.noise_source_stream(xs, np.deg2rad(-60.0 + 17 * a), fs, (F + 1) * 512, 11 + a) for a in range(A)])).to(dev)
sink = torch.zeros(2 * 1024 * 1024, dtype=torch.float32, device=dev)     # (kind 3 reads its dummy operands from the first 64 KiB, all kinds write behind 2 MiB)
side = torch.cuda.Stream(device=dev)
main = torch.cuda.current_stream().cuda_stream


def run(kind, iters, n_wg=256):
    ctx = api.Context(fs, xs, N, 0.5, 1, srp_precision=api.SRP_FP16X3, max_arrays=A)
    b = torch.empty(A, F, 1, dtype=torch.int32, device=dev); r = torch.empty(A, F, 1, dtype=torch.float32, device=dev)
    q = torch.empty(A, F, 1, dtype=torch.float32, device=dev); o = torch.zeros(A, 1, F * 512, dtype=torch.float32, device=dev)
    en = torch.zeros(A, F, ctx.D, dtype=torch.float32, device=dev); o2 = torch.zeros(A, 1, F * 512, dtype=torch.float32, device=dev)
    off = torch.full((A, F, 1), 0.1234, dtype=torch.float32, device=dev)     # an angle off the grid: k_beamform_ola
    ctx.process_frames_dev(pcm, F, b, r, q, None, o, stream=main)          # builds the tables, warms up
    torch.cuda.synchronize()
    ctx.reset()
    torch.cuda.synchronize()
    if kind >= 100:
        nb.neighbour_launch(kind - 100, n_wg, iters, C.c_void_p(sink.data_ptr()), C.c_void_p(side.cuda_stream))
    elif kind >= 0:
        hog.hog_launch(kind, n_wg, iters, C.c_void_p(sink.data_ptr()), C.c_void_p(side.cuda_stream))
    ctx.process_frames_dev(pcm, F, b, r, q, en, o, stream=main)
    ctx.process_frames_dev(pcm, F, None, off, None, None, o2, stream=main, localise=False, separate=True)
    torch.cuda.synchronize()
    out = (b.cpu().numpy().copy(), o.cpu().numpy().copy(), en.cpu().numpy().copy(), o2.cpu().numpy().copy())
    ctx.close()
    return out


ref = run(-1, 0)
again = run(-1, 0)
print("no neighbour, twice: bins equal %s, audio equal %s, energies equal %s, off-grid audio equal %s" % tuple(np.array_equal(ref[i], again[i]) for i in range(4)))
n_wg = int(sys.argv[1]) if len(sys.argv) > 1 else 256
for name, kind, iters in (("matrix cores, accumulators in AGPRs, no LDS", 0, 400000), ("50 KiB of static LDS, reads + writes + barriers", 1, 300000), ("plain vector work", 2, 3000000),
                          ("a contraction's whole loop: loads, LDS staging, barriers, MFMA", 3, 60000)) + tuple(
                              ("the loop with " + "+".join(n for b, n in ((1, "loads"), (2, "LDS writes"), (4, "MFMA"), (8, "barriers")) if f & b), 16 + f, 60000)
                              for f in (14, 13, 11, 7, 12, 10, 9, 6, 5, 3)) + (
                              ("standalone neighbour: LDS reads feed the MFMAs", 100, 20000), ("standalone neighbour: MFMAs on registers, LDS reads summed", 101, 20000),
                              ("standalone neighbour: LDS reads only", 102, 40000)):
    if len(sys.argv) > 2 and kind < 100:
        continue
    for rep in range(2):
        got = run(kind, iters, n_wg)
        bad = np.nonzero(np.abs(got[1] - ref[1]).reshape(A, F, 512).max(axis=2) > 0)
        bad2 = np.nonzero(np.abs(got[3] - ref[3]).reshape(A, F, 512).max(axis=2) > 0)
        bad_e = np.nonzero(np.abs(got[2] - ref[2]).max(axis=2) > 0)
        print("%-62s run %d: bins equal %s; differing: hops of k_beamform_wave %d (max %.2e), hops of k_beamform_ola %d (max %.2e), energy rows %d (max %.2e)" % (
            name, rep, np.array_equal(got[0], ref[0]), len(bad[0]), np.abs(got[1] - ref[1]).max(), len(bad2[0]), np.abs(got[3] - ref[3]).max(), len(bad_e[0]), np.abs(got[2] - ref[2]).max()))
