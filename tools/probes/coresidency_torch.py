"""Third-party victims: do rocFFT / rocBLAS / elementwise torch kernels return different bits when the standalone neighbour
(coresidency_standalone.hip) runs on another stream?  No code of this library is involved.  usage (GPU box): python tools/probes/coresidency_torch.py"""
import ctypes as C
import os

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
nb = C.CDLL(os.path.join(ROOT, "abtest", "libneighbour.so"))
nb.neighbour_launch.argtypes = [C.c_int, C.c_int, C.c_longlong, C.c_void_p, C.c_void_p]
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(5)
x = torch.randn(32768, 1024, device=dev, generator=g)
m1 = torch.randn(4096, 4096, device=dev, generator=g); m2 = torch.randn(4096, 4096, device=dev, generator=g)
h1 = m1.half(); h2 = m2.half()
sink = torch.zeros(2 * 1024 * 1024, dtype=torch.float32, device=dev)
side = torch.cuda.Stream(device=dev)
victims = (("rocFFT rfft 32768 x 1024", lambda: torch.view_as_real(torch.fft.rfft(x))), ("fp32 matmul 4096^3", lambda: m1 @ m2), ("fp16 matmul 4096^3", lambda: h1 @ h2),
           ("elementwise sin*x+cumsum", lambda: torch.cumsum(torch.sin(x) * x, dim=1)), ("softmax rows", lambda: torch.softmax(x, dim=1)), ("sort rows", lambda: torch.sort(x, dim=1).values))
for name, fn in victims:
    fn(); torch.cuda.synchronize()
    ref = fn().clone(); torch.cuda.synchronize()
    again = fn(); torch.cuda.synchronize()
    print("%-28s alone twice: %s" % (name, "identical" if torch.equal(ref, again) else "DIFFERENT (%d values)" % int((ref != again).sum())))
    for mode, hname in ((0, "LDS reads feed the MFMAs"), (1, "MFMAs on registers + LDS reads summed")):
        for rep in range(2):
            torch.cuda.synchronize()
            nb.neighbour_launch(mode, 256, 30000, C.c_void_p(sink.data_ptr()), C.c_void_p(side.cuda_stream))
            got = fn()
            running = not side.query()
            torch.cuda.synchronize()
            nbad = int((got != ref).sum())
            print("    neighbour %-40s run %d (neighbour %s at enqueue end): values that differ %d of %d%s" % (
                hname, rep, "still running" if running else "done", nbad, ref.numel(), "" if nbad == 0 else ", max |d| %.3e" % float((got.float() - ref.float()).abs().max())))
