"""PCIe-inclusive rate of the host-pointer entry points on the bench shape (8 arrays x 4096 frames, 8 microphones): fp32 PCM
and 16-bit PCM, from pageable host memory (one synchronous copy each way) and from page-locked buffers (mca_hip_host_alloc:
chunked, upload / kernels / download overlapped), with a bit-for-bit comparison of the two.  This is what a drop-in caller
of process() with host buffers sees; it is never bench.py's `value`.  Run on the GPU box: python tools/host_path_rate.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mcarray_amd import api, synth
fs, N, A, F = 48000, 1024, 8, 4096
rng = np.random.default_rng(0)
x = (rng.standard_normal((A, 8, (F + 1) * 512)) * 3000).astype(np.int16)
xf = x.astype(np.float32)
prec = getattr(api, "SRP_" + os.environ.get("MCA_SRP_PRECISION", "adaptive").upper())
ctx = api.Context(fs, synth.ULA8, N, 0.5, 1, srp_precision=prec, max_arrays=A)
pin_out = {"bin": api.PinnedBuffer((A, F, 1), np.int32), "doa": api.PinnedBuffer((A, F, 1), np.float32),
           "prob": api.PinnedBuffer((A, F, 1), np.float32), "out": api.PinnedBuffer((A, 1, F * 512), np.float32)}
into = {k: v.array for k, v in pin_out.items()}
for name, buf in (("fp32 PCM", xf), ("int16 PCM", x)):
    pin_in = api.PinnedBuffer(buf.shape, buf.dtype)
    pin_in.array[...] = buf
    res = {}
    for kind, src, kw in (("pageable", buf, {}), ("page-locked", pin_in.array, {"into": into})):
        ctx.reset()
        for _ in range(2): ctx.process_frames_host(src, **kw)
        t0 = time.perf_counter(); n = 5
        for _ in range(n): ctx.process_frames_host(src, **kw)
        dt = (time.perf_counter() - t0) / n
        ctx.reset()
        r = ctx.process_frames_host(src, **kw)
        res[kind] = {k: np.array(r[k]) for k in ("bin", "doa", "prob", "out")}
        print("%s, %s host buffers, %d arrays x %d frames: %.1f ms per call = %.2f M frames/s (%.1f GB/s of input over PCIe)"
              % (name, kind, A, F, dt * 1e3, A * F / dt / 1e6, buf.nbytes / dt / 1e9))
    same = all(np.array_equal(res["pageable"][k], res["page-locked"][k]) for k in res["pageable"])
    print("   pageable and page-locked results bit-identical:", same)
    pin_in.close()
