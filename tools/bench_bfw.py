"""A/B timing of the delay-and-sum stage alone on device buffers (bench shape: 8 arrays x 4096 frames, 8-mic ULA).

usage: python tools/bench_bfw.py [arrays] [frames] -- variants are MCA_HIP_BFW_VAR bit masks (kernels_wave.hip), 'ola' is
k_beamform_ola; every variant is checked against k_beamform_ola's audio.  Timings by HIP events over `reps` launches.
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mcarray_amd import api, synth  # noqa: E402


def main():
    A = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    F = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
    variants = sys.argv[3].split(",") if len(sys.argv) > 3 else ["ola"] + [str(v) for v in range(16)]
    fts = [int(x) for x in sys.argv[4].split(",")] if len(sys.argv) > 4 else [16]
    reps = 20
    fs, N = 48000, 1024
    dev = torch.device("cuda:0")
    ctx = api.Context(fs, synth.ULA8, N, 0.5, 1, srp_precision=api.SRP_FP16, device=0, max_arrays=A)
    g = torch.Generator(device=dev); g.manual_seed(1)
    pcm = (torch.rand((A, 8, (F + 1) * 512), device=dev, generator=g) - 0.5) * 0.2
    D = ctx.D
    rng = np.random.default_rng(3)
    bins_np = np.repeat(rng.integers(1, D - 1, size=(A, 1, 1)), F, axis=1).astype(np.int32)
    bins_np[:, F // 2:, :] = (bins_np[:, F // 2:, :] + 7) % (D - 1) + 1          # one change of angle per array
    grid = ctx.doa_grid()
    doa_bin = torch.from_numpy(bins_np).to(dev)
    doa_rad = torch.from_numpy(grid[bins_np].astype(np.float32)).to(dev)
    out = torch.zeros((A, 1, F * 512), device=dev)
    ref = None
    for ft in fts:
        os.environ["MCA_HIP_BFW_FT"] = str(ft)
        for v in variants:
            os.environ.pop("MCA_HIP_BFW_ABL", None)
            if v == "ola":
                os.environ["MCA_HIP_BF_OLA"] = "1"
            elif v.startswith("abl"):
                os.environ.pop("MCA_HIP_BF_OLA", None)
                os.environ["MCA_HIP_BFW_ABL"] = v[3:]
            else:
                os.environ.pop("MCA_HIP_BF_OLA", None)
                os.environ["MCA_HIP_BFW_VAR"] = v
            run = lambda: ctx.process_frames_dev(pcm, F, doa_bin, doa_rad, None, None, out, localise=False, separate=True, bins_are_grid=True)
            ctx.reset()
            out.zero_()
            run()
            torch.cuda.synchronize()
            o = out.clone()
            if ref is None:
                ref = o
            err = float((o - ref).abs().max() / ref.abs().max())
            for _ in range(5):
                run()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                run()
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / reps
            print("ft %3d  var %4s  %.4f ms  %.1f M frames/s  max|d|/max = %.2e" % (ft, v, ms, A * F / ms / 1e3, err), flush=True)


if __name__ == "__main__":
    main()
