"""Timing of the analysis stage (k_stft_phat_wave) and the delay-and-sum stage across batch shapes and PCM row paddings, from the
library's own hipEvent brackets.  usage: python tools/bench_spw.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mcarray_amd import api, synth  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    fs, N = 48000, 1024
    for A, F in ((8, 4096), (128, 256), (32, 1024), (128, 256)):
        for pad in (0, 64, 1024, 5 * 512):
            L = (F + 1) * 512 + pad
            ctx = api.Context(fs, synth.ULA8, N, 0.5, 1, srp_precision=api.SRP_ADAPTIVE, device=0, max_arrays=A)
            g = torch.Generator(device=dev); g.manual_seed(1)
            pcm = (torch.rand((A, 8, L), device=dev, generator=g) - 0.5) * 0.2
            b = torch.empty(A, F, 1, dtype=torch.int32, device=dev)
            d = torch.empty(A, F, 1, dtype=torch.float32, device=dev)
            pr = torch.empty(A, F, 1, dtype=torch.float32, device=dev)
            o = torch.empty(A, 1, F * 512, dtype=torch.float32, device=dev)
            run = lambda: ctx.process_frames_dev(pcm, F, b, d, pr, None, o)
            for _ in range(8):
                run()
            torch.cuda.synchronize()
            ctx.set_timing(True); ctx.reset_timing()
            for _ in range(20):
                run()
            torch.cuda.synchronize()
            kt = {name: ctx.get_timing(kid) for kid, name in api.KERNEL_NAMES.items()}
            print("%4d x %4d pad %5d:" % (A, F, pad), {k: round(ms / max(n, 1), 4) for k, (n, ms) in kt.items() if n}, flush=True)
            ctx.close()
            del pcm


if __name__ == "__main__":
    main()
