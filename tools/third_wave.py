#!/usr/bin/env python3
"""Round 5: what is a third wave per SIMD worth to the wave-per-run analysis kernel?  The 8-microphone instantiation needs
233 ... 248 registers (two waves per SIMD); the 4-microphone one of the SAME kernel needs 164 and 46 KiB of LDS per workgroup, so three
workgroups per CU -- three waves per SIMD -- are resident.  Padding its LDS (MCA_HIP_SPW_LDS_PAD, measurement build) takes that to two and
to one: the same code at 1 / 2 / 3 waves per SIMD.  usage (GPU box, MCA_HIP_LIB = a -DMCA_MEASURE build):
    MCA_HIP_SPW_LDS_PAD=<KiB> python tools/third_wave.py            (prints the analysis kernel's average launch)"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mcarray_amd import api, synth  # noqa: E402

A, F, N, hop = 16, 4096, 1024, 512
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(3)
pcm = (torch.randn(A, 4, (F + 1) * hop + 64, device=dev, generator=g) * 0.1).contiguous()
ctx = api.Context(48000, synth.REEM_C, N, 0.5, 1, srp_precision=api.SRP_FP16, max_arrays=A)
b = torch.empty(A, F, 1, dtype=torch.int32, device=dev)
r = torch.empty(A, F, 1, dtype=torch.float32, device=dev)
q = torch.empty(A, F, 1, dtype=torch.float32, device=dev)
st = torch.cuda.current_stream().cuda_stream
L = pcm.shape[2]
call = lambda: ctx._check(ctx._lib.mca_hip_localise_frames_dev(ctx.h, pcm.data_ptr(), 4 * L, L, A, F, b.data_ptr(), r.data_ptr(), q.data_ptr(), None, st))
for _ in range(5):
    call()
torch.cuda.synchronize()
ctx.set_timing_kernels([api.K_STFT_PHAT]); ctx.reset_timing()
n = int(os.environ.get("THIRD_WAVE_STEPS", "30"))
for _ in range(n):
    call()
torch.cuda.synchronize()
nl, ms = ctx.get_timing(api.K_STFT_PHAT)
print("LDS pad %3s KiB: k_stft_phat_wave<4 mics> %d launches, %.1f us per launch of %d frames" % (os.environ.get("MCA_HIP_SPW_LDS_PAD", "0"), nl, ms / nl * 1e3, A * F))
ctx.close()
