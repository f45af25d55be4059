"""The coarse (one fp16 plane) energy map of a 16-microphone ULA against the exact split, in units of the decision margin tau:
k_stft_phat_wave16 (whitened spectra packed to fp16 before the pair products) against k_stft_phat<16> (fp32 spectra; MCA_HIP_STFT_WG=1
in a -DMCA_MEASURE build).  tools/adaptive_check.py found 1.05 tau on one such configuration; the error model behind tau
(api.hip: tau_en) allows about 0.5.  usage (GPU box): MCA_HIP_LIB=abtest/lib_measure.so python tools/probes/r04_wave16_coarse_error.py"""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
    import torch
    import adaptive_check as ac
    from mcarray_amd import api
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(7)
    for M in (16, 8):
        for spacing in (0.02, 0.035, 0.05):
            for kind in ("static", "moving", "two", "noise"):
                xs = (spacing * np.arange(M)).tolist()
                A, F, S = 4, 2304, 1
                pcm = ac.synth(xs, A, F, kind, rng, dev)
                res = {}
                for name, prec in (("x3", api.SRP_FP16X3), ("fp16", api.SRP_FP16)):
                    ctx = api.Context(ac.FS, xs, ac.N, 0.5, S, srp_precision=prec, max_arrays=A)
                    res[name] = ac.run(ctx, pcm, F, S, 0)
                    P = ctx.P
                    ctx.close()
                err = (res["fp16"][2] - res["x3"][2]).abs() / (30.0 * P)
                sum_n2 = sum((M - 1 - g) ** 2 for g in range(M - 1))
                tau = 8.0 * np.sqrt(2.0) * 5.0e-4 * np.sqrt(0.5 * 513 * sum_n2) / (30.0 * P)
                flips = int((res["fp16"][0] != res["x3"][0]).sum())
                print("M=%2d spacing %.3f %-7s max err %.3f tau, rms %.4f tau, 99.99 %% quantile %.3f tau, fp16 flips %d of %d" % (
                    M, spacing, kind, float(err.max()) / tau, float((err.double() ** 2).mean().sqrt()) / tau,
                    float(torch.quantile(err.flatten()[::7].double(), 0.9999)) / tau, flips, A * F), flush=True)
    sys.exit(0)
for tag, env in (("wave kernels (shipped)", {}), ("fp32 spectra (MCA_HIP_STFT_WG=1)", {"MCA_HIP_STFT_WG": "1"})):
    print("==", tag, flush=True)
    subprocess.check_call([sys.executable, os.path.abspath(__file__), "child"], env=dict(os.environ, **env))
