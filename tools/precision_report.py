#!/usr/bin/env python3
"""Parity of the three SRP precisions against the CPU oracle on a sample of the bench workload geometry
(8-mic ULA, 361 angles): max energy-map error relative to the map's peak, DOA-bin mismatches, audio error.
Run on the GPU box: python tools/precision_report.py > profiles/rNN_precision_report.json"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mcarray_amd import api, synth  # noqa: E402
from oracle import pyoracle as po  # noqa: E402

fs, N, F, A = 48000, 1024, 384, 6
rng = np.random.default_rng(2026)
thetas = rng.uniform(-80, 80, size=A)
pcm = np.stack([synth.noise_source_stream(synth.ULA8, np.deg2rad(thetas[a]), fs, (F + 1) * 512, 4000 + a) for a in range(A)])
ora = [po.ssl_stream(fs, N, synth.ULA8, pcm[a].astype(np.float64), 1, 0.5, want_map=True) for a in range(A)]
out = {"workload": "%d arrays x %d frames, 8-mic ULA 0.04 m, 48 kHz, N=1024, 361 angles, white far-field source + 20 dB sensor noise" % (A, F),
       "frames": A * F, "precisions": {}}
os.environ["MCA_HIP_ADAPT_MIN_ROWS"] = "256"      # the sample is small: let the adaptive mode run on it
for name, prec in (("fp32", api.SRP_FP32), ("fp16x3", api.SRP_FP16X3), ("fp16", api.SRP_FP16), ("adaptive", api.SRP_ADAPTIVE)):
    ctx = api.Context(fs, synth.ULA8, N, 0.5, 1, srp_precision=prec, max_arrays=A)
    ctx.reset_timing()
    r = ctx.process_frames_host(pcm, want_energy=True)
    e_err = max(float(np.abs(r["energy"][a] - ora[a]["energy"]).max() / np.abs(ora[a]["energy"]).max()) for a in range(A))
    mism = int(sum((r["bin"][a] != ora[a]["bin"]).sum() for a in range(A)))
    p_err = max(float(np.abs(r["prob"][a] - ora[a]["prob"]).max()) for a in range(A))
    a_err = max(float(np.abs(r["out"][a] - ora[a]["out"]).max() / np.abs(ora[a]["out"]).max()) for a in range(A))
    out["precisions"][name] = {"max_energy_err_rel_to_peak": e_err, "doa_bin_mismatches": mism, "max_prob_abs_err": p_err,
                               "max_audio_err_rel_to_peak": a_err}
    if prec == api.SRP_ADAPTIVE:
        out["precisions"][name]["repair"] = ctx.repair_stats()
    ctx.close()
print(json.dumps(out, indent=1))
