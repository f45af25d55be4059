"""Writes tests/golden/pin_gcc_fixture.bin: one analysis frame of the golden stream ssl_reemc_d37 (4-microphone Reem-C array,
48 kHz, N = 1024), the per-pair delay tables of SteeringBeamforming::generateLookupTable (SteeringBeamforming.cpp:58-94) and
the per-pair correlations R_p[d] the ORACLE computes from them under both readings of
dsp::GeneralisedCrossCorrelation::calculateCorrelationsForPrecomputedTauMatrix (SteeringBeamforming.cpp:115-119): PHAT and NONE.
tools/pin_against_dspone.cpp feeds the same frame and delays through the real DSPONE class and says which one it implements.

Layout (little endian): char magic[8] = "MCAPIN1\\0"; int32 M, ccs_len, D, P, fs; double frames[M][ccs_len];
double delays[P][D]; double corr_phat[P][D]; double corr_none[P][D]."""
import os
import struct
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import np_twin as tw  # noqa: E402
from oracle import pyoracle as po  # noqa: E402


def main():
    g = np.load(os.path.join(ROOT, "tests", "golden", "ssl_reemc_d37.npz"))
    fs, N, xs = int(g["fs"]), int(g["N"]), g["xs"]
    frame = 5
    X = tw.stft_frames(g["pcm"].astype(np.float64), N)[frame]                    # [M][K] complex
    M, K = X.shape
    ccs = np.stack([tw.to_ccs(x) for x in X])                                   # [M][N + 2]
    delays = tw.delay_table(fs, tw.xyz_of(xs), float(g["step_deg"]))            # [P][D] (float delays as doubles)
    P, D = delays.shape
    corr = {}
    for w in ("phat", "none"):
        corr[w] = np.stack([tw.gcc_phat(X[i], X[j], delays[p], K, w).real for p, (i, j) in enumerate(tw.pair_list(M))])
    # the C oracle must say the same (it is what the GPU tests compare against)
    T = po.precompute_tau_matrix(delays[0], K)
    for w, code in (("phat", 0), ("none", 1)):
        c0 = po.gcc_tau_matrix(ccs[0], ccs[1], T, K, D, code)
        assert np.abs(c0[:, 0] - corr[w][0]).max() <= 1e-9 * np.abs(corr[w][0]).max(), w
    out = os.path.join(ROOT, "tests", "golden", "pin_gcc_fixture.bin")
    with open(out, "wb") as f:
        f.write(b"MCAPIN1\0")
        f.write(struct.pack("<5i", M, N + 2, D, P, fs))
        for a in (ccs, delays, corr["phat"], corr["none"]):
            f.write(np.ascontiguousarray(a, dtype="<f8").tobytes())
    print("wrote", out, os.path.getsize(out), "bytes; max |R| phat %.3f none %.3e" % (np.abs(corr["phat"]).max(), np.abs(corr["none"]).max()))


if __name__ == "__main__":
    main()
