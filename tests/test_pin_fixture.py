"""tests/golden/pin_gcc_fixture.bin -- the input and the two expected outputs of tools/pin_against_dspone.cpp (the harness that
settles DSPONE's GCC weighting on a machine that has DSPONE): the committed file must be what the C oracle computes."""
import os
import struct

import numpy as np

from oracle import pyoracle as po

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_pin_fixture_matches_the_oracle(golden_dir):
    raw = open(os.path.join(golden_dir, "pin_gcc_fixture.bin"), "rb").read()
    assert raw[:8] == b"MCAPIN1\0"
    M, ccs, D, P, fs = struct.unpack("<5i", raw[8:28])
    assert (M, ccs, D, P, fs) == (4, 1026, 37, 6, 48000)
    a = np.frombuffer(raw[28:], dtype="<f8")
    assert a.size == M * ccs + 3 * P * D
    frames = a[:M * ccs].reshape(M, ccs)
    delays, phat, none = (a[M * ccs + i * P * D:M * ccs + (i + 1) * P * D].reshape(P, D) for i in range(3))
    K = ccs // 2
    # the delay tables are those of the reference's generateLookupTable chain (float arithmetic, SteeringBeamforming.cpp:69-73)
    st = po.Steering(fs, [0.0, 0.07, 0.175, 0.21], ccs, 5.0)
    assert all(np.array_equal(st.delays(p), delays[p]) for p in range(P))
    pairs = [(i, j) for i in range(M) for j in range(i + 1, M)]
    for p, (i, j) in enumerate(pairs):
        T = po.precompute_tau_matrix(delays[p], K)
        for code, want in ((0, phat), (1, none)):
            got = po.gcc_tau_matrix(frames[i], frames[j], T, K, D, code)[:, 0]
            assert np.abs(got - want[p]).max() <= 1e-10 * np.abs(want).max()
    assert os.path.exists(os.path.join(ROOT, "tools", "pin_against_dspone.cpp"))
