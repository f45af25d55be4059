"""CPU: the C oracle (oracle/mca_oracle.c) against the committed golden vectors (made by the
independent numpy twin, tests/golden/make_golden.py) and against the twin on fresh random input."""
import glob
import os

import numpy as np
import pytest

from mcarray_amd import synth
from oracle import np_twin as tw
from oracle import pyoracle as po


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name + ".npz"))


@pytest.mark.parametrize("name", ["ssl_reemc_d37", "ssl_ula8_d361", "ssl_reemc_d37_s2"])
def test_ssl_stream_matches_golden(golden_dir, name):
    g = _load(golden_dir, name)
    S = int(g["n_sources"])
    r = po.ssl_stream(int(g["fs"]), int(g["N"]), g["xs"], g["pcm"].astype(np.float64), S, float(g["step_deg"]), want_map=True)
    assert np.array_equal(r["bin"], g["bin"])                       # bit-exact DOA bins
    np.testing.assert_allclose(r["doa"], g["doa"], rtol=0, atol=0)  # same float angle
    np.testing.assert_allclose(r["energy"], g["energy"], rtol=0, atol=1e-9)
    np.testing.assert_allclose(r["prob"], g["prob"], rtol=0, atol=1e-12)
    np.testing.assert_allclose(r["out"], g["out"], rtol=0, atol=2e-7)  # golden audio stored as float32


@pytest.mark.parametrize("name", ["mask_relative_both", "mask_full_both", "mask_factor_temporal", "mask_noisy_spatial"])
def test_masking_matches_golden(golden_dir, name):
    g = _load(golden_dir, name)
    m = po.Masking(int(g["fs"]), int(g["N"]), float(g["d"]), float(g["flo"]), float(g["fhi"]), int(g["method"]), int(g["alg"]))
    np.testing.assert_allclose(m.thresholds, g["thresholds"], rtol=0, atol=1e-14)
    np.testing.assert_allclose(m.center_freqs, g["center"], rtol=0, atol=1e-15)
    N = int(g["N"])
    X = po.stft_frames(np.stack([g["left"], g["right"]]).astype(np.float64), N)
    for t in range(X.shape[0]):
        _, _, dec = m.process(X[t, 0], X[t, 1])
        assert np.array_equal(dec, g["decisions"][t]), t
        np.testing.assert_allclose(m.short_time_power, g["Q"][t], rtol=1e-12)
    m2 = po.Masking(int(g["fs"]), N, float(g["d"]), float(g["flo"]), float(g["fhi"]), int(g["method"]), int(g["alg"]))
    ol, orr = m2.stream(g["left"].astype(np.float64), g["right"].astype(np.float64))
    np.testing.assert_allclose(np.stack([ol, orr]), g["out"], rtol=0, atol=2e-7)


def test_freqgcc_matches_golden(golden_dir):
    g = _load(golden_dir, "freqgcc_16k_d61")
    N = int(g["N"])
    fg = po.FreqGCC(int(g["fs"]), g["xs"], N + 2, False, float(g["step_deg"]))
    assert fg.D == 61
    X = po.stft_frames(g["pcm"].astype(np.float64), N)
    for t in range(X.shape[0]):
        voiced, corr, idx, doa, power = fg.process(X[t, 0], X[t, 1])
        assert voiced
        assert idx == g["argmax"][t]
        np.testing.assert_allclose(corr, g["corr"][t], rtol=0, atol=1e-10)
    # setProbability (BinauralLocalisation.cpp:569-631): grid points reproduce (corr-min)/sum, sub-threshold -> 0
    grid = np.array([po.doaidx2angle(i, 3.0) for i in range(61)], dtype=np.float64)
    pr = fg.set_probability(grid)
    c = g["corr"][-1]
    ref = (c - c.min()) / (c.sum() - 61 * c.min())
    ref[ref < 0.01] = 0
    np.testing.assert_allclose(pr[1:-1], ref[1:-1], atol=1e-9)
    mid = fg.set_probability(np.array([0.5 * (grid[40] + grid[41])]))
    assert min(ref[40], ref[41]) - 1e-9 <= mid[0] <= max(ref[40], ref[41]) + 1e-9


def test_freqgcc_gate_and_silence_rule_match_golden(golden_dir):
    """usePowerFloor = true: floor estimation, the gate, and the silence rule of BinauralLocalisation.cpp:530-560 -- the
    memory factors drop to zero after windowsToDecay (93) gated-out frames, so the burst that follows 94 of them restarts
    the recursions and the one that follows 93 does not.  Also: the frame that completes the floor estimation already sets
    the factors to their maxima, so the stream's first fired frame is smoothed against the zero state."""
    g = _load(golden_dir, "freqgcc_8k_gated_silence")
    fs, N = int(g["fs"]), int(g["N"])
    pcm = g["pcm_i16"].astype(np.float64) / 32768
    fg = po.FreqGCC(fs, g["xs"], N + 2, True, float(g["step_deg"]))
    X = po.stft_frames(pcm, N)
    fired = np.nonzero(g["fired"])[0]
    assert list(np.nonzero(g["restart"])[0]) == [150]
    assert fired[7] - fired[6] - 1 == 94 and fired[14] - fired[13] - 1 == 93
    prev_doa, k = 0.0, 0
    for t in range(X.shape[0]):
        voiced, corr, idx, doa, power = fg.process(X[t, 0], X[t, 1])
        assert voiced == bool(g["fired"][t]), t
        np.testing.assert_allclose(power, g["power"][t], rtol=1e-9)
        if voiced:
            assert idx == g["argmax"][t]
            np.testing.assert_allclose(corr, g["corr_fired"][k], rtol=0, atol=3e-7 * np.abs(corr).max())   # float32 fixture
            np.testing.assert_allclose(doa, g["doa"][t], rtol=0, atol=1e-12)
            np.testing.assert_allclose(fg.set_probability(np.array([prev_doa]))[0], g["prob"][t], rtol=0, atol=1e-9)
            prev_doa = doa
            k += 1
    # the restart: corr of frame 150 is the raw GCC-PHAT (no trace of the first burst), its DOA the raw grid angle
    t = 150
    assert g["doa"][t] == float(po.doaidx2angle(int(g["argmax"][t]), 3.0))
    # no restart after 93 gated-out frames: the DOA of frame 250 is pulled 0.6 : 0.4 towards the new source
    t = 250
    np.testing.assert_allclose(g["doa"][t], 0.6 * g["doa"][t - 1] + 0.4 * float(po.doaidx2angle(int(g["argmax"][t]), 3.0)), atol=1e-6)
    # the first fired frame: factors already at their maxima, zero state
    t = int(fired[0])
    np.testing.assert_allclose(g["doa"][t], 0.4 * float(po.doaidx2angle(int(g["argmax"][t]), 3.0)), atol=1e-6)


def test_multiband_matches_golden(golden_dir):
    g = _load(golden_dir, "multiband_48k_b15")
    N, nb = int(g["N"]), int(g["nbins"])
    m = po.Multiband(int(g["fs"]), g["xs"], N + 2, nb, False)
    X = po.stft_frames(g["pcm"].astype(np.float64), N)
    for t in range(X.shape[0]):
        r = m.process(X[t, 0], X[t, 1])
        assert r["fired"]
        assert np.array_equal(r["band_idx"], g["band_idx"][t])
        np.testing.assert_allclose(r["band_corr"], g["band_corr"][t], rtol=0, atol=1e-10)
        np.testing.assert_allclose(r["energy_in_doa"], g["energy_in_doa"][t], rtol=1e-12, atol=0)
        assert r["doa"] == g["doa"][t]
        np.testing.assert_allclose(r["prob"], g["prob"][t], rtol=1e-12)
        np.testing.assert_allclose(r["power"], g["power"][t], rtol=1e-12)


def test_multiband_oracle_vs_twin_gated():
    # floor estimation (3 s), the linear-vs-dB comparison and the hold-over on gated-out frames
    fs, N, F = 16000, 512, 130
    xs = [0.0, 0.086]
    rng = np.random.default_rng(8)
    L = (F + 1) * N // 2
    src = synth.noise_source_stream(xs, np.deg2rad(50.0), fs, L, 9).astype(np.float64)
    env = np.zeros(L)
    env[100 * 256:115 * 256] = 1.0
    pcm = 2.0 * rng.standard_normal((2, L)) + 100.0 / src.std() * src * env
    b = tw.multiband_stream(fs, N, xs, pcm, 15, True)
    m = po.Multiband(fs, xs, N + 2, 15, True)
    X = po.stft_frames(pcm, N)
    for t in range(F):
        r = m.process(X[t, 0], X[t, 1])
        assert r["fired"] == b["fired"][t], t
        assert r["doa"] == b["doa"][t] and np.isclose(r["prob"], b["prob"][t], rtol=1e-12)
        np.testing.assert_allclose(r["power"], b["power"][t], rtol=1e-12)
    assert 0 < b["fired"].sum() < 20 and not b["fired"][:94].any()


def test_all_golden_files_are_covered(golden_dir):
    names = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(golden_dir, "*.npz")))
    assert names == sorted(["ssl_reemc_d37", "ssl_ula8_d361", "ssl_reemc_d37_s2", "mask_relative_both", "mask_full_both",
                            "mask_factor_temporal", "mask_noisy_spatial", "freqgcc_16k_d61", "freqgcc_8k_gated_silence", "multiband_48k_b15",
                            "mvdr_ula16_48k", "mvdr_reemc_16k"])   # the mvdr_* files are covered by tests/test_oracle_mvdr.py


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_oracle_vs_twin_random(seed):
    rng = np.random.default_rng(seed)
    fs, N = 48000, 1024
    xs = np.sort(rng.uniform(0, 0.3, size=5))
    theta = rng.uniform(-80, 80)
    pcm = synth.noise_source_stream(xs, np.deg2rad(theta), fs, 6 * 512, seed + 50).astype(np.float64)
    a = po.ssl_stream(fs, N, xs, pcm, 2, 5.0, want_map=True)
    b = tw.ssl_stream(fs, N, xs, pcm, 2, 5.0)
    assert np.array_equal(a["bin"], b["bin"])
    np.testing.assert_allclose(a["energy"], b["energy"], atol=1e-9)
    np.testing.assert_allclose(a["out"], b["out"], atol=1e-12)


def test_fft_roundtrip_and_parseval():
    rng = np.random.default_rng(0)
    x = rng.standard_normal(1024)
    ccs = po.rfft_ccs(x)
    ref = np.fft.rfft(x)
    np.testing.assert_allclose(ccs[0::2], ref.real, atol=1e-10)
    np.testing.assert_allclose(ccs[1::2], ref.imag, atol=1e-10)
    np.testing.assert_allclose(po.irfft_ccs(ccs), x, atol=1e-12)
    import ctypes as C
    rows = (po.c_dp * 1)(ccs.ctypes.data_as(po.c_dp))
    p = po.lib().mca_or_fft_power(rows, 1, 1026)
    assert p == pytest.approx(np.mean(x ** 2), rel=1e-12)


def test_select_doa_edge_cases():
    # no interior maximum -> idx 0 => DOA bin 1, prob 0
    D, P = 37, 6
    doa, prob, b = po.select_doa(np.linspace(0, 1, D), P, 5.0, 1)
    assert b[0] == 1 and prob[0] == 0
    doa, prob, b = po.select_doa(np.zeros(D), P, 5.0, 2)
    assert list(b) == [1, 1] and list(prob) == [0, 0]
    # two separated peaks: the higher wins first, then the second
    E = np.zeros(D)
    E[10], E[25] = 5.0, 9.0
    E[9] = E[11] = 2.0
    E[24] = E[26] = 3.0
    doa, prob, b = po.select_doa(E, P, 5.0, 2)
    assert list(b) == [25, 10]
    t_doa, t_prob, t_b = tw.select_doa(E.copy(), P, tw.doa_step(5.0), 2)
    assert list(t_b) == [25, 10]
    np.testing.assert_allclose(prob, t_prob, atol=1e-15)


def test_silent_input_is_finite():
    fs, N = 48000, 1024
    pcm = np.zeros((4, 5 * 512))
    r = po.ssl_stream(fs, N, synth.REEM_C, pcm, 1, 5.0, want_map=True)
    assert np.all(np.isfinite(r["energy"])) and np.all(r["energy"] == 0)
    assert np.all(r["bin"] == 1) and np.all(r["out"] == 0)


def test_das_stream_is_the_separation_half_of_the_ssl_stream():
    """mca_or_das_stream (mcabeamf.cpp:77-122 around Beamformer.cpp:51-71) at the angles the localiser picked reproduces the
    audio of the whole SourceSeparationAndLocalisation stream exactly, and feeding the stream in two calls (overlap-add carry
    handed over) changes nothing."""
    from mcarray_amd import synth
    fs, N, F = 48000, 1024, 24
    xs = synth.REEM_C
    pcm = synth.noise_source_stream(xs, np.deg2rad(20.0), fs, (F + 1) * N // 2, 3).astype(np.float64)
    o = po.ssl_stream(fs, N, xs, pcm, 1, 5.0)
    d = po.das_stream(fs, N, xs, pcm, o["doa"][:, 0])
    assert np.array_equal(d, o["out"][0])
    tail = np.zeros(N // 2)
    a = po.das_stream(fs, N, xs, pcm[:, :(10 + 1) * 512], o["doa"][:10, 0], tail)
    b = po.das_stream(fs, N, xs, pcm[:, 10 * 512:], o["doa"][10:, 0], tail)
    assert np.array_equal(np.concatenate([a, b]), d)
