"""GPU parity of the multiband 2-microphone localiser (mca_hip_mb_*, kernels_multiband.hip) through the C ABI,
against the CPU oracle (MultibandBinarualLocalisation.cpp:145-258 restated in oracle/mca_oracle.c).

Tolerances (fp32 GPU vs fp64 oracle):
  * smoothed band correlations: |gpu - oracle| <= 2e-5 * max|corr| of the frame
  * per-band first-max index: exact, or a flagged numerical tie (oracle's own values at the two indices
    differ by < 1e-5 * max|corr|); frames with a flagged band are excluded from the exact DOA comparison
  * band energies / histogram: relative 1e-4; prob: 1e-4 absolute; power: relative 1e-4
"""
import numpy as np
import pytest

from mcarray_amd import api, synth
from oracle import pyoracle as po

pytestmark = pytest.mark.gpu


def _compare(loc, og, pcm, N, r, a=0, t0=0):
    X = po.stft_frames(pcm.astype(np.float64), N)
    F = X.shape[0]
    flagged = 0
    for t in range(F):
        o = og.process(X[t, 0], X[t, 1])
        scale = np.abs(o["band_corr"]).max() + 1e-300
        assert np.abs(r["band_corr"][a, t0 + t] - o["band_corr"]).max() <= 2e-5 * scale, t
        tie = False
        for b in range(loc.nbins):
            gi, oi = int(r["band_idx"][a, t0 + t, b]), int(o["band_idx"][b])
            if gi != oi:
                assert abs(o["band_corr"][b, gi] - o["band_corr"][b, oi]) < 1e-5 * scale, (t, b, gi, oi)
                tie = True
        assert bool(r["voiced"][a, t0 + t]) == o["fired"], t
        np.testing.assert_allclose(r["power"][a, t0 + t], o["power"], rtol=1e-4)
        if tie:
            flagged += 1
            continue
        np.testing.assert_allclose(r["energy_in_doa"][a, t0 + t], o["energy_in_doa"], rtol=1e-4, atol=1e-7 * o["energy_in_doa"].max())
        assert abs(r["doa"][a, t0 + t] - np.float32(o["doa"])) == 0, t
        assert abs(r["prob"][a, t0 + t] - o["prob"]) <= 1e-4, t
    return flagged


@pytest.mark.parametrize("fs,N,nbins", [(48000, 1024, 15), (48000, 1024, 25), (16000, 512, 15)])
def test_multiband_stream_matches_oracle(fs, N, nbins):
    F, A = 70, 2
    xs = synth.BINAURAL
    pcm = np.stack([synth.noise_source_stream(xs, np.deg2rad(-55.0 + 85.0 * a), fs, (F + 1) * N // 2, 40 + a) for a in range(A)])
    loc = api.MultibandBinarualLocalisation(fs, xs, nbins, False, max_arrays=A)
    assert loc.N == N and loc.D == 37
    og = po.Multiband(fs, xs, N + 2, nbins, False)
    np.testing.assert_allclose(loc.filters(), og.filters(), rtol=0, atol=0)
    r = loc.process(pcm, want_bands=True)
    flagged = 0
    for a in range(A):
        flagged += _compare(loc, po.Multiband(fs, xs, N + 2, nbins, False), pcm[a], N, r, a)
    assert flagged <= 0.1 * A * F, flagged
    # the localiser finds the sources (5 degree grid, broadband noise)
    for a in range(A):
        deg = np.rad2deg(r["doa"][a, 10:])
        assert np.median(np.abs(deg - (-55.0 + 85.0 * a))) <= 5.0
    loc.close()


def test_multiband_reference_sine_property_on_gpu():
    # test/test_mcarray.cpp:344-383 on the GPU path: 1 kHz sine, 25 bands, ungated, +-15 degrees
    fs, N, F = 48000, 1024, 30
    xs = [0.0, 0.086]
    angles = list(range(-90, 91, 10))
    pcm = np.stack([synth.sine_stream(xs, np.deg2rad(d), fs, (F + 1) * N // 2, 1000.0, 5000.0) for d in angles])
    loc = api.MultibandBinarualLocalisation(fs, xs, 25, False, max_arrays=len(angles))
    seen = []
    loc.set_callback(lambda doa, prob, power, n: seen.append(doa[0]))
    r = loc.process(pcm)
    assert np.all(r["voiced"] == 1) and len(seen) == F
    for i, d in enumerate(angles):
        assert np.all(np.abs(np.rad2deg(r["doa"][i]) - d) <= 15.0), (d, np.rad2deg(r["doa"][i]))
    loc.close()


def test_multiband_power_gate_and_state_across_calls():
    # after the 3 s estimation the reference compares the LINEAR frame power with the dB floor (:221,:225);
    # a faint background (power 4 -> floor 9 dB) stays below it, bursts of power ~1e4 fire
    fs, N, F = 48000, 1024, 230
    hop = N // 2
    xs = synth.BINAURAL
    rng = np.random.default_rng(5)
    L = (F + 1) * hop
    src = synth.noise_source_stream(xs, np.deg2rad(30.0), fs, L, 6).astype(np.float64)
    src *= 100.0 / src.std()
    env = np.zeros(L)
    for a, b in ((150, 165), (172, 180), (190, F - 1)):
        env[a * hop:b * hop] = 1.0
    pcm = (2.0 * rng.standard_normal((2, L)) + src * env).astype(np.float32)
    og = po.Multiband(fs, xs, N + 2, 15, True)
    loc = api.MultibandBinarualLocalisation(fs, xs, 15, True)
    r = loc.process(pcm, want_bands=True)
    _compare(loc, og, pcm, N, r)
    assert r["voiced"][0, :141].sum() == 0 and 0 < r["voiced"][0].sum() < F - 141
    quiet = r["voiced"][0] == 0
    assert np.all(r["prob"][0][quiet] == -100000.0)
    # gated-out frames keep the previous DOA (_doaMemoryFactorSilence = 1)
    t = 166
    assert not r["voiced"][0, t] and r["doa"][0, t] == r["doa"][0, 164]
    # the same stream in two calls: correlation memory, gate and DOA carry over
    two = api.MultibandBinarualLocalisation(fs, xs, 15, True)
    cut = 100
    ra = two.process(pcm[:, :(cut + 1) * hop], want_bands=True)
    rb = two.process(pcm[:, cut * hop:], want_bands=True)
    for k in ("voiced", "doa", "prob", "band_idx"):
        assert np.array_equal(np.concatenate([ra[k], rb[k]], axis=1), r[k]), k
    np.testing.assert_allclose(np.concatenate([ra["band_corr"], rb["band_corr"]], axis=1), r["band_corr"], rtol=0, atol=1e-5)
    two.reset()
    rc = two.process(pcm)
    assert np.array_equal(rc["voiced"], r["voiced"]) and np.array_equal(rc["doa"], r["doa"])


def test_multiband_invalid_arguments():
    with pytest.raises(api.MCArrayHipError):
        api.MultibandBinarualLocalisation(48000, synth.ULA8)                       # needs 2 microphones
    with pytest.raises(api.MCArrayHipError):
        api.MultibandBinarualLocalisation(48000, synth.BINAURAL, nbins=40)         # 40 x 37 > 1024 threads
    loc = api.MultibandBinarualLocalisation(48000, synth.BINAURAL)
    with pytest.raises(api.MCArrayHipError):
        loc.process(np.zeros((2, 2, 3 * 512), np.float32))                         # more arrays than max_arrays
    with pytest.raises(api.MCArrayHipError):
        loc.process(np.zeros((1, 2, 1000), np.float32))                            # not (F+1)*hop samples


def test_multiband_matches_golden(golden_dir):
    import os
    g = np.load(os.path.join(golden_dir, "multiband_48k_b15.npz"))
    loc = api.MultibandBinarualLocalisation(int(g["fs"]), g["xs"], int(g["nbins"]), False)
    r = loc.process(g["pcm"], want_bands=True)
    assert np.array_equal(r["band_idx"][0], g["band_idx"])
    scale = np.abs(g["band_corr"]).max()
    assert np.abs(r["band_corr"][0] - g["band_corr"]).max() <= 2e-5 * scale
    np.testing.assert_allclose(r["energy_in_doa"][0], g["energy_in_doa"], rtol=1e-4, atol=1e-7 * g["energy_in_doa"].max())
    assert np.array_equal(r["doa"][0], g["doa"].astype(np.float32))
    np.testing.assert_allclose(r["prob"][0], g["prob"], atol=1e-4)
    np.testing.assert_allclose(r["power"][0], g["power"], rtol=1e-4)
