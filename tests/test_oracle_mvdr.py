"""CPU: the MVDR part of the C oracle (oracle/mca_oracle.c, SURVEY A.9 -- BASELINE.json configs[3]) against the golden
vectors of the independent numpy twin (full-matrix numpy.linalg.solve instead of Cholesky + forward substitutions)
and against the properties the definition implies.  There is no reference counterpart (the reference's only
beamformer is the delay-and-sum of Beamformer.cpp:51-71); the one link to the reference is that w = d/M must
reproduce that delay-and-sum, which is checked here against the oracle's Beamformer restatement."""
import os

import numpy as np
import pytest

from mcarray_amd import synth
from oracle import np_twin as tw
from oracle import pyoracle as po


def _spec(o):
    return o["spec"][:, 0::2] + 1j * o["spec"][:, 1::2]


@pytest.mark.parametrize("name", ["mvdr_ula16_48k", "mvdr_reemc_16k"])
def test_mvdr_matches_golden(golden_dir, name):
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    fs, N = int(g["fs"]), int(g["N"])
    m = po.MVDR(fs, N, g["xs"])
    o = m.stream(g["pcm"].astype(np.float64), g["doa"].astype(np.float64), want_spec=True)
    sc = np.abs(g["spec"]).max()
    assert np.abs(_spec(o) - g["spec"]).max() <= 2e-6 * sc            # golden stored as complex64
    assert np.abs(o["out"] - g["out"]).max() <= 2e-6 * np.abs(g["out"]).max()
    phi = m.covariance()[::64]
    assert np.abs(phi - g["phi_last"]).max() <= 2e-6 * np.abs(g["phi_last"]).max()


def test_mvdr_matches_twin_on_random_input():
    rng = np.random.default_rng(3)
    fs, N, F = 16000, 256, 9
    xs = np.sort(rng.uniform(0, 0.3, 6))
    pcm = rng.standard_normal((6, (F + 1) * N // 2)) * 0.1
    doa = rng.uniform(-1.4, 1.4, F)
    o = po.MVDR(fs, N, xs, alpha=0.8, loading=1e-2).stream(pcm, doa, want_spec=True)
    r = tw.mvdr_stream(fs, N, xs, pcm, doa, alpha=0.8, loading=1e-2)
    assert np.abs(_spec(o) - r["spec"]).max() <= 1e-10 * np.abs(r["spec"]).max()
    assert np.abs(o["out"] - r["out"]).max() <= 1e-10 * np.abs(r["out"]).max()


def test_mvdr_is_distortionless():
    """w^H d = 1: a frame that is an exact plane wave from the look direction, x = s d, comes out as s whatever the covariance."""
    rng = np.random.default_rng(4)
    fs, N, M = 48000, 1024, 8
    xs = np.asarray(synth.ULA8)
    K = N // 2 + 1
    m = po.MVDR(fs, N, xs)
    # some history from another direction, so the covariance is far from white
    pcm = synth.noise_source_stream(xs, np.deg2rad(-40.0), fs, 12 * N // 2, 9).astype(np.float64)
    m.stream(pcm, -0.2)
    theta = 0.35
    k = np.arange(K)
    d = np.exp(1j * 2 * np.pi * k[None, :] * fs * xs[:, None] * np.sin(theta) / (N * synth.C_SOUND))
    s = rng.standard_normal(K) + 1j * rng.standard_normal(K)
    frames = np.zeros((M, N + 2))
    frames[:, 0::2] = (s * d).real
    frames[:, 1::2] = (s * d).imag
    y = m.process_frame(frames, theta)
    np.testing.assert_allclose(y[0::2] + 1j * y[1::2], s, rtol=0, atol=1e-9)


def test_mvdr_with_heavy_loading_is_the_reference_delay_and_sum():
    """loading -> infinity makes PhiL a multiple of the identity: w = d/M, Beamformer::processFrame (Beamformer.cpp:51-71)."""
    fs, N = 48000, 1024
    xs = synth.REEM_C
    pcm = synth.noise_source_stream(xs, np.deg2rad(30.0), fs, 4 * N // 2, 2).astype(np.float64)
    X = po.stft_frames(pcm, N)
    m = po.MVDR(fs, N, xs, loading=1e12)
    for t in range(X.shape[0]):
        y = m.process_frame(X[t], 0.4)
        ref = po.beamformer_process_frame(fs, xs, X[t], 0.4)
        np.testing.assert_allclose(y, ref, rtol=0, atol=1e-9 * np.abs(ref).max())


def test_mvdr_silence_falls_back_to_delay_and_sum():
    fs, N = 16000, 512
    xs = synth.REEM_C
    m = po.MVDR(fs, N, xs)
    z = np.zeros((4, N + 2))
    y = m.process_frame(z, 0.1)
    assert np.all(y == 0) and np.all(np.isfinite(y))
    # the first non-silent frame works from a rank-one covariance plus loading
    X = po.stft_frames(synth.noise_source_stream(xs, 0.3, fs, 2 * N // 2, 1).astype(np.float64), N)
    y = m.process_frame(X[0], 0.3)
    assert np.all(np.isfinite(y)) and np.abs(y).max() > 0


def test_mvdr_suppresses_an_interferer_better_than_delay_and_sum():
    """A broadband source away from the look direction, converged covariance: the MVDR output carries at least 10 dB less of
    it than the reference's delay-and-sum of the same frames (Beamformer.cpp:51-71)."""
    fs, N, F = 16000, 512, 120
    xs = np.asarray(synth.ULA8)
    itf = synth.noise_source_stream(xs, np.deg2rad(-50.0), fs, (F + 1) * N // 2, 22, snr_db=40).astype(np.float64)
    look = float(np.deg2rad(10.0))
    X = po.stft_frames(itf, N)
    m = po.MVDR(fs, N, xs)
    p_mvdr = p_das = 0.0
    for t in range(F):
        y = m.process_frame(X[t], look)
        if t >= 80:
            p_mvdr += np.sum(y ** 2)
            p_das += np.sum(po.beamformer_process_frame(fs, xs, X[t], look) ** 2)
    assert 10 * np.log10(p_das / p_mvdr) > 10.0, 10 * np.log10(p_das / p_mvdr)
