"""Deterministic synthetic multichannel input (host side, numpy only).

Plane-wave far-field model of SURVEY A.2: a source at angle theta (rad, >0
towards +x) reaches microphone m as x_m(t) = s(t + x_m*sin(theta)/c); the
fractional delays are applied in the frequency domain over the whole stream.
Geometries are the [BUILD-DEFINES] ones of SURVEY 8d plus the reference's
Reem-C array (test/test_mcarray.cpp:397).
"""
import numpy as np

C_SOUND = 346.1  # src/mcarray/microhponeArrayHelpers.cpp:42

ULA8 = [0.04 * m for m in range(8)]
ULA16 = [0.02 * m for m in range(16)]
BINAURAL = [0.0, 0.086]
REEM_C = [0.0, 0.07, 0.175, 0.21]


def delay_channels(s, xs, theta, fs):
    """s [L] -> [M][L] with x_m(t) = s(t + x_m sin(theta)/c) (circular over the stream)."""
    L = len(s)
    S = np.fft.rfft(s)
    f = np.fft.rfftfreq(L, d=1.0 / fs)
    xs = np.asarray(xs, dtype=np.float64)
    adv = xs * np.sin(theta) / C_SOUND  # seconds
    return np.fft.irfft(S[None, :] * np.exp(2j * np.pi * f[None, :] * adv[:, None]), n=L, axis=1)


def noise_source_stream(xs, theta, fs, n_samples, seed, sigma=0.1, snr_db=20.0):
    """White Gaussian source at `theta` plus independent sensor noise; float32 [M][n_samples]."""
    rng = np.random.default_rng(seed)
    s = rng.standard_normal(n_samples) * sigma
    x = delay_channels(s, xs, theta, fs)
    nstd = sigma * 10.0 ** (-snr_db / 20.0)
    x = x + rng.standard_normal(x.shape) * nstd
    return np.clip(x, -1.0, 1.0 - 2 ** -23).astype(np.float32)


def sine_stream(xs, theta, fs, n_samples, freq, amplitude=0.5, phase=0.0):
    """Pure tone from angle theta (the 'sine_f_1000_fs_48000' files of test/test_mcarray.cpp:403)."""
    t = np.arange(n_samples) / fs
    xs = np.asarray(xs, dtype=np.float64)
    adv = xs * np.sin(theta) / C_SOUND
    return (amplitude * np.cos(2 * np.pi * freq * (t[None, :] + adv[:, None]) + phase))


def tone16(n, magn, freq, phase=0.0):
    """wipp::tone(int16*, n, magn, freq, phase) stand-in [BUILD-DEFINES]: round(A cos(2 pi f n + phase))."""
    return np.round(magn * np.cos(2 * np.pi * freq * np.arange(n) + phase)).astype(np.int16)
