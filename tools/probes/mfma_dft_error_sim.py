"""Numerical experiment (CPU, numpy): what would a single-product fp16 MFMA DFT (1024 = 32 x 32, two DFT-32 stages as fp16 matrix
products with fp32 accumulation, fp16 intermediate) do to the COARSE SRP map of the adaptive precision, against what the shipped
coarse pass does (fp32 transform, one fp16 rounding of the merged PHAT sums and of the steering table)?  Bench geometry (8-mic
ULA 0.04 m, 48 kHz, N = 1024, 361 angles), one far-field white source + sensor noise.  Prints the error of the map in units of
the shipped coarse error and of the decision margin tau (api.hip), and the tail of the per-bin error of the whitened spectra.
Round 4: a COLOURED source (colour = "speech": flat to 300 Hz, then -12 dB per octave: 76 dB down at 24 kHz -- the long-term slope of
speech and most real signals) next to the white one of the bench.  PHAT gives every bin the same weight whatever its level, and a
transform whose error is relative to the frame's RMS (fp16 operands) leaves the bins that lie 50 dB below it without a usable
phase; the share of bins that would need the exact fix-up (modulus below 30 x the transform's error) is printed.
usage: python tools/probes/mfma_dft_error_sim.py [frames] [snr_db] [white|speech]"""
import sys

import numpy as np

FS, N, M, D, C_SOUND, DX = 48000, 1024, 8, 361, 346.1, 0.04
f16 = lambda a: a.astype(np.float16).astype(np.float32)


def cf16(z):
    return f16(z.real) + 1j * f16(z.imag)


def main(frames=48, snr_db=20.0, theta_deg=23.0, seed=3, colour="white"):
    rng = np.random.default_rng(seed)
    L = (frames + 1) * (N // 2)
    s = rng.standard_normal(L + 64) * 0.1
    S = np.fft.rfft(s)
    f = np.fft.rfftfreq(L + 64, 1.0 / FS)
    if colour == "speech":
        S = S * np.minimum(1.0, (300.0 / np.maximum(f, 1.0)) ** 2)                                  # -12 dB per octave above 300 Hz
        S *= np.sqrt((s ** 2).sum() / max((np.fft.irfft(S, n=L + 64) ** 2).sum(), 1e-300))         # same total power as the white source
    xs = DX * np.arange(M)
    adv = xs * np.sin(np.deg2rad(theta_deg)) / C_SOUND
    x = np.fft.irfft(S[None, :] * np.exp(2j * np.pi * f[None, :] * adv[:, None]), n=L + 64, axis=1)[:, :L]
    x = x + rng.standard_normal((M, L)) * 0.1 * 10 ** (-snr_db / 20)
    x = x.astype(np.float32)
    win = (0.5 - 0.5 * np.cos(2 * np.pi * np.arange(N) / N)).astype(np.float32)
    fr = np.stack([x[:, t * N // 2:t * N // 2 + N] * win for t in range(frames)], axis=1)        # [M][F][N] fp32

    # exact spectra (double) and the MFMA-style ones
    X = np.fft.fft(fr.astype(np.float64), axis=2)[:, :, :N // 2 + 1]                               # [M][F][513]
    n1 = np.arange(32)
    F32 = np.exp(-2j * np.pi * np.outer(n1, n1) / 32)
    F16 = cf16(F32)
    tw = np.exp(-2j * np.pi * np.outer(n1, n1) / 1024).astype(np.complex64)                        # W_1024^(k1 n2)
    Xm = np.empty_like(X)
    for p in range(M // 2):
        z = fr[2 * p] + 1j * fr[2 * p + 1]                                                         # [F][N] pair packing
        scale = 2.0 ** np.ceil(np.log2(np.abs(z).max(axis=1) + 1e-30))                             # block scaling per frame
        zs = cf16(z / scale[:, None]).reshape(frames, 32, 32)                                      # [n1][n2], n = 32 n1 + n2
        y = np.einsum("kn,fnm->fkm", F16, zs.astype(np.complex64)).astype(np.complex64)            # stage 1 (fp32 accumulate)
        y = cf16(y * tw[None])                                                                     # twiddle in fp32, round to fp16
        zz = np.einsum("fkm,mq->fkq", y.astype(np.complex64), F16).astype(np.complex64)            # stage 2: Z[k1 + 32 k2]
        Z = zz.transpose(0, 2, 1).reshape(frames, N) * scale[:, None]                              # index k = k1 + 32 k2
        Zr = np.conj(np.roll(Z[:, ::-1], 1, axis=1))                                               # conj Z[N - k]
        Xm[2 * p] = (0.5 * (Z + Zr))[:, :N // 2 + 1]
        Xm[2 * p + 1] = (-0.5j * (Z - Zr))[:, :N // 2 + 1]

    def whiten(Xc):
        a = np.abs(Xc)
        return np.where(a > 1e-15, Xc / np.maximum(a, 1e-300), 0)

    Wx, Wm = whiten(X), whiten(Xm)
    err_bin = np.abs(Wm - Wx)                                                                      # [M][F][513]
    print("whitened spectra, |error| of the MFMA-style transform: rms %.2e  99%% %.2e  99.9%% %.2e  max %.2e   (fp16 rounding of a unit value: 2.4e-4 rms)"
          % (np.sqrt((err_bin ** 2).mean()), np.quantile(err_bin, 0.99), np.quantile(err_bin, 0.999), err_bin.max()))

    # bins that an exact fix-up would have to recompute: modulus below 30 x the transform's error (3e-4 of the frame's RMS spectrum)
    rms_spec = np.sqrt((np.abs(X) ** 2).mean(axis=2, keepdims=True))
    need = np.abs(X) < 30 * 3.0e-4 * rms_spec
    print("bins below 30 x the transform's error (exact fix-up needed): %.2f %% of all (channel, frame, bin) = %.1f per frame of %d microphones; "
          "bins whose whitened error exceeds 0.1 (phase off by > 6 degrees): %.2f %%" % (100 * need.mean(), need.mean() * M * (N // 2 + 1), M, 100 * (err_bin > 0.1).mean()))
    # merged PHAT sums (ULA: index m = k g) and the steering contraction
    tau1 = (DX * np.sin(np.deg2rad(np.arange(D) * 0.5 - 90.0)) / C_SOUND * FS)                      # delay of spacing 1, samples
    def srp(W, round_sums):
        C = np.zeros((W.shape[1], D))
        ms = {}
        for g in range(1, M):
            Sg = sum(W[i] * np.conj(W[i + g]) for i in range(M - g))                               # [F][513]
            for k in range(N // 2 + 1):
                ms.setdefault(k * g, []).append(Sg[:, k])
        keys = sorted(ms)
        A = np.stack([sum(ms[m]) for m in keys], axis=1)                                           # [F][n_merged]
        B = np.exp(2j * np.pi * np.outer(np.array(keys), tau1) / N)                                 # [n_merged][D]
        if round_sums:
            A, B = cf16(A.astype(np.complex64)), cf16(B.astype(np.complex64))
        return (A.real @ B.real - A.imag @ B.imag)                                                 # Re(A B) per frame and angle
    C_exact = srp(Wx, False)
    C_ship = srp(Wx, True)                                                                         # shipped coarse pass: fp32 transform, fp16 operands
    C_mfma = srp(Wm, True)
    P = M * (M - 1) // 2
    e_ship, e_mfma = C_ship - C_exact, C_mfma - C_exact
    # decision margin of api.hip: sigma_C = 5e-4 sqrt(K/2 sum_g n_g^2), tau = 8 sqrt(2) 1.25 sigma_C in map units
    sum_n2 = sum((M - g) ** 2 for g in range(1, M))
    tau_map = 8 * np.sqrt(2) * 1.25 * 5.0e-4 * np.sqrt(0.5 * (N // 2 + 1) * sum_n2)
    print("coarse map error (map units; peak of the map %.0f):" % np.abs(C_exact).max())
    print("   shipped coarse pass : rms %.3f  max %.3f  = %.2f tau" % (np.sqrt((e_ship ** 2).mean()), np.abs(e_ship).max(), np.abs(e_ship).max() / tau_map))
    print("   MFMA-style transform: rms %.3f  max %.3f  = %.2f tau   (%.1f x the shipped rms)" % (
        np.sqrt((e_mfma ** 2).mean()), np.abs(e_mfma).max(), np.abs(e_mfma).max() / tau_map, np.sqrt((e_mfma ** 2).mean()) / np.sqrt((e_ship ** 2).mean())))


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 48, float(sys.argv[2]) if len(sys.argv) > 2 else 20.0, colour=sys.argv[3] if len(sys.argv) > 3 else "white")
