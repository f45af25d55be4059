"""Randomised parity sweep of the 2-channel masking module and the MVDR beamformer against the CPU oracle (a one-off check like
tools/fuzz_parity.py): random frame lengths, methods / algorithms, channel counts, geometries, memories, loadings, chunked calls.
usage (GPU box): python tools/fuzz_modules.py [cases] [seed]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mcarray_amd import api, synth  # noqa: E402
from oracle import pyoracle as po  # noqa: E402


def mask_case(rng):
    fs, N = [(8000, 512), (16000, 1024), (48000, 2048), (44100, 2048), (16000, 1024), (96000, 4096)][int(rng.integers(0, 6))]
    method = int(rng.choice([api.FACTOR, api.RELATIVE, api.FULL, api.NOISY]))
    alg = int(rng.integers(0, 3))
    hop, F = N // 2, int(rng.integers(2, 90))
    d = float(rng.uniform(0.05, 0.2))
    n = (F + 1) * hop
    src = rng.standard_normal(n) * 0.1
    nl = float(rng.choice([0.003, 0.03]))
    left = src + rng.standard_normal(n) * nl
    right = np.roll(src, int(rng.integers(0, 3))) * float(rng.uniform(0.5, 1.0)) + rng.standard_normal(n) * nl
    env = np.repeat(rng.choice([1.0, 0.2, 0.05, 0.6], F + 1), hop)
    pcm = np.stack([left * env, right * env]).astype(np.float32)
    flo, fhi = float(rng.uniform(100, 600)), float(min(rng.uniform(3000, 7000), 0.45 * fs))
    tag = "mask fs=%d N=%d method=%d alg=%d F=%d" % (fs, N, method, alg, F)
    m = api.FastBinauralMasking(fs, d, flo, fhi, method, alg, fft_size=N)
    cut = int(rng.integers(1, F)) if F > 2 and rng.integers(0, 2) else 0
    if cut:
        oa, da = m.process(pcm[:, :(cut + 1) * hop]); ob, db = m.process(pcm[:, cut * hop:])
        out, dec = np.concatenate([oa[0], ob[0]], axis=1), np.concatenate([da[0], db[0]], axis=0)
    else:
        o_, d_ = m.process(pcm); out, dec = o_[0], d_[0]
    o = po.Masking(fs, N, d, flo, fhi, method, alg)
    ol, orr = o.stream(pcm[0].astype(np.float64), pcm[1].astype(np.float64))
    o2 = po.Masking(fs, N, d, flo, fhi, method, alg)
    X = po.stft_frames(pcm.astype(np.float64), N)
    odec = np.array([o2.process(X[t, 0], X[t, 1])[2] for t in range(F)])
    ndiff = int((dec != odec).sum())
    ref = np.stack([ol, orr])
    err = np.abs(out - ref).max() / (np.abs(ref).max() + 1e-30)
    ok = ndiff <= 2 and (ndiff > 0 or err <= 2e-5)
    m.close()
    return ok, "%s cut=%d decisions differ %d audio err %.1e" % (tag, cut, ndiff, err)


def mvdr_case(rng):
    fs, N = [(8000, 256), (16000, 512), (48000, 1024), (48000, 1024), (96000, 2048)][int(rng.integers(0, 5))]
    M = int(rng.integers(2, 17))
    xs = np.sort(rng.uniform(0, 0.03 * M, M))
    F, A = int(rng.integers(1, 60)), int(rng.integers(1, 4))
    alpha, loading = float(rng.choice([0.0, 0.5, 0.9, 0.95, 0.99])), float(rng.choice([1e-3, 1e-2, 1e-1]))
    hop = N // 2
    pcm = np.stack([synth.noise_source_stream(xs, rng.uniform(-1.3, 1.3), fs, (F + 1) * hop, int(rng.integers(1, 1 << 30)))
                    + synth.noise_source_stream(xs, rng.uniform(-1.3, 1.3), fs, (F + 1) * hop, int(rng.integers(1, 1 << 30)), snr_db=50)
                    for _ in range(A)]).astype(np.float32)
    doa = rng.uniform(-1.4, 1.4, (A, F)).astype(np.float32)
    tag = "mvdr fs=%d N=%d M=%d A=%d F=%d alpha=%.2f loading=%.0e" % (fs, N, M, A, F, alpha, loading)
    bf = api.MvdrBeamformer(fs, xs, N, alpha, loading, max_streams=A)
    cut = int(rng.integers(1, F)) if F > 2 and rng.integers(0, 2) else 0
    if cut:
        ra = bf.process(pcm[:, :, :(cut + 1) * hop], doa[:, :cut], want_spec=True)
        rb = bf.process(pcm[:, :, cut * hop:], doa[:, cut:], want_spec=True)
        out, spec = np.concatenate([ra["out"], rb["out"]], axis=1), np.concatenate([ra["spec"], rb["spec"]], axis=1)
    else:
        r = bf.process(pcm, doa, want_spec=True); out, spec = r["out"], r["spec"]
    worst = 0.0
    for a in range(A):
        o = po.MVDR(fs, N, xs, alpha, loading).stream(pcm[a].astype(np.float64), doa[a].astype(np.float64), want_spec=True)
        sp = o["spec"][:, 0::2] + 1j * o["spec"][:, 1::2]
        worst = max(worst, np.abs(spec[a] - sp).max() / np.abs(sp).max(), np.abs(out[a] - o["out"]).max() / np.abs(o["out"]).max())
    bf.close()
    return worst <= 5e-4, "%s cut=%d worst rel err %.1e" % (tag, cut, worst)


def main(cases, seed):
    rng = np.random.default_rng(seed)
    bad = 0
    for case in range(cases):
        for fn in (mask_case, mvdr_case):
            try:
                ok, msg = fn(rng)
            except api.MCArrayHipError as e:
                ok, msg = False, "%s raised %s" % (fn.__name__, e)
            print(("ok   " if ok else "FAIL ") + "case %d: %s" % (case, msg), flush=True)
            bad += 0 if ok else 1
    print("%d cases x 2, %d failures" % (cases, bad))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(int(sys.argv[1]) if len(sys.argv) > 1 else 40, int(sys.argv[2]) if len(sys.argv) > 2 else 0))
