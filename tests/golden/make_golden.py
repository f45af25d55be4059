"""Generates the golden fixtures in this directory (run from the repo root:
`python tests/golden/make_golden.py`).

The reference cannot be executed here, so the vectors come from the
numpy twin (oracle/np_twin.py) -- an implementation independent of the C oracle --
on seeded synthetic inputs.  tests/test_oracle_golden.py checks the C oracle
against them on CPU; tests/test_gpu_parity.py checks the HIP path against them
on the GPU box.  Inputs are stored as float32 PCM (what the C-ABI consumes).
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from mcarray_amd import synth  # noqa: E402
from oracle import np_twin as tw  # noqa: E402


def ssl_case(name, xs, fs, N, step_deg, theta_deg, F, seed, n_sources=1):
    pcm = synth.noise_source_stream(xs, np.deg2rad(theta_deg), fs, (F + 1) * N // 2, seed)
    r = tw.ssl_stream(fs, N, xs, pcm.astype(np.float64), n_sources, step_deg)
    np.savez_compressed(os.path.join(HERE, name + ".npz"),
                        xs=np.asarray(xs), fs=fs, N=N, step_deg=step_deg, theta_deg=theta_deg, n_sources=n_sources,
                        pcm=pcm, bin=r["bin"], doa=r["doa"], prob=r["prob"],
                        out=r["out"].astype(np.float32), energy=r["energy"])
    print(name, "bins", r["bin"][:, 0])


def masking_case(name, fs, N, d, flo, fhi, method, alg, F, seed, delay=0, nlev=0.01):
    rng = np.random.default_rng(seed)
    n = (F + 1) * N // 2
    s = rng.standard_normal(n) * 0.1
    left = s + rng.standard_normal(n) * nlev
    right = np.roll(s, delay) * 0.9 + rng.standard_normal(n) * nlev
    # amplitude steps exercise the temporal mask
    env = np.repeat([1.0, 0.2, 1.0, 0.05, 0.6, 1.0, 0.1, 1.0][:F + 1], N // 2)[:n]
    left = (left * env).astype(np.float32)
    right = (right * env).astype(np.float32)
    m = tw.Masking(fs, N, d, flo, fhi, method, alg)
    X = tw.stft_frames(np.stack([left, right]).astype(np.float64), N)
    hop = N // 2
    out = np.zeros((2, F * hop))
    tail = np.zeros((2, hop))
    decs = np.zeros((F, 45), dtype=np.int32)
    Qs = np.zeros((F, 45))
    for t in range(F):
        oL, oR, decs[t] = m.process(X[t, 0], X[t, 1])
        Qs[t] = m.Q
        for c, Y in enumerate((oL, oR)):
            y = tw.irfft_ccs(Y, N)
            out[c, t * hop:(t + 1) * hop] = tail[c] + y[:hop]
            tail[c] = y[hop:]
    np.savez_compressed(os.path.join(HERE, name + ".npz"), fs=fs, N=N, d=d, flo=flo, fhi=fhi, method=method, alg=alg,
                        left=left, right=right, out=out.astype(np.float32), decisions=decs, Q=Qs,
                        thresholds=m.thr, center=m.center)
    print(name, "decisions per frame (enh/temp/spat):",
          [(int((d == 0).sum()), int((d == 1).sum()), int((d == 2).sum())) for d in decs])


def freqgcc_case(name, fs, N, d, step_deg, theta_deg, F, seed):
    xs = [0.0, d]
    pcm = synth.noise_source_stream(xs, np.deg2rad(theta_deg), fs, (F + 1) * N // 2, seed)
    X = tw.stft_frames(pcm.astype(np.float64), N)
    step = tw.doa_step(step_deg)
    delays = tw.delay_table(fs, xs, step_deg)[0]
    D = len(delays)
    prev = np.zeros(D)
    mem = np.float32(0)
    corrs = np.zeros((F, D))
    idxs = np.zeros(F, dtype=np.int32)
    for t in range(F):
        c = tw.gcc_phat(X[t, 0], X[t, 1], delays, N // 2 + 1).real
        c = float(np.float32(1) - mem) * c + float(mem) * prev
        prev = c
        corrs[t] = c
        idxs[t] = int(np.argmax(c))
        mem = np.float32(0.8)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), xs=np.asarray(xs), fs=fs, N=N, step_deg=step_deg,
                        theta_deg=theta_deg, pcm=pcm, corr=corrs, argmax=idxs)
    print(name, "argmax", idxs, "expected angle", [float(np.rad2deg(tw.doaidx2angle(i, step))) for i in idxs[:2]])


def freqgcc_gated_case(name, fs, N, d, F, bursts, seed):
    """usePowerFloor = true with silences around the 3 s decay window (BinauralLocalisation.cpp:530-560): 16-bit PCM (a quiet
    floor of ~1.5 LSB, loud bursts from two directions), so the fixture stays small and the input is exact in float32."""
    xs = [0.0, d]
    hop = N // 2
    n = (F + 1) * hop
    srcs = [synth.noise_source_stream(xs, np.deg2rad(a), fs, n, seed + i) for i, a in enumerate((35.0, -50.0))]
    x = srcs[0] * (1.5 / 32768 / 0.1)
    for k, (b0, b1) in enumerate(bursts):
        x[:, b0 * hop:b1 * hop] = srcs[k % 2][:, b0 * hop:b1 * hop]
    q = np.round(x * 32768).astype(np.int16)
    r = tw.freqgcc_stream(fs, N, xs, q.astype(np.float32) / 32768, 3.0, True)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), xs=np.asarray(xs), fs=fs, N=N, step_deg=3.0, pcm_i16=q,
                        fired=r["fired"], restart=r["restart"], argmax=r["argmax"], doa=r["doa"], prob=r["prob"],
                        power=r["power"], corr_fired=r["corr"][r["fired"]].astype(np.float32))
    print(name, "fired", np.nonzero(r["fired"])[0], "restarts at", np.nonzero(r["restart"])[0])


def multiband_case(name, fs, N, d, nbins, theta_deg, F, seed):
    xs = [0.0, d]
    pcm = synth.noise_source_stream(xs, np.deg2rad(theta_deg), fs, (F + 1) * N // 2, seed)
    r = tw.multiband_stream(fs, N, xs, pcm.astype(np.float64), nbins, False)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), xs=np.asarray(xs), fs=fs, N=N, nbins=nbins, theta_deg=theta_deg,
                        pcm=pcm, doa=r["doa"], prob=r["prob"], power=r["power"], band_idx=r["band_idx"],
                        band_corr=r["band_corr"], energy_in_doa=r["energy_in_doa"])
    print(name, "doa", np.rad2deg(r["doa"]))


def mvdr_case(name, xs, fs, N, F, seed, look_deg, interferer_deg):
    n = (F + 1) * N // 2
    pcm = (synth.noise_source_stream(xs, np.deg2rad(look_deg), fs, n, seed)
           + synth.noise_source_stream(xs, np.deg2rad(interferer_deg), fs, n, seed + 100, snr_db=60)).astype(np.float32)
    doa = (np.deg2rad(look_deg) + 0.002 * np.arange(F)).astype(np.float32)      # a slowly moving look direction
    r = tw.mvdr_stream(fs, N, xs, pcm.astype(np.float64), doa.astype(np.float64))
    np.savez_compressed(os.path.join(HERE, name + ".npz"), xs=np.asarray(xs), fs=fs, N=N, pcm=pcm, doa=doa,
                        out=r["out"].astype(np.float32), spec=r["spec"].astype(np.complex64), phi_last=r["phi"][::64].astype(np.complex64))
    print(name, "out rms", float(np.sqrt(np.mean(r["out"] ** 2))))


if __name__ == "__main__":
    ssl_case("ssl_reemc_d37", synth.REEM_C, 48000, 1024, 5.0, 20.0, 8, 11)
    ssl_case("ssl_ula8_d361", synth.ULA8, 48000, 1024, 0.5, -33.0, 6, 12)
    ssl_case("ssl_reemc_d37_s2", synth.REEM_C, 48000, 1024, 5.0, -45.0, 6, 13, n_sources=2)
    masking_case("mask_relative_both", 16000, 1024, 0.086, 500.0, 5000.0, 1, 0, 7, 21)
    masking_case("mask_full_both", 16000, 1024, 0.086, 500.0, 5000.0, 3, 0, 7, 22)
    masking_case("mask_factor_temporal", 16000, 1024, 0.086, 500.0, 5000.0, 0, 2, 7, 23)
    masking_case("mask_noisy_spatial", 16000, 1024, 0.086, 500.0, 5000.0, 4, 1, 7, 24, delay=1, nlev=0.003)
    freqgcc_case("freqgcc_16k_d61", 16000, 1024, 0.086, 3.0, 30.0, 6, 31)
    # 8 kHz -> 512-sample frames (0.075 s, BinauralLocalisation.h:196), windowsToDecay = 3 * 8000 / 256 = 93: the second burst
    # fires after exactly 94 gated-out frames (the recursions restart), the third after 93 (they do not)
    freqgcc_gated_case("freqgcc_8k_gated_silence", 8000, 512, 0.086, 256, [(50, 56), (151, 157), (251, 256)], 61)
    multiband_case("multiband_48k_b15", 48000, 1024, 0.086, 15, -35.0, 8, 41)
    mvdr_case("mvdr_ula16_48k", synth.ULA16, 48000, 1024, 6, 51, 25.0, -40.0)
    mvdr_case("mvdr_reemc_16k", synth.REEM_C, 16000, 512, 10, 52, -15.0, 55.0)
