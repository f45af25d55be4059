"""Channels whose levels are far apart (round 6; csrc/pair_balance.h).

The reference transforms every channel on its own (dsp::STFT(M, order), SourceSeparationAndLocalisation.cpp:52) and GCC-PHAT reads each
channel's own spectrum (SteeringBeamforming.cpp:110-119).  The wave-level analysis kernels of this build transform two channels as ONE
complex sequence; before round 6 the weaker channel then carried 2^-23 of the STRONGER one's magnitude as rounding noise, and PHAT, which
keeps only the phase, turned that into a map error of up to 3e-3 of the peak in every SRP precision (round 5's fuzz, seed 7305: a digitally
muted channel whose mute edges fall INSIDE a frame).  These tests feed exactly that input -- mute edges that are not hop-aligned, channels
60 / 80 / 100 dB below their partners -- to the 8-microphone ULA, an irregular 4-microphone array and the 16-microphone ULA, in all four
SRP precisions, in two calls (the state handed over), through host pointers and (lazy tails, candidate columns) device pointers, and hold
the results to the same bars as every other stream test: map <= the mode's tolerance, bins by the oracle-fragility bar, audio on every hop
whose bins agree."""
import os

import numpy as np
import pytest

from mcarray_amd import api, synth
from oracle import pyoracle as po
from parity_helpers import assert_audio_where_bins_agree, assert_bins, classify_bins

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

TOL_E = {api.SRP_FP32: 2e-5, api.SRP_FP16X3: 2e-5, api.SRP_FP16: 2e-4, api.SRP_ADAPTIVE: 2e-4}
IRR4 = [0.0, 0.031, 0.118, 0.164]            # an irregular array: one delay table per pair (no merged rows)


@pytest.fixture(autouse=True)
def adaptive_on_small_batches(monkeypatch):
    monkeypatch.setenv("MCA_HIP_ADAPT_FALLBACK", "0")
    monkeypatch.setenv("MCA_HIP_ADAPT_MIN_ROWS", "128")


def uneven_streams(xs, fs, n_samples, seed, mute=(21 * 512 + 137, 58 * 512 + 401)):
    """four streams of one geometry: [0] channel 1 digitally muted over a stretch whose edges fall inside frames, [1..3] one channel
    60 / 80 / 100 dB below the others (an odd and an even channel: either half of a transform pair); returns pcm [4][M][n]"""
    M = len(xs)
    pcm = np.stack([synth.noise_source_stream(xs, np.deg2rad(th), fs, n_samples, seed + i, snr_db=15.0) for i, th in enumerate((31.0, -52.0, 8.0, 67.0))]).astype(np.float32)
    pcm[0, 1, mute[0]:mute[1]] = 0.0                          # NOT hop-aligned: frames 20, 21 and 57, 58 hold the window's taper of the channel only
    pcm[1, 2] *= np.float32(10.0 ** (-60 / 20))
    pcm[2, M - 1] *= np.float32(10.0 ** (-80 / 20))
    pcm[3, 0] *= np.float32(10.0 ** (-100 / 20))
    return pcm


def check_against_oracle(r, pcm, fs, N, xs, S, step, prec, P, gate=False, weighting="phat", max_ties=3):
    hop = N // 2
    for a in range(pcm.shape[0]):
        if gate:
            o = po.ssl_stream_gated(fs, N, xs, pcm[a].astype(np.float64), S, step, True)
        else:
            o = po.ssl_stream(fs, N, xs, pcm[a].astype(np.float64), S, step, want_map=True, weighting=weighting)
        scale = np.abs(o["energy"]).max()
        err = np.abs(r["energy"][a] - o["energy"]).max() / scale
        assert err <= TOL_E[prec], "array %d: energy map error %.2e of the peak (allowed %.0e)" % (a, err, TOL_E[prec])
        assert_bins(r["bin"][a], o["bin"], o["energy"], P, max_ties=max_ties)
        nout = o["out"].shape[0]
        assert_audio_where_bins_agree(r["out"][a][:nout], o["out"], r["bin"][a], o["bin"], hop)


@pytest.mark.parametrize("prec", [api.SRP_FP32, api.SRP_FP16X3, api.SRP_FP16, api.SRP_ADAPTIVE])
@pytest.mark.parametrize("name,xs,step", [("ULA8", synth.ULA8, 0.5), ("IRR4", IRR4, 1.0), ("ULA16", synth.ULA16, 1.0)])
def test_uneven_channel_levels_match_the_oracle(name, xs, step, prec):
    fs, N, F, cut = 48000, 1024, 150, 71
    pcm = uneven_streams(xs, fs, (F + 1) * 512, 7300)
    ctx = api.Context(fs, xs, N, step, 1, srp_precision=prec, max_arrays=pcm.shape[0])
    ra = ctx.process_frames_host(pcm[:, :, :(cut + 1) * 512], want_energy=True)
    rb = ctx.process_frames_host(pcm[:, :, cut * 512:], want_energy=True)
    r = {k: np.concatenate([ra[k], rb[k]], axis=2 if k == "out" else 1) for k in ("bin", "energy", "out")}
    check_against_oracle(r, pcm, fs, N, xs, 1, step, prec, ctx.P)
    ctx.close()


def test_uneven_levels_with_two_sources_the_gate_and_without_phat():
    """the other instantiations of the analysis kernel: S = 2 (FP16X3 rows, two planes), the power gate (the frame power is the
    UNSCALED channels'), gcc_weighting NONE (X itself, not its phase: a scale left on the weak channel would show as 60 dB)"""
    fs, N, F = 48000, 1024, 120
    xs = synth.ULA8
    pcm = uneven_streams(xs, fs, (F + 1) * 512, 7400)
    ctx = api.Context(fs, xs, N, 1.0, 2, srp_precision=api.SRP_FP16X3, max_arrays=4)
    r = ctx.process_frames_host(pcm, want_energy=True)
    check_against_oracle(r, pcm, fs, N, xs, 2, 1.0, api.SRP_FP16X3, ctx.P, max_ties=6)
    ctx.close()
    Fg = 340                                                   # (the gate estimates its floor over the first 3 s = 282 frames: the mute comes after them)
    pcm_g = uneven_streams(xs, fs, (Fg + 1) * 512, 7450, mute=(295 * 512 + 137, 322 * 512 + 401))
    pcm_g[:, :, 288 * 512:] *= np.float32(4.0)                 # (... and the stream gets 12 dB louder: above the floor + 3 dB margin)
    ctx = api.Context(fs, xs, N, 1.0, 1, use_power_floor=True, srp_precision=api.SRP_FP32, max_arrays=4)
    r = ctx.process_frames_host(pcm_g, want_energy=True)
    assert (r["bin"][:, 292:, 0] >= 0).all()                   # (voiced)
    check_against_oracle(r, pcm_g, fs, N, xs, 1, 1.0, api.SRP_FP32, ctx.P, gate=True)
    ctx.close()
    ctx = api.Context(fs, xs, N, 1.0, 1, srp_precision=api.SRP_FP32, max_arrays=4, gcc_weighting=api.GCC_NONE)
    r = ctx.process_frames_host(pcm, want_energy=True)
    check_against_oracle(r, pcm, fs, N, xs, 1, 1.0, api.SRP_FP32, ctx.P, weighting="none")
    ctx.close()


@pytest.mark.parametrize("name,xs", [("ULA8", synth.ULA8), ("IRR4", IRR4)])
def test_uneven_levels_through_lazy_tails_and_candidate_columns(name, xs, monkeypatch):
    """device pointers, four calls (mca_hip_process_frames_dev): the coarse launch, the list-mode launch of the repair pass and the kept
    history all run the same balanced transform; a wide decision margin flags frames around the mute edges too"""
    monkeypatch.setenv("MCA_HIP_ADAPT_TAU_SCALE", "20")
    monkeypatch.setenv("MCA_HIP_ADAPT_CAND", "1")
    monkeypatch.setenv("MCA_HIP_ADAPT_MIN_ROWS", "64")
    fs, N, hop = 48000, 1024, 512
    sizes = [64, 80, 65, 96]
    F = sum(sizes)
    pcm = uneven_streams(xs, fs, (F + 1) * hop, 7500)
    pcm[0, 1, 100 * hop + 77:171 * hop + 300] = 0.0            # a second mute, across two call boundaries
    ctx = api.Context(fs, xs, N, 0.5, 1, srp_precision=api.SRP_ADAPTIVE, max_arrays=4, adaptive_fallback=False)
    ctx.reset_timing()
    dev = torch.device("cuda:0")
    parts, t0 = {"bin": [], "energy": [], "out": []}, 0
    for Fi in sizes:
        x = torch.from_numpy(np.ascontiguousarray(pcm[:, :, t0 * hop:(t0 + Fi + 1) * hop])).to(dev)
        b = torch.empty(4, Fi, 1, dtype=torch.int32, device=dev)
        rr = torch.empty(4, Fi, 1, dtype=torch.float32, device=dev)
        q = torch.empty(4, Fi, 1, dtype=torch.float32, device=dev)
        e = torch.empty(4, Fi, ctx.D, dtype=torch.float32, device=dev)
        o = torch.empty(4, 1, Fi * hop, dtype=torch.float32, device=dev)
        ctx.process_frames_dev(x, Fi, b, rr, q, e, o)
        torch.cuda.synchronize()
        parts["bin"].append(b.cpu().numpy()); parts["energy"].append(e.cpu().numpy()); parts["out"].append(o.cpu().numpy())
        t0 += Fi
    r = {k: np.concatenate(v, axis=2 if k == "out" else 1) for k, v in parts.items()}
    st = ctx.repair_stats()
    assert st["flagged"] > 0, st
    check_against_oracle(r, pcm, fs, N, xs, 1, 0.5, api.SRP_ADAPTIVE, ctx.P, max_ties=6)
    ctx.close()


def test_a_channel_with_one_sample_under_the_windows_zero_is_a_channel_of_zeros():
    """periodic Hann: w[0] = 0.  A channel whose only non-zero sample of a frame meets w[0] has an all-zero windowed frame: X = 0 in the
    reference's own transform, and here (alive = a non-zero WINDOWED sample), not its partner's rounding noise whitened to unit modulus"""
    fs, N, F = 48000, 1024, 40
    xs = synth.ULA8
    pcm = synth.noise_source_stream(xs, np.deg2rad(-24.0), fs, (F + 1) * 512, 77, snr_db=15.0).astype(np.float32)[None].copy()
    pcm[0, 5, :] = 0.0
    pcm[0, 5, 10 * 512] = 0.25                                # sample 0 of frame 10, sample 512 of frame 9
    for prec in (api.SRP_FP32, api.SRP_FP16):
        ctx = api.Context(fs, xs, N, 0.5, 1, srp_precision=prec, max_arrays=1)
        r = ctx.process_frames_host(pcm, want_energy=True)
        check_against_oracle(r, pcm, fs, N, xs, 1, 0.5, prec, ctx.P)
        ctx.close()


@pytest.mark.parametrize("prec", [api.SRP_FP32, api.SRP_FP16X3])
def test_uneven_levels_at_512_sample_frames(prec):
    """16 kHz / 512-sample frames (k_stft_phat_512: two channels per 512-point complex transform, fft512.h rfft512_pair): the same
    balance -- and a muted channel gives X = 0 there too, not its partner's rounding noise whitened to unit modulus"""
    fs, N, F, cut = 16000, 512, 150, 71
    xs = synth.ULA8
    pcm = uneven_streams(xs, fs, (F + 1) * 256, 7600, mute=(21 * 256 + 77, 58 * 256 + 201))
    ctx = api.Context(fs, xs, N, 1.0, 1, srp_precision=prec, max_arrays=4)
    ra = ctx.process_frames_host(pcm[:, :, :(cut + 1) * 256], want_energy=True)
    rb = ctx.process_frames_host(pcm[:, :, cut * 256:], want_energy=True)
    r = {k: np.concatenate([ra[k], rb[k]], axis=2 if k == "out" else 1) for k in ("bin", "energy", "out")}
    check_against_oracle(r, pcm, fs, N, xs, 1, 1.0, prec, ctx.P)
    ctx.close()


def test_multiband_localiser_with_one_ear_far_below_the_other():
    """MultibandBinarualLocalisation at 16 kHz (k_mb_analyse_512: the two channels of a frame are ONE 512-point transform;
    MultibandBinarualLocalisation.cpp:164-196 whitens the cross-spectrum): right channel 80 dB down, left channel muted for a while"""
    from test_gpu_multiband import _compare
    fs, N, nbins, F, A = 16000, 512, 15, 70, 2
    xs = synth.BINAURAL
    pcm = np.stack([synth.noise_source_stream(xs, np.deg2rad(-40.0 + 70.0 * a), fs, (F + 1) * N // 2, 60 + a) for a in range(A)]).astype(np.float32)
    pcm[0, 1] *= np.float32(1e-4)
    pcm[1, 0, 20 * 256 + 31:41 * 256 + 200] = 0.0
    loc = api.MultibandBinarualLocalisation(fs, xs, nbins, False, max_arrays=A)
    r = loc.process(pcm, want_bands=True)
    flagged = 0
    for a in range(A):
        flagged += _compare(loc, po.Multiband(fs, xs, N + 2, nbins, False), pcm[a], N, r, a)
    assert flagged <= 0.4 * A * F, flagged                     # (a muted ear: flat band correlations, ties by construction)
    loc.close()
