"""MCA_HIP_SRP_ADAPTIVE: one fp16 MFMA product per k-step for every frame, then an exact repair (hi + lo operand planes,
three products) of the frames whose peak pick is sensitive to the fp16 error and of the rows their smoothed energy depends
on (replaces the all-pairs, all-delays double-precision loop nest of SteeringBeamforming.cpp:104-130 + :132-195 at about the
cost of the plain fp16 mode).  The DOA bins must be those of the exact paths: against the CPU oracle on sizes it can reach
(adaptive forced onto small batches), and against MCA_HIP_SRP_FP16X3 / MCA_HIP_SRP_FP32 on the GPU at full size."""
import os
import sys

import numpy as np
import pytest

from mcarray_amd import api, synth
from oracle import pyoracle as po

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


@pytest.fixture(autouse=True)
def no_back_off(monkeypatch):
    monkeypatch.setenv("MCA_HIP_ADAPT_FALLBACK", "0")        # read by mca_hip_create: these tests are about coarse + repair itself
    monkeypatch.setenv("MCA_HIP_ADAPT_MAX_SOURCES", "4")     # ... also with several sources (by default such contexts run as FP16X3)


@pytest.fixture
def force_small(monkeypatch):
    monkeypatch.setenv("MCA_HIP_ADAPT_MIN_ROWS", "256")       # read by mca_hip_create: adaptive also for small batches


from parity_helpers import assert_bins as _assert_bins   # the same bar as tests/test_gpu_parity.py


@pytest.mark.parametrize("xs,step,S,thetas", [(synth.ULA8, 0.5, 1, (23.0, -61.5, 79.0)), (synth.REEM_C, 5.0, 2, (-40.0, 10.0, 55.0)),
                                               (synth.ULA16, 1.0, 1, (5.0, -20.0, 70.0))])
def test_adaptive_matches_oracle(force_small, xs, step, S, thetas):
    fs, N, F = 48000, 1024, 330
    A = len(thetas)
    pcm = np.stack([synth.noise_source_stream(xs, np.deg2rad(th), fs, (F + 1) * N // 2, 900 + i, snr_db=20.0 - 8 * i) for i, th in enumerate(thetas)])
    ctx = api.Context(fs, xs, N, step, S, srp_precision=api.SRP_ADAPTIVE, max_arrays=A)
    ctx.reset_timing()
    cut = 150                                                   # two calls: the state handed over must be exact as well
    ra = ctx.process_frames_host(pcm[:, :, :(cut + 1) * 512], want_energy=True)
    rb = ctx.process_frames_host(pcm[:, :, cut * 512:], want_energy=True)
    r = {k: np.concatenate([ra[k], rb[k]], axis=2 if k == "out" else 1) for k in ("bin", "energy", "out", "prob")}
    st = ctx.repair_stats()
    assert st["frames"] == A * F and st["flagged"] >= 2 * A and st["recomputed"] >= st["flagged"]      # (host-pointer calls: every call repairs its own last rows)
    for a in range(A):
        o = po.ssl_stream(fs, N, xs, pcm[a].astype(np.float64), S, step, want_map=True)
        ties = _assert_bins(r["bin"][a], o["bin"], o["energy"], ctx.P, max_ties=3)
        assert np.abs(r["energy"][a] - o["energy"]).max() <= 2e-4 * np.abs(o["energy"]).max()      # fp16-level map on unrepaired frames
        if not ties:
            assert np.abs(r["out"][a] - o["out"]).max() <= 2e-5 * np.abs(o["out"]).max() + 1e-7
    ctx.close()


IRR5 = [0.0, 0.028, 0.071, 0.102, 0.155]


@pytest.mark.parametrize("fs,N", [(96000, 2048), (16000, 512)])
@pytest.mark.parametrize("name,xs,step,tau_scale", [("ULA8", synth.ULA8, 0.5, "1"), ("ULA8-wide-margin", synth.ULA8, 0.5, "30"), ("REEMC", synth.REEM_C, 5.0, "1"),
                                                     ("IRR5", IRR5, 1.0, "40")])
def test_adaptive_at_other_frame_lengths_matches_the_oracle(monkeypatch, name, xs, step, tau_scale, fs, N):
    """Round 6: the adaptive mode on 2048- and 512-sample frames (k_stft_phat_2048 / k_stft_phat_512: coarse one-plane rows for every
    frame, their list mode for the exact rows of the repair units; 3 ... 8 microphones) -- before, such contexts ran ADAPTIVE as plain
    FP16X3.  One source and two sources of equal strength per array (near ties: content flags; the wide decision margin flags many
    more), two calls (the state handed over must be the exact one), bins against the oracle at the tie rule of the exact paths."""
    monkeypatch.setenv("MCA_HIP_ADAPT_MIN_ROWS", "128")
    monkeypatch.setenv("MCA_HIP_ADAPT_TAU_SCALE", tau_scale)
    hop, F, cut, A = N // 2, 150, 72, 2
    pcm = np.stack([synth.noise_source_stream(xs, np.deg2rad(37.0), fs, (F + 1) * hop, 700, snr_db=15.0),
                    0.5 * synth.noise_source_stream(xs, np.deg2rad(-52.0), fs, (F + 1) * hop, 701, snr_db=10.0) +
                    0.5 * synth.noise_source_stream(xs, np.deg2rad(18.0), fs, (F + 1) * hop, 702, snr_db=10.0)]).astype(np.float32)
    ctx = api.Context(fs, xs, N, step, 1, srp_precision=api.SRP_ADAPTIVE, max_arrays=A)
    ctx.reset_timing()
    ra = ctx.process_frames_host(pcm[:, :, :(cut + 1) * hop], want_energy=True)
    rb = ctx.process_frames_host(pcm[:, :, cut * hop:], want_energy=True)
    r = {k: np.concatenate([ra[k], rb[k]], axis=2 if k == "out" else 1) for k in ("bin", "energy", "out")}
    st = ctx.repair_stats()
    assert st["frames"] == A * F and st["flagged"] >= 2 * A and st["recomputed"] >= st["flagged"], st      # (the mode ran: every call repairs its own last rows)
    if name == "ULA8-wide-margin":
        assert st["flagged"] > 4 * A + 10, st                                                               # ... and content flags with the wide margin
    from parity_helpers import assert_audio_where_bins_agree
    for a in range(A):
        o = po.ssl_stream(fs, N, xs, pcm[a].astype(np.float64), 1, step, want_map=True)
        _assert_bins(r["bin"][a], o["bin"], o["energy"], ctx.P, max_ties=3)
        assert np.abs(r["energy"][a] - o["energy"]).max() <= 2e-4 * np.abs(o["energy"]).max()      # fp16-level map on unrepaired frames
        assert_audio_where_bins_agree(r["out"][a][:o["out"].shape[0]], o["out"], r["bin"][a], o["bin"], hop)
    ctx.close()


def test_sixteen_microphones_two_work_lists_match_the_oracle(force_small, monkeypatch):
    """Round 6 (k_scan_pick<PL, 2>): a 16-microphone context sends the flagged frames that take whole rows by construction (the last
    frame of every array and call: eager tails; unsure rows) to k_srp_gemm_repair + k_repair_patch and every other flagged frame to
    k_srp_cand at its candidate columns -- with MCA_HIP_ADAPT_CAND=1 only: the mode measured no faster than whole rows for every flagged
    frame (profiles/r06_m16_two_lists_negative.log) and is not what the policy picks.  Two equal sources per array and a wide decision
    margin (many content flags) in two calls, against the oracle; both kinds of frames occurred; the whole-row form (MCA_HIP_ADAPT_CAND=0)
    flags the same frames."""
    fs, N, F, A, cut = 48000, 1024, 260, 3, 120
    xs = synth.ULA16
    monkeypatch.setenv("MCA_HIP_ADAPT_TAU_SCALE", "40")
    # two sources of equal strength per array (their peaks trade places from frame to frame) and a wide decision margin: content flags
    pcm = np.stack([(synth.noise_source_stream(xs, np.deg2rad(th), fs, (F + 1) * 512, 950 + i, snr_db=10.0) * 0.5 +
                     synth.noise_source_stream(xs, np.deg2rad(th2), fs, (F + 1) * 512, 960 + i, snr_db=10.0) * 0.5)
                    for i, (th, th2) in enumerate(((12.0, -30.0), (-47.0, 25.0), (66.0, -5.0)))]).astype(np.float32)
    o = [po.ssl_stream(fs, N, xs, pcm[a].astype(np.float64), 1, 1.0, want_map=True) for a in range(A)]
    from parity_helpers import classify_bins, assert_audio_where_bins_agree
    res = {}
    for cand in ("1", "0"):
        monkeypatch.setenv("MCA_HIP_ADAPT_CAND", cand)
        ctx = api.Context(fs, xs, N, 1.0, 1, srp_precision=api.SRP_ADAPTIVE, max_arrays=A)
        ctx.reset_timing()
        ra = ctx.process_frames_host(pcm[:, :, :(cut + 1) * 512], want_energy=True)
        rb = ctx.process_frames_host(pcm[:, :, cut * 512:], want_energy=True)
        r = {k: np.concatenate([ra[k], rb[k]], axis=2 if k == "out" else 1) for k in ("bin", "energy", "out")}
        res[cand] = (r, ctx.repair_stats(), ctx.repair_columns())
        for a in range(A):
            ties, bad = classify_bins(r["bin"][a], o[a]["bin"], o[a]["energy"], ctx.P)
            assert not bad, (cand, a, bad[:5])
            assert len(ties) <= 12
            assert np.abs(r["energy"][a] - o[a]["energy"]).max() <= 2e-4 * np.abs(o[a]["energy"]).max()
            assert_audio_where_bins_agree(r["out"][a][:1], o[a]["out"], r["bin"][a], o[a]["bin"], 512)
        ctx.close()
    st, cols = res["1"][1], res["1"][2]
    assert cols["whole_row_frames"] >= 2 * A, cols               # the last frame of every array and call at least
    assert st["flagged"] > cols["whole_row_frames"] + 10, (st, cols)   # ... and content flags that took the candidate kernel
    assert 0 < cols["candidate_columns"] <= 60 * (st["flagged"] - cols["whole_row_frames"]), (st, cols)
    assert res["0"][2] == {"candidate_columns": 0, "whole_row_frames": 0}
    assert res["0"][1]["flagged"] == st["flagged"]


def test_adaptive_backs_off_to_fp16x3_while_most_rows_need_the_repair(force_small, monkeypatch):
    """Noise only: every pick is a near tie, every frame is flagged, and coarse + repair of everything costs twice the direct
    exact pass.  The last kernel of an adaptive call reports its totals through page-locked memory; the call TWO calls later consumes
    the report (a fixed lag: round 6), the calls from there on run as plain FP16X3 (no adaptive frames counted) for 8 calls, then one
    call probes again.  The bins stay the exact path's."""
    monkeypatch.setenv("MCA_HIP_ADAPT_FALLBACK", "1")
    fs, N, F, A = 48000, 1024, 256, 1
    xs = synth.ULA8
    rng = np.random.default_rng(5)
    n_calls = 13
    pcm = (0.1 * rng.standard_normal((A, len(xs), (n_calls * F + 1) * 512))).astype(np.float32)
    ctx = api.Context(fs, xs, N, 0.5, 1, srp_precision=api.SRP_ADAPTIVE, max_arrays=A)
    ref = api.Context(fs, xs, N, 0.5, 1, srp_precision=api.SRP_FP16X3, max_arrays=A)
    ctx.reset_timing()
    counted, bins, rbins = [], [], []
    for i in range(n_calls):
        chunk = pcm[:, :, i * F * 512:((i + 1) * F + 1) * 512]
        bins.append(ctx.process_frames_host(chunk)["bin"])
        rbins.append(ref.process_frames_host(chunk)["bin"])
        counted.append(ctx.repair_stats()["frames"])
    # calls 0 and 1 adaptive (call 0's report is consumed by call 2), calls 2..9 suspended, call 10 probes (adaptive again), call 11 waits
    # for the probe's report, call 12 consumes it: suspended (for 16 calls)
    assert counted[0] == A * F and counted[1] == 2 * A * F and counted[9] == counted[1], counted
    assert counted[10] == 3 * A * F and counted[12] == counted[10], counted
    st = ctx.repair_stats()
    assert st["recomputed"] > 0.3 * st["frames"], st
    got, want = np.concatenate(bins, axis=1), np.concatenate(rbins, axis=1)
    # suspended calls ARE the FP16X3 path on the same state: the adaptive calls hand over an exact state, so the two contexts
    # can only part on exact-level ties (noise only: there are some)
    assert np.mean(got != want) < 0.02, np.mean(got != want)
    o = po.ssl_stream(fs, N, xs, pcm[0].astype(np.float64), 1, 0.5, want_map=True)
    _assert_bins(got[0], o["bin"], o["energy"], ctx.P, max_ties=int(0.02 * n_calls * F))
    ctx.close(); ref.close()


def test_the_back_off_policy_is_reproducible_run_to_run(force_small, monkeypatch):
    """north_star: "bit-exact for DOA peak indices"; SteeringBeamforming.cpp:146-195 is deterministic.  Under adaptive_fallback = AUTO the
    switch to FP16X3 (and the choice between candidate columns and whole-row repair kernels) hangs on reports the GPU sends to the
    host.  Round 6: every report is consumed by the call two calls after its own, which waits for it -- so one stream of six
    device-pointer calls (source, source, NOISE ONLY, source, source, source: calls 0..3 adaptive, call 4 consumes the report of call 2
    and runs FP16X3 with call 5) returns the same bits whether the calls are fired back to back without a single synchronisation or
    one at a time with the device drained and the host asleep in between (before: the second form saw every report a call earlier)."""
    import time
    monkeypatch.setenv("MCA_HIP_ADAPT_FALLBACK", "1")
    monkeypatch.setenv("MCA_HIP_ADAPT_MIN_ROWS", "128")
    fs, N, hop, A, F, n_calls = 48000, 1024, 512, 2, 128, 6
    xs = synth.ULA8
    total = F * n_calls
    pcm = np.stack([synth.noise_source_stream(xs, np.deg2rad(-33.0 + 70 * a), fs, (total + 1) * hop, 8100 + a, snr_db=10.0).astype(np.float32) for a in range(A)])
    rng = np.random.default_rng(81)
    pcm[:, :, 2 * F * hop + hop:3 * F * hop] = (0.1 * rng.standard_normal((A, len(xs), F * hop - hop))).astype(np.float32)    # call 2: independent noise on every channel
    dev = torch.device("cuda:0")
    x_all = torch.from_numpy(pcm).to(dev)

    def run(drain):
        ctx = api.Context(fs, xs, N, 0.5, 1, srp_precision=api.SRP_ADAPTIVE, max_arrays=A)      # adaptive_fallback AUTO (the default)
        ctx.reset_timing()
        outs = []
        for i in range(n_calls):
            x = x_all[:, :, i * F * hop:((i + 1) * F + 1) * hop].contiguous()
            b = torch.empty(A, F, 1, dtype=torch.int32, device=dev)
            r = torch.empty(A, F, 1, dtype=torch.float32, device=dev)
            q = torch.empty(A, F, 1, dtype=torch.float32, device=dev)
            e = torch.empty(A, F, ctx.D, dtype=torch.float32, device=dev)
            o = torch.empty(A, 1, F * hop, dtype=torch.float32, device=dev)
            ctx.process_frames_dev(x, F, b, r, q, e, o)
            outs.append((x, b, r, q, e, o))
            if drain:
                torch.cuda.synchronize()
                time.sleep(0.05)
        torch.cuda.synchronize()
        frames = ctx.repair_stats()["frames"]
        res = [np.concatenate([t[j].cpu().numpy() for t in outs], axis=-1 if j == 5 else 1) for j in range(1, 6)]
        ctx.close()
        return frames, res
    fa, ra = run(False)
    fb, rb = run(True)
    assert fa == fb == 4 * A * F, (fa, fb)                  # calls 0..3 adaptive, 4 and 5 FP16X3 -- in both runs
    for name, u, v in zip(("bin", "doa", "prob", "energy", "audio"), ra, rb):
        assert np.array_equal(u, v), name
    for a in range(A):
        o = po.ssl_stream(fs, N, xs, pcm[a].astype(np.float64), 1, 0.5, want_map=True)
        _assert_bins(ra[0][a], o["bin"], o["energy"], ctx_P(xs), max_ties=int(0.03 * total))


def ctx_P(xs):
    return len(xs) * (len(xs) - 1) // 2


def test_adaptive_context_runs_small_and_gated_calls_as_fp16x3():
    """below the batch-size threshold and with the power gate an ADAPTIVE context IS the FP16X3 path: bit-identical outputs"""
    fs, N, F, A = 48000, 1024, 70, 2
    xs = synth.ULA8
    pcm = np.stack([synth.noise_source_stream(xs, np.deg2rad(-30.0 + 50 * a), fs, (F + 1) * N // 2, 40 + a) for a in range(A)])
    for gate in (False, True):
        ra = api.Context(fs, xs, N, 0.5, 1, use_power_floor=gate, srp_precision=api.SRP_ADAPTIVE, max_arrays=A).process_frames_host(pcm, want_energy=True)
        rx = api.Context(fs, xs, N, 0.5, 1, use_power_floor=gate, srp_precision=api.SRP_FP16X3, max_arrays=A).process_frames_host(pcm, want_energy=True)
        for k in ("bin", "doa", "prob", "energy", "out"):
            assert np.array_equal(ra[k], rx[k]), (gate, k)


@pytest.mark.parametrize("kind,M,S,step", [("static", 8, 1, 0.5), ("noise", 8, 1, 0.5), ("two", 5, 2, 1.0), ("moving", 4, 3, 3.0)])
def test_adaptive_equals_exact_paths_at_full_size(kind, M, S, step):
    """GPU only (the oracle cannot reach these sizes): bins of ADAPTIVE == bins of FP16X3, except frames on which the two
    exact-level paths FP16X3 and FP32 themselves disagree (fp32-level ties); the plain fp16 mode flips on the same data."""
    import adaptive_check as ac
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(1234 + M + S)
    xs = (0.04 * np.arange(M)).tolist() if M == 8 else np.sort(rng.uniform(0, 0.05 * M, M)).tolist()
    A, F, cut = 8, 2304, 1031            # two calls of >= 8192 rows each: both run coarse + repair
    pcm = ac.synth(xs, A, F, kind, rng, dev)
    from oracle import np_twin as tw
    res, en = {}, None
    for name, prec in (("x3", api.SRP_FP16X3), ("fp32", api.SRP_FP32), ("adaptive", api.SRP_ADAPTIVE)):
        ctx = api.Context(ac.FS, xs, ac.N, step, S, srp_precision=prec, max_arrays=A)
        ctx.reset_timing()
        out = ac.run(ctx, pcm, F, S, cut)
        res[name] = out[0]
        if name == "x3":
            en, P = out[2], ctx.P
        if name == "adaptive":
            st = ctx.repair_stats()
        ctx.close()
    assert st["frames"] == A * F
    tie = (res["x3"] != res["fp32"]).any(dim=2)                # frames the exact-level paths do not agree on
    flips = ((res["adaptive"] != res["x3"]).any(dim=2) & ~tie).nonzero().tolist()
    assert len(flips) <= 4, (len(flips), int(tie.sum()), st)
    # a remaining difference must be a tie at the exact paths' own error level (FP16X3 vs the fp64 oracle: 2e-6 of the
    # map's peak): selectDOA of the FP16X3 energy row changes under perturbations of that size
    prng = np.random.default_rng(5)
    for a_, t_ in flips:
        E = en[a_, t_].double().cpu().numpy()
        base = tw.select_doa(E, P, tw.doa_step(step), S)[2]
        unstable = any(not np.array_equal(tw.select_doa(E + prng.standard_normal(E.shape) * 2e-6 * np.abs(E).max(), P, tw.doa_step(step), S)[2], base)
                       for _ in range(32))
        assert unstable, (a_, t_, res["adaptive"][a_, t_].tolist(), res["x3"][a_, t_].tolist())
    if kind == "static":
        assert st["flagged"] < 0.05 * A * F                    # a clear source: few frames need the exact rows


def _bursty(xs, A, F, seed, dev=None):
    """[A][M][(F+1)*512] float32: a quiet lead-in (floor estimation: 141 frames at 48 kHz), then loud bursts and quiet gaps"""
    rng = np.random.default_rng(seed)
    pcm = np.stack([synth.noise_source_stream(xs, np.deg2rad(rng.uniform(-70, 70)), 48000, (F + 1) * 512, seed + 10 + a, snr_db=15.0) for a in range(A)])
    env = np.full(F + 1, 0.004)
    t = 150
    while t < F:
        b = int(rng.integers(3, 60))
        env[t:t + b] = 1.0
        t += b + int(rng.integers(2, 50))
    return (pcm * np.repeat(env, 512)[None, None, :]).astype(np.float32)


def test_adaptive_with_power_gate_matches_oracle(force_small):
    """usePowerFloor = true (the reference's default): the recursion only advances on voiced frames, so the rows a flagged
    frame depends on are its last 25 VOICED ones -- possibly far back across a silence -- and gated-out frames repeat the
    last FINAL pick.  Against the oracle, two calls."""
    fs, N, F, A, S = 48000, 1024, 420, 3, 2
    xs = synth.ULA8
    pcm = _bursty(xs, A, F, 5)
    ctx = api.Context(fs, xs, N, 0.5, S, use_power_floor=True, srp_precision=api.SRP_ADAPTIVE, max_arrays=A)
    ctx.reset_timing()
    cut = 260
    ra = ctx.process_frames_host(pcm[:, :, :(cut + 1) * 512], want_energy=True)
    rb = ctx.process_frames_host(pcm[:, :, cut * 512:], want_energy=True)
    r = {k: np.concatenate([ra[k], rb[k]], axis=2 if k == "out" else 1) for k in ("bin", "energy", "out", "prob", "voiced")}
    st = ctx.repair_stats()
    assert st["frames"] == A * F and st["flagged"] >= A           # (gated contexts repair the last voiced frame of every call: no lazy tails)
    fired = 0
    for a in range(A):
        o = po.ssl_stream_gated(fs, N, xs, pcm[a].astype(np.float64), S, 0.5, True)
        assert np.array_equal(r["voiced"][a].astype(bool), o["fired"].astype(bool))
        fired += int(o["fired"].sum())
        ties = _assert_bins(r["bin"][a], o["bin"], o["energy"], ctx.P, max_ties=3)
        assert np.abs(r["energy"][a] - o["energy"]).max() <= 2e-4 * np.abs(o["energy"]).max()
        if not ties:
            assert np.abs(r["out"][a][:o["out"].shape[0]] - o["out"]).max() <= 2e-5 * np.abs(o["out"]).max() + 1e-7
    assert 100 < fired < A * (F - 141)
    ctx.close()


@pytest.mark.parametrize("fs,N,lead", [(96000, 2048, 141), (16000, 512, 94)])
def test_adaptive_with_power_gate_at_other_frame_lengths(monkeypatch, fs, N, lead):
    """the gated adaptive mode on the 2048- and 512-sample analysis kernels (FFTPower of the coarse launch, the planning wave's walk
    over the voiced flags, eager tails): a quiet lead-in of 3 s, then bursts; against the oracle in two calls"""
    monkeypatch.setenv("MCA_HIP_ADAPT_MIN_ROWS", "128")
    hop, A, F, cut = N // 2, 2, 330, 200
    xs = synth.ULA8
    rng = np.random.default_rng(12)
    pcm = np.stack([synth.noise_source_stream(xs, np.deg2rad(rng.uniform(-70, 70)), fs, (F + 1) * hop, 40 + a, snr_db=15.0) for a in range(A)])
    env = np.full(F + 1, 0.004)
    t = lead + 9
    while t < F:
        b = int(rng.integers(3, 40))
        env[t:t + b] = 1.0
        t += b + int(rng.integers(2, 30))
    pcm = (pcm * np.repeat(env, hop)[None, None, :]).astype(np.float32)
    ctx = api.Context(fs, xs, N, 0.5, 1, use_power_floor=True, srp_precision=api.SRP_ADAPTIVE, max_arrays=A)
    ctx.reset_timing()
    ra = ctx.process_frames_host(pcm[:, :, :(cut + 1) * hop], want_energy=True)
    rb = ctx.process_frames_host(pcm[:, :, cut * hop:], want_energy=True)
    r = {k: np.concatenate([ra[k], rb[k]], axis=2 if k == "out" else 1) for k in ("bin", "energy", "out", "voiced")}
    st = ctx.repair_stats()
    assert st["frames"] == A * F and st["flagged"] >= A, st
    fired = 0
    from parity_helpers import assert_audio_where_bins_agree
    for a in range(A):
        o = po.ssl_stream_gated(fs, N, xs, pcm[a].astype(np.float64), 1, 0.5, True)
        assert np.array_equal(r["voiced"][a].astype(bool), o["fired"].astype(bool))
        fired += int(o["fired"].sum())
        _assert_bins(r["bin"][a], o["bin"], o["energy"], ctx.P, max_ties=3)
        assert np.abs(r["energy"][a] - o["energy"]).max() <= 2e-4 * np.abs(o["energy"]).max()
        assert_audio_where_bins_agree(r["out"][a][:o["out"].shape[0]], o["out"], r["bin"][a], o["bin"], hop)
    assert 40 < fired < A * (F - lead)
    ctx.close()


def test_adaptive_with_power_gate_equals_fp16x3_at_full_size():
    from oracle import np_twin as tw
    dev = torch.device("cuda:0")
    fs, N, F, A, S = 48000, 1024, 2304, 8, 1
    xs = synth.ULA8
    pcm = torch.from_numpy(_bursty(xs, A, F, 11)).to(dev)
    res = {}
    for name, prec in (("x3", api.SRP_FP16X3), ("adaptive", api.SRP_ADAPTIVE)):
        ctx = api.Context(fs, xs, N, 0.5, S, use_power_floor=True, srp_precision=prec, max_arrays=A)
        ctx.reset_timing()
        b = torch.empty(A, F, S, dtype=torch.int32, device=dev)
        r = torch.empty(A, F, S, dtype=torch.float32, device=dev)
        p = torch.empty(A, F, S, dtype=torch.float32, device=dev)
        e = torch.empty(A, F, ctx.D, dtype=torch.float32, device=dev)
        h = 1100
        for f0, f1 in ((0, h), (h, F)):
            part = pcm[:, :, f0 * 512:(f1 + 1) * 512].contiguous()
            bb, rr, pp, ee = (torch.empty_like(t[:, f0:f1]).contiguous() for t in (b, r, p, e))
            ctx.process_frames_dev(part, f1 - f0, bb, rr, pp, ee, None)
            torch.cuda.synchronize()
            b[:, f0:f1], e[:, f0:f1] = bb, ee
        if name == "adaptive":
            st = ctx.repair_stats()
        res[name] = (b.clone(), e.clone())
        P = ctx.P
        ctx.close()
    assert st["frames"] == A * F and st["flagged"] < 0.2 * A * F
    assert int((res["x3"][0] >= 0).sum()) > 2000                       # frames fired
    diff = (res["adaptive"][0] != res["x3"][0]).any(dim=2).nonzero().tolist()
    assert len(diff) <= 4, len(diff)
    prng = np.random.default_rng(5)
    for a_, t_ in diff:                                                # (a gated-out frame repeats a tie of the frame it copies)
        E = res["x3"][1][a_, t_].double().cpu().numpy()
        base = tw.select_doa(E, P, tw.doa_step(0.5), S)[2]
        assert any(not np.array_equal(tw.select_doa(E + prng.standard_normal(E.shape) * 2e-6 * np.abs(E).max(), P, tw.doa_step(0.5), S)[2], base)
                   for _ in range(32)), (a_, t_)


def test_adaptive_fine_grid_dp512(force_small):
    """a 0.4 degree grid (451 angles, Dp = 576 > 512 columns): the second pick's LDS tile exceeds 64 KiB and needs the
    dynamic-shared-memory opt-in (k_scan_repick); bins against the oracle."""
    fs, N, F = 48000, 1024, 330
    pcm = synth.noise_source_stream(synth.ULA8, np.deg2rad(-17.3), fs, (F + 1) * N // 2, 31)
    ctx = api.Context(fs, synth.ULA8, N, 0.4, 1, srp_precision=api.SRP_ADAPTIVE)
    assert ctx.D == 451
    r = ctx.process_frames_host(pcm[None], want_energy=False)
    assert ctx.repair_stats()["frames"] == F
    o = po.ssl_stream(fs, N, synth.ULA8, pcm.astype(np.float64), 1, 0.4, want_map=True, want_audio=False)
    _assert_bins(r["bin"][0], o["bin"], o["energy"], ctx.P, max_ties=2)
    ctx.close()


def test_adaptive_fallback_off_is_an_api_field_and_runs_are_bit_reproducible(monkeypatch):
    """mca_hip_config.adaptive_fallback = OFF (round 4; before: the environment only): every eligible call runs coarse + repair
    whatever the content, and two identical runs return the same bits in EVERY output -- on noise-only input too, where AUTO
    would back off at a call that depends on when the GPU's report reaches the host."""
    monkeypatch.delenv("MCA_HIP_ADAPT_FALLBACK", raising=False)       # the field decides, not the environment
    monkeypatch.setenv("MCA_HIP_ADAPT_MIN_ROWS", "256")
    fs, N, F, A, n_calls = 48000, 1024, 256, 2, 6
    xs = synth.ULA8
    rng = np.random.default_rng(11)
    pcm = (0.1 * rng.standard_normal((A, len(xs), (n_calls * F + 1) * 512))).astype(np.float32)
    pcm[1] = synth.noise_source_stream(xs, np.deg2rad(31.0), fs, (n_calls * F + 1) * 512, 77, snr_db=10.0)
    runs = []
    for _ in range(2):
        ctx = api.Context(fs, xs, N, 0.5, 1, srp_precision=api.SRP_ADAPTIVE, max_arrays=A, adaptive_fallback=False)
        ctx.reset_timing()
        outs = [ctx.process_frames_host(pcm[:, :, i * F * 512:((i + 1) * F + 1) * 512], want_energy=True) for i in range(n_calls)]
        assert ctx.repair_stats()["frames"] == n_calls * A * F           # no call backed off
        runs.append(outs)
        ctx.close()
    for a, b in zip(*runs):
        for k in ("bin", "doa", "prob", "energy", "out"):
            assert np.array_equal(a[k], b[k]), k


def test_graph_of_a_new_shape_while_the_mode_is_suspended(monkeypatch):
    """mca_hip_graph_create while the back-off has suspended the adaptive mode reserved the FP16X3 workspace only; the
    recording -- always coarse + repair -- then allocated inside the capture and failed.  Reserve is by the shape of the call."""
    monkeypatch.setenv("MCA_HIP_ADAPT_FALLBACK", "1")
    monkeypatch.setenv("MCA_HIP_ADAPT_MIN_ROWS", "256")
    fs, N, F, A = 48000, 1024, 256, 1
    xs = synth.ULA8
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(3)
    ctx = api.Context(fs, xs, N, 0.5, 1, srp_precision=api.SRP_ADAPTIVE, max_arrays=2)
    ctx.reset_timing()
    noise = (0.1 * rng.standard_normal((A, len(xs), (F + 1) * 512))).astype(np.float32)
    ctx.process_frames_host(noise)                     # adaptive: everything flagged, the report says so
    ctx.process_frames_host(noise)                     # adaptive as well: a report is consumed two calls after its own
    ctx.process_frames_host(noise)                     # consumes the first report: suspended from here on
    assert ctx.repair_stats()["frames"] == 2 * A * F
    F2, A2 = 320, 2                                    # a shape this context has not run in adaptive mode
    pcm = torch.from_numpy(np.stack([synth.noise_source_stream(xs, np.deg2rad(20.0 + 30 * a), fs, (F2 + 1) * 512, 5 + a) for a in range(A2)])).to(dev)
    b = torch.empty(A2, F2, 1, dtype=torch.int32, device=dev); r = torch.empty(A2, F2, 1, dtype=torch.float32, device=dev)
    q = torch.empty(A2, F2, 1, dtype=torch.float32, device=dev); o = torch.empty(A2, 1, F2 * 512, dtype=torch.float32, device=dev)
    ctx.reset()
    g = ctx.graph_create(pcm, F2, b, r, q, None, o)
    g.launch()
    torch.cuda.synchronize()
    ref = api.Context(fs, xs, N, 0.5, 1, srp_precision=api.SRP_FP16X3, max_arrays=2).process_frames_host(pcm.cpu().numpy())
    assert np.mean(b.cpu().numpy() != ref["bin"]) < 0.01
    assert np.abs(o.cpu().numpy() - ref["out"]).max() <= 1e-3 * np.abs(ref["out"]).max()     # (a flipped tie moves the steering by one bin)
    g.close(); ctx.close()


@pytest.mark.parametrize("notch", [False, True], ids=["white", "notch_at_3kHz"])
def test_sixteen_microphones_repair_the_frames_whose_real_bins_are_at_rounding_level(force_small, notch):
    """PHAT keeps only the SIGN of a bin, and the DC and Nyquist bins are real: when one of them sits at the rounding level of the
    transform, two implementations need not agree on it (seed 1 / case 23 of tools/adaptive_check.py).  With 16
    microphones the coarse rows (k_stft_phat_wave16) and the exact rows (k_stft_phat<16>) come from two kernels, so the coarse
    analysis marks such frames and k_scan_pick repairs them with their six successors whatever the map says.  Here the Nyquist bin of
    one channel is cancelled in one frame; the frames around it must turn up in the repair statistics, and the bins stay FP16X3's."""
    fs, N, F, hop = 48000, 1024, 512, 512
    xs = synth.ULA16
    pcm = synth.noise_source_stream(xs, np.deg2rad(31.0), fs, (F + 1) * hop, 77).astype(np.float64)
    if notch:
        # round 4 measured "rounding level" against the frame's bin 64 (3 kHz) alone.  A notch there (2.6 ... 3.4 kHz removed from
        # every channel) made that scale ~1e-7 of the spectrum's level and the cancelled Nyquist bin below went unmarked; the scale is now the
        # channel's mean bin power (Parseval), which a notch does not move.
        Xs = np.fft.rfft(pcm, axis=1)
        fr = np.fft.rfftfreq(pcm.shape[1], 1.0 / fs)
        Xs[:, (fr > 2600.0) & (fr < 3400.0)] = 0.0
        pcm = np.fft.irfft(Xs, n=pcm.shape[1], axis=1)
    base = pcm.astype(np.float32).copy()
    t0, ch = 300, 9
    w = np.hanning(N + 1)[:N]                                             # (periodic Hann, as the library's window)
    alt = (-1.0) ** np.arange(N)
    seg = pcm[ch, t0 * hop:t0 * hop + N]
    seg -= (alt * w * seg).sum() * alt * w / (w * w).sum()                # sum (-1)^n w[n] x[n] = 0: the frame's Nyquist bin
    pcm32 = pcm.astype(np.float32)
    X = np.fft.rfft(pcm32[ch, t0 * hop:t0 * hop + N].astype(np.float64) * w)
    assert abs(X[N // 2]) < 1e-6 * np.abs(X).max()                        # (what float32 samples leave of it)
    orig = base

    def run(x, prec):
        ctx = api.Context(fs, xs, N, 0.5, 1, srp_precision=prec)
        ctx.reset_timing()
        r = ctx.process_frames_host(x[None])
        st = ctx.repair_stats() if prec == api.SRP_ADAPTIVE else None
        ctx.close()
        return r["bin"][0], st
    b_plain, st_plain = run(orig, api.SRP_ADAPTIVE)
    b_mod, st_mod = run(pcm32, api.SRP_ADAPTIVE)
    b_x3, _ = run(pcm32, api.SRP_FP16X3)
    assert st_mod["flagged"] >= st_plain["flagged"] + 5, (st_plain, st_mod)     # frames t0 ... t0 + 6, unless some were flagged anyway
    assert np.array_equal(b_mod, b_x3)


def _dev_calls(ctx, pcm, F, n_calls, hop, sizes=None):
    """pcm [A][M][...] fed to mca_hip_process_frames_dev in consecutive calls of F (or sizes[i]) frames; returns bins [A][total], audio"""
    import torch
    dev = torch.device("cuda:0")
    A = pcm.shape[0]
    bins, outs, t0 = [], [], 0
    for i in range(n_calls):
        Fi = sizes[i] if sizes else F
        x = torch.from_numpy(np.ascontiguousarray(pcm[:, :, t0 * hop:(t0 + Fi + 1) * hop])).to(dev)
        b = torch.empty(A, Fi, 1, dtype=torch.int32, device=dev)
        r = torch.empty(A, Fi, 1, dtype=torch.float32, device=dev)
        q = torch.empty(A, Fi, 1, dtype=torch.float32, device=dev)
        o = torch.empty(A, 1, Fi * hop, dtype=torch.float32, device=dev)
        ctx.process_frames_dev(x, Fi, b, r, q, None, o)
        torch.cuda.synchronize()
        bins.append(b.cpu().numpy()[:, :, 0]); outs.append(o.cpu().numpy()[:, 0])
        t0 += Fi
    return np.concatenate(bins, axis=1), np.concatenate(outs, axis=1)


def test_lazy_tails_match_the_eager_form_and_the_oracle(force_small, monkeypatch):
    """Lazy tails (round 5; mca_internal.h HIST_FRAMES): an adaptive device-pointer call no longer recomputes its last 17 rows for the
    state it hands over; it keeps 16 frames of PCM, their coarse rows and the energies in front of them, and the NEXT call repairs them
    only where one of its first 16 frames is flagged.  Over six consecutive calls of a low-SNR stream (many near ties, also right behind the
    call boundaries): the bins equal the oracle's up to oracle-fragile frames, the lazy and the eager form (MCA_HIP_ADAPT_LAZY=0) differ
    only on such frames, the lazy form recomputes fewer rows, a sequence with a small (non-adaptive) call in between and a state blob
    taken in the middle continue exactly as the eager form's do."""
    import os
    fs, N, hop, A, F, n_calls = 48000, 1024, 512, 2, 160, 6
    xs = synth.ULA8
    total = F * n_calls
    # two sources of equal strength per array (their peaks trade places from frame to frame) and a decision margin 40 x the shipped one:
    # several per cent of the frames are flagged, among them first frames of calls -- the rows of the PREVIOUS call come from the history
    monkeypatch.setenv("MCA_HIP_ADAPT_TAU_SCALE", "40")
    pcm = np.stack([(synth.noise_source_stream(xs, np.deg2rad(20.0 - 50 * a), fs, (total + 1) * hop, 4100 + a, snr_db=10.0) * 0.5 +
                     synth.noise_source_stream(xs, np.deg2rad(-35.0 + 70 * a), fs, (total + 1) * hop, 4200 + a, snr_db=10.0) * 0.5).astype(np.float32)
                    for a in range(A)])
    o = [po.ssl_stream(fs, N, xs, pcm[a].astype(np.float64), 1, 0.5, want_map=True) for a in range(A)]

    def make(lazy):
        if not lazy:
            os.environ["MCA_HIP_ADAPT_LAZY"] = "0"
        os.environ["MCA_HIP_ADAPT_CAND"] = "1"            # the lazy form with candidate columns (the context is pinned: no policy reports to go by)
        c = api.Context(fs, xs, N, 0.5, 1, srp_precision=api.SRP_ADAPTIVE, max_arrays=A, adaptive_fallback=False)
        os.environ.pop("MCA_HIP_ADAPT_LAZY", None)
        os.environ.pop("MCA_HIP_ADAPT_CAND", None)
        return c
    res = {}
    for lazy in (True, False):
        ctx = make(lazy)
        ctx.reset_timing()
        b, audio = _dev_calls(ctx, pcm, F, n_calls, hop)
        res[lazy] = (b, audio, ctx.repair_stats())
        ctx.close()
    from parity_helpers import classify_bins, assert_audio_where_bins_agree
    n_ties = 0
    for a in range(A):
        for lazy in (True, False):
            ties, bad = classify_bins(res[lazy][0][a], o[a]["bin"][:, 0], o[a]["energy"], 28)
            assert not bad, ("lazy" if lazy else "eager", a, bad[:5])
            n_ties += len(ties)
            assert_audio_where_bins_agree(res[lazy][1][a][None], o[a]["out"], res[lazy][0][a], o[a]["bin"][:, 0], hop)
    assert n_ties <= 0.02 * 2 * A * total
    st_l, st_e = res[True][2], res[False][2]
    assert st_l["flagged"] > 40, st_l                            # content flags (the lazy form has no others), ~17 % of the frames:
    assert st_l["recomputed"] > 0.5 * A * total                   # ... rows behind call boundaries among them, i.e. history units
    # mixed sequence: lazy, lazy, a call too small for the adaptive mode (settles the debt), lazy, state blob, lazy
    sizes = [160, 160, 16, 160, 160, 304]
    assert sum(sizes) == total
    outs = {}
    for lazy in (True, False):
        ctx = make(lazy)
        b1, a1 = _dev_calls(ctx, pcm, 0, 4, hop, sizes[:4])
        blob = ctx.state_save()
        ctx.close()
        ctx2 = make(lazy)
        ctx2.state_load(blob)
        t0 = sum(sizes[:4])
        b2, a2 = _dev_calls(ctx2, pcm[:, :, t0 * hop:], 0, 2, hop, sizes[4:])
        ctx2.close()
        outs[lazy] = np.concatenate([b1, b2], axis=1)
    for a in range(A):
        for lazy in (True, False):
            ties, bad = classify_bins(outs[lazy][a], o[a]["bin"][:, 0], o[a]["energy"], 28)
            assert not bad, ("mixed", "lazy" if lazy else "eager", a, bad[:5])
    # with the shipped decision margin the same stream flags a handful of frames: the lazy form recomputes their rows, the eager form
    # also the last 20 rows of every array and call
    monkeypatch.delenv("MCA_HIP_ADAPT_TAU_SCALE")
    counts = {}
    for lazy in (True, False):
        ctx = make(lazy)
        ctx.reset_timing()
        _dev_calls(ctx, pcm, F, n_calls, hop)
        counts[lazy] = ctx.repair_stats()
        ctx.close()
    assert counts[False]["flagged"] == counts[True]["flagged"] + A * n_calls, counts
    assert counts[True]["recomputed"] + 12 * A * n_calls <= counts[False]["recomputed"], counts


def test_settling_the_history_in_passes_when_the_workspace_budget_is_small(force_small, monkeypatch):
    """settle_history ran ONE pass over the history rows of every array whatever the workspace held.  With a 4 MB budget
    (MCA_HIP_WS_MAX_MB: 128 two-plane rows per repair pass) and 16 arrays the history is 256 rows: two passes.  A lazy call, a call too
    small for the mode (settles the debt), a lazy call, a state blob (settles again), the stream continued from the blob: bins
    against the oracle throughout."""
    monkeypatch.setenv("MCA_HIP_WS_MAX_MB", "4")
    monkeypatch.setenv("MCA_HIP_ADAPT_TAU_SCALE", "20")
    fs, N, hop, A = 48000, 1024, 512, 16
    xs = synth.ULA8
    sizes = [64, 16, 64, 64]
    total = sum(sizes)
    pcm = np.stack([synth.noise_source_stream(xs, np.deg2rad(-70.0 + 9 * a), fs, (total + 1) * hop, 9100 + a, snr_db=8.0).astype(np.float32) for a in range(A)])
    ctx = api.Context(fs, xs, N, 0.5, 1, srp_precision=api.SRP_ADAPTIVE, max_arrays=A, adaptive_fallback=False)
    ctx.reset_timing()
    b1, a1 = _dev_calls(ctx, pcm, 0, 3, hop, sizes[:3])
    blob = ctx.state_save()
    st = ctx.repair_stats()
    ctx.close()
    assert st["frames"] == 2 * A * 64, st                       # (the 16-frame call ran as FP16X3)
    ctx2 = api.Context(fs, xs, N, 0.5, 1, srp_precision=api.SRP_ADAPTIVE, max_arrays=A, adaptive_fallback=False)
    ctx2.state_load(blob)
    t0 = sum(sizes[:3])
    b2, a2 = _dev_calls(ctx2, pcm[:, :, t0 * hop:], 0, 1, hop, sizes[3:])
    ctx2.close()
    bins, audio = np.concatenate([b1, b2], axis=1), np.concatenate([a1, a2], axis=1)
    from parity_helpers import classify_bins, assert_audio_where_bins_agree
    n_ties = 0
    for a in range(A):
        o = po.ssl_stream(fs, N, xs, pcm[a].astype(np.float64), 1, 0.5, want_map=True)
        ties, bad = classify_bins(bins[a], o["bin"][:, 0], o["energy"], 28)
        assert not bad, (a, bad[:5])
        n_ties += len(ties)
        assert_audio_where_bins_agree(audio[a][None], o["out"], bins[a], o["bin"][:, 0], hop)
    assert n_ties <= 0.02 * A * total, n_ties


def test_candidate_columns_and_frames_of_exact_zeros(force_small, monkeypatch):
    """Candidate columns (round 5; CandArgs in mca_internal.h): lazy calls recompute a flagged frame's rows at the delays its pick can be
    among.  A low-SNR stream with a stretch of digital silence in the middle (every channel exact zeros: those rows are zero in the coarse
    and in the exact map, they are counted as warm rows and not listed) and one channel muted for a while, in five device-pointer calls:
    bins against the oracle up to oracle-fragile frames (the silent stretch has flat maps: ties by construction), audio on every hop whose
    bins agree; mca_hip_get_repair_columns: a handful of columns per flagged frame, no frame that took every column outside the silence,
    nothing with MCA_HIP_ADAPT_CAND=0 (whole-row kernels) -- whose bins differ from the candidate form's on oracle-fragile frames only."""
    import os
    fs, N, hop, A, F, n_calls = 48000, 1024, 512, 2, 128, 5
    xs = synth.ULA8
    total = F * n_calls
    monkeypatch.setenv("MCA_HIP_ADAPT_TAU_SCALE", "20")
    pcm = np.stack([synth.noise_source_stream(xs, np.deg2rad(35.0 - 60 * a), fs, (total + 1) * hop, 5100 + a, snr_db=6.0).astype(np.float32) for a in range(A)])
    pcm[:, :, 250 * hop:330 * hop] = 0.0                       # 80 frames of digital silence, across a call boundary
    pcm[0, 3, 400 * hop:470 * hop] = 0.0                       # one muted channel
    o = [po.ssl_stream(fs, N, xs, pcm[a].astype(np.float64), 1, 0.5, want_map=True) for a in range(A)]
    from parity_helpers import classify_bins, assert_audio_where_bins_agree
    res = {}
    for cand in ("1", "0"):
        os.environ["MCA_HIP_ADAPT_CAND"] = cand
        ctx = api.Context(fs, xs, N, 0.5, 1, srp_precision=api.SRP_ADAPTIVE, max_arrays=A, adaptive_fallback=False)
        os.environ.pop("MCA_HIP_ADAPT_CAND", None)
        ctx.reset_timing()
        b, audio = _dev_calls(ctx, pcm, F, n_calls, hop)
        res[cand] = (b, audio, ctx.repair_stats(), ctx.repair_columns())
        assert ctx.D == 361
        ctx.close()
        for a in range(A):
            ties, bad = classify_bins(b[a], o[a]["bin"][:, 0], o[a]["energy"], 28)
            assert not bad, (cand, a, bad[:5])
            assert_audio_where_bins_agree(audio[a][None], o[a]["out"], b[a], o[a]["bin"][:, 0], hop)
    st, cols = res["1"][2], res["1"][3]
    assert st["flagged"] > 30, st
    assert cols["whole_row_frames"] <= 2 * A * 80, cols          # (only frames of the silent stretch -- flat maps -- can lack a lower bound; their rows are zero: not listed)
    narrow = st["flagged"] - cols["whole_row_frames"]
    assert narrow > 0 and 0 < cols["candidate_columns"] - 361 * cols["whole_row_frames"] <= 40 * narrow, (st, cols)
    assert res["0"][3] == {"candidate_columns": 0, "whole_row_frames": 0}, res["0"][3]
    assert res["0"][2]["flagged"] == st["flagged"]               # the same frames are flagged: the coarse pass is the same
    assert res["1"][2]["recomputed"] <= res["0"][2]["recomputed"]  # ... and the silent rows are not listed
