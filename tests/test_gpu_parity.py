"""GPU parity tests proper: the HIP path, called through the C ABI (libmcarray_hip.so), against
the CPU oracle and the committed golden vectors.  Run on the MI355X box with `-m gpu`.

Tolerances (fp32 GPU vs fp64 oracle), written where they are used:
  * DOA bin: bit-exact, except frames the oracle itself flags as fragile (mca_or_select_doa_fragile_local at 1e-6 x max(1, |the
    normalised energies compared|): peak ties, sign-chain ties, zero picks -- tests/parity_helpers.py), which are counted and bounded.
  * energy map E_t[d]: |gpu - oracle| <= TOL_E * max|E| with TOL_E = 2e-5 (fp32), 2e-5 (fp16x3), 2e-4 (fp16)
  * beamformed audio: |gpu - oracle| <= 2e-5 * max|out| + 1e-7
"""
import numpy as np
import pytest

from mcarray_amd import api, synth
from oracle import pyoracle as po

pytestmark = pytest.mark.gpu

TOL_E = {api.SRP_FP32: 2e-5, api.SRP_FP16X3: 2e-5, api.SRP_FP16: 2e-4}
PRECS = [api.SRP_FP32, api.SRP_FP16X3, api.SRP_FP16]


def _golden(golden_dir, name):
    import os
    return np.load(os.path.join(golden_dir, name + ".npz"))


from parity_helpers import assert_bins as _assert_bins   # exact, or a frame the oracle itself flags as fragile (counted)


@pytest.mark.parametrize("prec", PRECS)
@pytest.mark.parametrize("name", ["ssl_reemc_d37", "ssl_ula8_d361", "ssl_reemc_d37_s2"])
def test_stream_matches_golden(golden_dir, name, prec):
    g = _golden(golden_dir, name)
    S = int(g["n_sources"])
    ctx = api.Context(int(g["fs"]), g["xs"], int(g["N"]), float(g["step_deg"]), S, srp_precision=prec)
    r = ctx.process_frames_host(g["pcm"][None], want_energy=True)
    P = ctx.P
    _assert_bins(r["bin"][0], g["bin"], g["energy"], P)
    scale = np.abs(g["energy"]).max()
    assert np.abs(r["energy"][0] - g["energy"]).max() <= TOL_E[prec] * scale
    np.testing.assert_allclose(r["doa"][0], g["doa"].astype(np.float32), rtol=0, atol=0)
    nout = g["out"].shape[0]
    assert np.abs(r["out"][0, :nout] - g["out"]).max() <= 2e-5 * np.abs(g["out"]).max() + 1e-7
    ctx.close()


@pytest.mark.parametrize("prec", PRECS)
def test_stream_vs_oracle_random_ula8_d361(prec):
    fs, N, F, A = 48000, 1024, 40, 3
    xs = synth.ULA8
    rng = np.random.default_rng(7)
    thetas = rng.uniform(-80, 80, size=A)
    pcm = np.stack([synth.noise_source_stream(xs, np.deg2rad(thetas[a]), fs, (F + 1) * N // 2, 100 + a) for a in range(A)])
    ctx = api.Context(fs, xs, N, 0.5, 1, srp_precision=prec, max_arrays=A)
    assert ctx.D == 361 and ctx.P == 28 and ctx.G == 7
    r = ctx.process_frames_host(pcm, want_energy=True)
    for a in range(A):
        o = po.ssl_stream(fs, N, xs, pcm[a].astype(np.float64), 1, 0.5, want_map=True)
        _assert_bins(r["bin"][a], o["bin"], o["energy"], ctx.P, max_ties=2)
        scale = np.abs(o["energy"]).max()
        assert np.abs(r["energy"][a] - o["energy"]).max() <= TOL_E[prec] * scale
        assert np.abs(r["out"][a] - o["out"]).max() <= 2e-5 * np.abs(o["out"]).max() + 1e-7
        # the DOA must also be the true one (0.5 degree grid, +-1 bin)
        deg = np.rad2deg(r["doa"][a, 5:, 0])
        assert np.all(np.abs(deg - thetas[a]) <= 0.76), (thetas[a], deg)
    ctx.close()


@pytest.mark.parametrize("step,D,S", [(0.4, 451, 2), (1.0, 181, 3), (3.0, 61, 1)])
def test_other_angle_grids_vs_oracle(step, D, S):
    """the peak pick deals 2 / 6 / 8 consecutive positions to a lane by the number of angles (451 angles: 8 per lane, a map padded
    to 576 columns); several sources"""
    fs, N, F, A = 48000, 1024, 36, 2
    xs = synth.ULA8
    pcm = np.stack([synth.noise_source_stream(xs, np.deg2rad(th), fs, (F + 1) * N // 2, 300 + a) for a, th in enumerate((-41.0, 63.0))])
    ctx = api.Context(fs, xs, N, step, S, max_arrays=A)
    assert ctx.D == D
    r = ctx.process_frames_host(pcm, want_energy=True)
    for a in range(A):
        o = po.ssl_stream(fs, N, xs, pcm[a].astype(np.float64), S, step, want_map=True)
        _assert_bins(r["bin"][a], o["bin"], o["energy"], ctx.P, max_ties=3)
        assert np.abs(r["energy"][a] - o["energy"]).max() <= 2e-4 * np.abs(o["energy"]).max()
        assert np.abs(r["out"][a] - o["out"]).max() <= 2e-5 * np.abs(o["out"]).max() + 1e-7
    ctx.close()


@pytest.mark.parametrize("prec", [api.SRP_FP16X3, api.SRP_FP16])
def test_split_k_contraction_vs_oracle(prec):
    # >= 1024 rows with 361 angles selects the 256 x 384 split-K MFMA kernel (two partial maps summed by the scan)
    fs, N, F, A = 48000, 1024, 136, 8
    xs = synth.ULA8
    pcm = np.stack([synth.noise_source_stream(xs, np.deg2rad(-75.0 + 20.0 * a), fs, (F + 1) * N // 2, 500 + a) for a in range(A)])
    ctx = api.Context(fs, xs, N, 0.5, 1, srp_precision=prec, max_arrays=A)
    r = ctx.process_frames_host(pcm, want_energy=True)
    for a in (0, 5, 7):
        o = po.ssl_stream(fs, N, xs, pcm[a].astype(np.float64), 1, 0.5, want_map=True, want_audio=False)
        _assert_bins(r["bin"][a], o["bin"], o["energy"], ctx.P, max_ties=2)
        assert np.abs(r["energy"][a] - o["energy"]).max() <= TOL_E[prec] * np.abs(o["energy"]).max()
    ctx.close()


def test_nonuniform_array_no_merging():
    # Reem-C has 6 distinct pair distances: G == P, the un-merged kernel variant
    fs, N, F = 48000, 1024, 20
    xs = synth.REEM_C
    pcm = synth.noise_source_stream(xs, np.deg2rad(-35.0), fs, (F + 1) * N // 2, 5)
    ctx = api.Context(fs, xs, N, 5.0, 2)
    assert ctx.G == ctx.P == 6
    r = ctx.process_frames_host(pcm[None], want_energy=True)
    o = po.ssl_stream(fs, N, xs, pcm.astype(np.float64), 2, 5.0, want_map=True)
    _assert_bins(r["bin"][0], o["bin"], o["energy"], 6, max_ties=1)
    assert np.abs(r["energy"][0] - o["energy"]).max() <= 2e-5 * np.abs(o["energy"]).max()
    np.testing.assert_allclose(r["prob"][0], o["prob"], rtol=0, atol=2e-5)
    ctx.close()


@pytest.mark.parametrize("M", [3, 5, 16])
def test_generic_channel_counts(M):
    # runtime-M kernel variants (M = 3: ULA generic, 5: irregular generic, 16: ULA template)
    fs, N, F = 48000, 1024, 6
    rng = np.random.default_rng(M)
    xs = [0.02 * m for m in range(M)] if M != 5 else list(np.sort(rng.uniform(0, 0.3, 5)))
    pcm = synth.noise_source_stream(xs, np.deg2rad(25.0), fs, (F + 1) * N // 2, 9 + M)
    ctx = api.Context(fs, xs, N, 5.0, 1)
    r = ctx.process_frames_host(pcm[None], want_energy=True)
    o = po.ssl_stream(fs, N, xs, pcm.astype(np.float64), 1, 5.0, want_map=True)
    _assert_bins(r["bin"][0], o["bin"], o["energy"], ctx.P, max_ties=1)
    assert np.abs(r["energy"][0] - o["energy"]).max() <= 2e-5 * np.abs(o["energy"]).max()
    assert np.abs(r["out"][0] - o["out"]).max() <= 2e-5 * np.abs(o["out"]).max() + 1e-7
    ctx.close()


def test_frame_api_double_precision_matches_oracle():
    # mca::SteeringBeamforming::processFrame / mca::Beamformer::processFrame drop-ins run in double
    fs, N, F = 48000, 1024, 6
    xs = synth.REEM_C
    pcm = synth.noise_source_stream(xs, np.deg2rad(40.0), fs, (F + 1) * N // 2, 77).astype(np.float64)
    X = po.stft_frames(pcm, N)
    sb = api.SteeringBeamforming(fs, xs, N + 2, 4)
    bf = api.Beamformer(fs, xs, N + 2, 4)
    ost = po.Steering(fs, xs, N + 2, 5.0)
    for t in range(F):
        doa, prob, bins = sb.process_frame(X[t], 2)
        o = ost.process_frame(X[t], 2)
        assert np.array_equal(bins, o["bin"])
        np.testing.assert_allclose(doa, o["doa"], rtol=0, atol=0)
        np.testing.assert_allclose(prob, o["prob"], rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(sb.ctx.energy(), o["energy"], rtol=0, atol=1e-9)
        y = bf.process_frame(X[t], doa[0])
        yo = po.beamformer_process_frame(fs, xs, X[t], doa[0])
        np.testing.assert_allclose(y, yo, rtol=0, atol=1e-10 * np.abs(yo).max())
    # FFTLogPower of the power gate
    import ctypes as C
    rows = (po.c_dp * 4)(*[X[0, c].ctypes.data_as(po.c_dp) for c in range(4)])
    assert sb.ctx.fft_log_power(X[0]) == pytest.approx(po.lib().mca_or_fft_log_power(rows, 4, N + 2), abs=1e-9)


def test_frame_api_other_fft_size():
    # the dead reference test uses N = 2048 (test/test_mcarray.cpp:660-662): frame API takes any size
    fs, N = 48000, 2048
    xs = synth.REEM_C
    x = (synth.sine_stream(xs, np.deg2rad(45), fs, N, 800.0, 5000.0) + synth.sine_stream(xs, np.deg2rad(-45), fs, N, 2000.0, 5000.0))
    frames = np.stack([po.rfft_ccs(x[c]) for c in range(4)])
    bf = api.Beamformer(fs, xs, N + 2, 4)
    y = bf.process_frame(frames, np.deg2rad(45))
    yo = po.beamformer_process_frame(fs, xs, frames, np.deg2rad(45))
    np.testing.assert_allclose(y, yo, rtol=0, atol=1e-10 * np.abs(yo).max())


@pytest.mark.parametrize("prec", [api.SRP_FP32, api.SRP_ADAPTIVE])
def test_chunk_start_values_by_look_back_equal_the_carry_pass(monkeypatch, prec):
    """Ungated calls let every pick workgroup compose the start value of its 32-frame chunk from the chunk-local results of the four
    chunks before it (0.8^32 per chunk: the fifth back is 3e-16 of it) instead of running k_scan_carry over all chunks in order
    (MCA_HIP_SCAN_CARRY=1): same bins, energies equal to the last bits, over two calls, with a loud source that stops (the case in
    which a dropped term would show first)."""
    fs, N, F, A = 48000, 1024, 1280, 3
    xs = synth.ULA8
    monkeypatch.setenv("MCA_HIP_ADAPT_MIN_ROWS", "256")     # (read by mca_hip_create: the adaptive case runs coarse + repair at this size)
    monkeypatch.setenv("MCA_HIP_ADAPT_FALLBACK", "0")
    pcm = np.stack([synth.noise_source_stream(xs, np.deg2rad(-55.0 + 50 * a), fs, (F + 1) * N // 2, 300 + a, snr_db=15.0) for a in range(A)])
    pcm[:, :, (F // 2) * 512:] *= 1e-3                       # the energy falls by 60 dB in the middle of the stream
    res = {}
    for carry in (False, True):
        if carry:
            monkeypatch.setenv("MCA_HIP_SCAN_CARRY", "1")
        else:
            monkeypatch.delenv("MCA_HIP_SCAN_CARRY", raising=False)
        ctx = api.Context(fs, xs, N, 0.5, 1, srp_precision=prec, max_arrays=A)
        cut = 700
        ra = ctx.process_frames_host(pcm[:, :, :(cut + 1) * 512], want_energy=True)
        rb = ctx.process_frames_host(pcm[:, :, cut * 512:], want_energy=True)
        res[carry] = {k: np.concatenate([ra[k], rb[k]], axis=1) for k in ("bin", "energy")}
        ctx.close()
    assert np.array_equal(res[False]["bin"], res[True]["bin"])
    scale = np.abs(res[True]["energy"]).max(axis=2, keepdims=True) + 1e-30
    assert (np.abs(res[False]["energy"] - res[True]["energy"]) / scale).max() <= 4e-7


def test_state_carries_across_calls():
    # E_prev (SteeringBeamforming.h:69) and the overlap-add tail continue across process() calls
    fs, N, F = 48000, 1024, 48
    xs = synth.ULA8
    pcm = synth.noise_source_stream(xs, np.deg2rad(12.0), fs, (F + 1) * N // 2, 3)
    hop = N // 2
    one = api.Context(fs, xs, N, 0.5, 1)
    r1 = one.process_frames_host(pcm[None], want_energy=True)
    two = api.Context(fs, xs, N, 0.5, 1)
    h = F // 2
    ra = two.process_frames_host(pcm[None, :, :(h + 1) * hop], want_energy=True)
    rb = two.process_frames_host(pcm[None, :, h * hop:], want_energy=True)
    assert np.array_equal(np.concatenate([ra["bin"], rb["bin"]], axis=1), r1["bin"])
    # (the two runs cut the recursion into chunks at different frames: a few ulp of fp32 rounding between them)
    np.testing.assert_allclose(np.concatenate([ra["energy"], rb["energy"]], axis=1), r1["energy"], rtol=2e-6, atol=1e-3)
    np.testing.assert_allclose(np.concatenate([ra["out"], rb["out"]], axis=2), r1["out"], rtol=0, atol=1e-6)
    # reset gives a fresh module
    two.reset()
    rc = two.process_frames_host(pcm[None], want_energy=True)
    assert np.array_equal(rc["bin"], r1["bin"])
    np.testing.assert_array_equal(rc["out"], r1["out"])


def test_int16_pcm_equals_float_pcm():
    """mca_hip_process_frames_host_i16: 16-bit PCM (the reference's process(std::vector<int16_t*>&, ...) callers) is widened on
    the GPU and must give exactly what the same samples give as floats; the DOA does not depend on the scale of the input."""
    fs, N, F = 48000, 1024, 40
    xs = synth.ULA8
    x = synth.noise_source_stream(xs, np.deg2rad(-28.0), fs, (F + 1) * N // 2, 17)
    p16 = np.round(x * 20000.0).astype(np.int16)
    a = api.Context(fs, xs, N, 0.5, 1).process_frames_host(p16[None], want_energy=True)
    b = api.Context(fs, xs, N, 0.5, 1).process_frames_host(p16[None].astype(np.float32), want_energy=True)
    for k in ("bin", "doa", "prob", "energy", "out"):
        assert np.array_equal(a[k], b[k]), k
    c = api.Context(fs, xs, N, 0.5, 1).process_frames_host(x[None])
    assert (a["bin"] != c["bin"]).mean() <= 0.05          # 16-bit quantisation of the input may move a near-tie
    assert np.abs(a["out"] / 20000.0 - c["out"]).max() <= 1e-3 * np.abs(c["out"]).max() + 2e-4


def test_edge_cases_silence_single_frame_ragged():
    fs, N = 48000, 1024
    xs = synth.ULA8
    ctx = api.Context(fs, xs, N, 0.5, 1, max_arrays=2)
    z = np.zeros((2, 8, 4 * 512), dtype=np.float32)
    r = ctx.process_frames_host(z, want_energy=True)
    assert np.all(r["bin"] == 1) and np.all(r["prob"] == 0) and np.all(r["energy"] == 0) and np.all(r["out"] == 0)
    # one channel silent: PHAT of a zero bin contributes 0 (oracle: |G| = 0 -> 0)
    pcm = synth.noise_source_stream(xs, 0.3, fs, 2 * 512, 1)
    pcm[3] = 0
    ctx.reset()
    r = ctx.process_frames_host(pcm[None], want_energy=True)      # F = 1
    o = po.ssl_stream(fs, N, xs, pcm.astype(np.float64), 1, 0.5, want_map=True)
    assert np.array_equal(r["bin"][0], o["bin"])
    assert np.abs(r["energy"][0] - o["energy"]).max() <= 2e-5 * np.abs(o["energy"]).max()
    # ragged frame counts around the kernel tile sizes (8 frames/block, 16-frame runs, 128-frame scan chunks)
    for F in (7, 9, 17, 129, 131):
        pcm = synth.noise_source_stream(xs, -0.5, fs, (F + 1) * 512, F)
        ctx.reset()
        r = ctx.process_frames_host(pcm[None], want_energy=True)
        o = po.ssl_stream(fs, N, xs, pcm.astype(np.float64), 1, 0.5, want_map=True, want_audio=(F < 20))
        _assert_bins(r["bin"][0], o["bin"], o["energy"], 28, max_ties=2)
        assert np.abs(r["energy"][0] - o["energy"]).max() <= 2e-5 * np.abs(o["energy"]).max()
    ctx.close()


def test_invalid_arguments_are_rejected():
    with pytest.raises(api.MCArrayHipError):
        api.Context(48000, [0.0], 1024)                       # one microphone
    with pytest.raises(api.MCArrayHipError):
        api.Context(48000, synth.ULA8, 1024, doa_step_deg=0.1)   # > 512 steering angles
    ctx = api.Context(48000, synth.ULA8, 1536, 5.0)
    with pytest.raises(api.MCArrayHipError):                   # stream API needs a power-of-two frame length
        ctx.process_frames_host(np.zeros((1, 8, 3 * 768), dtype=np.float32))
    ctx = api.Context(48000, synth.ULA8, 1024, 5.0, max_arrays=1)
    with pytest.raises(api.MCArrayHipError):
        ctx.process_frames_host(np.zeros((2, 8, 2048), dtype=np.float32))   # more arrays than max_arrays


@pytest.mark.parametrize("prec", [api.SRP_FP32, api.SRP_FP16X3, api.SRP_FP16, api.SRP_ADAPTIVE])
def test_full_size_properties(prec):
    """BASELINE config 3 size (8 arrays x 4096 frames, 361 angles) through size-independent properties."""
    torch = pytest.importorskip("torch")
    fs, N, F, A = 48000, 1024, 4096, 8
    hop = N // 2
    dev = torch.device("cuda:0")
    gen = torch.Generator(device=dev).manual_seed(1234)
    L = (F + 1) * hop
    xs = torch.tensor(synth.ULA8, device=dev, dtype=torch.float64)
    theta = torch.linspace(-70, 70, A, device=dev, dtype=torch.float64) * np.pi / 180
    s = torch.randn(A, L, device=dev, dtype=torch.float64, generator=gen) * 0.1
    Sf = torch.fft.rfft(s, dim=1)
    f = torch.fft.rfftfreq(L, d=1.0 / fs).to(dev).to(torch.float64)
    adv = xs[None, :, None] * torch.sin(theta)[:, None, None] / 346.1
    x = torch.fft.irfft(Sf[:, None, :] * torch.exp(2j * np.pi * f[None, None, :] * adv), n=L, dim=2)
    x = x + torch.randn(A, 8, L, device=dev, dtype=torch.float64, generator=gen) * 0.01
    pcm = x.to(torch.float32).contiguous()

    def run(ctx, p, nA, energy=None):
        b = torch.empty(nA, F, 1, dtype=torch.int32, device=dev)
        d = torch.empty(nA, F, 1, dtype=torch.float32, device=dev)
        pr = torch.empty(nA, F, 1, dtype=torch.float32, device=dev)
        o = torch.empty(nA, 1, F * hop, dtype=torch.float32, device=dev)
        ctx.process_frames_dev(p, F, b, d, pr, energy, o, stream=None)
        torch.cuda.synchronize()
        return b, d, pr, o

    ctx = api.Context(fs, synth.ULA8, N, 0.5, 1, srp_precision=prec, max_arrays=A)
    b, d, pr, o = run(ctx, pcm, A)
    # (1) every array localises its own source within one grid step once the IIR has settled
    deg = d[:, 16:, 0].cpu().numpy() * 180 / np.pi
    assert np.all(np.abs(deg - np.linspace(-70, 70, A)[:, None]) <= 0.76)
    # (2) gain invariance: PHAT makes the DOA independent of level; power-of-two gain scales audio exactly
    ctx.reset()
    b2, d2, pr2, o2 = run(ctx, (pcm * 0.25).contiguous(), A)
    assert torch.equal(b, b2)
    assert torch.allclose(o2, o * 0.25, rtol=0, atol=1e-7)
    # (3) arrays are independent units: array 5 alone == array 5 in the batch (bit-exact)
    solo = api.Context(fs, synth.ULA8, N, 0.5, 1, srp_precision=prec, max_arrays=1)
    e3 = torch.empty(1, F, ctx.D, dtype=torch.float32, device=dev) if prec == api.SRP_ADAPTIVE else None
    b3, d3, pr3, o3 = run(solo, pcm[5:6].contiguous(), 1, e3)
    if prec == api.SRP_ADAPTIVE:
        # The two calls run their coarse pass on different kernels (4 096 rows: 128 x 192 tiles, K cut 8 ways; 32 768 rows: 256 x 384
        # tiles, 2 ways), so they flag different frames, and a repaired frame keeps 0.8^17 of its call's coarse error: a frame whose
        # pick is an exact-level tie may resolve differently.  Every differing frame must BE such a tie -- its own energy row (either
        # call's) fragile at the parity bar, 1e-6 of the row's peak, tests/parity_helpers.py -- and the audio is equal wherever the
        # bins are.
        diff = (b3[0] != b[5]).flatten()
        if int(diff.sum()):
            from parity_helpers import fragile
            ctx.reset()
            eb = torch.empty(A, F, ctx.D, dtype=torch.float32, device=dev)
            b_again = run(ctx, pcm, A, eb)[0]
            assert torch.equal(b_again, b)                      # (the same call on the same state: the same bins)
            for t in torch.nonzero(diff).flatten().tolist():
                assert fragile(e3[0, t].double().cpu().numpy(), ctx.P, 1) or fragile(eb[5, t].double().cpu().numpy(), ctx.P, 1), \
                    "array 5 alone and in the batch differ at frame %d (%d vs %d) and neither energy row is a tie at the parity bar" % (t, int(b3[0, t, 0]), int(b[5, t, 0]))
        assert int(diff.sum()) <= 4
        same_hops = (~diff).repeat_interleave(hop)
        same_hops[hop:] &= same_hops[:-hop].clone()            # a hop also carries the previous frame's second half
        assert torch.equal(o3[0, 0][same_hops], o[5, 0][same_hops])
    else:
        assert torch.equal(b3[0], b[5]) and torch.equal(o3[0], o[5])
    # (4) STFT -> delay-and-sum -> ISTFT is the identity for identical channels steered broadside
    same = pcm[0:1, 0:1, :].expand(1, 8, L).contiguous()
    d0 = torch.zeros(1, F, 1, dtype=torch.float32, device=dev)
    oid = torch.empty(1, 1, F * hop, dtype=torch.float32, device=dev)
    solo.reset()
    solo.process_frames_dev(same, F, None, d0, None, None, oid, localise=False)
    torch.cuda.synchronize()
    assert torch.allclose(oid[0, 0, hop:], same[0, 0, hop:F * hop], rtol=0, atol=2e-6)
    # (5) checksum: the energy recursion is linear -- sum_d E_t = 0.8 sum_d E_{t-1} + 0.2 sum_d C_t is implied by
    #     the map; check prob (= normalised energy at the peak) stays inside its analytic bounds
    assert float(pr.max()) <= (28 * 513 + 15 * 28) / (30 * 28) + 1e-3 and float(pr.min()) >= 0.0


# ---------------------------------------------------------------------------------------------
# 2-microphone GCC-PHAT path (FreqGCCBinauralLocalisation) and binaural masking
# ---------------------------------------------------------------------------------------------
def test_freqgcc_matches_golden_and_oracle(golden_dir):
    g = _golden(golden_dir, "freqgcc_16k_d61")
    fs, N = int(g["fs"]), int(g["N"])
    loc = api.FreqGCCBinauralLocalisation(fs, g["xs"], False, float(g["step_deg"]))
    assert loc.ctx.D == 61
    r = loc.process(g["pcm"], want_corr=True)
    assert np.array_equal(r["argmax"][0], g["argmax"])
    assert np.abs(r["corr"][0] - g["corr"]).max() <= 2e-5 * np.abs(g["corr"]).max()
    # longer stream vs the oracle, crossing the 32-frame chunks of the scan kernel
    F = 150
    pcm = synth.noise_source_stream(synth.BINAURAL, np.deg2rad(-42.0), fs, (F + 1) * N // 2, 8)
    loc = api.FreqGCCBinauralLocalisation(fs, synth.BINAURAL, False, 3.0)
    r = loc.process(pcm, want_corr=True)
    og = po.FreqGCC(fs, synth.BINAURAL, N + 2, False, 3.0)
    X = po.stft_frames(pcm.astype(np.float64), N)
    prev_doa = 0.0
    nbad = 0
    for t in range(F):
        voiced, corr, idx, doa, power = og.process(X[t, 0], X[t, 1])
        assert voiced
        if idx != r["argmax"][0, t]:
            assert abs(corr[idx] - corr[r["argmax"][0, t]]) < 1e-5 * np.abs(corr).max()   # numerical tie
            nbad += 1
        assert np.abs(r["corr"][0, t] - corr).max() <= 2e-5 * np.abs(corr).max()
        assert abs(r["doa"][0, t] - doa) <= 2e-5 + 0.06 * nbad
        pr = og.set_probability(np.array([prev_doa]))[0]           # setProbability of the previous DOA (:454)
        assert abs(r["prob"][0, t] - pr) <= 2e-4
        prev_doa = doa
    assert nbad <= 1
    # state carries over calls
    loc2 = api.FreqGCCBinauralLocalisation(fs, synth.BINAURAL, False, 3.0)
    h = 70
    ra = loc2.process(pcm[:, :(h + 1) * 512], want_corr=True)
    rb = loc2.process(pcm[:, h * 512:], want_corr=True)
    assert np.array_equal(np.concatenate([ra["argmax"], rb["argmax"]], axis=1), r["argmax"])
    np.testing.assert_allclose(np.concatenate([ra["doa"], rb["doa"]], axis=1), r["doa"], atol=1e-5)


def test_freqgcc_power_gate_matches_oracle():
    """usePowerFloor = true (the reference's default, BinauralLocalisation.h:191): 3 s of floor estimation, then only frames
    6 dB above the floor fire (BinauralLocalisation.cpp:387-404, :425-434); the recursions skip the others, whose outputs
    repeat the last fired frame.  Two calls, a stream with loud bursts and quiet gaps, bursts that straddle the chunks."""
    fs, N, F = 16000, 1024, 330
    hop = N // 2
    xs = synth.BINAURAL
    A = 2
    pcm = np.stack([synth.noise_source_stream(xs, np.deg2rad(-42.0 + 70 * a), fs, (F + 1) * hop, 8 + a) for a in range(A)])
    env = np.full(F + 1, 0.01)
    for (b0, b1) in [(70, 95), (120, 121), (140, 215), (260, 300)]:       # loud bursts (in hops)
        env[b0:b1] = 1.0
    pcm = (pcm * np.repeat(env, hop)[None, None, :]).astype(np.float32)
    pcm[1] *= 0.5
    loc = api.FreqGCCBinauralLocalisation(fs, xs, True, 3.0, max_arrays=A)
    h = 150
    ra = loc.process(pcm[:, :, :(h + 1) * hop], want_corr=True)
    rb = loc.process(pcm[:, :, h * hop:], want_corr=True)
    r = {k: np.concatenate([ra[k], rb[k]], axis=1) for k in ("argmax", "doa", "prob", "corr", "voiced", "power")}
    fired_total = 0
    for a in range(A):
        og = po.FreqGCC(fs, xs, N + 2, True, 3.0)
        X = po.stft_frames(pcm[a].astype(np.float64), N)
        prev_doa, last = 0.0, None
        nbad = 0
        for t in range(F):
            voiced, corr, idx, doa, power = og.process(X[t, 0], X[t, 1])
            assert bool(r["voiced"][a, t]) == voiced, (a, t)
            assert abs(r["power"][a, t] - power) <= 1e-4 * abs(power) + 1e-6, (a, t)
            if voiced:
                fired_total += 1
                if idx != r["argmax"][a, t]:
                    assert abs(corr[idx] - corr[r["argmax"][a, t]]) < 1e-5 * np.abs(corr).max()   # numerical tie
                    nbad += 1
                assert np.abs(r["corr"][a, t] - corr).max() <= 2e-5 * np.abs(corr).max(), (a, t)
                assert abs(r["doa"][a, t] - doa) <= 2e-5 + 0.06 * nbad, (a, t)
                pr = og.set_probability(np.array([prev_doa]))[0]
                assert abs(r["prob"][a, t] - pr) <= 2e-4, (a, t)
                prev_doa = doa
                last = t
            elif last is None:
                assert r["argmax"][a, t] == -1 and r["doa"][a, t] == 0 and r["prob"][a, t] == -1     # :339-340
                assert np.all(r["corr"][a, t] == 0)
            else:
                for k in ("argmax", "doa", "prob"):
                    assert r[k][a, t] == r[k][a, last], (a, t, k)
                assert np.array_equal(r["corr"][a, t], r["corr"][a, last])
        assert nbad <= 2
    assert 150 < fired_total <= 2 * 148                        # the bursts after the 47 frames of floor estimation, both arrays


def _check_gated_golden(r, g, F):
    fired = g["fired"][:F]
    assert np.array_equal(r["voiced"][0].astype(bool), fired)
    k = 0
    last = None
    for t in range(F):
        if fired[t]:
            c = g["corr_fired"][k]
            assert r["argmax"][0, t] == g["argmax"][t], t
            assert np.abs(r["corr"][0, t] - c).max() <= 2e-5 * np.abs(c).max(), t
            assert abs(r["doa"][0, t] - g["doa"][t]) <= 2e-5, (t, r["doa"][0, t], g["doa"][t])
            assert abs(r["prob"][0, t] - g["prob"][t]) <= 2e-4, t
            k += 1
            last = t
        elif last is None:
            assert r["argmax"][0, t] == -1 and r["doa"][0, t] == 0 and r["prob"][0, t] == -1
        else:
            assert r["argmax"][0, t] == r["argmax"][0, last] and r["doa"][0, t] == r["doa"][0, last]


def test_freqgcc_silence_rule_matches_golden(golden_dir):
    """The silence rule of BinauralLocalisation.cpp:530-560 (3 s of gated-out frames zero both memory factors): the golden
    stream fires after 94 gated-out frames (restart: corr = R_t, DOA = raw angle) and after 93 (no restart); the frame that
    completes the floor estimation already sets the factors to their maxima, so the first fired frame is smoothed against
    the zero state.  One call; the silences cut by call boundaries (the counter is carried); a checkpoint inside a silence."""
    g = _golden(golden_dir, "freqgcc_8k_gated_silence")
    fs, N = int(g["fs"]), int(g["N"])
    hop = N // 2
    pcm = g["pcm_i16"].astype(np.float32) / 32768
    F = pcm.shape[1] // hop - 1
    assert list(np.nonzero(g["restart"])[0]) == [150]
    loc = api.FreqGCCBinauralLocalisation(fs, g["xs"], True, float(g["step_deg"]))
    assert loc.ctx.N == 512 and loc.ctx.D == 61
    r = loc.process(pcm, want_corr=True)
    _check_gated_golden(r, g, F)
    # restart frame: raw grid angle; frame 250 (93 gated-out frames before it): pulled 0.6 : 0.4
    grid = loc.ctx.doa_grid()
    assert r["doa"][0, 150] == grid[r["argmax"][0, 150]]
    assert abs(r["doa"][0, 250] - (0.6 * r["doa"][0, 249] + 0.4 * grid[r["argmax"][0, 250]])) < 1e-5
    # the same stream in calls that cut both silences, the estimation phase and a burst
    for cuts in ([100, 200], [30, 47, 53, 149, 150, 151, 230], [48, 49, 156, 157]):
        loc.ctx.reset()
        parts = []
        b = [0] + cuts + [F]
        for i in range(len(b) - 1):
            parts.append(loc.process(pcm[:, b[i] * hop:(b[i + 1] + 1) * hop], want_corr=True))
        rc = {k: np.concatenate([q[k] for q in parts], axis=1) for k in ("argmax", "doa", "prob", "corr", "voiced")}
        _check_gated_golden(rc, g, F)
        assert np.array_equal(rc["argmax"], r["argmax"])
        np.testing.assert_allclose(rc["doa"], r["doa"], atol=1e-6)
    # checkpoint in the middle of the long silence, resumed in a fresh context
    loc.ctx.reset()
    ra = loc.process(pcm[:, :(120 + 1) * hop], want_corr=True)
    blob = loc.ctx.state_save()
    loc2 = api.FreqGCCBinauralLocalisation(fs, g["xs"], True, float(g["step_deg"]))
    loc2.ctx.state_load(blob)
    rb = loc2.process(pcm[:, 120 * hop:], want_corr=True)
    rc = {k: np.concatenate([ra[k], rb[k]], axis=1) for k in ("argmax", "doa", "prob", "corr", "voiced")}
    _check_gated_golden(rc, g, F)


def test_freqgcc_silence_rule_vs_oracle_long_stream():
    """1024-sample frames at 16 kHz (windowsToDecay = 93), two arrays with different burst patterns in one batch, silences
    of 92 ... 200 frames, chunked scan (> 128 fired frames), against the C oracle frame by frame."""
    fs, N, F = 16000, 1024, 900
    hop = N // 2
    xs = synth.BINAURAL
    # a burst over hops [b0, b1) fires frames b0 - 1 ... b1 - 1, so the next burst follows b0' - b1 - 1 gated-out frames
    patterns = [[(50, 60), (155, 160), (254, 420), (621, 640), (734, 740)],       # gaps: 94 (restart), 93 (none), 200 (restart), 93
                [(60, 200), (294, 300), (501, 510), (605, 700)]]                   # gaps: 93 (none), 200 (restart), 94 (restart)
    pcm = []
    for a, bursts in enumerate(patterns):
        x = synth.noise_source_stream(xs, np.deg2rad(-40.0 + 60 * a), fs, (F + 1) * hop, 90 + a)
        y = synth.noise_source_stream(xs, np.deg2rad(25.0 - 70 * a), fs, (F + 1) * hop, 95 + a)
        env = np.full(F + 1, 0.005)
        sel = np.zeros(F + 1, bool)
        for i, (b0, b1) in enumerate(bursts):
            env[b0:b1] = 1.0
            if i % 2:
                sel[b0:b1] = True
        e, s_ = np.repeat(env, hop), np.repeat(sel, hop)
        pcm.append(np.where(s_[None, :], y, x) * e[None, :])
    pcm = np.stack(pcm).astype(np.float32)
    loc = api.FreqGCCBinauralLocalisation(fs, xs, True, 3.0, max_arrays=2)
    h = 450
    ra = loc.process(pcm[:, :, :(h + 1) * hop], want_corr=True)
    rb = loc.process(pcm[:, :, h * hop:], want_corr=True)
    r = {k: np.concatenate([ra[k], rb[k]], axis=1) for k in ("argmax", "doa", "prob", "corr", "voiced")}
    restarts = []
    for a in range(2):
        og = po.FreqGCC(fs, xs, N + 2, True, 3.0)
        X = po.stft_frames(pcm[a].astype(np.float64), N)
        prev_doa, run, nbad = 0.0, 0, 0
        grid = loc.ctx.doa_grid()
        for t in range(F):
            voiced, corr, idx, doa, power = og.process(X[t, 0], X[t, 1])
            assert bool(r["voiced"][a, t]) == voiced, (a, t)
            if voiced:
                if run >= 94:
                    restarts.append((a, t))
                    assert abs(doa - grid[idx]) < 1e-6                       # the oracle restarted ...
                    assert r["doa"][a, t] == grid[r["argmax"][a, t]]         # ... and so did the GPU
                if idx != r["argmax"][a, t]:
                    assert abs(corr[idx] - corr[r["argmax"][a, t]]) < 1e-5 * np.abs(corr).max()   # numerical tie
                    nbad += 1
                assert np.abs(r["corr"][a, t] - corr).max() <= 2e-5 * np.abs(corr).max(), (a, t)
                assert abs(r["doa"][a, t] - doa) <= 2e-5 + 0.06 * nbad, (a, t)
                assert abs(r["prob"][a, t] - og.set_probability(np.array([prev_doa]))[0]) <= 2e-4, (a, t)
                prev_doa, run = doa, 0
            else:
                run += 1
        assert nbad <= 2
    assert len(restarts) == 4, restarts


def _mask_margins(m, X, t_count):
    """oracle decisions plus a flag per (frame, band): decision sits within 1e-4 (relative) of a threshold"""
    decs, near = [], []
    for t in range(t_count):
        Q_before = m.short_time_power
        _, _, dec = m.process(X[t, 0], X[t, 1])
        decs.append(dec)
    return np.array(decs)


@pytest.mark.parametrize("name", ["mask_relative_both", "mask_full_both", "mask_factor_temporal", "mask_noisy_spatial"])
def test_masking_stream_matches_golden(golden_dir, name):
    g = _golden(golden_dir, name)
    fs, N = int(g["fs"]), int(g["N"])
    m = api.FastBinauralMasking(fs, float(g["d"]), float(g["flo"]), float(g["fhi"]), int(g["method"]), int(g["alg"]))
    thr, cen = m.thresholds()
    np.testing.assert_allclose(thr, g["thresholds"], atol=1e-14)
    np.testing.assert_allclose(cen, g["center"], atol=1e-15)
    out, dec = m.process(np.stack([g["left"], g["right"]]))
    ndiff = int((dec[0] != g["decisions"]).sum())
    assert ndiff == 0, "decisions differ in %d (frame, band) cells" % ndiff
    assert np.abs(out[0] - g["out"]).max() <= 2e-5 * np.abs(g["out"]).max() + 1e-7


def test_masking_hook_double_matches_oracle(golden_dir):
    g = _golden(golden_dir, "mask_relative_both")
    fs, N = int(g["fs"]), int(g["N"])
    X = po.stft_frames(np.stack([g["left"], g["right"]]).astype(np.float64), N)
    for method, alg in ((api.RELATIVE, api.BOTH), (api.NOISY, api.SPATIAL), (api.FACTOR, api.TEMPORAL), (api.FULL, api.BOTH)):
        m = api.FastBinauralMasking(fs, float(g["d"]), float(g["flo"]), float(g["fhi"]), method, alg)
        o = po.Masking(fs, N, float(g["d"]), float(g["flo"]), float(g["fhi"]), method, alg)
        for t in range(X.shape[0]):
            l, r, dec = m.process_parametrisation(X[t, 0], X[t, 1])
            ol, orr, odec = o.process(X[t, 0], X[t, 1])
            assert np.array_equal(dec, odec), (method, alg, t)
            np.testing.assert_allclose(l, ol, rtol=0, atol=1e-10 * np.abs(ol).max())
            np.testing.assert_allclose(r, orr, rtol=0, atol=1e-10 * np.abs(orr).max())
        m.close()


def test_masking_long_stream_chunks_and_reference_windows():
    """The reference's own masking tests (test/test_mcarray.cpp:892-1065) through the HIP path, long enough to
    cross the 64-frame chunks of the kernel: band-power windows 70+-10 dB, 2+-0.5 dB -> 5+-1 dB."""
    from scipy import signal
    fs, N = 16000, 1024
    magn = 5000
    delay = int(0.1 * fs)
    tonestep = int(0.1 * fs)
    n = 50 * 1024
    tone = np.zeros(n)
    i, sfreq, interest_start = 0, np.float32(0.01), 0
    while i < n - tonestep and sfreq < 0.5:
        if np.float32(0.2) - np.float32(0.005) < sfreq < np.float32(0.2) + np.float32(0.005):
            interest_start = i
        tone[i:i + tonestep] = synth.tone16(tonestep, magn, float(sfreq))
        i += tonestep
        sfreq = np.float32(sfreq + np.float32(0.01))
    sig = np.zeros(n)
    tb = tone.copy()
    for k in range(6):
        tb = np.trunc(tb / 2)
        sig[delay * k:] += tb[:n - delay * k]
    F = n // 512 - 1
    pcm = np.stack([sig, sig])[:, :(F + 1) * 512].astype(np.float32)
    m = api.FastBinauralMasking(fs, 0.086, 500, 5000, api.FULL, api.BOTH)
    out, dec = m.process(pcm)
    o = po.Masking(fs, N, 0.086, 500, 5000, po.FULL, po.BOTH)
    ol, orr = o.stream(pcm[0].astype(np.float64), pcm[1].astype(np.float64))
    # same decisions as the oracle except threshold ties (identical channels make ncorr == 1 exactly: no spatial ties)
    X = po.stft_frames(pcm.astype(np.float64), N)
    o2 = po.Masking(fs, N, 0.086, 500, 5000, po.FULL, po.BOTH)
    odec = np.array([o2.process(X[t, 0], X[t, 1])[2] for t in range(F)])
    assert (dec[0] != odec).mean() < 2e-3
    agree = (dec[0] == odec).all(axis=1)
    # frames whose decisions all agree must match the oracle's audio (two frames overlap per output hop)
    ok = agree[1:] & agree[:-1]
    hop = 512
    err = np.abs(out[0, 0] - ol).reshape(F, hop).max(axis=1)[1:]
    assert err[ok].max() <= 2e-5 * np.abs(ol).max() + 1e-3
    sigf = signal.firwin(257, [0.19, 0.21], pass_zero=False, fs=1.0)
    intf = signal.firwin(257, [0.15, 0.20], pass_zero=False, fs=1.0)
    sl = slice(interest_start, interest_start + tonestep)
    diff = lambda x: po.log_power(signal.lfilter(sigf, 1.0, x)[sl]) - po.log_power(signal.lfilter(intf, 1.0, x)[sl])
    assert abs(diff(sig[:F * hop]) - 2) < 0.5
    assert abs(diff(out[0, 0].astype(np.float64)) - 5) < 1.0


# ---------------------------------------------------------------------------------------------
# power gate (usePowerFloor = true, the reference's default) in the batched stream path
# ---------------------------------------------------------------------------------------------
def _gated_signal(xs, fs, F, seed):
    rng = np.random.default_rng(seed)
    L = (F + 1) * 512
    pcm = rng.standard_normal((len(xs), L)) * 0.001                 # sensor-noise floor for the first 3 s and the gaps
    src = synth.noise_source_stream(xs, np.deg2rad(30.0), fs, L, seed + 1).astype(np.float64)
    env = np.zeros(L)
    for a, b in ((150, 165), (172, 180), (190, F - 1)):
        env[a * 512:b * 512] = 1.0
    return (pcm + src * env).astype(np.float32)


@pytest.mark.parametrize("prec", [api.SRP_FP32, api.SRP_FP16X3])
def test_power_gate_stream_matches_oracle(prec):
    fs, N, F = 48000, 1024, 230
    xs = synth.REEM_C
    pcm = _gated_signal(xs, fs, F, 3)
    o = po.ssl_stream_gated(fs, N, xs, pcm.astype(np.float64), 1, 5.0, True)
    assert 0 < o["fired"].sum() < F and o["fired"][:141].sum() == 0          # 3 s of floor estimation never fire
    ctx = api.Context(fs, xs, N, 5.0, 1, use_power_floor=True, srp_precision=prec)
    r = ctx.process_frames_host(pcm[None], want_energy=True)
    assert np.array_equal(r["voiced"][0], o["fired"])
    np.testing.assert_allclose(r["power"][0][141:], o["power"][141:], rtol=0, atol=2e-3)        # dB
    np.testing.assert_allclose(r["power"][0][:140], o["power"][:140], rtol=2e-5)                # running sum (linear)
    assert np.array_equal(r["bin"][0], o["bin"])
    np.testing.assert_allclose(r["doa"][0], o["doa"].astype(np.float32), rtol=0, atol=0)
    np.testing.assert_allclose(r["prob"][0], o["prob"], rtol=0, atol=2e-5)
    assert np.abs(r["energy"][0] - o["energy"]).max() <= 2e-5 * np.abs(o["energy"]).max()
    assert np.abs(r["out"][0] - o["out"]).max() <= 2e-5 * np.abs(o["out"]).max() + 1e-7
    # chunked streaming: the floor estimation, E_prev, the last DOA and the OLA tail all carry across calls
    two = api.Context(fs, xs, N, 5.0, 1, use_power_floor=True, srp_precision=prec)
    cut = 100
    ra = two.process_frames_host(pcm[None, :, :(cut + 1) * 512], want_energy=True)
    rb = two.process_frames_host(pcm[None, :, cut * 512:], want_energy=True)
    for k in ("voiced", "bin", "doa"):
        assert np.array_equal(np.concatenate([ra[k], rb[k]], axis=1), r[k]), k
    np.testing.assert_allclose(np.concatenate([ra["out"], rb["out"]], axis=2), r["out"], rtol=0, atol=1e-6)
    # a third call that starts inside a silent gap keeps reporting the previous DOA
    rc = two.process_frames_host((pcm[None, :, :6 * 512] * 0).astype(np.float32))
    assert np.all(rc["voiced"] == 0) and np.all(rc["bin"] == r["bin"][0, -1]) and np.all(rc["doa"] == r["doa"][0, -1])
    two.reset()
    rd = two.process_frames_host(pcm[None])
    assert np.array_equal(rd["bin"], r["bin"])


def test_exact_chunked_scan_long_stream():
    # 700 frames cross several 128-frame scan chunks: the chunk composition E_start[c+1] = 0.8^n E_start[c] + b
    # must reproduce the frame-by-frame recursion of the oracle
    fs, N, F = 48000, 1024, 700
    xs = synth.REEM_C
    pcm = synth.noise_source_stream(xs, np.deg2rad(-20.0), fs, (F + 1) * 512, 77)
    ctx = api.Context(fs, xs, N, 5.0, 2)
    r = ctx.process_frames_host(pcm[None], want_energy=True, want_audio=False)
    o = po.ssl_stream(fs, N, xs, pcm.astype(np.float64), 2, 5.0, want_map=True, want_audio=False)
    _assert_bins(r["bin"][0], o["bin"], o["energy"], 6, max_ties=3)
    assert np.abs(r["energy"][0] - o["energy"]).max() <= 2e-5 * np.abs(o["energy"]).max()


def test_workspace_chunking_gives_identical_results():
    """The A-operand workspace is bounded (4 GiB by default): long batches are processed in chunks of frames.
    Forcing a tiny budget must not change a single output bit (run in a subprocess: the budget is read once)."""
    import subprocess, sys, textwrap
    code = textwrap.dedent('''
        import numpy as np, sys, hashlib
        from mcarray_amd import api, synth
        fs, N, F, A = 48000, 1024, 300, 3
        pcm = np.stack([synth.noise_source_stream(synth.ULA8, np.deg2rad(-50 + 45 * a), fs, (F + 1) * 512, 900 + a) for a in range(A)])
        ctx = api.Context(fs, synth.ULA8, N, 0.5, 1, srp_precision=api.SRP_FP16X3, max_arrays=A)
        r = ctx.process_frames_host(pcm, want_energy=True)
        h = hashlib.sha256()
        for k in ("bin", "doa", "prob", "energy", "out"):
            h.update(np.ascontiguousarray(r[k]).tobytes())
        print(h.hexdigest())
    ''')
    import os
    env = dict(os.environ)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env["PYTHONPATH"] = root + os.pathsep + env.get("PYTHONPATH", "")
    a = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=600)
    env["MCA_HIP_WS_MAX_MB"] = "8"          # 8 MiB / 57.6 KB per row -> ~145 rows -> 48-frame chunks for 3 arrays
    b = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=600)
    assert a.returncode == 0 and b.returncode == 0, a.stderr + b.stderr
    assert a.stdout.strip().splitlines()[-1] == b.stdout.strip().splitlines()[-1]


@pytest.mark.parametrize("fs,N", [(48000, 1024), (16000, 512)])
def test_mcbeam_cli_wav_roundtrip(tmp_path, fs, N):
    """tools/mcbeam.py (the counterpart of src/programs/mcabeamf.cpp): 4-channel 16-bit WAV in, mono WAV + DOA text out;
    the frame length follows the sample rate like the reference's module (SourceSeparationAndLocalisation.cpp:52)."""
    import subprocess, sys, wave, os
    F, hop = 40, N // 2
    xs = synth.REEM_C
    x = synth.noise_source_stream(xs, np.deg2rad(25.0), fs, (F + 1) * hop, 12)
    pcm16 = np.clip(np.round(x * 32768), -32768, 32767).astype("<i2")
    wav_in, wav_out, doa_txt = str(tmp_path / "in.wav"), str(tmp_path / "out.wav"), str(tmp_path / "doa.txt")
    with wave.open(wav_in, "wb") as w:
        w.setnchannels(4); w.setsampwidth(2); w.setframerate(fs); w.writeframes(pcm16.T.tobytes())
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "mcbeam.py"), "-i", wav_in, "-o", wav_out, "-d", doa_txt,
                        "--mics", "0,0.07,0.175,0.21"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    doa = np.loadtxt(doa_txt)
    assert doa.shape == (F, 2) and np.all(np.abs(doa[5:, 0] - 25.0) <= 5.0)
    with wave.open(wav_out, "rb") as w:
        assert w.getnchannels() == 1 and w.getframerate() == fs and w.getnframes() == F * hop
        y = np.frombuffer(w.readframes(F * hop), dtype="<i2").astype(np.float64) / 32768
    o = po.ssl_stream(fs, N, xs, pcm16.astype(np.float64) / 32768, 1, 5.0)
    assert np.abs(y - o["out"][0]).max() <= 1.5 / 32768 + 2e-5


def test_source_localisation_is_the_analysis_only_sibling():
    # mca::SourceLocalisation (SourceLocalisation.cpp:63-79): processFrameLocalisation only -- same DOAs, no audio
    fs, F = 48000, 25
    xs = synth.REEM_C
    pcm = synth.noise_source_stream(xs, np.deg2rad(-15.0), fs, (F + 1) * 512, 19)
    got = []
    sl = api.SourceLocalisation(fs, xs, 2, False)
    sl.set_callback(lambda doa, prob, power, n: got.append((doa.copy(), n)))
    r = sl.process(pcm)
    assert r["out"] is None and len(got) == F and got[0][1] == 2
    o = po.ssl_stream(fs, 1024, xs, pcm.astype(np.float64), 2, 5.0, want_map=True, want_audio=False)
    _assert_bins(r["bin"][0], o["bin"], o["energy"], 6, max_ties=1)
    np.testing.assert_allclose(np.array([g[0] for g in got]), np.rad2deg(r["doa"][0].astype(np.float64)), rtol=0, atol=0)


def test_state_save_and_load_resume_a_stream():
    # checkpoint after the first half of a gated 2-source stream, resume in a NEW context: identical to one run
    fs, N, F = 48000, 1024, 220
    xs = synth.REEM_C
    pcm = _gated_signal(xs, fs, F, 9)
    one = api.Context(fs, xs, N, 5.0, 2, use_power_floor=True)
    r = one.process_frames_host(pcm[None], want_energy=True)
    a = api.Context(fs, xs, N, 5.0, 2, use_power_floor=True)
    cut = 160                                        # inside the first burst, after the floor estimation
    ra = a.process_frames_host(pcm[None, :, :(cut + 1) * 512], want_energy=True)
    blob = a.state_save()
    a.close()
    b = api.Context(fs, xs, N, 5.0, 2, use_power_floor=True)
    b.state_load(blob)
    rb = b.process_frames_host(pcm[None, :, cut * 512:], want_energy=True)
    for k in ("voiced", "bin", "doa", "prob"):
        assert np.array_equal(np.concatenate([ra[k], rb[k]], axis=1), r[k]), k
    np.testing.assert_allclose(np.concatenate([ra["energy"], rb["energy"]], axis=1), r["energy"], rtol=0, atol=1e-6 * np.abs(r["energy"]).max())
    np.testing.assert_allclose(np.concatenate([ra["out"], rb["out"]], axis=2), r["out"], rtol=0, atol=1e-6)
    # a blob from another configuration is refused
    other = api.Context(fs, xs, N, 3.0, 2, use_power_floor=True)
    with pytest.raises(api.MCArrayHipError, match="different configuration"):
        other.state_load(blob)
    with pytest.raises(api.MCArrayHipError):
        b.state_load(blob[:100])


def test_page_locked_host_buffers_equal_pageable_ones():
    """The host-pointer entry point with page-locked buffers (mca_hip_host_alloc: arrays uploaded in chunks, upload / kernels /
    download overlapped on three streams) returns bit-for-bit what the synchronous pageable path returns: fp32 and 16-bit
    PCM, several arrays (chunked), one array, with the power gate (single chunk) and with state carried over two calls."""
    fs, N = 48000, 1024
    hop = N // 2
    xs = synth.ULA8
    for A, F, gate, prec in ((5, 70, False, api.SRP_FP16X3), (1, 40, False, api.SRP_FP32), (3, 200, True, api.SRP_FP16X3), (6, 1400, False, api.SRP_ADAPTIVE)):
        pcm = np.stack([synth.noise_source_stream(xs, np.deg2rad(-50.0 + 30 * a), fs, (F + 1) * hop, 300 + a) for a in range(A)])
        if gate:
            pcm[:, :, :150 * hop] *= 0.01
        i16 = np.round(pcm * 20000).astype(np.int16)
        for src in (pcm, i16):
            pin = api.PinnedBuffer(src.shape, src.dtype)
            pin.array[...] = src
            outs = []
            for buf, pinned in ((src, False), (pin.array, True)):
                ctx = api.Context(fs, xs, N, 0.5, 1, use_power_floor=gate, srp_precision=prec, max_arrays=A)
                into = None
                keep = []
                h = F // 2
                parts = []
                for sl in (slice(0, (h + 1) * hop), slice(h * hop, None)):
                    part = np.ascontiguousarray(buf[:, :, sl])
                    if pinned:
                        pp = api.PinnedBuffer(part.shape, part.dtype)
                        pp.array[...] = part
                        nf = part.shape[2] // hop - 1
                        ob = {"bin": api.PinnedBuffer((A, nf, 1), np.int32), "out": api.PinnedBuffer((A, 1, nf * hop), np.float32)}
                        keep += [pp] + list(ob.values())
                        r = ctx.process_frames_host(pp.array, want_energy=True, into={k: v.array for k, v in ob.items()})
                    else:
                        r = ctx.process_frames_host(part, want_energy=True)
                    parts.append({k: np.array(v) for k, v in r.items() if v is not None})
                outs.append({k: np.concatenate([q[k] for q in parts], axis=2 if k == "out" else 1) for k in parts[0]})
                ctx.close()
            for k in outs[0]:
                assert np.array_equal(outs[0][k], outs[1][k]), (A, F, gate, prec, src.dtype, k)
            pin.close()


def test_masking_context_rejects_a_changing_stream_count():
    """the masking context tracks the module's first-frame behaviour once for all its streams (FastBinauralMasking.cpp:186-197):
    a later call with another n_streams would silently give some slots the wrong path, so it is refused until a reset"""
    fs, N, F = 16000, 1024, 12
    rng = np.random.default_rng(3)
    pcm = (rng.standard_normal((3, 2, (F + 1) * 512)) * 0.1).astype(np.float32)
    m = api.FastBinauralMasking(fs, 0.086, 500.0, 5000.0, po.FULL, po.BOTH, max_streams=3)
    m.process(pcm[:2])
    with pytest.raises(api.MCArrayHipError, match="n_streams differs"):
        m.process(pcm)
    m.process(pcm[:2])
    m.reset()
    m.process(pcm)


def test_kernel_timing_and_mask():
    """mca_hip_set_timing / mca_hip_set_timing_mask / mca_hip_get_timing: event pairs around every kernel group, or around the
    chosen ones only (what bench.py does inside its timed region); the outputs do not depend on it."""
    fs, N, F = 48000, 1024, 64
    xs = synth.ULA8
    pcm = synth.noise_source_stream(xs, np.deg2rad(20.0), fs, (F + 1) * N // 2, 5)
    ctx = api.Context(fs, xs, N, 0.5, 1)
    r0 = ctx.process_frames_host(pcm[None])
    ctx.reset()
    ctx.set_timing(True)
    ctx.reset_timing()
    r1 = ctx.process_frames_host(pcm[None])
    n_stft, ms_stft = ctx.get_timing(api.K_STFT_PHAT)
    n_bf, ms_bf = ctx.get_timing(api.K_BEAMFORM)
    assert n_stft >= 1 and n_bf >= 1 and ms_stft > 0 and ms_bf > 0
    ctx.reset()
    ctx.set_timing_kernels([api.K_BEAMFORM])
    ctx.reset_timing()
    r2 = ctx.process_frames_host(pcm[None])
    assert ctx.get_timing(api.K_STFT_PHAT)[0] == 0 and ctx.get_timing(api.K_SRP_GEMM)[0] == 0
    assert ctx.get_timing(api.K_BEAMFORM)[0] == n_bf
    ctx.set_timing(False)
    for r in (r1, r2):
        assert np.array_equal(r["bin"], r0["bin"])
        np.testing.assert_array_equal(r["out"], r0["out"])



def test_configs4_per_gpu_shape_adaptive_vs_oracle():
    """BASELINE configs[4]'s per-GPU shape -- 128 arrays x 256 frames -- in the shipped default (ADAPTIVE), against the oracle on
    sampled arrays (first, middle, last: 3 x 256 frames), and array independence at this shape."""
    torch = pytest.importorskip("torch")
    fs, N, F, A = 48000, 1024, 256, 128
    hop = N // 2
    rng = np.random.default_rng(44)
    thetas = rng.uniform(-80, 80, A)
    sample = (0, 63, 127)
    pcm = np.empty((A, 8, (F + 1) * hop), dtype=np.float32)
    for a in range(A):
        pcm[a] = synth.noise_source_stream(synth.ULA8, np.deg2rad(thetas[a]), fs, (F + 1) * hop, 7000 + a)
    ctx = api.Context(fs, synth.ULA8, N, 0.5, 1, srp_precision=api.SRP_ADAPTIVE, max_arrays=A)
    r = ctx.process_frames_host(pcm, want_energy=False)
    st = ctx.repair_stats()
    assert st["frames"] == A * F and st["recomputed"] < 0.25 * A * F, st           # the adaptive path ran; the recomputed rows stay a fraction
    for a in sample:
        o = po.ssl_stream(fs, N, synth.ULA8, pcm[a].astype(np.float64), 1, 0.5, want_map=True)
        _assert_bins(r["bin"][a], o["bin"], o["energy"], ctx.P, max_ties=2)
        ok = (r["bin"][a] == o["bin"]).all()
        if ok:
            assert np.abs(r["out"][a] - o["out"]).max() <= 2e-5 * np.abs(o["out"]).max() + 1e-7
        assert np.all(np.abs(np.rad2deg(r["doa"][a, 16:, 0]) - thetas[a]) <= 0.76)
    ctx.close()


def test_gcc_weighting_none_localises_the_references_literal_sines():
    """mca_hip_config.gcc_weighting = NONE (plain cross-spectrum instead of PHAT; MCA_HIP_SRP_FP32): the reference's literal SRP
    stimulus -- 1 kHz sines on the 4-microphone Reem-C array, test_mcarray.cpp:384-423 -- localises within its 7 degrees at every
    angle on the GPU, bins equal to the oracle's under the same weighting; under PHAT (the default) it does not
    (tests/test_oracle_reference_properties.py keeps that as a strict xfail).  DESIGN.md section 2."""
    fs, N, F = 48000, 1024, 12
    ctx = api.Context(fs, synth.REEM_C, N, 5.0, 1, srp_precision=api.SRP_FP32, gcc_weighting=api.GCC_NONE, max_arrays=17)
    angles = list(range(-80, 81, 10))
    pcm = np.stack([synth.sine_stream(synth.REEM_C, np.deg2rad(a), fs, (F + 1) * N // 2, 1000.0, 5000.0) for a in angles]).astype(np.float32)
    r = ctx.process_frames_host(pcm, want_energy=True)
    for i, a in enumerate(angles):
        o = po.ssl_stream(fs, N, synth.REEM_C, pcm[i].astype(np.float64), 1, 5.0, want_map=True, weighting="none")
        _assert_bins(r["bin"][i], o["bin"], o["energy"], ctx.P, max_ties=0)
        assert np.abs(r["energy"][i] - o["energy"]).max() <= 2e-5 * np.abs(o["energy"]).max()
        assert np.all(np.abs(np.rad2deg(r["doa"][i, :, 0]) - a) <= 7.0)
        assert np.abs(r["out"][i] - o["out"]).max() <= 2e-5 * np.abs(o["out"]).max() + 1e-7
    ctx.close()
    # a broadband source under NONE, 8 microphones, the 0.5 degree grid
    pcm = synth.noise_source_stream(synth.ULA8, np.deg2rad(33.0), fs, 41 * 512, 5)
    c8 = api.Context(fs, synth.ULA8, N, 0.5, 1, srp_precision=api.SRP_FP32, gcc_weighting=api.GCC_NONE)
    r = c8.process_frames_host(pcm[None], want_energy=True)
    o = po.ssl_stream(fs, N, synth.ULA8, pcm.astype(np.float64), 1, 0.5, want_map=True, weighting="none")
    _assert_bins(r["bin"][0], o["bin"], o["energy"], c8.P, max_ties=1)
    assert np.abs(r["energy"][0] - o["energy"]).max() <= 2e-5 * np.abs(o["energy"]).max()
    c8.close()
    # fp16 operands cannot carry un-normalised spectra: refused, not silently whitened
    with pytest.raises(api.MCArrayHipError):
        api.Context(fs, synth.ULA8, N, 0.5, 1, srp_precision=api.SRP_ADAPTIVE, gcc_weighting=api.GCC_NONE)


def test_gcc_weighting_none_long_stream_loud_then_digital_silence():
    """under gcc_weighting NONE the map scales with the amplitude squared and has no bound, so the four-chunk look-back of
    the ungated scan (which drops 0.8^128 of an earlier energy: below the last bit only of a BOUNDED map) is not taken: after a loud
    passage at the reference's own 16-bit scale (test_mcarray.cpp:397: amplitude 5000) that falls to exact zeros, the recursion
    E = 0.8f E + ... must keep carrying the old peak, frame by frame, like the reference's (SteeringBeamforming.cpp:132-144).  Checked
    against the oracle with a PER-FRAME scale (the global one is blind to a decayed tail), over two calls."""
    fs, N, F, loud = 48000, 1024, 420, 90
    xs = synth.ULA8
    pcm = (5000.0 * synth.noise_source_stream(xs, np.deg2rad(-24.0), fs, (F + 1) * N // 2, 12, snr_db=25.0)).astype(np.float32)
    pcm[:, (loud + 1) * 512:] = 0.0
    ctx = api.Context(fs, xs, N, 5.0, 1, srp_precision=api.SRP_FP32, gcc_weighting=api.GCC_NONE)
    cut = 200
    ra = ctx.process_frames_host(pcm[None, :, :(cut + 1) * 512], want_energy=True)
    rb = ctx.process_frames_host(pcm[None, :, cut * 512:], want_energy=True)
    en = np.concatenate([ra["energy"], rb["energy"]], axis=1)[0]
    o = po.ssl_stream(fs, N, xs, pcm.astype(np.float64), 1, 5.0, want_map=True, weighting="none")
    peak = np.abs(o["energy"]).max(axis=1)
    assert peak[loud - 1] > 1e12                              # 16-bit scale: nothing like PHAT's |C| <= P
    live = peak > 1e-20                                       # (above fp32's denormals: ~330 silent frames)
    assert live.sum() > loud + 250
    rel = np.abs(en - o["energy"]).max(axis=1)[live] / peak[live]
    assert rel.max() <= 5e-5, (int(np.argmax(rel)), float(rel.max()))
    _assert_bins(np.concatenate([ra["bin"], rb["bin"]], axis=1)[0][:loud], o["bin"][:loud], o["energy"][:loud], ctx.P, max_ties=1)
    ctx.close()


@pytest.mark.parametrize("M,S,step", [(8, 3, 1.0), (5, 4, 3.0), (7, 2, 0.5), (3, 2, 5.0), (2, 3, 3.0)])
def test_several_sources_share_the_forward_transforms(M, S, step):
    """k_beamform_wave_ms (round 4): the S beamformed outputs of a frame (processFrameSeparation,
    BeamformingSeparationAndLocalisation.cpp:113-118; Beamformer.cpp:51-71) come from ONE set of forward transforms of the channel pairs,
    each source with its own steering rows, inverse transform and overlap-add carry.  Against the oracle: bins (the parity bar), every
    separated channel's audio, over two calls (the carries of all sources continue) and with an odd channel count (a half-empty pair)."""
    fs, N, F, cut = 48000, 1024, 150, 67
    rng = np.random.default_rng(100 * M + S)
    xs = np.sort(rng.uniform(0, 0.05 * M, M)).tolist() if M != 8 else synth.ULA8
    th = [-50.0, 15.0, 60.0, -10.0][:S]
    pcm = sum(synth.noise_source_stream(xs, np.deg2rad(a), fs, (F + 1) * N // 2, 300 + i, snr_db=30.0) * (1.0 - 0.2 * i) for i, a in enumerate(th)).astype(np.float32)
    ctx = api.Context(fs, xs, N, step, S, srp_precision=api.SRP_FP32)
    ra = ctx.process_frames_host(pcm[None, :, :(cut + 1) * 512], want_energy=True)
    rb = ctx.process_frames_host(pcm[None, :, cut * 512:], want_energy=True)
    r = {k: np.concatenate([ra[k], rb[k]], axis=2 if k == "out" else 1) for k in ("bin", "out")}
    o = po.ssl_stream(fs, N, xs, pcm.astype(np.float64), S, step, want_map=True)
    _assert_bins(r["bin"][0], o["bin"], o["energy"], ctx.P, max_ties=3)
    nout = o["out"].shape[0]                                    # min(M, S) separated channels
    assert nout == min(M, S)
    # every hop of every source whose bins agree (all of them unless a tie was counted above): never skipped as a whole
    from parity_helpers import assert_audio_where_bins_agree
    compared = assert_audio_where_bins_agree(r["out"][0][:nout], o["out"], r["bin"][0], o["bin"], 512)
    assert compared >= nout * (F - 8)
    ctx.close()


@pytest.mark.parametrize("gate", [False, True])
def test_sixteen_microphone_wave_analysis_matches_oracle(gate):
    """k_stft_phat_wave16 (round 4): a 16-microphone uniform linear array with ONE fp16 operand plane (MCA_HIP_SRP_FP16 here; the
    ADAPTIVE coarse pass is test_adaptive_matches_oracle[ULA16]) -- whitened spectra packed to fp16, 120 pair products per bin on
    v_dot2_f32_f16 (SteeringBeamforming.cpp:104-130 up to the steering sum).  Against the oracle at the plain-fp16 bars (energy map
    2e-4 of the peak, bins at the mode's own tie level), with the power gate (the frame power comes from the same kernel), a dead
    channel (exact zeros: X = 0 as in the reference's own transform, not its partner's rounding noise whitened to unit modulus)
    and runs of frames that do not fill a wave's share."""
    fs, N, F = 48000, 1024, 173 if gate else 77
    xs = synth.ULA16
    pcm = synth.noise_source_stream(xs, np.deg2rad(-33.0), fs, (F + 1) * N // 2, 61, snr_db=25.0).astype(np.float32)
    if gate:
        pcm[:, :150 * 512] *= 1e-3                                     # quiet lead-in: 3 s of floor estimation, then the source
    pcm[11] = 0.0                                                       # a dead microphone
    ctx = api.Context(fs, xs, N, 1.0, 1, use_power_floor=gate, srp_precision=api.SRP_FP16)
    ra = ctx.process_frames_host(pcm[None, :, :(40 + 1) * 512], want_energy=True)
    rb = ctx.process_frames_host(pcm[None, :, 40 * 512:], want_energy=True)
    r = {k: np.concatenate([ra[k], rb[k]], axis=2 if k == "out" else 1) for k in ("bin", "energy", "out") + (("voiced",) if gate else ())}
    o = po.ssl_stream_gated(fs, N, xs, pcm.astype(np.float64), 1, 1.0, gate)
    if gate:
        assert np.array_equal(r["voiced"][0], o["fired"]) and 0 < o["fired"].sum() < F
    assert np.abs(r["energy"][0] - o["energy"]).max() <= 2e-4 * np.abs(o["energy"]).max()
    from parity_helpers import fragile
    for t in np.nonzero(r["bin"][0][:, 0] != o["bin"][:, 0])[0]:
        assert fragile(o["energy"][t], ctx.P, 1, 2e-4), "frame %d: gpu %d oracle %d" % (t, r["bin"][0][t, 0], o["bin"][t, 0])
    from parity_helpers import assert_audio_where_bins_agree
    assert_audio_where_bins_agree(r["out"][0], o["out"], r["bin"][0], o["bin"], 512)
    ctx.close()
