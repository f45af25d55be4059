"""GPU parity of the MVDR-style beamformer (mca_hip_mvdr_*, kernels_mvdr.hip; BASELINE.json configs[3]) through the
C ABI, against the CPU oracle (SURVEY A.9 restated in oracle/mca_oracle.c) and the golden vectors of the numpy twin.
There is no reference counterpart (the reference has delay-and-sum only); the link to the reference is the
delay-and-sum limit, checked against the oracle's restatement of Beamformer.cpp:51-71.

Tolerances (fp32 GPU vs fp64 oracle).  The solve runs through the Cholesky factor of the loaded covariance, whose
condition number reaches sqrt(M / loading) ~ 130 while a stream's first M-1 covariances are rank deficient:
  * beamformed spectra:  |gpu - oracle| <= 5e-4 * max|spectrum| of the call   (measured <= 1.7e-4)
  * audio:               |gpu - oracle| <= 5e-4 * max|audio|                  (measured <= 1.3e-4)
  * covariance state:    |gpu - oracle| <= 5e-6 * max|Phi|                    (measured <= 6.1e-7)
"""
import os

import numpy as np
import pytest

from mcarray_amd import api, synth
from oracle import pyoracle as po

pytestmark = pytest.mark.gpu

SPEC_TOL, AUDIO_TOL, COV_TOL = 5e-4, 5e-4, 5e-6


def _scene(xs, fs, N, F, a):
    n = (F + 1) * N // 2
    return (synth.noise_source_stream(xs, np.deg2rad(20.0 - 30 * a), fs, n, 5 + a)
            + synth.noise_source_stream(xs, np.deg2rad(-50.0 + 40 * a), fs, n, 15 + a, snr_db=60)).astype(np.float32)


def _ospec(o):
    return o["spec"][:, 0::2] + 1j * o["spec"][:, 1::2]


@pytest.mark.parametrize("xs,fs,N,F", [
    (synth.ULA16, 48000, 1024, 24),       # BASELINE configs[3] geometry
    (synth.ULA8, 48000, 1024, 20),
    (synth.REEM_C, 16000, 512, 20),       # the reference's 4-microphone test array
    (synth.BINAURAL, 16000, 1024, 12),
    ([0.0, 0.03, 0.07, 0.10, 0.20], 8000, 256, 30),
    (synth.ULA16, 96000, 2048, 5),
])
def test_mvdr_stream_matches_oracle(xs, fs, N, F):
    A = 3
    pcm = np.stack([_scene(xs, fs, N, F, a) for a in range(A)])
    doa = (np.deg2rad(20.0 - 30 * np.arange(A))[:, None] + 0.01 * np.arange(F)[None, :]).astype(np.float32)
    bf = api.MvdrBeamformer(fs, xs, N, max_streams=A)
    r = bf.process(pcm, doa, want_spec=True)
    ogs = []
    for a in range(A):
        og = po.MVDR(fs, N, xs)
        o = og.stream(pcm[a].astype(np.float64), doa[a].astype(np.float64), want_spec=True)
        sp = _ospec(o)
        assert np.abs(r["spec"][a] - sp).max() <= SPEC_TOL * np.abs(sp).max(), a
        assert np.abs(r["out"][a] - o["out"]).max() <= AUDIO_TOL * np.abs(o["out"]).max(), a
        assert np.abs(bf.covariance(a) - og.covariance()).max() <= COV_TOL * np.abs(og.covariance()).max(), a
        ogs.append(og)
    # a second call continues the recursion and the overlap-add
    r2 = bf.process(pcm, doa[:, ::-1].copy(), want_spec=True)
    o2 = ogs[1].stream(pcm[1].astype(np.float64), doa[1, ::-1].astype(np.float64), want_spec=True)
    assert np.abs(r2["spec"][1] - _ospec(o2)).max() <= SPEC_TOL * np.abs(_ospec(o2)).max()
    # the oracle's stream() restarts its overlap-add tail per call; the GPU carries it: compare past the first hop
    hop = N // 2
    assert np.abs(r2["out"][1, hop:] - o2["out"][hop:]).max() <= AUDIO_TOL * np.abs(o2["out"]).max()


@pytest.mark.parametrize("M", [3, 6, 7, 9, 11, 13, 15])
def test_mvdr_every_row_slot_count(M):
    """The solve kernel deals the rows of the triangle cyclically over four lanes (1 ... 4 row slots per lane): channel counts
    that leave the last slot partly empty, a bin count that is no multiple of the 16 problems of a wave, one frame per call."""
    fs, N, F, A = 16000, 256, 9, 2
    rng = np.random.default_rng(M)
    xs = np.sort(rng.uniform(0.0, 0.04 * M, M))
    pcm = np.stack([_scene(xs, fs, N, F, a) for a in range(A)])
    doa = rng.uniform(-1.3, 1.3, (A, F)).astype(np.float32)
    bf = api.MvdrBeamformer(fs, xs, N, max_streams=A)
    hop = N // 2
    specs = [bf.process(pcm[:, :, t * hop:(t + 2) * hop], doa[:, t:t + 1], want_spec=True)["spec"] for t in range(F)]
    spec = np.concatenate(specs, axis=1)
    for a in range(A):
        og = po.MVDR(fs, N, xs)
        sp = _ospec(og.stream(pcm[a].astype(np.float64), doa[a].astype(np.float64), want_spec=True))
        assert np.abs(spec[a] - sp).max() <= SPEC_TOL * np.abs(sp).max(), (M, a)
        assert np.abs(bf.covariance(a) - og.covariance()).max() <= COV_TOL * np.abs(og.covariance()).max()


@pytest.mark.parametrize("name", ["mvdr_ula16_48k", "mvdr_reemc_16k"])
def test_mvdr_matches_golden(golden_dir, name):
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    bf = api.MvdrBeamformer(int(g["fs"]), g["xs"], int(g["N"]))
    r = bf.process(g["pcm"], g["doa"][None], want_spec=True)
    assert np.abs(r["spec"][0] - g["spec"]).max() <= SPEC_TOL * np.abs(g["spec"]).max()
    assert np.abs(r["out"][0] - g["out"]).max() <= AUDIO_TOL * np.abs(g["out"]).max()
    assert np.abs(bf.covariance(0)[::64] - g["phi_last"]).max() <= COV_TOL * np.abs(g["phi_last"]).max()


def test_mvdr_tail_workgroups_cut_along_the_frames_match_oracle():
    """Calls of more than 512 solve workgroups (here 64 streams x 513 bins / 64 = 513) put the workgroups behind the last whole
    round into a second launch cut along the frames, each piece repeating the covariance recursion of the frames before its own
    (api_mvdr.hip): spectra (the last bins on their own as well), audio and the covariance of sampled streams -- the last
    stream is the one in the tail -- against the oracle, over two calls."""
    fs, N, F, A = 16000, 1024, 10, 64
    xs = synth.REEM_C
    base = np.stack([_scene(xs, fs, N, 2 * F, a) for a in range(3)])
    pick = np.arange(A) % 3
    pcm = base[pick]
    doa = (np.deg2rad(20.0 - 30 * pick)[:, None] + 0.01 * np.arange(2 * F)[None, :]).astype(np.float32)
    hop = N // 2
    bf = api.MvdrBeamformer(fs, xs, N, max_streams=A)
    r1 = bf.process(pcm[:, :, :(F + 1) * hop].copy(), doa[:, :F].copy(), want_spec=True)
    r2 = bf.process(pcm[:, :, F * hop:].copy(), doa[:, F:].copy(), want_spec=True)
    spec = np.concatenate([r1["spec"], r2["spec"]], axis=1)
    out = np.concatenate([r1["out"], r2["out"]], axis=1)
    for a in (0, 31, 63):
        og = po.MVDR(fs, N, xs)
        o = og.stream(pcm[a].astype(np.float64), doa[a].astype(np.float64), want_spec=True)
        sp = _ospec(o)
        assert np.abs(spec[a] - sp).max() <= SPEC_TOL * np.abs(sp).max(), a
        assert np.abs(spec[a][:, -64:] - sp[:, -64:]).max() <= SPEC_TOL * np.abs(sp).max(), a
        assert np.abs(out[a] - o["out"]).max() <= AUDIO_TOL * np.abs(o["out"]).max(), a
        assert np.abs(bf.covariance(a) - og.covariance()).max() <= COV_TOL * np.abs(og.covariance()).max(), a
    # streams with the same input give the same bits wherever they sit in the batch
    assert np.array_equal(spec[0], spec[3]) and np.array_equal(out[1], out[61])


def test_mvdr_chunked_calls_equal_one_call():
    """The state (covariances, traces, overlap-add tail) carried between calls reproduces a single long call bit for bit."""
    fs, N, F = 48000, 1024, 48
    xs = synth.ULA16
    pcm = _scene(xs, fs, N, F, 0)[None]
    doa = np.full((1, F), 0.3, dtype=np.float32)
    one = api.MvdrBeamformer(fs, xs, N).process(pcm, doa, want_spec=True)
    bf = api.MvdrBeamformer(fs, xs, N)
    hop = N // 2
    outs, specs = [], []
    for (t0, t1) in [(0, 1), (1, 18), (18, 19), (19, 48)]:
        r = bf.process(pcm[:, :, t0 * hop:(t1 + 1) * hop], doa[:, t0:t1], want_spec=True)
        outs.append(r["out"]); specs.append(r["spec"])
    assert np.array_equal(np.concatenate(specs, axis=1), one["spec"])
    assert np.array_equal(np.concatenate(outs, axis=1), one["out"])
    bf.reset()
    again = bf.process(pcm, doa, want_spec=True)
    assert np.array_equal(again["out"], one["out"])


def test_mvdr_full_size_properties():
    """BASELINE configs[3] at full size (16 microphones, 256 streams x 64 frames) through size-independent properties:
    streams are independent (bit-exact whatever their position in the batch), a power-of-two gain scales the output
    exactly (the weights depend on Phi only up to a factor), digital silence gives silence, and the oracle agrees on
    a sample of streams."""
    fs, N, F, A = 48000, 1024, 64, 256
    xs = synth.ULA16
    hop = N // 2
    rng = np.random.default_rng(7)
    base = np.stack([_scene(xs, fs, N, F, a) for a in range(4)])
    pcm = np.empty((A, 16, (F + 1) * hop), dtype=np.float32)
    doa = np.empty((A, F), dtype=np.float32)
    src = rng.integers(0, 4, A)
    gain = 2.0 ** rng.integers(-3, 3, A)
    look = rng.uniform(-1.2, 1.2, 4).astype(np.float32)
    for a in range(A):
        pcm[a] = base[src[a]] * gain[a]
        doa[a] = look[src[a]]
    pcm[200] = 0.0
    bf = api.MvdrBeamformer(fs, xs, N, max_streams=A)
    r = bf.process(pcm, doa)
    assert np.all(np.isfinite(r["out"]))
    assert np.all(r["out"][200] == 0.0)
    first = {}
    for a in range(A):
        if a == 200:
            continue
        if src[a] not in first:
            first[src[a]] = a
        b = first[src[a]]
        assert np.array_equal(r["out"][a] / gain[a], r["out"][b] / gain[b]), (a, b)
    for s, a in first.items():
        o = po.MVDR(fs, N, xs).stream(pcm[a].astype(np.float64), doa[a].astype(np.float64))
        assert np.abs(r["out"][a] - o["out"]).max() <= AUDIO_TOL * np.abs(o["out"]).max(), a


def test_mvdr_heavy_loading_is_delay_and_sum():
    """loading -> infinity: w = d/M, the reference's Beamformer::processFrame (Beamformer.cpp:51-71) as restated in the oracle."""
    fs, N, F = 48000, 1024, 6
    xs = synth.ULA8
    pcm = _scene(xs, fs, N, F, 1)
    bf = api.MvdrBeamformer(fs, xs, N, loading=1e9)
    r = bf.process(pcm, 0.4, want_spec=True)
    X = po.stft_frames(pcm.astype(np.float64), N)
    for t in range(F):
        ref = po.beamformer_process_frame(fs, xs, X[t], float(np.float32(0.4)))
        refc = ref[0::2] + 1j * ref[1::2]
        assert np.abs(r["spec"][0, t] - refc).max() <= 2e-5 * np.abs(refc).max(), t


def test_mvdr_rejects_bad_configurations():
    with pytest.raises(api.MCArrayHipError):
        api.MvdrBeamformer(48000, synth.ULA8, 1000)                    # not a power of two
    with pytest.raises(api.MCArrayHipError):
        api.MvdrBeamformer(48000, [0.01 * m for m in range(17)], 1024)  # more than 16 microphones
    with pytest.raises(api.MCArrayHipError):
        api.MvdrBeamformer(48000, synth.ULA8, 1024, loading=0.0)
    with pytest.raises(api.MCArrayHipError):
        api.MvdrBeamformer(48000, synth.ULA8, 1024, alpha=1.0)
    bf = api.MvdrBeamformer(48000, synth.ULA8, 1024, max_streams=1)
    with pytest.raises(api.MCArrayHipError):
        bf.process(np.zeros((2, 8, 1024), dtype=np.float32), 0.0)      # more streams than the context holds


def test_module_state_blobs_resume_streams():
    """mca_hip_{mvdr,mask,mb}_state_save/_load: a second context of the same configuration continues a stream bit for bit;
    a context of another configuration refuses the blob."""
    fs, N, F = 16000, 512, 40
    hop = N // 2
    # MVDR
    xs = synth.REEM_C
    pcm = _scene(xs, fs, N, F, 0)[None]
    a = api.MvdrBeamformer(fs, xs, N)
    one = a.process(pcm, 0.2)["out"]
    b, c = api.MvdrBeamformer(fs, xs, N), api.MvdrBeamformer(fs, xs, N)
    first = b.process(pcm[:, :, :(15 + 1) * hop], 0.2)["out"]
    c.state_load(b.state_save())
    rest = c.process(pcm[:, :, 15 * hop:], 0.2)["out"]
    assert np.array_equal(np.concatenate([first, rest], axis=1), one)
    with pytest.raises(api.MCArrayHipError):
        api.MvdrBeamformer(fs, xs, N, alpha=0.9).state_load(b.state_save())
    # masking (1024 = tuned kernel, 512 = any-length kernel)
    for Nm in (1024, 512):
        h2 = Nm // 2
        rng = np.random.default_rng(Nm)
        x2 = (rng.standard_normal((1, 2, (F + 1) * h2)) * 0.1).astype(np.float32)
        x2[0, 1] = np.roll(x2[0, 0], 1)
        m1 = api.FastBinauralMasking(fs, 0.086, 300.0, 5000.0, api.RELATIVE, api.BOTH, fft_size=Nm)
        o1, d1 = m1.process(x2)
        m2, m3 = (api.FastBinauralMasking(fs, 0.086, 300.0, 5000.0, api.RELATIVE, api.BOTH, fft_size=Nm) for _ in range(2))
        oa, da = m2.process(x2[:, :, :(15 + 1) * h2])
        m3.state_load(m2.state_save())
        ob, db = m3.process(x2[:, :, 15 * h2:])
        assert np.array_equal(np.concatenate([da, db], axis=1), d1)
        assert np.abs(np.concatenate([oa, ob], axis=2) - o1).max() <= 1e-6 * np.abs(o1).max()
        with pytest.raises(api.MCArrayHipError):
            api.FastBinauralMasking(fs, 0.086, 300.0, 5000.0, api.FULL, api.BOTH, fft_size=Nm).state_load(m2.state_save())
    # multiband localiser
    xb = synth.noise_source_stream(synth.BINAURAL, 0.5, 48000, (F + 1) * 512, 9)[None]
    l1 = api.MultibandBinarualLocalisation(48000, synth.BINAURAL, 15, False)
    r1 = l1.process(xb)
    l2, l3 = (api.MultibandBinarualLocalisation(48000, synth.BINAURAL, 15, False) for _ in range(2))
    ra = l2.process(xb[:, :, :(15 + 1) * 512])
    l3.state_load(l2.state_save())
    rb = l3.process(xb[:, :, 15 * 512:])
    assert np.array_equal(np.concatenate([ra["doa"], rb["doa"]], axis=1), r1["doa"])
    assert np.array_equal(np.concatenate([ra["prob"], rb["prob"]], axis=1), r1["prob"])
    with pytest.raises(api.MCArrayHipError):
        api.MultibandBinarualLocalisation(48000, synth.BINAURAL, 12, False).state_load(l2.state_save())


def test_localise_then_mvdr_16_microphones():
    """BASELINE configs[3] end to end: the 16-microphone localiser (SRP over 361 angles) supplies the look direction per frame,
    the MVDR beamformer steers with it; both stages against the oracle's (DOA bins bit-exact, then the MVDR audio)."""
    fs, N, F, A = 48000, 1024, 48, 3
    xs = synth.ULA16
    pcm = np.stack([_scene(xs, fs, N, F, a) for a in range(A)])
    loc = api.Context(fs, xs, N, 0.5, 1, max_arrays=A)
    r = loc.process_frames_host(pcm, want_audio=False)
    bf = api.MvdrBeamformer(fs, xs, N, max_streams=A)
    out = bf.process(pcm, r["doa"][:, :, 0])["out"]
    for a in range(A):
        o = po.ssl_stream(fs, N, xs, pcm[a].astype(np.float64), 1, 0.5, want_audio=False)
        assert np.array_equal(r["bin"][a], o["bin"]), a
        om = po.MVDR(fs, N, xs).stream(pcm[a].astype(np.float64), r["doa"][a, :, 0].astype(np.float64))
        assert np.abs(out[a] - om["out"]).max() <= AUDIO_TOL * np.abs(om["out"]).max(), a
    # the localiser settles on one of the two (equally loud) sources of every scene
    for a in range(A):
        med = np.rad2deg(np.median(r["doa"][a, 8:, 0]))
        assert min(abs(med - (20.0 - 30 * a)), abs(med - (-50.0 + 40 * a))) <= 1.0, (a, med)
