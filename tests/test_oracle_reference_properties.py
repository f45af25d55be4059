"""Pins the CPU oracle to every known answer / property the reference's own tests hold
for the hot path.  CPU only.

* ArrayDescription known-answer table  -- test/test_mcarray.cpp:518-580
* SRP-PHAT DOA within +-7 deg, 4-mic Reem-C, -80..80 deg -- :384-423: on a SUBSTITUTED broadband source (passes); the
  reference's literal 1 kHz sines are kept as strict xfails with the measured errors (PHAT cannot localise a pure tone)
* delay-and-sum inter-source attenuation >= 5.5 dB, 3 two-tone scenes    -- :631-800
* masking band-power windows 70+-10 dB, 2+-0.5 dB -> 5+-1 dB              -- :892-1065
* multiband 2-mic localiser within +-15 deg on 1 kHz sines, -90..90 deg   -- :344-383
"""
import numpy as np
import pytest
from scipy import signal

from mcarray_amd import synth
from oracle import pyoracle as po


def test_array_description_known_answers():
    # test/test_mcarray.cpp:524-578
    xyz = np.array([[0.0, 2, 3], [0.035 * 2, 2, 3], [0.035 * 5, 2, 3], [0.035 * 6, 2, 3]])
    expect = [[0.000, 0.070, 0.175, 0.210],
              [0.070, 0.000, 0.105, 0.140],
              [0.175, 0.105, 0.000, 0.035],
              [0.210, 0.140, 0.035, 0.000]]
    for i in range(4):
        for j in range(4):
            # EXPECT_DOUBLE_EQ = within 4 ulp
            assert po.distance(xyz, i, j) == pytest.approx(expect[i][j], rel=0, abs=4 * np.spacing(0.21))
    assert po.max_distance(xyz) == pytest.approx(0.210, abs=4 * np.spacing(0.21))


def test_grid_sizes():
    # SteeringBeamforming.cpp:39-40 (5 deg -> 37), BinauralLocalisation.cpp:328-329 (3 deg -> 61)
    assert po.num_steps(5.0) == 37
    assert po.num_steps(3.0) == 61
    assert po.num_steps(0.5) == 361
    assert po.doaidx2angle(18, 5.0) == pytest.approx(0.0, abs=1e-6)
    assert po.doaidx2angle(0, 5.0) == pytest.approx(-np.pi / 2, abs=1e-6)


@pytest.mark.parametrize("doa_deg", list(range(-80, 81, 10)))
def test_srp_doa_within_7_degrees_substituted_broadband_source(doa_deg):
    # SUBSTITUTED SIGNAL (the literal one: test_srp_doa_within_7_degrees_literal_sine below).  testBeamformingSoundLocalisation: 4-mic Reem-C, @48 kHz, 1 source, no power floor, tolerance 7 degrees,
    # DOA -80..80 step 10 (test/test_mcarray.cpp:387-417).  N from calculateOrderFromSampleRate(48000, 0.025).
    # The reference reads 1 kHz sine recordings that are not in its tree, and its core never feeds data
    # (test/test_mcarray.cpp:611), so that test never ran.  A noiseless pure tone carries one coherent phase
    # in every bin, which PHAT weighting (north_star: GCC-PHAT) cannot localise -- for ANY implementation --
    # so the property is checked on a far-field broadband source from the same angles and geometry.
    fs = 48000
    N = 1 << 10
    assert po.lib().mca_or_order_from_sample_rate(fs, 0.025) == 10
    F = 12
    pcm = synth.noise_source_stream(synth.REEM_C, np.deg2rad(doa_deg), fs, (F + 1) * N // 2, seed=doa_deg + 1000)
    r = po.ssl_stream(fs, N, synth.REEM_C, pcm.astype(np.float64), 1, 5.0, want_audio=False)
    deg = np.rad2deg(r["doa"][:, 0])
    assert np.all(np.abs(deg - doa_deg) <= 7.0), deg


def _sine_errors(snr_db, weighting="phat"):
    """worst |DOA - truth| over 12 frames per angle for the reference's literal stimulus: 1 kHz sine, amplitude 5000,
    Reem-C, 48 kHz, -80..80 step 10, optional white sensor noise at `snr_db`."""
    from oracle import np_twin as tw
    fs, N, F = 48000, 1024, 12
    errs = []
    for doa_deg in range(-80, 81, 10):
        pcm = synth.sine_stream(synth.REEM_C, np.deg2rad(doa_deg), fs, (F + 1) * N // 2, 1000.0, 5000.0)
        if snr_db is not None:
            rng = np.random.default_rng(doa_deg + 5000)
            pcm = pcm + rng.standard_normal(pcm.shape) * (5000 / np.sqrt(2)) * 10 ** (-snr_db / 20)
        # the reference's whole flow (computeCorrelations -> computeEnergyInDOA -> selectDOA) in the C oracle, with the GCC
        # weighting as a parameter (mca_or_gcc_tau_matrix); the numpy twin is held to the same answer
        r = po.ssl_stream(fs, N, synth.REEM_C, pcm.astype(np.float64), 1, 5.0, want_audio=False, weighting=weighting)
        deg = np.rad2deg(r["doa"][:, 0])
        if doa_deg in (-80, 0, 30):
            t = tw.ssl_stream(fs, N, synth.REEM_C, pcm.astype(np.float64), 1, 5.0, weighting=weighting)
            assert np.array_equal(t["bin"], r["bin"])
        errs.append(float(np.abs(deg - doa_deg).max()))
    return errs


@pytest.mark.xfail(strict=True, reason="the reference's LITERAL stimulus (test_mcarray.cpp:384-423: noiseless 1 kHz sines, files not in its "
                   "tree, test never ran): measured worst error per angle -80..80 = [15, 5, 5, 45, 35, 30, 20, 10, 0, 10, 20, 30, 35, 45, "
                   "30, 40, 50] degrees.  A pure tone puts one coherent phase in every FFT bin through the window's leakage; PHAT (north_star: "
                   "GCC-PHAT) gives all 513 bins unit weight, so the map is not that of the tone's inter-microphone delay.")
def test_srp_doa_within_7_degrees_literal_sine():
    errs = _sine_errors(None)
    assert max(errs) <= 7.0, errs


@pytest.mark.parametrize("snr_db", [40, 20])
@pytest.mark.xfail(strict=True, reason="literal 1 kHz sines plus white sensor noise: measured worst error per angle at 40 dB = [50, 25, 25, 35, 40, "
                   "35, 30, 35, 80, 20, 20, 50, 20, 35, 130, 140, 35], at 20 dB = [50, 25, 125, 35, 85, 90, 30, 35, 80, 80, 40, 50, 20, 25, 130, "
                   "140, 135] degrees: noise does not rescue the property at ANY SNR (60, 30 and 10 dB measured too) -- the 512 noise-only bins "
                   "weigh as much as the tone's bin under PHAT.  See test_srp_doa_within_7_degrees_literal_sine_weighting_none.")
def test_srp_doa_within_7_degrees_literal_sine_with_sensor_noise(snr_db):
    errs = _sine_errors(snr_db)
    assert max(errs) <= 7.0, errs


@pytest.mark.parametrize("snr_db", [None, 40, 20])
def test_srp_doa_within_7_degrees_literal_sine_weighting_none(snr_db):
    """The SAME literal stimulus under gcc_weighting NONE (plain cross-spectrum, mca_or_gcc_tau_matrix / MCA_HIP_GCC_NONE):
    the reference's +-7 degree property (test_mcarray.cpp:390,417) holds at all 17 angles -- in fact the grid angle itself is
    returned -- noiseless and with sensor noise at 40 / 20 dB.  Evidence about the [INFERRED] weighting of
    dsp::GeneralisedCrossCorrelation: either DSPONE's GCC is not PHAT-weighted, or the reference's sine test did
    not pass for its author.  north_star mandates GCC-PHAT, so PHAT stays the default of the oracle and of the HIP path;
    tools/pin_against_dspone.cpp settles the question on a machine that has DSPONE."""
    errs = _sine_errors(snr_db, weighting="none")
    assert max(errs) <= 7.0, errs
    assert max(errs) < 1e-3, errs          # the grid angle itself (float grid: 1e-5 degrees)


SCENES = [  # test/test_mcarray.cpp:640-656
    (800.0, 45.0, 2000.0, -45.0),
    (1000.0, 20.0, 4000.0, -20.0),
    (1000.0, 80.0, 1500.0, 10.0),
]


@pytest.mark.parametrize("f1,d1,f2,d2", SCENES)
def test_delay_and_sum_attenuation(f1, d1, f2, d2):
    # dead test testBeamformingSeparation (test/test_mcarray.cpp:631-800): Reem-C, fs 48 kHz, N = 2048,
    # spectral peak = max |X| over bins floor(f/fs*N)-4 .. +3 (:727-729); attenuation >= 5.5 dB (:757,:785)
    fs, N = 48000, 2048
    xs = synth.REEM_C
    x = (synth.sine_stream(xs, np.deg2rad(d1), fs, N, f1, amplitude=5000.0)
         + synth.sine_stream(xs, np.deg2rad(d2), fs, N, f2, amplitude=5000.0))
    frames = np.stack([po.rfft_ccs(x[c]) for c in range(4)])

    def peak(ccs, f):
        mag = np.hypot(ccs[0::2], ccs[1::2])
        b = int(f / fs * N)
        return mag[b - 4:b + 4].max()

    in1 = np.mean([peak(frames[c], f1) for c in range(4)])
    in2 = np.mean([peak(frames[c], f2) for c in range(4)])
    out = po.beamformer_process_frame(fs, xs, frames, np.deg2rad(d1))
    att1 = 20 * np.log10((peak(out, f1) / in1) / (peak(out, f2) / in2))
    out = po.beamformer_process_frame(fs, xs, frames, np.deg2rad(d2))
    att2 = 20 * np.log10((peak(out, f2) / in2) / (peak(out, f1) / in1))
    assert att1 >= 5.5 and att2 >= 5.5, (att1, att2)


def _bandpass(lo, hi):
    # dsp::BandPassFIRFilter(256, lo, hi) stand-in [BUILD-DEFINES]: firwin(257), frequencies in cycles/sample
    return signal.firwin(257, [lo, hi], pass_zero=False, fs=1.0)


def test_spatial_masking_power_windows():
    # testSpatialMaskingCore (test/test_mcarray.cpp:892-958); FastBinauralMasking(16000, 0.086, 500, 5000, FULL)
    fs, N = 16000, 1024
    n = 5 * 1024
    magn, delay = 5000, 6
    interest = synth.tone16(n, magn, 0.1).astype(np.float64)
    interf_l = synth.tone16(n, magn, 0.3).astype(np.float64)
    interf_r = np.zeros(n)
    interf_r[:n - delay] = interf_l[delay:]
    L = interest + interf_l
    R = interest + interf_r
    m = po.Masking(fs, N, 0.086, 500, 5000, po.FULL, po.BOTH)
    ol, orr = m.stream(L, R)
    sigf = _bandpass(0.05, 0.15)
    intf = _bandpass(0.25, 0.35)
    p_sig = po.log_power(signal.lfilter(sigf, 1.0, ol))
    p_int = po.log_power(signal.lfilter(intf, 1.0, ol))
    assert abs(70 - p_sig) <= 10   # :943
    assert abs(70 - p_int) <= 10   # :956 (passes although the interferer is not attenuated, as in the reference)


def test_temporal_masking_power_windows():
    # testTemporalMaskingCore (test/test_mcarray.cpp:960-1065)
    fs, N = 16000, 1024
    magn = 5000
    delay = int(0.1 * fs)
    tonestep = int(0.1 * fs)
    freqstep = np.float32(0.01)
    n = 50 * 1024
    tone = np.zeros(n)
    i, sfreq, interest_start = 0, np.float32(0.01), 0
    interest_freq = np.float32(0.2)
    while i < n - tonestep and sfreq < 0.5:
        if interest_freq - freqstep / 2 < sfreq < interest_freq + freqstep / 2:
            interest_start = i
        tone[i:i + tonestep] = synth.tone16(tonestep, magn, float(sfreq))
        i += tonestep
        sfreq = np.float32(sfreq + freqstep)
    sig = np.zeros(n)
    tb = tone.copy()
    for k in range(6):                      # i = 0..nreverb (:1014-1018), int16 halving
        tb = np.trunc(tb / 2)
        sig[delay * k:] += tb[:n - delay * k]
    m = po.Masking(fs, N, 0.086, 500, 5000, po.FULL, po.BOTH)
    ol, orr = m.stream(sig, sig)
    out_len = len(ol)
    sigf = _bandpass(0.19, 0.21)
    intf = _bandpass(0.15, 0.20)
    sl = slice(interest_start, interest_start + tonestep)

    def diff(x):
        return po.log_power(signal.lfilter(sigf, 1.0, x[:out_len])[sl]) - po.log_power(signal.lfilter(intf, 1.0, x[:out_len])[sl])

    before = diff(sig)
    after = diff(ol)
    assert abs(before - 2) < 0.5, before     # :1039-1045
    assert abs(after - 5) < 1.0, after       # :1061-1064


@pytest.mark.parametrize("doa_deg", list(range(-90, 91, 10)))
def test_multiband_sine_within_15_degrees(doa_deg):
    """test/test_mcarray.cpp:344-383 (testMultibandBinauralLocalisation): 1 kHz sine, 48 kHz, 0.086 m pair,
    25 sub-bands, usePowerFloor=false, every callback within +-15 degrees of the true angle."""
    fs, N, F = 48000, 1024, 40
    xs = [0.0, 0.086]
    pcm = synth.sine_stream(xs, np.deg2rad(doa_deg), fs, (F + 1) * N // 2, 1000.0, 5000.0)
    X = po.stft_frames(pcm, N)
    m = po.Multiband(fs, xs, N + 2, 25, False)
    assert m.D == 37
    for t in range(F):
        r = m.process(X[t, 0], X[t, 1])
        assert r["fired"]
        assert abs(np.rad2deg(r["doa"]) - doa_deg) <= 15.0, (t, np.rad2deg(r["doa"]))
        assert 0.0 <= r["prob"] <= 1.0
