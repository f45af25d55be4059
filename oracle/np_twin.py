"""numpy twin of the C oracle (oracle/mca_oracle.c).

TEST INFRASTRUCTURE ONLY.  An independent, vectorised restatement of the same
spec used to cross-check the C oracle in the CPU test suite
and to emit the golden fixtures under tests/golden/ (tests/golden/make_golden.py).
It is never imported by the product package.  "parity unpinned" against an
executed reference (see mca_oracle.h).
"""
import numpy as np

C_SOUND = 346.1  # microhponeArrayHelpers.cpp:42
f32 = np.float32


def doa_step(step_deg):
    return f32(step_deg * np.pi / 180.0)  # SteeringBeamforming.cpp:39


def num_steps(step):
    return int(np.round(np.pi / float(step)) + 1)  # :40  (C round() = half away from zero; ties do not occur)


def doaidx2angle(idx, step):
    # microhponeArrayHelpers.cpp:117-120 : float product, double subtraction, float return
    prod = f32(idx) * f32(step)
    return f32(np.float64(prod) - np.pi / 2)


def delay_samples(doa, dist, fs):
    # :46-72 : (float dist * sin(float doa) in double) / 346.1 -> float ; * fs in float
    d = f32(np.float64(f32(dist)) * np.sin(np.float64(f32(doa))) / C_SOUND)
    return f32(d * f32(fs))


def pair_list(M):
    return [(i, j) for i in range(M) for j in range(i + 1, M)]


def xyz_of(x):
    a = np.asarray(x, dtype=np.float64)
    if a.ndim == 1:
        a = np.stack([a, np.zeros_like(a), np.zeros_like(a)], axis=1)
    return a


def distance(xyz, i, j):
    d = xyz[j] - xyz[i]
    return float(np.sqrt(d[0] ** 2 + d[1] ** 2 + d[2] ** 2))


def delay_table(fs, xyz, step_deg):
    """[P][D] float64 holding the float delays of generateLookupTable (:58-94)."""
    xyz = xyz_of(xyz)
    step = doa_step(step_deg)
    D = num_steps(step)
    pairs = pair_list(len(xyz))
    tab = np.empty((len(pairs), D))
    for p, (i, j) in enumerate(pairs):
        dist = distance(xyz, i, j)
        for d in range(D):
            tab[p, d] = float(delay_samples(doaidx2angle(d, step), dist, fs))
    return tab


def hann(N):
    return 0.5 - 0.5 * np.cos(2 * np.pi * np.arange(N) / N)


def to_c(ccs):
    """CCS double[..., N+2] -> complex[..., K]"""
    return ccs[..., 0::2] + 1j * ccs[..., 1::2]


def to_ccs(X):
    out = np.empty(X.shape[:-1] + (2 * X.shape[-1],))
    out[..., 0::2] = X.real
    out[..., 1::2] = X.imag
    return out


def stft_frames(pcm, N):
    """[M][(F+1)*hop] -> complex [F][M][K]"""
    pcm = np.asarray(pcm, dtype=np.float64)
    hop = N // 2
    F = pcm.shape[1] // hop - 1
    w = hann(N)
    idx = np.arange(N)[None, :] + hop * np.arange(F)[:, None]
    fr = pcm[:, idx] * w  # [M][F][N]
    return np.fft.rfft(fr, axis=-1).transpose(1, 0, 2)


def gcc_phat(A, B, tau, K, weighting="phat"):
    """R[d] = Re sum_k Ghat[k] exp(+j 2 pi k tau_d / N)  (SURVEY A.3); weighting "phat" (Ghat = G / |G|) or "none" (Ghat = G)"""
    G = A * np.conj(B)
    if weighting == "phat":
        Gh = G / np.maximum(np.abs(G), 1e-30)
    elif weighting == "none":
        Gh = G
    else:
        raise ValueError(weighting)
    N = 2 * (K - 1)
    ph = 2 * np.pi * np.outer(tau, np.arange(K)) / N
    return (Gh[None, :] * np.exp(1j * ph)).sum(axis=1)


def srp_map(X, delays, weighting="phat"):
    """X complex [M][K]; delays [P][D] -> per-pair R [P][D] (real)."""
    M, K = X.shape
    pairs = pair_list(M)
    R = np.empty((len(pairs), delays.shape[1]))
    for p, (i, j) in enumerate(pairs):
        R[p] = gcc_phat(X[i], X[j], delays[p], K, weighting).real
    return R


MU = float(f32(0.8))
ONE_MINUS_MU = float(f32(1) - f32(0.8))


def energy_update(E_prev, R):
    """computeEnergyInDOA :132-144 (summation order p = 0..P-1)."""
    E = MU * E_prev
    for p in range(R.shape[0]):
        E = E + ONE_MINUS_MU * R[p]
    return E


def median3(x):
    xp = np.concatenate([x[:1], x, x[-1:]])
    return np.median(np.stack([xp[:-2], xp[1:-1], xp[2:]]), axis=0)


def select_doa(E, n_pairs, step, n_sources=1):
    """selectDOA :146-195. returns (doa[S], prob[S], bin[S])"""
    mn = -15.0 * n_pairs
    En = (E - mn) / (-2 * mn)
    fd = En[1:] - En[:-1]
    fd = np.where(fd < 0, 1.0, np.where(fd > 0, 0.0, fd))
    fd = median3(fd)
    sd = (fd[1:] - fd[:-1]) * En[1:-1]
    doa, prob, bins = [], [], []
    for _ in range(n_sources):
        idx = int(np.argmax(sd))  # first max
        prob.append(sd[idx])
        sd[idx] = 0
        bins.append(idx + 1)
        doa.append(float(doaidx2angle(idx + 1, step)))
    return np.array(doa), np.array(prob), np.array(bins, dtype=np.int32)


def beamformer(fs, xyz, X, doa):
    """Beamformer::processFrame (Beamformer.cpp:51-71). X complex [M][K] -> complex [K]"""
    xyz = xyz_of(xyz)
    M, K = X.shape
    N = 2 * (K - 1)
    k = np.arange(K)
    out = np.zeros(K, dtype=complex)
    for c in range(M):
        slope = 2 * np.pi * fs / N / C_SOUND * xyz[c, 0] * np.cos(doa + np.pi / 2)
        out += X[c] * np.exp(1j * slope * k)
    return out / M


def irfft_ccs(Y, N):
    Y = Y.copy()
    Y[..., 0] = Y[..., 0].real
    Y[..., -1] = Y[..., -1].real
    return np.fft.irfft(Y, n=N, axis=-1)


def ssl_stream(fs, N, xyz, pcm, n_sources=1, step_deg=5.0, weighting="phat"):
    xyz = xyz_of(xyz)
    M = len(xyz)
    hop = N // 2
    X = stft_frames(pcm, N)
    F = X.shape[0]
    step = doa_step(step_deg)
    delays = delay_table(fs, xyz, step_deg)
    D = delays.shape[1]
    P = delays.shape[0]
    E = np.zeros(D)
    bins = np.empty((F, n_sources), dtype=np.int32)
    doas = np.empty((F, n_sources))
    probs = np.empty((F, n_sources))
    emap = np.empty((F, D))
    nout = min(M, n_sources)
    out = np.zeros((nout, F * hop))
    tail = np.zeros((nout, hop))
    for t in range(F):
        R = srp_map(X[t], delays, weighting)
        E = energy_update(E, R)
        emap[t] = E
        doas[t], probs[t], bins[t] = select_doa(E.copy(), P, step, n_sources)
        for s in range(nout):
            y = irfft_ccs(beamformer(fs, xyz, X[t], doas[t, s]), N)
            out[s, t * hop:(t + 1) * hop] = tail[s] + y[:hop]
            tail[s] = y[hop:]
    return dict(bin=bins, doa=doas, prob=probs, out=out, energy=emap)


# ---- masking -------------------------------------------------
def hz2mel(f):
    return 2595.0 * np.log10(1.0 + f / 700.0)


def mel2hz(m):
    return 700.0 * (10.0 ** (m / 2595.0) - 1.0)


def mel_filterbank(N, nbins, fs, fmin, fmax):
    K = N // 2 + 1
    edges = mel2hz(hz2mel(fmin) + (hz2mel(fmax) - hz2mel(fmin)) * np.arange(nbins + 2) / (nbins + 1))
    f = np.arange(K) * fs / N
    H = np.zeros((nbins, K))
    for b in range(nbins):
        f0, f1, f2 = edges[b], edges[b + 1], edges[b + 2]
        up = (f > f0) & (f <= f1)
        dn = (f > f1) & (f < f2)
        H[b, up] = (f[up] - f0) / (f1 - f0)
        H[b, dn] = (f2 - f[dn]) / (f2 - f1)
    return H, edges[1:-1] / fs


class Masking:
    LAM = float(f32(0.04))
    ONE_MINUS_LAM = float(f32(1) - f32(0.04))
    REJECT = float(f32(0.999))
    RHO = float(f32(0.01))

    def __init__(self, fs, N, d, flo, fhi, method=1, alg=0):
        self.fs, self.N, self.K = fs, N, N // 2 + 1
        self.method, self.alg = method, alg
        self.H, self.center = mel_filterbank(N, 45, fs, float(f32(flo)), float(f32(fhi)))
        self.thr = np.cos(self.center * fs * 2 * np.pi * d * np.sin(10 * np.pi / 180) / C_SOUND)
        self.Q = np.zeros(45)
        self.noise = np.zeros(45)
        self.first = 0

    def _mask(self, F, b, factor):
        Kh = self.N // 2  # "first N doubles" = N/2 complex bins
        if self.method == 3:
            F[:Kh] /= 1000
        elif self.method == 1:
            f = np.mean(np.abs(F) ** 2) * self.RHO
            f = self.RHO if self.Q[b] < 1e-10 else f / self.Q[b]
            F[:Kh] *= np.sqrt(f)
        elif self.method == 0:
            F[:Kh] /= factor
        elif self.method == 4:
            pw = np.sqrt(np.mean(np.abs(F[:Kh]) ** 2))
            f = self.noise[b] / pw if pw > 0 else 1.0
            if self.first >= 2:
                F[:Kh] *= f
        return F

    def process(self, L, R):
        """L, R complex [K] -> (outL, outR, decisions[45])"""
        if self.method == 5:
            return L, R, np.zeros(45, dtype=np.int32)
        Kh = self.N // 2
        oL = np.zeros(self.K, dtype=complex)
        oR = np.zeros(self.K, dtype=complex)
        dec = np.zeros(45, dtype=np.int32)
        for b in range(45):
            Lb = L * self.H[b]
            Rb = R * self.H[b]
            mix = Lb[:Kh] / 2 + Rb[:Kh] / 2
            P = np.sqrt(np.mean(np.abs(mix) ** 2))
            self.Q[b] = self.Q[b] * self.LAM + self.ONE_MINUS_LAM * P
            temp = P < self.REJECT * self.Q[b]
            spat = False
            if self.alg in (0, 1):
                num = np.mean((np.conj(Lb) * Rb).real)
                if num == 0:
                    nc = 0.0
                else:
                    den = np.sqrt(np.mean(np.abs(Lb) ** 2) * np.mean(np.abs(Rb) ** 2))
                    nc = 1.0 if den == 0 else num / den
                spat = nc < self.thr[b]
                if self.alg == 1:
                    temp = False
            if spat:
                Lb = self._mask(Lb, b, 10.0)
                Rb = self._mask(Rb, b, 10.0)
                dec[b] = 2
            elif temp:
                Lb = self._mask(Lb, b, 3.0)
                Rb = self._mask(Rb, b, 3.0)
                dec[b] = 1
            oL += Lb
            oR += Rb
        self.first += 1
        if self.first < 2:
            self.noise = self.Q.copy()
        return oL, oR, dec


# ---- multiband 2-mic localiser (MultibandBinarualLocalisation.cpp:52-258) -------------------
def linear_filterbank(N, nbins, fs, fmin, fmax):
    """[BUILD-DEFINES] LINEAR bank of dsp::SubBandSTFTAnalysis: unit-peak triangles, linearly spaced edges."""
    K = N // 2 + 1
    edges = fmin + (fmax - fmin) * np.arange(nbins + 2) / (nbins + 1)
    f = np.arange(K) * fs / N
    H = np.zeros((nbins, K))
    for b in range(nbins):
        f0, f1, f2 = edges[b], edges[b + 1], edges[b + 2]
        up = (f > f0) & (f <= f1)
        dn = (f > f1) & (f < f2)
        H[b, up] = (f[up] - f0) / (f1 - f0)
        H[b, dn] = (f2 - f[dn]) / (f2 - f1)
    return H


def fft_power(X, ccs_len):
    """dsp::SignalPower::FFTPower over the first ccs_len doubles of each channel's CCS buffer."""
    n, k = ccs_len - 2, ccs_len // 2
    w = np.full(k, 2.0)
    w[0] = w[-1] = 1.0
    return float(np.mean([(w * np.abs(x[:k]) ** 2).sum() / n ** 2 for x in X]))


def multiband_stream(fs, N, xs, pcm, nbins=15, use_power_floor=True):
    """whole stream; returns dict(fired, doa, prob, power, band_idx, band_corr, energy_in_doa) per frame."""
    xyz = xyz_of(xs)
    dist = distance(xyz, 0, 1)
    step = f32(5 * np.pi / 180)
    D = int(np.floor(np.pi / float(step)) + 1)
    ang = np.array([f32(float(f32(i) * step) - np.pi / 2) for i in range(D)], dtype=np.float32)
    tau = np.array([float(f32(f32((float(f32(dist)) * np.sin(float(a))) / 346.1) * f32(fs))) for a in ang])
    fmax = float(f32(346.1 / float(f32(2) * f32(dist))))
    H = linear_filterbank(N, nbins, fs, 100.0, fmax)
    X = stft_frames(np.asarray(pcm, dtype=np.float64), N)
    F, K = X.shape[0], N // 2 + 1
    mem, omm = float(f32(0.4)), float(f32(1) - f32(0.4))
    prev = np.zeros((nbins, D))
    acc, consumed, est, cur, pr = 0.0, 0, False, 0.0, -1.0
    out = dict(fired=np.zeros(F, bool), doa=np.zeros(F), prob=np.zeros(F), power=np.zeros(F),
               band_idx=np.zeros((F, nbins), np.int32), band_corr=np.zeros((F, nbins, D)), energy_in_doa=np.zeros((F, D)))
    for t in range(F):
        E = np.zeros(D)
        for b in range(nbins):
            L, R = X[t, 0] * H[b], X[t, 1] * H[b]
            c = omm * gcc_phat(L, R, tau, K).real + mem * prev[b]
            prev[b] = c
            idx = int(np.argmax(c))
            E[idx] += fft_power([L, R], N + 2)
            out["band_idx"][t, b] = idx
            out["band_corr"][t, b] = c
        if not est:
            half = (N + 2) // 2
            acc += fft_power([X[t, 0], X[t, 1]], half) * (2 * half - 2)
            consumed += 2 * half - 2
            if consumed >= int(3 * fs):
                est = True
                acc = 10 * np.log10(acc / consumed) + 3.0
            power = acc
        else:
            power = fft_power([X[t, 0], X[t, 1]], N + 2)
        if power > acc or not use_power_floor:
            s = E.sum()
            i = int(np.argmax(E))
            pr = E[i] / s if s != 0 else 0.0
            cur = float(ang[i])
            out["fired"][t] = True
        else:
            pr = -100000.0
        out["doa"][t], out["prob"][t], out["power"][t], out["energy_in_doa"][t] = cur, pr, power, E
    return out


# ---- FreqGCCBinauralLocalisation, whole stream incl. power gate and silence rule (BinauralLocalisation.cpp:387-631) ----
def freqgcc_set_probability(corr, doa, step):
    """setProbability :569-631 for one DOA."""
    D = len(corr)
    mn = corr.min()
    total = corr.sum() - mn * D
    a = float(f32(max(float(f32(doa)), -np.pi / 2)))         # angle2DOAidx, microhponeArrayHelpers.cpp:110-115:
    a = float(f32(min(a, np.pi / 2)))                        # clamps in double, stored back into the float argument
    idx = int((a + np.pi / 2) / float(step))
    angle = float(doaidx2angle(idx, step))
    if 0 < idx < D - 1:
        if angle > doa:
            pc, pd, nc, nd = corr[idx - 1], float(doaidx2angle(idx - 1, step)), corr[idx], angle
        else:
            pc, pd, nc, nd = corr[idx], angle, corr[idx + 1], float(doaidx2angle(idx + 1, step))
        pr = (nc - pc) / (nd - pd) * (doa - pd) + pc
    else:
        pr = corr[idx]
    pr = (pr - mn) / total if total > 0 else 0.0
    return 0.0 if pr < 0.01 else pr


def freqgcc_stream(fs, N, xs, pcm, step_deg=3.0, use_power_floor=True):
    """Per frame: fired, power, argmax, smoothed corr, smoothed DOA and prob (of the previous DOA, :454).

    The memory factors follow :523-524 (a frame fired -> maxima), :536-542 (fewer than windowsToDecay gated-out
    frames since -> maxima) and :556-557 (more -> zero): written here as a run length of gated-out frames."""
    xyz = xyz_of(xs)
    step = doa_step(step_deg)
    tau = delay_table(fs, xyz, step_deg)[0]
    D, K = len(tau), N // 2 + 1
    X = stft_frames(np.asarray(pcm, dtype=np.float64), N)
    F = X.shape[0]
    mu_max, dm_max = float(f32(0.8)), float(f32(0.6))
    windows_to_decay = (3 * fs) // (N // 2)
    out = dict(fired=np.zeros(F, bool), power=np.zeros(F), argmax=np.full(F, -1, np.int32), corr=np.zeros((F, D)),
               doa=np.zeros(F), prob=np.full(F, -1.0), restart=np.zeros(F, bool))
    prev, cur_doa, cur_prob = np.zeros(D), 0.0, -1.0
    acc, consumed, est, floor = 0.0, 0, False, 0.0
    mu, dm = 0.0, 0.0                   # :323-324
    gated_run = 0                       # gated-out frames since the estimate exists / since the last fired frame
    for t in range(F):
        if not est:
            acc += fft_power([X[t, 0], X[t, 1]], N + 2) * N + 1e-10
            consumed += N
            if consumed >= int(3 * fs):
                est = True
                floor = 10 * np.log10(acc / consumed) + 6.0
                acc = floor
            power = acc
            thr = acc
        else:
            power = 10 * np.log10(fft_power([X[t, 0], X[t, 1]], N + 2))
            thr = floor
        out["power"][t] = power
        if power > thr or not use_power_floor:
            c = float(f32(1) - f32(mu)) * gcc_phat(X[t, 0], X[t, 1], tau, K).real + mu * prev   # 1 - float factor: float arithmetic
            prev = c
            cur_prob = freqgcc_set_probability(c, cur_doa, step)
            i = int(np.argmax(c))
            cur_doa = dm * cur_doa + float(f32(1) - f32(dm)) * float(doaidx2angle(i, step))
            out["fired"][t], out["argmax"][t], out["restart"][t] = True, i, mu == 0.0
            mu, dm, gated_run = mu_max, dm_max, 0
        elif est:
            mu, dm = (mu_max, dm_max) if gated_run < windows_to_decay else (0.0, 0.0)
            gated_run += 1
        out["corr"][t], out["doa"][t], out["prob"][t] = prev, cur_doa, cur_prob
    return out


# ---- MVDR-style beamformer with a per-bin spatial covariance (no reference counterpart) ----
def mvdr_stream(fs, N, xs, pcm, doa_rad, alpha=0.95, loading=1e-3):
    """Independent of the C oracle: full-matrix numpy.linalg.solve instead of the Cholesky / forward-substitution form.
    pcm [M][(F+1)*hop]; doa_rad [F].  Returns dict(out [F*hop], spec [F][K] complex, phi [K][M][M])."""
    x = np.asarray(xs, dtype=np.float64)
    if x.ndim == 2:
        x = x[:, 0]
    M = len(x)
    hop, K = N // 2, N // 2 + 1
    X = stft_frames(pcm, N)                                   # complex [F][M][K]
    F = X.shape[0]
    doa = np.broadcast_to(np.asarray(doa_rad, dtype=np.float64), (F,))
    Phi = np.zeros((K, M, M), dtype=np.complex128)
    k = np.arange(K, dtype=np.float64)
    spec = np.zeros((F, K), dtype=np.complex128)
    out = np.zeros(F * hop)
    tail = np.zeros(hop)
    eye = np.eye(M)
    for t in range(F):
        Xc = X[t].T                                                   # [K][M]
        slope = 2 * np.pi * fs / N / C_SOUND * x * np.cos(doa[t] + np.pi / 2)   # Beamformer.cpp:59
        d = np.exp(-1j * k[:, None] * slope[None, :])                 # [K][M]: conj of the phasor X_m is multiplied by
        Phi = alpha * Phi + (1 - alpha) * Xc[:, :, None] * np.conj(Xc[:, None, :])
        tr = np.real(np.trace(Phi, axis1=1, axis2=2))
        for kk in range(K):
            if not tr[kk] > 1e-30:
                spec[t, kk] = np.vdot(d[kk], Xc[kk]) / M
                continue
            PL = Phi[kk] + loading * tr[kk] / M * eye
            g = np.linalg.solve(PL, d[kk])
            w = g / np.vdot(d[kk], g)
            spec[t, kk] = np.vdot(w, Xc[kk])
        y = irfft_ccs(spec[t], N)
        out[t * hop:(t + 1) * hop] = tail + y[:hop]
        tail = y[hop:]
    return dict(out=out, spec=spec, phi=Phi)
