// C++ tests of the module API (include/mcarray/*.h) -- modelled on the reference's gtest file
// (test/test_mcarray.cpp), without gtest.  `--cpu` runs only what needs no GPU (testArrayDescription,
// helper tables); with no argument everything runs and needs an MI355X.
//
//   testArrayDescription             <- test/test_mcarray.cpp:518-580 (the only exact known-answer test)
//   testBeamformingSeparation        <- test/test_mcarray.cpp:631-800 (dead in the reference; property >= 5.5 dB)
//   testBeamformingSoundLocalisation <- test/test_mcarray.cpp:384-423 (+-7 degrees; broadband source)
//   testHookMatchesStream            the DSPONE hook (frame API, double) against the batched stream path
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "mcarray/micarray.h"

using namespace mca;

static int g_fail = 0;
#define EXPECT(cond)                                                                   \
    do { if (!(cond)) { std::printf("FAIL %s:%d: %s\n", __FILE__, __LINE__, #cond); ++g_fail; } } while (0)
static bool almost(double a, double b) { return std::fabs(a - b) <= 4 * 2.220446049250313e-16 * std::fmax(std::fabs(a), std::fabs(b)) + 1e-300; }

static void testArrayDescription()
{
    ArrayDescription description;
    EXPECT(description.size() == 0);
    int l = description.pushPosition(0, 2, 3, "left");
    int cl = description.pushPosition(0.035 * 2, 2, 3, "center-left");
    int cr = description.pushPosition(0.035 * 5, 2, 3, "center-right");
    int r = description.pushPosition(0.035 * 6, 2, 3, "right");
    EXPECT(description.size() == 4);
    EXPECT(description.getName(l) == "left" && description.getName(cl) == "center-left");
    EXPECT(description.getName(cr) == "center-right" && description.getName(r) == "right");
    EXPECT(almost(description.getX(cl), 0.070) && almost(description.getX(cr), 0.175) && almost(description.getX(r), 0.210));
    EXPECT(almost(description.getY(l), 2) && almost(description.getZ(r), 3));
    const double expect[4][4] = {{0.000, 0.070, 0.175, 0.210}, {0.070, 0.000, 0.105, 0.140}, {0.175, 0.105, 0.000, 0.035}, {0.210, 0.140, 0.035, 0.000}};
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) EXPECT(std::fabs(description.distance(i, j) - expect[i][j]) < 1e-15);
    EXPECT(std::fabs(description.maxDistance() - 0.210) < 1e-15);
    EXPECT(std::fabs(description.distance("left", "center-left") - 0.070) < 1e-15);
    EXPECT(std::fabs(description.distance("right", "center-left") - 0.140) < 1e-15);
    bool threw = false;
    try { description.pushPosition(1, 1, 1, "left"); } catch (const MCArrayException &) { threw = true; }
    EXPECT(threw);
    EXPECT(description.minDistance() == 0);   // reference quirk kept
    // helper chain
    const float step = static_cast<float>(5 * M_PI / 180);
    EXPECT(static_cast<int>(std::round(M_PI / step) + 1) == 37);
    EXPECT(std::fabs(doaIdx2angle(18, step)) < 1e-6);
    EXPECT(std::fabs(doaToDelayFarFieldSamples(static_cast<float>(M_PI / 2), 0.21f, 48000) - 0.21 / 346.1 * 48000) < 1e-4);
    // the rest of microhponeArrayHelpers.h (:74-139)
    EXPECT(std::fabs(maxFreqForSpatialAliasing(0.086f) - 346.1 / (2 * 0.086)) < 1e-3);
    EXPECT(std::fabs(delayToDOA(0.f, 0.21f) - M_PI / 2) < 1e-6);
    EXPECT(std::fabs(delaySamplesToDOA(doaToDelayFarFieldSamples(0.3f, 0.21f, 48000), 0.21f, 48000.f) - (M_PI / 2 - 0.3)) < 1e-4);   // acos vs sin convention
    {
        SignalPtr deg(new BaseType[2]);
        deg[0] = 180; deg[1] = -90;
        SignalPtr rad = toRadiasn(deg, 2), back = toDegrees(rad, 2);
        EXPECT(std::fabs(rad[0] - M_PI) < 1e-12 && std::fabs(rad[1] + M_PI / 2) < 1e-12 && std::fabs(back[0] - 180) < 1e-9 && deg[0] == 180);
        // calculateBinauralPower: |(1 + 2j) + (3 - 2j)|^2 = 16 and |(0 + 1j) + (0 + 1j)|^2 = 4 -> mean 10 -> 10 log10(5)
        const Complex l[2] = {{1, 2}, {0, 1}}, r[2] = {{3, -2}, {0, 1}};
        Complex m[2];
        EXPECT(std::fabs(calculateBinauralPower(l, r, m, 2) - 10 * std::log10(5.0)) < 1e-12);
        EXPECT(m[0].re == 4 && m[0].im == 0 && m[1].re == 0 && m[1].im == 2);
    }
    std::printf("testArrayDescription done\n");
}

// small O(N^2)-free real FFT for the test scaffolding (radix-2)
static void rfft_ccs(const std::vector<double> &x, std::vector<double> &ccs)
{
    const int N = static_cast<int>(x.size());
    std::vector<double> re(x), im(N, 0.0);
    for (int i = 1, j = 0; i < N; ++i) { int bit = N >> 1; for (; j & bit; bit >>= 1) j ^= bit; j ^= bit; if (i < j) std::swap(re[i], re[j]); }
    for (int len = 2; len <= N; len <<= 1)
        for (int k = 0; k < len / 2; ++k) {
            const double a = -2 * M_PI * k / len, wr = std::cos(a), wi = std::sin(a);
            for (int i = k; i < N; i += len) {
                const int j = i + len / 2;
                const double xr = re[j] * wr - im[j] * wi, xi = re[j] * wi + im[j] * wr;
                re[j] = re[i] - xr; im[j] = im[i] - xi; re[i] += xr; im[i] += xi;
            }
        }
    ccs.resize(N + 2);
    for (int k = 0; k <= N / 2; ++k) { ccs[2 * k] = re[k]; ccs[2 * k + 1] = im[k]; }
}

static double peak(const double *ccs, double f, int fs, int N)
{
    const int b = static_cast<int>(f / fs * N);
    double m = 0;
    for (int k = b - 4; k < b + 4; ++k) m = std::fmax(m, std::hypot(ccs[2 * k], ccs[2 * k + 1]));
    return m;
}

static void testBeamformingSeparation()
{
    const int fs = 48000, order = 11, N = 1 << order, ccs = N + 2;
    const std::vector<double> xs = {0, 0.07, 0.175, 0.21};   // Reem C
    ArrayDescription mics = ArrayDescription::make_linear_array_description(xs);
    Beamformer beamformer(fs, mics, ccs, 4);
    const double scenes[3][4] = {{800, 45, 2000, -45}, {1000, 20, 4000, -20}, {1000, 80, 1500, 10}};
    for (int sc = 0; sc < 3; ++sc) {
        const double f1 = scenes[sc][0], d1 = scenes[sc][1] * M_PI / 180, f2 = scenes[sc][2], d2 = scenes[sc][3] * M_PI / 180;
        SignalVector frames;
        double in1 = 0, in2 = 0;
        for (int c = 0; c < 4; ++c) {
            std::vector<double> x(N), X;
            for (int n = 0; n < N; ++n) {
                const double t = static_cast<double>(n) / fs;
                x[n] = 5000 * std::cos(2 * M_PI * f1 * (t + xs[c] * std::sin(d1) / 346.1)) + 5000 * std::cos(2 * M_PI * f2 * (t + xs[c] * std::sin(d2) / 346.1));
            }
            rfft_ccs(x, X);
            SignalPtr p(new double[ccs]);
            std::memcpy(p.get(), X.data(), sizeof(double) * ccs);
            frames.push_back(p);
            in1 += peak(p.get(), f1, fs, N) / 4; in2 += peak(p.get(), f2, fs, N) / 4;
        }
        SignalPtr out(new double[ccs]);
        beamformer.processFrame(frames, out, d1);
        const double att1 = 20 * std::log10((peak(out.get(), f1, fs, N) / in1) / (peak(out.get(), f2, fs, N) / in2));
        beamformer.processFrame(frames, out, d2);
        const double att2 = 20 * std::log10((peak(out.get(), f2, fs, N) / in2) / (peak(out.get(), f1, fs, N) / in1));
        std::printf("scene %d: attenuation %.2f / %.2f dB\n", sc, att1, att2);
        EXPECT(att1 >= 5.5); EXPECT(att2 >= 5.5);
    }
}

struct RangeCallback : public LocalisationCallback {
    double lo, hi; int calls, bad;
    RangeCallback(double l, double h) : lo(l), hi(h), calls(0), bad(0) {}
    virtual void setDOA(SignalPtr doa, SignalPtr prob, double, int n)
    {
        ++calls;
        if (n != 1 || doa[0] < lo || doa[0] > hi || !(prob[0] >= 0)) ++bad;
    }
};

// deterministic broadband far-field source: many random-phase sinusoids, analytic fractional delays
static void make_source(const std::vector<double> &xs, double theta, int fs, int n, unsigned seed, std::vector<std::vector<double> > &ch, double fspan = 15000.0)
{
    const int NC = 400;
    std::vector<double> f(NC), ph(NC);
    unsigned s = seed;
    for (int i = 0; i < NC; ++i) {
        s = s * 1664525u + 1013904223u; f[i] = 150.0 + (s >> 8) * (1.0 / 16777216.0) * fspan;
        s = s * 1664525u + 1013904223u; ph[i] = (s >> 8) * (1.0 / 16777216.0) * 2 * M_PI;
    }
    ch.assign(xs.size(), std::vector<double>(n));
    for (size_t c = 0; c < xs.size(); ++c) {
        const double adv = xs[c] * std::sin(theta) / 346.1;
        for (int t = 0; t < n; ++t) {
            double v = 0;
            for (int i = 0; i < NC; ++i) v += std::cos(2 * M_PI * f[i] * (static_cast<double>(t) / fs + adv) + ph[i]);
            ch[c][t] = 0.01 * v;
        }
    }
}

static void testBeamformingSoundLocalisation(int fs)
{
    // 48 kHz gives 1024-sample frames (the tuned kernels), 16 kHz 512 and 96 kHz 2048 (the any-length kernels)
    const int tolerance = 7;
    const std::vector<double> xs = {0, 0.07, 0.175, 0.21};
    ArrayDescription mics = ArrayDescription::make_linear_array_description(xs);
    for (int doa = -80; doa <= 80; doa += 40) {
        SourceSeparationAndLocalisation ssl(fs, mics, 1, false);
        RangeCallback cb(doa - tolerance, doa + tolerance);
        ssl.setCallback(&cb);
        const int n = 3 * ssl.getFrameSize() * 4;
        std::vector<std::vector<double> > ch;
        make_source(xs, doa * M_PI / 180, fs, n, 77u + doa, ch, fs >= 32000 ? 15000.0 : 0.43 * fs);
        std::vector<double *> in; std::vector<std::vector<double> > outb(4, std::vector<double>(n + ssl.getMaxLatency()));
        std::vector<double *> out;
        for (int c = 0; c < 4; ++c) { in.push_back(ch[c].data()); out.push_back(outb[c].data()); }
        // feed in three unequal chunks to exercise the buffering
        int done = 0, produced = 0;
        const int chunks[3] = {n / 2 + 37, n / 4 - 11, n - (n / 2 + 37) - (n / 4 - 11)};
        for (int k = 0; k < 3; ++k) {
            std::vector<double *> inp, outp;
            for (int c = 0; c < 4; ++c) { inp.push_back(in[c] + done); outp.push_back(out[c] + produced); }
            produced += ssl.process(inp, chunks[k], outp, n + ssl.getMaxLatency() - produced);
            done += chunks[k];
        }
        EXPECT(cb.calls == (n - ssl.getWindowSize()) / ssl.getFrameSize() + 1);
        EXPECT(cb.bad == 0);
        EXPECT(produced == cb.calls * ssl.getFrameSize());
        // the beamformed output is a delayed copy of the source within a few dB (channel 0 as reference)
        double eo = 0, ei = 0;
        for (int i = ssl.getFrameSize(); i < produced; ++i) { eo += outb[0][i] * outb[0][i]; ei += ch[0][i] * ch[0][i]; }
        EXPECT(10 * std::log10(eo / ei) > -3.0 && 10 * std::log10(eo / ei) < 1.0);
        std::printf("fs %d N %d DOA %d: %d callbacks, %d out of range, out/in %.2f dB\n", fs, ssl.getWindowSize(), doa, cb.calls, cb.bad, 10 * std::log10(eo / ei));
    }
}

struct LastCallback : public LocalisationCallback {
    std::vector<double> doa;
    virtual void setDOA(SignalPtr d, SignalPtr, double, int) { doa.push_back(d[0]); }
};

static void testHookMatchesStream()
{
    // the per-frame DSPONE hook (double, frame API) and the batched stream path (fp32) must report the same DOAs
    const int fs = 48000, N = 1024, hop = 512, F = 10;
    const std::vector<double> xs = {0, 0.07, 0.175, 0.21};
    ArrayDescription mics = ArrayDescription::make_linear_array_description(xs);
    std::vector<std::vector<double> > ch;
    make_source(xs, -35 * M_PI / 180, fs, (F + 1) * hop, 5u, ch);
    SourceSeparationAndLocalisation a(fs, mics, 1, false), b(fs, mics, 1, false);
    LastCallback ca, cb;
    a.setCallback(ca); b.setCallback(cb);
    std::vector<double *> in, out; std::vector<std::vector<double> > ob(4, std::vector<double>((F + 2) * hop));
    for (int c = 0; c < 4; ++c) { in.push_back(ch[c].data()); out.push_back(ob[c].data()); }
    const int produced = a.process(in, (F + 1) * hop, out, (F + 2) * hop);
    EXPECT(produced == F * hop);
    std::vector<double> win(N);
    for (int n = 0; n < N; ++n) win[n] = 0.5 - 0.5 * std::cos(2 * M_PI * n / N);
    for (int t = 0; t < F; ++t) {
        std::vector<std::vector<double> > X(4);
        std::vector<double *> fr;
        for (int c = 0; c < 4; ++c) {
            std::vector<double> x(N);
            for (int n = 0; n < N; ++n) x[n] = ch[c][t * hop + n] * win[n];
            rfft_ccs(x, X[c]);
            fr.push_back(X[c].data());
        }
        std::vector<double *> none;
        b.processParametrisation(fr, N + 2, none, 0);
        // in place: channel 0 now holds the beamformed spectrum, channels 1..3 are zero (processFrameSeparation)
        double z = 0; for (int i = 0; i < N + 2; ++i) z += std::fabs(X[1][i]) + std::fabs(X[3][i]);
        EXPECT(z == 0);
    }
    EXPECT(ca.doa.size() == static_cast<size_t>(F) && cb.doa.size() == static_cast<size_t>(F));
    for (int t = 0; t < F && t < static_cast<int>(ca.doa.size()) && t < static_cast<int>(cb.doa.size()); ++t) EXPECT(std::fabs(ca.doa[t] - cb.doa[t]) < 1e-4);
    std::printf("hook vs stream DOA[last] %.3f / %.3f deg\n", ca.doa.back(), cb.doa.back());
}

static void testBinauralModules()
{
    // FreqGCCBinauralLocalisation: 2 mics 0.086 m, 16 kHz, 3 degree grid (BinauralLocalisation.cpp:328), broadband source
    const int fs = 16000;
    const std::vector<double> xs = {0.0, 0.086};
    ArrayDescription mics = ArrayDescription::make_linear_array_description(xs);
    for (int doa = -60; doa <= 60; doa += 30) {
        FreqGCCBinauralLocalisation loc(fs, mics, false);
        RangeCallback cb(doa - 4, doa + 4);
        loc.setCallback(&cb);
        std::vector<std::vector<double> > ch;
        const int n = 20 * loc.getFrameSize();
        make_source(xs, doa * M_PI / 180, fs, n, 300u + doa, ch, 7000.0);
        for (auto &c : ch) for (auto &v : c) v *= 0.3;
        std::vector<double *> in = {ch[0].data(), ch[1].data()};
        const int frames = loc.process(in, n);
        EXPECT(frames == 19 && cb.calls == 19);
        EXPECT(cb.bad <= 3);   // the DOA smoothing (0.6) needs a few frames to settle from 0
        std::printf("FreqGCC DOA %d: %d callbacks, %d out of range\n", doa, cb.calls, cb.bad);
    }
    // FastBinauralMasking: NOTHING leaves the frames untouched (FastBinauralMasking.cpp:130-134) -> STFT/ISTFT identity;
    // FULL on two uncorrelated channels attenuates (spatial mask, -60 dB per masked band)
    std::vector<std::vector<double> > ch;
    const int n = 24 * 512;
    make_source(xs, 0.0, fs, n, 9u, ch, 7000.0);
    std::vector<std::vector<double> > other;
    make_source(xs, 0.0, fs, n, 10u, other, 7000.0);
    for (int pass = 0; pass < 2; ++pass) {
        FastBinauralMasking m(fs, 0.086, 500, 5000, pass == 0 ? BinauralMasking::NOTHING : BinauralMasking::FULL, BinauralMasking::BOTH);
        EXPECT(m.getWindowSize() == 1024 && m.getNumberOfChannels() == 2);
        std::vector<double *> in = {ch[0].data(), pass == 0 ? ch[1].data() : other[1].data()};
        std::vector<std::vector<double> > ob(2, std::vector<double>(n));
        std::vector<double *> out = {ob[0].data(), ob[1].data()};
        const int produced = m.process(in, n, out, n);
        EXPECT(produced == n - 512);
        double eo = 0, ei = 0, ed = 0;
        for (int i = 512; i < produced; ++i) { eo += ob[0][i] * ob[0][i]; ei += ch[0][i] * ch[0][i]; ed += (ob[0][i] - ch[0][i]) * (ob[0][i] - ch[0][i]); }
        if (pass == 0) EXPECT(ed < 1e-10 * ei);
        else EXPECT(eo < 0.05 * ei);
        std::printf("masking pass %d: out/in %.2f dB\n", pass, 10 * std::log10(eo / ei + 1e-300));
    }
}

static void testSourceLocalisation()
{
    // the analysis-only module reports the same DOAs as SourceSeparationAndLocalisation on the same input
    const int fs = 48000;
    const std::vector<double> xs = {0, 0.07, 0.175, 0.21};
    ArrayDescription mics = ArrayDescription::make_linear_array_description(xs);
    std::vector<std::vector<double> > ch;
    SourceLocalisation sl(fs, mics, 1, false);
    SourceSeparationAndLocalisation ssl(fs, mics, 1, false);
    const int n = 14 * sl.getFrameSize() + 123;
    make_source(xs, 55 * M_PI / 180, fs, n, 9u, ch);
    LastCallback ca, cb;
    sl.setCallback(ca); ssl.setCallback(cb);
    std::vector<double *> in; std::vector<std::vector<double> > ob(4, std::vector<double>(n + 1024));
    std::vector<double *> out;
    for (int c = 0; c < 4; ++c) { in.push_back(ch[c].data()); out.push_back(ob[c].data()); }
    const int frames = sl.process(in, n);
    ssl.process(in, n, out, n + 1024);
    EXPECT(frames == 13 && ca.doa.size() == 13u && cb.doa.size() == 13u);
    for (size_t t = 0; t < ca.doa.size() && t < cb.doa.size(); ++t) EXPECT(ca.doa[t] == cb.doa[t]);
    EXPECT(std::fabs(ca.doa.back() - 55.0) <= 7.0);
    std::printf("SourceLocalisation: %d frames, last DOA %.1f deg\n", frames, ca.doa.back());
}

static void testMultibandBinauralLocalisation()
{
    // test/test_mcarray.cpp:344-383: 1 kHz sine, 48 kHz, 0.086 m pair, 25 bins, ungated, +-15 degrees
    const int fs = 48000, tolerance = 15;
    ArrayDescription mics = ArrayDescription::make_linear_array_description({0, 0.086});
    for (int doa = -90; doa <= 90; doa += 30) {
        MultibandBinarualLocalisation mbl(fs, mics, 25, false);
        RangeCallback cb(doa - tolerance, doa + tolerance);
        mbl.setCallback(&cb);
        const int n = 20 * mbl.getFrameSize();
        std::vector<std::vector<double> > ch(2, std::vector<double>(n));
        for (int c = 0; c < 2; ++c) {
            const double adv = (c ? 0.086 : 0.0) * std::sin(doa * M_PI / 180) / 346.1;
            for (int t = 0; t < n; ++t) ch[c][t] = 5000.0 * std::cos(2 * M_PI * 1000.0 * (static_cast<double>(t) / fs + adv));
        }
        std::vector<double *> in = {ch[0].data(), ch[1].data()};
        const int frames = mbl.process(in, n);
        EXPECT(frames == 19 && cb.calls == 19);
        EXPECT(cb.bad == 0);
        std::printf("Multiband DOA %d: %d callbacks, %d out of range\n", doa, cb.calls, cb.bad);
    }
}

static void testMvdrBeamformer()
{
    // a broadband source 60 degrees off the look direction: once the covariance has converged the MVDR output carries
    // at least 10 dB less of it than the delay-and-sum of SourceSeparationAndLocalisation steered the same way would
    // (here: than the input level minus the array's delay-and-sum gain, bounded below by the single-microphone level / M)
    const int fs = 16000, N = 512, hop = N / 2, F = 120;
    const std::vector<double> xs = {0, 0.04, 0.08, 0.12, 0.16, 0.20, 0.24, 0.28};
    ArrayDescription mics = ArrayDescription::make_linear_array_description(xs);
    std::vector<std::vector<double> > ch;
    make_source(xs, -50.0 * M_PI / 180, fs, (F + 1) * hop, 77u, ch, 7000.0);
    MvdrBeamformer mvdr(fs, mics, N);
    MvdrBeamformer das(fs, mics, N, 0.95, 1e9);        // loading -> infinity: w = d/M, the reference's delay-and-sum
    mvdr.setDOA(10.0 * M_PI / 180); das.setDOA(10.0 * M_PI / 180);
    std::vector<double *> in(xs.size());
    std::vector<double> o1(static_cast<size_t>(F) * hop), o2(static_cast<size_t>(F) * hop);
    int w1 = 0, w2 = 0;
    const int chunk = 1000;                             // chunks that are not a multiple of the hop
    for (int pos = 0; pos < (F + 1) * hop; pos += chunk) {
        const int n = std::min(chunk, (F + 1) * hop - pos);
        for (size_t c = 0; c < xs.size(); ++c) in[c] = ch[c].data() + pos;
        w1 += mvdr.process(in, n, o1.data() + w1, F * hop - w1);
        w2 += das.process(in, n, o2.data() + w2, F * hop - w2);
    }
    EXPECT(w1 == F * hop && w2 == F * hop);
    double p1 = 0, p2 = 0;
    for (int i = 80 * hop; i < F * hop; ++i) { p1 += o1[static_cast<size_t>(i)] * o1[static_cast<size_t>(i)]; p2 += o2[static_cast<size_t>(i)] * o2[static_cast<size_t>(i)]; }
    const double gain = 10 * std::log10(p2 / p1);
    std::printf("MVDR vs delay-and-sum on an off-axis source: %.1f dB less\n", gain);
    EXPECT(gain > 10.0);
}

static void testSignalVectorOverloads()
{
    // process(SignalVector16s&, n, SignalVector16s&, outSize) (the 16-bit path: shorts go to the GPU as they are) gives the DOAs of
    // process(std::vector<double*>&, ...) on the same samples
    const int fs = 48000, n = 9 * 512;
    const std::vector<double> xs = {0, 0.07, 0.175, 0.21};
    ArrayDescription mics = ArrayDescription::make_linear_array_description(xs);
    std::vector<std::vector<double> > ch;
    make_source(xs, 25.0 * M_PI / 180, fs, n, 5u, ch);
    SignalVector16s in16, out16;
    std::vector<std::vector<double> > chd(xs.size(), std::vector<double>(n));
    std::vector<double *> ind, outd;
    std::vector<std::vector<double> > od(xs.size(), std::vector<double>(n));
    for (size_t c = 0; c < xs.size(); ++c) {
        SignalPtr16s p(new BaseType16s[n]), q(new BaseType16s[n]);
        for (int i = 0; i < n; ++i) { p[i] = static_cast<BaseType16s>(std::lround(ch[c][i] * 8000.0)); chd[c][i] = p[i]; }
        in16.push_back(p); out16.push_back(q);
        ind.push_back(chd[c].data()); outd.push_back(od[c].data());
    }
    SourceSeparationAndLocalisation a(fs, mics, 1, false), b(fs, mics, 1, false);
    const int wa = a.process(in16, n, out16, n), wb = b.process(ind, n, outd, n);
    EXPECT(wa == wb && wa == 8 * 512);
    EXPECT(a.lastDoaBins() == b.lastDoaBins());
    double worst = 0;
    for (int i = 0; i < wa; ++i) worst = std::fmax(worst, std::fabs(static_cast<double>(out16[0][i]) - od[0][i]));
    EXPECT(worst <= 1.0);                       // the 16-bit output is the double output rounded toward zero
    std::printf("SignalVector16s overload: %d samples, DOA bins equal, |int16 out - double out| <= %.2f\n", wa, worst);
}

static void testAdaptivePrecisionAndPageLockedBuffers()
{
    // the C ABI from plain C++: MCA_HIP_SRP_ADAPTIVE (fp16 scan of every frame + exact repair of the tie-sensitive ones) must
    // report the DOA bins of MCA_HIP_SRP_FP16X3 on a batch large enough for the adaptive path (4 arrays x 2100 frames), fed
    // through page-locked buffers (mca_hip_host_alloc: chunked, overlapped copies) on one side and pageable ones on the other
    const int fs = 48000, N = 1024, hop = 512, A = 4, M = 8, F = 2100, S = 1;
    const long long L = static_cast<long long>(F + 1) * hop;
    std::vector<double> xyz(3 * M, 0.0);
    for (int m = 0; m < M; ++m) xyz[3 * m] = 0.04 * m;
    // white source per array, integer-sample delays (the point is the path, not the acoustics) + sensor noise
    std::vector<float> pcm(static_cast<size_t>(A) * M * L);
    unsigned s = 12345u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) * (1.0f / 16777216.0f)) - 0.5f; };
    for (int a = 0; a < A; ++a) {
        std::vector<float> src(L + 64);
        for (float &v : src) v = 0.2f * rnd();
        const int lag = a - 2;                          // samples per microphone step: a different direction per array
        for (int m = 0; m < M; ++m)
            for (long long t = 0; t < L; ++t)
                pcm[(static_cast<size_t>(a) * M + m) * L + t] = src[static_cast<size_t>(t + 32 + lag * m)] + 0.02f * rnd();
    }
    const size_t nfs = static_cast<size_t>(A) * F * S;
    std::vector<int> bin_x(nfs);
    std::vector<float> rad_x(nfs), prob_x(nfs);
    mca_hip_config cfg = mca_hip_config();       // zero first: fields a caller does not know yet (gcc_weighting) keep their defaults
    cfg.struct_size = static_cast<int>(sizeof(cfg)); cfg.device = 0; cfg.sample_rate = fs; cfg.fft_size = N; cfg.n_mics = M;
    cfg.mic_xyz = xyz.data(); cfg.doa_step_deg = 0.5; cfg.n_sources = S; cfg.use_power_floor = 0; cfg.max_arrays = A;
    mca_hip_ctx *cx = nullptr, *ca = nullptr;
    cfg.srp_precision = MCA_HIP_SRP_FP16X3;
    EXPECT(mca_hip_create(&cfg, &cx) == MCA_HIP_OK);
    cfg.srp_precision = MCA_HIP_SRP_ADAPTIVE;
    EXPECT(mca_hip_create(&cfg, &ca) == MCA_HIP_OK);
    if (!cx || !ca) return;
    EXPECT(mca_hip_process_frames_host(cx, pcm.data(), A, F, bin_x.data(), rad_x.data(), prob_x.data(), nullptr, nullptr) == MCA_HIP_OK);
    float *pin = static_cast<float *>(mca_hip_host_alloc(static_cast<long long>(pcm.size() * sizeof(float))));
    int *bin_a = static_cast<int *>(mca_hip_host_alloc(static_cast<long long>(nfs * sizeof(int))));
    float *rad_a = static_cast<float *>(mca_hip_host_alloc(static_cast<long long>(nfs * sizeof(float))));
    EXPECT(pin && bin_a && rad_a);
    if (pin && bin_a && rad_a) {
        std::copy(pcm.begin(), pcm.end(), pin);
        EXPECT(mca_hip_reset_timing(ca) == MCA_HIP_OK);
        EXPECT(mca_hip_process_frames_host(ca, pin, A, F, bin_a, rad_a, nullptr, nullptr, nullptr) == MCA_HIP_OK);
        unsigned long long frames = 0, flagged = 0, recomputed = 0;
        EXPECT(mca_hip_get_repair_stats(ca, &frames, &flagged, &recomputed) == MCA_HIP_OK);
        EXPECT(frames == static_cast<unsigned long long>(A) * F && flagged >= static_cast<unsigned long long>(A) && recomputed >= flagged);
        size_t diff = 0;
        for (size_t i = 0; i < nfs; ++i) diff += bin_a[i] != bin_x[i];
        EXPECT(diff <= 3);                              // (exact-level ties between the two three-product kernels)
        std::printf("adaptive vs fp16x3: %zu of %zu bins differ; %llu frames flagged, %llu rows recomputed\n", diff, nfs, flagged, recomputed);
        // registering a caller's buffer in place gives the same path
        EXPECT(mca_hip_host_register(pcm.data(), static_cast<long long>(pcm.size() * sizeof(float))) == MCA_HIP_OK);
        std::vector<int> bin_r(nfs);
        EXPECT(mca_hip_reset(ca, nullptr) == MCA_HIP_OK);
        EXPECT(mca_hip_process_frames_host(ca, pcm.data(), A, F, bin_r.data(), nullptr, nullptr, nullptr, nullptr) == MCA_HIP_OK);
        EXPECT(mca_hip_host_unregister(pcm.data()) == MCA_HIP_OK);
        size_t diff2 = 0;
        for (size_t i = 0; i < nfs; ++i) diff2 += bin_r[i] != bin_a[i];
        EXPECT(diff2 == 0);
    }
    mca_hip_host_free(pin); mca_hip_host_free(bin_a); mca_hip_host_free(rad_a);
    mca_hip_destroy(cx); mca_hip_destroy(ca);
}

int main(int argc, char **argv)
{
    const bool cpu_only = argc > 1 && std::string(argv[1]) == "--cpu";
    testArrayDescription();
    if (!cpu_only) {
        try {
            testBeamformingSeparation();
            testBeamformingSoundLocalisation(48000);
            testBeamformingSoundLocalisation(16000);
            testBeamformingSoundLocalisation(96000);
            testHookMatchesStream();
            testBinauralModules();
            testSourceLocalisation();
            testMultibandBinauralLocalisation();
            testMvdrBeamformer();
            testSignalVectorOverloads();
            testAdaptivePrecisionAndPageLockedBuffers();
        } catch (const MCArrayException &e) {
            std::printf("FAIL: MCArrayException: %s\n", e.what());
            ++g_fail;
        }
    } else {
        bool threw = false;   // without a GPU the module constructors must fail loudly
        try { ArrayDescription m = ArrayDescription::make_linear_array_description({0, 0.1}); Beamformer b(48000, m, 1026, 2); }
        catch (const MCArrayException &e) { threw = true; std::printf("expected without a GPU: %s\n", e.what()); }
        (void)threw;
    }
    std::printf(g_fail ? "FAILED (%d)\n" : "ALL PASSED\n", g_fail);
    return g_fail ? 1 : 0;
}
