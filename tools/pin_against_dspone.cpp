// pin_against_dspone.cpp -- settles the one [INFERRED] assumption that decides the DOA index: the weighting inside
// dsp::GeneralisedCrossCorrelation::calculateCorrelationsForPrecomputedTauMatrix (DSPONE; call site
// /root/reference/src/mcarray/SteeringBeamforming.cpp:115-119).
//
// DSPONE is not vendored by the reference and is absent from this build's image, so this program DOES NOT BUILD HERE.  On a
// machine that has libdspone-dev / libwipp-dev (the reference's own dependencies, .travis.yml:21-22):
//
//     g++ -std=c++11 tools/pin_against_dspone.cpp -o pin_against_dspone -ldspone -lwipp
//     ./pin_against_dspone tests/golden/pin_gcc_fixture.bin
//
// It reads the committed fixture (tools/make_pin_fixture.py: one analysis frame of the golden stream ssl_reemc_d37 as CCS
// spectra, the per-pair delay tables of SteeringBeamforming::generateLookupTable, and the per-pair correlations this build's
// oracle computes under PHAT and under NONE), runs the SAME calls the reference makes --
//     dsp::GeneralisedCrossCorrelation gcc(K, ONESIDEDFFT);                          (SteeringBeamforming.cpp:84-85)
//     gcc.precomputeTauMatrix(delays, D, K, ONESIDEDFFT);                            (:87-88)
//     gcc.calculateCorrelationsForPrecomputedTauMatrix(A, B, out, K, D, ONESIDEDFFT) (:115-119), real part (:122)
// -- and prints, per microphone pair, the largest difference against both expectations (relative to the largest expected
// value), then the verdict.  Exit code 0: DSPONE matches PHAT (the default of mca_hip_config.gcc_weighting and of the oracle)
// within 1e-9; 1: it matches NONE (set gcc_weighting = MCA_HIP_GCC_NONE / mca_or_*_set_weighting(MCA_OR_GCC_NONE)); 2: neither
// (the sign of the exponent, the bin range or the zero guard differ -- SURVEY A.3 lists the alternatives; the per-delay table
// printed with -v shows which).
#include <dspone/algorithm/gralCrossCorrelation.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

static bool read_exact(FILE *f, void *dst, size_t bytes) { return std::fread(dst, 1, bytes, f) == bytes; }

int main(int argc, char **argv)
{
    const bool verbose = argc > 2 && std::strcmp(argv[2], "-v") == 0;
    if (argc < 2) { std::fprintf(stderr, "usage: %s tests/golden/pin_gcc_fixture.bin [-v]\n", argv[0]); return 3; }
    FILE *f = std::fopen(argv[1], "rb");
    if (!f) { std::perror(argv[1]); return 3; }
    char magic[8];
    int32_t hdr[5];
    if (!read_exact(f, magic, 8) || std::memcmp(magic, "MCAPIN1", 8) != 0 || !read_exact(f, hdr, sizeof(hdr))) { std::fprintf(stderr, "not a pin fixture\n"); return 3; }
    const int M = hdr[0], ccs = hdr[1], D = hdr[2], P = hdr[3], fs = hdr[4];
    const int K = ccs / 2;                                          // _complexFFTCCSLength (SteeringBeamforming.cpp:36)
    std::vector<double> frames((size_t)M * ccs), delays((size_t)P * D), phat((size_t)P * D), none((size_t)P * D);
    if (!read_exact(f, frames.data(), frames.size() * 8) || !read_exact(f, delays.data(), delays.size() * 8) ||
        !read_exact(f, phat.data(), phat.size() * 8) || !read_exact(f, none.data(), none.size() * 8)) { std::fprintf(stderr, "truncated fixture\n"); return 3; }
    std::fclose(f);
    std::printf("fixture: %d microphones, fs %d, CCS length %d (K = %d bins), %d steering angles, %d pairs\n", M, fs, ccs, K, D, P);

    double max_phat = 0, max_none = 0, scale_phat = 0, scale_none = 0;
    for (size_t i = 0; i < phat.size(); ++i) { scale_phat = std::fmax(scale_phat, std::fabs(phat[i])); scale_none = std::fmax(scale_none, std::fabs(none[i])); }
    std::vector<double> ccorr((size_t)2 * D);
    int p = 0;
    for (int i = 0; i < M; ++i)
        for (int j = i + 1; j < M; ++j, ++p) {                      // pair order of generateLookupTable (:62-64)
            dsp::GeneralisedCrossCorrelation gcc(K, dsp::GeneralisedCrossCorrelation::ONESIDEDFFT);
            gcc.precomputeTauMatrix(&delays[(size_t)p * D], D, K, dsp::GeneralisedCrossCorrelation::ONESIDEDFFT);
            dsp::Complex *A = reinterpret_cast<dsp::Complex *>(&frames[(size_t)i * ccs]);
            dsp::Complex *B = reinterpret_cast<dsp::Complex *>(&frames[(size_t)j * ccs]);
            gcc.calculateCorrelationsForPrecomputedTauMatrix(A, B, reinterpret_cast<dsp::Complex *>(ccorr.data()), K, D,
                                                             dsp::GeneralisedCrossCorrelation::ONESIDEDFFT);
            double dp = 0, dn = 0;
            for (int d = 0; d < D; ++d) {
                const double r = ccorr[2 * d];                      // wipp::real (:122)
                dp = std::fmax(dp, std::fabs(r - phat[(size_t)p * D + d]));
                dn = std::fmax(dn, std::fabs(r - none[(size_t)p * D + d]));
                if (verbose) std::printf("  pair (%d,%d) d %2d: dspone % .9e   phat % .9e   none % .9e\n", i, j, d, r, phat[(size_t)p * D + d], none[(size_t)p * D + d]);
            }
            std::printf("pair (%d,%d): max |dspone - phat| / max|phat| = %.3e    max |dspone - none| / max|none| = %.3e\n", i, j, dp / scale_phat, dn / scale_none);
            max_phat = std::fmax(max_phat, dp / scale_phat);
            max_none = std::fmax(max_none, dn / scale_none);
        }
    const double tol = 1e-9;
    if (max_phat <= tol) { std::printf("VERDICT: DSPONE's GCC is PHAT-weighted as the build assumes (MCA_HIP_GCC_PHAT): parity pinned for this stage.\n"); return 0; }
    if (max_none <= tol) { std::printf("VERDICT: DSPONE's GCC is UN-weighted: configure gcc_weighting = MCA_HIP_GCC_NONE (oracle: MCA_OR_GCC_NONE).\n"); return 1; }
    std::printf("VERDICT: neither reading matches (phat %.3e, none %.3e): see SURVEY A.3 for the other conventions; rerun with -v.\n", max_phat, max_none);
    return 2;
}
