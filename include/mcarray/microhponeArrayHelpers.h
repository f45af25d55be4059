// DOA <-> delay helpers (reference include/mcarray/microhponeArrayHelpers.h, sic; src/...Helpers.cpp:38-120).
// Host-side scalar functions in the reference's float/double sequence.  They only produce the tables the
// device consumes; the library computes the same chain internally (mcarray_amd/csrc/api.hip).
#ifndef MCA_HIP_MICROHPONEARRAYHELPERS_H
#define MCA_HIP_MICROHPONEARRAYHELPERS_H
#include <cmath>

#include "complex.h"
#include "mcadefs.h"

namespace mca {

inline double getSpeedOfSound() { return 346.1; }                       // :38-43
inline float doaToDelayFarField(float doa, float microDist, bool useDegrees = false)   // :46-67
{
    if (useDegrees) doa = static_cast<float>(doa * M_PI / 180);
    // [BUILD-DEFINES] sin evaluated in double (gcc-4.8 + <math.h>, the reference's CI toolchain)
    return static_cast<float>((static_cast<double>(microDist) * std::sin(static_cast<double>(doa))) / getSpeedOfSound());
}
inline float doaToDelayFarFieldSamples(float doa, float microDist, int sampleRate)      // :69-72
{
    return doaToDelayFarField(doa, microDist) * static_cast<float>(sampleRate);
}
inline float delayToDOA(float delay, float microDist)                  // :74-77
{
    return static_cast<float>(std::acos(static_cast<double>(delay) * getSpeedOfSound() / static_cast<double>(microDist)));
}
inline float delaySamplesToDOA(float delay, float microDist, float sampleRate)          // :79-83
{
    delay = delay / sampleRate;
    return delayToDOA(delay, microDist);
}
inline float maxFreqForSpatialAliasing(float microphoneDistance)       // :85-89 (float in, float out)
{
    return static_cast<float>(getSpeedOfSound() / (2 * microphoneDistance));
}
inline float toDegrees(float radians) { return static_cast<float>(radians * M_1_PI * 180); }
inline float toRadians(float degrees) { return static_cast<float>(degrees * M_PI / 180); }
inline SignalPtr toDegrees(SignalPtr radians, int length)               // :91-98 (allocates, like the reference)
{
    SignalPtr degrees(new BaseType[length]);
    for (int i = 0; i < length; ++i) degrees[i] = (180 / M_PI) * radians[i];
    return degrees;
}
// The reference declares `toRadiasn` (sic, microhponeArrayHelpers.h:46) and defines `toRadians(SignalPtr, int)` (:100-107),
// whose body scales its own uninitialised result into the ARGUMENT: no caller can link against the one or use the other.
// Both names are provided here with the evident meaning (degrees -> radians into a fresh array, the input untouched).
inline SignalPtr toRadians(SignalPtr degrees, int length)
{
    SignalPtr radians(new BaseType[length]);
    for (int i = 0; i < length; ++i) radians[i] = (M_PI / 180) * degrees[i];
    return radians;
}
inline SignalPtr toRadiasn(SignalPtr degrees, int length) { return toRadians(degrees, length); }
inline float angle2DOAidx(float angle, float doaStep)                   // :110-115
{
    angle = static_cast<float>(std::max(static_cast<double>(angle), -M_PI_2));
    angle = static_cast<float>(std::min(static_cast<double>(angle), M_PI_2));
    return static_cast<float>(static_cast<int>((angle + M_PI_2) / doaStep));
}
inline float doaIdx2angle(int idx, float doaStep)                       // :117-120
{
    const float prod = static_cast<float>(idx) * doaStep;
    return static_cast<float>(static_cast<double>(prod) - M_PI_2);
}

// 10 log10 of half the mean squared magnitude of left + right, which is also left in `mixed` (:122-139; unused by the
// reference's own modules).  length complex values per buffer.
inline double calculateBinauralPower(const Complex *left, const Complex *right, Complex *mixed, int length)
{
    double mean = 0;
    for (int i = 0; i < length; ++i) {
        mixed[i].re = left[i].re + right[i].re;
        mixed[i].im = left[i].im + right[i].im;
        const double magn = std::sqrt(mixed[i].re * mixed[i].re + mixed[i].im * mixed[i].im);   // wipp::magnitude, then wipp::sqr
        mean += magn * magn;
    }
    mean /= length;
    return 10 * std::log10(mean / 2);
}

}  // namespace mca
#endif
