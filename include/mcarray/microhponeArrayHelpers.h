// DOA <-> delay helpers (reference include/mcarray/microhponeArrayHelpers.h, sic; src/...Helpers.cpp:38-120).
// Host-side scalar functions in the reference's float/double sequence.  They only produce the tables the
// device consumes; the library computes the same chain internally (mcarray_amd/csrc/api.hip).
#ifndef MCA_HIP_MICROHPONEARRAYHELPERS_H
#define MCA_HIP_MICROHPONEARRAYHELPERS_H
#include <cmath>

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
inline float toDegrees(float radians) { return static_cast<float>(radians * M_1_PI * 180); }
inline float toRadians(float degrees) { return static_cast<float>(degrees * M_PI / 180); }
inline SignalPtr toDegrees(SignalPtr radians, int length)               // :91-98 (allocates, like the reference)
{
    SignalPtr degrees(new BaseType[length]);
    for (int i = 0; i < length; ++i) degrees[i] = (180 / M_PI) * radians[i];
    return degrees;
}
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

}  // namespace mca
#endif
