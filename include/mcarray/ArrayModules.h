// The enums of mca::BinauralMasking that FastBinauralMasking needs (reference include/mcarray/ArrayModules.h:81,89).
// The facades themselves (SoundLocalisation, BinauralMasking) expose no process() and are out of scope.
#ifndef MCA_HIP_ARRAYMODULES_H
#define MCA_HIP_ARRAYMODULES_H
namespace mca {
class BinauralMasking {
public:
    typedef enum { FACTOR = 0, RELATIVE = 1, FULL = 3, NOISY = 4, NOTHING = 5 } MaskingMethod;
    typedef enum { BOTH = 0, SPATIAL = 1, TEMPORAL = 2 } MaskingAlg;
};
}  // namespace mca
#endif
