// mca::LocalisationCallback -- the synchronous upcall of the path
// (reference include/mcarray/SoundLocalisationCallback.h:37-54): doa in DEGREES, arrays owned by the module.
#ifndef MCA_HIP_SOUNDLOCALISATIONCALLBACK_H
#define MCA_HIP_SOUNDLOCALISATIONCALLBACK_H
#include "mcadefs.h"
namespace mca {
class LocalisationCallback {
public:
    LocalisationCallback() {}
    virtual ~LocalisationCallback() {}
    virtual void setDOA(SignalPtr doa, SignalPtr prob, double power, int numOfSources) = 0;
};
}  // namespace mca
#endif
