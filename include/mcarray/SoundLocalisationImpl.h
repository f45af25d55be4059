// mca::SoundLocalisationImpl -- base holding the callback pointer, current DOA/prob and the power-floor
// bookkeeping (reference include/mcarray/SoundLocalisationImpl.h:44-87).  The particle-filter members of the
// reference are out of scope.
#ifndef MCA_HIP_SOUNDLOCALISATIONIMPL_H
#define MCA_HIP_SOUNDLOCALISATIONIMPL_H
#include "ArrayDescription.h"
#include "SoundLocalisationCallback.h"
#include "mcadefs.h"

namespace mca {

class SoundLocalisationImpl {
public:
    explicit SoundLocalisationImpl(ArrayDescription microphonePositions)
        : _ptrCallback(nullptr), _microphonePositions(microphonePositions), _powerFloor(0), _noiseEstimated(false), _samplesConsumedForNoise(0) {}
    virtual ~SoundLocalisationImpl() {}
    void setCallback(LocalisationCallback &callback) { _ptrCallback = &callback; }
    void setCallback(LocalisationCallback *callback) { _ptrCallback = callback; }
    // default of the reference: uniform probability (SoundLocalisationImpl.cpp:55-58)
    virtual void setProbability(const double *, double *probs, int size) { for (int i = 0; i < size; ++i) probs[i] = 1.0 / size; }

protected:
    static constexpr double _durationToEstimatePowerFloor = 3;   // seconds (SoundLocalisationImpl.h:77)
    LocalisationCallback *_ptrCallback;
    const ArrayDescription _microphonePositions;
    SignalPtr _currentDOA;
    SignalPtr _prob;
    double _powerFloor;
    bool _noiseEstimated;
    int _samplesConsumedForNoise;
};

}  // namespace mca
#endif
