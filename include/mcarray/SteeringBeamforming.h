// mca::SteeringBeamforming -- SRP-PHAT DOA scan with the reference's peak picking, same constructor and
// processFrame signature as the reference (include/mcarray/SteeringBeamforming.h:43,54;
// src/mcarray/SteeringBeamforming.cpp:34-195).  State (_prevEnergyInDOA) lives in the GPU context.
// Extension over the reference: the DOA grid step is a constructor parameter (reference: 5 degrees, :39).
#ifndef MCA_HIP_STEERINGBEAMFORMING_H
#define MCA_HIP_STEERINGBEAMFORMING_H
#include <memory>
#include <vector>

#include "HipContext.h"
#include "mcadefs.h"
#include "microhponeArrayHelpers.h"

namespace mca {

class SteeringBeamforming {
public:
    SteeringBeamforming(int sampleRate, ArrayDescription microphonePositions, int fftCCSLength, unsigned int nchannels,
                        double doaStepDeg = 5.0)
        : fftCCSLength_(fftCCSLength), nchannels_(nchannels)
    {
        if (nchannels != microphonePositions.size()) throw MCArrayException("nchannels does not match the array description");
        ctx_.reset(new detail::HipContext(sampleRate, microphonePositions, fftCCSLength - 2, doaStepDeg, 4, false));
    }
    virtual ~SteeringBeamforming() {}

    // wienerCoefs is accepted and ignored, like the reference (SteeringBeamforming.cpp:114).
    // LIMIT of this build: numOfSources is 1 ... 4 (MCA_MAX_SOURCES, mcarray_amd/csrc/mca_internal.h: the per-frame pick buffers, the state
    // blob and the overlap-add carries of the stream kernels are sized by it); the reference's loop (SteeringBeamforming.cpp:185-194) takes
    // any count.  A larger value throws MCArrayException (MCA_HIP_ERR_INVALID_ARGUMENT from the C ABI) -- it is never truncated silently.
    void processFrame(const SignalVector &analysisFrames, SignalPtr DOA, SignalPtr prob, int numOfSources, SignalVector &wienerCoefs)
    {
        (void)wienerCoefs;
        std::vector<const double *> rows(nchannels_);
        for (unsigned c = 0; c < nchannels_; ++c) rows[c] = analysisFrames[c].get();
        ctx_->check(mca_hip_steering_process_frame(ctx_->get(), rows.data(), fftCCSLength_, DOA.get(), prob.get(), lastBins_, numOfSources));
    }
    int numSteps() const { return mca_hip_num_steps(ctx_->get()); }
    const int *lastDoaBins() const { return lastBins_; }   // maxIdx+1 of selectDOA for the last frame

private:
    int fftCCSLength_;
    unsigned nchannels_;
    int lastBins_[4] = {0, 0, 0, 0};
    std::shared_ptr<detail::HipContext> ctx_;
};

}  // namespace mca
#endif
