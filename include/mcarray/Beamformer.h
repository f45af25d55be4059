// mca::Beamformer -- frequency-domain delay-and-sum, same constructor and processFrame signature as the
// reference (include/mcarray/Beamformer.h:39,49; src/mcarray/Beamformer.cpp:33-71).  The arithmetic runs on the
// GPU in double (mca_hip_beamformer_process_frame, kernel k_frame_beamform<double>).
#ifndef MCA_HIP_BEAMFORMER_H
#define MCA_HIP_BEAMFORMER_H
#include <memory>
#include <vector>

#include "HipContext.h"
#include "mcadefs.h"

namespace mca {

class Beamformer {
public:
    Beamformer(int sampleRate, ArrayDescription microphonePositions, int fftCCSLength, unsigned int nchannels)
        : fftCCSLength_(fftCCSLength), nchannels_(nchannels)
    {
        if (nchannels != microphonePositions.size()) throw MCArrayException("nchannels does not match the array description");
        ctx_.reset(new detail::HipContext(sampleRate, microphonePositions, fftCCSLength - 2, 5.0, 1, false));
    }
    virtual ~Beamformer() {}

    void processFrame(SignalVector &inputAnalysisFrames, SignalPtr outputFrame, double DOA)
    {
        std::vector<const double *> rows(nchannels_);
        for (unsigned c = 0; c < nchannels_; ++c) rows[c] = inputAnalysisFrames[c].get();
        ctx_->check(mca_hip_beamformer_process_frame(ctx_->get(), rows.data(), fftCCSLength_, outputFrame.get(), DOA));
    }

private:
    int fftCCSLength_;
    unsigned nchannels_;
    std::shared_ptr<detail::HipContext> ctx_;
};

}  // namespace mca
#endif
