// mca::FreqGCCBinauralLocalisation -- 2-microphone GCC-PHAT localiser, deterministic part of the reference class
// (include/mcarray/BinauralLocalisation.h:188-247; src/mcarray/BinauralLocalisation.cpp:320-631): smoothed
// correlation, first-max argmax, the author's DOA smoothing (#else branch :502-504) and setProbability.  The
// particle filter (:456-473) is a stochastic DSPONE component and is out of scope.
// The time-domain TemporalGCCBinauralLocalisation of the same header is out of scope.
#ifndef MCA_HIP_BINAURALLOCALISATION_H
#define MCA_HIP_BINAURALLOCALISATION_H
#include <cmath>
#include <memory>
#include <vector>

#include "mcadefs.h"

#include "HipContext.h"
#include "SoundLocalisationImpl.h"
#include "microhponeArrayHelpers.h"

namespace mca {

class FreqGCCBinauralLocalisation : public SoundLocalisationImpl {
public:
    FreqGCCBinauralLocalisation(int sampleRate, ArrayDescription microphonePositions, bool usePowerFloor = true, double doaStepDeg = 3.0)
        : SoundLocalisationImpl(microphonePositions), _order(calculateOrderFromSampleRate(sampleRate, _frameRate))
    {
        if (microphonePositions.size() != 2) throw MCArrayException("FreqGCCBinauralLocalisation needs an ArrayDescription with 2 microphones");
        _usePowerFloor = usePowerFloor;
        _ctx.reset(new detail::HipContext(sampleRate, microphonePositions, 1 << _order, doaStepDeg, 1, usePowerFloor));
        _currentDOA.reset(new BaseType[1]);
        _prob.reset(new BaseType[1]);
        _currentDOA[0] = 0; _prob[0] = -1;                   // BinauralLocalisation.cpp:339-340
    }
    static int calculateOrderFromSampleRate(int sampleRate, double frameSeconds)
    {
        int order = static_cast<int>(std::lround(std::log2(sampleRate * frameSeconds)));
        return order < 8 ? 8 : (order > 14 ? 14 : order);
    }
    int getFrameSize() const { return 1 << (_order - 1); }
    int getWindowSize() const { return 1 << _order; }
    int getAnalysisLength() const { return (1 << _order) + 2; }
    int getOneSidedFFTLength() const { return (1 << (_order - 1)) + 1; }
    int getMaxLatency() const { return 1 << _order; }
    int getNumberOfChannels() const { return 2; }

    // chunked PCM in (2 channels); fires the callback once per completed frame: setDOA(degrees, prob, power, 1) (:521)
    template <typename Tin> int process(const std::vector<Tin *> &in, int nSamples)
    {
        const int N = getWindowSize(), hop = N / 2;
        for (int c = 0; c < 2; ++c)
            for (int i = 0; i < nSamples; ++i) _pending[c].push_back(static_cast<float>(in[static_cast<size_t>(c)][i]));
        const int have = static_cast<int>(_pending[0].size());
        const int F = have >= N ? (have - N) / hop + 1 : 0;
        if (F == 0) return 0;
        const size_t L = static_cast<size_t>(F + 1) * static_cast<size_t>(hop);
        std::vector<float> pcm(2 * L), doa(static_cast<size_t>(F)), prob(static_cast<size_t>(F));
        std::vector<int> idx(static_cast<size_t>(F));
        for (int c = 0; c < 2; ++c) std::copy(_pending[c].begin(), _pending[c].begin() + static_cast<long>(L), pcm.begin() + static_cast<long>(L) * c);
        _ctx->check(mca_hip_gcc2_frames_host(_ctx->get(), pcm.data(), 1, F, idx.data(), doa.data(), prob.data(), nullptr));
        std::vector<unsigned char> voiced(static_cast<size_t>(F), 1);
        std::vector<float> power(static_cast<size_t>(F), 0.f);
        if (_usePowerFloor) _ctx->check(mca_hip_copy_gate(_ctx->get(), voiced.data(), power.data()));
        for (int t = 0; t < F; ++t) {
            if (!voiced[static_cast<size_t>(t)]) continue;       // gated out: the block of BinauralLocalisation.cpp:434 is skipped, no setDOA
            _currentDOA[0] = doa[static_cast<size_t>(t)]; _prob[0] = prob[static_cast<size_t>(t)];
            if (_ptrCallback) _ptrCallback->setDOA(toDegrees(_currentDOA, 1), _prob, static_cast<double>(power[static_cast<size_t>(t)]), 1);
        }
        for (int c = 0; c < 2; ++c) _pending[c].erase(_pending[c].begin(), _pending[c].begin() + static_cast<long>(F) * hop);
        _lastArgmax = idx;
        return F;
    }
    const std::vector<int> &lastArgmax() const { return _lastArgmax; }

    // the SignalVector / SignalVector16s overloads the reference's callers use (test_mcarray.cpp:618; mcadefs.h:86-88)
    int process(const SignalVector &in, int nSamples)
    {
        std::vector<const BaseType *> pi;
        for (size_t c = 0; c < in.size(); ++c) pi.push_back(in[c].get());
        return process(pi, nSamples);
    }
    int process(const SignalVector16s &in, int nSamples)
    {
        std::vector<const BaseType16s *> pi;
        for (size_t c = 0; c < in.size(); ++c) pi.push_back(in[c].get());
        return process(pi, nSamples);
    }

private:
    static constexpr float _frameRate = 0.075f;      // BinauralLocalisation.h:196
    const int _order;
    bool _usePowerFloor = true;
    std::shared_ptr<detail::HipContext> _ctx;
    std::vector<float> _pending[2];
    std::vector<int> _lastArgmax;
};

}  // namespace mca
#endif
