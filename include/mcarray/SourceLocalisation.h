// mca::SourceLocalisation -- the localisation-only stream module for M > 2 microphones: same constructor, setCallback and
// processParametrisation hook as the reference (include/mcarray/SourceLocalisation.h:38-70; src/mcarray/SourceLocalisation.cpp:51-90).
//
// The reference IS-A dsp::STFTAnalysis (DSPONE, absent here): analysis only, no audio comes back.  Two ways in, as for
// SourceSeparationAndLocalisation: the unchanged per-frame DSPONE hook (frame API, double on the GPU), and
// process(in, nSamples) -- the stand-in for dsp::ShortTimeAnalysis::process -- which runs every completed frame of
// the chunk through the batched stream API (STFT, GCC-PHAT, SRP scan, selectDOA, gate) in one device call.
#ifndef MCA_HIP_SOURCELOCALISATION_H
#define MCA_HIP_SOURCELOCALISATION_H
#include <cmath>
#include <memory>
#include <vector>

#include "mcadefs.h"

#include "BeamformingSeparationAndLocalistaion.h"

namespace mca {

class SourceLocalisation {
public:
    SourceLocalisation(int sampleRate, ArrayDescription microphonePositions, unsigned int numOfSources, bool usePowerFloor = true,
                       double doaStepDeg = 5.0, int srpPrecision = MCA_HIP_SRP_FP32)
        : _sampleRate(sampleRate), _nchannels(static_cast<int>(microphonePositions.size())), _numOfSources(static_cast<int>(numOfSources)),
          _order(calculateOrderFromSampleRate(sampleRate, _frameRate)), _usePowerFloor(usePowerFloor)
    {
        const int N = 1 << _order;
        _impl.reset(new BeamformingSeparationAndLocalisation(sampleRate, N + 2, microphonePositions, numOfSources, usePowerFloor, doaStepDeg));
        _stream.reset(new detail::HipContext(sampleRate, microphonePositions, N, doaStepDeg, _numOfSources, usePowerFloor, srpPrecision));
        _pending.assign(static_cast<size_t>(_nchannels), std::vector<float>());
        for (int c = 0; c < _nchannels; ++c) _subBandWeights.push_back(SignalPtr(new BaseType[N + 2]));
    }
    virtual ~SourceLocalisation() {}

    void setCallback(LocalisationCallback &callback) { _callback = &callback; _impl->setCallback(callback); }     // .cpp:86-89
    void setCallback(LocalisationCallback *callback) { _callback = callback; _impl->setCallback(callback); }      // .cpp:81-84

    static int calculateOrderFromSampleRate(int sampleRate, double frameSeconds)
    {
        int order = static_cast<int>(std::lround(std::log2(sampleRate * frameSeconds)));
        return order < 8 ? 8 : (order > 14 ? 14 : order);
    }
    int getWindowSize() const { return 1 << _order; }
    int getOneSidedFFTLength() const { return (1 << (_order - 1)) + 1; }
    int getMaxLatency() const { return 1 << _order; }
    int getAnalysisLength() const { return (1 << _order) + 2; }
    int getFrameSize() const { return 1 << (_order - 1); }
    int getNumberOfChannels() const { return _nchannels; }

    // the DSPONE hook (SourceLocalisation.cpp:63-79): localisation only, frames untouched
    virtual void processParametrisation(std::vector<double *> &analysisFrames, int analysisLength,
                                        std::vector<double *> &dataChannels, int dataLength)
    {
        (void)dataChannels; (void)dataLength;
        if (analysisLength != getAnalysisLength()) throw MCArrayException("analysisLength does not match the module's FFT size");
        SignalVector af;
        for (double *p : analysisFrames) af.push_back(SignalPtr(p, [](double *) {}));   // null_deleter (.cpp:33-49)
        _impl->processFrameLocalisation(af, _subBandWeights);                            // :76
    }

    // chunked PCM in (one pointer per channel); fires the callback once per frame that passes the gate.
    // Returns the number of frames completed by this chunk.
    template <typename Tin> int process(const std::vector<Tin *> &in, int nSamples)
    {
        const int N = getWindowSize(), hop = N / 2;
        for (int c = 0; c < _nchannels; ++c)
            for (int i = 0; i < nSamples; ++i) _pending[static_cast<size_t>(c)].push_back(static_cast<float>(in[static_cast<size_t>(c)][i]));
        const int have = static_cast<int>(_pending[0].size());
        const int F = have >= N ? (have - N) / hop + 1 : 0;
        if (F == 0) return 0;
        const size_t L = static_cast<size_t>(F + 1) * static_cast<size_t>(hop);
        std::vector<float> pcm(L * static_cast<size_t>(_nchannels));
        for (int c = 0; c < _nchannels; ++c) std::copy(_pending[static_cast<size_t>(c)].begin(), _pending[static_cast<size_t>(c)].begin() + static_cast<long>(L), pcm.begin() + static_cast<long>(L * static_cast<size_t>(c)));
        const size_t FS = static_cast<size_t>(F) * static_cast<size_t>(_numOfSources);
        std::vector<int> bins(FS);
        std::vector<float> doa(FS), prob(FS);
        _stream->check(mca_hip_process_frames_host(_stream->get(), pcm.data(), 1, F, bins.data(), doa.data(), prob.data(), nullptr, nullptr));
        std::vector<unsigned char> voiced(static_cast<size_t>(F), 1);
        std::vector<float> power(static_cast<size_t>(F), 0.f);
        if (_usePowerFloor) _stream->check(mca_hip_copy_gate(_stream->get(), voiced.data(), power.data()));
        if (_callback) {
            for (int t = 0; t < F; ++t) {
                if (!voiced[static_cast<size_t>(t)]) continue;     // BeamformingSeparationAndLocalisation.cpp:87-94
                SignalPtr d(new BaseType[_numOfSources]), p(new BaseType[_numOfSources]);
                for (int s = 0; s < _numOfSources; ++s) {
                    d[s] = (180 / M_PI) * static_cast<double>(doa[static_cast<size_t>(t * _numOfSources + s)]);
                    p[s] = static_cast<double>(prob[static_cast<size_t>(t * _numOfSources + s)]);
                }
                _callback->setDOA(d, p, static_cast<double>(power[static_cast<size_t>(t)]), _numOfSources);
            }
        }
        for (int c = 0; c < _nchannels; ++c) _pending[static_cast<size_t>(c)].erase(_pending[static_cast<size_t>(c)].begin(), _pending[static_cast<size_t>(c)].begin() + static_cast<long>(F) * hop);
        _lastBins = bins;
        return F;
    }
    const std::vector<int> &lastDoaBins() const { return _lastBins; }

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
    static constexpr float _frameRate = 0.025f;     // SourceLocalisation.h:52
    const int _sampleRate;
    int _nchannels, _numOfSources, _order;
    bool _usePowerFloor;
    LocalisationCallback *_callback = nullptr;
    std::unique_ptr<BeamformingSeparationAndLocalisation> _impl;
    std::unique_ptr<detail::HipContext> _stream;
    std::vector<std::vector<float> > _pending;
    SignalVector _subBandWeights;
    std::vector<int> _lastBins;
};

}  // namespace mca
#endif
