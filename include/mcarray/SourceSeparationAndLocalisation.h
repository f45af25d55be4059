// mca::SourceSeparationAndLocalisation -- the stream-level module for M > 2 microphones, same constructor,
// setCallback and processParametrisation hook as the reference (include/mcarray/SourceSeparationAndLocalisation.h:47-71;
// src/mcarray/SourceSeparationAndLocalisation.cpp:51-107).
//
// The reference IS-A dsp::STFT (DSPONE, absent here).  Two ways in:
//   * processParametrisation(std::vector<double*>&, ...) -- the DSPONE per-frame hook, unchanged signature: a
//     DSPONE build calls it from its own STFT; frames are modified in place (frame API, double on the GPU).
//   * process(in, nSamples, out, outSize) -- a stand-in for the dsp::ShortTimeProcess::process overloads the
//     reference's callers use (mcabeamf.cpp:112, test_mcarray.cpp:869): buffers chunked PCM, runs every complete
//     frame of the chunk through the batched stream API in ONE device call (STFT, GCC-PHAT, SRP, pick,
//     delay-and-sum, ISTFT, overlap-add all on the GPU; any frame length 2^order the sample rate gives), fires
//     the callback once per frame.  [BUILD-DEFINES]
//     framing: N = 2^order, hop N/2, periodic Hann, plain overlap-add.
#ifndef MCA_HIP_SOURCESEPARATIONANDLOCALISATION_H
#define MCA_HIP_SOURCESEPARATIONANDLOCALISATION_H
#include <cmath>
#include <cstdint>
#include <memory>
#include <type_traits>
#include <vector>

#include "mcadefs.h"

#include "BeamformingSeparationAndLocalistaion.h"

namespace mca {

class SourceSeparationAndLocalisation {
public:
    SourceSeparationAndLocalisation(int sampleRate, ArrayDescription microphonePositions, unsigned int numOfSources,
                                    bool usePowerFloor = true, double doaStepDeg = 5.0, int srpPrecision = MCA_HIP_SRP_FP32)
        : _sampleRate(sampleRate), _nchannels(static_cast<int>(microphonePositions.size())), _numOfSources(static_cast<int>(numOfSources)),
          _order(calculateOrderFromSampleRate(sampleRate, _frameRate)), _usePowerFloor(usePowerFloor)
    {
        const int N = 1 << _order;
        _impl.reset(new BeamformingSeparationAndLocalisation(sampleRate, N + 2, microphonePositions, numOfSources, usePowerFloor, doaStepDeg));
        _stream.reset(new detail::HipContext(sampleRate, microphonePositions, N, doaStepDeg, _numOfSources, usePowerFloor, srpPrecision));
        _pending.assign(static_cast<size_t>(_nchannels), std::vector<float>());
        for (int c = 0; c < _nchannels; ++c) _wienerCoefs.push_back(SignalPtr(new BaseType[N + 2]));
    }
    virtual ~SourceSeparationAndLocalisation() {}

    void setCallback(LocalisationCallback &callback) { _callback = &callback; _impl->setCallback(callback); }
    void setCallback(LocalisationCallback *callback) { _callback = callback; _impl->setCallback(callback); }

    // [BUILD-DEFINES] stand-in for dsp::STFT::calculateOrderFromSampleRate
    static int calculateOrderFromSampleRate(int sampleRate, double frameSeconds)
    {
        int order = static_cast<int>(std::lround(std::log2(sampleRate * frameSeconds)));
        return order < 8 ? 8 : (order > 14 ? 14 : order);
    }
    int getWindowSize() const { return 1 << _order; }
    int getAnalysisLength() const { return (1 << _order) + 2; }
    int getOneSidedFFTLength() const { return (1 << (_order - 1)) + 1; }
    int getFrameSize() const { return 1 << (_order - 1); }       // hop
    int getMaxLatency() const { return 1 << _order; }
    int getNumberOfChannels() const { return _nchannels; }

    // The DSPONE hook (SourceSeparationAndLocalisation.cpp:66-94): localise, then separate in place.
    virtual void processParametrisation(std::vector<double *> &analysisFrames, int analysisLength,
                                        std::vector<double *> &dataChannels, int dataLength)
    {
        (void)dataChannels; (void)dataLength;
        if (analysisLength != getAnalysisLength()) throw MCArrayException("analysisLength does not match the module's FFT size");
        SignalVector sf;
        for (double *p : analysisFrames) sf.push_back(SignalPtr(p, [](double *) {}));   // non-owning, like null_deleter (.cpp:33-49)
        _impl->processFrameLocalisation(sf, _wienerCoefs);     // .cpp:87
        _impl->processFrameSeparation(sf, sf);                 // .cpp:92
    }

    // process(): chunked PCM in (one pointer per channel), beamformed PCM out (one pointer per source channel;
    // channels beyond numOfSources are zero-filled like processFrameSeparation does).  Returns samples written.
    template <typename Tin, typename Tout>
    int process(const std::vector<Tin *> &in, int nSamples, const std::vector<Tout *> &out, int outSize)
    {
        const int N = getWindowSize(), hop = N / 2;
        for (int c = 0; c < _nchannels; ++c) {
            std::vector<float> &buf = _pending[static_cast<size_t>(c)];
            const size_t old = buf.size();
            buf.resize(old + static_cast<size_t>(nSamples));
            for (int i = 0; i < nSamples; ++i) buf[old + static_cast<size_t>(i)] = static_cast<float>(in[static_cast<size_t>(c)][i]);
        }
        const int have = static_cast<int>(_pending[0].size());
        const int F = have >= N ? (have - N) / hop + 1 : 0;
        if (F == 0) return 0;
        if (F * hop > outSize) throw MCArrayException("output buffer too small for the frames completed by this chunk");
        const size_t L = static_cast<size_t>(F + 1) * static_cast<size_t>(hop);
        std::vector<float> pcm(L * static_cast<size_t>(_nchannels));
        for (int c = 0; c < _nchannels; ++c) std::copy(_pending[static_cast<size_t>(c)].begin(), _pending[static_cast<size_t>(c)].begin() + static_cast<long>(L), pcm.begin() + static_cast<long>(L * static_cast<size_t>(c)));
        const size_t FS = static_cast<size_t>(F) * static_cast<size_t>(_numOfSources);
        std::vector<int> bins(FS);
        std::vector<float> doa(FS), prob(FS), audio(FS * static_cast<size_t>(hop));
        if (std::is_same<typename std::remove_cv<Tin>::type, short>::value) {
            // 16-bit PCM (the process(std::vector<int16_t*>&, ...) overloads of the reference's callers, mcabeamf.cpp:112):
            // the shorts go up as they are, half the PCIe bytes; the buffered floats hold them exactly
            std::vector<short> pcm16(pcm.size());
            for (size_t i = 0; i < pcm.size(); ++i) pcm16[i] = static_cast<short>(pcm[i]);
            _stream->check(mca_hip_process_frames_host_i16(_stream->get(), pcm16.data(), 1, F, bins.data(), doa.data(), prob.data(), nullptr, audio.data()));
        } else {
            _stream->check(mca_hip_process_frames_host(_stream->get(), pcm.data(), 1, F, bins.data(), doa.data(), prob.data(), nullptr, audio.data()));
        }
        std::vector<unsigned char> voiced(static_cast<size_t>(F), 1);
        std::vector<float> power(static_cast<size_t>(F), 0.f);
        if (_usePowerFloor) _stream->check(mca_hip_copy_gate(_stream->get(), voiced.data(), power.data()));
        if (_callback) {
            for (int t = 0; t < F; ++t) {
                if (!voiced[static_cast<size_t>(t)]) continue;     // gated out: no setDOA (BeamformingSeparationAndLocalisation.cpp:87-94)
                SignalPtr d(new BaseType[_numOfSources]), p(new BaseType[_numOfSources]);
                for (int s = 0; s < _numOfSources; ++s) {
                    d[s] = (180 / M_PI) * static_cast<double>(doa[static_cast<size_t>(t * _numOfSources + s)]);   // toDegrees
                    p[s] = static_cast<double>(prob[static_cast<size_t>(t * _numOfSources + s)]);
                }
                _callback->setDOA(d, p, static_cast<double>(power[static_cast<size_t>(t)]), _numOfSources);
            }
        }
        for (size_t c = 0; c < out.size(); ++c)
            for (int i = 0; i < F * hop; ++i)
                out[c][i] = static_cast<int>(c) < _numOfSources ? static_cast<Tout>(audio[c * static_cast<size_t>(F) * static_cast<size_t>(hop) + static_cast<size_t>(i)]) : static_cast<Tout>(0);
        for (int c = 0; c < _nchannels; ++c) _pending[static_cast<size_t>(c)].erase(_pending[static_cast<size_t>(c)].begin(), _pending[static_cast<size_t>(c)].begin() + static_cast<long>(F) * hop);
        _lastBins = bins;
        return F * hop;
    }
    const std::vector<int> &lastDoaBins() const { return _lastBins; }

    // the SignalVector / SignalVector16s overloads the reference's callers use (test_mcarray.cpp:869,937; mcadefs.h:86-88)
    int process(const SignalVector &in, int nSamples, SignalVector &out, int outSize)
    {
        std::vector<const BaseType *> pi; std::vector<BaseType *> po;
        for (size_t c = 0; c < in.size(); ++c) pi.push_back(in[c].get());
        for (size_t c = 0; c < out.size(); ++c) po.push_back(out[c].get());
        return process(pi, nSamples, po, outSize);
    }
    int process(const SignalVector16s &in, int nSamples, SignalVector16s &out, int outSize)
    {
        std::vector<const BaseType16s *> pi; std::vector<BaseType16s *> po;
        for (size_t c = 0; c < in.size(); ++c) pi.push_back(in[c].get());
        for (size_t c = 0; c < out.size(); ++c) po.push_back(out[c].get());
        return process(pi, nSamples, po, outSize);
    }

private:
    static constexpr float _frameRate = 0.025f;     // SourceSeparationAndLocalisation.h:60
    const int _sampleRate;
    int _nchannels, _numOfSources, _order;
    bool _usePowerFloor;
    LocalisationCallback *_callback = nullptr;
    std::unique_ptr<BeamformingSeparationAndLocalisation> _impl;
    std::unique_ptr<detail::HipContext> _stream;
    std::vector<std::vector<float> > _pending;
    SignalVector _wienerCoefs;
    std::vector<int> _lastBins;
};

}  // namespace mca
#endif
