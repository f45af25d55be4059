// mca::FastBinauralMasking -- 2-channel spatial + temporal T-F masking, same constructor, getters and
// processParametrisation hook as the reference (include/mcarray/FastBinauralMasking.h:71-104;
// src/mcarray/FastBinauralMasking.cpp:51-538).  The hook runs on the GPU in double
// (mca_hip_mask_process_frame); process() is the batched stream path (STFT, masking, ISTFT, overlap-add on
// the GPU in one call), the stand-in for the dsp::STFT::process() the reference inherits (test_mcarray.cpp:937).
#ifndef MCA_HIP_FASTBINAURALMASKING_H
#define MCA_HIP_FASTBINAURALMASKING_H
#include <cmath>
#include <string>
#include <vector>

#include "../mcarray_hip.h"
#include "ArrayModules.h"
#include "mcadefs.h"
#include "mcarray_exception.h"

namespace mca {

class FastBinauralMasking {
public:
    typedef BinauralMasking::MaskingMethod MaskingMethod;
    typedef BinauralMasking::MaskingAlg MaskingAlg;

    FastBinauralMasking(int samplerate, double microDistance, float lowFreq, float highFreq,
                        MaskingMethod mmethod = BinauralMasking::RELATIVE, MaskingAlg algorithm = BinauralMasking::BOTH)
        : _microDistance(microDistance), _order(calculateOrderFromSampleRate(samplerate, _frameRate))
    {
        mca_hip_mask_config cfg;
        cfg.struct_size = static_cast<int>(sizeof(cfg));
        cfg.device = 0;
        cfg.sample_rate = samplerate;
        cfg.fft_size = 1 << _order;
        cfg.micro_distance = microDistance;
        cfg.low_freq = lowFreq;
        cfg.high_freq = highFreq;
        cfg.method = static_cast<int>(mmethod);
        cfg.algorithm = static_cast<int>(algorithm);
        cfg.max_streams = 1;
        if (mca_hip_mask_create(&cfg, &_ctx) != MCA_HIP_OK) throw MCArrayException(std::string("mca_hip_mask_create: ") + mca_hip_mask_last_error(nullptr));
    }
    virtual ~FastBinauralMasking() { mca_hip_mask_destroy(_ctx); }
    FastBinauralMasking(const FastBinauralMasking &) = delete;
    FastBinauralMasking &operator=(const FastBinauralMasking &) = delete;

    static int calculateOrderFromSampleRate(int sampleRate, double frameSeconds)   // [BUILD-DEFINES], SURVEY A.1
    {
        int order = static_cast<int>(std::lround(std::log2(sampleRate * frameSeconds)));
        return order < 8 ? 8 : (order > 14 ? 14 : order);
    }
    int getWindowSize() const { return 1 << _order; }
    int getAnalysisLength() const { return (1 << _order) + 2; }
    int getFrameSize() const { return 1 << (_order - 1); }
    int getOneSidedFFTLength() const { return (1 << (_order - 1)) + 1; }     // _oneSidedFFTLength (FastBinauralMasking.cpp:93)
    int getMaxLatency() const { return 1 << _order; }
    int getNumberOfChannels() const { return 2; }
    int getNonMaskingAngle() { return 10; }                               // _phi in degrees (.h:99,113)
    float getMicroPhoneDistance() { return static_cast<float>(_microDistance); }
    float getSpatialMaskingFactor() { return 1 / 10.f; }                  // .h:103,117
    float getTemporalMaskingFactor() { return 1 / 3.f; }                  // .h:104,116

    // the DSPONE hook (FastBinauralMasking.cpp:126-210): analysisFrames[0,1] are modified in place
    virtual void processParametrisation(std::vector<double *> &analysisFrames, int analysisLength,
                                        std::vector<double *> &dataChannels, int dataLength)
    {
        (void)dataChannels; (void)dataLength;
        if (analysisFrames.size() != 2) throw MCArrayException("Binaural masking is only working for 2 channels.");   // .cpp:90
        check(mca_hip_mask_process_frame(_ctx, analysisFrames[0], analysisFrames[1], analysisLength, nullptr));
    }

    // chunked PCM in, masked PCM out (2 channels each); returns samples written per channel
    template <typename Tin, typename Tout>
    int process(const std::vector<Tin *> &in, int nSamples, const std::vector<Tout *> &out, int outSize)
    {
        if (in.size() != 2 || out.size() != 2) throw MCArrayException("Binaural masking is only working for 2 channels.");
        const int N = getWindowSize(), hop = N / 2;
        for (int c = 0; c < 2; ++c)
            for (int i = 0; i < nSamples; ++i) _pending[c].push_back(static_cast<float>(in[static_cast<size_t>(c)][i]));
        const int have = static_cast<int>(_pending[0].size());
        const int F = have >= N ? (have - N) / hop + 1 : 0;
        if (F == 0) return 0;
        if (F * hop > outSize) throw MCArrayException("output buffer too small for the frames completed by this chunk");
        const size_t L = static_cast<size_t>(F + 1) * static_cast<size_t>(hop);
        std::vector<float> pcm(2 * L), res(2 * static_cast<size_t>(F) * static_cast<size_t>(hop));
        for (int c = 0; c < 2; ++c) std::copy(_pending[c].begin(), _pending[c].begin() + static_cast<long>(L), pcm.begin() + static_cast<long>(L) * c);
        check(mca_hip_mask_frames_host(_ctx, pcm.data(), 1, F, res.data(), nullptr));
        for (int c = 0; c < 2; ++c) {
            for (int i = 0; i < F * hop; ++i) out[static_cast<size_t>(c)][i] = static_cast<Tout>(res[static_cast<size_t>(c) * static_cast<size_t>(F) * static_cast<size_t>(hop) + static_cast<size_t>(i)]);
            _pending[c].erase(_pending[c].begin(), _pending[c].begin() + static_cast<long>(F) * hop);
        }
        return F * hop;
    }

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
    void check(int rc) const { if (rc != MCA_HIP_OK) throw MCArrayException(std::string("libmcarray_hip: ") + mca_hip_mask_last_error(_ctx)); }
    static constexpr float _frameRate = 0.050f;      // FastBinauralMasking.h:112
    const double _microDistance;
    const int _order;
    mca_hip_mask_ctx *_ctx = nullptr;
    std::vector<float> _pending[2];
};

}  // namespace mca
#endif
