// mca::MultibandBinarualLocalisation -- 2-microphone localiser: GCC-PHAT per sub-band, DOA histogram weighted by the
// band energies (include/mcarray/MultibandBinarualLocalisation.h:36-103 of the reference -- the file name keeps its
// spelling; src/mcarray/MultibandBinarualLocalisation.cpp:52-258).
//
// The reference IS-A dsp::SubBandSTFTAnalysis (DSPONE, absent here) whose engine calls processSetup /
// processOneSubband / processSumamry per frame.  This class offers the caller-facing side -- the constructor,
// setCallback and a process(in, nSamples) stand-in for the dsp::ShortTimeAnalysis::process overloads -- and runs all
// completed frames of a chunk in ONE device call (STFT, band split, per-band GCC-PHAT, smoothing, histogram, gate).
// [BUILD-DEFINES] framing as elsewhere (N = 2^order, hop N/2, periodic Hann) and the LINEAR filter bank: nbins
// unit-peak triangles between 100 Hz and maxFreqForSpatialAliasing(distance(0,1)).
#ifndef MCA_HIP_MULTIBANDBINARUALLOCALISATION_H
#define MCA_HIP_MULTIBANDBINARUALLOCALISATION_H
#include <cmath>
#include <string>
#include <vector>

#include "mcadefs.h"

#include "../mcarray_hip.h"
#include "SoundLocalisationImpl.h"
#include "microhponeArrayHelpers.h"

namespace mca {

class MultibandBinarualLocalisation : public SoundLocalisationImpl {
public:
    MultibandBinarualLocalisation(int sampleRate, ArrayDescription microphonePositions, int nbins = 15, bool usePowerFloor = true)
        : SoundLocalisationImpl(microphonePositions), _order(calculateOrderFromSampleRate(sampleRate, _frameRate)), _nbins(nbins)
    {
        if (microphonePositions.size() != 2) throw MCArrayException("MultibandBinarualLocalisation needs an ArrayDescription with 2 microphones");
        std::vector<double> xyz = microphonePositions.xyz();
        mca_hip_mb_config cfg;
        cfg.struct_size = static_cast<int>(sizeof(cfg));
        cfg.device = 0;
        cfg.sample_rate = sampleRate;
        cfg.fft_size = 1 << _order;
        cfg.mic_xyz = xyz.data();
        cfg.nbins = nbins;
        cfg.use_power_floor = usePowerFloor ? 1 : 0;
        cfg.max_arrays = 1;
        if (mca_hip_mb_create(&cfg, &_ctx) != MCA_HIP_OK) throw MCArrayException(std::string("mca_hip_mb_create: ") + mca_hip_mb_last_error(nullptr));
        _currentDOA.reset(new BaseType[1]);
        _prob.reset(new BaseType[1]);
        _currentDOA[0] = 0; _prob[0] = -1;                   // MultibandBinarualLocalisation.cpp:79-82
    }
    virtual ~MultibandBinarualLocalisation() { mca_hip_mb_destroy(_ctx); }
    MultibandBinarualLocalisation(const MultibandBinarualLocalisation &) = delete;
    MultibandBinarualLocalisation &operator=(const MultibandBinarualLocalisation &) = delete;

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
    int getNumberOfBins() const { return _nbins; }

    // chunked PCM in (2 channels); fires setDOA(degrees, prob, power, 1) once per frame that passes the gate (:225-248).
    // Returns the number of frames completed by this chunk.
    template <typename Tin> int process(const std::vector<Tin *> &in, int nSamples)
    {
        const int N = getWindowSize(), hop = N / 2;
        for (int c = 0; c < 2; ++c)
            for (int i = 0; i < nSamples; ++i) _pending[c].push_back(static_cast<float>(in[static_cast<size_t>(c)][i]));
        const int have = static_cast<int>(_pending[0].size());
        const int F = have >= N ? (have - N) / hop + 1 : 0;
        if (F == 0) return 0;
        const size_t L = static_cast<size_t>(F + 1) * static_cast<size_t>(hop);
        std::vector<float> pcm(2 * L), doa(static_cast<size_t>(F)), prob(static_cast<size_t>(F)), power(static_cast<size_t>(F));
        std::vector<unsigned char> voiced(static_cast<size_t>(F));
        for (int c = 0; c < 2; ++c) std::copy(_pending[c].begin(), _pending[c].begin() + static_cast<long>(L), pcm.begin() + static_cast<long>(L) * c);
        if (mca_hip_mb_frames_host(_ctx, pcm.data(), 1, F, doa.data(), prob.data(), voiced.data(), power.data(), nullptr, nullptr, nullptr) != MCA_HIP_OK)
            throw MCArrayException(std::string("libmcarray_hip: ") + mca_hip_mb_last_error(_ctx));
        for (int t = 0; t < F; ++t) {
            _currentDOA[0] = doa[static_cast<size_t>(t)]; _prob[0] = prob[static_cast<size_t>(t)];
            if (voiced[static_cast<size_t>(t)] && _ptrCallback) _ptrCallback->setDOA(toDegrees(_currentDOA, 1), _prob, power[static_cast<size_t>(t)], 1);
        }
        for (int c = 0; c < 2; ++c) _pending[c].erase(_pending[c].begin(), _pending[c].begin() + static_cast<long>(F) * hop);
        return F;
    }

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
    static constexpr float _frameRate = 0.025f;      // MultibandBinarualLocalisation.h:43
    const int _order, _nbins;
    mca_hip_mb_ctx *_ctx = nullptr;
    std::vector<float> _pending[2];
};

}  // namespace mca
#endif
