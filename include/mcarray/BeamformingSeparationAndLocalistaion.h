// mca::BeamformingSeparationAndLocalisation -- per-frame orchestrator (file name sic, as in the reference:
// include/mcarray/BeamformingSeparationAndLocalistaion.h:40-44; src/mcarray/BeamformingSeparationAndLocalisation.cpp:29-119).
// Power gate bookkeeping (scalars) is host logic; every per-bin operation (FFT power, SRP scan, peak pick,
// delay-and-sum) runs on the GPU through the C ABI.
#ifndef MCA_HIP_BEAMFORMINGSEPARATIONANDLOCALISATION_H
#define MCA_HIP_BEAMFORMINGSEPARATIONANDLOCALISATION_H
#include <algorithm>
#include <cmath>
#include <cstring>
#include <memory>
#include <vector>

#include "Beamformer.h"
#include "SoundLocalisationImpl.h"
#include "SteeringBeamforming.h"

namespace mca {

class BeamformingSeparationAndLocalisation : public SoundLocalisationImpl {
public:
    BeamformingSeparationAndLocalisation(int sampleRate, int fftCCSLength, ArrayDescription microphonePositions,
                                         unsigned int numOfSources, bool usePowerFloor, double doaStepDeg = 5.0)
        : SoundLocalisationImpl(microphonePositions), _nchannels(static_cast<unsigned>(microphonePositions.size())),
          _sampleRate(sampleRate), _fftCCSLength(fftCCSLength), _usePowerFloor(usePowerFloor), _numOfSources(numOfSources),
          _ctx(new detail::HipContext(sampleRate, microphonePositions, fftCCSLength - 2, doaStepDeg, static_cast<int>(numOfSources), usePowerFloor))
    {
        // (limit of this build, MCA_MAX_SOURCES = 4; the reference has none: BeamformingSeparationAndLocalisation.cpp:113-118 loops min(M, S))
        if (numOfSources < 1 || numOfSources > 4) throw MCArrayException("numOfSources must be in [1,4]");
        for (unsigned c = 0; c < _nchannels; ++c) _inputFrames.push_back(SignalPtr(new BaseType[_fftCCSLength]));
        _currentDOA.reset(new BaseType[_numOfSources]);
        _prob.reset(new BaseType[_numOfSources]);
        for (unsigned s = 0; s < _numOfSources; ++s) { _currentDOA[s] = 0; _prob[s] = -1.0; }   // .cpp:51-52
    }
    virtual ~BeamformingSeparationAndLocalisation() {}

    void processFrameLocalisation(SignalVector &analysisFrames, SignalVector &wienerCoefs)   // .cpp:74-101
    {
        (void)wienerCoefs;
        std::vector<const double *> rows = rowsOf(analysisFrames);
        double powerDb = 0;
        _ctx->check(mca_hip_fft_log_power(_ctx->get(), rows.data(), _fftCCSLength, &powerDb));
        BaseType power;
        if (!_noiseEstimated && _usePowerFloor) power = setPowerFloor(std::pow(10.0, powerDb / 10.0));   // .cpp:80-81
        else power = powerDb;                                                                             // .cpp:83
        if ((power > _powerFloor) || !_usePowerFloor) {                                                   // .cpp:87
            int bins[4];
            _ctx->check(mca_hip_steering_process_frame(_ctx->get(), rows.data(), _fftCCSLength, _currentDOA.get(), _prob.get(), bins,
                                                       static_cast<int>(_numOfSources)));
            if (_ptrCallback) _ptrCallback->setDOA(toDegrees(_currentDOA, static_cast<int>(_numOfSources)), _prob, power, static_cast<int>(_numOfSources));   // .cpp:93
        }
    }

    void processFrameSeparation(SignalVector &inputFrames, SignalVector &outputFrames)                    // .cpp:103-119
    {
        unsigned c;
        for (c = 0; c < _nchannels; ++c) std::memcpy(_inputFrames[c].get(), inputFrames[c].get(), sizeof(BaseType) * _fftCCSLength);
        std::vector<const double *> rows = rowsOf(_inputFrames);
        for (c = 0; c < std::min(_nchannels, _numOfSources); ++c)
            _ctx->check(mca_hip_beamformer_process_frame(_ctx->get(), rows.data(), _fftCCSLength, outputFrames[c].get(), _currentDOA[c]));
        for (; c < _nchannels; ++c) std::memset(outputFrames[c].get(), 0, sizeof(BaseType) * _fftCCSLength);
    }

    const SignalPtr &currentDOA() const { return _currentDOA; }

private:
    std::vector<const double *> rowsOf(const SignalVector &v) const
    {
        std::vector<const double *> rows(_nchannels);
        for (unsigned c = 0; c < _nchannels; ++c) rows[c] = v[c].get();
        return rows;
    }
    BaseType setPowerFloor(double linearPower)                                                            // .cpp:55-72
    {
        const int neededSamples = static_cast<int>(_durationToEstimatePowerFloor * _sampleRate);
        _powerFloor += linearPower * (_fftCCSLength - 2);
        _samplesConsumedForNoise += (_fftCCSLength - 2);
        if (_samplesConsumedForNoise >= neededSamples) {
            _noiseEstimated = true;
            _powerFloor /= _samplesConsumedForNoise;
            _powerFloor = 10 * std::log10(_powerFloor) + _noiseMarginDB;
        }
        return _powerFloor;
    }

    const unsigned int _nchannels;
    int _sampleRate;
    int _fftCCSLength;
    bool _usePowerFloor;
    static constexpr double _noiseMarginDB = 3;      // BeamformingSeparationAndLocalistaion.h:52
    unsigned int _numOfSources;
    SignalVector _inputFrames;
    std::shared_ptr<detail::HipContext> _ctx;
};

}  // namespace mca
#endif
