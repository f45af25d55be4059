// mca::MvdrBeamformer -- frequency-domain beamformer with a per-bin spatial covariance (BASELINE.json configs[3]).
// NOT in the reference (its only beamformer is the delay-and-sum of Beamformer.h:39,49 / Beamformer.cpp:51-71); the
// class follows the shape of the reference's stream modules (constructor = whole configuration, process() over chunks
// of PCM with one pointer per channel, SourceSeparationAndLocalisation.h:47) and Beamformer's steering convention
// (Beamformer.cpp:59).  Definition: SURVEY A.9 / include/mcarray_hip.h (mca_hip_mvdr_*).
#ifndef MCA_HIP_MVDRBEAMFORMER_H
#define MCA_HIP_MVDRBEAMFORMER_H
#include <algorithm>
#include <string>
#include <vector>

#include "../mcarray_hip.h"
#include "ArrayDescription.h"
#include "mcarray_exception.h"

namespace mca {

class MvdrBeamformer {
public:
    MvdrBeamformer(int sampleRate, ArrayDescription microphonePositions, int fftSize = 1024, double alpha = 0.95,
                   double loading = 1e-3, int device = 0)
        : _nchannels(static_cast<int>(microphonePositions.size())), _N(fftSize)
    {
        std::vector<double> xyz = microphonePositions.xyz();
        mca_hip_mvdr_config cfg;
        cfg.struct_size = static_cast<int>(sizeof(cfg));
        cfg.device = device;
        cfg.sample_rate = sampleRate;
        cfg.fft_size = fftSize;
        cfg.n_mics = _nchannels;
        cfg.mic_xyz = xyz.data();
        cfg.alpha = alpha;
        cfg.loading = loading;
        cfg.max_streams = 1;
        const int rc = mca_hip_mvdr_create(&cfg, &_ctx);
        if (rc != MCA_HIP_OK) throw MCArrayException(std::string("mca_hip_mvdr_create: ") + mca_hip_mvdr_last_error(nullptr));
        _pending.assign(static_cast<size_t>(_nchannels), std::vector<float>());
    }
    virtual ~MvdrBeamformer() { mca_hip_mvdr_destroy(_ctx); }
    MvdrBeamformer(const MvdrBeamformer &) = delete;
    MvdrBeamformer &operator=(const MvdrBeamformer &) = delete;

    int getWindowSize() const { return _N; }
    int getFrameSize() const { return _N / 2; }
    int getMaxLatency() const { return _N; }
    int getNumberOfChannels() const { return _nchannels; }
    void setDOA(double doaRadians) { _doa = doaRadians; }      // look direction of the frames completed from now on
    void reset()
    {
        check(mca_hip_mvdr_reset(_ctx, nullptr));
        for (std::vector<float> &b : _pending) b.clear();
    }

    // chunked PCM in (one pointer per channel), beamformed PCM out; returns the samples written (a multiple of the hop)
    template <typename Tin, typename Tout>
    int process(const std::vector<Tin *> &in, int nSamples, Tout *out, int outSize)
    {
        const int hop = _N / 2;
        for (int c = 0; c < _nchannels; ++c) {
            std::vector<float> &buf = _pending[static_cast<size_t>(c)];
            const size_t old = buf.size();
            buf.resize(old + static_cast<size_t>(nSamples));
            for (int i = 0; i < nSamples; ++i) buf[old + static_cast<size_t>(i)] = static_cast<float>(in[static_cast<size_t>(c)][i]);
        }
        const int have = static_cast<int>(_pending[0].size());
        const int F = have >= _N ? (have - _N) / hop + 1 : 0;
        if (F == 0) return 0;
        if (F * hop > outSize) throw MCArrayException("output buffer too small for the frames completed by this chunk");
        const size_t L = static_cast<size_t>(F + 1) * static_cast<size_t>(hop);
        std::vector<float> pcm(L * static_cast<size_t>(_nchannels));
        for (int c = 0; c < _nchannels; ++c)
            std::copy(_pending[static_cast<size_t>(c)].begin(), _pending[static_cast<size_t>(c)].begin() + static_cast<long>(L), pcm.begin() + static_cast<long>(L * static_cast<size_t>(c)));
        std::vector<float> doa(static_cast<size_t>(F), static_cast<float>(_doa)), audio(static_cast<size_t>(F) * static_cast<size_t>(hop));
        check(mca_hip_mvdr_frames_host(_ctx, pcm.data(), 1, F, doa.data(), audio.data(), nullptr));
        for (int i = 0; i < F * hop; ++i) out[i] = static_cast<Tout>(audio[static_cast<size_t>(i)]);
        for (int c = 0; c < _nchannels; ++c)
            _pending[static_cast<size_t>(c)].erase(_pending[static_cast<size_t>(c)].begin(), _pending[static_cast<size_t>(c)].begin() + static_cast<long>(F) * hop);
        return F * hop;
    }

private:
    void check(int rc) const
    {
        if (rc != MCA_HIP_OK) throw MCArrayException(std::string("libmcarray_hip: ") + mca_hip_mvdr_last_error(_ctx));
    }
    int _nchannels, _N;
    double _doa = 0.0;
    mca_hip_mvdr_ctx *_ctx = nullptr;
    std::vector<std::vector<float> > _pending;
};

}  // namespace mca
#endif
