// RAII owner of one mca_hip_ctx (include/mcarray_hip.h) + status -> MCArrayException translation.
// Internal helper of the module classes in this directory; not part of the reference's API.
#ifndef MCA_HIP_HIPCONTEXT_H
#define MCA_HIP_HIPCONTEXT_H
#include <string>
#include <vector>

#include "../mcarray_hip.h"
#include "ArrayDescription.h"
#include "mcarray_exception.h"

namespace mca {
namespace detail {

class HipContext {
public:
    HipContext(int sampleRate, const ArrayDescription &mics, int fftSize, double doaStepDeg, int numOfSources,
               bool usePowerFloor, int srpPrecision = MCA_HIP_SRP_FP32, int maxArrays = 1, int device = 0)
    {
        std::vector<double> xyz = mics.xyz();
        mca_hip_config cfg = mca_hip_config();     // zeroed: fields appended later (gcc_weighting: 0 = PHAT) keep their defaults
        cfg.struct_size = static_cast<int>(sizeof(cfg));
        cfg.device = device;
        cfg.sample_rate = sampleRate;
        cfg.fft_size = fftSize;
        cfg.n_mics = static_cast<int>(mics.size());
        cfg.mic_xyz = xyz.data();
        cfg.doa_step_deg = doaStepDeg;
        cfg.n_sources = numOfSources;
        cfg.use_power_floor = usePowerFloor ? 1 : 0;
        cfg.srp_precision = srpPrecision;
        cfg.max_arrays = maxArrays;
        const int rc = mca_hip_create(&cfg, &ctx_);
        if (rc != MCA_HIP_OK) throw MCArrayException(std::string("mca_hip_create: ") + mca_hip_last_error(nullptr));
    }
    ~HipContext() { mca_hip_destroy(ctx_); }
    HipContext(const HipContext &) = delete;
    HipContext &operator=(const HipContext &) = delete;
    mca_hip_ctx *get() const { return ctx_; }
    void check(int rc) const
    {
        if (rc != MCA_HIP_OK) throw MCArrayException(std::string("libmcarray_hip: ") + mca_hip_last_error(ctx_));
    }

private:
    mca_hip_ctx *ctx_ = nullptr;
};

}  // namespace detail
}  // namespace mca
#endif
