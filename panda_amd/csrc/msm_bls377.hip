// msm_bls377.hip -- the MSM kernels and driver of msm_impl.h instantiated for BLS12-377.
#define PANDA_MSM_IMPL
#include "msm_impl.h"

namespace panda {

hipError_t msm_execute_bls377(const panda_msm_configuration &cfg, const MsmRegistration *reg, MsmTuning tuning, float *phase_ms, bool *stale,
                              const MsmPipeline *pipe)
{
    return msm_execute<CurveBls377>(cfg, reg, tuning, phase_ms, stale, pipe);
}

hipError_t msm_build_registration_bls377(MsmRegistration &r, hipStream_t s) { return build_registration<Bls377Fq>(r, s); }

} // namespace panda
