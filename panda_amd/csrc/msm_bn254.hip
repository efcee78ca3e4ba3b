// msm_bn254.hip -- the MSM kernels and driver of msm_impl.h instantiated for BN254.
#define PANDA_MSM_IMPL
#include "msm_impl.h"

namespace panda {

hipError_t msm_execute_bn254(const panda_msm_configuration &cfg, const MsmRegistration *reg, MsmTuning tuning, float *phase_ms, bool *stale,
                              const MsmPipeline *pipe)
{
    return msm_execute<CurveBn254>(cfg, reg, tuning, phase_ms, stale, pipe);
}

hipError_t msm_build_registration_bn254(MsmRegistration &r, hipStream_t s) { return build_registration<Bn254Fq>(r, s); }

} // namespace panda
