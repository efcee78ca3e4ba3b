// msm_bn254g2.hip -- the MSM kernels and driver of msm_impl.h instantiated for BN254 G2 (coordinates in Fq2, fe29_ext2.h).
#define PANDA_MSM_IMPL
#include "msm_impl.h"

namespace panda {

hipError_t msm_execute_bn254_g2(const panda_msm_configuration &cfg, const MsmRegistration *reg, MsmTuning tuning, float *phase_ms, bool *stale,
                                const MsmPipeline *pipe)
{
    return msm_execute<CurveBn254G2>(cfg, reg, tuning, phase_ms, stale, pipe);
}

hipError_t msm_build_registration_bn254_g2(MsmRegistration &r, hipStream_t s) { return build_registration<Ext2<Bn254Fq>>(r, s); }

} // namespace panda
