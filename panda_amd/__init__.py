"""panda_amd -- MI355X-native MSM + NTT behind panda's C ABI.

The package holds only what the hot path needs: `csrc/` (hand-written HIP kernels for gfx950 and the
`extern "C"` boundary declared in include/panda_interface.h), `gpu_ffi` (ctypes mirror of the reference's
Rust gpu_ffi) and `gpu_manager` (mirror of the reference's Rust gpu_manager: device/pool/stream ownership
and the staging protocol of panda_msm_bn254_gpu* / panda_ntt_bn254_gpu*).
Nothing here imports the CPU oracle under oracle/.
"""
from . import gpu_ffi, gpu_manager  # noqa: F401
from .gpu_ffi import JACOBIAN, PROJECTIVE, PandaGpuError  # noqa: F401
from .gpu_manager import (  # noqa: F401
    BLS12_377, BLS12_381, BN254, PandaGpuManager, panda_intt_bn254_gpu, panda_msm_bn254_gpu, panda_msm_bn254_gpu_host,
    panda_msm_bn254_gpu_with_cached_bases, panda_msm_bn254_gpu_with_cached_bases_batched, panda_msm_bn254_gpu_with_cached_input, panda_msm_bn254_gpu_with_cached_scalars,
    panda_ntt_bn254_gpu, panda_ntt_bn254_gpu_v1)
