"""ctypes binding of the C ABI in include/panda_interface.h.

Python counterpart of the reference's Rust `gpu_ffi` layer (src/gpu_ffi/binding.rs:3-115 for the
extern block, src/gpu_ffi/common.rs:5-208 for the repr(C) structs and the error enum): same symbol
names, same by-value handle and configuration structs, same "non-zero means failure" convention.
The library is the in-tree HIP build (panda_amd/csrc/libpanda-cuda.so); there is no fallback:
loading fails loudly if it has not been built.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libpanda-cuda.so")


class PandaGpuError(RuntimeError):
    """Mirrors enum PandaGpuError (gpu_ffi/common.rs:5-38); `kind` carries the variant name."""

    def __init__(self, kind: str, code: int = 0):
        super().__init__(f"{kind} (panda_error={code})")
        self.kind, self.code = kind, code


class PandaStream(C.Structure):  # gpu_ffi/common.rs:40-44
    _fields_ = [("handle", C.c_void_p)]


class PandaEvent(C.Structure):  # gpu_ffi/common.rs:89-93
    _fields_ = [("handle", C.c_void_p)]


class PandaMemPool(C.Structure):  # gpu_ffi/common.rs:134-138
    _fields_ = [("handle", C.c_void_p)]


JACOBIAN, PROJECTIVE = 0, 1  # PandaMSMResultCoordinateType, gpu_ffi/common.rs:160-166


class MSMConfiguration(C.Structure):  # gpu_ffi/common.rs:168-185, panda_interface.cuh:70-79
    _fields_ = [("mem_pool", PandaMemPool), ("stream", PandaStream), ("bases", C.c_void_p), ("scalars", C.c_void_p),
                ("results", C.c_void_p), ("log_scalars_count", C.c_uint), ("msm_result_coordinate_type", C.c_int)]


class NTTConfiguration(C.Structure):  # gpu_ffi/common.rs:187-196
    _fields_ = [("mem_pool", PandaMemPool), ("stream", PandaStream), ("d_src", C.c_void_p), ("d_dst", C.c_void_p),
                ("log_n", C.c_uint), ("flag", C.POINTER(C.c_uint))]


class NttconfigurationV1(C.Structure):  # gpu_ffi/common.rs:198-208
    _fields_ = [("mem_pool", PandaMemPool), ("stream", PandaStream), ("d_src", C.c_void_p), ("d_dst", C.c_void_p),
                ("omega", C.c_void_p), ("log_n", C.c_uint), ("flag", C.POINTER(C.c_uint))]


class PandaMultiGpu(C.Structure):  # additive: include/panda_interface.h panda_multi_gpu (one process, one worker thread + RCCL communicator per device)
    _fields_ = [("handle", C.c_void_p)]


MULTI_RCCL, MULTI_LOOPBACK = 0, 1
CLOCK_WORDS, CLOCK_STAMP_BYTES = 12, 32768  # PANDA_CLOCK_WORDS, PANDA_CLOCK_STAMP_BYTES


class NttSlabConfiguration(C.Structure):  # additive: include/panda_interface.h panda_ntt_slab_configuration
    _fields_ = [("stream", PandaStream), ("d_slab", C.c_void_p), ("d_scratch", C.c_void_p), ("omega", C.c_void_p),
                ("log_n", C.c_uint), ("log_ranks", C.c_uint), ("rank", C.c_uint), ("flag", C.POINTER(C.c_uint))]


# every symbol include/panda_interface.h declares; tests assert the library exports each one
REFERENCE_SYMBOLS = [
    "panda_get_device_number", "panda_get_device", "panda_set_device", "panda_stream_create", "panda_stream_wait_event",
    "panda_stream_sync", "panda_stream_destroy", "panda_launch_host_fn", "panda_event_create", "panda_event_record",
    "panda_event_sync", "panda_event_query", "panda_event_destroy", "panda_mem_get_info", "panda_malloc", "panda_malloc_host",
    "panda_free", "panda_free_host", "panda_host_register", "panda_host_unregister", "panda_memcpy", "panda_memcpy_async",
    "panda_memset", "panda_memset_async", "panda_mem_pool_create", "panda_mem_pool_destroy", "panda_malloc_from_pool_async",
    "panda_free_async", "panda_msm_setup_bn254", "panda_msm_execute_bn254", "panda_msm_execute_bn254_host", "panda_msm_tear_down",
    "panda_ntt_setup_bn254", "panda_ntt_execute_bn254", "panda_ntt_tear_down", "panda_ntt_execute_bn254_v1",
]
RUST_ONLY_SYMBOLS = ["panda_stream_synchronize", "panda_stream_query", "panda_device_enable_peer_access", "panda_device_disable_peer_access"]
ADDITIVE_SYMBOLS = [
    "panda_msm_register_bases", "panda_msm_unregister_bases", "panda_msm_execute_from_host", "panda_msm_precompute_bases", "panda_msm_registered_info", "panda_msm_set_chunk_entries", "panda_msm_set_overlap", "panda_msm_set_accumulate_variant", "panda_msm_set_wide_merge", "panda_msm_set_reduce_group", "panda_msm_plain_window_plan", "panda_msm_verify_registered", "panda_msm_set_paranoid", "panda_msm_set_phase_timing", "panda_msm_setup_bls12_377", "panda_msm_execute_bls12_377", "panda_msm_execute_bls12_377_host", "panda_msm_set_window_bits",
    "panda_msm_last_phase_ms", "panda_msm_phase_name", "panda_ntt_last_device_ms", "panda_ntt_pass_plan", "panda_ntt_set_streamed_tables", "panda_ntt_execute_bn254_inverse", "panda_ntt_execute_bls12_377_v1", "panda_ntt_execute_bls12_377_inverse", "panda_msm_combine_bn254",
    "panda_msm_combine_bls12_377", "panda_msm_setup_bn254_g2", "panda_msm_execute_bn254_g2", "panda_msm_execute_bn254_g2_host", "panda_msm_combine_bn254_g2", "panda_msm_setup_bls12_381", "panda_msm_execute_bls12_381", "panda_msm_execute_bls12_381_host", "panda_msm_combine_bls12_381",
    "panda_ntt_execute_bls12_381_v1", "panda_ntt_execute_bls12_381_inverse", "panda_ntt_execute_bn254_coset", "panda_ntt_execute_bn254_coset_inverse", "panda_ntt_execute_bn254_bitrev_out", "panda_ntt_execute_bn254_inverse_bitrev_in", "panda_ntt_slab_step1_bn254", "panda_ntt_slab_step2_bn254", "panda_ntt_slab_step1_bn254_enqueue", "panda_ntt_slab_step2_bn254_enqueue", "panda_ntt_slab_inverse_step1_bn254_enqueue", "panda_ntt_slab_inverse_step2_bn254_enqueue", "panda_gen_scalars", "panda_gen_bases",
    "panda_debug_field_op", "panda_debug_curve_op", "panda_version",
    "panda_ntt_table_builds", "panda_msm_set_chunk_first", "panda_set_clock_stamps", "panda_msm_last_clock", "panda_ntt_last_clock", "panda_clock_stamp", "panda_clock_delta",
    "panda_multi_gpu_create", "panda_multi_gpu_destroy", "panda_multi_gpu_device_count", "panda_msm_execute_bn254_multi", "panda_msm_execute_bls12_377_multi",
    "panda_msm_execute_bn254_from_host_multi", "panda_msm_execute_bls12_377_from_host_multi",
    "panda_msm_execute_bls12_381_multi", "panda_msm_execute_bn254_g2_multi", "panda_msm_execute_bls12_381_from_host_multi", "panda_msm_execute_bn254_g2_from_host_multi",
    "panda_ntt_execute_bn254_multi", "panda_ntt_execute_bn254_inverse_multi", "panda_ntt_execute_bn254_multi_batch", "panda_ntt_execute_bn254_inverse_multi_batch", "panda_multi_gpu_last_phase_ms",
    "panda_ntt_execute_bls12_377_bitrev_out", "panda_ntt_execute_bls12_377_inverse_bitrev_in", "panda_ntt_execute_bls12_377_coset", "panda_ntt_execute_bls12_377_coset_inverse",
    "panda_ntt_slab_step1_bls12_377_enqueue", "panda_ntt_slab_step2_bls12_377_enqueue", "panda_ntt_slab_inverse_step1_bls12_377_enqueue", "panda_ntt_slab_inverse_step2_bls12_377_enqueue",
    "panda_ntt_execute_bls12_377_multi", "panda_ntt_execute_bls12_377_inverse_multi", "panda_ntt_execute_bls12_377_multi_batch", "panda_ntt_execute_bls12_377_inverse_multi_batch",
    "panda_ntt_execute_bls12_381_bitrev_out", "panda_ntt_execute_bls12_381_inverse_bitrev_in", "panda_ntt_execute_bls12_381_coset", "panda_ntt_execute_bls12_381_coset_inverse", "panda_ntt_slab_step1_bls12_381_enqueue", "panda_ntt_slab_step2_bls12_381_enqueue", "panda_ntt_slab_inverse_step1_bls12_381_enqueue", "panda_ntt_slab_inverse_step2_bls12_381_enqueue", "panda_ntt_execute_bls12_381_multi", "panda_ntt_execute_bls12_381_inverse_multi", "panda_ntt_execute_bls12_381_multi_batch", "panda_ntt_execute_bls12_381_inverse_multi_batch",
]
ALL_SYMBOLS = REFERENCE_SYMBOLS + RUST_ONLY_SYMBOLS + ADDITIVE_SYMBOLS

_lib = None


def load() -> C.CDLL:
    """dlopen the HIP library.  torch, when installed, is imported first so that the process ends up
    with ONE libamdhip64 (torch bundles its own copy under the same SONAME)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: build the HIP extension first (python -c 'import __graft_entry__ as g; g.build()' "
            "or make -C panda_amd/csrc).  There is no CPU fallback for the product path.")
    try:
        import torch  # noqa: F401
    except Exception:  # pragma: no cover - torch is plumbing, not a requirement of the library
        pass
    lib = C.CDLL(LIB_PATH)
    u, sz, vp = C.c_uint, C.c_size_t, C.c_void_p
    sig = {
        "panda_get_device_number": [C.POINTER(C.c_int)], "panda_get_device": [C.POINTER(C.c_int)], "panda_set_device": [C.c_int],
        "panda_stream_create": [C.POINTER(PandaStream), C.c_bool], "panda_stream_wait_event": [PandaStream, PandaEvent],
        "panda_stream_sync": [PandaStream], "panda_stream_synchronize": [PandaStream], "panda_stream_query": [PandaStream],
        "panda_stream_destroy": [PandaStream], "panda_launch_host_fn": [PandaStream, vp, vp],
        "panda_event_create": [C.POINTER(PandaEvent), C.c_bool, C.c_bool], "panda_event_record": [PandaEvent, PandaStream],
        "panda_event_sync": [PandaEvent], "panda_event_query": [PandaEvent], "panda_event_destroy": [PandaEvent],
        "panda_mem_get_info": [C.POINTER(sz), C.POINTER(sz)], "panda_malloc": [C.POINTER(vp), sz], "panda_malloc_host": [C.POINTER(vp), sz],
        "panda_free": [vp], "panda_free_host": [vp], "panda_host_register": [vp, sz], "panda_host_unregister": [vp],
        "panda_memcpy": [vp, vp, sz], "panda_memcpy_async": [vp, vp, sz, PandaStream], "panda_memset": [vp, C.c_int, sz],
        "panda_memset_async": [vp, C.c_int, sz, PandaStream], "panda_mem_pool_create": [C.POINTER(PandaMemPool), C.c_int],
        "panda_mem_pool_destroy": [PandaMemPool], "panda_malloc_from_pool_async": [C.POINTER(vp), sz, PandaMemPool, PandaStream],
        "panda_free_async": [vp, PandaStream], "panda_msm_setup_bn254": [], "panda_msm_execute_bn254": [MSMConfiguration],
        "panda_msm_execute_bn254_host": [MSMConfiguration], "panda_msm_tear_down": [], "panda_ntt_setup_bn254": [vp],
        "panda_ntt_execute_bn254": [NTTConfiguration], "panda_ntt_tear_down": [], "panda_ntt_execute_bn254_v1": [NttconfigurationV1],
        "panda_device_enable_peer_access": [C.c_int], "panda_device_disable_peer_access": [C.c_int],
        "panda_msm_execute_from_host": [u, MSMConfiguration, vp, u, PandaStream],
        "panda_msm_register_bases": [u, vp, u, PandaStream], "panda_msm_unregister_bases": [vp], "panda_msm_precompute_bases": [u, vp, u, u, PandaStream],
        "panda_msm_registered_info": [vp, C.POINTER(u), C.POINTER(u), C.POINTER(sz)], "panda_msm_set_chunk_entries": [u], "panda_msm_set_overlap": [u, u], "panda_msm_set_accumulate_variant": [u], "panda_msm_set_wide_merge": [u], "panda_msm_set_reduce_group": [u], "panda_msm_plain_window_plan": [u, u, C.POINTER(u), C.POINTER(u)], "panda_msm_verify_registered": [vp, PandaStream], "panda_msm_set_paranoid": [u], "panda_msm_set_phase_timing": [u], "panda_msm_setup_bls12_377": [], "panda_msm_execute_bls12_377": [MSMConfiguration], "panda_msm_execute_bls12_377_host": [MSMConfiguration],
        "panda_msm_set_window_bits": [u], "panda_msm_last_phase_ms": [C.POINTER(C.c_float)], "panda_ntt_last_device_ms": [C.POINTER(C.c_float)], "panda_ntt_pass_plan": [u, C.POINTER(C.c_uint), C.POINTER(C.c_uint)], "panda_ntt_set_streamed_tables": [u], "panda_ntt_execute_bn254_inverse": [NttconfigurationV1], "panda_ntt_execute_bls12_377_v1": [NttconfigurationV1], "panda_ntt_execute_bls12_377_inverse": [NttconfigurationV1],
        "panda_msm_combine_bn254": [vp, u, C.c_int, vp], "panda_msm_combine_bls12_377": [vp, u, C.c_int, vp], "panda_msm_combine_bls12_381": [vp, u, C.c_int, vp],
        "panda_msm_setup_bn254_g2": [], "panda_msm_execute_bn254_g2": [MSMConfiguration], "panda_msm_execute_bn254_g2_host": [MSMConfiguration],
        "panda_msm_combine_bn254_g2": [vp, u, C.c_int, vp],
        "panda_msm_setup_bls12_381": [], "panda_msm_execute_bls12_381": [MSMConfiguration], "panda_msm_execute_bls12_381_host": [MSMConfiguration],
        "panda_ntt_execute_bls12_381_v1": [NttconfigurationV1], "panda_ntt_execute_bls12_381_inverse": [NttconfigurationV1],
        "panda_ntt_execute_bn254_bitrev_out": [NttconfigurationV1], "panda_ntt_execute_bn254_inverse_bitrev_in": [NttconfigurationV1],
        "panda_ntt_execute_bn254_coset": [NttconfigurationV1, vp], "panda_ntt_execute_bn254_coset_inverse": [NttconfigurationV1, vp],
        "panda_ntt_slab_step1_bn254": [NttSlabConfiguration], "panda_ntt_slab_step2_bn254": [NttSlabConfiguration],
        "panda_ntt_slab_step1_bn254_enqueue": [NttSlabConfiguration], "panda_ntt_slab_step2_bn254_enqueue": [NttSlabConfiguration],
        "panda_ntt_slab_inverse_step1_bn254_enqueue": [NttSlabConfiguration], "panda_ntt_slab_inverse_step2_bn254_enqueue": [NttSlabConfiguration],
        "panda_gen_scalars": [u, C.c_uint64, C.c_uint64, C.c_uint64, vp, PandaStream], "panda_gen_bases": [u, C.c_uint64, C.c_uint64, C.c_uint64, vp, PandaStream],
        "panda_debug_field_op": [u, u, vp, vp, vp, sz, PandaStream], "panda_debug_curve_op": [u, u, vp, vp, vp, sz, PandaStream],
        "panda_ntt_table_builds": [C.POINTER(C.c_uint64)], "panda_msm_set_chunk_first": [u], "panda_set_clock_stamps": [u], "panda_msm_last_clock": [C.POINTER(C.c_uint64)],
        "panda_ntt_last_clock": [C.POINTER(C.c_uint64)], "panda_clock_stamp": [PandaStream, vp], "panda_clock_delta": [vp, vp, C.POINTER(C.c_uint64)],
        "panda_multi_gpu_create": [C.POINTER(PandaMultiGpu), C.POINTER(C.c_int), u, u], "panda_multi_gpu_destroy": [PandaMultiGpu],
        "panda_multi_gpu_device_count": [PandaMultiGpu, C.POINTER(u)],
        "panda_msm_execute_bn254_multi": [PandaMultiGpu, C.POINTER(MSMConfiguration), vp], "panda_msm_execute_bls12_377_multi": [PandaMultiGpu, C.POINTER(MSMConfiguration), vp],
        "panda_msm_execute_bn254_from_host_multi": [PandaMultiGpu, C.POINTER(MSMConfiguration), C.POINTER(vp), u, vp],
        "panda_msm_execute_bls12_377_from_host_multi": [PandaMultiGpu, C.POINTER(MSMConfiguration), C.POINTER(vp), u, vp],
        "panda_msm_execute_bls12_381_multi": [PandaMultiGpu, C.POINTER(MSMConfiguration), vp], "panda_msm_execute_bn254_g2_multi": [PandaMultiGpu, C.POINTER(MSMConfiguration), vp],
        "panda_msm_execute_bls12_381_from_host_multi": [PandaMultiGpu, C.POINTER(MSMConfiguration), C.POINTER(vp), u, vp],
        "panda_msm_execute_bn254_g2_from_host_multi": [PandaMultiGpu, C.POINTER(MSMConfiguration), C.POINTER(vp), u, vp],
        "panda_ntt_execute_bn254_multi": [PandaMultiGpu, C.POINTER(NttSlabConfiguration)], "panda_ntt_execute_bn254_inverse_multi": [PandaMultiGpu, C.POINTER(NttSlabConfiguration)],
        "panda_ntt_execute_bn254_multi_batch": [PandaMultiGpu, C.POINTER(NttSlabConfiguration), C.c_uint], "panda_ntt_execute_bn254_inverse_multi_batch": [PandaMultiGpu, C.POINTER(NttSlabConfiguration), C.c_uint],
        "panda_multi_gpu_last_phase_ms": [PandaMultiGpu, u, C.POINTER(C.c_float)],
        "panda_ntt_execute_bls12_377_bitrev_out": [NttconfigurationV1], "panda_ntt_execute_bls12_377_inverse_bitrev_in": [NttconfigurationV1],
        "panda_ntt_execute_bls12_377_coset": [NttconfigurationV1, vp], "panda_ntt_execute_bls12_377_coset_inverse": [NttconfigurationV1, vp],
        "panda_ntt_slab_step1_bls12_377_enqueue": [NttSlabConfiguration], "panda_ntt_slab_step2_bls12_377_enqueue": [NttSlabConfiguration],
        "panda_ntt_slab_inverse_step1_bls12_377_enqueue": [NttSlabConfiguration], "panda_ntt_slab_inverse_step2_bls12_377_enqueue": [NttSlabConfiguration],
        "panda_ntt_execute_bls12_377_multi": [PandaMultiGpu, C.POINTER(NttSlabConfiguration)], "panda_ntt_execute_bls12_377_inverse_multi": [PandaMultiGpu, C.POINTER(NttSlabConfiguration)],
        "panda_ntt_execute_bls12_377_multi_batch": [PandaMultiGpu, C.POINTER(NttSlabConfiguration), C.c_uint],
        "panda_ntt_execute_bls12_377_inverse_multi_batch": [PandaMultiGpu, C.POINTER(NttSlabConfiguration), C.c_uint],
    }
    for name in [n for n in list(sig) if "bls12_377" in n and n.replace("bls12_377", "bls12_381") in ADDITIVE_SYMBOLS]:
        sig.setdefault(name.replace("bls12_377", "bls12_381"), sig[name])
    for name, args in sig.items():
        try:
            fn = getattr(lib, name)
        except AttributeError:
            if LIB_PATH.endswith(os.path.join("csrc", "libpanda-cuda.so")):
                raise  # the in-tree build must export everything
            continue   # tools/*_bench.py pointed LIB_PATH at an earlier round's build (PANDA_LIB) for an A/B run
        fn.argtypes = args
        fn.restype = C.c_uint  # PandaError = c_uint, gpu_ffi/mod.rs:8
    lib.panda_msm_phase_name.argtypes = [u]
    lib.panda_msm_phase_name.restype = C.c_char_p
    lib.panda_version.argtypes = []
    lib.panda_version.restype = C.c_char_p
    _lib = lib
    return lib


def check(code: int, kind: str) -> None:
    if code != 0:
        raise PandaGpuError(kind, code)
