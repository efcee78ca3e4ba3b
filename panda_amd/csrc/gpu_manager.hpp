// gpu_manager.hpp -- C++ mirror of the reference's Rust host layer (src/gpu_manager/{wrapper,unit,common}.rs)
// over the C ABI in include/panda_interface.h.
//
// The reference's host side is compiled Rust; this image has no rustc, so the same layer is written in C++:
// same type and function names, same ownership (the manager owns device id, memory pool and four streams,
// wrapper.rs:8-19), same staging protocol (H2D on the h2d stream, event, execute on the exec stream, D2H,
// free; unit.rs:10-101) and the same error enum (gpu_ffi/common.rs:5-38) returned instead of Result<_, _>.
// Deliberate differences: the pinned result buffer is freed (the reference leaks it, unit.rs:67-100), and
// cached scalars stay valid across calls because the library does not overwrite them.
#pragma once
#include <cstdint>
#include <cstring>
#include <vector>

#include "../../include/panda_interface.h"

namespace panda_host {

enum class PandaGpuError { // gpu_ffi/common.rs:5-38, plus Ok
    Ok = 0,
    GetDeviceCountError,
    SetDeviceError,
    DeviceGetDeviceMemoryInfoError,
    CreateContextError,
    InitUnitTypeError,
    MSMBasesAddrError,
    NTTOmegaAddrError,
    SetBasesErr,
    SchedulingErr,
    GetExponentAddressErr,
    GetResultAddressesErr,
    StartProcessingErr,
    FinishProcessingErr,
    DestroyContextErr,
    BasesIndexErr,
    MemPoolCreateErr,
    AsyncPoolMallocErr,
    AsyncMemcopyErr,
    NttExecErr,
    StremCreateErr,
    StreamDestroyErr,
    StreamWaitEventErr,
    StreamSyncErr,
    EventCreateErr,
    EventRecordErr,
    EventDestroyErr,
    EventSyncErr,
};

enum class PandaGpuManagerInitUnitType { None, MSM, NTT, ALL }; // wrapper.rs:23-29

constexpr size_t FIELD_ELEMENT_LEN = 32; // gpu_manager/mod.rs:14

struct Bytes { // &[u8]
    const uint8_t *data;
    size_t len;
};

struct PandaDeviceInfo {
    uint64_t free, total;
};

inline uint32_t log_2(size_t num) // gpu_manager/common.rs:5-15
{
    uint32_t pow = 0;
    while (((size_t)1 << (pow + 1)) <= num) pow++;
    return pow;
}

PandaGpuError get_device_number(int *count);              // wrapper.rs:315-323
PandaGpuError device_info(int device_id, PandaDeviceInfo *out); // wrapper.rs:325-338
PandaGpuError set_device(size_t device_id);               // wrapper.rs:340-347

class PandaGpuManager {
  public:
    // PandaGpuManager::new (wrapper.rs:32-53)
    static PandaGpuError create(size_t device_id, PandaGpuManager *out);
    // PandaGpuManager::init_all (wrapper.rs:55-113)
    static PandaGpuError init_all(size_t device_id, PandaGpuManagerInitUnitType type, const std::vector<Bytes> *bases, const Bytes *omega,
                                  PandaGpuManager *out);
    static PandaGpuError init_hardware(size_t device_id, panda_mem_pool *pool);          // wrapper.rs:115-120
    static PandaGpuError init_msm(const std::vector<Bytes> &bases, std::vector<void *> *d_ptrs); // wrapper.rs:122-152
    static PandaGpuError init_msm_cached_bases(Bytes bases, void **d_ptr);               // wrapper.rs:154-169
    static PandaGpuError init_msm_cached_scalars(Bytes scalars, void **d_ptr);           // wrapper.rs:171-186
    static PandaGpuError init_ntt(Bytes omega);                                          // wrapper.rs:199-210
    // additive: let the library keep the radix-converted copy of cached base set `index` (2^log_n points) between calls
    PandaGpuError register_cached_bases(size_t index, uint32_t log_n);
    // additive: the same plus precomputed window tables (panda_msm_precompute_bases); window_bits 0 = built-in policy
    PandaGpuError precompute_cached_bases(size_t index, uint32_t log_n, uint32_t window_bits = 0);

    void set_config(panda_msm_result_coordinate_type t) { msm_result_coordinate_type_ = t; } // wrapper.rs:212-214
    panda_mem_pool get_mem_pool() const { return mem_pool_; }
    panda_stream get_stream() const { return default_stream_; }
    panda_stream get_h2d_stream() const { return h2d_stream_; }
    panda_stream get_d2h_stream() const { return d2h_stream_; }
    panda_stream get_exec_stream() const { return exec_stream_; }
    panda_msm_result_coordinate_type get_msm_result_coordinate_type() const { return msm_result_coordinate_type_; }
    void *get_params_bases_ptr_mut(size_t index) const { return index < d_bases.size() ? d_bases[index] : nullptr; }
    void *get_params_scalars_ptr_mut(size_t index) const { return index < d_scalars.size() ? d_scalars[index] : nullptr; }
    size_t get_params_scalars_len(size_t index) const { return index < scalars_len.size() ? scalars_len[index] : 0; }
    PandaGpuError wait_h2d() const; // wrapper.rs:260-266
    PandaGpuError wait_exec() const; // wrapper.rs:268-273
    PandaGpuError sync() const;      // wrapper.rs:285-291
    size_t device_id() const { return device_id_; }
    PandaGpuError deinit();          // wrapper.rs:297-312 (also releases streams and the library's scratch)

    std::vector<void *> d_bases, d_scalars; // pub in the reference (wrapper.rs:15-17)
    std::vector<size_t> scalars_len;
    std::vector<void *> registered_bases;

  private:
    size_t device_id_ = 0;
    panda_mem_pool mem_pool_{};
    panda_stream default_stream_{}, h2d_stream_{}, d2h_stream_{}, exec_stream_{};
    panda_msm_result_coordinate_type msm_result_coordinate_type_ = JACOBIAN;
};

// gpu_manager/common.rs:17-76
PandaGpuError malloc_from_pool_async(void **ptr, size_t size, panda_mem_pool pool, panda_stream stream);
PandaGpuError memcpy_async(void *dst, const void *src, size_t size, panda_stream stream);
PandaGpuError free_async(void *ptr, panda_stream stream);
PandaGpuError memory_alloc_and_copy(const PandaGpuManager &gm, Bytes h_values, panda_stream stream, void **d_values);

unsigned pipeline_ranges(uint32_t log_n); // additive: ranges the with_cached_bases call pipelines its upload in (1 = not pipelined)

// gpu_manager/unit.rs -- results are 96 bytes X||Y||Z (Montgomery limbs)
PandaGpuError panda_msm_bn254_gpu(const PandaGpuManager &gm, Bytes scalars, Bytes bases, std::vector<uint8_t> *result);                    // :10-101
PandaGpuError panda_msm_bn254_gpu_with_cached_bases(const PandaGpuManager &gm, Bytes scalars, size_t bases_index, std::vector<uint8_t> *result);   // :103-188
// additive: a run of MSMs over one cached base set, scalar uploads overlapped with execution (h2d / exec streams + events)
PandaGpuError panda_msm_bn254_gpu_with_cached_bases_batched(const PandaGpuManager &gm, const std::vector<Bytes> &scalars, size_t bases_index,
                                                            std::vector<std::vector<uint8_t>> *results);
PandaGpuError panda_msm_bn254_gpu_with_cached_scalars(const PandaGpuManager &gm, size_t scalars_index, Bytes bases, std::vector<uint8_t> *result); // :190-275
PandaGpuError panda_msm_bn254_gpu_with_cached_input(const PandaGpuManager &gm, size_t scalars_index, size_t bases_index, std::vector<uint8_t> *result); // :277-361
PandaGpuError panda_msm_bn254_gpu_host(const PandaGpuManager &gm, Bytes scalars, Bytes bases, std::vector<uint8_t> *result);               // :363-416
PandaGpuError panda_ntt_bn254_gpu(const PandaGpuManager &gm, uint8_t *scalars, size_t len, uint32_t log_n);                                // :418-479
PandaGpuError panda_ntt_bn254_gpu_v1(const PandaGpuManager &gm, uint8_t *scalars, size_t len, Bytes omega, uint32_t log_n);                // :481-543
// additive: inverse transform with n^-1 fused
PandaGpuError panda_intt_bn254_gpu(const PandaGpuManager &gm, uint8_t *scalars, size_t len, Bytes omega, uint32_t log_n);

// Additive (no reference counterpart: wrapper.rs:38 opens ONE device): the sharded calls from host slices, one process.  One
// PandaGpuManager per device stages that device's share exactly as the single-GPU calls do; the sharded operation itself is one C call
// on a panda_multi_gpu handle (csrc/multi_gpu.hip: worker thread + RCCL communicator per device).
class PandaMultiGpuManager {
  public:
    // transport: PANDA_MULTI_RCCL (one rank per device) or PANDA_MULTI_LOOPBACK (device copies; devices may repeat)
    static PandaGpuError create(const std::vector<int> &devices, unsigned transport, PandaMultiGpuManager *out);
    PandaGpuError deinit();
    size_t ranks() const { return managers.size(); }
    // cached bases: rank d keeps points [d n/G, (d+1) n/G) on its device, optionally with precomputed window tables
    PandaGpuError init_msm_cached_bases(Bytes bases, bool tables);
    // MSM of `scalars` against the cached bases: rank d gets its slice of the scalars; result = 96 bytes X||Y||Z
    PandaGpuError msm_bn254_with_cached_bases(Bytes scalars, std::vector<uint8_t> *result);
    // forward transform of 2^log_n elements given in natural order; `data` receives y in natural order (the slabs are decimated
    // on the way in and the output layout y[k1 m + q m/G + k2'] at rank q's [k1][k2'] is undone on the way out)
    PandaGpuError ntt_bn254(uint8_t *data, size_t len, Bytes omega, uint32_t log_n);
    // the same for several polynomials of one size behind ONE call (panda_ntt_execute_bn254_multi_batch: the exchange of polynomial t runs
    // beside the kernels of its neighbours); every polys[t] is transformed in place
    PandaGpuError ntt_bn254_batch(const std::vector<uint8_t *> &polys, size_t len, Bytes omega, uint32_t log_n);

    std::vector<PandaGpuManager> managers;
    panda_multi_gpu handle{};

  private:
    std::vector<int> devices_;
    std::vector<void *> d_bases_; // per rank
    uint32_t bases_log_per_ = 0;
    bool tables_ = false;
};

} // namespace panda_host
