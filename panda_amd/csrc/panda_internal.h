// panda_internal.h -- helpers shared by the translation units of libpanda-cuda (HIP build).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "../../include/panda_interface.h"

// Early-return on a failing runtime call, printing where (the reference's HANDLE_RESULT_CUDA,
// src/cuda/core/common/common.cuh:14-22, does the same on stdout; here it goes to stderr: callers such as bench.py
// speak JSON lines on stdout).
#define PANDA_TRY(expr)                                                                                        \
    do {                                                                                                       \
        hipError_t panda_err__ = (expr);                                                                       \
        if (panda_err__ != hipSuccess) {                                                                       \
            fprintf(stderr, "[panda-hip] error %d (%s) at %s:%d: %s\n", (int)panda_err__,                      \
                    hipGetErrorName(panda_err__), __FILE__, __LINE__, #expr);                                  \
            return panda_err__;                                                                                \
        }                                                                                                      \
    } while (0)

namespace panda {

// Grow-only device scratch arena, one per host thread and device.  The reference allocates and
// frees seven scratch buffers per MSM call with cudaMallocAsync (msm_cuda.cuh:604-610,757-763);
// with 288 GB of HBM it is cheaper to keep one arena alive between calls.  Released by
// panda_msm_tear_down() / panda_ntt_tear_down().
struct Arena {
    void *base = nullptr;
    size_t capacity = 0;
    size_t used = 0;
    int device = -1;

    hipError_t reserve(size_t bytes);
    void reset() { used = 0; }
    void *take(size_t bytes) // 256-byte aligned carve; reserve() must have been called with the total
    {
        size_t off = (used + 255) & ~(size_t)255;
        if (off + bytes > capacity) return nullptr;
        used = off + bytes;
        return (char *)base + off;
    }
    hipError_t release();
    Arena() = default;
    Arena(const Arena &) = delete;
    Arena &operator=(const Arena &) = delete;
    ~Arena(); // a host thread that exits without tear_down does not leak its scratch
};

Arena &thread_arena();
hipError_t release_thread_arena();
// a second stream of this host thread on the current device (created on first use, destroyed with the arena): the other lane of an
// MSM that runs in point ranges
hipError_t thread_helper_stream(hipStream_t *out);

// Pinned host memory of this host thread that kernels write into directly (the stale-registration flag, the window sums or the
// finished result of an MSM): what a call hands back to its host side arrives with the kernel that produced it, without a
// device-to-host copy of its own behind it (each cost ~10 us of a 1.5 ms call).  MAILBOX_WORDS u32, allocated on first use.
constexpr size_t MAILBOX_WORDS = 16384;
hipError_t thread_mailbox(uint32_t **host_words);

// Clock stamps (measurement only, off unless panda_set_clock_stamps(1)): a marker kernel on the launch stream reads s_memtime (shader
// cycles) and s_memrealtime (100 MHz) directly before and directly behind the kernel(s) of interest, so a run's record carries CYCLES and
// the clock they ran at beside the milliseconds -- a slow device and a slower kernel read differently (cycles are the code's, MHz the
// box's).  What the first version of this got wrong (profiles/r06_clock_stamps.txt): s_memtime is NOT one counter per XCD -- two waves of
// one XCD on different shader engines read values ~10^8 apart -- so a stamp is only ever compared with a stamp taken on the SAME CU:
// the marker runs 2048 one-wave workgroups (every CU gets several) and wave w stores {s_memtime, s_memrealtime} into the slot of its CU,
// slot = XCC_ID * 256 + HW_ID[15:8] (shader engine, array, CU).  The XCDs hold different clocks (+-4 % on one device at one moment), and
// the workgroups of a grid are dealt round the XCDs, so a kernel ends when the SLOWEST-clocked XCD is through: that XCD's cycle count is
// the work, the others idle at the end and show more.  A stamp block is CLOCK_STAMP_SLOTS x 2 u64; blocks live in pinned host memory of
// the calling thread (thread_stamp_blocks: four of them, MSM before / behind, NTT before / behind).
constexpr unsigned CLOCK_STAMP_SLOTS = 2048;
struct ClockDelta {
    uint64_t cycles = 0;       // of the XCD with the fewest (the slowest clock: the one the kernel waits for); median over that XCD's CUs
    uint64_t cycles_mean = 0;  // mean over the XCDs: what rocprofv3's GRBM_GUI_ACTIVE / 8 shows for the same launch
    uint64_t ticks = 0;        // 10 ns, median over all paired CUs
    unsigned xcds = 0;         // XCDs with at least one CU stamped on both sides
    uint64_t per_xcd[8] = {};  // cycles of every XCD (0: none paired)
};
bool clock_stamps_enabled();
void set_clock_stamps_enabled(bool on);
hipError_t thread_stamp_blocks(uint64_t **host_blocks); // 4 blocks of CLOCK_STAMP_SLOTS x 2 u64, allocated on first use
hipError_t enqueue_clock_stamp(hipStream_t s, uint64_t *block); // zeroes nothing: the host clears the block before the call
ClockDelta clock_delta(const uint64_t *before, const uint64_t *after);
ClockDelta &thread_msm_clock(); // of the last MSM / NTT of this host thread (zero when stamps are off)
ClockDelta &thread_ntt_clock();
void clock_delta_out(const ClockDelta &d, uint64_t *out); // the PANDA_CLOCK_WORDS u64 of panda_*_last_clock

// drops every cached-bases registration whose buffer lies in the allocation `ptr` belongs to (msm.hip); called by
// panda_free / panda_free_async before the memory goes back to the allocator
void registry_forget_allocation(const void *ptr);

static inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

// Boundary check for caller buffers: true when `ptr` lies in an allocation made through this library (panda_malloc,
// panda_malloc_from_pool_async: shim.hip keeps their extents) and that allocation ends before ptr + bytes.  A caller that passes a
// buffer shorter than log_n implies (round 2: a 4 KB buffer with log_n = 10 on a 128-byte-per-point curve) would otherwise make a
// kernel read past it and take the process down; with this the entry point answers panda_error_invalid_value.  Pointers from other
// allocators pass unchecked: the runtime's own description of them (hipMemGetAddressRange, which round 3 used) is one mapped CHUNK for
// virtual-memory mappings such as torch's expandable segments, so a valid buffer spanning several chunks would be refused -- the check
// only ever refuses what is provably too short.  No runtime call: a lookup in a small ordered map.  Extents leave the map with
// panda_free / panda_free_async and, for pool memory, with panda_mem_pool_destroy; memory released behind the library's back (hipFree, a
// pool trim) keeps a stale extent until its address is tracked again -- release what panda_malloc* gave through panda_free*.
bool extent_too_short(const void *ptr, size_t bytes);
void track_allocation(const void *ptr, size_t bytes, const void *pool = nullptr);
void untrack_pool(const void *pool); // panda_mem_pool_destroy: the pool's extents go with it
void untrack_allocation(const void *ptr);

} // namespace panda
