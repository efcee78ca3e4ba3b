// msm_sort.h -- scalars -> signed window digits -> per-bucket point lists (the front half of the MSM pipeline).
// Implemented in msm_sort.hip; consumed by msm.hip.  Replaces the reference's calc_lens / allo_arrs / fill_arrs
// global-atomic passes (src/cuda/core/unit/msm/msm_cuda.cuh:159-282).
#pragma once
#include "panda_internal.h"

namespace panda {

// BITS+1 scalar bits (one spare for the signed-digit carry) cut into W windows whose widths differ by at most one.
constexpr unsigned MAX_WINDOWS = 72; // 4-bit windows over 256 + 8 scalar bits at most
struct WindowPlan {
    unsigned W;
    unsigned char width[MAX_WINDOWS];
    unsigned short lo[MAX_WINDOWS];
};

WindowPlan make_window_plan(unsigned total_bits, unsigned c);
// plan for scalar field `fr` (0 = BN254 Fr, 1 = BLS12-377 Fr) whose top window can never carry out
WindowPlan make_safe_window_plan(unsigned fr, unsigned c);

struct SortResult {
    const uint32_t *off;    // [lists][NB + 1] start of each bucket's run, off[NB] = number of entries of the list
    const uint32_t *sorted; // [lists][stride] entries: bit 31 = negate, low bits = index into the base array
    unsigned lists;         // W (plain) or 1 (tabled)
    unsigned NB;            // buckets per list
    uint64_t stride;        // n (plain) or W*n (tabled)
};

// Where the sorted entries point.  Default ({0, 0}): at rows [0, n) of an n-row base array / of tables of n rows each.  A call that
// runs in point-range chunks sorts 2^log_n scalars that belong to rows [row0, row0 + 2^log_n) of tables of 2^row_shift rows.
struct SortPlacement {
    unsigned row_shift; // 0 = default placement
    uint32_t row0;
};

// Stale-registration check carried by the first kernel of the sort (msm.hip, "Staleness"): REG_SAMPLES rows of the caller's wire
// buffer are compared with the copies taken at registration; *flag_dev = 0 / 1 either way, *flag_host = 1 on a mismatch.
// wire == nullptr: nothing to check.
constexpr unsigned REG_SAMPLES = 64;
struct SampleCheck {
    const uint32_t *wire, *samples;
    uint64_t n;
    unsigned row_words;
    uint32_t *flag_dev, *flag_host;
};
// sample t is row 0, row n-1, or a fixed pseudo-random row
static __host__ __device__ inline uint64_t sample_row(unsigned t, uint64_t n) { return t == 0 ? 0 : (t == 1 ? n - 1 : (((uint64_t)t * 0x9E3779B97F4A7C15ull) >> 20) & (n - 1)); }

// optional events recorded on `stream` between the phases (may be null)
struct SortEvents {
    hipEvent_t digits_done, partition_done;
};

// Plain mode: W independent lists (one per window), entries index the caller's n bases.  Windows of up to 16 bits take a two-level
// sort, wider ones (up to 2^19 buckets per window) the three levels of the tabled mode with one list per window.
bool msm_sort_plain_supported(unsigned log_n, const WindowPlan &plan);
size_t msm_sort_plain_bytes(unsigned log_n, const WindowPlan &plan);
hipError_t msm_sort_plain(hipStream_t stream, Arena &arena, unsigned fr, const void *scalars, unsigned log_n, const WindowPlan &plan, SortEvents ev,
                          SortResult *out, SortPlacement place = SortPlacement{0, 0}, SampleCheck check = SampleCheck{});

// Overlap of the sort's back half with the accumulation (tabled mode; replaces the serial phases of msm_cuda.cuh:611-755).  Level 1
// partitions every window's entries by the TOP bits of the bucket id, so everything behind it -- level 2, level 3, k_accumulate -- can
// run per range of level-1 partitions: the buckets are cut at a partition boundary into a FRONT part and the REST.  The front's levels
// 2 and 3 run on the caller's stream; the rest's start on `helper` once the front is sorted (`ev_front`) -- beside whatever the caller
// enqueues behind the sort, i.e. the accumulation of the front -- and `rest_done` is recorded on `helper` when the whole list is sorted.
// The entries before pos[cut_cell] and the bucket offsets up to and including off[cut_bucket] are final in the caller's stream order when
// msm_sort_tabled returns.  Measured on MI355X and not the library's policy (profiles/r05_overlap_sort_accumulate.txt).
struct SortSplit {
    hipStream_t helper;          // in: second stream (nullptr: no split)
    unsigned front_of_128;       // in: size of the front part, in 1/128 of the bucket space (rounded to what the geometry allows)
    hipEvent_t ev_front, rest_done; // in: two events the caller owns (no timing needed)
    bool active;                 // out: the sort was split (false: sizes too small for it; everything ran on the caller's stream)
    const uint32_t *pos;         // out: position of every level-3 cell's first entry in the list, pos[cells] = number of entries
    unsigned cells, cut_cell;    // out: cells in all, first cell of the rest
    unsigned cut_bucket;         // out: first bucket of the rest
};

// Tabled mode: one list over all windows; entry index = k * n + i names row i of table k (= 2^lo[k] * base i), so
// every window falls into the same 2^(c-1) buckets and the window sums need no Horner step.
bool msm_sort_tabled_supported(unsigned log_n, const WindowPlan &plan);
// level-3 merge of the tabled sort: 0 = the wide variant (cells of up to 32 k entries read once) for the dense half of the cells where the
// plan calls for it and 256-thread workgroups for the sparse half, 1 = the wide variant for every cell, 2 = neither (large cells then take
// the two-pass path), 3 = the 256-thread variant for every cell
void msm_sort_set_wide_merge(unsigned mode);
size_t msm_sort_tabled_bytes(unsigned log_n, const WindowPlan &plan);
hipError_t msm_sort_tabled(hipStream_t stream, Arena &arena, unsigned fr, const void *scalars, unsigned log_n, const WindowPlan &plan, SortEvents ev,
                           SortResult *out, SortPlacement place = SortPlacement{0, 0}, SampleCheck check = SampleCheck{}, SortSplit *split = nullptr);

} // namespace panda
