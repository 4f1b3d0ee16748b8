// Library options: the tuning knobs that select a kernel FORM (never a result: every form is bit-identical, the tests
// run each of them).  One process-wide table of atomics, written by nfa_set_option (include/nerfacc_hip.h) and read by the
// launch planners; the NFA_* environment variables seed it ONCE, when the library is loaded, so that
// `NFA_EMIT=rays python tools/...` still works while a host process that calls setenv() later changes nothing and no
// planner calls getenv() on the per-call path (getenv is not safe against a concurrent setenv and
// the header promises re-entrancy).
#pragma once
#include <atomic>
#include <cstdint>

namespace nfa {

enum Option : int {
    OPT_E = 0,           // samples per lane of the tiled streaming kernels: 1 | 2 | 4
    OPT_TILE,            // nominal tile of those kernels, a multiple of 64 * E
    OPT_SPLIT_P,         // lanes per ray of the one-level count pass: 1 | 2 | 4 | 8 | 16
    OPT_SEG_P,           // several levels, cone_angle = 0: 8 | 32
    OPT_CONE_P,          // cone-angle count pass: 8 | 16 | 32 | 64
    OPT_CONE,            // 0: the general lane-per-ray kernel for cone_angle != 0
    OPT_SPLIT_L2,        // 16 lanes per ray: grid image in LDS (0) / read from L2 (1)
    OPT_COUNT_L2,        // lane-per-ray count and fill kernels: image in LDS (0) / from L2 (1)
    OPT_EMIT,            // emit pass: 1 = rays (16 lanes per ray), 2 = samples (a lane per sample), 3 = tiles (a wave expands a block of rays)
    OPT_SCAN_RW,         // packed scan: rows per wave 4 | 16
    OPT_SPLIT_BLK,       // workgroup size of the 16-lane count pass: 256 | 512
    OPT_SPLIT_XT,        // 0: no crossing-time arrays in the 512-thread form
    OPT_SEGMENTS,        // 0: several levels take the lane-per-ray count pass
    OPT_SPLIT_CAP,       // grids read from L2: entries of a part's boundary list, 16 | 24 | 32 (16 / 24: 8 and 16 lanes per ray only)
    OPT_EMIT_RB,         // tile form of the emit pass: log2 of the rays a wave takes (0 ... 6)
    OPT_CHUNK_PREFETCH,  // 0: sample_occgrid of the torch extension never launches the next ray slice's count pass ahead of its call
    OPT_SPECULATIVE_EMIT,// 0: the extension's sample_occgrid launches the emit pass after the read-back
    OPT_SKIP,            // lane-per-ray lattice count pass: 0 = voxel by voxel (rounds 1-4), 1 = empty-space macro steps (brick distances from L2)
    OPT_FUSED_SAMPLE,    // nfa_traverse_sample's single-launch form: 0 never, 1 up to 80 samples of capacity per ray (default), 2 always inside its window
    OPT_FUSED_VIS,       // 1: the visibility filter of small calls as one launch (static one-pass kernel); default 0 (measured slower)
    OPT_FOLD_FILL,       // 0: nfa_rendering_fwd always fills the rays without a sample with a launch of its own (1: inside its kernel up to 2^20 samples)
    OPT_SYNC_SPIN_US,    // bound of every look-back's wait inside the single-launch forms, microseconds (default 2000; 0: give up at once — tests)
    OPT_COUNT
};

constexpr int64_t kOptUnset = INT64_MIN;

struct OptionTable {
    std::atomic<int64_t> v[OPT_COUNT];
};
OptionTable &option_table();        // options.hip; seeded from the environment on first use

// value of an option, or `dflt` while it is unset (= "auto")
inline int64_t opt(Option o, int64_t dflt) {
    const int64_t x = option_table().v[o].load(std::memory_order_relaxed);
    return x == kOptUnset ? dflt : x;
}
inline bool opt_is_set(Option o) { return option_table().v[o].load(std::memory_order_relaxed) != kOptUnset; }

}  // namespace nfa
