// common.hpp — shared host/device helpers for libnerfacc_hip.so (gfx950 / wave64 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>

#include "../../include/nerfacc_hip.h"

#define NFA_EXPORT extern "C" __attribute__((visibility("default")))

namespace nfa {

// ----------------------------------------------------------------------------------------
// host side: error reporting + launch geometry
// ----------------------------------------------------------------------------------------
char *last_error_buffer();  // thread-local, defined in grid.hip

inline int fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(last_error_buffer(), 512, fmt, ap);
    va_end(ap);
    return code;
}

#define NFA_REQUIRE(cond, ...)                                            \
    do {                                                                  \
        if (!(cond)) return ::nfa::fail(NFA_ERR_INVALID_ARG, __VA_ARGS__); \
    } while (0)

inline int check_launch(const char *what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(NFA_ERR_LAUNCH, "%s: %s", what, hipGetErrorString(e));
    return NFA_OK;
}

constexpr int kWave = 64;           // CDNA wavefront
constexpr int kBlock = 256;         // 4 waves: one per SIMD of a CU
constexpr int kWavesPerBlock = kBlock / kWave;
constexpr int kNumCU = 256;         // MI355X
constexpr int kNumXCD = 8;

inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

// grid for a grid-stride elementwise kernel: enough blocks to fill the chip (>= 8 waves/SIMD
// worth), capped so tiny problems do not launch empty blocks.
inline unsigned blocks_for(int64_t n_threads_needed) {
    int64_t b = ceil_div(n_threads_needed, kBlock);
    const int64_t cap = (int64_t)kNumCU * 8;
    if (b > cap) b = cap;
    if (b < 1) b = 1;
    return (unsigned)b;
}

// ----------------------------------------------------------------------------------------
// device side: wave64 primitives
// ----------------------------------------------------------------------------------------
__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63u); }

__device__ __forceinline__ unsigned long long lanes_le(int lane) {  // bits [0, lane]
    return (lane >= 63) ? ~0ull : ((2ull << lane) - 1ull);
}
__device__ __forceinline__ unsigned long long lanes_lt(int lane) {  // bits [0, lane)
    return (1ull << lane) - 1ull;
}

struct OpSum {
    static __device__ __forceinline__ float identity() { return 0.0f; }
    static __device__ __forceinline__ float apply(float a, float b) { return a + b; }
};
struct OpProd {
    static __device__ __forceinline__ float identity() { return 1.0f; }
    static __device__ __forceinline__ float apply(float a, float b) { return a * b; }
};

// store of a per-sample output that a LATER kernel reads.  Non-temporal stores (-DNFA_NT_STORES) were measured and
// are NOT the default: at N = 2^24 weight_fwd drops from 3.93 to 3.35 TB/s and rendering_fwd from 3.53 to 3.40 with
// them (profiles/r02_streaming.md) — the 1.15-1.22x write traffic of these kernels is not a write-allocate effect
// that bypassing the cache removes.
template <class T>
__device__ __forceinline__ void st_stream(T *p, T v) {
#ifdef NFA_NT_STORES
    __builtin_nontemporal_store(v, p);
#else
    *p = v;
#endif
}

// --- cross-lane moves on the DPP path ----------------------------------------------------
// ds_bpermute shuffles go through the LDS crossbar (~100+ cycles each, and a scan is a chain of
// dependent ones); DPP operand modifiers move data inside the VALU in a few cycles.  gfx950
// still has the GFX9 full set: row_shr/row_shl (within a 16-lane row), row_bcast:15/31
// (last lane of a row / of the lower half to the following rows) and wave_shr/wave_shl:1.
constexpr int kDppRowShl = 0x100, kDppRowShr = 0x110, kDppWaveShl1 = 0x130, kDppWaveShr1 = 0x138,
              kDppRowBcast15 = 0x142, kDppRowBcast31 = 0x143;

// lanes without a source (outside the row / wave) keep `old`
template <int CTRL>
__device__ __forceinline__ float dpp_f(float old, float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, v),
                                                                 CTRL, 0xf, 0xf, false));
}
template <int CTRL>
__device__ __forceinline__ int64_t dpp_i64(int64_t v) {
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(uint32_t)(uint64_t)v, CTRL, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(uint32_t)((uint64_t)v >> 32), CTRL, 0xf, 0xf, false);
    return (int64_t)(((uint64_t)(uint32_t)hi << 32) | (uint64_t)(uint32_t)lo);
}
// value of lane-1 (lane 0 gets `fill`) / lane+1 (lane 63 gets `fill`)
__device__ __forceinline__ float lane_prev_f(float v, float fill) { return dpp_f<kDppWaveShr1>(fill, v); }
__device__ __forceinline__ float lane_next_f(float v, float fill) { return dpp_f<kDppWaveShl1>(fill, v); }
__device__ __forceinline__ int64_t lane_prev_i64(int64_t v) { return dpp_i64<kDppWaveShr1>(v); }
__device__ __forceinline__ int64_t lane_next_i64(int64_t v) { return dpp_i64<kDppWaveShl1>(v); }

// wave-uniform read of one lane (v_readlane_b32: VALU -> SGPR, no LDS); SRC is a constant
template <int SRC>
__device__ __forceinline__ float readlane_f(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), SRC));
}
template <int SRC>
__device__ __forceinline__ int64_t readlane_i64(int64_t v) {
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(uint64_t)v, SRC);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)((uint64_t)v >> 32), SRC);
    return (int64_t)(((uint64_t)hi << 32) | lo);
}

// sum over the 64 lanes, result in every lane: four row_shr adds inside each 16-lane row on the
// DPP path, then the four row totals through v_readlane (no LDS crossbar, no barrier)
__device__ __forceinline__ int64_t wave_sum_i64(int64_t v) {
    v += dpp_i64<kDppRowShr + 1>(v);
    v += dpp_i64<kDppRowShr + 2>(v);
    v += dpp_i64<kDppRowShr + 4>(v);
    v += dpp_i64<kDppRowShr + 8>(v);
    return readlane_i64<15>(v) + readlane_i64<31>(v) + readlane_i64<47>(v) + readlane_i64<63>(v);
}

// Segmented inclusive scan over the 64 lanes of a wave, forward (towards higher lanes).
// `dist` = number of lanes between this lane and the first lane of its segment inside this
// wave-chunk (0 for a segment head).  Four row_shr steps scan each 16-lane row, row_bcast:15
// then row_bcast:31 stitch the rows; a lane only accepts a partner inside its own segment.
template <class Op>
__device__ __forceinline__ float wave_seg_scan_fwd(float v, int dist) {
    const int lane = lane_id();
    const int rl = lane & 15;
    const int dr = dist < rl ? dist : rl;
    float u;
    u = dpp_f<kDppRowShr + 1>(Op::identity(), v); if (dr >= 1) v = Op::apply(u, v);
    u = dpp_f<kDppRowShr + 2>(Op::identity(), v); if (dr >= 2) v = Op::apply(u, v);
    u = dpp_f<kDppRowShr + 4>(Op::identity(), v); if (dr >= 4) v = Op::apply(u, v);
    u = dpp_f<kDppRowShr + 8>(Op::identity(), v); if (dr >= 8) v = Op::apply(u, v);
    u = dpp_f<kDppRowBcast15>(Op::identity(), v); if ((lane & 16) && dist > rl) v = Op::apply(u, v);
    u = dpp_f<kDppRowBcast31>(Op::identity(), v); if ((lane & 32) && dist > (lane & 31)) v = Op::apply(u, v);
    return v;
}
// Same, towards lower lanes; `dist` = lanes between this lane and the last lane of its segment.
// There is no downward row broadcast, so the rows are stitched with three v_readlane.
template <class Op>
__device__ __forceinline__ float wave_seg_scan_bwd(float v, int dist) {
    const int lane = lane_id();
    const int rr = 15 - (lane & 15);
    const int row = lane >> 4;
    const int dr = dist < rr ? dist : rr;
    float u;
    u = dpp_f<kDppRowShl + 1>(Op::identity(), v); if (dr >= 1) v = Op::apply(v, u);
    u = dpp_f<kDppRowShl + 2>(Op::identity(), v); if (dr >= 2) v = Op::apply(v, u);
    u = dpp_f<kDppRowShl + 4>(Op::identity(), v); if (dr >= 4) v = Op::apply(v, u);
    u = dpp_f<kDppRowShl + 8>(Op::identity(), v); if (dr >= 8) v = Op::apply(v, u);
    const bool reaches = dist > rr;
    u = readlane_f<48>(v); if (row == 2 && reaches) v = Op::apply(v, u);
    u = readlane_f<32>(v); if (row == 1 && reaches) v = Op::apply(v, u);
    u = readlane_f<16>(v); if (row == 0 && reaches) v = Op::apply(v, u);
    return v;
}

// distance from `lane` back to the closest set bit of `heads` at or below it; if there is
// none the segment started in an earlier chunk: returns lane (distance to lane 0) and sets
// `open` (the carry of the previous chunk applies).
__device__ __forceinline__ int dist_to_head(unsigned long long heads, int lane, bool &open) {
    const unsigned long long m = heads & lanes_le(lane);
    open = (m == 0ull);
    return open ? lane : lane - (63 - __clzll((long long)m));
}
// distance from `lane` forward to the closest set bit of `tails` at or above it; none => the
// segment continues into the next chunk (`open`), distance to lane 63.
__device__ __forceinline__ int dist_to_tail(unsigned long long tails, int lane, bool &open) {
    const unsigned long long m = tails & ~lanes_lt(lane);
    open = (m == 0ull);
    return open ? 63 - lane : (__ffsll((long long)m) - 1) - lane;
}


// ----------------------------------------------------------------------------------------
// Snapped tiling of a key-grouped array (DESIGN.md "segment-snapped wave tiles").
//
// The flat sample array is cut into nominal tiles of `tile` elements, one wave each.  Each
// wave then moves both of its boundaries forward to the next segment head (first element of
// a ray), so every ray lies wholly inside one wave's range: no carry ever crosses waves, no
// inter-workgroup communication, no atomics, bit-reproducible.  A wave whose nominal tile
// holds no head owns nothing (the ray that covers it belongs to an earlier wave).
// ----------------------------------------------------------------------------------------

// first position p in [from, limit) with keys[p] != keys[p-1] (position 0 counts as a head);
// returns `limit` if there is none.  Wave-uniform result; all 64 lanes must call.
__device__ __forceinline__ int64_t find_head(const int64_t *__restrict__ keys, int64_t from, int64_t limit) {
    if (from <= 0) return 0 < limit ? 0 : limit;
    const int lane = lane_id();
    for (int64_t base = from; base < limit; base += 64) {
        const int64_t i = base + lane;
        bool h = false;
        if (i < limit) h = keys[i] != keys[i - 1];
        const unsigned long long b = __ballot(h);
        if (b) return base + (__ffsll((long long)b) - 1);
    }
    return limit;
}

struct TileRange {
    int64_t begin, end;
};

// range owned by wave `w` for nominal tile size `tile` over n elements
__device__ __forceinline__ TileRange snapped_tile(const int64_t *__restrict__ keys, int64_t n, int64_t w, int64_t tile) {
    TileRange r;
    const int64_t nb = w * tile;
    int64_t ne = nb + tile;
    if (ne > n) ne = n;
    if (nb >= n) { r.begin = r.end = n; return r; }
    r.begin = find_head(keys, nb, ne);
    if (r.begin >= ne) { r.begin = r.end = n; return r; }   // no head in the nominal tile
    r.end = (ne >= n) ? n : find_head(keys, ne, n);
    return r;
}

// nominal tile size: multiples of 64, small enough to give every SIMD of the chip a few waves
// on mid-size inputs.  Capped at 576 (nine chunks, not a power of two): with big tiles every
// resident wave streams its own region `tile` elements apart, and at 2048 x 4 B = 8 KiB spacing
// the concurrent 256-byte requests alias onto a subset of the HBM channels (measured at
// N = 2^24: weight_bwd 3.77 -> 4.80 TB/s, keyed scan 4.03 -> 4.77 TB/s, profiles/r01_tile_sweep.md).
inline int64_t pick_tile(int64_t n) {
    if (const char *e = getenv("NFA_TILE")) {           // tuning knob (multiple of 64)
        const int64_t v = atoll(e);
        if (v >= 64 && v % 64 == 0) return v;
    }
    const int64_t target_waves = (int64_t)kNumCU * 4 * 4;
    int64_t t = ceil_div(ceil_div(n, target_waves), 64) * 64;
    if (t < 256) t = 256;
    if (t > 576) t = 576;
    return t;
}

}  // namespace nfa
