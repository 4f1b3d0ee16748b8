// common.hpp — shared host/device helpers for libnerfacc_hip.so (gfx950 / wave64 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <initializer_list>
#include <type_traits>

#include "../../include/nerfacc_hip.h"
#include "options.hpp"

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
// the wave's index inside its workgroup as a SCALAR (threadIdx.x >> 6 is wave-uniform, but the compiler keeps it in a vector register
// and then carries every tile base, loop bound and chunk address of a walk as 64-bit VALU arithmetic with vector compares in front of
// the loop branches; through v_readfirstlane all of that moves to the scalar unit)
__device__ __forceinline__ int wave_in_block() { return __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); }

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

// store of a per-sample output that a LATER kernel reads: a plain store.  (Non-temporal stores were measured in rounds 1 and 2:
// weight_fwd 3.93 -> 3.35 TB/s, rendering_fwd 3.53 -> 3.40 at N = 2^24, profiles/r02_streaming.md; a pure fill does not care,
// tools/ubench/ceiling.py.)
template <class T>
__device__ __forceinline__ void st_stream(T *p, T v) { *p = v; }

// NT: non-temporal.  On MI355X a read-only 16-byte-lane stream reaches 7.0 TB/s with non-temporal loads against 6.2-6.3 with
// plain ones (tools/ubench/ceiling.py), and weight_fwd / rendering_fwd gain 10 % at N = 2^24 (4.43 -> 4.89, 4.16 -> 4.59 TB/s);
// the kernels that move few bytes per sample (scans 16 B, accumulate 12-24 B) LOSE 12-20 % with them at that size — their
// inputs are served by the 256 MB Infinity Cache when a benchmark repeats a call, which a non-temporal load forgoes.  So the
// choice is per kernel (profiles/r03_streaming.md); -DNFA_NT_LOADS forces them everywhere.
template <bool NT = false, class V>
__device__ __forceinline__ V ld_stream(const V *p) {
#ifdef NFA_NT_LOADS
    return __builtin_nontemporal_load(p);
#else
    if (NT) return __builtin_nontemporal_load(p);
    return *p;
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

// the same with a wave-uniform lane number that is not a constant
__device__ __forceinline__ int64_t readlane_i64_dyn(int64_t v, int src) {
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(uint64_t)v, src);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)((uint64_t)v >> 32), src);
    return (int64_t)(((uint64_t)hi << 32) | lo);
}

// the first active lane's value as a wave-uniform (scalar) 64-bit integer
__device__ __forceinline__ int64_t readfirstlane_i64(int64_t v) {
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uint64_t)v);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)((uint64_t)v >> 32));
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
// Ray-owning wave tiles of a key-grouped array (DESIGN.md "aligned wave tiles").
//
// The flat sample array is cut into nominal tiles of `tile` elements, one wave each.  A wave OWNS the rays
// whose first element (forward walks) / last element (backward walks) lies in its nominal tile, so every ray is
// wholly inside one wave's range: no carry ever crosses waves, no inter-workgroup communication, no atomics,
// bit-reproducible.  A wave whose nominal tile holds no such element owns nothing (the ray that covers it belongs
// to another wave).
//
// The wave walks chunks of 64 x E elements aligned to the NOMINAL grid (base = multiple of 64 E), lane l holding
// the E consecutive elements base + l E ...: every load and store is an aligned 4 E-byte (keys: 8 E-byte) vector
// per lane and a whole number of 128-byte lines per wave.  (Chunks aligned to the start of the owned range made
// every access straddle lines and needed one memory instruction per element and array: weight_fwd 4.7 vs 5.5 TB/s
// on 63- vs 64-sample rays, profiles/r02_streaming.md; and the range had to be searched for first — two dependent
// memory round trips per wave before the first useful load.)  Ownership is decided from the chunk's own keys:
// elements of the first / last chunk that belong to a neighbouring wave's ray are masked off.  PF further chunks
// are requested before the current one is used (payloads are raw loaded values: anything computed in `load` would
// wait for the data there); with `spec` — launches too small to fill the chip — the chunk past the nominal tile
// is requested ahead as well.
//
// Per-ray results of forward walks: the last element of a ray is known from the heads of its own chunk, except
// for the chunk's very last element, whose successor lives in the next chunk.  The walker reports `tail` for
// everything it knows and calls `flush(key)` at the start of the next chunk when that element turns out to have
// ended its ray (the kernel's carries then hold the ray's totals).
// ----------------------------------------------------------------------------------------
#ifndef NFA_PF
#define NFA_PF 1      // chunks requested ahead by the tiled walkers (tuning knob)
#endif
struct NoPayload {};

template <int E, class P>
struct RayChunk {
    int64_t key[E];
    P p;
};

// aligned E-wide loads / stores (i0 is a multiple of E by construction; elements at or beyond n are filled / skipped)
// `full` (std::true_type): the caller knows — wave-uniformly — that the whole chunk lies inside [0, n): no per-lane test, no exec
// masking, no branch (walk_rays_* decide that once per chunk with a scalar compare; round 5: the per-array `i0 + E <= n` tests were
// half of the walk loops' ~75 branches)
template <int E, class T, bool NT = false, class Full = std::false_type>
__device__ __forceinline__ void ld_vec(const T *__restrict__ p, int64_t i0, int64_t n, T fill, T (&out)[E], Full = Full()) {
    typedef T vec_t __attribute__((ext_vector_type(E)));
    if (Full::value || i0 + E <= n) {
        const vec_t v = ld_stream<NT>(reinterpret_cast<const vec_t *>(p + i0));
#pragma unroll
        for (int e = 0; e < E; ++e) out[e] = v[e];
    } else {
#pragma unroll
        for (int e = 0; e < E; ++e) out[e] = (i0 + e < n) ? p[i0 + e] : fill;
    }
}
// S interleaved channels per element (rgb: S = 3): out[e][c]
template <int E, int S, bool NT = false, class Full = std::false_type>
__device__ __forceinline__ void ld_vec_strided(const float *__restrict__ p, int64_t i0, int64_t n, float fill, float (&out)[E][S], Full = Full()) {
    if (E > 1 && (Full::value || i0 + E <= n)) {
        typedef float vec_t __attribute__((ext_vector_type(E)));
        float flat[E * S];
#pragma unroll
        for (int j = 0; j < S; ++j) {
            const vec_t v = ld_stream<NT>(reinterpret_cast<const vec_t *>(p + i0 * S + j * E));
#pragma unroll
            for (int e = 0; e < E; ++e) flat[j * E + e] = v[e];
        }
#pragma unroll
        for (int e = 0; e < E; ++e)
#pragma unroll
            for (int c = 0; c < S; ++c) out[e][c] = flat[e * S + c];
    } else {
#pragma unroll
        for (int e = 0; e < E; ++e)
#pragma unroll
            for (int c = 0; c < S; ++c) out[e][c] = (i0 + e < n) ? p[(i0 + e) * S + c] : fill;
    }
}
template <int E, class T>
__device__ __forceinline__ void st_vec(T *__restrict__ p, int64_t i0, const bool (&act)[E], const T (&v)[E]) {
    typedef T vec_t __attribute__((ext_vector_type(E)));
    bool all = true;
#pragma unroll
    for (int e = 0; e < E; ++e) all = all && act[e];
    if (E > 1 && __ballot(all) == ~0ull) {                 // every lane stores its whole vector (interior chunks): a scalar branch
        vec_t o;
#pragma unroll
        for (int e = 0; e < E; ++e) o[e] = v[e];
        *reinterpret_cast<vec_t *>(p + i0) = o;
    } else if (E > 1 && all) {
        vec_t o;
#pragma unroll
        for (int e = 0; e < E; ++e) o[e] = v[e];
        *reinterpret_cast<vec_t *>(p + i0) = o;
    } else {
#pragma unroll
        for (int e = 0; e < E; ++e)
            if (act[e]) p[i0 + e] = v[e];
    }
}

// Segment flags of one chunk, lane view.  Forward: head[e] (elements outside the owned range count as heads);
// the lane-level scan runs over lanes that contain a head.
template <int E>
struct SegFwd {
    bool head[E], open;
    int dist;
};
template <int E>
struct SegBwd {
    bool tail[E], open;
    int dist;
};

// Segmented scans of a chunk: E elements per lane serially, the lane totals through the 64-lane DPP scan, one
// carry register across chunks.
template <class Op, int E>
__device__ __forceinline__ void seg_scan_fwd(const float (&v)[E], const SegFwd<E> &s, float &carry, float (&incl)[E], float (&excl)[E]) {
    const int lane = lane_id();
    float x[E];
    x[0] = v[0];
#pragma unroll
    for (int e = 1; e < E; ++e) x[e] = s.head[e] ? v[e] : Op::apply(x[e - 1], v[e]);
    float tot = wave_seg_scan_fwd<Op>(x[E - 1], s.dist);
    if (s.open) tot = Op::apply(carry, tot);
    float pre = lane_prev_f(tot, Op::identity());
    if (lane == 0) pre = carry;
    carry = readlane_f<63>(tot);
    bool cont = true;                                   // no head so far in this lane: the segment came in from the left
#pragma unroll
    for (int e = 0; e < E; ++e) {
        cont = cont && !s.head[e];
        incl[e] = cont ? Op::apply(pre, x[e]) : x[e];
        excl[e] = s.head[e] ? Op::identity() : (e == 0 ? pre : incl[e > 0 ? e - 1 : 0]);
    }
}
template <class Op, int E>
__device__ __forceinline__ void seg_scan_bwd(const float (&v)[E], const SegBwd<E> &s, float &carry, float (&incl)[E], float (&excl)[E]) {
    const int lane = lane_id();
    float x[E];
    x[E - 1] = v[E - 1];
#pragma unroll
    for (int e = E - 2; e >= 0; --e) x[e] = s.tail[e] ? v[e] : Op::apply(v[e], x[e + 1]);
    float tot = wave_seg_scan_bwd<Op>(x[0], s.dist);
    if (s.open) tot = Op::apply(tot, carry);
    float post = lane_next_f(tot, Op::identity());
    if (lane == 63) post = carry;
    carry = readlane_f<0>(tot);
    bool cont = true;
#pragma unroll
    for (int e = E - 1; e >= 0; --e) {
        cont = cont && !s.tail[e];
        incl[e] = cont ? Op::apply(x[e], post) : x[e];
        excl[e] = s.tail[e] ? Op::identity() : (e == E - 1 ? post : incl[e < E - 1 ? e + 1 : E - 1]);
    }
}

// payload types whose kernels stream their keys with non-temporal loads too (see ld_stream) say `static constexpr bool kStreamKeys = true;`
template <class P, class = void>
struct StreamKeys { static constexpr bool value = false; };
template <class P>
struct StreamKeys<P, decltype((void)P::kStreamKeys)> { static constexpr bool value = P::kStreamKeys; };

template <int E, class P, class Load>
__device__ __forceinline__ RayChunk<E, P> fetch_chunk(const int64_t *__restrict__ keys, int64_t n, int64_t base, int lane, Load &load) {
    RayChunk<E, P> c;
    const int64_t i0 = base + (int64_t)lane * E;
    if (base + 64 * E <= n) {                               // (scalar: `base` derives from wave_in_block()) the chunk is whole
        ld_vec<E, int64_t, StreamKeys<P>::value>(keys, i0, n, (int64_t)0, c.key, std::true_type());
        c.p = load(i0, std::true_type());
    } else {
        ld_vec<E, int64_t, StreamKeys<P>::value>(keys, i0, n, (int64_t)0, c.key);
        c.p = load(i0, std::false_type());
    }
    return c;
}

// position (lane * E + e) of the first / last set flag of a chunk; f[e] per lane.  Wave-uniform; -1 if none.
template <int E>
__device__ __forceinline__ int first_flag(const bool (&f)[E]) {
    int fe = E;
#pragma unroll
    for (int e = E - 1; e >= 0; --e) if (f[e]) fe = e;
    const unsigned long long b = __ballot(fe < E);
    if (!b) return -1;
    const int l = __ffsll((long long)b) - 1;
    return l * E + __builtin_amdgcn_readlane(fe, l);
}
template <int E>
__device__ __forceinline__ int last_flag(const bool (&f)[E]) {
    int fe = -1;
#pragma unroll
    for (int e = 0; e < E; ++e) if (f[e]) fe = e;
    const unsigned long long b = __ballot(fe >= 0);
    if (!b) return -1;
    const int l = 63 - __clzll((long long)b);
    return l * E + __builtin_amdgcn_readlane(fe, l);
}

// body(i0, act[E], key[E], SegFwd<E>, tail[E], payload) for every chunk of the rays whose head lies in nominal tile w,
// ascending; flush(key) when the element that closed the previous chunk ended ray `key`.
template <int E, int PF, class P, class Load, class Body, class Flush>
__device__ __forceinline__ void walk_rays_fwd(const int64_t *__restrict__ keys, int64_t n, int64_t w, int64_t tile, int spec,
                                              Load load, Body body, Flush flush)
{
    constexpr int64_t CH = 64 * E;
    const int lane = lane_id();
    const int64_t nb = w * tile;
    if (nb >= n) return;
    const int64_t ne = nb + tile < n ? nb + tile : n;
    int64_t lim = ne + (spec ? CH : 0);                     // chunks below `lim` are requested ahead
    if (lim > n) lim = n;
    RayChunk<E, P> q[PF + 1] = {};            // (slots past the last requested chunk are copied around by the shifts below, never used)
    q[0] = fetch_chunk<E, P>(keys, n, nb, lane, load);
#pragma unroll
    for (int d = 1; d < PF; ++d)
        if (nb + CH * d < lim) q[d] = fetch_chunk<E, P>(keys, n, nb + CH * d, lane, load);
    int64_t edge = nb > 0 ? keys[nb - 1] : 0;               // the key before the current chunk
    bool started = false, pending = false;                  // pending: the previous chunk's last element was walked, its tail unknown
    for (int64_t base = nb;; base += CH) {
        if (PF > 0 && base + CH * PF < lim) q[PF] = fetch_chunk<E, P>(keys, n, base + CH * PF, lane, load);
        const int64_t i0 = base + (int64_t)lane * E;
        int64_t pk = lane_prev_i64(q[0].key[E - 1]);
        if (lane == 0) pk = edge;
        bool hf[E];
        hf[0] = i0 < n && (i0 == 0 || q[0].key[0] != pk);
#pragma unroll
        for (int e = 1; e < E; ++e) hf[e] = i0 + e < n && q[0].key[e] != q[0].key[e - 1];
        int lo = 0, hi = CH;
        bool done = base + CH >= n, skip = false;
        if (base < ne) {
            if (!started) {
                const int f = first_flag<E>(hf);
                if (f >= 0) { lo = f; started = true; }
                else if (base + CH >= ne) return;                   // no head in the nominal tile: nothing owned
                else skip = true;
            }
        } else {                                                    // past the tile: the straddling ray ends at the next head
            const int f = first_flag<E>(hf);
            if (f >= 0) { hi = f; done = true; }
        }
        if (!skip) {
            if (pending && (__ballot(hf[0]) & 1ull)) flush(edge);
            bool act[E], tail[E];
            SegFwd<E> s;
            bool any = false;
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const int pos = lane * E + e;
                act[e] = i0 + e < n && pos >= lo && pos < hi;
                s.head[e] = !act[e] || hf[e];
                any = any || s.head[e];
            }
            s.dist = dist_to_head(__ballot(any), lane, s.open);
            // tail[e]: the next element is a head, lies past the array or past the range
            int nh0 = s.head[0] ? 1 : 0;
            nh0 = __builtin_amdgcn_update_dpp(1, nh0, kDppWaveShl1, 0xf, 0xf, false);   // lane 63: unknown here, see `pending`
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const bool next_head = e + 1 < E ? s.head[e + 1 < E ? e + 1 : 0] : (lane == 63 ? base + CH >= n : nh0 != 0);
                tail[e] = act[e] && next_head;
            }
            body(i0, act, q[0].key, s, tail, q[0].p);
            pending = true;                                         // (if the walk goes on, the chunk's last element was active)
        }
        edge = readlane_i64<63>(q[0].key[E - 1]);
        if (done) break;
#pragma unroll
        for (int d = 0; d < PF; ++d) q[d] = q[d + 1];
        if (PF == 0 || base + CH >= lim) q[0] = fetch_chunk<E, P>(keys, n, base + CH, lane, load);
    }
}

// body(i0, act[E], key[E], SegBwd<E>, payload) for every chunk of the rays whose tail lies in nominal tile w, descending
template <int E, int PF, class P, class Load, class Body>
__device__ __forceinline__ void walk_rays_bwd(const int64_t *__restrict__ keys, int64_t n, int64_t w, int64_t tile, int spec,
                                              Load load, Body body)
{
    constexpr int64_t CH = 64 * E;
    const int lane = lane_id();
    const int64_t nb = w * tile;
    if (nb >= n) return;
    const int64_t ne = nb + tile < n ? nb + tile : n;
    int64_t lim = nb - (spec ? CH : 0);                     // chunks at or above `lim` are requested ahead
    if (lim < 0) lim = 0;
    const int64_t top = ((ne - 1) / CH) * CH;
    RayChunk<E, P> q[PF + 1] = {};            // (slots past the last requested chunk are copied around by the shifts below, never used)
    q[0] = fetch_chunk<E, P>(keys, n, top, lane, load);
#pragma unroll
    for (int d = 1; d < PF; ++d)
        if (top - CH * d >= lim) q[d] = fetch_chunk<E, P>(keys, n, top - CH * d, lane, load);
    int64_t edge = top + CH < n ? keys[top + CH] : 0;       // the key after the current chunk
    bool started = false;
    for (int64_t base = top;; base -= CH) {
        if (PF > 0 && base - CH * PF >= lim) q[PF] = fetch_chunk<E, P>(keys, n, base - CH * PF, lane, load);
        const int64_t i0 = base + (int64_t)lane * E;
        int64_t nk = lane_next_i64(q[0].key[0]);
        if (lane == 63) nk = edge;
        bool tf[E];
#pragma unroll
        for (int e = 0; e + 1 < E; ++e) tf[e] = i0 + e < n && (i0 + e + 1 >= n || q[0].key[e] != q[0].key[e + 1]);
        tf[E - 1] = i0 + E - 1 < n && (i0 + E >= n || q[0].key[E - 1] != nk);
        int lo = 0, hi = CH;
        bool done = base == 0, skip = false;
        if (base >= nb) {
            if (!started) {
                const int f = last_flag<E>(tf);
                if (f >= 0) { hi = f + 1; started = true; }
                else if (base == nb) return;                        // no tail in the nominal tile: nothing owned
                else skip = true;
            }
        } else {                                                    // below the tile: the straddling ray begins after the last tail
            const int f = last_flag<E>(tf);
            if (f >= 0) { lo = f + 1; done = true; }
        }
        if (!skip) {
            bool act[E];
            SegBwd<E> s;
            bool any = false;
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const int pos = lane * E + e;
                act[e] = i0 + e < n && pos >= lo && pos < hi;
                s.tail[e] = !act[e] || tf[e];
                any = any || s.tail[e];
            }
            s.dist = dist_to_tail(__ballot(any), lane, s.open);
            body(i0, act, q[0].key, s, q[0].p);
        }
        edge = readlane_i64<0>(q[0].key[0]);
        if (done) break;
#pragma unroll
        for (int d = 0; d < PF; ++d) q[d] = q[d + 1];
        if (PF == 0 || base - CH < lim) q[0] = fetch_chunk<E, P>(keys, n, base - CH, lane, load);
    }
}

// Launch plan of the tiled kernels: elements per lane, nominal tile, whether to request the chunk past the tile early.
// Tiles are 9 chunks (not a power of two: with big power-of-two tiles every resident wave streams its own region
// `tile` elements apart and the concurrent requests alias onto a subset of the HBM channels — measured at N = 2^24:
// weight_bwd 3.77 -> 4.80 TB/s, profiles/r01_tile_sweep.md), shrunk on inputs that cannot give every SIMD a few waves.
struct TilePlan {
    int e;
    int64_t tile;
    int spec;
};
// every pointer of a vectorised walk must be 16-byte aligned (tensors from the caching allocator are; views with an
// odd storage offset are not): otherwise one element per lane
inline bool aligned16(std::initializer_list<const void *> ptrs) {
    for (const void *q : ptrs)
        if ((reinterpret_cast<uintptr_t>(q) & 15u) != 0) return false;
    return true;
}
inline TilePlan pick_plan(int64_t n, bool vec_ok = true) {
    TilePlan p;
    // (2 samples per lane from 128 k samples: at roofline scale it measures equal to 4 per lane within noise for every kernel except
    // rendering_bwd, which needs 170 VGPRs at 4 and runs 13-15 % faster at 2; at the training size (2.6e5 samples) it is 5-15 %
    // ahead of one per lane back to back, and below ~1e5 samples one per lane wins: profiles/r02_streaming.md)
    p.e = n >= ((int64_t)1 << 17) ? 2 : 1;
    p.e = (int)opt(OPT_E, p.e);                          // tuning knobs (options.hpp)
    if (!vec_ok) p.e = 1;
    const int64_t ch = 64 * p.e;
    const int64_t target_waves = (int64_t)kNumCU * 4 * 4;
    int64_t t = ceil_div(ceil_div(n, target_waves), ch) * ch;
    if (t < ch) t = ch;                                  // (the one-element-per-lane plan has the smallest tile: workspaces are sized by it)
    if (t > 9 * ch) t = 9 * ch;
    {
        const int64_t v = opt(OPT_TILE, t);
        if (v >= ch && v % ch == 0) t = v;
    }
    p.tile = t;
    p.spec = t < 9 * ch;
    return p;
}
inline unsigned tile_blocks(int64_t n, int64_t tile) { return (unsigned)ceil_div(ceil_div(n, tile), kWavesPerBlock); }

// hipLaunchKernelGGL of KERNEL<E> for the plan's E
#define NFA_LAUNCH_TILED(KERNEL, plan, n, stream, ...)                                                                     \
    do {                                                                                                                   \
        const dim3 g_(tile_blocks(n, (plan).tile)), b_(kBlock);                                                            \
        if ((plan).e == 4) hipLaunchKernelGGL((KERNEL<4>), g_, b_, 0, stream, __VA_ARGS__);                                 \
        else if ((plan).e == 2) hipLaunchKernelGGL((KERNEL<2>), g_, b_, 0, stream, __VA_ARGS__);                            \
        else hipLaunchKernelGGL((KERNEL<1>), g_, b_, 0, stream, __VA_ARGS__);                                               \
    } while (0)


// ----------------------------------------------------------------------------------------
// occupancy threshold shared by occgrid.hip (nfa_grid_threshold) and grid.hip (nfa_grid_threshold_packed)
// ----------------------------------------------------------------------------------------
constexpr int kReduceBlocks = 1024;     // (128 blocks left a 128^3 grid's 8 MB to 64 dependent loads per thread: 18.8 us, 0.45 TB/s)
// {sum, count} partial pairs of the visible cells -> min(mean, occ_thre) (NaN mean: nothing passes); fixed-order tree,
// called by the first wave of a workgroup (all 64 lanes)
__device__ __forceinline__ float threshold_from_partials(const double *__restrict__ partials, int n_partials, float occ_thre) {
    const int l = lane_id();
    double a = 0.0, b = 0.0;
    for (int i = l; i < n_partials; i += 64) { a += partials[2 * i]; b += partials[2 * i + 1]; }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) { a += __shfl_down(a, off, 64); b += __shfl_down(b, off, 64); }
    a = __shfl(a, 0, 64);
    b = __shfl(b, 0, 64);
    const float mean = b > 0.0 ? (float)(a / b) : __builtin_nanf("");
    return (mean != mean) ? mean : fminf(mean, occ_thre);
}
// launches grid_mean_partials_kernel (occgrid.hip); returns the number of partial pairs written to `partials`
int launch_grid_mean_partials(const float *occs, int64_t n_cells, double *partials, hipStream_t s);

}  // namespace nfa
