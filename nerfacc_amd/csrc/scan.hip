// scan.hip — per-ray (segmented) inclusive/exclusive sum and product scans for gfx950.
//
// Replaces nerfacc/cuda/csrc/scan.cu + include/utils_scan.cuh (chunked by packed_info) and
// scan_cub.cu (keyed by ray index, cub::DeviceScan::*ByKey) behind include/nerfacc_hip.h.
//
// MI355X mapping
//   keyed:   ray-owning wave tiles (common.hpp): each wave owns a range of whole rays,
//            walks it in line-aligned 64-element chunks, ballot()s the segment heads, runs a 6-step
//            segmented shuffle scan and carries one register across chunks.  Single pass,
//            8+4 bytes read and 4 written per element, no LDS, no atomics, no inter-workgroup
//            traffic, results independent of scheduling.
//   chunked: a row (ray) per 16-lane quarter wave — arbitrary (start,count) rows, as the
//            reference allows — four rows per wave, 64 elements per row and trip as ONE 16-byte access per lane,
//            four elements per lane serially, the lane totals through a 16-wide DPP scan with a carry.
#include "common.hpp"

namespace nfa {
namespace {

// ----------------------------------------------------------------------------------------
// keyed
// ----------------------------------------------------------------------------------------
// Optional fusions for the product backward (scan.cu:199-210): v = in * mul before the scan,
// result / max(div, 1e-10) after it.
template <int E>
struct ScanIn { float v[E], m[E], d[E]; };
template <class Op, bool INCL, bool REV, int E>
__global__ __launch_bounds__(kBlock) void scan_keyed_kernel(
    const int64_t *__restrict__ keys, const float *__restrict__ in, const float *__restrict__ mul,
    const float *__restrict__ div, float *__restrict__ out, int64_t n, int64_t tile, int spec)
{
    const int64_t w = (int64_t)blockIdx.x * kWavesPerBlock + wave_in_block();
    const float ident = Op::identity();
    float carry = ident;
    auto load = [&](int64_t i0, auto full) {
        ScanIn<E> p;
        ld_vec<E>(in, i0, n, ident, p.v, full);
        if (mul) ld_vec<E>(mul, i0, n, 1.0f, p.m, full);
        if (div) ld_vec<E>(div, i0, n, 1.0f, p.d, full);
        return p;
    };
    auto values = [&](const bool (&act)[E], const ScanIn<E> &p, float (&v)[E]) {
#pragma unroll
        for (int e = 0; e < E; ++e) v[e] = act[e] ? (mul ? p.v[e] * p.m[e] : p.v[e]) : ident;
    };
    auto store = [&](int64_t i0, const bool (&act)[E], const float (&incl)[E], const float (&excl)[E], const ScanIn<E> &p) {
        float r[E];
#pragma unroll
        for (int e = 0; e < E; ++e) {
            r[e] = INCL ? incl[e] : excl[e];
            if (div) r[e] = r[e] / fmaxf(p.d[e], 1e-10f);
        }
        st_vec<E>(out, i0, act, r);
    };
    if (!REV) {
        walk_rays_fwd<E, NFA_PF, ScanIn<E>>(keys, n, w, tile, spec, load,
            [&](int64_t i0, const bool (&act)[E], const int64_t (&)[E], const SegFwd<E> &s, const bool (&)[E], const ScanIn<E> &p) {
                float v[E], incl[E], excl[E];
                values(act, p, v);
                seg_scan_fwd<Op, E>(v, s, carry, incl, excl);
                store(i0, act, incl, excl, p);
            },
            [](int64_t) {});
    } else {
        walk_rays_bwd<E, NFA_PF, ScanIn<E>>(keys, n, w, tile, spec, load,
            [&](int64_t i0, const bool (&act)[E], const int64_t (&)[E], const SegBwd<E> &s, const ScanIn<E> &p) {
                float v[E], incl[E], excl[E];
                values(act, p, v);
                seg_scan_bwd<Op, E>(v, s, carry, incl, excl);
                store(i0, act, incl, excl, p);
            });
    }
}

// ----------------------------------------------------------------------------------------
// chunked by (start, count): 16 lanes per row
// ----------------------------------------------------------------------------------------
// Each quarter wave (one DPP row) owns a row; per trip it takes 4 x 16 elements with the four loads issued before the
// first scan (one 64-byte request per row and trip kept the HBM queues nearly empty: 2.1 TB/s), scans each 16-group
// on the DPP path (row_shr 1/2/4/8 — no ds_bpermute) and carries the row total through v_readlane.
template <class Op>
__device__ __forceinline__ float row16_incl_scan(float v) {
    v = Op::apply(dpp_f<kDppRowShr + 1>(Op::identity(), v), v);
    v = Op::apply(dpp_f<kDppRowShr + 2>(Op::identity(), v), v);
    v = Op::apply(dpp_f<kDppRowShr + 4>(Op::identity(), v), v);
    v = Op::apply(dpp_f<kDppRowShr + 8>(Op::identity(), v), v);
    return v;
}
// value of lane 15 of this lane's 16-lane row, in every lane of the row
__device__ __forceinline__ float row16_last(float v, int lane) {
    const float a = readlane_f<15>(v), b = readlane_f<31>(v), c = readlane_f<47>(v), d = readlane_f<63>(v);
    return (lane & 32) ? ((lane & 16) ? d : c) : ((lane & 16) ? b : a);
}

// four consecutive elements of a row per lane (in walk order: descending addresses when `reverse`).  The row's start is
// arbitrary, so the 16-byte access is only 4-byte aligned: global memory takes that (unaligned access mode), and it is one
// memory instruction per 64 elements of a row instead of four.
struct alignas(4) F4u { float v[4]; };
__device__ __forceinline__ void ld4_walk(const float *__restrict__ p, int64_t lo, bool full, const bool (&act)[4], const int64_t (&idx)[4],
                                         bool reverse, float fill, float (&out)[4]) {
    if (full) {
        F4u t;
        __builtin_memcpy(&t, p + lo, sizeof(t));
#pragma unroll
        for (int e = 0; e < 4; ++e) out[e] = reverse ? t.v[3 - e] : t.v[e];
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) out[e] = act[e] ? p[idx[e]] : fill;
    }
}

// `rw` rows per wave and outer trip (4 or 16), one per lane, DEALT to the four 16-lane groups as they finish their rows (round 3):
// with a fixed row per group the wave's trips were those of its longest row — 2.9 trips per four rows of 0..192 elements against
// the 2.0 a row needs on average; a group that is done now takes the wave's next non-empty row.  rw grows with the input so that
// small inputs keep one row per group (a wave per four rows: parallelism first) and the launch always has >= 8 k waves.
// (Also measured: 64 rows per wave — fewer waves than the chip wants at 2^24 elements; the next trip's loads issued one trip
// ahead — 36 -> 47 us, the kernel is bound by its instructions, not by bytes in flight: profiles/r03_streaming.md.)
template <class Op, bool INCL>
__global__ __launch_bounds__(kBlock) void scan_packed_kernel(
    const int64_t *__restrict__ starts, const int64_t *__restrict__ cnts, int64_t n_rows,
    const float *__restrict__ in, const float *__restrict__ mul, const float *__restrict__ div,
    float *__restrict__ out, int reverse, int normalize, int rw)
{
    const int lane = lane_id(), sub = lane & 15, grp = lane >> 4;
    const int64_t wave = (int64_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6), n_waves = (int64_t)gridDim.x * kWavesPerBlock;
    for (int64_t wb = wave * rw; wb < n_rows; wb += n_waves * rw) {
        // lane l < rw looks at row wb + l; the non-empty rows are packed into the low lanes (ds_permute: a lane SENDS to its rank;
        // every lane sends — a lane only receives while it is active: the empty rows go behind the non-empty ones, a permutation)
        const int64_t my_row = wb + lane;
        const bool mine = lane < rw && my_row < n_rows;
        const int64_t my_start = mine ? starts[my_row] : 0, my_cnt = mine ? cnts[my_row] : 0;
        const unsigned long long ne = __ballot(my_cnt > 0);
        const int n_ne = __popcll(ne);
        const int dst4 = (my_cnt > 0 ? __popcll(ne & lanes_lt(lane)) : n_ne + __popcll(~ne & lanes_lt(lane))) * 4;
        const int s_lo = __builtin_amdgcn_ds_permute(dst4, (int)(my_start & 0xffffffffll));
        const int s_hi = __builtin_amdgcn_ds_permute(dst4, (int)(my_start >> 32));
        const int c_lo = __builtin_amdgcn_ds_permute(dst4, (int)(my_cnt & 0xffffffffll));
        const int c_hi = __builtin_amdgcn_ds_permute(dst4, (int)(my_cnt >> 32));
        int taken = 0;
        bool busy = false;
        int64_t start = 0, cnt = 0, c = 0;
        float carry = Op::identity();
        for (;;) {
            // deal the next rows to the idle groups, in group order
            const unsigned long long idle_b = __ballot(!busy);
            const unsigned idle_g = (unsigned)((idle_b & 1ull) | ((idle_b >> 15) & 2ull) | ((idle_b >> 30) & 4ull) | ((idle_b >> 45) & 8ull));
            const int k = taken + __popc(idle_g & ((1u << grp) - 1u));
            const int src = (k < n_ne ? k : 0) * 4;
            const int a_lo = __builtin_amdgcn_ds_bpermute(src, s_lo), a_hi = __builtin_amdgcn_ds_bpermute(src, s_hi);
            const int b_lo = __builtin_amdgcn_ds_bpermute(src, c_lo), b_hi = __builtin_amdgcn_ds_bpermute(src, c_hi);
            if (!busy && k < n_ne) {
                start = ((int64_t)a_hi << 32) | (uint32_t)a_lo;
                cnt = ((int64_t)b_hi << 32) | (uint32_t)b_lo;
                c = 0;
                carry = Op::identity();
                busy = true;
            }
            taken += __popc(idle_g);
            if (!__ballot(busy)) break;
            const int64_t k0 = c + 4 * sub;                 // this lane's first element of the trip, in walk order
            bool act[4];
            int64_t idx[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                act[e] = busy && k0 + e < cnt;
                idx[e] = reverse ? (start + cnt - 1 - (k0 + e)) : (start + k0 + e);
            }
            const bool full = act[3];
            const int64_t lo = reverse ? idx[3] : idx[0];   // lowest address of the four
            float v[4], dv[4];
            ld4_walk(in, lo, full, act, idx, reverse != 0, Op::identity(), v);
            if (mul) {
                float m[4];
                ld4_walk(mul, lo, full, act, idx, reverse != 0, 1.0f, m);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = act[e] ? v[e] * m[e] : Op::identity();
            }
            if (div) ld4_walk(div, lo, full, act, idx, reverse != 0, 1.0f, dv);
            // lane-local scan, then the 16 lane totals of the row on the DPP path
            float x[4];
            x[0] = v[0];
#pragma unroll
            for (int e = 1; e < 4; ++e) x[e] = Op::apply(x[e - 1], v[e]);
            const float incl_lane = Op::apply(carry, row16_incl_scan<Op>(x[3]));
            const float pre = dpp_f<kDppRowShr + 1>(carry, incl_lane);            // lane 0 of the row has no source: keeps `carry`
            carry = row16_last(incl_lane, lane);
            float r[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float incl = Op::apply(pre, x[e]);
                const float excl = e == 0 ? pre : Op::apply(pre, x[e > 0 ? e - 1 : 0]);
                r[e] = INCL ? incl : excl;
                if (div) r[e] = r[e] / fmaxf(dv[e], 1e-10f);
            }
            if (full) {
                F4u t;
#pragma unroll
                for (int e = 0; e < 4; ++e) t.v[e] = reverse ? r[3 - e] : r[e];
                __builtin_memcpy(out + lo, &t, sizeof(t));
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (act[e]) out[idx[e]] = r[e];
            }
            c += 64;
            if (busy && c >= cnt) {                         // this group's row is done
                if (normalize) {  // utils_scan.cuh:101-109 / 228-236: divide by the row total
                    const float den = fmaxf(carry, 1e-10f);
                    for (int64_t q = 0; q < cnt; q += 16) {
                        const int64_t kk = q + sub;
                        if (kk >= cnt) continue;
                        if (!INCL && kk == 0) continue;
                        const int64_t i = reverse ? (start + cnt - 1 - kk) : (start + kk);
                        out[i] = out[i] / den;
                    }
                }
                busy = false;
            }
        }
    }
}

template <class Op, bool INCL, bool REV>
void launch_keyed_dir(const int64_t *keys, const float *in, const float *mul, const float *div, float *out,
                      int64_t n, const TilePlan &pl, hipStream_t s) {
    const dim3 g(tile_blocks(n, pl.tile)), b(kBlock);
    if (pl.e == 4) hipLaunchKernelGGL((scan_keyed_kernel<Op, INCL, REV, 4>), g, b, 0, s, keys, in, mul, div, out, n, pl.tile, pl.spec);
    else if (pl.e == 2) hipLaunchKernelGGL((scan_keyed_kernel<Op, INCL, REV, 2>), g, b, 0, s, keys, in, mul, div, out, n, pl.tile, pl.spec);
    else hipLaunchKernelGGL((scan_keyed_kernel<Op, INCL, REV, 1>), g, b, 0, s, keys, in, mul, div, out, n, pl.tile, pl.spec);
}
template <class Op, bool INCL>
void launch_keyed(const int64_t *keys, const float *in, const float *mul, const float *div, float *out,
                  int64_t n, bool reverse, hipStream_t s) {
    const TilePlan pl = pick_plan(n, aligned16({keys, in, mul, div, out}));
    if (reverse) launch_keyed_dir<Op, INCL, true>(keys, in, mul, div, out, n, pl, s);
    else launch_keyed_dir<Op, INCL, false>(keys, in, mul, div, out, n, pl, s);
}

template <class Op, bool INCL>
void launch_packed(const int64_t *starts, const int64_t *cnts, int64_t n_rows, const float *in,
                   const float *mul, const float *div, float *out, int reverse, int normalize, hipStream_t s) {
    // rows per wave: 4 (a row per group) until the launch has ~8 k waves of 16; NFA_SCAN_RW = 4 | 16 overrides
    int rw = n_rows >= 16 * 8192 ? 16 : 4;
    rw = (int)opt(OPT_SCAN_RW, rw);
    const unsigned nb = blocks_for(ceil_div(n_rows, rw) * kWave);
    hipLaunchKernelGGL((scan_packed_kernel<Op, INCL>), dim3(nb), dim3(kBlock), 0, s, starts, cnts, n_rows, in, mul, div, out,
                       reverse, normalize, rw);
}

}  // namespace
}  // namespace nfa

using namespace nfa;

NFA_EXPORT int nfa_scan_keyed(const int64_t *keys, const float *inputs, float *outputs, int64_t n,
                              int32_t op, int32_t inclusive, int32_t reverse, void *stream)
{
    NFA_REQUIRE(n >= 0, "scan_keyed: n < 0");
    NFA_REQUIRE(op == NFA_OP_SUM || op == NFA_OP_PROD, "scan_keyed: bad op %d", op);
    if (n == 0) return NFA_OK;
    NFA_REQUIRE(keys && inputs && outputs, "scan_keyed: NULL pointer");
    hipStream_t s = (hipStream_t)stream;
    if (op == NFA_OP_SUM) {
        if (inclusive) launch_keyed<OpSum, true>(keys, inputs, nullptr, nullptr, outputs, n, reverse, s);
        else launch_keyed<OpSum, false>(keys, inputs, nullptr, nullptr, outputs, n, reverse, s);
    } else {
        if (inclusive) launch_keyed<OpProd, true>(keys, inputs, nullptr, nullptr, outputs, n, reverse, s);
        else launch_keyed<OpProd, false>(keys, inputs, nullptr, nullptr, outputs, n, reverse, s);
    }
    return check_launch("scan_keyed_kernel");
}

NFA_EXPORT int nfa_scan_packed(const int64_t *chunk_starts, const int64_t *chunk_cnts, int64_t n_rays,
                               const float *inputs, float *outputs, int64_t n,
                               int32_t op, int32_t inclusive, int32_t reverse, int32_t normalize, void *stream)
{
    NFA_REQUIRE(n >= 0 && n_rays >= 0, "scan_packed: negative size");
    NFA_REQUIRE(op == NFA_OP_SUM || op == NFA_OP_PROD, "scan_packed: bad op %d", op);
    NFA_REQUIRE(!(normalize && op == NFA_OP_PROD), "scan_packed: normalize is only defined for sums");
    NFA_REQUIRE(!(normalize && reverse), "scan_packed: backward does not support normalize (scan.cu:25-26)");
    if (n == 0 || n_rays == 0) return NFA_OK;
    NFA_REQUIRE(chunk_starts && chunk_cnts && inputs && outputs, "scan_packed: NULL pointer");
    hipStream_t s = (hipStream_t)stream;
    if (op == NFA_OP_SUM) {
        if (inclusive) launch_packed<OpSum, true>(chunk_starts, chunk_cnts, n_rays, inputs, nullptr, nullptr, outputs, reverse, normalize, s);
        else launch_packed<OpSum, false>(chunk_starts, chunk_cnts, n_rays, inputs, nullptr, nullptr, outputs, reverse, normalize, s);
    } else {
        if (inclusive) launch_packed<OpProd, true>(chunk_starts, chunk_cnts, n_rays, inputs, nullptr, nullptr, outputs, reverse, 0, s);
        else launch_packed<OpProd, false>(chunk_starts, chunk_cnts, n_rays, inputs, nullptr, nullptr, outputs, reverse, 0, s);
    }
    return check_launch("scan_packed_kernel");
}

NFA_EXPORT int nfa_prod_backward(const int64_t *keys, const int64_t *chunk_starts, const int64_t *chunk_cnts,
                                 int64_t n_rays, const float *inputs, const float *outputs,
                                 const float *grad_outputs, float *grad_inputs, int64_t n, int32_t inclusive,
                                 void *stream)
{
    NFA_REQUIRE(n >= 0, "prod_backward: n < 0");
    if (n == 0) return NFA_OK;
    NFA_REQUIRE(inputs && outputs && grad_outputs && grad_inputs, "prod_backward: NULL pointer");
    NFA_REQUIRE((keys != nullptr) != (chunk_starts != nullptr && chunk_cnts != nullptr),
                "prod_backward: give either keys or (chunk_starts, chunk_cnts)");
    hipStream_t s = (hipStream_t)stream;
    if (keys) {
        if (inclusive) launch_keyed<OpSum, true>(keys, grad_outputs, outputs, inputs, grad_inputs, n, true, s);
        else launch_keyed<OpSum, false>(keys, grad_outputs, outputs, inputs, grad_inputs, n, true, s);
    } else {
        if (n_rays == 0) return NFA_OK;
        if (inclusive) launch_packed<OpSum, true>(chunk_starts, chunk_cnts, n_rays, grad_outputs, outputs, inputs, grad_inputs, 1, 0, s);
        else launch_packed<OpSum, false>(chunk_starts, chunk_cnts, n_rays, grad_outputs, outputs, inputs, grad_inputs, 1, 0, s);
    }
    return check_launch("prod_backward");
}
