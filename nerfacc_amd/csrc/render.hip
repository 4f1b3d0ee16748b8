// render.hip — fused volumetric-rendering kernels for gfx950.
//
// In the reference this part of the hot path is ~45 small ATen kernels per training step
// (nerfacc/volrend.py: sigma*dt, exp, exclusive scan, mul, three index_add_ with float
// atomics, clamp/div, background blend — and their autograd twins) plus three host-syncing
// boolean-mask compactions in OccGridEstimator.sampling (occ_grid.py:194-220).  Here each
// logical op is ONE streaming kernel over the flat sample array, built on the
// ray-owning, line-aligned wave tiles of common.hpp:
//   nfa_render_weight_from_density_{fwd,bwd}   volrend.py:219-278, 326-376
//   nfa_visibility_compact                     volrend.py:435-494 + occ_grid.py:216-220
//   nfa_accumulate_along_rays{,_bwd}           volrend.py:497-587
//   nfa_rendering_{fwd,bwd}                    volrend.py:104-164 (after rgb_sigma_fn)
//   nfa_pack_info / nfa_unpack_info            pack.py:10-49
// All are HBM-streaming (bytes per sample in DESIGN.md), need no LDS, and are deterministic
// for ray-sorted input (per-ray sums are formed inside one wave, in a fixed order).
#include <algorithm>

#include "common.hpp"
#include "lookback.hpp"     // hand-offs between the workgroups of one launch (the single-launch form of the filter)

namespace nfa {
namespace {

constexpr float kEpsF32 = 1.1920928955078125e-07f;   // torch.finfo(torch.float32).eps

__device__ __forceinline__ int64_t wave_index() { return (int64_t)blockIdx.x * kWavesPerBlock + wave_in_block(); }

// ----------------------------------------------------------------------------------------
// weights / transmittance / alpha from density
// ----------------------------------------------------------------------------------------
template <int E>
struct WeightFwdIn { float t0[E], t1[E], sg[E], pf[E]; static constexpr bool kStreamKeys = true; };
template <int E>
__global__ __launch_bounds__(kBlock) void weight_fwd_kernel(
    const int64_t *__restrict__ keys, const float *__restrict__ ts, const float *__restrict__ te,
    const float *__restrict__ sigmas, const float *__restrict__ prefix, int64_t n, int64_t tile, int spec,
    float *__restrict__ weights, float *__restrict__ trans, float *__restrict__ alphas)
{
    float carry = 0.0f;
    walk_rays_fwd<E, NFA_PF, WeightFwdIn<E>>(keys, n, wave_index(), tile, spec,
        [&](int64_t i0, auto full) {
            WeightFwdIn<E> p;
            ld_vec<E, float, true>(ts, i0, n, 0.0f, p.t0, full);          // non-temporal: common.hpp, ld_stream
            ld_vec<E, float, true>(te, i0, n, 0.0f, p.t1, full);
            ld_vec<E, float, true>(sigmas, i0, n, 0.0f, p.sg, full);
            if (prefix) ld_vec<E, float, true>(prefix, i0, n, 1.0f, p.pf, full);
            return p;
        },
        [&](int64_t i0, const bool (&act)[E], const int64_t (&)[E], const SegFwd<E> &s, const bool (&)[E], const WeightFwdIn<E> &p) {
            float sd[E], incl[E], acc[E], a[E], T[E], wgt[E];
#pragma unroll
            for (int e = 0; e < E; ++e) sd[e] = act[e] ? p.sg[e] * (p.t1[e] - p.t0[e]) : 0.0f;
            seg_scan_fwd<OpSum, E>(sd, s, carry, incl, acc);
#pragma unroll
            for (int e = 0; e < E; ++e) {
                a[e] = 1.0f - expf(-sd[e]);
                T[e] = expf(-acc[e]);
                if (prefix) T[e] = T[e] * p.pf[e];
                wgt[e] = T[e] * a[e];
            }
            if (alphas) st_vec<E>(alphas, i0, act, a);
            if (trans) st_vec<E>(trans, i0, act, T);
            if (weights) st_vec<E>(weights, i0, act, wgt);
        },
        [](int64_t) {});
}

// d/dsigma of sum(g_w w + g_T T + g_a a):
//   g_sd_i = (g_w_i T_i + g_a_i)(1 - a_i) - sum_{j>i, same ray} (g_w_j w_j + g_T_j T_j)
template <int E>
struct WeightBwdIn { float T[E], a[E], gw[E], gT[E], ga[E], t0[E], t1[E]; };
template <int E>
__global__ __launch_bounds__(kBlock) void weight_bwd_kernel(
    const int64_t *__restrict__ keys, const float *__restrict__ ts, const float *__restrict__ te,
    const float *__restrict__ trans, const float *__restrict__ alphas,
    const float *__restrict__ g_w, const float *__restrict__ g_T, const float *__restrict__ g_a,
    int64_t n, int64_t tile, int spec, float *__restrict__ g_sigmas)
{
    float carry = 0.0f;
    walk_rays_bwd<E, NFA_PF, WeightBwdIn<E>>(keys, n, wave_index(), tile, spec,
        [&](int64_t i0, auto full) {
            WeightBwdIn<E> p;
            ld_vec<E>(trans, i0, n, 0.0f, p.T, full);
            ld_vec<E>(alphas, i0, n, 0.0f, p.a, full);
            ld_vec<E>(ts, i0, n, 0.0f, p.t0, full);
            ld_vec<E>(te, i0, n, 0.0f, p.t1, full);
#pragma unroll
            for (int e = 0; e < E; ++e) p.gw[e] = p.gT[e] = p.ga[e] = 0.0f;
            if (g_w) ld_vec<E>(g_w, i0, n, 0.0f, p.gw, full);
            if (g_T) ld_vec<E>(g_T, i0, n, 0.0f, p.gT, full);
            if (g_a) ld_vec<E>(g_a, i0, n, 0.0f, p.ga, full);
            return p;
        },
        [&](int64_t i0, const bool (&act)[E], const int64_t (&)[E], const SegBwd<E> &s, const WeightBwdIn<E> &p) {
            float q[E], incl[E], suffix[E], gs[E];
#pragma unroll
            for (int e = 0; e < E; ++e) q[e] = act[e] ? p.gw[e] * (p.T[e] * p.a[e]) + p.gT[e] * p.T[e] : 0.0f;
            seg_scan_bwd<OpSum, E>(q, s, carry, incl, suffix);
#pragma unroll
            for (int e = 0; e < E; ++e) gs[e] = ((p.gw[e] * p.T[e] + p.ga[e]) * (1.0f - p.a[e]) - suffix[e]) * (p.t1[e] - p.t0[e]);
            st_vec<E>(g_sigmas, i0, act, gs);
        });
}

// ----------------------------------------------------------------------------------------
// visibility filter + compaction (three kernels, no host round trip in between)
// ----------------------------------------------------------------------------------------
// 1) keep mask, per-wave-tile kept counts and the range each wave owned (for the compaction's walk)
template <int E>
struct VisIn { float d[E], t0[E], t1[E]; };
// Keep / head bit planes of the filter (between the mask pass and the compaction).  Per chunk of 64 E samples and per SLOT, 2 E
// words: [keep bits of element e = 0 .. E-1 | head bits of element e], bit l = lane l's element.  A chunk has two slots because
// two waves can own elements of it — the wave of the nominal tile it lies in ("home", slot 0) and the one earlier wave whose last
// ray straddles into it (slot 1); each writes whole words of its own slot (bits of elements it does not own are 0), so there
// is no read-modify-write and no byte-per-sample mask: 0.5 byte per sample instead of 1 written and 1 read.
template <int E>
__device__ __forceinline__ int64_t vis_plane_words(int64_t base, int slot) { return ((base / (64 * E)) * 2 + slot) * (2 * E); }
// keep[e] of one chunk: transmittance before the sample >= eps (and alpha >= alpha_thre); `carry` is the walk's scan carry
template <int E, class P>
__device__ __forceinline__ void vis_keep(int from_alpha, const bool (&act)[E], const P &p, const SegFwd<E> &s, float &carry,
                                         float eps, float alpha_thre, uint8_t (&keep)[E])
{
    float x[E], a[E] = {}, incl[E], ex[E];
    if (from_alpha) {   // exclusive product of (1 - alpha), volrend.py:207-209
#pragma unroll
        for (int e = 0; e < E; ++e) { a[e] = p.d[e]; x[e] = act[e] ? 1.0f - a[e] : 1.0f; }
        seg_scan_fwd<OpProd, E>(x, s, carry, incl, ex);
    } else {
#pragma unroll
        for (int e = 0; e < E; ++e) x[e] = act[e] ? p.d[e] * (p.t1[e] - p.t0[e]) : 0.0f;
        // alpha is only compared with alpha_thre: with alpha_thre == 0 — configs[1]'s setting, train_ngp_nerf_occ.py:77 — nobody
        // reads it, and its expf is a quarter of this pass's VALU work (round 4; the branch is wave-uniform, the kept samples
        // are bit-identical: the transmittance path below is untouched)
        if (alpha_thre > 0.0f) {
#pragma unroll
            for (int e = 0; e < E; ++e) a[e] = 1.0f - expf(-x[e]);
        }
        seg_scan_fwd<OpSum, E>(x, s, carry, incl, ex);
#pragma unroll
        for (int e = 0; e < E; ++e) ex[e] = expf(-ex[e]);
    }
#pragma unroll
    for (int e = 0; e < E; ++e) {
        bool k = act[e] && (ex[e] >= eps);
        if (alpha_thre > 0.0f) k = k && (a[e] >= alpha_thre);
        keep[e] = k ? 1 : 0;
    }
}

template <int E>
__global__ __launch_bounds__(kBlock) void visibility_mask_kernel(
    const int64_t *__restrict__ keys, const float *__restrict__ ts, const float *__restrict__ te,
    const float *__restrict__ dens, int from_alpha, int64_t n, int64_t tile, int spec, float eps, float alpha_thre,
    uint8_t *__restrict__ mask, uint64_t *__restrict__ planes, int64_t *__restrict__ tile_cnts, int64_t *__restrict__ tile_rng,
    int64_t *__restrict__ wg_sums)
{
    __shared__ int64_t s_kept[kWavesPerBlock];
    const int64_t w = wave_index();
    const int lane = lane_id();
    int64_t kept = 0, rb = 0, re = 0;
    bool first = true;
    float carry = from_alpha ? 1.0f : 0.0f;
    if (w * tile < n)
    walk_rays_fwd<E, NFA_PF, VisIn<E>>(keys, n, w, tile, spec,
        [&](int64_t i0, auto full) {
            VisIn<E> p;
            ld_vec<E>(dens, i0, n, 0.0f, p.d, full);
            if (!from_alpha) { ld_vec<E>(ts, i0, n, 0.0f, p.t0, full); ld_vec<E>(te, i0, n, 0.0f, p.t1, full); }
            return p;
        },
        [&](int64_t i0, const bool (&act)[E], const int64_t (&)[E], const SegFwd<E> &s, const bool (&)[E], const VisIn<E> &p) {
            uint8_t keep[E];
            vis_keep<E>(from_alpha, act, p, s, carry, eps, alpha_thre, keep);
            if (mask) st_vec<E>(mask, i0, act, keep);
            // the chunk's keep / head / active bits, bit l of word e = element (lane l, e); all wave-uniform from here on
            unsigned long long kb[E], hb[E];
            int lo = 64 * E, hi = -1;
#pragma unroll
            for (int e = 0; e < E; ++e) {
                kb[e] = __ballot(keep[e] != 0);
                hb[e] = __ballot(act[e] && s.head[e]);
                kept += __popcll(kb[e]);
                const unsigned long long ab = __ballot(act[e]);
                if (ab) {
                    const int f = (__ffsll((long long)ab) - 1) * E + e, l = (63 - __clzll((long long)ab)) * E + e;
                    lo = f < lo ? f : lo;
                    hi = l > hi ? l : hi;
                }
            }
            const int64_t base = readfirstlane_i64(i0);           // (lane 0's i0)
            if (hi >= 0) {
                if (first) { rb = base + lo; first = false; }
                re = base + hi + 1;
            }
            // the words go to the chunk's "home" slot (chunk inside this wave's nominal tile) or to its "straddler" slot (beyond it:
            // the tail of this wave's last ray; the wave of that tile fills the home slot with ITS elements) — see vis_plane_words
            if (planes) {
                uint64_t *pw = planes + vis_plane_words<E>(base, base < (w + 1) * tile ? 0 : 1);
                unsigned long long v = 0;
#pragma unroll
                for (int e = 0; e < E; ++e) { v = lane == e ? kb[e] : v; v = lane == E + e ? hb[e] : v; }
                if (lane < 2 * E) pw[lane] = v;
            }
        },
        [](int64_t) {});
    if (lane == 0 && w * tile < n) { tile_cnts[w] = kept; tile_rng[2 * w] = rb; tile_rng[2 * w + 1] = re; }
    // survivors of the workgroup's four tiles: the compaction adds up one number per workgroup before its own (a quarter of the loads)
    if (lane == 0) s_kept[threadIdx.x >> 6] = kept;
    __syncthreads();
    if (threadIdx.x == 0 && wg_sums) wg_sums[blockIdx.x] = s_kept[0] + s_kept[1] + s_kept[2] + s_kept[3];
}

// 2) (only beyond kVisFusedTiles tiles) one wave per group of 64 tiles: exclusive prefix of the
// group's tile counts in place + the group total — thousands of independent waves instead of one
// workgroup crawling through every count
__global__ __launch_bounds__(kBlock) void visibility_group_scan_kernel(int64_t *__restrict__ tile_cnts, int64_t n_tiles,
                                                                       int64_t *__restrict__ group_sums)
{
    const int64_t g = (int64_t)blockIdx.x * kWavesPerBlock + wave_in_block();
    const int lane = lane_id();
    const int64_t j = g * 64 + lane;
    if (g * 64 >= n_tiles) return;
    const int64_t v = j < n_tiles ? tile_cnts[j] : 0;
    int64_t inc = v;                                  // inclusive scan over the wave (DPP, as wave_seg_scan_fwd)
    { const int64_t u = dpp_i64<kDppRowShr + 1>(inc); if ((lane & 15) >= 1) inc += u; }
    { const int64_t u = dpp_i64<kDppRowShr + 2>(inc); if ((lane & 15) >= 2) inc += u; }
    { const int64_t u = dpp_i64<kDppRowShr + 4>(inc); if ((lane & 15) >= 4) inc += u; }
    { const int64_t u = dpp_i64<kDppRowShr + 8>(inc); if ((lane & 15) >= 8) inc += u; }
    { const int64_t u = dpp_i64<kDppRowBcast15>(inc); if (lane & 16) inc += u; }
    { const int64_t u = dpp_i64<kDppRowBcast31>(inc); if (lane & 32) inc += u; }
    if (j < n_tiles) tile_cnts[j] = inc - v;
    if (lane == 63) group_sums[g] = inc;
}

// The compaction of ONE wave's range [rb, re) of the mask pass (the samples it owned there), survivors to o_*[dst ...].
// What is read per sample: the bit planes (0.5 byte), t_starts / t_ends as aligned vectors, and the KEY ONLY AT RAY HEADS (keys are
// constant along a ray: every other element takes the key of the nearest head before it — from its own lane, from a lower lane
// through ds_bpermute, or from the previous chunk) instead of 8 bytes for every sample.  Chunks without a survivor (the cut-off
// tails of opaque rays) are skipped without a load.  `tile_end`: end of the wave's nominal tile (chunks beyond it: the straddler slot).
template <int E>
__device__ __forceinline__ void vis_compact_range(const int64_t *__restrict__ keys, const float *__restrict__ ts, const float *__restrict__ te,
                                                  const uint64_t *__restrict__ planes, int64_t rb, int64_t re, int64_t tile_end, int64_t dst,
                                                  int64_t n, int64_t *__restrict__ o_keys, float *__restrict__ o_ts, float *__restrict__ o_te)
{
    constexpr int64_t CH = 64 * E;
    const int lane = lane_id();
    int64_t carry_key = 0;
    for (int64_t base = (rb / CH) * CH; base < re; base += CH) {
        const uint64_t *pw = planes + vis_plane_words<E>(base, base < tile_end ? 0 : 1);
        unsigned long long kb[E], hb[E], any_k = 0, any_h = 0;
#pragma unroll
        for (int e = 0; e < E; ++e) { kb[e] = pw[e]; hb[e] = pw[E + e]; any_k |= kb[e]; any_h |= hb[e]; }
        const int64_t i0 = base + (int64_t)lane * E;
        bool keep[E], head[E];
        int64_t key[E];
        int64_t lk = 0;                                   // key of the lane's last head (lanes of any_h)
#pragma unroll
        for (int e = 0; e < E; ++e) {
            keep[e] = (kb[e] >> lane) & 1ull;
            head[e] = (hb[e] >> lane) & 1ull;
            key[e] = 0;
            if (head[e]) { key[e] = keys[i0 + e]; lk = key[e]; }
        }
        if (any_k) {
            float t0[E], t1[E];
            ld_vec<E>(ts, i0, n, 0.0f, t0);
            ld_vec<E>(te, i0, n, 0.0f, t1);
            // the key that comes into the lane: the last head of the nearest lower lane that has one, else the previous chunk's
            const unsigned long long lower = any_h & lanes_lt(lane);
            const int src = lower ? 63 - __clzll((long long)lower) : lane;
            const int64_t from_lane = __shfl(lk, src, 64);
            int64_t in_key = lower ? from_lane : carry_key;
            int below = 0;
#pragma unroll
            for (int e = 0; e < E; ++e) below += __popcll(kb[e] & lanes_lt(lane));
            int64_t k = dst + below;
            int64_t okey[E];
#pragma unroll
            for (int e = 0; e < E; ++e) {
                in_key = head[e] ? key[e] : in_key;
                okey[e] = in_key;
                dst += __popcll(kb[e]);
            }
            bool all = true;
#pragma unroll
            for (int e = 0; e < E; ++e) all = all && keep[e];
            if (E > 1 && all) {
                // the lane's E survivors are neighbours in the output: one store per array (the destination is only element-aligned;
                // global memory takes unaligned vectors), so that a wave's store covers whole lines instead of every E-th element
                typedef int64_t kvec_t __attribute__((ext_vector_type(E), aligned(8)));
                typedef float fvec_t __attribute__((ext_vector_type(E), aligned(4)));
                kvec_t vk;
                fvec_t v0, v1;
#pragma unroll
                for (int e = 0; e < E; ++e) { vk[e] = okey[e]; v0[e] = t0[e]; v1[e] = t1[e]; }
                *reinterpret_cast<kvec_t *>(o_keys + k) = vk;
                *reinterpret_cast<fvec_t *>(o_ts + k) = v0;
                *reinterpret_cast<fvec_t *>(o_te + k) = v1;
            } else {
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    if (keep[e]) {
                        st_stream(o_keys + k, okey[e]);
                        st_stream(o_ts + k, t0[e]);
                        st_stream(o_te + k, t1[e]);
                        ++k;
                    }
                }
            }
        }
        if (any_h) {                                      // the key that leaves the chunk: its last head
            int last_e = 0;
#pragma unroll
            for (int e = 0; e < E; ++e) if (hb[e] >> (63 - __clzll((long long)any_h)) & 1ull) last_e = e;
            const int l = 63 - __clzll((long long)any_h);
            int64_t ck = key[0];
#pragma unroll
            for (int e = 1; e < E; ++e) ck = last_e == e ? key[e] : ck;
            carry_key = (int64_t)(((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)((uint64_t)ck >> 32), l) << 32) |
                                  (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(uint64_t)ck, l));
        }
    }
}

// 3) stream compaction with ballot ranks.  Where a wave's survivors go:
//   mode 0  tile_offs[w] is the global exclusive prefix (single-workgroup scan, huge inputs only)
//   mode 1  fused form for up to kVisFusedTiles tiles (every training-size call): tile_offs holds the
//           tile COUNTS and each wave sums the ones before its own (<= 64 L2 loads per lane)
//   mode 2  tile_offs holds the prefix inside the wave's group of 64 tiles (kernel 2) and group_sums the
//           group totals: the wave adds the totals of the groups before its own
// The last wave stores the total in modes 1 and 2.
constexpr int64_t kVisFusedTiles = 4096, kVisGroupedTiles = 64 * 4096;
constexpr int64_t kVisOnePassChunks = 4;               // most chunks (64 E samples) per tile of the one-pass form
template <int E>
__global__ __launch_bounds__(kBlock) void visibility_compact_kernel(
    const int64_t *__restrict__ keys, const float *__restrict__ ts, const float *__restrict__ te,
    const uint64_t *__restrict__ planes, const int64_t *__restrict__ tile_offs, const int64_t *__restrict__ group_sums,
    const int64_t *__restrict__ tile_rng, int mode, int64_t n, int64_t tile, int64_t n_tiles, int64_t *__restrict__ n_out, int64_t stamp,
    int64_t *__restrict__ o_keys, float *__restrict__ o_ts, float *__restrict__ o_te)
{
    const int64_t w_ = wave_index();
    if (w_ >= n_tiles) return;
    const int lane = lane_id();
    const int64_t rb = readfirstlane_i64(tile_rng[2 * w_]), re = readfirstlane_i64(tile_rng[2 * w_ + 1]);
    int64_t dst;
    if (mode == 1) {
        // group_sums holds the survivors per WORKGROUP of the mask pass here: add up the workgroups before this wave's, then
        // the tiles of its own workgroup before it (<= 1024 + 3 loads per wave instead of <= 4096, eight in flight)
        const int64_t wg = w_ / kWavesPerBlock;
        int64_t p = 0;
        int64_t j = lane;
        for (; j + 7 * 64 < wg; j += 8 * 64) {
            int64_t v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = group_sums[j + u * 64];
#pragma unroll
            for (int u = 0; u < 8; ++u) p += v[u];
        }
        for (; j < wg; j += 64) p += group_sums[j];
        const int64_t k = wg * kWavesPerBlock + lane;
        if (lane < kWavesPerBlock && k < w_) p += tile_offs[k];
        dst = wave_sum_i64(p);
        if (w_ == n_tiles - 1 && lane == 0) { *n_out = dst + tile_offs[w_]; if (stamp) { __threadfence_system(); n_out[1] = stamp; } }
    } else if (mode == 2) {
        const int64_t g = w_ >> 6;
        int64_t p = 0;
        for (int64_t j = lane; j < g; j += 64) p += group_sums[j];
        dst = wave_sum_i64(p) + tile_offs[w_];
        if (w_ == n_tiles - 1 && lane == 0) { *n_out = dst - tile_offs[w_] + group_sums[g]; if (stamp) { __threadfence_system(); n_out[1] = stamp; } }
    } else {
        dst = tile_offs[w_];
    }
    if (!o_keys || rb >= re) return;
    vis_compact_range<E>(keys, ts, te, planes, rb, re, (w_ + 1) * tile, dst, n, o_keys, o_ts, o_te);
}

// ----------------------------------------------------------------------------------------
// The one-pass form of the filter (option `fused_vis` = 1; never chosen automatically: on MI355X it measures slower than the two
// kernels above although it moves a third fewer bytes — round 5 built it for any size with atomic tickets and persistent
// workgroups (1.3-1.7x slower at 2^24: every tile waits for the slowest of the ~1000 groups in flight before it,
// profiles/r05_streaming.md section 3); round 6 kept only this STATIC form for calls whose workgroups are all resident at once
// (19.5 us against 9.2 + 6.9 at the training size, profiles/r06_small_n.md section 3: a publish + look-back costs two memory-side
// round trips, more than the kernel boundary it replaces).
//
// The two-kernel form reads keys / t_starts / t_ends twice and moves bit planes in between: 45 bytes per sample for 36
// algorithmic ones.  Here a wave walks its rays ONCE: the survivors of its tile are packed into LDS in sample order as they are
// found (ballot ranks), the workgroup's survivor count is published as one word of the caller's sync block (lookback.hpp: static
// ids — one tile group per workgroup by its id —, bounded waits, the block left zero), the destination is the sum of the words of
// the workgroups before, and the LDS image leaves as one contiguous, coalesced copy.  n_out[0] = -1: the look-back gave up,
// nothing was stored (the caller runs the two kernels).
//
// A ray that straddles far beyond the nominal tile can give a wave more survivors than its LDS image holds.  From the chunk that
// would overflow on, the wave only counts and leaves keep bytes (in the caller's mask, or in the workspace); once its destination
// is known it compacts that remainder as the compaction kernel would.
// ----------------------------------------------------------------------------------------
template <int E>
__global__ __launch_bounds__(kBlock) void visibility_onepass_kernel(
    const int64_t *__restrict__ keys, const float *__restrict__ ts, const float *__restrict__ te,
    const float *__restrict__ dens, int from_alpha, int64_t n, int64_t tile, float eps, float alpha_thre,
    uint8_t *__restrict__ mask, uint8_t *__restrict__ ov_mask, uint64_t *__restrict__ sync, uint64_t spin, int64_t n_tiles, int cap,
    int64_t *__restrict__ n_out, int64_t stamp, int64_t *__restrict__ o_keys, float *__restrict__ o_ts, float *__restrict__ o_te)
{
    extern __shared__ __align__(16) uint8_t vis_smem[];
    __shared__ int64_t s_kept[kWavesPerBlock], s_excl;
    const int wv = threadIdx.x >> 6, lane = lane_id();
    const int64_t n_groups = (n_tiles + kWavesPerBlock - 1) / kWavesPerBlock;
    // the wave's LDS image: keys[cap] | t_starts[cap] | t_ends[cap]  (addressed through vis_smem itself: pointers derived from it
    // and selected against the global outputs lose their LDS address space and the compiler parks them in scratch)
    const uint32_t img = (uint32_t)wv * (uint32_t)cap * 16u;
#define L_KEYS(j) (*(int64_t *)(vis_smem + img + 8u * (uint32_t)(j)))
#define L_TS(j) (*(float *)(vis_smem + img + 8u * (uint32_t)cap + 4u * (uint32_t)(j)))
#define L_TE(j) (*(float *)(vis_smem + img + 12u * (uint32_t)cap + 4u * (uint32_t)(j)))
    const int64_t t = blockIdx.x;                      // (the host launches exactly n_groups workgroups)
    const int64_t w = t * kWavesPerBlock + wv;
    int cnt = 0;                       // survivors in the LDS image
    int64_t kept = 0;                  // survivors of the tile
    bool over = false;                 // the image is full: chunks from ov_b on are only counted, their keep bytes are in ov_mask
    int64_t ov_b = 0, ov_e = 0;
    float carry = from_alpha ? 1.0f : 0.0f;
    walk_rays_fwd<E, NFA_PF, VisIn<E>>(keys, n, w, tile, 0,
        [&](int64_t i0, auto full) {
            VisIn<E> p;
            ld_vec<E>(dens, i0, n, 0.0f, p.d, full);
            ld_vec<E>(ts, i0, n, 0.0f, p.t0, full);
            ld_vec<E>(te, i0, n, 0.0f, p.t1, full);
            return p;
        },
        [&](int64_t i0, const bool (&act)[E], const int64_t (&key)[E], const SegFwd<E> &s, const bool (&)[E], const VisIn<E> &p) {
            uint8_t keep[E];
            vis_keep<E>(from_alpha, act, p, s, carry, eps, alpha_thre, keep);
            if (mask) st_vec<E>(mask, i0, act, keep);
            // rank of element (lane, e) among the chunk's survivors, in sample order (index = base + lane E + e)
            int below = 0, chunk_kept = 0;
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const unsigned long long b = __ballot(keep[e] != 0);
                below += __popcll(b & lanes_lt(lane));
                chunk_kept += __popcll(b);
            }
            if (!over && cnt + chunk_kept > cap) { over = true; ov_b = i0 - (int64_t)lane * E; }      // (wave-uniform)
            if (!over) {
                int r = cnt + below;
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    if (keep[e]) {
                        L_KEYS(r) = key[e];
                        L_TS(r) = p.t0[e];
                        L_TE(r) = p.t1[e];
                        ++r;
                    }
                }
                cnt += chunk_kept;
            } else {
                if (!mask) st_vec<E>(ov_mask, i0, act, keep);
                int le = -1;
#pragma unroll
                for (int e = 0; e < E; ++e) if (act[e]) le = e;
                const unsigned long long am = __ballot(le >= 0);
                if (am) {
                    const int l = 63 - __clzll((long long)am);
                    ov_e = i0 - (int64_t)lane * E + l * E + __builtin_amdgcn_readlane(le, l) + 1;
                }
            }
            kept += chunk_kept;
        },
        [](int64_t) {});
    if (lane == 0) s_kept[wv] = kept;
    __syncthreads();
    if (wv == 0) {                      // one word per workgroup: the four tiles' survivors
        const int64_t tot = s_kept[0] + s_kept[1] + s_kept[2] + s_kept[3];
        const int64_t excl = sync_publish_and_lookback(sync, t, tot, 0, 0, lane, spin);
        if (lane == 0) s_excl = excl;
        // the call's result comes from the workgroup that leaves last: only it knows that no look-back of the launch gave up
        sync_deposit(sync, excl, excl + tot, t == n_groups - 1, lane);
        sync_leave(sync, n_groups, lane, [&](bool gave_up, int64_t total) {
            if (lane != 0) return;
            *n_out = gave_up ? -1 : total;
            if (stamp) { __threadfence_system(); n_out[1] = stamp; }
        });
    }
    __syncthreads();
    int64_t dst = s_excl;
    if (dst < 0) return;
#pragma unroll
    for (int k = 0; k < kWavesPerBlock - 1; ++k) dst += k < wv ? s_kept[k] : 0;
    for (int j = lane; j < cnt; j += 64) {          // the LDS image leaves as one contiguous copy
        st_stream(o_keys + dst + j, L_KEYS(j));
        st_stream(o_ts + dst + j, L_TS(j));
        st_stream(o_te + dst + j, L_TE(j));
    }
    if (over) {                                     // the counted-only remainder, as visibility_compact_kernel does it
        __threadfence();                            // this wave's keep bytes, stored by other lanes, before they are loaded
        dst += cnt;
        for (int64_t base = ov_b; base < ov_e; base += 64) {
            const int64_t i = base + lane;
            const bool keepb = i < ov_e && ov_mask[i];
            const unsigned long long b = __ballot(keepb);
            if (keepb) {
                const int64_t k = dst + __popcll(b & lanes_lt(lane));
                st_stream(o_keys + k, keys[i]);
                st_stream(o_ts + k, ts[i]);
                st_stream(o_te + k, te[i]);
            }
            dst += __popcll(b);
        }
    }
#undef L_KEYS
#undef L_TS
#undef L_TE
}

// ----------------------------------------------------------------------------------------
// accumulate_along_rays: out[r, c0 + c] += sum_i w_i v[i, c0 + c]   (DC channels per launch)
// ----------------------------------------------------------------------------------------
template <int E, int DC>
struct AccumIn { float w[E], v[E][DC]; };
template <int E, int DC>
__global__ __launch_bounds__(kBlock) void accumulate_kernel(
    const int64_t *__restrict__ keys, const float *__restrict__ weights, const float *__restrict__ values,
    int64_t n, int64_t tile, int spec, int D, int c0, int64_t n_rays, float *__restrict__ out)
{
    const int lane = lane_id();
    float carry[DC];
#pragma unroll
    for (int c = 0; c < DC; ++c) carry[c] = 0.0f;
    // two further chunks requested ahead: this kernel moves only 12 + 4 DC bytes per sample, and with one chunk in
    // flight per wave the bytes in flight per CU (not the HBM) bounded it (Little's law)
    walk_rays_fwd<E, (E == 1 ? 2 : 1), AccumIn<E, DC>>(keys, n, wave_index(), tile, spec,
        [&](int64_t i0, auto full) {
            AccumIn<E, DC> p;
            ld_vec<E>(weights, i0, n, 0.0f, p.w, full);
            if (values && D == DC) ld_vec_strided<E, DC>(values, i0, n, 1.0f, p.v, full);
            else {
#pragma unroll
                for (int e = 0; e < E; ++e)
#pragma unroll
                    for (int c = 0; c < DC; ++c) p.v[e][c] = (values && i0 + e < n) ? values[(i0 + e) * D + c0 + c] : 1.0f;
            }
            return p;
        },
        [&](int64_t, const bool (&act)[E], const int64_t (&key)[E], const SegFwd<E> &s, const bool (&tail)[E], const AccumIn<E, DC> &p) {
#pragma unroll
            for (int c = 0; c < DC; ++c) {
                float x[E], incl[E], excl[E];
#pragma unroll
                for (int e = 0; e < E; ++e) x[e] = act[e] ? p.w[e] * p.v[e][c] : 0.0f;
                seg_scan_fwd<OpSum, E>(x, s, carry[c], incl, excl);
#pragma unroll
                for (int e = 0; e < E; ++e)
                    if (tail[e] && key[e] >= 0 && key[e] < n_rays) unsafeAtomicAdd(out + key[e] * D + c0 + c, incl[e]);
            }
        },
        [&](int64_t key) {
            if (lane == 0 && key >= 0 && key < n_rays) {
#pragma unroll
                for (int c = 0; c < DC; ++c) unsafeAtomicAdd(out + key * D + c0 + c, carry[c]);
            }
        });
}

__global__ __launch_bounds__(kBlock) void accumulate_bwd_kernel(
    const int64_t *__restrict__ keys, const float *__restrict__ weights, const float *__restrict__ values,
    const float *__restrict__ g_out, int64_t n, int D, int64_t n_rays, float *__restrict__ g_w, float *__restrict__ g_v)
{
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
        const int64_t r = keys[i];
        const bool in = r >= 0 && r < n_rays;     // the forward pass skipped such samples (e.g. the -1 gaps of unpack_info): zero gradient
        const float w = weights[i];
        float acc = 0.0f;
        for (int c = 0; c < D; ++c) {
            const float g = in ? g_out[r * D + c] : 0.0f;
            if (values) { acc += g * values[i * D + c]; if (g_v) g_v[i * D + c] = w * g; }
            else acc += g;
        }
        if (g_w) g_w[i] = acc;
    }
}

// ----------------------------------------------------------------------------------------
// fused rendering
// ----------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void fill_rays_kernel(int64_t n_rays, const float *__restrict__ bkgd,
                                                           float *__restrict__ colors, float *__restrict__ opac,
                                                           float *__restrict__ depth)
{
    for (int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x; r < n_rays; r += (int64_t)gridDim.x * kBlock) {
        colors[3 * r] = bkgd ? bkgd[0] : 0.0f;
        colors[3 * r + 1] = bkgd ? bkgd[1] : 0.0f;
        colors[3 * r + 2] = bkgd ? bkgd[2] : 0.0f;
        opac[r] = 0.0f;
        depth[r] = 0.0f;
    }
}

template <int E>
struct RenderFwdIn { float t0[E], t1[E], sg[E], rgb[E][3]; static constexpr bool kStreamKeys = true; };
// The rays WITHOUT a sample get (background, 0, 0) in the same launch (round 6; a launch of its own before — fill_rays_kernel, still
// what calls without samples and calls beyond kFillFoldMaxSamples take): ray_indices ascend (this entry point's contract: the
// reference builds its chunk starts in ray order, pack.py:38-46), so the rays without a sample are the gaps between consecutive
// keys, before the first and behind the last.  Workgroups main_blocks ... gridDim.x - 1 of the launch do nothing else: they stream
// over the keys once more (8 of the call's 60 bytes per sample — why large calls keep the separate launch, whose few microseconds
// do not matter there), a lane fills a gap of up to two rays in front of its element, the wave together a longer one.  The walk
// of the other workgroups is untouched: filling the gaps at its ray heads costs 20 VGPRs there (69 -> 89 at two samples per lane).
constexpr int64_t kFillFoldMaxSamples = (int64_t)1 << 20;
__device__ __forceinline__ void fill_ray_gaps(const int64_t *__restrict__ keys, int64_t n, int64_t n_rays, int64_t block, int64_t n_blocks,
                                              float bk0, float bk1, float bk2, float *__restrict__ colors, float *__restrict__ opac,
                                              float *__restrict__ depth)
{
    const int lane = lane_id();
    auto ray_empty = [&](int64_t r) {
        colors[3 * r] = bk0; colors[3 * r + 1] = bk1; colors[3 * r + 2] = bk2;
        opac[r] = 0.0f;
        depth[r] = 0.0f;
    };
    const int64_t stride = n_blocks * kBlock;
    // element i closes the gap (keys[i - 1], keys[i]); "element n" the one behind the last key
    for (int64_t i0 = block * kBlock; i0 <= n; i0 += stride) {
        const int64_t i = i0 + threadIdx.x;
        int64_t a = 0, b = 0;
        if (i <= n) {
            a = i > 0 ? keys[i - 1] + 1 : 0;
            b = i < n ? keys[i] : n_rays;
            a = a < 0 ? 0 : a;
            b = b > n_rays ? n_rays : b;
        }
        if (a < b) ray_empty(a);
        if (a + 1 < b) ray_empty(a + 1);
        unsigned long long m = __ballot(a + 2 < b);
        while (m) {
            const int l = __ffsll((long long)m) - 1;
            m &= m - 1;
            const int64_t a0 = readlane_i64_dyn(a, l) + 2, b0 = readlane_i64_dyn(b, l);
            for (int64_t r = a0 + lane; r < b0; r += 64) ray_empty(r);
        }
    }
}

template <int E>
__global__ __launch_bounds__(kBlock) void rendering_fwd_kernel(
    const int64_t *__restrict__ keys, const float *__restrict__ ts, const float *__restrict__ te,
    const float *__restrict__ sigmas, const float *__restrict__ rgbs, int64_t n, int64_t tile, int spec, int64_t n_rays,
    const float *__restrict__ bkgd, int expected_depths, int main_blocks,
    float *__restrict__ weights, float *__restrict__ trans, float *__restrict__ alphas,
    float *__restrict__ colors, float *__restrict__ opac, float *__restrict__ depth)
{
    const int lane = lane_id();
    float c_sd = 0.f, c_r = 0.f, c_g = 0.f, c_b = 0.f, c_w = 0.f, c_m = 0.f;
    float bk0 = 0.f, bk1 = 0.f, bk2 = 0.f;
    if (bkgd) { bk0 = bkgd[0]; bk1 = bkgd[1]; bk2 = bkgd[2]; }
    if ((int)blockIdx.x >= main_blocks) {
        fill_ray_gaps(keys, n, n_rays, (int64_t)blockIdx.x - main_blocks, (int64_t)gridDim.x - main_blocks, bk0, bk1, bk2, colors, opac, depth);
        return;
    }
    auto ray_out = [&](int64_t key, float sr, float sg, float sb, float sw, float sm) {
        if (key < 0 || key >= n_rays) return;
        const float rem = 1.0f - sw;
        colors[3 * key] = bkgd ? sr + bk0 * rem : sr;
        colors[3 * key + 1] = bkgd ? sg + bk1 * rem : sg;
        colors[3 * key + 2] = bkgd ? sb + bk2 * rem : sb;
        opac[key] = sw;
        depth[key] = expected_depths ? sm / fmaxf(sw, kEpsF32) : sm;
    };
    walk_rays_fwd<E, NFA_PF, RenderFwdIn<E>>(keys, n, wave_index(), tile, spec,
        [&](int64_t i0, auto full) {
            RenderFwdIn<E> p;
            ld_vec<E, float, true>(ts, i0, n, 0.0f, p.t0, full);          // non-temporal: common.hpp, ld_stream
            ld_vec<E, float, true>(te, i0, n, 0.0f, p.t1, full);
            ld_vec<E, float, true>(sigmas, i0, n, 0.0f, p.sg, full);
            ld_vec_strided<E, 3, true>(rgbs, i0, n, 0.0f, p.rgb, full);
            return p;
        },
        [&](int64_t i0, const bool (&act)[E], const int64_t (&key)[E], const SegFwd<E> &s, const bool (&tail)[E], const RenderFwdIn<E> &p) {
            float sd[E], incl[E], acc[E], a[E], T[E], w[E];
#pragma unroll
            for (int e = 0; e < E; ++e) sd[e] = act[e] ? p.sg[e] * (p.t1[e] - p.t0[e]) : 0.0f;
            seg_scan_fwd<OpSum, E>(sd, s, c_sd, incl, acc);
#pragma unroll
            for (int e = 0; e < E; ++e) {
                a[e] = 1.0f - expf(-sd[e]);
                T[e] = expf(-acc[e]);
                w[e] = act[e] ? T[e] * a[e] : 0.0f;
            }
            st_vec<E>(alphas, i0, act, a);
            st_vec<E>(trans, i0, act, T);
            st_vec<E>(weights, i0, act, w);
            float x[E], sr[E], sg[E], sb[E], sw[E], sm[E], ex[E];
#pragma unroll
            for (int e = 0; e < E; ++e) x[e] = w[e] * p.rgb[e][0];
            seg_scan_fwd<OpSum, E>(x, s, c_r, sr, ex);
#pragma unroll
            for (int e = 0; e < E; ++e) x[e] = w[e] * p.rgb[e][1];
            seg_scan_fwd<OpSum, E>(x, s, c_g, sg, ex);
#pragma unroll
            for (int e = 0; e < E; ++e) x[e] = w[e] * p.rgb[e][2];
            seg_scan_fwd<OpSum, E>(x, s, c_b, sb, ex);
            seg_scan_fwd<OpSum, E>(w, s, c_w, sw, ex);
#pragma unroll
            for (int e = 0; e < E; ++e) x[e] = w[e] * ((p.t0[e] + p.t1[e]) / 2.0f);
            seg_scan_fwd<OpSum, E>(x, s, c_m, sm, ex);
#pragma unroll
            for (int e = 0; e < E; ++e)
                if (tail[e]) ray_out(key[e], sr[e], sg[e], sb[e], sw[e], sm[e]);
        },
        [&](int64_t key) { if (lane == 0) ray_out(key, c_r, c_g, c_b, c_w, c_m); });
}

// EXT: the caller has gradients w.r.t. the per-sample extras (weights / trans / alphas — a loss on extras["weights"], e.g. the
// distortion loss).  The usual training step has none, and their three payload slots per element and queued chunk were what kept
// the kernel at 98 VGPRs (4 waves per SIMD; forcing 5 through __launch_bounds__ spilled 20 registers and ran 1.9x slower,
// profiles/r05_streaming.md): without them (round 6) it fits 5.
template <int E, bool EXT>
struct RenderBwdIn { float T[E], a[E], gw[EXT ? E : 1], gT[EXT ? E : 1], ga[EXT ? E : 1], t0[E], t1[E], rgb[E][3]; };
template <int E, bool EXT>
__global__ __launch_bounds__(kBlock) void rendering_bwd_kernel(
    const int64_t *__restrict__ keys, const float *__restrict__ ts, const float *__restrict__ te,
    const float *__restrict__ rgbs, const float *__restrict__ trans,
    const float *__restrict__ alphas, const float *__restrict__ opac, const float *__restrict__ depth,
    int64_t n, int64_t tile, int spec, int64_t n_rays, const float *__restrict__ bkgd, int expected_depths,
    const float *__restrict__ g_colors, const float *__restrict__ g_opac, const float *__restrict__ g_depth,
    const float *__restrict__ g_w_ext, const float *__restrict__ g_T_ext, const float *__restrict__ g_a_ext,
    float *__restrict__ g_sigmas, float *__restrict__ g_rgbs)
{
    float carry = 0.0f;
    float bk0 = 0.f, bk1 = 0.f, bk2 = 0.f;
    if (bkgd) { bk0 = bkgd[0]; bk1 = bkgd[1]; bk2 = bkgd[2]; }
    walk_rays_bwd<E, NFA_PF, RenderBwdIn<E, EXT>>(keys, n, wave_index(), tile, spec,
        [&](int64_t i0, auto full) {
            RenderBwdIn<E, EXT> p;
            // (the weights are not loaded: the forward pass stored w = T * alpha — rendering_fwd_kernel above, one rounding — and the
            // same product of the same two floats is formed below: 4 of 60 bytes per sample less)
            ld_vec<E>(trans, i0, n, 0.0f, p.T, full);
            ld_vec<E>(alphas, i0, n, 0.0f, p.a, full);
            ld_vec<E>(ts, i0, n, 0.0f, p.t0, full);
            ld_vec<E>(te, i0, n, 0.0f, p.t1, full);
            ld_vec_strided<E, 3>(rgbs, i0, n, 0.0f, p.rgb, full);
            if constexpr (EXT) {
#pragma unroll
                for (int e = 0; e < E; ++e) p.gw[e] = p.gT[e] = p.ga[e] = 0.0f;
                if (g_w_ext) ld_vec<E>(g_w_ext, i0, n, 0.0f, p.gw, full);
                if (g_T_ext) ld_vec<E>(g_T_ext, i0, n, 0.0f, p.gT, full);
                if (g_a_ext) ld_vec<E>(g_a_ext, i0, n, 0.0f, p.ga, full);
            }
            return p;
        },
        [&](int64_t i0, const bool (&act)[E], const int64_t (&key)[E], const SegBwd<E> &s, const RenderBwdIn<E, EXT> &p) {
            float gw[E], q[E], incl[E], suffix[E], gs[E], grgb[E][3], w[E];
            bool wr[E];
#pragma unroll
            for (int e = 0; e < E; ++e) {
                w[e] = p.T[e] * p.a[e];
                gw[e] = EXT ? p.gw[EXT ? e : 0] : 0.0f;
                wr[e] = act[e] && key[e] >= 0 && key[e] < n_rays;
                grgb[e][0] = grgb[e][1] = grgb[e][2] = 0.0f;
                if (wr[e]) {
                    const int64_t k = key[e];
                    float gc0 = 0.f, gc1 = 0.f, gc2 = 0.f;
                    if (g_colors) { gc0 = g_colors[3 * k]; gc1 = g_colors[3 * k + 1]; gc2 = g_colors[3 * k + 2]; }
                    float go = g_opac ? g_opac[k] : 0.0f;
                    float gacc = 0.0f;                      // dL/d(sum w m)
                    if (g_depth) {
                        const float gd = g_depth[k];
                        if (expected_depths) {
                            const float O = opac[k];
                            gacc = gd / fmaxf(O, kEpsF32);
                            if (O > kEpsF32) go -= gd * depth[k] / O;
                        } else gacc = gd;
                    }
                    if (bkgd) go -= gc0 * bk0 + gc1 * bk1 + gc2 * bk2;
                    gw[e] += gc0 * p.rgb[e][0] + gc1 * p.rgb[e][1] + gc2 * p.rgb[e][2] + go + gacc * ((p.t0[e] + p.t1[e]) / 2.0f);
                    grgb[e][0] = w[e] * gc0; grgb[e][1] = w[e] * gc1; grgb[e][2] = w[e] * gc2;
                }
                q[e] = act[e] ? (EXT ? gw[e] * w[e] + p.gT[EXT ? e : 0] * p.T[e] : gw[e] * w[e]) : 0.0f;
            }
            if (g_rgbs) {
                bool all = true;
#pragma unroll
                for (int e = 0; e < E; ++e) all = all && wr[e];
                if (E > 1 && __ballot(all) == ~0ull) {
                    // interior chunks: the lane's 3 E floats are contiguous (and 4 E-byte aligned: i0 is a multiple of E) — three E-wide
                    // stores instead of 3 E scalar ones (the mirror of ld_vec_strided)
                    typedef float vec_t __attribute__((ext_vector_type(E)));
                    float flat[3 * E];
#pragma unroll
                    for (int e = 0; e < E; ++e)
#pragma unroll
                        for (int c = 0; c < 3; ++c) flat[e * 3 + c] = grgb[e][c];
#pragma unroll
                    for (int j = 0; j < 3; ++j) {
                        vec_t v;
#pragma unroll
                        for (int e = 0; e < E; ++e) v[e] = flat[j * E + e];
                        *reinterpret_cast<vec_t *>(g_rgbs + 3 * i0 + j * E) = v;
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < E; ++e)
                        if (wr[e]) { g_rgbs[3 * (i0 + e)] = grgb[e][0]; g_rgbs[3 * (i0 + e) + 1] = grgb[e][1]; g_rgbs[3 * (i0 + e) + 2] = grgb[e][2]; }
                }
            }
            seg_scan_bwd<OpSum, E>(q, s, carry, incl, suffix);
#pragma unroll
            for (int e = 0; e < E; ++e) gs[e] = ((EXT ? gw[e] * p.T[e] + p.ga[EXT ? e : 0] : gw[e] * p.T[e]) * (1.0f - p.a[e]) - suffix[e]) * (p.t1[e] - p.t0[e]);
            if (g_sigmas) st_vec<E>(g_sigmas, i0, act, gs);
        });
}

// ----------------------------------------------------------------------------------------
// pack_info / unpack_info
// ----------------------------------------------------------------------------------------
// pack.py:38-46: cnt[r] = #{i : ray_indices[i] == r} (index_add_ of ones), start[r] = cumsum(cnt)[r] - cnt[r] — for
// ANY order of ray_indices.  Ascending input (what every producer on this path emits) takes one launch:
// start[r] = first index with key >= r by binary search, the next ray's start from the neighbouring lane
// (lane 63 gallops forward from its own start instead: counts are small).  A descent anywhere in the input is
// detected first (pack_check_kernel) and switches to the histogram form of the reference: wave-aggregated int64
// atomics (one per run of equal keys, deterministic: integers) + one exclusive scan.  The flag lives in
// packed[0] — the start of ray 0, which is 0 in either form — so no workspace is needed; every kernel of the
// sequence is launched unconditionally and returns at once when the flag says it is not its turn.
__global__ __launch_bounds__(kBlock) void pack_check_kernel(const int64_t *__restrict__ keys, int64_t n, int64_t *__restrict__ packed)
{
    bool bad = false;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x + 1; i < n; i += (int64_t)gridDim.x * kBlock)
        bad |= keys[i] < keys[i - 1];
    if (__ballot(bad) && lane_id() == 0) packed[0] = 1;      // plain store: every writer writes the same value
}

__global__ __launch_bounds__(kBlock) void pack_info_kernel(const int64_t *__restrict__ keys, int64_t n, int64_t n_rays,
                                                           int64_t *__restrict__ packed)
{
    if (__builtin_nontemporal_load(packed) != 0) return;      // unsorted input: the histogram kernels do the work
    const int lane = lane_id();
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t r0 = (int64_t)blockIdx.x * kBlock; r0 < n_rays; r0 += stride) {
        const int64_t r = r0 + threadIdx.x;
        int64_t lo = 0, hi = n;                 // first index with key >= r  (n for r >= n_rays: the loop still converges)
        while (lo < hi) { const int64_t m = lo + ((hi - lo) >> 1); if (keys[m] < r) lo = m + 1; else hi = m; }
        const int64_t first = lo;
        int64_t next = lane_next_i64(first);
        if (lane == 63) {                       // first index with key > r: gallop, then bisect the last gap
            int64_t step = 1, a = first, b = first;
            while (b < n && keys[b] <= r) { a = b + 1; b += step; step <<= 1; }
            if (b > n) b = n;
            while (a < b) { const int64_t m = a + ((b - a) >> 1); if (keys[m] <= r) a = m + 1; else b = m; }
            next = a;
        }
        if (r < n_rays && r > 0) packed[2 * r] = first;       // packed[0] (= 0) doubles as the flag and is left alone
        if (r < n_rays) packed[2 * r + 1] = next - first;
    }
}

// the three kernels of the unsorted form; all of them return at once for sorted input (flag 0)
__global__ __launch_bounds__(kBlock) void pack_hist_zero_kernel(int64_t n_rays, int64_t *__restrict__ packed)
{
    if (__builtin_nontemporal_load(packed) == 0) return;
    for (int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x; r < n_rays; r += (int64_t)gridDim.x * kBlock) packed[2 * r + 1] = 0;
}

__global__ __launch_bounds__(kBlock) void pack_hist_kernel(const int64_t *__restrict__ keys, int64_t n, int64_t n_rays,
                                                           int64_t *__restrict__ packed)
{
    if (__builtin_nontemporal_load(packed) == 0) return;
    const int lane = lane_id();
    const int64_t n64 = (n + 63) & ~(int64_t)63;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n64; i += (int64_t)gridDim.x * kBlock) {
        const bool active = i < n;
        const int64_t k = active ? keys[i] : -1;
        const int64_t prev = lane_prev_i64(k);
        const bool head = active && (lane == 0 || prev != k);            // first lane of a run of equal keys in this wave
        const unsigned long long heads = __ballot(head);
        const unsigned long long act = __ballot(active);
        if (head && k >= 0 && k < n_rays) {
            const unsigned long long later = heads & ~lanes_le(lane);
            const int end = later ? (__ffsll((long long)later) - 1) : (64 - (int)__clzll((long long)act));
            atomicAdd((unsigned long long *)(packed + 2 * k + 1), (unsigned long long)(end - lane));
        }
    }
}

__global__ __launch_bounds__(1024) void pack_hist_scan_kernel(int64_t n_rays, int64_t *__restrict__ packed)
{
    if (__builtin_nontemporal_load(packed) == 0) return;
    __shared__ int64_t wsum[16];
    __shared__ int64_t carry_s;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    const int lane = lane_id(), wave = threadIdx.x >> 6;
    for (int64_t base = 0; base < n_rays; base += 1024) {
        const int64_t r = base + threadIdx.x;
        const int64_t v = r < n_rays ? packed[2 * r + 1] : 0;
        int64_t inc = v;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int64_t u = __shfl_up(inc, off, 64);
            if (lane >= off) inc += u;
        }
        if (lane == 63) wsum[wave] = inc;
        __syncthreads();
        int64_t woff = 0, tot = 0;
        for (int w = 0; w < 16; ++w) { const int64_t t = wsum[w]; if (w < wave) woff += t; tot += t; }
        const int64_t carry = carry_s;
        if (r < n_rays && r > 0) packed[2 * r] = carry + woff + inc - v;
        __syncthreads();
        if (threadIdx.x == 0) carry_s = carry + tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) packed[0] = 0;        // clear the flag last: the start of ray 0
}

__global__ __launch_bounds__(kBlock) void unpack_info_kernel(const int64_t *__restrict__ starts, const int64_t *__restrict__ cnts,
                                                             int64_t n_rays, int64_t n, int64_t *__restrict__ keys)
{
    const int sub = threadIdx.x & 15;
    for (int64_t r = (int64_t)blockIdx.x * (kBlock / 16) + (threadIdx.x >> 4); r < n_rays; r += (int64_t)gridDim.x * (kBlock / 16)) {
        const int64_t s = starts[r], c = cnts[r];
        for (int64_t k = sub; k < c; k += 16)
            if (s + k >= 0 && s + k < n) keys[s + k] = r;
    }
}

// sample midpoints in world space: p_i = o[r_i] + d[r_i] * ((t0_i + t1_i) / 2) — the line every
// rgb_sigma_fn / sigma_fn of the reference's examples starts with (examples/utils.py:96-101), six
// ATen launches there.  One lane per output float (coalesced [N,3] stores); same operation order as
// the torch expression, no contraction: bit-identical.  `dirs` (nullable) = d[r_i], the view
// directions handed to the field.
__global__ __launch_bounds__(kBlock) void sample_positions_kernel(
    const float *__restrict__ rays_o, const float *__restrict__ rays_d, int64_t n_rays,
    const int64_t *__restrict__ keys, const float *__restrict__ ts, const float *__restrict__ te, int64_t n,
    float *__restrict__ positions, float *__restrict__ dirs)
{
    const int64_t total = 3 * n;
    for (int64_t k = (int64_t)blockIdx.x * kBlock + threadIdx.x; k < total; k += (int64_t)gridDim.x * kBlock) {
        const int64_t i = k / 3;
        const int c = (int)(k - 3 * i);
        const int64_t r = keys[i];
        float o = 0.0f, d = 0.0f;
        if (r >= 0 && r < n_rays) { o = rays_o[3 * r + c]; d = rays_d[3 * r + c]; }
        const float mid = (ts[i] + te[i]) / 2.0f;
        positions[k] = o + d * mid;
        if (dirs) dirs[k] = d;
    }
}


}  // namespace
}  // namespace nfa

using namespace nfa;

NFA_EXPORT int nfa_render_weight_from_density_fwd(const int64_t *ray_indices, const float *t_starts,
                                                  const float *t_ends, const float *sigmas,
                                                  const float *prefix_trans, int64_t n,
                                                  float *weights, float *trans, float *alphas, void *stream)
{
    NFA_REQUIRE(n >= 0, "render_weight_from_density_fwd: n < 0");
    if (n == 0) return NFA_OK;
    NFA_REQUIRE(ray_indices && t_starts && t_ends && sigmas, "render_weight_from_density_fwd: NULL input");
    const TilePlan pl = pick_plan(n, aligned16({ray_indices, t_starts, t_ends, sigmas, prefix_trans, weights, trans, alphas}));
    NFA_LAUNCH_TILED(weight_fwd_kernel, pl, n, (hipStream_t)stream,
                     ray_indices, t_starts, t_ends, sigmas, prefix_trans, n, pl.tile, pl.spec, weights, trans, alphas);
    return check_launch("weight_fwd_kernel");
}

NFA_EXPORT int nfa_render_weight_from_density_bwd(const int64_t *ray_indices, const float *t_starts,
                                                  const float *t_ends, const float *sigmas, const float *trans,
                                                  const float *alphas, const float *g_weights,
                                                  const float *g_trans, const float *g_alphas, int64_t n,
                                                  float *g_sigmas, void *stream)
{
    (void)sigmas;
    NFA_REQUIRE(n >= 0, "render_weight_from_density_bwd: n < 0");
    if (n == 0) return NFA_OK;
    NFA_REQUIRE(ray_indices && t_starts && t_ends && trans && alphas && g_sigmas, "render_weight_from_density_bwd: NULL pointer");
    const TilePlan pl = pick_plan(n, aligned16({ray_indices, t_starts, t_ends, trans, alphas, g_weights, g_trans, g_alphas, g_sigmas}));
    NFA_LAUNCH_TILED(weight_bwd_kernel, pl, n, (hipStream_t)stream,
                     ray_indices, t_starts, t_ends, trans, alphas, g_weights, g_trans, g_alphas, n, pl.tile, pl.spec, g_sigmas);
    return check_launch("weight_bwd_kernel");
}

// workspace layout: [ front: bit planes of the mask pass (vis_plane_words; 32 bytes per 64 samples) — or, one-pass form, the keep
// bytes of overflowing tiles: n ][ tile_cnts: T int64 ][ tile_offs: T int64 ][ tile_rng: 2 T int64 ]
// (sized for the smallest tile any plan of this n can pick: one element per lane)
static inline int64_t vis_tiles(int64_t n) { return ceil_div(n > 0 ? n : 1, pick_plan(n, false).tile); }
static inline int64_t vis_front_bytes(int64_t n) {
    const int64_t m = n > 0 ? n : 1;
    const int64_t planes = (ceil_div(m, 64) + 4) * 32;              // (a chunk of 64 E samples: 2 slots x 2 E words; + one chunk of 4 E)
    return ceil_div(planes > m ? planes : m, 16) * 16;
}
NFA_EXPORT int64_t nfa_visibility_workspace_bytes(int64_t n) {
    // (4 words per tile behind the front; the one-pass form only uses the front, for the keep bytes of overflowing tiles)
    return vis_front_bytes(n) + 4 * (int64_t)sizeof(int64_t) * vis_tiles(n);
}

NFA_EXPORT int nfa_visibility_compact(const int64_t *ray_indices, const float *t_starts, const float *t_ends,
                                      const float *dens, int32_t from_alpha, int64_t n, float early_stop_eps,
                                      float alpha_thre, int64_t *out_ray_indices, float *out_t_starts,
                                      float *out_t_ends, uint8_t *out_mask, int64_t *n_out, void *workspace,
                                      void *stream)
{
    return nfa_visibility_compact_stamped(ray_indices, t_starts, t_ends, dens, from_alpha, n, early_stop_eps, alpha_thre, out_ray_indices,
                                          out_t_starts, out_t_ends, out_mask, n_out, 0, workspace, stream);
}

static int visibility_compact_impl(const int64_t *ray_indices, const float *t_starts, const float *t_ends, const float *dens, int32_t from_alpha,
                                   int64_t n, float early_stop_eps, float alpha_thre, int64_t *out_ray_indices, float *out_t_starts,
                                   float *out_t_ends, uint8_t *out_mask, int64_t *n_out, int64_t stamp, void *workspace, void *sync, int resume,
                                   void *stream);

static bool vis_fusable(int64_t n, const TilePlan &pl) {
    if (n <= 0 || opt(OPT_FUSED_VIS, 0) == 0) return false;
    const int64_t T = ceil_div(n, pl.tile), n_wg = tile_blocks(n, pl.tile);
    return T <= kVisFusedTiles && n_wg <= 3 * kNumCU && n_wg <= kSyncMaxBlocks;
}
// does nfa_visibility_compact_sync take its single-launch form for n samples (pointers 16-byte aligned or not)?
NFA_EXPORT int nfa_visibility_compact_fused(int64_t n, int32_t aligned16_ptrs) { return vis_fusable(n, pick_plan(n, aligned16_ptrs != 0)) ? 1 : 0; }

NFA_EXPORT int nfa_visibility_compact_stamped(const int64_t *ray_indices, const float *t_starts, const float *t_ends,
                                              const float *dens, int32_t from_alpha, int64_t n, float early_stop_eps,
                                              float alpha_thre, int64_t *out_ray_indices, float *out_t_starts,
                                              float *out_t_ends, uint8_t *out_mask, int64_t *n_out, int64_t stamp, void *workspace,
                                              void *stream)
{
    return visibility_compact_impl(ray_indices, t_starts, t_ends, dens, from_alpha, n, early_stop_eps, alpha_thre, out_ray_indices, out_t_starts,
                                   out_t_ends, out_mask, n_out, stamp, workspace, nullptr, 0, stream);
}

NFA_EXPORT int nfa_visibility_compact_sync(const int64_t *ray_indices, const float *t_starts, const float *t_ends,
                                           const float *dens, int32_t from_alpha, int64_t n, float early_stop_eps,
                                           float alpha_thre, int64_t *out_ray_indices, float *out_t_starts,
                                           float *out_t_ends, uint8_t *out_mask, int64_t *n_out, int64_t stamp, void *workspace,
                                           void *sync, void *stream)
{
    return visibility_compact_impl(ray_indices, t_starts, t_ends, dens, from_alpha, n, early_stop_eps, alpha_thre, out_ray_indices, out_t_starts,
                                   out_t_ends, out_mask, n_out, stamp, workspace, sync, 0, stream);
}

NFA_EXPORT int nfa_visibility_compact_resume(const int64_t *ray_indices, const float *t_starts, const float *t_ends,
                                             const float *dens, int32_t from_alpha, int64_t n, float early_stop_eps,
                                             float alpha_thre, int64_t *out_ray_indices, float *out_t_starts,
                                             float *out_t_ends, uint8_t *out_mask, int64_t *n_out, int64_t stamp, void *workspace, void *stream)
{
    return visibility_compact_impl(ray_indices, t_starts, t_ends, dens, from_alpha, n, early_stop_eps, alpha_thre, out_ray_indices, out_t_starts,
                                   out_t_ends, out_mask, n_out, stamp, workspace, nullptr, 1, stream);
}

static int visibility_compact_impl(const int64_t *ray_indices, const float *t_starts, const float *t_ends, const float *dens, int32_t from_alpha,
                                   int64_t n, float early_stop_eps, float alpha_thre, int64_t *out_ray_indices, float *out_t_starts,
                                   float *out_t_ends, uint8_t *out_mask, int64_t *n_out, int64_t stamp, void *workspace, void *sync, int resume,
                                   void *stream)
{
    NFA_REQUIRE(n >= 0, "visibility_compact: n < 0");
    NFA_REQUIRE(n_out != nullptr, "visibility_compact: n_out is NULL");
    hipStream_t s = (hipStream_t)stream;
    if (n == 0) { (void)hipMemsetAsync(n_out, 0, sizeof(int64_t), s); return NFA_OK; }
    NFA_REQUIRE(ray_indices && t_starts && t_ends && dens && workspace, "visibility_compact: NULL pointer");
    if (out_ray_indices) NFA_REQUIRE(out_t_starts && out_t_ends, "visibility_compact: compacted outputs must be given together");
    // (the byte mask is only written when the caller asks for it; the compaction reads the bit planes at the head of the workspace)
    const TilePlan pl = pick_plan(n, aligned16({ray_indices, t_starts, t_ends, dens, out_mask}));
    const int64_t tile = pl.tile, T = ceil_div(n, tile);
    uint64_t *planes = (uint64_t *)workspace;
    int64_t *tile_cnts = (int64_t *)((uint8_t *)workspace + vis_front_bytes(n));
    int64_t *tile_offs = tile_cnts + T, *tile_rng = tile_offs + T;
    const int mode = T <= kVisFusedTiles ? 1 : (T <= kVisGroupedTiles ? 2 : 0);
    // The single-launch form (`fused_vis` = 1, opt-in): the static one-pass kernel — every workgroup resident at once, a state per
    // workgroup in the sync block, survivors staged in LDS and flushed behind the look-back.  Measured at the training size
    // (profiles/r06_small_n.md): 19.5 us against 9.2 + 6.9 for the two kernels below — publishing a count and polling the counts of
    // the workgroups before it costs two memory-side round trips (agent-scope atomics: the XCDs' L2s are not coherent), more than
    // the kernel boundary it replaces.  Not taken automatically.
    if (sync && resume == 0 && out_ray_indices && pl.e <= 2 && vis_fusable(n, pl)) {
        const int64_t ch = 64 * pl.e;
        // tiles of one chunk while that keeps the launch at two workgroups per CU or fewer, longer tiles beyond
        int64_t chunks = ceil_div(n, ch * kWavesPerBlock * 2 * kNumCU);
        chunks = chunks < 1 ? 1 : (chunks > kVisOnePassChunks ? kVisOnePassChunks : chunks);
        const int64_t otile = chunks * ch, OT = ceil_div(n, otile), groups = ceil_div(OT, kWavesPerBlock);
        const int cap = (int)(otile + ch);
        uint8_t *ov_mask = out_mask ? out_mask : (uint8_t *)workspace;
        const size_t lds = (size_t)kWavesPerBlock * cap * 16;
        if (groups <= 3 * kNumCU && groups <= kSyncMaxBlocks) {
            const dim3 g((unsigned)groups), b(kBlock);
            if (pl.e == 2) hipLaunchKernelGGL((visibility_onepass_kernel<2>), g, b, lds, s, ray_indices, t_starts, t_ends, dens, from_alpha, n, otile,
                                              early_stop_eps, alpha_thre, out_mask, ov_mask, (uint64_t *)sync, sync_spin_ticks(), OT, cap, n_out, stamp, out_ray_indices, out_t_starts, out_t_ends);
            else hipLaunchKernelGGL((visibility_onepass_kernel<1>), g, b, lds, s, ray_indices, t_starts, t_ends, dens, from_alpha, n, otile,
                                    early_stop_eps, alpha_thre, out_mask, ov_mask, (uint64_t *)sync, sync_spin_ticks(), OT, cap, n_out, stamp, out_ray_indices, out_t_starts, out_t_ends);
            return check_launch("visibility_onepass_kernel");
        }
    }
    NFA_LAUNCH_TILED(visibility_mask_kernel, pl, n, s, ray_indices, t_starts, t_ends,
                     dens, from_alpha, n, tile, pl.spec, early_stop_eps, alpha_thre, out_mask, planes, tile_cnts, tile_rng,
                     mode == 1 ? tile_offs : (int64_t *)nullptr);      // (mode 1: the otherwise unused second quarter of the workspace)
    if (int rc = check_launch("visibility_mask_kernel")) return rc;
    int64_t *group_sums = tile_offs;                       // (the second half of the workspace; T / 64 <= T entries)
    if (mode == 2) {
        const int64_t groups = ceil_div(T, 64);
        hipLaunchKernelGGL(visibility_group_scan_kernel, dim3((unsigned)ceil_div(groups, kWavesPerBlock)), dim3(kBlock), 0, s,
                           tile_cnts, T, group_sums);
        if (int rc = check_launch("visibility_group_scan_kernel")) return rc;
    } else if (mode == 0) {
        if (int rc = nfa_exclusive_sum_i64(tile_cnts, T, tile_offs, n_out, stream)) return rc;
        if (!out_ray_indices) return NFA_OK;
    }
    NFA_LAUNCH_TILED(visibility_compact_kernel, pl, n, s, ray_indices, t_starts, t_ends,
                     (const uint64_t *)planes, (const int64_t *)(mode == 0 ? tile_offs : tile_cnts), (const int64_t *)group_sums,
                     (const int64_t *)tile_rng, mode, n, tile, T, n_out, stamp, out_ray_indices, out_t_starts, out_t_ends);
    return check_launch("visibility_compact_kernel");
}

NFA_EXPORT int nfa_accumulate_along_rays(const int64_t *ray_indices, const float *weights, const float *values,
                                         int64_t n, int32_t D, int64_t n_rays, float *outputs, void *stream)
{
    NFA_REQUIRE(n >= 0 && n_rays >= 0, "accumulate_along_rays: negative size");
    NFA_REQUIRE(D >= 1, "accumulate_along_rays: D must be >= 1");
    NFA_REQUIRE(values != nullptr || D == 1, "accumulate_along_rays: D must be 1 when values is NULL");
    if (n == 0 || n_rays == 0) return NFA_OK;
    NFA_REQUIRE(ray_indices && weights && outputs, "accumulate_along_rays: NULL pointer");
    const TilePlan pl = pick_plan(n, aligned16({ray_indices, weights, values}));
    const int64_t tile = pl.tile;
    const dim3 grid(tile_blocks(n, tile)), block(kBlock);
    hipStream_t s = (hipStream_t)stream;
#define NFA_ACC(DC)                                                                                                              \
    do {                                                                                                                         \
        if (pl.e == 4) hipLaunchKernelGGL((accumulate_kernel<4, DC>), grid, block, 0, s, ray_indices, weights, values, n, tile, pl.spec, D, c0, n_rays, outputs); \
        else if (pl.e == 2) hipLaunchKernelGGL((accumulate_kernel<2, DC>), grid, block, 0, s, ray_indices, weights, values, n, tile, pl.spec, D, c0, n_rays, outputs); \
        else hipLaunchKernelGGL((accumulate_kernel<1, DC>), grid, block, 0, s, ray_indices, weights, values, n, tile, pl.spec, D, c0, n_rays, outputs); \
        c0 += DC;                                                                                                                \
    } while (0)
    int c0 = 0;
    while (c0 < D) {
        const int rem = D - c0;
        if (rem >= 4) NFA_ACC(4);
        else if (rem == 3) NFA_ACC(3);
        else if (rem == 2) NFA_ACC(2);
        else NFA_ACC(1);
    }
#undef NFA_ACC
    return check_launch("accumulate_kernel");
}

NFA_EXPORT int nfa_accumulate_along_rays_bwd(const int64_t *ray_indices, const float *weights, const float *values,
                                             const float *g_outputs, int64_t n, int32_t D, int64_t n_rays,
                                             float *g_weights, float *g_values, void *stream)
{
    NFA_REQUIRE(n >= 0 && D >= 1 && n_rays >= 0, "accumulate_along_rays_bwd: bad size");
    if (n == 0) return NFA_OK;
    NFA_REQUIRE(ray_indices && weights && g_outputs, "accumulate_along_rays_bwd: NULL pointer");
    NFA_REQUIRE(values != nullptr || (D == 1 && g_values == nullptr), "accumulate_along_rays_bwd: values is NULL");
    hipLaunchKernelGGL(accumulate_bwd_kernel, dim3(blocks_for(n)), dim3(kBlock), 0, (hipStream_t)stream,
                       ray_indices, weights, values, g_outputs, n, D, n_rays, g_weights, g_values);
    return check_launch("accumulate_bwd_kernel");
}

NFA_EXPORT int nfa_rendering_fwd(const int64_t *ray_indices, const float *t_starts, const float *t_ends,
                                 const float *sigmas, const float *rgbs, int64_t n, int64_t n_rays,
                                 const float *bkgd, int32_t expected_depths, float *weights, float *trans,
                                 float *alphas, float *colors, float *opacities, float *depths, void *stream)
{
    NFA_REQUIRE(n >= 0 && n_rays >= 0, "rendering_fwd: negative size");
    if (n_rays == 0) return NFA_OK;
    NFA_REQUIRE(colors && opacities && depths, "rendering_fwd: NULL per-ray output");
    hipStream_t s = (hipStream_t)stream;
    const bool fold = n > 0 && n <= kFillFoldMaxSamples && opt(OPT_FOLD_FILL, 1) != 0;
    if (!fold) {       // (n = 0: nothing but rays without a sample)
        hipLaunchKernelGGL(fill_rays_kernel, dim3(blocks_for(n_rays)), dim3(kBlock), 0, s, n_rays, bkgd, colors, opacities, depths);
        if (int rc = check_launch("fill_rays_kernel")) return rc;
    }
    if (n == 0) return NFA_OK;
    NFA_REQUIRE(ray_indices && t_starts && t_ends && sigmas && rgbs && weights && trans && alphas, "rendering_fwd: NULL pointer");
    const TilePlan pl = pick_plan(n, aligned16({ray_indices, t_starts, t_ends, sigmas, rgbs, weights, trans, alphas}));
    const unsigned main_blocks = tile_blocks(n, pl.tile);
    // the gap workgroups: four keys per thread, at most one workgroup per CU
    const unsigned gap_blocks = fold ? (unsigned)std::min<int64_t>(kNumCU, ceil_div(n + 1, 4 * kBlock)) : 0u;
    const dim3 g_(main_blocks + gap_blocks), b_(kBlock);
#define NFA_RENDER_FWD(EE) hipLaunchKernelGGL((rendering_fwd_kernel<EE>), g_, b_, 0, s, ray_indices, t_starts, t_ends, sigmas, rgbs, n, pl.tile, pl.spec, n_rays, \
                                              bkgd, expected_depths, (int)main_blocks, weights, trans, alphas, colors, opacities, depths)
    if (pl.e == 4) NFA_RENDER_FWD(4); else if (pl.e == 2) NFA_RENDER_FWD(2); else NFA_RENDER_FWD(1);
#undef NFA_RENDER_FWD
    return check_launch("rendering_fwd_kernel");
}

NFA_EXPORT int nfa_rendering_bwd(const int64_t *ray_indices, const float *t_starts, const float *t_ends,
                                 const float *sigmas, const float *rgbs, const float *weights,
                                 const float *trans, const float *alphas, const float *opacities,
                                 const float *depths, int64_t n, int64_t n_rays, const float *bkgd,
                                 int32_t expected_depths, const float *g_colors, const float *g_opacities,
                                 const float *g_depths, const float *g_weights, const float *g_trans,
                                 const float *g_alphas, float *g_sigmas, float *g_rgbs, void *stream)
{
    (void)sigmas;
    NFA_REQUIRE(n >= 0 && n_rays >= 0, "rendering_bwd: negative size");
    if (n == 0) return NFA_OK;
    (void)weights;      // (not read since round 5: the kernel forms w = trans * alphas, exactly what the forward pass stored; nullable)
    NFA_REQUIRE(ray_indices && t_starts && t_ends && rgbs && trans && alphas, "rendering_bwd: NULL pointer");
    NFA_REQUIRE(!(g_depths && expected_depths) || (opacities && depths), "rendering_bwd: opacities/depths needed for g_depths");
    const TilePlan pl = pick_plan(n, aligned16({ray_indices, t_starts, t_ends, rgbs, trans, alphas, g_weights, g_trans, g_alphas, g_sigmas, g_rgbs}));
    const dim3 g_(tile_blocks(n, pl.tile)), b_(kBlock);
#define NFA_RENDER_BWD(EE, XX) hipLaunchKernelGGL((rendering_bwd_kernel<EE, XX>), g_, b_, 0, (hipStream_t)stream, ray_indices, t_starts, t_ends, rgbs, trans, \
                                                  alphas, opacities, depths, n, pl.tile, pl.spec, n_rays, bkgd, expected_depths, g_colors, g_opacities, g_depths, \
                                                  g_weights, g_trans, g_alphas, g_sigmas, g_rgbs)
    if (g_weights || g_trans || g_alphas) { if (pl.e == 4) NFA_RENDER_BWD(4, true); else if (pl.e == 2) NFA_RENDER_BWD(2, true); else NFA_RENDER_BWD(1, true); }
    else { if (pl.e == 4) NFA_RENDER_BWD(4, false); else if (pl.e == 2) NFA_RENDER_BWD(2, false); else NFA_RENDER_BWD(1, false); }
#undef NFA_RENDER_BWD
    return check_launch("rendering_bwd_kernel");
}

NFA_EXPORT int nfa_pack_info(const int64_t *ray_indices, int64_t n, int64_t n_rays, int64_t *packed_info, void *stream)
{
    NFA_REQUIRE(n >= 0 && n_rays >= 0, "pack_info: negative size");
    if (n_rays == 0) return NFA_OK;
    NFA_REQUIRE(packed_info && (ray_indices || n == 0), "pack_info: NULL pointer");
    hipStream_t s = (hipStream_t)stream;
    if (n_rays == 0) return NFA_OK;
    if (hipMemsetAsync(packed_info, 0, 2 * sizeof(int64_t), s) != hipSuccess) return fail(NFA_ERR_LAUNCH, "pack_info: memset failed");
    if (n > 1) hipLaunchKernelGGL(pack_check_kernel, dim3(blocks_for(n)), dim3(kBlock), 0, s, ray_indices, n, packed_info);
    hipLaunchKernelGGL(pack_info_kernel, dim3(blocks_for(n_rays)), dim3(kBlock), 0, s, ray_indices, n, n_rays, packed_info);
    if (n > 1) {
        hipLaunchKernelGGL(pack_hist_zero_kernel, dim3(blocks_for(n_rays)), dim3(kBlock), 0, s, n_rays, packed_info);
        hipLaunchKernelGGL(pack_hist_kernel, dim3(blocks_for(n)), dim3(kBlock), 0, s, ray_indices, n, n_rays, packed_info);
        hipLaunchKernelGGL(pack_hist_scan_kernel, dim3(1), dim3(1024), 0, s, n_rays, packed_info);
    }
    return check_launch("pack_info_kernel");
}

NFA_EXPORT int nfa_unpack_info(const int64_t *chunk_starts, const int64_t *chunk_cnts, int64_t n_rays,
                               int64_t *ray_indices, int64_t n, void *stream)
{
    NFA_REQUIRE(n >= 0 && n_rays >= 0, "unpack_info: negative size");
    if (n == 0) return NFA_OK;
    NFA_REQUIRE(ray_indices, "unpack_info: NULL pointer");
    hipStream_t s = (hipStream_t)stream;
    (void)hipMemsetAsync(ray_indices, 0xFF, sizeof(int64_t) * (size_t)n, s);   // -1 everywhere
    if (n_rays == 0) return NFA_OK;
    NFA_REQUIRE(chunk_starts && chunk_cnts, "unpack_info: NULL pointer");
    hipLaunchKernelGGL(unpack_info_kernel, dim3(blocks_for(n_rays * 16)), dim3(kBlock), 0, s, chunk_starts, chunk_cnts, n_rays, n, ray_indices);
    return check_launch("unpack_info_kernel");
}

NFA_EXPORT int nfa_sample_positions(const float *rays_o, const float *rays_d, int64_t n_rays,
                                    const int64_t *ray_indices, const float *t_starts, const float *t_ends, int64_t n,
                                    float *positions, float *dirs, void *stream)
{
    NFA_REQUIRE(n >= 0 && n_rays >= 0, "sample_positions: negative size");
    if (n == 0) return NFA_OK;
    NFA_REQUIRE(rays_o && rays_d && ray_indices && t_starts && t_ends && positions, "sample_positions: NULL pointer");
    hipLaunchKernelGGL(sample_positions_kernel, dim3(blocks_for(3 * n)), dim3(kBlock), 0, (hipStream_t)stream,
                       rays_o, rays_d, n_rays, ray_indices, t_starts, t_ends, n, positions, dirs);
    return check_launch("sample_positions_kernel");
}
