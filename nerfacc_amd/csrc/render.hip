// render.hip — fused volumetric-rendering kernels for gfx950.
//
// In the reference this part of the hot path is ~45 small ATen kernels per training step
// (nerfacc/volrend.py: sigma*dt, exp, exclusive scan, mul, three index_add_ with float
// atomics, clamp/div, background blend — and their autograd twins) plus three host-syncing
// boolean-mask compactions in OccGridEstimator.sampling (occ_grid.py:194-220).  Here each
// logical op is ONE streaming kernel over the flat sample array, built on the
// segment-snapped wave tiles of common.hpp:
//   nfa_render_weight_from_density_{fwd,bwd}   volrend.py:219-278, 326-376
//   nfa_visibility_compact                     volrend.py:435-494 + occ_grid.py:216-220
//   nfa_accumulate_along_rays{,_bwd}           volrend.py:497-587
//   nfa_rendering_{fwd,bwd}                    volrend.py:104-164 (after rgb_sigma_fn)
//   nfa_pack_info / nfa_unpack_info            pack.py:10-49
// All are HBM-streaming (bytes per sample in DESIGN.md), need no LDS, and are deterministic
// for ray-sorted input (per-ray sums are formed inside one wave, in a fixed order).
#include "common.hpp"

namespace nfa {
namespace {

constexpr float kEpsF32 = 1.1920928955078125e-07f;   // torch.finfo(torch.float32).eps

// Forward-direction segment flags of one 64-element chunk.
struct SegFwd {
    bool head, open;
    int dist;
    unsigned long long heads;
};
__device__ __forceinline__ SegFwd seg_fwd(int64_t key, bool active, bool first_chunk, int64_t edge_key, int lane) {
    SegFwd s;
    const int64_t pk = lane_prev_i64(key);
    s.head = !active || key != pk;
    if (lane == 0) s.head = first_chunk || key != edge_key;
    s.heads = __ballot(s.head);
    s.dist = dist_to_head(s.heads, lane, s.open);
    return s;
}
// tail flag for a forward walk: the sample is the last of its ray.  Every lane loads the key after
// its own together with its other inputs (same cache lines, one memory round trip) — peeking
// keys[i + 1] from lane 63 only, after the scan flags, made each chunk wait for memory twice.
__device__ __forceinline__ int64_t load_next_key(const int64_t *__restrict__ keys, int64_t i, int64_t end, int64_t key) {
    return (i + 1 < end) ? keys[i + 1] : ~key;
}
// Reverse-direction flags.
struct SegBwd {
    bool tail, open;
    int dist;
};
__device__ __forceinline__ SegBwd seg_bwd(int64_t key, bool active, int64_t i, int64_t end, int64_t edge_key, int lane) {
    SegBwd s;
    const int64_t nk = lane_next_i64(key);
    s.tail = !active || (i + 1 >= end) || key != nk;
    if (lane == 63 && active && i + 1 < end) s.tail = key != edge_key;
    const unsigned long long tails = __ballot(s.tail);
    s.dist = dist_to_tail(tails, lane, s.open);
    return s;
}

__device__ __forceinline__ float seg_incl_fwd(float v, const SegFwd &s, float &carry) {
    float incl = wave_seg_scan_fwd<OpSum>(v, s.dist);
    if (s.open) incl += carry;
    carry = readlane_f<63>(incl);
    return incl;
}
__device__ __forceinline__ float seg_excl_fwd(float v, const SegFwd &s, float &carry, int lane) {
    float incl = wave_seg_scan_fwd<OpSum>(v, s.dist);
    if (s.open) incl += carry;
    float excl = lane_prev_f(incl, 0.0f);
    if (s.head) excl = 0.0f;
    else if (lane == 0) excl = carry;
    carry = readlane_f<63>(incl);
    return excl;
}
__device__ __forceinline__ float seg_excl_bwd(float v, const SegBwd &s, float &carry, int lane) {
    float incl = wave_seg_scan_bwd<OpSum>(v, s.dist);
    if (s.open) incl += carry;
    float excl = lane_next_f(incl, 0.0f);
    if (s.tail) excl = 0.0f;
    else if (lane == 63) excl = carry;
    carry = readlane_f<0>(incl);
    return excl;
}

#define NFA_WAVE_TILE_PROLOGUE(keys, n, tile)                                        \
    const int64_t w_ = (int64_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);    \
    const TileRange tr = snapped_tile(keys, n, w_, tile);                             \
    if (tr.begin >= tr.end) return;                                                   \
    const int lane = lane_id();                                                       \
    const int64_t n_chunks = (tr.end - tr.begin + 63) >> 6;

// ----------------------------------------------------------------------------------------
// weights / transmittance / alpha from density
// ----------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void weight_fwd_kernel(
    const int64_t *__restrict__ keys, const float *__restrict__ ts, const float *__restrict__ te,
    const float *__restrict__ sigmas, const float *__restrict__ prefix, int64_t n, int64_t tile,
    float *__restrict__ weights, float *__restrict__ trans, float *__restrict__ alphas)
{
    NFA_WAVE_TILE_PROLOGUE(keys, n, tile)
    float carry = 0.0f;
    int64_t edge_key = 0;
    for (int64_t c = 0; c < n_chunks; ++c) {
        const int64_t i = tr.begin + c * 64 + lane;
        const bool active = i < tr.end;
        int64_t key = 0;
        float sd = 0.0f;
        if (active) { key = keys[i]; sd = sigmas[i] * (te[i] - ts[i]); }
        const SegFwd s = seg_fwd(key, active, c == 0, edge_key, lane);
        const float acc = seg_excl_fwd(sd, s, carry, lane);
        edge_key = readlane_i64<63>(key);
        if (active) {
            const float a = 1.0f - expf(-sd);
            float T = expf(-acc);
            if (prefix) T = T * prefix[i];
            if (alphas) st_stream(alphas + i, a);
            if (trans) st_stream(trans + i, T);
            if (weights) st_stream(weights + i, T * a);
        }
    }
}

// d/dsigma of sum(g_w w + g_T T + g_a a):
//   g_sd_i = (g_w_i T_i + g_a_i)(1 - a_i) - sum_{j>i, same ray} (g_w_j w_j + g_T_j T_j)
__global__ __launch_bounds__(kBlock) void weight_bwd_kernel(
    const int64_t *__restrict__ keys, const float *__restrict__ ts, const float *__restrict__ te,
    const float *__restrict__ trans, const float *__restrict__ alphas,
    const float *__restrict__ g_w, const float *__restrict__ g_T, const float *__restrict__ g_a,
    int64_t n, int64_t tile, float *__restrict__ g_sigmas)
{
    NFA_WAVE_TILE_PROLOGUE(keys, n, tile)
    float carry = 0.0f;
    int64_t edge_key = 0;
    for (int64_t c = n_chunks - 1; c >= 0; --c) {
        const int64_t i = tr.begin + c * 64 + lane;
        const bool active = i < tr.end;
        int64_t key = 0;
        float T = 0.f, a = 0.f, gw = 0.f, gT = 0.f, ga = 0.f;
        if (active) {
            key = keys[i];
            T = trans[i];
            a = alphas[i];
            if (g_w) gw = g_w[i];
            if (g_T) gT = g_T[i];
            if (g_a) ga = g_a[i];
        }
        const SegBwd s = seg_bwd(key, active, i, tr.end, edge_key, lane);
        const float q = gw * (T * a) + gT * T;
        const float suffix = seg_excl_bwd(q, s, carry, lane);
        edge_key = readlane_i64<0>(key);
        if (active) g_sigmas[i] = ((gw * T + ga) * (1.0f - a) - suffix) * (te[i] - ts[i]);
    }
}

// ----------------------------------------------------------------------------------------
// visibility filter + compaction (three kernels, no host round trip in between)
// ----------------------------------------------------------------------------------------
// 1) keep mask and per-wave-tile kept counts
__global__ __launch_bounds__(kBlock) void visibility_mask_kernel(
    const int64_t *__restrict__ keys, const float *__restrict__ ts, const float *__restrict__ te,
    const float *__restrict__ dens, int from_alpha, int64_t n, int64_t tile, float eps, float alpha_thre,
    uint8_t *__restrict__ mask, int64_t *__restrict__ tile_cnts)
{
    const int64_t w = (int64_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    if (w * tile >= n) return;
    const TileRange tr = snapped_tile(keys, n, w, tile);
    const int lane = lane_id();
    int64_t kept = 0;
    if (tr.begin < tr.end) {
        const int64_t n_chunks = (tr.end - tr.begin + 63) >> 6;
        float carry = from_alpha ? 1.0f : 0.0f;
        int64_t edge_key = 0;
        for (int64_t c = 0; c < n_chunks; ++c) {
            const int64_t i = tr.begin + c * 64 + lane;
            const bool active = i < tr.end;
            int64_t key = 0;
            float x = 0.0f, a = 0.0f;
            if (active) {
                key = keys[i];
                if (from_alpha) { a = dens[i]; x = 1.0f - a; }
                else { x = dens[i] * (te[i] - ts[i]); a = 1.0f - expf(-x); }
            }
            const SegFwd s = seg_fwd(key, active, c == 0, edge_key, lane);
            float T;
            if (from_alpha) {   // exclusive product of (1 - alpha), volrend.py:207-209
                float incl = wave_seg_scan_fwd<OpProd>(active ? x : 1.0f, s.dist);
                if (s.open) incl = carry * incl;
                T = lane_prev_f(incl, 0.0f);
                if (s.head) T = 1.0f;
                else if (lane == 0) T = carry;
                carry = readlane_f<63>(incl);
            } else {
                T = expf(-seg_excl_fwd(x, s, carry, lane));
            }
            edge_key = readlane_i64<63>(key);
            bool keep = active && (T >= eps);
            if (alpha_thre > 0.0f) keep = keep && (a >= alpha_thre);
            if (active) mask[i] = keep ? 1 : 0;
            kept += __popcll(__ballot(keep));
        }
    }
    if (lane == 0) tile_cnts[w] = kept;
}

// 2) (only beyond kVisFusedTiles tiles) one wave per group of 64 tiles: exclusive prefix of the
// group's tile counts in place + the group total — thousands of independent waves instead of one
// workgroup crawling through every count
__global__ __launch_bounds__(kBlock) void visibility_group_scan_kernel(int64_t *__restrict__ tile_cnts, int64_t n_tiles,
                                                                       int64_t *__restrict__ group_sums)
{
    const int64_t g = (int64_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
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

// 3) stream compaction with ballot ranks.  Where a wave's survivors go:
//   mode 0  tile_offs[w] is the global exclusive prefix (single-workgroup scan, huge inputs only)
//   mode 1  fused form for up to kVisFusedTiles tiles (every training-size call): tile_offs holds the
//           tile COUNTS and each wave sums the ones before its own (<= 64 L2 loads per lane)
//   mode 2  tile_offs holds the prefix inside the wave's group of 64 tiles (kernel 2) and group_sums the
//           group totals: the wave adds the totals of the groups before its own
// The last wave stores the total in modes 1 and 2.
constexpr int64_t kVisFusedTiles = 4096, kVisGroupedTiles = 64 * 4096;
__global__ __launch_bounds__(kBlock) void visibility_compact_kernel(
    const int64_t *__restrict__ keys, const float *__restrict__ ts, const float *__restrict__ te,
    const uint8_t *__restrict__ mask, const int64_t *__restrict__ tile_offs, const int64_t *__restrict__ group_sums,
    int mode, int64_t n_tiles, int64_t *__restrict__ n_out, int64_t n, int64_t tile,
    int64_t *__restrict__ o_keys, float *__restrict__ o_ts, float *__restrict__ o_te)
{
    const int64_t w_ = (int64_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    if (w_ >= n_tiles) return;
    const int lane = lane_id();
    const TileRange tr = snapped_tile(keys, n, w_, tile);
    int64_t dst;
    if (mode == 1) {
        int64_t p = 0;
        for (int64_t j = lane; j < w_; j += 64) p += tile_offs[j];
        dst = wave_sum_i64(p);
        if (w_ == n_tiles - 1 && lane == 0) *n_out = dst + tile_offs[w_];
    } else if (mode == 2) {
        const int64_t g = w_ >> 6;
        int64_t p = 0;
        for (int64_t j = lane; j < g; j += 64) p += group_sums[j];
        dst = wave_sum_i64(p) + tile_offs[w_];
        if (w_ == n_tiles - 1 && lane == 0) *n_out = dst - tile_offs[w_] + group_sums[g];
    } else {
        dst = tile_offs[w_];
    }
    if (!o_keys) return;
    if (tr.begin >= tr.end) return;
    const int64_t n_chunks = (tr.end - tr.begin + 63) >> 6;
    for (int64_t c = 0; c < n_chunks; ++c) {
        const int64_t i = tr.begin + c * 64 + lane;
        const bool keep = (i < tr.end) && mask[i];
        const unsigned long long b = __ballot(keep);
        if (keep) {
            const int64_t k = dst + __popcll(b & lanes_lt(lane));
            st_stream(o_keys + k, keys[i]);
            st_stream(o_ts + k, ts[i]);
            st_stream(o_te + k, te[i]);
        }
        dst += __popcll(b);
    }
}

// ----------------------------------------------------------------------------------------
// accumulate_along_rays: out[r, c0 + c] += sum_i w_i v[i, c0 + c]   (DC channels per launch)
// ----------------------------------------------------------------------------------------
template <int DC>
__global__ __launch_bounds__(kBlock) void accumulate_kernel(
    const int64_t *__restrict__ keys, const float *__restrict__ weights, const float *__restrict__ values,
    int64_t n, int64_t tile, int D, int c0, int64_t n_rays, float *__restrict__ out)
{
    NFA_WAVE_TILE_PROLOGUE(keys, n, tile)
    float carry[DC];
#pragma unroll
    for (int c = 0; c < DC; ++c) carry[c] = 0.0f;
    int64_t edge_key = 0;
    // U chunks per trip, all their loads issued before the first scan: this kernel moves only
    // 12 + 4 DC bytes per sample, and with one chunk in flight per wave the bytes in flight per CU
    // (not the HBM) bounded it (Little's law)
    constexpr int U = 3;
    for (int64_t ch0 = 0; ch0 < n_chunks; ch0 += U) {
        int64_t key[U], nkey[U];
        float w[U], v[U][DC];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t i = tr.begin + (ch0 + u) * 64 + lane;
            key[u] = 0; nkey[u] = -1; w[u] = 0.0f;
#pragma unroll
            for (int c = 0; c < DC; ++c) v[u][c] = 1.0f;
            if (i < tr.end) {
                key[u] = keys[i];
                nkey[u] = load_next_key(keys, i, tr.end, key[u]);
                w[u] = weights[i];
                if (values) {
#pragma unroll
                    for (int c = 0; c < DC; ++c) v[u][c] = values[i * D + c0 + c];
                }
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t ch = ch0 + u;
            if (ch >= n_chunks) break;
            const int64_t i = tr.begin + ch * 64 + lane;
            const bool active = i < tr.end;
            const SegFwd s = seg_fwd(key[u], active, ch == 0, edge_key, lane);
            const bool tail = active && nkey[u] != key[u];
            edge_key = readlane_i64<63>(key[u]);
#pragma unroll
            for (int c = 0; c < DC; ++c) {
                const float tot = seg_incl_fwd(w[u] * v[u][c], s, carry[c]);
                if (active && tail && key[u] >= 0 && key[u] < n_rays) unsafeAtomicAdd(out + key[u] * D + c0 + c, tot);
            }
        }
    }
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

__global__ __launch_bounds__(kBlock) void rendering_fwd_kernel(
    const int64_t *__restrict__ keys, const float *__restrict__ ts, const float *__restrict__ te,
    const float *__restrict__ sigmas, const float *__restrict__ rgbs, int64_t n, int64_t tile, int64_t n_rays,
    const float *__restrict__ bkgd, int expected_depths,
    float *__restrict__ weights, float *__restrict__ trans, float *__restrict__ alphas,
    float *__restrict__ colors, float *__restrict__ opac, float *__restrict__ depth)
{
    NFA_WAVE_TILE_PROLOGUE(keys, n, tile)
    float c_sd = 0.f, c_r = 0.f, c_g = 0.f, c_b = 0.f, c_w = 0.f, c_m = 0.f;
    int64_t edge_key = 0;
    float bk0 = 0.f, bk1 = 0.f, bk2 = 0.f;
    if (bkgd) { bk0 = bkgd[0]; bk1 = bkgd[1]; bk2 = bkgd[2]; }
    for (int64_t c = 0; c < n_chunks; ++c) {
        const int64_t i = tr.begin + c * 64 + lane;
        const bool active = i < tr.end;
        int64_t key = 0, nkey = -1;
        float t0 = 0.f, t1 = 0.f, sd = 0.f, r = 0.f, g = 0.f, b = 0.f;
        if (active) {
            key = keys[i];
            nkey = load_next_key(keys, i, tr.end, key);
            t0 = ts[i]; t1 = te[i];
            sd = sigmas[i] * (t1 - t0);
            r = rgbs[3 * i]; g = rgbs[3 * i + 1]; b = rgbs[3 * i + 2];
        }
        const SegFwd s = seg_fwd(key, active, c == 0, edge_key, lane);
        const bool tail = active && nkey != key;
        edge_key = readlane_i64<63>(key);
        const float acc = seg_excl_fwd(sd, s, c_sd, lane);
        float w = 0.0f;
        if (active) {
            const float a = 1.0f - expf(-sd);
            const float T = expf(-acc);
            w = T * a;
            st_stream(alphas + i, a);
            st_stream(trans + i, T);
            st_stream(weights + i, w);
        }
        const float sr = seg_incl_fwd(w * r, s, c_r);
        const float sg = seg_incl_fwd(w * g, s, c_g);
        const float sb = seg_incl_fwd(w * b, s, c_b);
        const float sw = seg_incl_fwd(w, s, c_w);
        const float sm = seg_incl_fwd(w * ((t0 + t1) / 2.0f), s, c_m);
        if (active && tail && key >= 0 && key < n_rays) {
            const float rem = 1.0f - sw;
            colors[3 * key] = bkgd ? sr + bk0 * rem : sr;
            colors[3 * key + 1] = bkgd ? sg + bk1 * rem : sg;
            colors[3 * key + 2] = bkgd ? sb + bk2 * rem : sb;
            opac[key] = sw;
            depth[key] = expected_depths ? sm / fmaxf(sw, kEpsF32) : sm;
        }
    }
}

__global__ __launch_bounds__(kBlock) void rendering_bwd_kernel(
    const int64_t *__restrict__ keys, const float *__restrict__ ts, const float *__restrict__ te,
    const float *__restrict__ rgbs, const float *__restrict__ weights, const float *__restrict__ trans,
    const float *__restrict__ alphas, const float *__restrict__ opac, const float *__restrict__ depth,
    int64_t n, int64_t tile, int64_t n_rays, const float *__restrict__ bkgd, int expected_depths,
    const float *__restrict__ g_colors, const float *__restrict__ g_opac, const float *__restrict__ g_depth,
    const float *__restrict__ g_w_ext, const float *__restrict__ g_T_ext, const float *__restrict__ g_a_ext,
    float *__restrict__ g_sigmas, float *__restrict__ g_rgbs)
{
    NFA_WAVE_TILE_PROLOGUE(keys, n, tile)
    float carry = 0.0f;
    int64_t edge_key = 0;
    float bk0 = 0.f, bk1 = 0.f, bk2 = 0.f;
    if (bkgd) { bk0 = bkgd[0]; bk1 = bkgd[1]; bk2 = bkgd[2]; }
    for (int64_t c = n_chunks - 1; c >= 0; --c) {
        const int64_t i = tr.begin + c * 64 + lane;
        const bool active = i < tr.end;
        int64_t key = 0;
        float w = 0.f, T = 0.f, a = 0.f, gw = 0.f, gT = 0.f, ga = 0.f, dt = 0.f;
        if (active) {
            key = keys[i];
            w = weights[i]; T = trans[i]; a = alphas[i];
            const float t0 = ts[i], t1 = te[i];
            dt = t1 - t0;
            if (g_w_ext) gw = g_w_ext[i];
            if (g_T_ext) gT = g_T_ext[i];
            if (g_a_ext) ga = g_a_ext[i];
            if (key >= 0 && key < n_rays) {
                float gc0 = 0.f, gc1 = 0.f, gc2 = 0.f;
                if (g_colors) { gc0 = g_colors[3 * key]; gc1 = g_colors[3 * key + 1]; gc2 = g_colors[3 * key + 2]; }
                float go = g_opac ? g_opac[key] : 0.0f;
                float gacc = 0.0f;                      // dL/d(sum w m)
                if (g_depth) {
                    const float gd = g_depth[key];
                    if (expected_depths) {
                        const float O = opac[key];
                        gacc = gd / fmaxf(O, kEpsF32);
                        if (O > kEpsF32) go -= gd * depth[key] / O;
                    } else gacc = gd;
                }
                if (bkgd) go -= gc0 * bk0 + gc1 * bk1 + gc2 * bk2;
                gw += gc0 * rgbs[3 * i] + gc1 * rgbs[3 * i + 1] + gc2 * rgbs[3 * i + 2] + go + gacc * ((t0 + t1) / 2.0f);
                if (g_rgbs) { g_rgbs[3 * i] = w * gc0; g_rgbs[3 * i + 1] = w * gc1; g_rgbs[3 * i + 2] = w * gc2; }
            }
        }
        const SegBwd s = seg_bwd(key, active, i, tr.end, edge_key, lane);
        const float suffix = seg_excl_bwd(gw * w + gT * T, s, carry, lane);
        edge_key = readlane_i64<0>(key);
        if (active && g_sigmas) g_sigmas[i] = ((gw * T + ga) * (1.0f - a) - suffix) * dt;
    }
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

inline unsigned tile_blocks(int64_t n, int64_t tile) { return (unsigned)ceil_div(ceil_div(n, tile), kWavesPerBlock); }

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
    const int64_t tile = pick_tile(n);
    hipLaunchKernelGGL(weight_fwd_kernel, dim3(tile_blocks(n, tile)), dim3(kBlock), 0, (hipStream_t)stream,
                       ray_indices, t_starts, t_ends, sigmas, prefix_trans, n, tile, weights, trans, alphas);
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
    const int64_t tile = pick_tile(n);
    hipLaunchKernelGGL(weight_bwd_kernel, dim3(tile_blocks(n, tile)), dim3(kBlock), 0, (hipStream_t)stream,
                       ray_indices, t_starts, t_ends, trans, alphas, g_weights, g_trans, g_alphas, n, tile, g_sigmas);
    return check_launch("weight_bwd_kernel");
}

// workspace layout: [ mask: n bytes, padded to 16 ][ tile_cnts: T int64 ][ tile_offs: T int64 ]
static inline int64_t vis_tiles(int64_t n) { return ceil_div(n > 0 ? n : 1, pick_tile(n)); }
NFA_EXPORT int64_t nfa_visibility_workspace_bytes(int64_t n) {
    return ceil_div(n > 0 ? n : 1, 16) * 16 + 2 * (int64_t)sizeof(int64_t) * vis_tiles(n);
}

NFA_EXPORT int nfa_visibility_compact(const int64_t *ray_indices, const float *t_starts, const float *t_ends,
                                      const float *dens, int32_t from_alpha, int64_t n, float early_stop_eps,
                                      float alpha_thre, int64_t *out_ray_indices, float *out_t_starts,
                                      float *out_t_ends, uint8_t *out_mask, int64_t *n_out, void *workspace,
                                      void *stream)
{
    NFA_REQUIRE(n >= 0, "visibility_compact: n < 0");
    NFA_REQUIRE(n_out != nullptr, "visibility_compact: n_out is NULL");
    hipStream_t s = (hipStream_t)stream;
    if (n == 0) { (void)hipMemsetAsync(n_out, 0, sizeof(int64_t), s); return NFA_OK; }
    NFA_REQUIRE(ray_indices && t_starts && t_ends && dens && workspace, "visibility_compact: NULL pointer");
    const int64_t tile = pick_tile(n), T = vis_tiles(n);
    uint8_t *mask = out_mask ? out_mask : (uint8_t *)workspace;
    int64_t *tile_cnts = (int64_t *)((uint8_t *)workspace + ceil_div(n, 16) * 16);
    int64_t *tile_offs = tile_cnts + T;
    hipLaunchKernelGGL(visibility_mask_kernel, dim3(tile_blocks(n, tile)), dim3(kBlock), 0, s, ray_indices, t_starts, t_ends,
                       dens, from_alpha, n, tile, early_stop_eps, alpha_thre, mask, tile_cnts);
    if (int rc = check_launch("visibility_mask_kernel")) return rc;
    if (out_ray_indices) NFA_REQUIRE(out_t_starts && out_t_ends, "visibility_compact: compacted outputs must be given together");
    const int mode = T <= kVisFusedTiles ? 1 : (T <= kVisGroupedTiles ? 2 : 0);
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
    hipLaunchKernelGGL(visibility_compact_kernel, dim3(tile_blocks(n, tile)), dim3(kBlock), 0, s, ray_indices, t_starts, t_ends,
                       mask, mode == 0 ? tile_offs : tile_cnts, group_sums, mode, T, n_out, n, tile,
                       out_ray_indices, out_t_starts, out_t_ends);
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
    const int64_t tile = pick_tile(n);
    const dim3 grid(tile_blocks(n, tile)), block(kBlock);
    hipStream_t s = (hipStream_t)stream;
    int c0 = 0;
    while (c0 < D) {
        const int rem = D - c0;
        if (rem >= 4) { hipLaunchKernelGGL(accumulate_kernel<4>, grid, block, 0, s, ray_indices, weights, values, n, tile, D, c0, n_rays, outputs); c0 += 4; }
        else if (rem == 3) { hipLaunchKernelGGL(accumulate_kernel<3>, grid, block, 0, s, ray_indices, weights, values, n, tile, D, c0, n_rays, outputs); c0 += 3; }
        else if (rem == 2) { hipLaunchKernelGGL(accumulate_kernel<2>, grid, block, 0, s, ray_indices, weights, values, n, tile, D, c0, n_rays, outputs); c0 += 2; }
        else { hipLaunchKernelGGL(accumulate_kernel<1>, grid, block, 0, s, ray_indices, weights, values, n, tile, D, c0, n_rays, outputs); c0 += 1; }
    }
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
    hipLaunchKernelGGL(fill_rays_kernel, dim3(blocks_for(n_rays)), dim3(kBlock), 0, s, n_rays, bkgd, colors, opacities, depths);
    if (int rc = check_launch("fill_rays_kernel")) return rc;
    if (n == 0) return NFA_OK;
    NFA_REQUIRE(ray_indices && t_starts && t_ends && sigmas && rgbs && weights && trans && alphas, "rendering_fwd: NULL pointer");
    const int64_t tile = pick_tile(n);
    hipLaunchKernelGGL(rendering_fwd_kernel, dim3(tile_blocks(n, tile)), dim3(kBlock), 0, s, ray_indices, t_starts, t_ends,
                       sigmas, rgbs, n, tile, n_rays, bkgd, expected_depths, weights, trans, alphas, colors, opacities, depths);
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
    NFA_REQUIRE(ray_indices && t_starts && t_ends && rgbs && weights && trans && alphas, "rendering_bwd: NULL pointer");
    NFA_REQUIRE(!(g_depths && expected_depths) || (opacities && depths), "rendering_bwd: opacities/depths needed for g_depths");
    const int64_t tile = pick_tile(n);
    hipLaunchKernelGGL(rendering_bwd_kernel, dim3(tile_blocks(n, tile)), dim3(kBlock), 0, (hipStream_t)stream,
                       ray_indices, t_starts, t_ends, rgbs, weights, trans, alphas, opacities, depths, n, tile, n_rays, bkgd,
                       expected_depths, g_colors, g_opacities, g_depths, g_weights, g_trans, g_alphas, g_sigmas, g_rgbs);
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
