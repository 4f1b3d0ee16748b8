// pdf.hip — inverse-cdf importance sampling and per-ray searchsorted for gfx950.
//
// Replaces nerfacc/cuda/csrc/pdf.cu (importance_sampling_kernel :98-167,
// compute_intervels_kernel :169-241, searchsorted_kernel :245-286) behind
// include/nerfacc_hip.h.  One lane per output value; each lane binary-searches its own ray's
// (<= a few hundred) edges, which sit in L1/L2 after the first touch.  The two reference
// kernels of importance_sampling are fused into one launch: a lane computes its sample and
// re-derives its left neighbour's (same cached edges) to form the interval edge between them.
#include "common.hpp"

namespace nfa {
namespace {

__device__ __forceinline__ int64_t upper_bound_f(const float *__restrict__ v, int64_t lo, int64_t hi, float x) {
    while (lo < hi) {
        const int64_t mid = lo + ((hi - lo) >> 1);
        if (!(v[mid] > x)) lo = mid + 1; else hi = mid;
    }
    return lo;
}

__device__ __forceinline__ int64_t clamp64(int64_t v, int64_t lo, int64_t hi) { return v < lo ? lo : (v > hi ? hi : v); }

__device__ __forceinline__ void ray_span(const nfa_ray_segments &s, int64_t ray, int64_t &base, int64_t &last) {
    if (s.chunk_starts) { base = s.chunk_starts[ray]; last = base + s.chunk_cnts[ray] - 1; }
    else { base = ray * s.n_edges_per_ray; last = base + s.n_edges_per_ray - 1; }
}

// sample `sid` of `n_out` on one ray: pdf.cu:126-165
__device__ __forceinline__ float draw_sample(const float *__restrict__ vals, const float *__restrict__ cdfs,
                                             int64_t base, int64_t last, int64_t sid, int64_t n_out, float bias) {
    const float u0 = cdfs[base], u1 = cdfs[last];
    const float du = (u1 - u0) / (float)n_out;
    const float u = u0 + ((float)sid + bias) * du;
    const int64_t p = upper_bound_f(cdfs, base, last, u);
    const int64_t p0 = clamp64(p - 1, base, last), p1 = clamp64(p, base, last);
    const float ul = cdfs[p0], uh = cdfs[p1], tl = vals[p0], th = vals[p1];
    if (uh - ul < 1e-10f) return (tl + th) * 0.5f;
    return (u - ul) * ((th - tl) / (uh - ul)) + tl;
}

__global__ __launch_bounds__(kBlock) void importance_sampling_kernel(
    nfa_ray_segments seg, const float *__restrict__ cdfs, int64_t n_out, const float *__restrict__ jitter,
    float *__restrict__ out_edges, float *__restrict__ out_mids)
{
    const int64_t total = seg.n_rays * n_out;
    for (int64_t tid = (int64_t)blockIdx.x * kBlock + threadIdx.x; tid < total; tid += (int64_t)gridDim.x * kBlock) {
        const int64_t ray = tid / n_out, sid = tid - ray * n_out;
        int64_t base, last;
        ray_span(seg, ray, base, last);
        float *edges = out_edges + ray * (n_out + 1);
        if (last < base) {                       // a ray without edges: nothing to invert
            out_mids[tid] = 0.0f;
            edges[sid] = 0.0f;
            if (sid == n_out - 1) edges[n_out] = 0.0f;
            continue;
        }
        const float bias = jitter ? jitter[ray] : 0.5f;
        const float t = draw_sample(seg.vals, cdfs, base, last, sid, n_out, bias);
        out_mids[tid] = t;
        const float tmin = seg.vals[base], tmax = seg.vals[last];
        // pdf.cu:207-239: edges are midpoints of neighbouring samples, ends mirrored + clamped
        if (sid == 0) {
            const float t_next = (n_out > 1) ? draw_sample(seg.vals, cdfs, base, last, 1, n_out, bias) : t;
            edges[0] = fmaxf(t - (t_next - t) * 0.5f, tmin);
            if (n_out == 1) edges[1] = fminf(t, tmax);
        } else {
            const float t_prev = draw_sample(seg.vals, cdfs, base, last, sid - 1, n_out, bias);
            edges[sid] = (t + t_prev) * 0.5f;
            if (sid == n_out - 1) edges[sid + 1] = fminf(t + (t - t_prev) * 0.5f, tmax);
        }
    }
}

__device__ __forceinline__ int64_t chunk_of(const int64_t *__restrict__ starts, int64_t n_chunks, int64_t item) {
    int64_t lo = 0, hi = n_chunks;               // pdf.cu:65-80: last chunk with start <= item
    while (lo < hi) { const int64_t m = lo + ((hi - lo) >> 1); if (!(starts[m] > item)) lo = m + 1; else hi = m; }
    return lo - 1;
}

// importance_sampling with a PER-RAY number of intervals (nerfacc.cpp:100-105, pdf.cu:294-357: the overload whose reference
// implementation sizes its outputs with memalloc_data's missing argument and so allocates nothing).  Semantics as the reference's
// two kernels state them for flattened outputs (pdf.cu:112-116, 207-239): ray r gets n_r samples at sm_starts[r] .. and, when n_r > 0,
// n_r + 1 interval edges at iv_starts[r] .. flagged is_left (all but the last) / is_right (all but the first); a ray with n_r = 0
// gets nothing.  One lane per output sample; the per-sample arithmetic is draw_sample / the edge rule of the batched kernel above, so
// a ray's values equal the batched call's with n_intervals = n_r bit for bit.  n_r = 1 writes both edges (the reference reads the
// next ray's first sample there: pdf.cu:211 "FIXME: out of bounds?").
__global__ __launch_bounds__(kBlock) void importance_sampling_ragged_kernel(
    nfa_ray_segments seg, const float *__restrict__ cdfs, const int64_t *__restrict__ sm_starts, const int64_t *__restrict__ sm_cnts,
    const int64_t *__restrict__ iv_starts, int64_t n_samples, const float *__restrict__ jitter,
    float *__restrict__ sm_vals, int64_t *__restrict__ sm_ray_indices, float *__restrict__ iv_vals, int64_t *__restrict__ iv_ray_indices,
    uint8_t *__restrict__ iv_is_left, uint8_t *__restrict__ iv_is_right)
{
    for (int64_t tid = (int64_t)blockIdx.x * kBlock + threadIdx.x; tid < n_samples; tid += (int64_t)gridDim.x * kBlock) {
        const int64_t ray = chunk_of(sm_starts, seg.n_rays, tid);        // (rays without samples share their start with the next ray: the
        const int64_t n_out = sm_cnts[ray], sid = tid - sm_starts[ray];  //  LAST chunk that starts at or before tid is the one that owns it)
        int64_t base, last;
        ray_span(seg, ray, base, last);
        const int64_t e0 = iv_starts[ray];
        auto edge = [&](int64_t k, float v, bool left, bool right) {
            iv_vals[e0 + k] = v;
            if (iv_ray_indices) iv_ray_indices[e0 + k] = ray;
            if (iv_is_left) iv_is_left[e0 + k] = left ? 1 : 0;
            if (iv_is_right) iv_is_right[e0 + k] = right ? 1 : 0;
        };
        if (sm_ray_indices) sm_ray_indices[tid] = ray;
        if (last < base) {                       // a ray without edges: nothing to invert
            sm_vals[tid] = 0.0f;
            edge(sid, 0.0f, true, sid > 0);
            if (sid == n_out - 1) edge(n_out, 0.0f, false, true);
            continue;
        }
        const float bias = jitter ? jitter[ray] : 0.5f;
        const float t = draw_sample(seg.vals, cdfs, base, last, sid, n_out, bias);
        sm_vals[tid] = t;
        const float tmin = seg.vals[base], tmax = seg.vals[last];
        if (sid == 0) {
            const float t_next = (n_out > 1) ? draw_sample(seg.vals, cdfs, base, last, 1, n_out, bias) : t;
            edge(0, fmaxf(t - (t_next - t) * 0.5f, tmin), true, false);
            if (n_out == 1) edge(1, fminf(t, tmax), false, true);
        } else {
            const float t_prev = draw_sample(seg.vals, cdfs, base, last, sid - 1, n_out, bias);
            edge(sid, (t + t_prev) * 0.5f, true, true);
            if (sid == n_out - 1) edge(sid + 1, fminf(t + (t - t_prev) * 0.5f, tmax), false, true);
        }
    }
}

__global__ __launch_bounds__(kBlock) void searchsorted_kernel(
    nfa_ray_segments query, nfa_ray_segments key, int64_t *__restrict__ ids_left, int64_t *__restrict__ ids_right)
{
    const bool q_batched = query.chunk_starts == nullptr;
    for (int64_t tid = (int64_t)blockIdx.x * kBlock + threadIdx.x; tid < query.n_edges; tid += (int64_t)gridDim.x * kBlock) {
        int64_t ray;
        if (q_batched) ray = tid / query.n_edges_per_ray;
        else if (query.ray_indices) ray = query.ray_indices[tid];
        else ray = chunk_of(query.chunk_starts, query.n_rays, tid);
        int64_t base, last;
        ray_span(key, ray, base, last);
        const int64_t p = upper_bound_f(key.vals, base, last, query.vals[tid]);
        const int64_t l = clamp64(p - 1, base, last), r = clamp64(p, base, last);
        ids_left[tid] = q_batched ? l - base : l;
        ids_right[tid] = q_batched ? r - base : r;
    }
}

int check_segments(const nfa_ray_segments *s, const char *who) {
    NFA_REQUIRE(s != nullptr, "%s: segments is NULL", who);
    NFA_REQUIRE(s->n_rays >= 0 && s->n_edges >= 0, "%s: negative size", who);
    if (s->n_edges > 0) NFA_REQUIRE(s->vals != nullptr, "%s: vals is NULL", who);
    if (s->chunk_starts) NFA_REQUIRE(s->chunk_cnts != nullptr, "%s: chunk_starts without chunk_cnts", who);
    else NFA_REQUIRE(s->n_edges_per_ray >= 0 && s->n_rays * s->n_edges_per_ray == s->n_edges, "%s: batched shape mismatch", who);
    return NFA_OK;
}

// normalised s in [0, 1] -> ray distance t (prop_net.py:215-229): the reference's torch expressions, operation for operation
// (no contraction: the Makefile builds with -ffp-contract=off), in one launch instead of five
//   uniform:  s * t_max + (1 - s) * t_min           lindisp:  1 / (s * (1 / t_max) + (1 - s) * (1 / t_min))
__global__ __launch_bounds__(kBlock) void transform_stot_kernel(const float *__restrict__ s, int64_t n, float c_max, float c_min,
                                                                int lindisp, float *__restrict__ t)
{
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
        const float v = s[i];
        const float r = v * c_max + (1.0f - v) * c_min;
        t[i] = lindisp ? 1.0f / r : r;
    }
}

// cdf at the n + 1 edges of a proposal level from the level's densities (prop_net.py:99-112: render_transmittance_from_density on
// batched tensors — sigma * (t_end - t_start), exclusive cumsum, exp(-.) — then 1 - cat([trans, 0])): one wave per ray, 64
// samples per trip, the DPP wave scan of common.hpp with a carry.  The reference spends ~13 elementwise launches + a cumsum on it
// per level (and as many in backward), each launch-bound at 4096 x 256.  `trans` (optional) is what the backward needs.
__global__ __launch_bounds__(kBlock) void edge_cdfs_fwd_kernel(const float *__restrict__ t_edges, const float *__restrict__ sigmas,
                                                               int64_t n_rays, int64_t S, float *__restrict__ cdfs, float *__restrict__ trans)
{
    const int lane = lane_id();
    for (int64_t row = (int64_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6); row < n_rays; row += (int64_t)gridDim.x * kWavesPerBlock) {
        const float *t = t_edges + row * (S + 1);
        const float *sg = sigmas + row * S;
        float carry = 0.0f;
        for (int64_t c = 0; c < S; c += 64) {
            const int64_t j = c + lane;
            const float x = j < S ? sg[j] * (t[j + 1] - t[j]) : 0.0f;
            const float incl = wave_seg_scan_fwd<OpSum>(x, lane);
            float before = lane_prev_f(incl, 0.0f);                  // (no incl - x: x may be inf, the opaque last sample)
            before = carry + (lane == 0 ? 0.0f : before);
            carry = carry + readlane_f<63>(incl);
            if (j < S) {
                const float T = expf(-before);
                cdfs[row * (S + 1) + j] = 1.0f - T;
                if (trans) trans[row * S + j] = T;
            }
        }
        if (lane == 0) cdfs[row * (S + 1) + S] = 1.0f;
    }
}

// g_sigma_i = (t_{i+1} - t_i) * sum_{i < j < S} g_cdfs_j T_j   (cdfs_j = 1 - T_j, T_j = exp(-sum_{i<j} sigma_i delta_i); the last edge
// is the constant 1)
__global__ __launch_bounds__(kBlock) void edge_cdfs_bwd_kernel(const float *__restrict__ t_edges, const float *__restrict__ trans,
                                                               const float *__restrict__ g_cdfs, int64_t n_rays, int64_t S,
                                                               float *__restrict__ g_sigmas)
{
    const int lane = lane_id();
    for (int64_t row = (int64_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6); row < n_rays; row += (int64_t)gridDim.x * kWavesPerBlock) {
        const float *t = t_edges + row * (S + 1);
        float carry = 0.0f;
        for (int64_t c = ((S - 1) / 64) * 64; c >= 0; c -= 64) {
            const int64_t j = c + lane;
            const float y = j < S ? g_cdfs[row * (S + 1) + j] * trans[row * S + j] : 0.0f;
            const float incl = wave_seg_scan_bwd<OpSum>(y, 63 - lane);          // sum over lanes >= this one
            float after = lane_next_f(incl, 0.0f);
            after = (lane == 63 ? 0.0f : after) + carry;
            carry = carry + readlane_f<0>(incl);
            if (j < S) g_sigmas[row * S + j] = (t[j + 1] - t[j]) * after;
        }
    }
}

// Histogram-envelope loss of one proposal level, batched layout (prop_net.py:232-256, `_pdf_loss`): searchsorted of the query
// edges in the key edges, w = cdf mass of the query interval, w_outer = mass of the key intervals that overlap it,
// loss = clip(w - w_outer, 0)^2 / (w + eps) — the reference's float operations in its order, one lane per query interval, one
// launch instead of searchsorted + ~10 ATen ops (and ~15 in backward).  il / ir (indices into the key row) and
// coef = d loss / d w_outer are what the backward reads.
__global__ __launch_bounds__(kBlock) void pdf_loss_fwd_kernel(const float *__restrict__ q, const float *__restrict__ cq,
                                                              const float *__restrict__ k, const float *__restrict__ ck,
                                                              int64_t n_rays, int64_t nq, int64_t nk, float eps,
                                                              float *__restrict__ loss, int32_t *__restrict__ il, int32_t *__restrict__ ir,
                                                              float *__restrict__ coef)
{
    const int64_t total = n_rays * nq;
    for (int64_t tid = (int64_t)blockIdx.x * kBlock + threadIdx.x; tid < total; tid += (int64_t)gridDim.x * kBlock) {
        const int64_t ray = tid / nq, i = tid - ray * nq;
        const float *qr = q + ray * (nq + 1), *cqr = cq + ray * (nq + 1);
        const int64_t base = ray * (nk + 1), last = base + nk;
        const int64_t pl = upper_bound_f(k, base, last, qr[i]), pr = upper_bound_f(k, base, last, qr[i + 1]);
        const int64_t l = clamp64(pl - 1, base, last), r = clamp64(pr, base, last);      // searchsorted_kernel's (left, right)
        const float w = cqr[i + 1] - cqr[i];
        const float w_outer = ck[r] - ck[l];
        const float d = w - w_outer;
        const float c = d < 0.0f ? 0.0f : d;               // torch.clip(min = 0): a NaN stays a NaN
        const float den = w + eps;
        loss[tid] = (c * c) / den;
        if (il) { il[tid] = (int32_t)(l - base); ir[tid] = (int32_t)(r - base); coef[tid] = -2.0f * c / den; }
    }
}

// d loss / d cdfs_key: w_outer = ck[ir] - ck[il], so key edge e collects + g coef of the intervals whose right index is e and
// - g coef of those whose left index is e.  One wave per ray.  When il and ir ascend along the ray (query edges that ascend, as the
// ones importance_sampling returns do) each edge's intervals are one stretch, found by bisection and summed in order; the wave
// checks that first, and a ray whose ids do not ascend (unsorted or NaN query edges: ADVICE r3 — the forward is still the
// reference's, and so must the gradient be) has every edge scan all of the ray's intervals, in order.  No atomics either way: the
// same bits every run.
__global__ __launch_bounds__(kBlock) void pdf_loss_bwd_kernel(const float *__restrict__ g_loss, const int32_t *__restrict__ il,
                                                              const int32_t *__restrict__ ir, const float *__restrict__ coef,
                                                              int64_t n_rays, int64_t nq, int64_t nk, float *__restrict__ g_ck)
{
    const int lane = lane_id();
    for (int64_t ray = (int64_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6); ray < n_rays; ray += (int64_t)gridDim.x * kWavesPerBlock) {
        const int64_t row = ray * nq;
        bool bad = false;
        for (int64_t i = lane; i + 1 < nq; i += 64) bad = bad || il[row + i] > il[row + i + 1] || ir[row + i] > ir[row + i + 1];
        const bool ascending = __ballot(bad) == 0ull;
        for (int64_t e = lane; e <= nk; e += 64) {
            float acc = 0.0f;
            if (ascending) {
                auto stretch = [&](const int32_t *__restrict__ ids, int64_t &lo, int64_t &hi) {      // [lo, hi): ids == e
                    int64_t a = 0, b = nq;
                    while (a < b) { const int64_t m = a + ((b - a) >> 1); if (ids[row + m] < e) a = m + 1; else b = m; }
                    lo = a;
                    b = nq;
                    while (a < b) { const int64_t m = a + ((b - a) >> 1); if (ids[row + m] <= e) a = m + 1; else b = m; }
                    hi = a;
                };
                int64_t lo, hi;
                stretch(ir, lo, hi);
                for (int64_t i = lo; i < hi; ++i) acc += g_loss[row + i] * coef[row + i];
                stretch(il, lo, hi);
                for (int64_t i = lo; i < hi; ++i) acc -= g_loss[row + i] * coef[row + i];
            } else {
                float sub = 0.0f;
                for (int64_t i = 0; i < nq; ++i) {
                    const float v = g_loss[row + i] * coef[row + i];
                    if (ir[row + i] == e) acc += v;
                    if (il[row + i] == e) sub += v;
                }
                acc -= sub;
            }
            g_ck[ray * (nk + 1) + e] = acc;
        }
    }
}

}  // namespace
}  // namespace nfa

using namespace nfa;

NFA_EXPORT int nfa_pdf_loss_fwd(const float *query_vals, const float *cdfs_query, const float *key_vals, const float *cdfs_key,
                                int64_t n_rays, int64_t n_query, int64_t n_key, float eps, float *loss, int32_t *ids_left,
                                int32_t *ids_right, float *coef, void *stream)
{
    NFA_REQUIRE(n_rays >= 0 && n_query >= 1 && n_key >= 1, "pdf_loss_fwd: n_rays < 0 or an empty level");
    if (n_rays == 0) return NFA_OK;
    NFA_REQUIRE(query_vals && cdfs_query && key_vals && cdfs_key && loss, "pdf_loss_fwd: NULL pointer");
    NFA_REQUIRE((ids_left != nullptr) == (ids_right != nullptr) && (ids_left != nullptr) == (coef != nullptr),
                "pdf_loss_fwd: ids_left, ids_right and coef must be given together");
    hipLaunchKernelGGL(pdf_loss_fwd_kernel, dim3(blocks_for(n_rays * n_query)), dim3(kBlock), 0, (hipStream_t)stream, query_vals, cdfs_query,
                       key_vals, cdfs_key, n_rays, n_query, n_key, eps, loss, ids_left, ids_right, coef);
    return check_launch("pdf_loss_fwd_kernel");
}

NFA_EXPORT int nfa_pdf_loss_bwd(const float *g_loss, const int32_t *ids_left, const int32_t *ids_right, const float *coef,
                                int64_t n_rays, int64_t n_query, int64_t n_key, float *g_cdfs_key, void *stream)
{
    NFA_REQUIRE(n_rays >= 0 && n_query >= 1 && n_key >= 1, "pdf_loss_bwd: n_rays < 0 or an empty level");
    if (n_rays == 0) return NFA_OK;
    NFA_REQUIRE(g_loss && ids_left && ids_right && coef && g_cdfs_key, "pdf_loss_bwd: NULL pointer");
    hipLaunchKernelGGL(pdf_loss_bwd_kernel, dim3(blocks_for(n_rays * kWave)), dim3(kBlock), 0, (hipStream_t)stream, g_loss, ids_left,
                       ids_right, coef, n_rays, n_query, n_key, g_cdfs_key);
    return check_launch("pdf_loss_bwd_kernel");
}

NFA_EXPORT int nfa_edge_cdfs_fwd(const float *t_edges, const float *sigmas, int64_t n_rays, int64_t n_samples, float *cdfs, float *trans,
                                 void *stream)
{
    NFA_REQUIRE(n_rays >= 0 && n_samples >= 1, "edge_cdfs_fwd: n_rays < 0 or n_samples < 1");
    if (n_rays == 0) return NFA_OK;
    NFA_REQUIRE(t_edges && sigmas && cdfs, "edge_cdfs_fwd: NULL pointer");
    hipLaunchKernelGGL(edge_cdfs_fwd_kernel, dim3(blocks_for(n_rays * kWave)), dim3(kBlock), 0, (hipStream_t)stream, t_edges, sigmas, n_rays,
                       n_samples, cdfs, trans);
    return check_launch("edge_cdfs_fwd_kernel");
}

NFA_EXPORT int nfa_edge_cdfs_bwd(const float *t_edges, const float *trans, const float *g_cdfs, int64_t n_rays, int64_t n_samples,
                                 float *g_sigmas, void *stream)
{
    NFA_REQUIRE(n_rays >= 0 && n_samples >= 1, "edge_cdfs_bwd: n_rays < 0 or n_samples < 1");
    if (n_rays == 0) return NFA_OK;
    NFA_REQUIRE(t_edges && trans && g_cdfs && g_sigmas, "edge_cdfs_bwd: NULL pointer");
    hipLaunchKernelGGL(edge_cdfs_bwd_kernel, dim3(blocks_for(n_rays * kWave)), dim3(kBlock), 0, (hipStream_t)stream, t_edges, trans, g_cdfs,
                       n_rays, n_samples, g_sigmas);
    return check_launch("edge_cdfs_bwd_kernel");
}

NFA_EXPORT int nfa_transform_stot(const float *s_vals, int64_t n, double t_min, double t_max, int32_t lindisp, float *t_vals, void *stream)
{
    NFA_REQUIRE(n >= 0, "transform_stot: n < 0");
    if (n == 0) return NFA_OK;
    NFA_REQUIRE(s_vals && t_vals, "transform_stot: NULL pointer");
    // (the scalars as torch forms them: 1 / t in double FROM THE CALLER'S DOUBLE, rounded to float once, when it meets the float
    // tensor — a float parameter would round t first and differ by an ulp for ~27 % of scalars: t = 1e-3 gave 999.99994)
    const float c_max = lindisp ? (float)(1.0 / t_max) : (float)t_max, c_min = lindisp ? (float)(1.0 / t_min) : (float)t_min;
    hipLaunchKernelGGL(transform_stot_kernel, dim3(blocks_for(n)), dim3(kBlock), 0, (hipStream_t)stream, s_vals, n, c_max, c_min,
                       (int)lindisp, t_vals);
    return check_launch("transform_stot_kernel");
}

NFA_EXPORT int nfa_importance_sampling(const nfa_ray_segments *segments, const float *cdfs, int64_t n_intervals,
                                       const float *jitter, float *out_edges, float *out_mids, void *stream)
{
    if (int rc = check_segments(segments, "importance_sampling")) return rc;
    NFA_REQUIRE(n_intervals >= 0, "importance_sampling: n_intervals < 0");
    if (segments->n_rays == 0 || n_intervals == 0) return NFA_OK;
    NFA_REQUIRE(cdfs && out_edges && out_mids, "importance_sampling: NULL pointer");
    hipLaunchKernelGGL(importance_sampling_kernel, dim3(blocks_for(segments->n_rays * n_intervals)), dim3(kBlock), 0,
                       (hipStream_t)stream, *segments, cdfs, n_intervals, jitter, out_edges, out_mids);
    return check_launch("importance_sampling_kernel");
}

NFA_EXPORT int nfa_importance_sampling_ragged(const nfa_ray_segments *segments, const float *cdfs, const int64_t *sm_starts,
                                              const int64_t *sm_cnts, const int64_t *iv_starts, int64_t n_samples, const float *jitter,
                                              float *sm_vals, int64_t *sm_ray_indices, float *iv_vals, int64_t *iv_ray_indices,
                                              uint8_t *iv_is_left, uint8_t *iv_is_right, void *stream)
{
    if (int rc = check_segments(segments, "importance_sampling_ragged")) return rc;
    NFA_REQUIRE(n_samples >= 0, "importance_sampling_ragged: n_samples < 0");
    if (segments->n_rays == 0 || n_samples == 0) return NFA_OK;
    NFA_REQUIRE(cdfs && sm_starts && sm_cnts && iv_starts && sm_vals && iv_vals, "importance_sampling_ragged: NULL pointer");
    hipLaunchKernelGGL(importance_sampling_ragged_kernel, dim3(blocks_for(n_samples)), dim3(kBlock), 0, (hipStream_t)stream, *segments, cdfs,
                       sm_starts, sm_cnts, iv_starts, n_samples, jitter, sm_vals, sm_ray_indices, iv_vals, iv_ray_indices, iv_is_left, iv_is_right);
    return check_launch("importance_sampling_ragged_kernel");
}

NFA_EXPORT int nfa_searchsorted(const nfa_ray_segments *query, const nfa_ray_segments *key,
                                int64_t *ids_left, int64_t *ids_right, void *stream)
{
    if (int rc = check_segments(query, "searchsorted(query)")) return rc;
    if (int rc = check_segments(key, "searchsorted(key)")) return rc;
    if (query->n_edges == 0) return NFA_OK;
    NFA_REQUIRE(ids_left && ids_right, "searchsorted: NULL output");
    NFA_REQUIRE(key->n_edges > 0, "searchsorted: empty key");
    hipLaunchKernelGGL(searchsorted_kernel, dim3(blocks_for(query->n_edges)), dim3(kBlock), 0, (hipStream_t)stream,
                       *query, *key, ids_left, ids_right);
    return check_launch("searchsorted_kernel");
}
