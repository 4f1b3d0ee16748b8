/*
 * oracle/nerfacc_oracle.c — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Scalar, single-threaded CPU restatement of the nerfacc 0.5.3 OccGrid
 * sampling + volumetric-rendering hot path.  It exists only so that tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg have something to
 * check the HIP kernels against (and to time).  Nothing under nerfacc_amd/
 * may import, link or call it.
 *
 * Every function cites the reference file:line it restates (paths relative to
 * the nerfacc repository root).  It is written from the reference's
 * *semantics*; loops, names and structure are this project's own.
 *
 * Floating-point discipline (matters for bit-exact sample counts):
 *   - compiled with -ffp-contract=off, so the compiler never fuses;
 *   - the places where nvcc's default (--fmad=true) fuses a multiply into a
 *     following add in the reference's CUDA path are written as explicit
 *     fmaf() here AND in the HIP kernels (nerfacc_amd/csrc/traverse.hip), so
 *     both sides round identically.  The list is in DESIGN.md ("FMA model").
 *   - float->int conversion saturates and maps NaN to 0, like the GPU's
 *     v_cvt_i32_f32 / CUDA's cvt.rzi.s32.f32 (x86's cvttss2si does not).
 *
 * Pinned against: the tests/golden/ fixtures (generated from the importable parts
 * of the Python reference by tests/golden/make_golden.py), the hand-computed
 * known answers in the reference's tests, and — for the
 * exact per-ray sample lists of traverse_grids, which the reference's tests hold
 * no vector for — tests/golden/k2_reference.npz: outputs of the reference's OWN
 * grid.cu compiled for the host (oracle/ref_shim -> oracle/_ref) and driven through
 * the reference's Python layer (tests/golden/make_k2_golden.py).  14 cases,
 * 10.2 M samples, bit for bit (tests/test_k2_reference.py).
 *
 * Threads: every entry point is single-threaded unless orc_set_threads(n > 1) was
 * called (bench.py's all-cores CPU baseline does).  Rays are independent, so the
 * parallel forms split the ray range (or the sample range at ray boundaries) and
 * run the same scalar code per part: results do not depend on the thread count.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_API __attribute__((visibility("default")))

static int orc_threads = 1;
ORC_API void orc_set_threads(int n) { orc_threads = n > 1 ? n : 1; }
ORC_API int orc_get_threads(void) { return orc_threads; }

/* cut [0, n) into `parts` ranges of about equal size whose boundaries fall on ray boundaries
 * (keys[i] != keys[i-1]); bounds has parts + 1 entries */
static void orc_split(int64_t n, const int64_t *keys, int parts, int64_t *bounds)
{
    bounds[0] = 0;
    for (int p = 1; p < parts; ++p) {
        int64_t i = n * p / parts;
        if (i < bounds[p - 1]) i = bounds[p - 1];
        while (i > 0 && i < n && keys[i] == keys[i - 1]) ++i;
        bounds[p] = i;
    }
    bounds[parts] = n;
}

/* ------------------------------------------------------------------ */
/* small helpers                                                        */
/* ------------------------------------------------------------------ */

static inline int f2i_sat(float x) {
    if (x != x) return 0;
    if (x >= 2147483648.0f) return INT32_MAX;
    if (x <= -2147483648.0f) return INT32_MIN;
    return (int)x; /* truncation toward zero */
}

static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* helper_math clamp(f,a,b) = fminf(fmaxf(f,a),b); grid.cu:23-28 (_calc_dt) */
static inline float march_dt(float t, float cone_angle, float dt_min, float dt_max) {
    return fminf(fmaxf(t * cone_angle, dt_min), dt_max);
}

/* ------------------------------------------------------------------ */
/* K1: ray / AABB slab test.  utils_grid.cuh:10-55, grid.cu:284-313      */
/* ------------------------------------------------------------------ */

static int slab_test(const float o[3], const float inv[3], const float *box,
                     float near, float far, float *t0_out, float *t1_out) {
    float t0 = 0.f, t1 = 0.f;
    for (int ax = 0; ax < 3; ++ax) {
        const float lo = box[ax], hi = box[3 + ax];
        float a, b;
        if (inv[ax] >= 0) { a = (lo - o[ax]) * inv[ax]; b = (hi - o[ax]) * inv[ax]; }
        else              { a = (hi - o[ax]) * inv[ax]; b = (lo - o[ax]) * inv[ax]; }
        if (ax == 0) { t0 = a; t1 = b; continue; }
        if (t0 > b || a > t1) return 0;
        if (a > t0) t0 = a;
        if (b < t1) t1 = b;
    }
    if (t1 <= 0) return 0;
    *t0_out = fmaxf(t0, near);
    *t1_out = fminf(t1, far);
    return 1;
}

ORC_API void orc_ray_aabb_intersect(
    int64_t n_rays, const float *rays_o, const float *rays_d,
    int64_t n_aabbs, const float *aabbs, float near, float far, float miss,
    float *t_mins, float *t_maxs, uint8_t *hits)
{
    for (int64_t r = 0; r < n_rays; ++r) {
        const float *o = rays_o + 3 * r, *d = rays_d + 3 * r;
        /* data_spec_packed.cuh:49 — inv_dir = 1/dir, +-inf for dir == 0 */
        const float inv[3] = {1.0f / d[0], 1.0f / d[1], 1.0f / d[2]};
        for (int64_t g = 0; g < n_aabbs; ++g) {
            float a, b;
            const int hit = slab_test(o, inv, aabbs + 6 * g, near, far, &a, &b);
            const int64_t k = r * n_aabbs + g;
            t_mins[k] = hit ? a : miss;
            t_maxs[k] = hit ? b : miss;
            hits[k] = (uint8_t)hit;
        }
    }
}

/* ------------------------------------------------------------------ */
/* K2: multi-level occupancy-grid traversal.  grid.cu:68-282,            */
/* utils_grid.cuh:58-142, utils_contraction.cuh:19-24                    */
/* ------------------------------------------------------------------ */

typedef struct {
    float tdist[3], delta[3];
    int step[3], cur[3], overflow[3];
} dda_t;

/* utils_grid.cuh:58-114 with the nvcc-fmad contraction model made explicit */
static void dda_setup(dda_t *s, const float o[3], const float d[3], const float inv[3],
                      float tmin, float tmax, float eps, const float *box, const int res[3])
{
    const float t_in = tmin + eps, t_out = tmax - eps;
    int fin[3];
    for (int ax = 0; ax < 3; ++ax) {
        const float lo = box[ax], hi = box[3 + ax];
        const float resf = (float)res[ax];
        const float vox = (hi - lo) / resf;                 /* voxel_size      */
        const float p_in = fmaf(d[ax], t_in, o[ax]);        /* ray_start (fma) */
        const float p_out = fmaf(d[ax], t_out, o[ax]);      /* ray_end   (fma) */
        /* roi_to_unit then * res, truncated, clamped */
        s->cur[ax] = clampi(f2i_sat(((p_in - lo) / (hi - lo)) * resf), 0, res[ax] - 1);
        fin[ax] = clampi(f2i_sat(((p_out - lo) / (hi - lo)) * resf), 0, res[ax] - 1);
        const int first_plane = s->cur[ax] + (d[ax] > 0 ? 1 : 0);
        /* ((aabb.min + (idx*vox - ray_start)) * inv_dir) + tmin :
           inner mul-sub and outer mul-add are each one fma under nvcc */
        const float inner = fmaf((float)first_plane, vox, -p_in);
        const float t_plane = fmaf(lo + inner, inv[ax], tmin);
        const float sgn = (d[ax] == 0.0f) ? 0.0f : (d[ax] > 0.0f ? 1.0f : -1.0f);
        s->step[ax] = (int)sgn;
        s->tdist[ax] = (d[ax] == 0.0f) ? tmax : t_plane;
        s->delta[ax] = (d[ax] == 0.0f) ? tmax : (vox * inv[ax]) * sgn;
        s->overflow[ax] = fin[ax] + s->step[ax];            /* grid.cu:183 */
    }
}

/* utils_grid.cuh:116-142: advance along the axis with the strictly smallest
 * tdist (ties resolved x -> y -> z by the strict '<'); 0 = left the segment */
static int dda_advance(dda_t *s) {
    int ax;
    if (s->tdist[0] < s->tdist[1] && s->tdist[0] < s->tdist[2]) ax = 0;
    else if (s->tdist[1] < s->tdist[2]) ax = 1;
    else ax = 2;
    s->cur[ax] += s->step[ax];
    s->tdist[ax] += s->delta[ax];
    return s->cur[ax] != s->overflow[ax];
}

/* advance the marching lattice: repeat t += dt (dt fixed) until the midpoint
 * t + dt/2 reaches `target`.  grid.cu:157-161 and :199-203.  The reference
 * spins forever when t + dt == t or on NaN; we stop instead (documented). */
static float lattice_skip(float t, float dt, float target) {
    while (t + dt * 0.5f < target) {
        const float nt = t + dt;
        if (nt == t) return target;
        t = nt;
    }
    return t;
}

typedef struct {
    /* interval edges (may be NULL) */
    float *iv_vals; int64_t *iv_ray; uint8_t *iv_left; uint8_t *iv_right;
    /* samples (may be NULL) */
    float *sm_vals; int64_t *sm_ray; uint8_t *sm_valid;
    /* the interval of every sample, written directly: what occ_grid.py:174-175 extracts from the edges with
     * vals[is_left] / vals[is_right] (may be NULL; needs sm_ray) */
    float *t_starts; float *t_ends;
} orc_emit_t;

/* one ray of traverse_grids_kernel; `emit` NULL => counting pass */
static void traverse_ray(
    int64_t tid, const float *rays_o, const float *rays_d,
    int n_grids, const int res[3], const uint8_t *binaries, const float *aabbs,
    const uint8_t *hits, const float *t_sorted, const int64_t *t_indices,
    float near, float far, float step_size, float cone_angle, int32_t steps_limit,
    const orc_emit_t *emit, int64_t iv_base, int64_t sm_base,
    int64_t *n_iv_out, int64_t *n_sm_out, float *t_term_out)
{
    const float eps = 1e-6f;                                 /* grid.cu:95 */
    const float *o = rays_o + 3 * tid, *d = rays_d + 3 * tid;
    const float inv[3] = {1.0f / d[0], 1.0f / d[1], 1.0f / d[2]};
    const uint8_t *hit = hits + tid * n_grids;
    const float *ts = t_sorted + tid * n_grids * 2;
    const int64_t *ti = t_indices + tid * n_grids * 2;
    const int64_t cells = (int64_t)res[0] * res[1] * res[2];

    int64_t n_iv = 0, n_sm = 0;
    float t_last = near;
    int continuous = 0;

    for (int i = 0; i + 1 < 2 * n_grids; ++i) {
        /* which grid are we inside between event i and event i+1?  grid.cu:131-146 */
        int64_t level = ti[i] % n_grids;
        if (!hit[level]) continue;
        if (!(ti[i] < n_grids)) {            /* a leaving event */
            if (ti[i + 1] < n_grids) continue; /* next one enters => gap */
            level = ti[i + 1] % n_grids;
            if (!hit[level]) continue;
        }
        const float seg_lo = fmaxf(ts[i], near);
        const float seg_hi = fminf(ts[i + 1], far);
        if (seg_lo >= seg_hi) continue;

        if (!continuous) {                    /* grid.cu:153-163 */
            if (step_size <= 0.0f) t_last = seg_lo;
            else t_last = lattice_skip(t_last, march_dt(t_last, cone_angle, step_size, 1e10f), seg_lo);
        }

        dda_t s;
        dda_setup(&s, o, d, inv, seg_lo, seg_hi, eps, aabbs + 6 * level, res);

        while (steps_limit <= 0 || n_sm < steps_limit) {
            float t_cell = fminf(s.tdist[0], fminf(s.tdist[1], s.tdist[2]));
            t_cell = fminf(t_cell, seg_hi);
            const int64_t cell = ((int64_t)s.cur[0] * res[1] + s.cur[1]) * res[2] + s.cur[2]
                                 + level * cells;
            if (!binaries[cell]) {            /* empty: jump the lattice past it */
                if (step_size <= 0.0f) t_last = t_cell;
                else t_last = lattice_skip(t_last, march_dt(t_last, cone_angle, step_size, 1e10f), t_cell);
                continuous = 0;
            } else {
                while (steps_limit <= 0 || n_sm < steps_limit) {
                    float t_next;
                    if (step_size <= 0.0f) t_next = t_cell;
                    else {
                        const float dt = march_dt(t_last, cone_angle, step_size, 1e10f);
                        if (t_last + dt * 0.5f >= t_cell) break;
                        t_next = t_last + dt;
                    }
                    if (n_iv_out) {           /* interval edges, grid.cu:219-245 */
                        if (!continuous) {
                            if (emit && emit->iv_vals) {
                                const int64_t k = iv_base + n_iv;
                                emit->iv_vals[k] = t_last; emit->iv_ray[k] = tid; emit->iv_left[k] = 1;
                                emit->iv_vals[k + 1] = t_next; emit->iv_ray[k + 1] = tid; emit->iv_right[k + 1] = 1;
                            }
                            n_iv += 2;
                        } else {
                            if (emit && emit->iv_vals) {
                                const int64_t k = iv_base + n_iv;
                                emit->iv_vals[k] = t_next; emit->iv_ray[k] = tid;
                                emit->iv_left[k - 1] = 1; emit->iv_right[k] = 1;
                            }
                            n_iv += 1;
                        }
                    }
                    if (emit && emit->t_starts) {
                        const int64_t k = sm_base + n_sm;
                        emit->t_starts[k] = t_last; emit->t_ends[k] = t_next;
                        if (!emit->sm_vals) emit->sm_ray[k] = tid;
                    }
                    if (emit && emit->sm_vals) { /* sample midpoint, grid.cu:248-255 */
                        const int64_t k = sm_base + n_sm;
                        emit->sm_vals[k] = (t_next + t_last) * 0.5f;
                        emit->sm_ray[k] = tid;
                        if (emit->sm_valid) emit->sm_valid[k] = 1;
                    }
                    n_sm += 1;
                    continuous = 1;
                    t_last = t_next;
                    if (t_next >= t_cell) break;
                }
            }
            if (!dda_advance(&s)) break;
        }
    }
    if (t_term_out) *t_term_out = t_last;
    if (n_iv_out) *n_iv_out = n_iv;
    if (n_sm_out) *n_sm_out = n_sm;
}

/* Counting pass (first_pass=true in grid.cu:413): per-ray edge / sample counts.
 * rays_mask may be NULL (the reference's two-pass mode passes nullptr). */
ORC_API void orc_traverse_count(
    int64_t n_rays, const float *rays_o, const float *rays_d, const uint8_t *rays_mask,
    int32_t n_grids, const int32_t *res, const uint8_t *binaries, const float *aabbs,
    const uint8_t *hits, const float *t_sorted, const int64_t *t_indices,
    const float *near_planes, const float *far_planes,
    float step_size, float cone_angle, int32_t steps_limit,
    int64_t *iv_cnts, int64_t *sm_cnts, float *terminate_planes)
{
    const int r3[3] = {res[0], res[1], res[2]};
#pragma omp parallel for schedule(dynamic, 64) num_threads(orc_threads) if (orc_threads > 1)
    for (int64_t r = 0; r < n_rays; ++r) {
        if (rays_mask && !rays_mask[r]) continue;
        int64_t a = 0, b = 0;
        traverse_ray(r, rays_o, rays_d, n_grids, r3, binaries, aabbs, hits, t_sorted, t_indices,
                     near_planes[r], far_planes[r], step_size, cone_angle, steps_limit,
                     NULL, 0, 0, &a, &b, terminate_planes ? terminate_planes + r : NULL);
        if (iv_cnts) iv_cnts[r] = a;
        if (sm_cnts) sm_cnts[r] = b;
    }
}

/* Filling pass (grid.cu:445 / :375).  *_starts give each ray's output offset.
 * Rays whose given count is 0 are skipped like grid.cu:103-106 when
 * `skip_empty` (two-pass mode); counts are rewritten with the actual numbers. */
ORC_API void orc_traverse_fill(
    int64_t n_rays, const float *rays_o, const float *rays_d, const uint8_t *rays_mask,
    int32_t n_grids, const int32_t *res, const uint8_t *binaries, const float *aabbs,
    const uint8_t *hits, const float *t_sorted, const int64_t *t_indices,
    const float *near_planes, const float *far_planes,
    float step_size, float cone_angle, int32_t steps_limit,
    const int64_t *iv_starts, int64_t *iv_cnts, const int64_t *sm_starts, int64_t *sm_cnts,
    float *iv_vals, int64_t *iv_ray, uint8_t *iv_left, uint8_t *iv_right,
    float *sm_vals, int64_t *sm_ray, uint8_t *sm_valid,
    float *terminate_planes, float *t_starts, float *t_ends)
{
    const int r3[3] = {res[0], res[1], res[2]};
    orc_emit_t e = {iv_vals, iv_ray, iv_left, iv_right, sm_vals, sm_ray, sm_valid, t_starts, t_ends};
#pragma omp parallel for schedule(dynamic, 64) num_threads(orc_threads) if (orc_threads > 1)
    for (int64_t r = 0; r < n_rays; ++r) {
        if (rays_mask && !rays_mask[r]) continue;
        if (iv_cnts && iv_cnts[r] == 0) continue;
        if (sm_cnts && sm_cnts[r] == 0) continue;
        int64_t a = 0, b = 0;
        traverse_ray(r, rays_o, rays_d, n_grids, r3, binaries, aabbs, hits, t_sorted, t_indices,
                     near_planes[r], far_planes[r], step_size, cone_angle, steps_limit,
                     &e, iv_starts ? iv_starts[r] : 0, sm_starts ? sm_starts[r] : 0,
                     iv_cnts ? &a : NULL, &b, terminate_planes ? terminate_planes + r : NULL);
        if (iv_cnts) iv_cnts[r] = a;
        if (sm_cnts) sm_cnts[r] = b;
    }
}

/* ------------------------------------------------------------------ */
/* Mask compactions of OccGridEstimator.sampling, occ_grid.py:174-176     */
/* (t_starts = vals[is_left], t_ends = vals[is_right]) and :218-220       */
/* (x = x[masks]) — in the reference these are torch boolean-mask         */
/* gathers; here one loop over rays so that the all-cores baseline of     */
/* bench.py has no serial numpy pass in it.  Same results as the gathers. */
/* ------------------------------------------------------------------ */
ORC_API void orc_intervals_to_samples(
    int64_t n_rays, const int64_t *iv_starts, const int64_t *iv_cnts, const float *iv_vals,
    const uint8_t *iv_left, const uint8_t *iv_right, const int64_t *sm_starts,
    float *t_starts, float *t_ends)
{
#pragma omp parallel for schedule(dynamic, 64) num_threads(orc_threads) if (orc_threads > 1)
    for (int64_t r = 0; r < n_rays; ++r) {
        int64_t a = sm_starts[r], b = sm_starts[r];
        for (int64_t k = iv_starts[r]; k < iv_starts[r] + iv_cnts[r]; ++k) {
            if (iv_left[k]) t_starts[a++] = iv_vals[k];
            if (iv_right[k]) t_ends[b++] = iv_vals[k];
        }
    }
}

/* kept[r] = number of samples of ray r with keep != 0 */
ORC_API void orc_count_kept(int64_t n_rays, const int64_t *starts, const int64_t *cnts, const uint8_t *keep, int64_t *kept)
{
#pragma omp parallel for schedule(dynamic, 64) num_threads(orc_threads) if (orc_threads > 1)
    for (int64_t r = 0; r < n_rays; ++r) {
        int64_t c = 0;
        for (int64_t k = starts[r]; k < starts[r] + cnts[r]; ++k) c += keep[k] != 0;
        kept[r] = c;
    }
}

/* out_starts = exclusive sum of the kept counts (caller); stable per-ray scatter of the survivors */
ORC_API void orc_compact_samples(
    int64_t n_rays, const int64_t *starts, const int64_t *cnts, const uint8_t *keep, const int64_t *out_starts,
    const int64_t *ray_indices, const float *t_starts, const float *t_ends,
    int64_t *o_ray_indices, float *o_t_starts, float *o_t_ends)
{
#pragma omp parallel for schedule(dynamic, 64) num_threads(orc_threads) if (orc_threads > 1)
    for (int64_t r = 0; r < n_rays; ++r) {
        int64_t o = out_starts[r];
        for (int64_t k = starts[r]; k < starts[r] + cnts[r]; ++k)
            if (keep[k]) { o_ray_indices[o] = ray_indices[k]; o_t_starts[o] = t_starts[k]; o_t_ends[o] = t_ends[k]; ++o; }
    }
}

/* ------------------------------------------------------------------ */
/* pack_info.  pack.py:38-46                                            */
/* ------------------------------------------------------------------ */
ORC_API void orc_pack_info(int64_t n, const int64_t *ray_indices, int64_t n_rays, int64_t *packed_info)
{
    for (int64_t r = 0; r < n_rays; ++r) { packed_info[2 * r] = 0; packed_info[2 * r + 1] = 0; }
    for (int64_t i = 0; i < n; ++i) packed_info[2 * ray_indices[i] + 1] += 1;
    int64_t run = 0;
    for (int64_t r = 0; r < n_rays; ++r) { packed_info[2 * r] = run; run += packed_info[2 * r + 1]; }
}

/* ------------------------------------------------------------------ */
/* Segmented scans.  scan.py:14-282, utils_scan.cuh:28-112, 153-239,     */
/* scan.cu:199-210 (prod backward).  Sequential fp32, index order.      */
/* op: 0 = sum, 1 = prod.  reverse: scan from the chunk's last element.  */
/* ------------------------------------------------------------------ */
ORC_API void orc_scan_packed(
    int64_t n_rays, const int64_t *starts, const int64_t *cnts, const float *in, float *out,
    int op, int inclusive, int reverse, int normalize)
{
    for (int64_t r = 0; r < n_rays; ++r) {
        const int64_t b = starts[r], c = cnts[r];
        if (c == 0) continue;
        float run = op ? 1.0f : 0.0f;
        for (int64_t k = 0; k < c; ++k) {
            const int64_t i = reverse ? (b + c - 1 - k) : (b + k);
            const float v = in[i];
            if (inclusive) { run = op ? run * v : run + v; out[i] = run; }
            else           { out[i] = run; run = op ? run * v : run + v; }
        }
        if (normalize) {                      /* utils_scan.cuh:101-109, 228-236 */
            const float den = fmaxf(run, 1e-10f);
            for (int64_t k = 0; k < c; ++k) {
                const int64_t i = b + k;
                if (inclusive || (reverse ? k != c - 1 : k != 0)) out[i] /= den;
            }
        }
    }
}

/* keyed by (sorted) ray indices: scan_cub.cu:18-56 semantics */
ORC_API void orc_scan_keyed(
    int64_t n, const int64_t *keys, const float *in, float *out, int op, int inclusive, int reverse)
{
    float run = op ? 1.0f : 0.0f;
    for (int64_t k = 0; k < n; ++k) {
        const int64_t i = reverse ? (n - 1 - k) : k;
        const int64_t p = reverse ? i + 1 : i - 1;
        if (k == 0 || keys[p] != keys[i]) run = op ? 1.0f : 0.0f;
        const float v = in[i];
        if (inclusive) { run = op ? run * v : run + v; out[i] = run; }
        else           { out[i] = run; run = op ? run * v : run + v; }
    }
}

/* ------------------------------------------------------------------ */
/* Volumetric rendering.  volrend.py:219-278, 326-376, 379-494, 497-561  */
/* keyed by ray_indices (contiguous runs); sequential fp32.              */
/* ------------------------------------------------------------------ */

/* transmittance / alpha / weight from density; prefix_trans optional */
static void weight_from_density_range(
    int64_t b0, int64_t b1, const int64_t *ray_indices, const float *t_starts, const float *t_ends,
    const float *sigmas, const float *prefix_trans, float *weights, float *trans, float *alphas);

ORC_API void orc_render_weight_from_density(
    int64_t n, const int64_t *ray_indices, const float *t_starts, const float *t_ends,
    const float *sigmas, const float *prefix_trans, float *weights, float *trans, float *alphas)
{
    if (orc_threads > 1 && n > 4096) {
        const int P = orc_threads;
        int64_t *bounds = (int64_t *)malloc(sizeof(int64_t) * (size_t)(P + 1));
        orc_split(n, ray_indices, P, bounds);
#pragma omp parallel for schedule(static, 1) num_threads(P)
        for (int p = 0; p < P; ++p)
            weight_from_density_range(bounds[p], bounds[p + 1], ray_indices, t_starts, t_ends, sigmas, prefix_trans, weights, trans, alphas);
        free(bounds);
        return;
    }
    weight_from_density_range(0, n, ray_indices, t_starts, t_ends, sigmas, prefix_trans, weights, trans, alphas);
}

static void weight_from_density_range(
    int64_t b0, int64_t b1, const int64_t *ray_indices, const float *t_starts, const float *t_ends,
    const float *sigmas, const float *prefix_trans, float *weights, float *trans, float *alphas)
{
    float acc = 0.0f;
    for (int64_t i = b0; i < b1; ++i) {
        if (i == b0 || ray_indices[i] != ray_indices[i - 1]) acc = 0.0f;
        const float sd = sigmas[i] * (t_ends[i] - t_starts[i]);
        const float a = 1.0f - expf(-sd);
        float T = expf(-acc);
        if (prefix_trans) T = T * prefix_trans[i];
        if (alphas) alphas[i] = a;
        if (trans) trans[i] = T;
        if (weights) weights[i] = T * a;
        acc += sd;
    }
}

/* gradient of sum_i (gw_i w_i + gT_i T_i + ga_i a_i) w.r.t. sigma.
 * Derived from volrend.py:271-278 (T = exp(-excl_sum(sd)) * prefix,
 * a = 1 - exp(-sd), w = T a) — what autograd produces for the reference. */
static void weight_from_density_bwd_range(
    int64_t b0, int64_t b1, const int64_t *ray_indices, const float *t_starts, const float *t_ends,
    const float *sigmas, const float *prefix_trans,
    const float *g_w, const float *g_T, const float *g_a, float *g_sigmas)
{
    /* forward recompute in double to keep the oracle's own error negligible */
    const int64_t n = b1 - b0;
    if (n <= 0) return;
    double *T = (double *)malloc(sizeof(double) * (size_t)n);
    double acc = 0.0;
    for (int64_t i = b0; i < b1; ++i) {
        if (i == b0 || ray_indices[i] != ray_indices[i - 1]) acc = 0.0;
        T[i - b0] = exp(-acc) * (prefix_trans ? (double)prefix_trans[i] : 1.0);
        acc += (double)sigmas[i] * ((double)t_ends[i] - (double)t_starts[i]);
    }
    double suffix = 0.0;
    for (int64_t i = b1 - 1; i >= b0; --i) {
        if (i == b1 - 1 || ray_indices[i] != ray_indices[i + 1]) suffix = 0.0;
        const double dt = (double)t_ends[i] - (double)t_starts[i];
        const double sd = (double)sigmas[i] * dt;
        const double one_m_a = exp(-sd), a = 1.0 - one_m_a;
        const double Ti = T[i - b0];
        const double gw = g_w ? g_w[i] : 0.0, gT = g_T ? g_T[i] : 0.0, ga = g_a ? g_a[i] : 0.0;
        const double g_sd = (gw * Ti + ga) * one_m_a - suffix;
        g_sigmas[i] = (float)(g_sd * dt);
        suffix += gw * Ti * a + gT * Ti;
    }
    free(T);
}

ORC_API void orc_render_weight_from_density_bwd(
    int64_t n, const int64_t *ray_indices, const float *t_starts, const float *t_ends,
    const float *sigmas, const float *prefix_trans,
    const float *g_w, const float *g_T, const float *g_a, float *g_sigmas)
{
    const int P = (orc_threads > 1 && n > 4096) ? orc_threads : 1;
    int64_t *bounds = (int64_t *)malloc(sizeof(int64_t) * (size_t)(P + 1));
    orc_split(n, ray_indices, P, bounds);
#pragma omp parallel for schedule(static, 1) num_threads(P) if (P > 1)
    for (int p = 0; p < P; ++p)
        weight_from_density_bwd_range(bounds[p], bounds[p + 1], ray_indices, t_starts, t_ends, sigmas, prefix_trans,
                                      g_w, g_T, g_a, g_sigmas);
    free(bounds);
}

/* from alphas: T = exclusive_prod(1 - a) (* prefix); volrend.py:167-216, 281-323 */
ORC_API void orc_render_weight_from_alpha(
    int64_t n, const int64_t *ray_indices, const float *alphas, const float *prefix_trans,
    float *weights, float *trans)
{
    float run = 1.0f;
    for (int64_t i = 0; i < n; ++i) {
        if (i == 0 || ray_indices[i] != ray_indices[i - 1]) run = 1.0f;
        float T = run;
        if (prefix_trans) T = T * prefix_trans[i];
        if (trans) trans[i] = T;
        if (weights) weights[i] = T * alphas[i];
        run = run * (1.0f - alphas[i]);
    }
}

/* visibility mask; volrend.py:429-432, 490-494 (trans/alphas given) */
ORC_API void orc_visibility(
    int64_t n, const float *trans, const float *alphas, float early_stop_eps, float alpha_thre,
    uint8_t *vis)
{
#pragma omp parallel for schedule(static) num_threads(orc_threads) if (orc_threads > 1 && n > 4096)
    for (int64_t i = 0; i < n; ++i) {
        int v = trans[i] >= early_stop_eps;
        if (alpha_thre > 0.0f) v = v && (alphas[i] >= alpha_thre);
        vis[i] = (uint8_t)v;
    }
}

/* accumulate_along_rays; volrend.py:547-561.  out [n_rays, D] is ADDED to
 * (callers zero it first for the out-of-place form).  values may be NULL (D=1). */
ORC_API void orc_accumulate_along_rays(
    int64_t n, const int64_t *ray_indices, const float *weights, const float *values, int64_t D,
    float *out)
{
    for (int64_t i = 0; i < n; ++i) {
        float *row = out + ray_indices[i] * D;
        if (!values) row[0] += weights[i];
        else for (int64_t c = 0; c < D; ++c) row[c] += weights[i] * values[i * D + c];
    }
}

/* fused rendering(); volrend.py:15-164 (rgb_sigma_fn outputs given).
 * colors [R,3], opacities [R,1], depths [R,1]; weights/trans/alphas [N]. */
ORC_API void orc_rendering(
    int64_t n, int64_t n_rays, const int64_t *ray_indices,
    const float *t_starts, const float *t_ends, const float *sigmas, const float *rgbs,
    const float *bkgd /* 3 or NULL */, int expected_depths,
    float *weights, float *trans, float *alphas,
    float *colors, float *opacities, float *depths)
{
    orc_render_weight_from_density(n, ray_indices, t_starts, t_ends, sigmas, NULL, weights, trans, alphas);
    memset(colors, 0, sizeof(float) * 3 * (size_t)n_rays);
    memset(opacities, 0, sizeof(float) * (size_t)n_rays);
    memset(depths, 0, sizeof(float) * (size_t)n_rays);
    /* parts end on ray boundaries, so no two parts add into the same ray (sorted ray_indices, as everywhere here) */
    const int P = (orc_threads > 1 && n > 4096) ? orc_threads : 1;
    int64_t *bounds = (int64_t *)malloc(sizeof(int64_t) * (size_t)(P + 1));
    orc_split(n, ray_indices, P, bounds);
#pragma omp parallel for schedule(static, 1) num_threads(P) if (P > 1)
    for (int p = 0; p < P; ++p)
        for (int64_t i = bounds[p]; i < bounds[p + 1]; ++i) {
            const int64_t r = ray_indices[i];
            const float w = weights[i];
            for (int c = 0; c < 3; ++c) colors[3 * r + c] += w * rgbs[3 * i + c];
            opacities[r] += w;
            depths[r] += w * ((t_starts[i] + t_ends[i]) / 2.0f);
        }
    free(bounds);
    const float eps = 1.1920928955078125e-07f; /* torch.finfo(float32).eps */
#pragma omp parallel for schedule(static) num_threads(orc_threads) if (orc_threads > 1 && n_rays > 4096)
    for (int64_t r = 0; r < n_rays; ++r) {
        if (expected_depths) depths[r] = depths[r] / fmaxf(opacities[r], eps);
        if (bkgd) for (int c = 0; c < 3; ++c) colors[3 * r + c] += bkgd[c] * (1.0f - opacities[r]);
    }
}

/* ------------------------------------------------------------------ */
/* pdf: importance_sampling (stratified=false) and searchsorted,         */
/* batched layout [R, E].  pdf.cu:98-167, 169-241, 245-286               */
/* ------------------------------------------------------------------ */
static int64_t upper_bound_f(const float *v, int64_t lo, int64_t hi, float x) {
    while (lo < hi) {
        const int64_t mid = lo + ((hi - lo) >> 1);
        if (!(v[mid] > x)) lo = mid + 1; else hi = mid;
    }
    return lo;
}

ORC_API void orc_importance_sampling_batched(
    int64_t n_rays, int64_t n_edges, const float *vals, const float *cdfs, int64_t n_out,
    float *out_edges /* [R, n_out+1] */, float *out_mids /* [R, n_out] */)
{
    for (int64_t r = 0; r < n_rays; ++r) {
        const int64_t base = r * n_edges, last = base + n_edges - 1;
        const float u0 = cdfs[base], u1 = cdfs[last];
        const float du = (u1 - u0) / (float)n_out;
        float *m = out_mids + r * n_out, *e = out_edges + r * (n_out + 1);
        for (int64_t s = 0; s < n_out; ++s) {
            const float u = u0 + ((float)s + 0.5f) * du;
            int64_t p = upper_bound_f(cdfs, base, last, u);
            int64_t p0 = p - 1; if (p0 > last) p0 = last; if (p0 < base) p0 = base;
            int64_t p1 = p;     if (p1 > last) p1 = last; if (p1 < base) p1 = base;
            const float ul = cdfs[p0], uh = cdfs[p1], tl = vals[p0], th = vals[p1];
            m[s] = (uh - ul < 1e-10f) ? (tl + th) * 0.5f : (u - ul) * ((th - tl) / (uh - ul)) + tl;
        }
        const float tmin = vals[base], tmax = vals[last];
        for (int64_t s = 0; s < n_out; ++s) {
            if (s == 0) {
                /* pdf.cu:208-213 reads one past the end when n_out == 1; we
                   define that case as a zero half width */
                const float nxt = (n_out > 1) ? m[1] : m[0];
                e[0] = fmaxf(m[0] - (nxt - m[0]) * 0.5f, tmin);
                if (n_out == 1) e[1] = fminf(m[0], tmax);
            } else {
                e[s] = (m[s] + m[s - 1]) * 0.5f;
                if (s == n_out - 1) e[s + 1] = fminf(m[s] + (m[s] - m[s - 1]) * 0.5f, tmax);
            }
        }
    }
}

ORC_API void orc_searchsorted_batched(
    int64_t n_rays, int64_t n_query, const float *query, int64_t n_key, const float *key,
    int64_t *ids_left, int64_t *ids_right)
{
    for (int64_t r = 0; r < n_rays; ++r) {
        const int64_t base = r * n_key, last = base + n_key - 1;
        for (int64_t q = 0; q < n_query; ++q) {
            const int64_t p = upper_bound_f(key, base, last, query[r * n_query + q]);
            int64_t a = p - 1; if (a > last) a = last; if (a < base) a = base;
            int64_t b = p;     if (b > last) b = last; if (b < base) b = base;
            ids_left[r * n_query + q] = a - base;
            ids_right[r * n_query + q] = b - base;
        }
    }
}

/* debugging aid: the voxel sequence of one ray through one level (cells visited by the DDA of
 * traverse_ray for segment [tmin,tmax)): returns the count, fills cx/cy/cz/t_exit/occ[max_n]
 * and the initial tdist/delta/step in state[9] */
ORC_API int64_t orc_debug_cells(const float *o, const float *d, float tmin, float tmax, const float *box,
                                const int32_t *res, const uint8_t *binaries, int64_t max_n,
                                int32_t *cx, int32_t *cy, int32_t *cz, float *t_exit, uint8_t *occ, float *state)
{
    const float inv[3] = {1.0f / d[0], 1.0f / d[1], 1.0f / d[2]};
    const int r3[3] = {res[0], res[1], res[2]};
    dda_t s;
    dda_setup(&s, o, d, inv, tmin, tmax, 1e-6f, box, r3);
    for (int k = 0; k < 3; ++k) { state[k] = s.tdist[k]; state[3 + k] = s.delta[k]; state[6 + k] = (float)s.step[k]; }
    int64_t n = 0;
    for (;;) {
        float t_cell = fminf(s.tdist[0], fminf(s.tdist[1], s.tdist[2]));
        t_cell = fminf(t_cell, tmax);
        if (n < max_n) {
            cx[n] = s.cur[0]; cy[n] = s.cur[1]; cz[n] = s.cur[2]; t_exit[n] = t_cell;
            occ[n] = binaries[((int64_t)s.cur[0] * r3[1] + s.cur[1]) * r3[2] + s.cur[2]];
        }
        ++n;
        if (!dda_advance(&s)) break;
    }
    return n;
}
