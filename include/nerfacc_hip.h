/*
 * nerfacc_hip.h — C ABI of libnerfacc_hip.so, the MI355X (gfx950) implementation of the
 * nerfacc OccGrid sampling + volumetric-rendering hot path.
 *
 * This is the drop-in boundary.  In the reference the same boundary is the pybind11
 * module `nerfacc.csrc` (nerfacc/cuda/csrc/nerfacc.cpp:126-163) whose attributes are looked
 * up lazily by nerfacc/cuda/__init__.py:8-53.  Every entry point below names the reference
 * function it stands in for.  The reference passes torch::Tensor; here everything is a raw
 * device pointer + sizes + a stream, so the library has no torch (or Python) dependency and
 * can be bound from ctypes / cffi / pybind alike (INTEGRATION.md shows the binding).
 *
 * Conventions
 *   - all pointers are DEVICE pointers to contiguous arrays unless marked [host];
 *     float = fp32 values, int64_t = offsets/indices/counts, uint8_t = torch.bool storage;
 *   - `stream` is a hipStream_t passed as void* (NULL = the legacy default stream);
 *     every call only enqueues work on that stream and returns — no call synchronises;
 *   - nothing is allocated: outputs and scratch are caller-provided (sizes are either
 *     stated or returned by the matching *_workspace_bytes function);
 *   - inputs are never written;
 *   - return value: NFA_OK (0) or an NFA_ERR_* code; nfa_last_error() gives the text of the
 *     last failure on the calling thread.  Arguments are validated before anything is launched.
 *   - n == 0 is legal everywhere and enqueues nothing.
 *   - ray_indices / keys must be grouped by ray (each ray's samples contiguous, as produced
 *     by traversal); sortedness is not required except where stated.
 */
#ifndef NERFACC_HIP_H
#define NERFACC_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NFA_OK 0
#define NFA_ERR_INVALID_ARG 1
#define NFA_ERR_LAUNCH 2
#define NFA_ERR_UNSUPPORTED 3

#define NFA_MAX_GRID_LEVELS 8 /* grid.cu:18 */

/* library identification: "nerfacc_hip <version> gfx950" */
const char *nfa_version(void);
/* text of the last error raised on this thread ("" if none) */
const char *nfa_last_error(void);

/* ---------------------------------------------------------------------------------------
 * Options: which FORM of a kernel serves a call (lanes per ray, where the grid image lives, the emit form ...).
 * Every form gives bit-identical results; the defaults ("auto") pick by size and grid kind.  The table is
 * process-wide, reads and writes are atomic (safe against concurrent calls of any entry point), and the NFA_<NAME>
 * environment variables seed it exactly once, when the library is loaded: no entry point reads the environment.
 *   nfa_set_option(name, value)  name: "split_p" or "NFA_SPLIT_P" (case-insensitive); value: the text an environment
 *                                variable would hold; NULL, "" or "auto" = back to automatic.  Unknown names and
 *                                values outside the option's set are NFA_ERR_INVALID_ARG and change nothing.
 *   nfa_get_option               *is_set = 0 while automatic; *value = the forced value (emit: 1 rays, 2 samples)
 *   nfa_reset_options            every option back to its state at load time
 *   nfa_option_count / _name / _doc   enumerate the table (names and one-line descriptions)
 * No reference counterpart (the reference has one form per kernel).
 * ------------------------------------------------------------------------------------- */
int nfa_set_option(const char *name, const char *value);
int nfa_get_option(const char *name, int64_t *value, int32_t *is_set);
void nfa_reset_options(void);
int32_t nfa_option_count(void);
const char *nfa_option_name(int32_t index);
const char *nfa_option_doc(int32_t index);

/* ---------------------------------------------------------------------------------------
 * Grid: ray/AABB test and multi-level occupancy-grid traversal
 * ------------------------------------------------------------------------------------- */

/* replaces ray_aabb_intersect (nerfacc.cpp:63-69, grid.cu:477-519).
 * t_mins, t_maxs: [n_rays, n_aabbs] float; hits: [n_rays, n_aabbs] bool. */
int nfa_ray_aabb_intersect(const float *rays_o, const float *rays_d, int64_t n_rays,
                           const float *aabbs, int64_t n_aabbs, float near_plane, float far_plane,
                           float miss_value, float *t_mins, float *t_maxs, uint8_t *hits, void *stream);

/* Packed occupancy.  The traversal kernels do not read the 1-byte-per-voxel `binaries` tensor
 * (occ_grid.py:72-75) directly: it is first packed into 4x4x4 bricks, one uint64 per brick
 * (bit = (x&3)*16 + (y&3)*4 + (z&3)), bricks x-major like the voxels, followed in the same
 * buffer by a 12-word header (word 0: number of non-empty bricks; words 1..8: occupied voxels of level 0..7 — what
 * `nonzero(binaries[level])` would count, so that the grid update can size it without a host sync), a bitmap of the
 * non-empty bricks, its rank prefix, the compacted non-empty bricks (the form the kernels stage into LDS) and — round 5 — one
 * NIBBLE per brick: the Chebyshev distance, in bricks and within its level, to the nearest non-empty brick, capped at 4 (what the
 * empty-space macro steps of the lane-per-ray count pass size their jumps from; written by nfa_pack_binaries /
 * nfa_grid_threshold_packed with the rest).
 * nfa_packed_grid_words: size of that buffer in uint64 words for [n_grids, rx, ry, rz] — ALWAYS size the buffer with this call
 * (the layout grew in round 5; a buffer sized by an older formula is too small). */
int64_t nfa_packed_grid_words(int32_t n_grids, int32_t rx, int32_t ry, int32_t rz);
int nfa_pack_binaries(const uint8_t *binaries, int32_t n_grids, int32_t rx, int32_t ry, int32_t rz,
                      uint64_t *bricks, void *stream);

/* Occupancy-grid maintenance, the device side of OccGridEstimator._update (occ_grid.py:366-404;
 * pure torch in the reference).  Random numbers are the caller's (torch generator, reference
 * call order); everything is asynchronous on `stream`.
 * nfa_grid_cell_points   :377-384  points[i] = lo + ((coords(id_i) + jitter[i]) / res) * (hi - lo);
 *                        cell_ids nullable = cells 0..n-1; ids are level-local, x-major; `aabb` is
 *                        the level's 6 floats in DEVICE memory.
 * nfa_grid_ema_update    :388-390  occs[id_i] = max(occs[id_i] * ema_decay, occ_new[i]), every
 *                        candidate formed from the old grid; with repeated ids one candidate wins
 *                        (as index_put_).  `occs` points at the level's first cell; scratch: n floats.
 * nfa_grid_threshold     :392-404  binaries[c] = occs[c] > min(mean(occs[occs >= 0]), occ_thre) over
 *                        all levels; threshold_out (nullable, device) receives the threshold. */
int nfa_grid_cell_points(const int64_t *cell_ids, int64_t n, const float *jitter,
                         int32_t rx, int32_t ry, int32_t rz, const float *aabb, float *points, void *stream);
int nfa_grid_ema_update(float *occs, const int64_t *cell_ids, int64_t n, const float *occ_new,
                        float ema_decay, float *scratch, void *stream);
/* nfa_grid_mark_invisible :262-332  occs[id_i] = (some camera sees the cell at depth >= near_plane and none sees it nearer) ? 0 : -1
 *                        for the cells of ONE level (`occs` points at the level's first cell, `aabb` = its 6 floats, ids
 *                        level-local, nullable = cells 0..n-1).  w2c_R [n_cams,9] / w2c_T [n_cams,3]: world-to-camera rotation
 *                        and translation (row-major); K [n_cams,9], or [1,9] shared by all cameras when k_shared != 0. */
int nfa_grid_mark_invisible(float *occs, const int64_t *cell_ids, int64_t n, int32_t rx, int32_t ry, int32_t rz,
                            const float *aabb, const float *w2c_R, const float *w2c_T, const float *K,
                            int32_t n_cams, int32_t k_shared, float width, float height, float near_plane, void *stream);
int64_t nfa_grid_threshold_workspace_bytes(void);
int nfa_grid_threshold(const float *occs, int64_t n_cells, float occ_thre, void *workspace,
                       uint8_t *binaries, float *threshold_out, void *stream);
/* nfa_grid_threshold followed by nfa_pack_binaries, fused (the comparison pass writes the bool grid AND the bricks:
 * four launches instead of five, and the bool grid is not read back).  occs: float[n_grids * rx * ry * rz];
 * binaries: the same number of bytes; bricks: nfa_packed_grid_words(...) words; workspace as nfa_grid_threshold. */
int nfa_grid_threshold_packed(const float *occs, int32_t n_grids, int32_t rx, int32_t ry, int32_t rz, float occ_thre,
                              void *workspace, uint8_t *binaries, float *threshold_out, uint64_t *bricks, void *stream);
/* The occupied cells of one level in ascending order — `torch.nonzero(binaries[lvl].flatten())[:, 0]` of
 * OccGridEstimator._sample_uniform_and_occupied_cells (occ_grid.py:356) — in one launch: out[k] = the k-th flat index c with
 * cells[c] != 0.  cells: the level's n_cells bool bytes (0 / 1, 16-byte aligned); capacity: entries `out` holds (the count is in the
 * packed grid's header, nfa_pack_binaries: no read-back on this path; indices beyond it are not stored).  sync: NFA_SYNC_BYTES of
 * zeroed device memory, left zero (see nfa_traverse_sample).  Up to 2^27 cells per call. */
int nfa_grid_occupied_cells(const uint8_t *cells, int64_t n_cells, int64_t *out, int64_t capacity, void *sync, void *stream);

/* Arguments of traverse_grids (nerfacc.cpp:71-98, grid.cu:320-474).  The reference does
 * count -> cumsum + .item() -> allocate -> fill inside one C++ call; a C ABI cannot allocate
 * torch tensors, so the two halves are separate calls and the caller allocates in between
 * from `totals`.  Pointers documented "nullable" may be NULL. */
typedef struct nfa_traverse_args {
    /* rays */
    int64_t n_rays;
    const float *rays_o;        /* [n_rays, 3] */
    const float *rays_d;        /* [n_rays, 3] */
    const uint8_t *rays_mask;   /* [n_rays] nullable: rays with 0 are skipped (grid.cu:100) */
    /* grids */
    int32_t n_grids;            /* <= NFA_MAX_GRID_LEVELS */
    int32_t res[3];
    const uint64_t *bricks;     /* from nfa_pack_binaries */
    int64_t n_nonempty_bricks;  /* bricks[n_bricks] (the header word nfa_pack_binaries writes) if the
                                   caller has read it back, else -1.  When given it MUST be that value:
                                   it sizes the kernels' LDS image of the grid; with -1 (or a grid too
                                   large for LDS) the kernels read the packed grid from L2 instead */
    const float *aabbs;         /* [n_grids, 6] */
    /* sorted ray/grid intersections (grid.py:156-162).  All three nullable together: when
     * NULL the kernel runs the slab test and the per-ray sort of the 2*n_grids events itself. */
    const uint8_t *hits;        /* [n_rays, n_grids] */
    const float *t_sorted;      /* [n_rays, 2*n_grids] */
    const int64_t *t_indices;   /* [n_rays, 2*n_grids] */
    /* options */
    const float *near_planes;   /* [n_rays]; nullable: then near_plane (scalar, below) is every ray's near plane */
    const float *far_planes;    /* [n_rays]; nullable: then far_plane */
    float step_size;
    float cone_angle;
    int32_t traverse_steps_limit; /* <= 0: unlimited */
    /* per-ray counts / offsets: written by nfa_traverse_count, read by nfa_traverse_fill.
     * iv_* describe interval edges (RaySegmentsSpec intervals), sm_* samples. iv pair nullable. */
    int64_t *iv_cnts, *iv_starts; /* [n_rays] */
    int64_t *sm_cnts, *sm_starts; /* [n_rays] */
    int64_t *totals;            /* [4] = {n_edges (counted also when no interval outputs are asked for), n_samples, n_overflow_rays, 0}; device-visible (pinned host ok) */
    /* fill outputs, each nullable (grid.cu:219-255) */
    float *iv_vals; int64_t *iv_ray_indices; uint8_t *iv_is_left; uint8_t *iv_is_right; /* [n_edges]; masks pre-zeroed */
    float *sm_vals; int64_t *sm_ray_indices; uint8_t *sm_is_valid;                      /* [n_samples] */
    float *t_starts; float *t_ends; /* [n_samples]: interval of each sample, written directly
                                       (what occ_grid.py:174-175 extracts with is_left/is_right) */
    float *terminate_planes;    /* [n_rays] nullable */
    /* per-ray planes formed in the kernel exactly as OccGridEstimator.sampling forms them with torch ops
     * (occ_grid.py:154-163; same float operations, same order): near = [near_planes or near_plane],
     * clamped from below by t_min, plus jitter * jitter_scale (stratified: rand * render_step_size);
     * far = [far_planes or far_plane] clamped from above by t_max.  All three pointers nullable. */
    float near_plane, far_plane;
    const float *t_min, *t_max; /* [n_rays] */
    const float *jitter;        /* [n_rays] uniform [0,1) numbers of the caller's generator */
    float jitter_scale;
    /* bytes of the workspace handed to nfa_traverse_count / _offsets / _fill / _emit_speculative with these args.  0 (or
     * anything below nfa_traverse_workspace_bytes_for(args)) = the caller sized it with nfa_traverse_workspace_bytes(n_rays):
     * count passes that need the larger workspace (cone_angle != 0: one lane per level segment + a serial chain over
     * per-voxel records) are then not selected. */
    int64_t workspace_bytes;
} nfa_traverse_args;

/* pass 1 (grid.cu:413), one kernel: per-ray counts into iv_cnts / sm_cnts, terminate_planes when
 * given, per-workgroup partial sums and each ray's samples as run-length records (lattice start,
 * length) in `workspace` — the records let pass 2 emit samples without touching the grid again.
 * workspace: nfa_traverse_workspace_bytes(n_rays) bytes of device scratch. */
int64_t nfa_traverse_workspace_bytes(int64_t n_rays);
/* the workspace these args can make use of (>= nfa_traverse_workspace_bytes(args->n_rays)); pass the size actually
 * allocated in args->workspace_bytes.  Reads n_rays, n_grids, res, step_size, cone_angle, traverse_steps_limit and which of
 * rays_mask / t_sorted are given. */
int64_t nfa_traverse_workspace_bytes_for(const nfa_traverse_args *args);
int nfa_traverse_count(const nfa_traverse_args *args, void *workspace, void *stream);
/* pass 1b (the cumsum of data_spec.hpp:90), one kernel: iv_starts / sm_starts = exclusive sums of
 * the counts, totals = {n_edges, n_samples, rays whose runs did not fit (re-traversed by pass 2), 0}.
 * Same args and workspace as the nfa_traverse_count call it follows. */
int nfa_traverse_offsets(const nfa_traverse_args *args, const void *workspace, void *stream);
/* The same; totals[3] receives `stamp` (the plain call stores 0 there), last and behind a system-scope fence: a host that
 * polls that word in coherent pinned memory may read the totals as soon as it sees the stamp. */
int nfa_traverse_offsets_stamped(const nfa_traverse_args *a, const void *workspace, int64_t stamp, void *stream);
/* pass 2 (grid.cu:445 / the single over-allocated pass :375): write edges / samples at
 * iv_starts / sm_starts.
 * workspace != NULL: the workspace nfa_traverse_count AND nfa_traverse_offsets filled for the SAME args, with
 *   n_samples = totals[1] and n_overflow = totals[2] as read back by the caller; one lane per
 *   output sample or 16 lanes per ray, chosen on the device from the totals in the workspace
 *   (skip_empty / rewrite_counts must be 1 / 0, no rays_mask).
 * workspace == NULL: the grid is traversed again, one lane per ray (n_samples / n_overflow
 *   ignored).  `skip_empty`: skip rays whose stored count is 0 (grid.cu:103-106).
 *   `rewrite_counts`: store the actual per-ray counts back (over-allocated mode, grid.cu:277-280). */
int nfa_traverse_fill(const nfa_traverse_args *args, int32_t skip_empty, int32_t rewrite_counts,
                      const void *workspace, int64_t n_samples, int64_t n_overflow, void *stream);

/* The emit pass of the two-pass mode launched BEFORE the host has read the totals back (sampling outputs only): the
 * outputs hold `capacity` samples — the caller's guess — and the kernel takes the true total from the device copy that
 * nfa_traverse_offsets left in `workspace`; if the total exceeds the capacity the launch does nothing and the caller
 * calls nfa_traverse_fill with exactly sized outputs after its read-back.  Rays flagged as overflowed by the count
 * pass are NOT written here: call nfa_traverse_fill(a, 1, 0, workspace, 0, n_overflow, stream) for them. */
int nfa_traverse_emit_speculative(const nfa_traverse_args *a, const void *workspace, int64_t capacity, void *stream);

/* Sync block of the single-launch forms below (nfa_traverse_sample, nfa_visibility_compact_sync): NFA_SYNC_BYTES of device memory that
 * is ZERO before the first call that is handed it and is used by the calls of ONE stream at a time; every kernel that uses it leaves it
 * zero again (a caller allocates it once per stream, zero-filled, and keeps it).  Inside a launch it carries one 64-bit state per
 * workgroup [status | value]: a workgroup publishes the sum of its part and looks back over the workgroups before it — the hand-off
 * that otherwise takes a kernel boundary (the reference: cumsum + .item() between its two traversal passes, data_spec.hpp:86-96).
 * Every wait is bounded (2 ms): a launch that cannot finish its look-back says so in its result and the caller goes on with the
 * separate kernels; results are identical either way. */
#define NFA_SYNC_BYTES 16384

/* nfa_traverse_count + nfa_traverse_offsets_stamped + nfa_traverse_emit_speculative (sampling outputs: sm_* / t_starts / t_ends, no
 * interval outputs) — as ONE launch when `sync` is given and the call has a fused form (nfa_traverse_sample_fused(args) != 0: one
 * level, step_size > 0, cone_angle = 0, 3072 ... 8192 rays, a grid whose sparse image fits LDS, n_nonempty_bricks given), else as the
 * three launches in order (also when `capacity` is more than 80 samples per ray: with rays that long the emit kernel's spread over the
 * chip beats the launch's one wave per four rays).  `capacity`: samples the outputs hold (the caller's guess; 0 = count and offsets only); rays flagged as
 * overflowed are not written (nfa_traverse_fill(a, 1, 0, workspace, 0, n_overflow, stream), as after nfa_traverse_emit_speculative).
 * totals / stamp as nfa_traverse_offsets_stamped; after a fused launch totals[1] == -1 means its look-back gave up: per-ray counts,
 * run records and wave sums are complete — call nfa_traverse_offsets[_stamped] and go on as after nfa_traverse_count.  With
 * totals[1] > capacity nothing useful was stored: call nfa_traverse_fill with outputs of the right size.
 * *fused (host, nullable) = 1 when the single launch was taken. */
int nfa_traverse_sample_fused(const nfa_traverse_args *args);
int nfa_traverse_sample(const nfa_traverse_args *args, void *workspace, int64_t capacity, int64_t stamp, void *sync, int32_t *fused,
                        void *stream);

/* chunk_starts = cumsum(cnts) - cnts, total -> *total (data_spec.hpp:86-106). total nullable. */
int nfa_exclusive_sum_i64(const int64_t *cnts, int64_t n, int64_t *starts, int64_t *total, void *stream);

/* ---------------------------------------------------------------------------------------
 * pack_info (pack.py:10-49): ray_indices [n] (any order; ascending input takes the one-launch path) -> packed_info [n_rays, 2]
 * ------------------------------------------------------------------------------------- */
int nfa_pack_info(const int64_t *ray_indices, int64_t n, int64_t n_rays, int64_t *packed_info, void *stream);
/* inverse: packed_info -> ray_indices [n]; elements outside every chunk get -1 */
int nfa_unpack_info(const int64_t *chunk_starts, const int64_t *chunk_cnts, int64_t n_rays,
                    int64_t *ray_indices, int64_t n, void *stream);

/* ---------------------------------------------------------------------------------------
 * Segmented scans (scan.cu, scan_cub.cu).  op: 0 sum, 1 product.
 * ------------------------------------------------------------------------------------- */
#define NFA_OP_SUM 0
#define NFA_OP_PROD 1

/* chunked by (chunk_starts, chunk_cnts): inclusive_sum / exclusive_sum /
 * {inclusive,exclusive}_prod_forward (nerfacc.cpp:17-37, scan.cu:9-126,128-166,215-257).
 * `reverse` scans each chunk from its last element (= the reference's backward=true).
 * `normalize` divides by the chunk total (utils_scan.cuh:101-109). */
int nfa_scan_packed(const int64_t *chunk_starts, const int64_t *chunk_cnts, int64_t n_rays,
                    const float *inputs, float *outputs, int64_t n,
                    int32_t op, int32_t inclusive, int32_t reverse, int32_t normalize, void *stream);

/* keyed by ray index: {inclusive,exclusive}_sum_cub, {inclusive,exclusive}_prod_cub_forward
 * (nerfacc.cpp:40-60, scan_cub.cu:66-182,220-250). */
int nfa_scan_keyed(const int64_t *keys, const float *inputs, float *outputs, int64_t n,
                   int32_t op, int32_t inclusive, int32_t reverse, void *stream);

/* {inclusive,exclusive}_prod(_cub)_backward (scan.cu:169-214,259-304; scan_cub.cu:184-218,
 * 252-287): grad_in = revscan(grad_out * out) / clamp_min(in, 1e-10), fused in one kernel.
 * Exactly one of (keys) / (chunk_starts, chunk_cnts) is given. */
int nfa_prod_backward(const int64_t *keys, const int64_t *chunk_starts, const int64_t *chunk_cnts,
                      int64_t n_rays, const float *inputs, const float *outputs,
                      const float *grad_outputs, float *grad_inputs, int64_t n, int32_t inclusive,
                      void *stream);

/* ---------------------------------------------------------------------------------------
 * Fused volumetric rendering (volrend.py; pure torch + one scan in the reference)
 * ------------------------------------------------------------------------------------- */

/* render_weight_from_density / render_transmittance_from_density (volrend.py:219-278,326-376):
 * sd = sigma*(t_end-t_start); alpha = 1-exp(-sd); trans = exp(-excl_sum(sd)) [* prefix_trans];
 * weight = trans*alpha.  Outputs nullable individually. */
int nfa_render_weight_from_density_fwd(const int64_t *ray_indices, const float *t_starts,
                                       const float *t_ends, const float *sigmas,
                                       const float *prefix_trans /* nullable */, int64_t n,
                                       float *weights, float *trans, float *alphas, void *stream);
/* its vector-Jacobian product w.r.t. sigmas; any of g_weights/g_trans/g_alphas nullable.
 * trans / alphas are the forward outputs. */
int nfa_render_weight_from_density_bwd(const int64_t *ray_indices, const float *t_starts,
                                       const float *t_ends, const float *sigmas, const float *trans,
                                       const float *alphas, const float *g_weights,
                                       const float *g_trans, const float *g_alphas, int64_t n,
                                       float *g_sigmas, void *stream);

/* Sample midpoints in world space, positions[i] = rays_o[r_i] + rays_d[r_i] * ((t_starts[i] + t_ends[i]) / 2)
 * with r_i = ray_indices[i]: what the reference's examples compute with six torch ops at the top of
 * every sigma_fn / rgb_sigma_fn (examples/utils.py:96-101, 105-118).  Same float operation order
 * as that expression.  dirs (nullable, [n,3]) receives rays_d[r_i]. */
int nfa_sample_positions(const float *rays_o, const float *rays_d, int64_t n_rays,
                         const int64_t *ray_indices, const float *t_starts, const float *t_ends, int64_t n,
                         float *positions, float *dirs, void *stream);

/* render_visibility_from_density + the three mask compactions of OccGridEstimator.sampling
 * (occ_grid.py:194-220, volrend.py:435-494) in one go: keep sample i iff
 * trans_i >= early_stop_eps and (alpha_thre <= 0 or alpha_i >= alpha_thre).
 * Outputs are compacted into the first *n_out entries of out_* (capacity n each).
 * workspace: nfa_visibility_workspace_bytes(n) bytes of scratch (uninitialised is fine; 16-byte aligned; private to the call
 * until the stream has passed it): keep / head bit planes between the two kernels (0.5 byte per sample), per-tile counts and
 * ranges.  out_mask (bool bytes) is written only when given.  n_out: [1], device-visible.
 * ray_indices grouped by ray (a ray = a run of equal keys, as everywhere on this path): the compaction reads the key at the
 * head of each run and hands it down the run. */
int64_t nfa_visibility_workspace_bytes(int64_t n);
int nfa_visibility_compact(const int64_t *ray_indices, const float *t_starts, const float *t_ends,
                           const float *sigmas /* or alphas when from_alpha */, int32_t from_alpha,
                           int64_t n, float early_stop_eps, float alpha_thre,
                           int64_t *out_ray_indices, float *out_t_starts, float *out_t_ends,
                           uint8_t *out_mask /* [n] nullable */, int64_t *n_out, void *workspace,
                           void *stream);
/* The same with a completion stamp for callers that poll instead of synchronising the stream: n_out is [2] in coherent pinned
 * host memory and the kernel stores `stamp` (non-zero) into n_out[1] right after the count, behind a system-scope fence — the
 * host may read n_out[0] as soon as it sees the stamp, while the compaction itself is still running (inputs of up to
 * 2^18 wave tiles — about 3e8 samples; beyond that the total is written by a separate scan kernel and the stamp never comes:
 * synchronise the stream instead). */
int nfa_visibility_compact_stamped(const int64_t *ray_indices, const float *t_starts, const float *t_ends,
                                   const float *sigmas, int32_t from_alpha, int64_t n, float early_stop_eps, float alpha_thre,
                                   int64_t *out_ray_indices, float *out_t_starts, float *out_t_ends, uint8_t *out_mask,
                                   int64_t *n_out, int64_t stamp, void *workspace, void *stream);

/* nfa_visibility_compact_stamped as ONE launch when `sync` (NFA_SYNC_BYTES, see nfa_traverse_sample) is given, option `fused_vis`
 * is 1 and the call is small enough for every workgroup to be resident at once (up to ~3 workgroups per CU): one pass over the
 * samples, the survivors of a tile staged in LDS, a workgroup sums the survivors of the workgroups before it by look-back and
 * flushes — the hand-off that otherwise takes a kernel boundary.  Opt-in: on MI355X it measures slower than the two kernels
 * (profiles/r06_small_n.md).  Everything else takes the kernels of nfa_visibility_compact_stamped.  n_out[0] == -1 (behind the stamp)
 * says the look-back gave up (bounded wait): nothing was stored — call nfa_visibility_compact_resume with the same arguments (it
 * runs the kernels of nfa_visibility_compact_stamped).  Results are identical. */
int nfa_visibility_compact_fused(int64_t n, int32_t aligned16_ptrs);     /* 1: the call below takes the single launch (given `sync` and compacted outputs) */
int nfa_visibility_compact_sync(const int64_t *ray_indices, const float *t_starts, const float *t_ends,
                                const float *sigmas, int32_t from_alpha, int64_t n, float early_stop_eps, float alpha_thre,
                                int64_t *out_ray_indices, float *out_t_starts, float *out_t_ends, uint8_t *out_mask,
                                int64_t *n_out, int64_t stamp, void *workspace, void *sync, void *stream);
int nfa_visibility_compact_resume(const int64_t *ray_indices, const float *t_starts, const float *t_ends,
                                  const float *sigmas, int32_t from_alpha, int64_t n, float early_stop_eps, float alpha_thre,
                                  int64_t *out_ray_indices, float *out_t_starts, float *out_t_ends, uint8_t *out_mask,
                                  int64_t *n_out, int64_t stamp, void *workspace, void *stream);

/* accumulate_along_rays / accumulate_along_rays_ (volrend.py:497-587):
 * outputs[r, :] += sum_{i in r} w_i * values[i, :]  (values NULL: D = 1, values = 1). */
int nfa_accumulate_along_rays(const int64_t *ray_indices, const float *weights,
                              const float *values /* [n, D] nullable */, int64_t n, int32_t D,
                              int64_t n_rays, float *outputs /* [n_rays, D] */, void *stream);
/* VJP: g_weights [n] and/or g_values [n, D] (either nullable) from g_outputs [n_rays, D]; samples whose
 * ray index lies outside [0, n_rays) were skipped by the forward pass and get a zero gradient */
int nfa_accumulate_along_rays_bwd(const int64_t *ray_indices, const float *weights,
                                  const float *values, const float *g_outputs, int64_t n, int32_t D,
                                  int64_t n_rays, float *g_weights, float *g_values, void *stream);

/* rendering() after rgb_sigma_fn (volrend.py:104-164), one kernel: weights/trans/alphas [n],
 * colors [n_rays,3], opacities [n_rays,1], depths [n_rays,1] incl. depth normalisation
 * (expected_depths) and background blend (bkgd [3], nullable).  ray_indices sorted. */
int nfa_rendering_fwd(const int64_t *ray_indices, const float *t_starts, const float *t_ends,
                      const float *sigmas, const float *rgbs /* [n,3] */, int64_t n, int64_t n_rays,
                      const float *bkgd, int32_t expected_depths, float *weights, float *trans,
                      float *alphas, float *colors, float *opacities, float *depths, void *stream);
/* VJP of the above w.r.t. sigmas and rgbs.  g_colors/g_opacities/g_depths per ray,
 * g_weights/g_trans/g_alphas per sample; all six nullable.  opacities/depths: forward outputs.
 * trans / alphas must be the forward pass's outputs for the same inputs: the kernel forms the weights as trans * alphas (exactly
 * what the forward pass stored).  `weights` (and `sigmas`) are NOT read and may be NULL — the parameters stay for the callers written
 * against rounds 1-4; passing weights that are not trans * alphas of the same forward pass changes nothing. */
int nfa_rendering_bwd(const int64_t *ray_indices, const float *t_starts, const float *t_ends,
                      const float *sigmas, const float *rgbs, const float *weights,
                      const float *trans, const float *alphas, const float *opacities,
                      const float *depths, int64_t n, int64_t n_rays, const float *bkgd,
                      int32_t expected_depths, const float *g_colors, const float *g_opacities,
                      const float *g_depths, const float *g_weights, const float *g_trans,
                      const float *g_alphas, float *g_sigmas, float *g_rgbs, void *stream);

/* ---------------------------------------------------------------------------------------
 * pdf (pdf.cu): importance_sampling (int overload, nerfacc.cpp:100-112) and searchsorted
 * (nerfacc.cpp:114-117).
 * ------------------------------------------------------------------------------------- */
/* device view of a RaySegmentsSpec (data_spec.hpp:6-14, data_spec_packed.cuh:10-41):
 * batched [n_rays, n_edges_per_ray] when chunk_starts == NULL, else flattened [n_edges] with
 * per-ray (chunk_starts, chunk_cnts) and optional ray_indices. */
typedef struct nfa_ray_segments {
    const float *vals;
    const int64_t *chunk_starts; /* [n_rays] nullable => batched */
    const int64_t *chunk_cnts;   /* [n_rays] */
    const int64_t *ray_indices;  /* [n_edges] nullable */
    int64_t n_edges;             /* total number of values */
    int64_t n_rays;
    int64_t n_edges_per_ray;     /* batched only */
} nfa_ray_segments;

/* Resample every ray to n_intervals intervals by inverting its cdf (pdf.cu:98-241).
 * Outputs are batched: out_mids [n_rays, n_intervals], out_edges [n_rays, n_intervals+1].
 * jitter: per-ray offsets in [0,1) (nullable => 0.5, i.e. stratified = false).  The reference
 * draws one Philox uniform per ray inside the kernel (pdf.cu:138-144); here the caller's
 * generator produces them, which keeps the library free of RNG state. */
int nfa_importance_sampling(const nfa_ray_segments *segments, const float *cdfs, int64_t n_intervals,
                            const float *jitter, float *out_edges, float *out_mids, void *stream);
/* The same with a PER-RAY number of intervals — the Tensor overload of the reference's importance_sampling (nerfacc.cpp:100-105,
 * pdf.cu:294-357; its own implementation allocates no output elements, so this follows what its kernels state: pdf.cu:112-116,
 * 207-239).  Flattened outputs: ray r has sm_cnts[r] samples from sm_starts[r] (= exclusive sum of sm_cnts) and, when sm_cnts[r] > 0,
 * sm_cnts[r] + 1 edges from iv_starts[r] (= exclusive sum of (cnt + 1) * (cnt > 0)); n_samples = sum of sm_cnts.  iv_is_left is set on
 * all but a ray's last edge, iv_is_right on all but its first.  Index / flag outputs are nullable.  A ray's values equal those of
 * nfa_importance_sampling with n_intervals = sm_cnts[r]. */
int nfa_importance_sampling_ragged(const nfa_ray_segments *segments, const float *cdfs, const int64_t *sm_starts,
                                   const int64_t *sm_cnts, const int64_t *iv_starts, int64_t n_samples, const float *jitter,
                                   float *sm_vals, int64_t *sm_ray_indices, float *iv_vals, int64_t *iv_ray_indices,
                                   uint8_t *iv_is_left, uint8_t *iv_is_right, void *stream);
/* For every query value find (left, right) with key[left] <= q < key[right] in the same ray's
 * key edges, clamped to the ray (pdf.cu:245-286).  ids are relative to the ray for a batched
 * query and absolute positions in key.vals for a flattened one, as in the reference. */
int nfa_searchsorted(const nfa_ray_segments *query, const nfa_ray_segments *key,
                     int64_t *ids_left, int64_t *ids_right, void *stream);
/* PropNetEstimator's map from normalised s in [0,1] to ray distance (prop_net.py:215-229, `_transform_stot`):
 * uniform (lindisp = 0): s t_max + (1 - s) t_min;  lindisp: 1 / (s / t_max + (1 - s) / t_min) — the reference's float
 * operations in the reference's order, one launch.  s_vals, t_vals: [n].  t_min / t_max are doubles, as the Python floats the
 * reference divides (`1.0 / t` in double, rounded to float once). */
int nfa_transform_stot(const float *s_vals, int64_t n, double t_min, double t_max, int32_t lindisp, float *t_vals, void *stream);
/* cdf at the n_samples + 1 edges of a proposal level from its densities, batched layout (prop_net.py:99-112:
 * render_transmittance_from_density, then 1 - cat([trans, 0])): cdfs[r, j] = 1 - exp(-sum_{i<j} sigma_i (t_{i+1} - t_i)) for
 * j < n_samples, cdfs[r, n_samples] = 1.  t_edges, cdfs: [n_rays, n_samples + 1]; sigmas, trans (nullable; what the backward
 * reads): [n_rays, n_samples].  The backward returns d loss / d sigmas from d loss / d cdfs. */
int nfa_edge_cdfs_fwd(const float *t_edges, const float *sigmas, int64_t n_rays, int64_t n_samples, float *cdfs, float *trans,
                      void *stream);
int nfa_edge_cdfs_bwd(const float *t_edges, const float *trans, const float *g_cdfs, int64_t n_rays, int64_t n_samples,
                      float *g_sigmas, void *stream);
/* Histogram-envelope loss of a proposal level against the final samples, batched layout (prop_net.py:232-256, `_pdf_loss`):
 * loss[r, i] = clip(w - w_outer, 0)^2 / (w + eps) with w = cdfs_query[r, i + 1] - cdfs_query[r, i] and w_outer the cdf mass of
 * the key intervals overlapping query interval i (searchsorted of the query edges in the key edges, pdf.cu:245-286).
 * query_vals, cdfs_query: [n_rays, n_query + 1]; key_vals, cdfs_key: [n_rays, n_key + 1]; loss: [n_rays, n_query].
 * ids_left, ids_right (int32, indices into the key row) and coef (d loss / d w_outer): [n_rays, n_query], all three or none —
 * the backward's inputs; it returns d loss / d cdfs_key [n_rays, n_key + 1] (deterministic: no atomics).  Rays whose query edges ascend
 * (as the edges importance_sampling returns do) find each key edge's query intervals by bisection; a ray whose ids do not ascend
 * (unsorted or NaN query edges) scans all of its intervals per edge — the same gradient as the reference's gather backward. */
int nfa_pdf_loss_fwd(const float *query_vals, const float *cdfs_query, const float *key_vals, const float *cdfs_key,
                     int64_t n_rays, int64_t n_query, int64_t n_key, float eps, float *loss, int32_t *ids_left,
                     int32_t *ids_right, float *coef, void *stream);
int nfa_pdf_loss_bwd(const float *g_loss, const int32_t *ids_left, const int32_t *ids_right, const float *coef,
                     int64_t n_rays, int64_t n_query, int64_t n_key, float *g_cdfs_key, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* NERFACC_HIP_H */
