// grid.hip — ray/AABB test, occupancy-brick packing and multi-level grid traversal for gfx950.
//
// Replaces nerfacc/cuda/csrc/grid.cu (+ include/utils_grid.cuh) behind the C ABI of
// include/nerfacc_hip.h.  Semantics are those restated in oracle/nerfacc_oracle.c; the
// float-op order and the explicit fmaf() sites are identical on both sides so that per-ray
// sample counts agree bit for bit (this file is compiled with -ffp-contract=off).
//
// MI355X mapping (DESIGN.md section "traversal"):
//   * the 1-byte-per-voxel grid is repacked into 4x4x4 bricks of one uint64 (8x smaller:
//     128^3 -> 256 KiB, L2-resident on every XCD); a ray keeps its current brick in two
//     VGPRs, so a global load happens once per brick crossed instead of once per voxel;
//   * one lane per ray, 256-thread workgroups, rays block-contiguous so that the per-ray
//     counts of a wave are 64 consecutive int64 (coalesced) and a wave's samples form one
//     contiguous output range;
//   * counting and packing: pass 1 block-reduces its counts (wave shuffles + LDS), a second
//     tiny kernel turns block sums into every ray's offset (exclusive sum) and the totals;
//     pass 2 re-walks and writes.  One 16-byte readback per traverse_grids call.
#include "common.hpp"

namespace nfa {

char *last_error_buffer() {
    static thread_local char buf[512] = {0};
    return buf;
}

namespace {

// ----------------------------------------------------------------------------------------
// shared device math
// ----------------------------------------------------------------------------------------
__device__ __forceinline__ float march_dt(float t, float cone_angle, float dt_min) {
    return fminf(fmaxf(t * cone_angle, dt_min), 1e10f);   // grid.cu:23-28
}

// slab test, utils_grid.cuh:10-55
__device__ __forceinline__ bool slab_test(const float o[3], const float inv[3], const float *__restrict__ box,
                                          float near, float far, float &t0, float &t1) {
    float a0, a1;
#pragma unroll
    for (int ax = 0; ax < 3; ++ax) {
        const float lo = box[ax], hi = box[3 + ax];
        float a, b;
        if (inv[ax] >= 0) { a = (lo - o[ax]) * inv[ax]; b = (hi - o[ax]) * inv[ax]; }
        else              { a = (hi - o[ax]) * inv[ax]; b = (lo - o[ax]) * inv[ax]; }
        if (ax == 0) { a0 = a; a1 = b; continue; }
        if (a0 > b || a > a1) return false;
        if (a > a0) a0 = a;
        if (b < a1) a1 = b;
    }
    if (a1 <= 0) return false;
    t0 = fmaxf(a0, near);
    t1 = fminf(a1, far);
    return true;
}

__device__ __forceinline__ int f2i(float x) { return (int)x; }  // v_cvt_i32_f32: trunc, saturating, NaN -> 0
__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

struct Dda {
    float tx, ty, tz;   // t at which the ray crosses the next x / y / z voxel plane
    float dx, dy, dz;   // t between successive planes per axis
    int sx, sy, sz;     // index step per axis (-1, 0, +1)
    int cx, cy, cz;     // current voxel
    int ox, oy, oz;     // first out-of-segment index per axis (final + step)
};

__device__ __forceinline__ void dda_axis(float o, float d, float inv, float lo, float hi, int res,
                                         float tmin, float tmax, float t_in, float t_out,
                                         float &tdist, float &delta, int &step, int &cur, int &overflow) {
    const float resf = (float)res;
    const float vox = (hi - lo) / resf;
    const float p_in = fmaf(d, t_in, o);
    const float p_out = fmaf(d, t_out, o);
    cur = clampi(f2i(((p_in - lo) / (hi - lo)) * resf), 0, res - 1);
    const int fin = clampi(f2i(((p_out - lo) / (hi - lo)) * resf), 0, res - 1);
    const int first_plane = cur + (d > 0 ? 1 : 0);
    const float inner = fmaf((float)first_plane, vox, -p_in);
    const float t_plane = fmaf(lo + inner, inv, tmin);
    const float sgn = (d == 0.0f) ? 0.0f : (d > 0.0f ? 1.0f : -1.0f);
    step = (int)sgn;
    tdist = (d == 0.0f) ? tmax : t_plane;
    delta = (d == 0.0f) ? tmax : (vox * inv) * sgn;
    overflow = fin + step;
}

// utils_grid.cuh:58-114
__device__ __forceinline__ void dda_setup(Dda &s, const float o[3], const float d[3], const float inv[3],
                                          float tmin, float tmax, const float *__restrict__ box,
                                          const int res[3]) {
    const float eps = 1e-6f;
    const float t_in = tmin + eps, t_out = tmax - eps;
    dda_axis(o[0], d[0], inv[0], box[0], box[3], res[0], tmin, tmax, t_in, t_out, s.tx, s.dx, s.sx, s.cx, s.ox);
    dda_axis(o[1], d[1], inv[1], box[1], box[4], res[1], tmin, tmax, t_in, t_out, s.ty, s.dy, s.sy, s.cy, s.oy);
    dda_axis(o[2], d[2], inv[2], box[2], box[5], res[2], tmin, tmax, t_in, t_out, s.tz, s.dz, s.sz, s.cz, s.oz);
}

// utils_grid.cuh:116-142
__device__ __forceinline__ bool dda_advance(Dda &s) {
    if (s.tx < s.ty && s.tx < s.tz) { s.cx += s.sx; s.tx += s.dx; return s.cx != s.ox; }
    if (s.ty < s.tz)                { s.cy += s.sy; s.ty += s.dy; return s.cy != s.oy; }
    s.cz += s.sz; s.tz += s.dz; return s.cz != s.oz;
}

// grid.cu:157-161 / 199-203 (with the t + dt == t escape, see oracle)
__device__ __forceinline__ float lattice_skip(float t, float dt, float target) {
    while (t + dt * 0.5f < target) {
        const float nt = t + dt;
        if (nt == t) return target;
        t = nt;
    }
    return t;
}

// ----------------------------------------------------------------------------------------
// K1
// ----------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void ray_aabb_kernel(
    const float *__restrict__ rays_o, const float *__restrict__ rays_d, int64_t n_rays,
    const float *__restrict__ aabbs, int64_t n_aabbs, float near, float far, float miss,
    float *__restrict__ t_mins, float *__restrict__ t_maxs, uint8_t *__restrict__ hits)
{
    const int64_t total = n_rays * n_aabbs;
    for (int64_t k = (int64_t)blockIdx.x * kBlock + threadIdx.x; k < total; k += (int64_t)gridDim.x * kBlock) {
        const int64_t r = k / n_aabbs, g = k - r * n_aabbs;
        const float o[3] = {rays_o[3 * r], rays_o[3 * r + 1], rays_o[3 * r + 2]};
        const float inv[3] = {1.0f / rays_d[3 * r], 1.0f / rays_d[3 * r + 1], 1.0f / rays_d[3 * r + 2]};
        float a = 0.f, b = 0.f;
        const bool hit = slab_test(o, inv, aabbs + 6 * g, near, far, a, b);
        t_mins[k] = hit ? a : miss;
        t_maxs[k] = hit ? b : miss;
        hits[k] = hit ? 1 : 0;
    }
}

// ----------------------------------------------------------------------------------------
// brick packing: one wave-lane per brick; each lane gathers its 4x4x4 voxels (16 loads of
// 4 contiguous bytes along z).  Runs once per grid update, 2 MiB read at 128^3.
// ----------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void pack_bricks_kernel(
    const uint8_t *__restrict__ binaries, int n_grids, int rx, int ry, int rz,
    int nbx, int nby, int nbz, uint64_t *__restrict__ bricks)
{
    const int64_t per_grid = (int64_t)nbx * nby * nbz;
    const int64_t total = per_grid * n_grids;
    for (int64_t b = (int64_t)blockIdx.x * kBlock + threadIdx.x; b < total; b += (int64_t)gridDim.x * kBlock) {
        const int64_t g = b / per_grid;
        int64_t rem = b - g * per_grid;
        const int bx = (int)(rem / ((int64_t)nby * nbz));
        rem -= (int64_t)bx * nby * nbz;
        const int by = (int)(rem / nbz), bz = (int)(rem - (int64_t)by * nbz);
        const uint8_t *grid = binaries + g * (int64_t)rx * ry * rz;
        uint64_t bits = 0;
#pragma unroll
        for (int dx = 0; dx < 4; ++dx) {
            const int x = bx * 4 + dx;
#pragma unroll
            for (int dy = 0; dy < 4; ++dy) {
                const int y = by * 4 + dy;
                if (x >= rx || y >= ry) continue;
                const uint8_t *row = grid + ((int64_t)x * ry + y) * rz + bz * 4;
#pragma unroll
                for (int dz = 0; dz < 4; ++dz) {
                    if (bz * 4 + dz < rz && row[dz]) bits |= 1ull << (dx * 16 + dy * 4 + dz);
                }
            }
        }
        bricks[b] = bits;
    }
}

// ----------------------------------------------------------------------------------------
// K2: traversal
// ----------------------------------------------------------------------------------------
struct GridView {
    const uint64_t *__restrict__ bricks;
    int res[3];
    int nbx, nby, nbz;
    int64_t bricks_per_grid;
};

struct BrickCache {
    int64_t id;
    uint64_t bits;
};

__device__ __forceinline__ bool occupied(const GridView &g, BrickCache &c, int level, int x, int y, int z) {
    const int64_t id = (((int64_t)(x >> 2) * g.nby + (y >> 2)) * g.nbz + (z >> 2)) + level * g.bricks_per_grid;
    if (id != c.id) { c.bits = g.bricks[id]; c.id = id; }
    return (c.bits >> (((x & 3) << 4) | ((y & 3) << 2) | (z & 3))) & 1ull;
}

// sorted ray/grid events: either the caller's arrays or computed in-kernel
template <bool PRECOMPUTED>
struct Events;

template <>
struct Events<true> {
    const uint8_t *hit;
    const float *t;
    const int64_t *id;
    __device__ __forceinline__ void init(const nfa_traverse_args &a, int64_t r, const float *, const float *) {
        hit = a.hits + r * a.n_grids;
        t = a.t_sorted + r * a.n_grids * 2;
        id = a.t_indices + r * a.n_grids * 2;
    }
    __device__ __forceinline__ bool hits(int level) const { return hit[level] != 0; }
    __device__ __forceinline__ float time(int i) const { return t[i]; }
    __device__ __forceinline__ int index(int i) const { return (int)id[i]; }
};

template <>
struct Events<false> {
    // grid.py:156-162 done per ray in registers/scratch: slab test against every level with
    // near = -inf, far = +inf, miss = +inf, then an ascending stable sort of the 2G times.
    float t[2 * NFA_MAX_GRID_LEVELS];
    int id[2 * NFA_MAX_GRID_LEVELS];
    bool hit[NFA_MAX_GRID_LEVELS];
    __device__ __forceinline__ void init(const nfa_traverse_args &a, int64_t, const float *o, const float *inv) {
        const int G = a.n_grids;
        for (int g = 0; g < G; ++g) {
            float t0 = 0.f, t1 = 0.f;
            const bool h = slab_test(o, inv, a.aabbs + 6 * g, -INFINITY, INFINITY, t0, t1);
            hit[g] = h;
            t[g] = h ? t0 : INFINITY;
            t[G + g] = h ? t1 : INFINITY;
            id[g] = g;
            id[G + g] = G + g;
        }
        for (int i = 1; i < 2 * G; ++i) {           // insertion sort, stable
            const float tv = t[i];
            const int iv = id[i];
            int j = i - 1;
            while (j >= 0 && t[j] > tv) { t[j + 1] = t[j]; id[j + 1] = id[j]; --j; }
            t[j + 1] = tv;
            id[j + 1] = iv;
        }
    }
    __device__ __forceinline__ bool hits(int level) const { return hit[level]; }
    __device__ __forceinline__ float time(int i) const { return t[i]; }
    __device__ __forceinline__ int index(int i) const { return id[i]; }
};

// One ray of grid.cu:95-281.  FILL=false counts, FILL=true writes at the given offsets.
template <bool FILL, bool PRECOMPUTED>
__device__ __forceinline__ void traverse_ray(
    const nfa_traverse_args &a, const GridView &gv, int64_t r,
    int64_t iv_base, int64_t sm_base, int64_t &n_iv_out, int64_t &n_sm_out, float &t_term)
{
    const float o[3] = {a.rays_o[3 * r], a.rays_o[3 * r + 1], a.rays_o[3 * r + 2]};
    const float d[3] = {a.rays_d[3 * r], a.rays_d[3 * r + 1], a.rays_d[3 * r + 2]};
    const float inv[3] = {1.0f / d[0], 1.0f / d[1], 1.0f / d[2]};
    const float near = a.near_planes[r], far = a.far_planes[r];
    const float step_size = a.step_size, cone = a.cone_angle;
    const int limit = a.traverse_steps_limit;
    const int G = a.n_grids;

    Events<PRECOMPUTED> ev;
    ev.init(a, r, o, inv);

    int64_t n_iv = 0, n_sm = 0;
    float t_last = near;
    bool continuous = false;
    BrickCache cache;
    cache.id = -1;
    cache.bits = 0;

    for (int i = 0; i + 1 < 2 * G; ++i) {
        int e = ev.index(i);
        int level = e % G;
        if (!ev.hits(level)) continue;
        if (e >= G) {                               // leaving `level`: are we inside another grid?
            const int e1 = ev.index(i + 1);
            if (e1 < G) continue;
            level = e1 % G;
            if (!ev.hits(level)) continue;
        }
        const float seg_lo = fmaxf(ev.time(i), near);
        const float seg_hi = fminf(ev.time(i + 1), far);
        if (seg_lo >= seg_hi) continue;

        if (!continuous) {
            if (step_size <= 0.0f) t_last = seg_lo;
            else t_last = lattice_skip(t_last, march_dt(t_last, cone, step_size), seg_lo);
        }

        Dda s;
        dda_setup(s, o, d, inv, seg_lo, seg_hi, a.aabbs + 6 * level, gv.res);

        while (limit <= 0 || n_sm < limit) {
            const float t_cell = fminf(fminf(s.tx, fminf(s.ty, s.tz)), seg_hi);
            if (!occupied(gv, cache, level, s.cx, s.cy, s.cz)) {
                if (step_size <= 0.0f) t_last = t_cell;
                else t_last = lattice_skip(t_last, march_dt(t_last, cone, step_size), t_cell);
                continuous = false;
            } else {
                while (limit <= 0 || n_sm < limit) {
                    float t_next;
                    if (step_size <= 0.0f) t_next = t_cell;
                    else {
                        const float dt = march_dt(t_last, cone, step_size);
                        if (t_last + dt * 0.5f >= t_cell) break;
                        t_next = t_last + dt;
                    }
                    if (FILL) {
                        if (a.iv_vals) {
                            const int64_t k = iv_base + n_iv;
                            if (!continuous) {
                                a.iv_vals[k] = t_last;      a.iv_ray_indices[k] = r;     a.iv_is_left[k] = 1;
                                a.iv_vals[k + 1] = t_next;  a.iv_ray_indices[k + 1] = r; a.iv_is_right[k + 1] = 1;
                            } else {
                                a.iv_vals[k] = t_next;      a.iv_ray_indices[k] = r;
                                a.iv_is_left[k - 1] = 1;    a.iv_is_right[k] = 1;
                            }
                        }
                        const int64_t k = sm_base + n_sm;
                        if (a.sm_vals) a.sm_vals[k] = (t_next + t_last) * 0.5f;
                        if (a.sm_ray_indices) a.sm_ray_indices[k] = r;
                        if (a.sm_is_valid) a.sm_is_valid[k] = 1;
                        if (a.t_starts) { a.t_starts[k] = t_last; a.t_ends[k] = t_next; }
                    }
                    n_iv += continuous ? 1 : 2;
                    n_sm += 1;
                    continuous = true;
                    t_last = t_next;
                    if (t_next >= t_cell) break;
                }
            }
            if (!dda_advance(s)) break;
        }
    }
    n_iv_out = n_iv;
    n_sm_out = n_sm;
    t_term = t_last;
}

__device__ __forceinline__ int64_t wave_sum_i64(int64_t v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_down(v, off, 64);
    return v;   // valid in lane 0
}

// pass 1.  Block b owns rays [256 b, 256 b + 256).  block_sums[2 b + {0,1}] = this block's
// {edge, sample} totals.
template <bool PRECOMPUTED>
__global__ __launch_bounds__(kBlock) void traverse_count_kernel(nfa_traverse_args a, GridView gv,
                                                                int64_t *__restrict__ block_sums)
{
    __shared__ int64_t part[2][kWavesPerBlock];
    const int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    int64_t n_iv = 0, n_sm = 0;
    if (r < a.n_rays && !(a.rays_mask && !a.rays_mask[r])) {
        float t_term;
        traverse_ray<false, PRECOMPUTED>(a, gv, r, 0, 0, n_iv, n_sm, t_term);
    }
    if (r < a.n_rays) {
        if (a.iv_cnts) a.iv_cnts[r] = n_iv;
        a.sm_cnts[r] = n_sm;
    }
    const int64_t w_iv = wave_sum_i64(n_iv), w_sm = wave_sum_i64(n_sm);
    const int wave = threadIdx.x >> 6;
    if (lane_id() == 0) { part[0][wave] = w_iv; part[1][wave] = w_sm; }
    __syncthreads();
    if (threadIdx.x == 0) {
        int64_t s0 = 0, s1 = 0;
        for (int w = 0; w < kWavesPerBlock; ++w) { s0 += part[0][w]; s1 += part[1][w]; }
        block_sums[2 * blockIdx.x] = s0;
        block_sums[2 * blockIdx.x + 1] = s1;
    }
}

// block-level exclusive scan of one int64 per thread (256 threads); returns the exclusive
// prefix, `total` gets the block total (all threads).
__device__ __forceinline__ int64_t block_excl_scan_i64(int64_t v, int64_t *lds /* [kWavesPerBlock] */, int64_t &total) {
    const int lane = lane_id(), wave = threadIdx.x >> 6;
    int64_t inc = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int64_t u = __shfl_up(inc, off, 64);
        if (lane >= off) inc += u;
    }
    if (lane == 63) lds[wave] = inc;
    __syncthreads();
    int64_t wave_off = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < kWavesPerBlock; ++w) {
        const int64_t s = lds[w];
        if (w < wave) wave_off += s;
        tot += s;
    }
    __syncthreads();
    total = tot;
    return wave_off + inc - v;
}

// pass 1b: offsets.  Every block first adds up the block sums before it (<= R/256 values,
// L2-resident), then scans its own 256 counts.  The last block also stores the totals.
__global__ __launch_bounds__(kBlock) void traverse_offsets_kernel(
    const int64_t *__restrict__ iv_cnts, int64_t *__restrict__ iv_starts,
    const int64_t *__restrict__ sm_cnts, int64_t *__restrict__ sm_starts,
    int64_t n_rays, const int64_t *__restrict__ block_sums, int64_t *__restrict__ totals)
{
    __shared__ int64_t lds[kWavesPerBlock];
    __shared__ int64_t base[2];
    const int b = blockIdx.x;
    // prefix over earlier blocks
    int64_t p0 = 0, p1 = 0;
    for (int j = threadIdx.x; j < b; j += kBlock) { p0 += block_sums[2 * j]; p1 += block_sums[2 * j + 1]; }
    int64_t t0, t1;
    block_excl_scan_i64(p0, lds, t0);
    block_excl_scan_i64(p1, lds, t1);
    if (threadIdx.x == 0) { base[0] = t0; base[1] = t1; }
    __syncthreads();
    const int64_t r = (int64_t)b * kBlock + threadIdx.x;
    const bool in = r < n_rays;
    int64_t tot;
    if (iv_cnts) {
        const int64_t c = in ? iv_cnts[r] : 0;
        const int64_t e = block_excl_scan_i64(c, lds, tot);
        if (in) iv_starts[r] = base[0] + e;
        if (b == (int)gridDim.x - 1 && threadIdx.x == 0) totals[0] = base[0] + tot;
    } else if (b == (int)gridDim.x - 1 && threadIdx.x == 0) totals[0] = 0;
    {
        const int64_t c = in ? sm_cnts[r] : 0;
        const int64_t e = block_excl_scan_i64(c, lds, tot);
        if (in) sm_starts[r] = base[1] + e;
        if (b == (int)gridDim.x - 1 && threadIdx.x == 0) totals[1] = base[1] + tot;
    }
}

// pass 2
template <bool PRECOMPUTED>
__global__ __launch_bounds__(kBlock) void traverse_fill_kernel(nfa_traverse_args a, GridView gv,
                                                               int skip_empty, int rewrite_counts)
{
    const int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (r >= a.n_rays) return;
    if (a.rays_mask && !a.rays_mask[r]) return;
    if (skip_empty) {
        if (a.iv_cnts && a.iv_cnts[r] == 0) return;
        if (a.sm_cnts[r] == 0) return;
    }
    int64_t n_iv, n_sm;
    float t_term;
    traverse_ray<true, PRECOMPUTED>(a, gv, r, a.iv_starts ? a.iv_starts[r] : 0, a.sm_starts[r], n_iv, n_sm, t_term);
    if (a.terminate_planes) a.terminate_planes[r] = t_term;
    if (rewrite_counts) {
        if (a.iv_cnts) a.iv_cnts[r] = n_iv;
        a.sm_cnts[r] = n_sm;
    }
}

// generic exclusive sum of int64 counts (data_spec.hpp:86-106), single workgroup of 1024:
// rounds of 1024 coalesced elements with a running carry.  Used for the small per-ray count
// arrays of the over-allocated traversal mode only.
__global__ __launch_bounds__(1024) void excl_sum_i64_kernel(const int64_t *__restrict__ cnts, int64_t n,
                                                            int64_t *__restrict__ starts, int64_t *__restrict__ total)
{
    __shared__ int64_t wsum[16];
    __shared__ int64_t carry_s;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    const int lane = lane_id(), wave = threadIdx.x >> 6;
    for (int64_t base = 0; base < n; base += 1024) {
        const int64_t i = base + threadIdx.x;
        const int64_t v = i < n ? cnts[i] : 0;
        int64_t inc = v;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int64_t u = __shfl_up(inc, off, 64);
            if (lane >= off) inc += u;
        }
        if (lane == 63) wsum[wave] = inc;
        __syncthreads();
        int64_t woff = 0, tot = 0;
        for (int w = 0; w < 16; ++w) { const int64_t s = wsum[w]; if (w < wave) woff += s; tot += s; }
        const int64_t carry = carry_s;
        if (i < n) starts[i] = carry + woff + inc - v;
        __syncthreads();
        if (threadIdx.x == 0) carry_s = carry + tot;
        __syncthreads();
    }
    if (threadIdx.x == 0 && total) *total = carry_s;
}

int validate_traverse(const nfa_traverse_args *a) {
    NFA_REQUIRE(a != nullptr, "traverse: args is NULL");
    NFA_REQUIRE(a->n_rays >= 0, "traverse: n_rays < 0");
    NFA_REQUIRE(a->n_grids >= 1 && a->n_grids <= NFA_MAX_GRID_LEVELS, "traverse: n_grids=%d not in [1,%d]", a->n_grids, NFA_MAX_GRID_LEVELS);
    NFA_REQUIRE(a->res[0] > 0 && a->res[1] > 0 && a->res[2] > 0, "traverse: bad resolution");
    if (a->n_rays == 0) return NFA_OK;
    NFA_REQUIRE(a->rays_o && a->rays_d && a->bricks && a->aabbs && a->near_planes && a->far_planes, "traverse: NULL input");
    const int given = (a->hits != nullptr) + (a->t_sorted != nullptr) + (a->t_indices != nullptr);
    NFA_REQUIRE(given == 0 || given == 3, "traverse: hits/t_sorted/t_indices must be given together");
    NFA_REQUIRE(a->sm_cnts && a->sm_starts, "traverse: sm_cnts/sm_starts are required");
    NFA_REQUIRE((a->iv_cnts == nullptr) == (a->iv_starts == nullptr), "traverse: iv_cnts/iv_starts go together");
    return NFA_OK;
}

GridView make_view(const nfa_traverse_args *a) {
    GridView gv;
    gv.bricks = a->bricks;
    for (int k = 0; k < 3; ++k) gv.res[k] = a->res[k];
    gv.nbx = (a->res[0] + 3) / 4;
    gv.nby = (a->res[1] + 3) / 4;
    gv.nbz = (a->res[2] + 3) / 4;
    gv.bricks_per_grid = (int64_t)gv.nbx * gv.nby * gv.nbz;
    return gv;
}

}  // namespace
}  // namespace nfa

using namespace nfa;

NFA_EXPORT const char *nfa_version(void) { return "nerfacc_hip 0.1.0 gfx950"; }
NFA_EXPORT const char *nfa_last_error(void) { return last_error_buffer(); }

NFA_EXPORT int nfa_ray_aabb_intersect(const float *rays_o, const float *rays_d, int64_t n_rays,
                                      const float *aabbs, int64_t n_aabbs, float near_plane, float far_plane,
                                      float miss_value, float *t_mins, float *t_maxs, uint8_t *hits, void *stream)
{
    NFA_REQUIRE(n_rays >= 0 && n_aabbs >= 0, "ray_aabb_intersect: negative size");
    const int64_t total = n_rays * n_aabbs;
    if (total == 0) return NFA_OK;
    NFA_REQUIRE(rays_o && rays_d && aabbs && t_mins && t_maxs && hits, "ray_aabb_intersect: NULL pointer");
    hipLaunchKernelGGL(ray_aabb_kernel, dim3(blocks_for(total)), dim3(kBlock), 0, (hipStream_t)stream,
                       rays_o, rays_d, n_rays, aabbs, n_aabbs, near_plane, far_plane, miss_value, t_mins, t_maxs, hits);
    return check_launch("ray_aabb_kernel");
}

NFA_EXPORT int64_t nfa_packed_grid_words(int32_t n_grids, int32_t rx, int32_t ry, int32_t rz) {
    if (n_grids <= 0 || rx <= 0 || ry <= 0 || rz <= 0) return 0;
    return (int64_t)n_grids * ((rx + 3) / 4) * ((ry + 3) / 4) * ((rz + 3) / 4);
}

NFA_EXPORT int nfa_pack_binaries(const uint8_t *binaries, int32_t n_grids, int32_t rx, int32_t ry, int32_t rz,
                                 uint64_t *bricks, void *stream)
{
    const int64_t words = nfa_packed_grid_words(n_grids, rx, ry, rz);
    NFA_REQUIRE(words > 0, "pack_binaries: empty grid");
    NFA_REQUIRE(binaries && bricks, "pack_binaries: NULL pointer");
    hipLaunchKernelGGL(pack_bricks_kernel, dim3(blocks_for(words)), dim3(kBlock), 0, (hipStream_t)stream,
                       binaries, n_grids, rx, ry, rz, (rx + 3) / 4, (ry + 3) / 4, (rz + 3) / 4, bricks);
    return check_launch("pack_bricks_kernel");
}

NFA_EXPORT int64_t nfa_traverse_workspace_bytes(int64_t n_rays) {
    return 2 * (int64_t)sizeof(int64_t) * (ceil_div(n_rays > 0 ? n_rays : 1, kBlock));
}

NFA_EXPORT int nfa_traverse_count(const nfa_traverse_args *a, void *workspace, void *stream)
{
    if (int rc = validate_traverse(a)) return rc;
    NFA_REQUIRE(a->totals != nullptr, "traverse_count: totals is NULL");
    if (a->n_rays == 0) {
        (void)hipMemsetAsync(a->totals, 0, 2 * sizeof(int64_t), (hipStream_t)stream);
        return NFA_OK;
    }
    NFA_REQUIRE(workspace != nullptr, "traverse_count: workspace is NULL");
    const GridView gv = make_view(a);
    const unsigned nb = (unsigned)ceil_div(a->n_rays, kBlock);
    int64_t *block_sums = (int64_t *)workspace;
    if (a->t_sorted)
        hipLaunchKernelGGL(traverse_count_kernel<true>, dim3(nb), dim3(kBlock), 0, (hipStream_t)stream, *a, gv, block_sums);
    else
        hipLaunchKernelGGL(traverse_count_kernel<false>, dim3(nb), dim3(kBlock), 0, (hipStream_t)stream, *a, gv, block_sums);
    if (int rc = check_launch("traverse_count_kernel")) return rc;
    hipLaunchKernelGGL(traverse_offsets_kernel, dim3(nb), dim3(kBlock), 0, (hipStream_t)stream,
                       a->iv_cnts, a->iv_starts, a->sm_cnts, a->sm_starts, a->n_rays, block_sums, a->totals);
    return check_launch("traverse_offsets_kernel");
}

NFA_EXPORT int nfa_traverse_fill(const nfa_traverse_args *a, int32_t skip_empty, int32_t rewrite_counts, void *stream)
{
    if (int rc = validate_traverse(a)) return rc;
    if (a->n_rays == 0) return NFA_OK;
    if (a->iv_vals) NFA_REQUIRE(a->iv_ray_indices && a->iv_is_left && a->iv_is_right && a->iv_starts,
                                "traverse_fill: interval outputs must be given together");
    if (a->t_starts) NFA_REQUIRE(a->t_ends != nullptr, "traverse_fill: t_starts without t_ends");
    const GridView gv = make_view(a);
    const unsigned nb = (unsigned)ceil_div(a->n_rays, kBlock);
    if (a->t_sorted)
        hipLaunchKernelGGL(traverse_fill_kernel<true>, dim3(nb), dim3(kBlock), 0, (hipStream_t)stream, *a, gv, skip_empty, rewrite_counts);
    else
        hipLaunchKernelGGL(traverse_fill_kernel<false>, dim3(nb), dim3(kBlock), 0, (hipStream_t)stream, *a, gv, skip_empty, rewrite_counts);
    return check_launch("traverse_fill_kernel");
}

NFA_EXPORT int nfa_exclusive_sum_i64(const int64_t *cnts, int64_t n, int64_t *starts, int64_t *total, void *stream)
{
    NFA_REQUIRE(n >= 0, "exclusive_sum_i64: n < 0");
    if (n == 0) {
        if (total) (void)hipMemsetAsync(total, 0, sizeof(int64_t), (hipStream_t)stream);
        return NFA_OK;
    }
    NFA_REQUIRE(cnts && starts, "exclusive_sum_i64: NULL pointer");
    hipLaunchKernelGGL(excl_sum_i64_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, cnts, n, starts, total);
    return check_launch("excl_sum_i64_kernel");
}
