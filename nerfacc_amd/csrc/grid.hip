// grid.hip — ray/AABB test, occupancy-brick packing and multi-level grid traversal for gfx950.
//
// Replaces nerfacc/cuda/csrc/grid.cu (+ include/utils_grid.cuh) behind the C ABI of
// include/nerfacc_hip.h.  Semantics are those restated in oracle/nerfacc_oracle.c; the
// float-op order and the explicit fmaf() sites are identical on both sides so that per-ray
// sample counts agree bit for bit (this file is compiled with -ffp-contract=off).
//
// MI355X mapping (DESIGN.md section "traversal"):
//   * occupancy: the 1-byte-per-voxel grid is repacked into 4x4x4 bricks of one uint64, plus a
//     bitmap of non-empty bricks, its rank prefix and the compacted non-empty bricks.  A
//     NeRF-like 128^3 grid is then ~4 KiB + 4 KiB + ~10-30 KiB and is staged into LDS by every
//     workgroup, so the voxel walk never waits on global memory;
//   * pass 1, one lane per ray: the DDA loop only walks voxels and appends occupied<->empty
//     transitions to a per-lane LDS list.  All lattice arithmetic happens afterwards, once
//     per transition, with the exact closed forms of lattice.hpp — under SIMT a rare
//     expensive branch inside the voxel loop would be paid by the whole wave at almost every
//     step, and a nested per-voxel lattice loop (the reference's shape) costs the slowest
//     lane's trip count at every voxel;
//   * pass 1 records every ray's samples as run-length records; block sums + a tiny second
//     kernel give offsets and totals (one 32-byte readback per traverse_grids call);
//   * pass 2, one lane per OUTPUT SAMPLE: binary-search the ray, pick the run, jump to the
//     lattice point in closed form, store — fully coalesced, no grid access, no divergence.
#include <cstdlib>
#include <mutex>
#include <unordered_map>

#include "common.hpp"
#include "lattice.hpp"

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

// utils_grid.cuh:116-142: step along the axis whose next plane is strictly nearest (ties go
// z, then y, then x by the strict '<' chain).  Written with selects: the three-way branch of
// the reference makes a wave execute all three arms at almost every voxel.
__device__ __forceinline__ bool dda_advance(Dda &s) {
    const bool ax = (s.tx < s.ty) && (s.tx < s.tz);
    const bool ay = !ax && (s.ty < s.tz);
    const bool az = !ax && !ay;
    s.cx += ax ? s.sx : 0;
    s.cy += ay ? s.sy : 0;
    s.cz += az ? s.sz : 0;
    s.tx = ax ? s.tx + s.dx : s.tx;
    s.ty = ay ? s.ty + s.dy : s.ty;
    s.tz = az ? s.tz + s.dz : s.tz;
    // (bitwise on purpose: a select between the three overflow indices makes the compiler
    //  spill them to scratch and index them, one scratch load per voxel)
    const bool hit_x = s.cx == s.ox, hit_y = s.cy == s.oy, hit_z = s.cz == s.oz;
    return !((ax & hit_x) | (ay & hit_y) | (az & hit_z));
}

// grid.cu:157-161 / 199-203: advance the marching lattice, t += dt with dt fixed, until the
// midpoint t + dt/2 reaches `target` (with the t + dt == t escape, see oracle).  The adds are
// inherently sequential (bit-exact parity needs the same rounding chain), so the cost is in
// the loop control: eight predicated steps per branch keep the wave out of divergent
// short-trip loops.
__device__ __forceinline__ float lattice_skip(float t, float dt, float target) {
    const float h = dt * 0.5f;
    while (t + h < target) {
        bool stuck = false;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const bool go = t + h < target;
            const float nt = t + dt;
            stuck = stuck || (go && nt == t);
            t = go ? nt : t;
        }
        if (stuck) return target;
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
// occupancy packing.  Buffer layout (uint64 words), n_bricks = G * nbx * nby * nbz,
// n_words = ceil(n_bricks / 32):
//   [0, n_bricks)                       dense bricks (bit = (x&3)*16 + (y&3)*4 + (z&3))
//   [n_bricks, +4)                      header: {n_compact, 0, 0, 0}
//   then coarse[n_words] u32 (1 bit per brick), prefix[n_words] u32 (non-empty bricks before
//   the word), each padded to 8 bytes, then compact[n_bricks] (the non-empty bricks in order).
// ----------------------------------------------------------------------------------------
struct PackedLayout {
    int64_t n_bricks, n_words, off_header, off_coarse, off_prefix, off_compact, total_words;
};
__host__ __device__ inline PackedLayout packed_layout(int n_grids, int rx, int ry, int rz) {
    PackedLayout L;
    L.n_bricks = (int64_t)n_grids * ((rx + 3) / 4) * ((ry + 3) / 4) * ((rz + 3) / 4);
    L.n_words = (L.n_bricks + 31) / 32;
    L.off_header = L.n_bricks;
    L.off_coarse = L.off_header + 12;          // header: [0] non-empty bricks, [1 + g] occupied voxels of level g (g < 8), 3 spare
    L.off_prefix = L.off_coarse + (L.n_words + 1) / 2;
    L.off_compact = L.off_prefix + (L.n_words + 1) / 2;
    L.total_words = L.off_compact + L.n_bricks;
    return L;
}

// one lane per brick: gather its 4x4x4 voxels (16 loads of 4 contiguous bytes along z); the
// wave's ballot of "non-empty" is the coarse bitmap (two u32 words per wave).
// FROM_OCCS: the voxels come from the float occupancies (`occs > min(mean, occ_thre)`, occ_grid.py:392-404) and the
// bool grid is an OUTPUT — threshold and bit-pack in one pass (nfa_grid_threshold_packed).
template <bool FROM_OCCS>
__global__ __launch_bounds__(kBlock) void pack_bricks_kernel(
    const uint8_t *__restrict__ binaries, int n_grids, int rx, int ry, int rz,
    int nbx, int nby, int nbz, uint64_t *__restrict__ bricks, uint32_t *__restrict__ coarse, int64_t *__restrict__ level_counts,
    const float *__restrict__ occs, const double *__restrict__ partials, int n_partials, float occ_thre,
    uint8_t *__restrict__ binaries_out, float *__restrict__ thre_out)
{
    float thre = 0.0f;
    if (FROM_OCCS) {
        __shared__ float s_thre;
        if (threadIdx.x < 64) {
            const float th = threshold_from_partials(partials, n_partials, occ_thre);
            if (threadIdx.x == 0) {
                s_thre = th;
                if (blockIdx.x == 0 && thre_out) *thre_out = th;
            }
        }
        __syncthreads();
        thre = s_thre;
    }
    const int64_t per_grid = (int64_t)nbx * nby * nbz;
    const int64_t total = per_grid * n_grids;
    const int64_t rounded = (total + 63) / 64 * 64;
    for (int64_t b = (int64_t)blockIdx.x * kBlock + threadIdx.x; b < rounded; b += (int64_t)gridDim.x * kBlock) {
        uint64_t bits = 0;
        int64_t g = 0;
        if (b < total) {
            g = b / per_grid;
            int64_t rem = b - g * per_grid;
            const int bx = (int)(rem / ((int64_t)nby * nbz));
            rem -= (int64_t)bx * nby * nbz;
            const int by = (int)(rem / nbz), bz = (int)(rem - (int64_t)by * nbz);
            const int64_t grid_off = g * (int64_t)rx * ry * rz;
#pragma unroll
            for (int dx = 0; dx < 4; ++dx) {
                const int x = bx * 4 + dx;
#pragma unroll
                for (int dy = 0; dy < 4; ++dy) {
                    const int y = by * 4 + dy;
                    if (x >= rx || y >= ry) continue;
                    const int64_t row = grid_off + ((int64_t)x * ry + y) * rz + bz * 4;
#pragma unroll
                    for (int dz = 0; dz < 4; ++dz) {
                        if (bz * 4 + dz >= rz) continue;
                        bool on;
                        if (FROM_OCCS) { on = occs[row + dz] > thre; binaries_out[row + dz] = on ? 1 : 0; }
                        else on = binaries[row + dz] != 0;
                        if (on) bits |= 1ull << (dx * 16 + dy * 4 + dz);
                    }
                }
            }
            bricks[b] = bits;
        }
        const unsigned long long any = __ballot(bits != 0ull);
        const int lane = lane_id();
        // occupied voxels per level (integer atomics: deterministic): lets OccGridEstimator size `nonzero` without a sync.
        // One atomic per wave — every brick adding to the same word serialises in the L2 (105 us for 32 k bricks) —
        // unless the wave straddles two levels.
        if (any) {
            const int64_t g0 = __builtin_amdgcn_readfirstlane((int)g);
            if (__ballot(b < total && g != g0) == 0ull) {
                const int64_t cnt = wave_sum_i64(__popcll(bits));
                if (lane == 0) atomicAdd((unsigned long long *)(level_counts + g0), (unsigned long long)cnt);
            } else if (bits) {
                atomicAdd((unsigned long long *)(level_counts + g), (unsigned long long)__popcll(bits));
            }
        }
        const int64_t w = b >> 5;                       // coarse word of this lane's brick
        if ((lane & 31) == 0 && (b < total))
            coarse[w] = (uint32_t)(lane ? (any >> 32) : any);
    }
}

// rank prefix over the coarse words (single workgroup; n_words is 1024 for 128^3)
__global__ __launch_bounds__(1024) void rank_bricks_kernel(const uint32_t *__restrict__ coarse, int64_t n_words,
                                                           uint32_t *__restrict__ prefix, int64_t *__restrict__ header)
{
    __shared__ uint32_t wsum[16];
    __shared__ uint32_t carry_s;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    const int lane = lane_id(), wave = threadIdx.x >> 6;
    for (int64_t base = 0; base < n_words; base += 1024) {
        const int64_t i = base + threadIdx.x;
        const uint32_t v = i < n_words ? (uint32_t)__popc(coarse[i]) : 0u;
        uint32_t inc = v;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t u = __shfl_up(inc, off, 64);
            if (lane >= off) inc += u;
        }
        if (lane == 63) wsum[wave] = inc;
        __syncthreads();
        uint32_t woff = 0, tot = 0;
        for (int w = 0; w < 16; ++w) { const uint32_t x = wsum[w]; if (w < wave) woff += x; tot += x; }
        const uint32_t carry = carry_s;
        if (i < n_words) prefix[i] = carry + woff + inc - v;
        __syncthreads();
        if (threadIdx.x == 0) carry_s = carry + tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) header[0] = carry_s;          // ([1..8]: per-level voxel counts, accumulated by pack_bricks_kernel)
}

__global__ __launch_bounds__(kBlock) void compact_bricks_kernel(const uint64_t *__restrict__ bricks, int64_t n_bricks,
                                                                const uint32_t *__restrict__ coarse,
                                                                const uint32_t *__restrict__ prefix,
                                                                uint64_t *__restrict__ compact)
{
    for (int64_t b = (int64_t)blockIdx.x * kBlock + threadIdx.x; b < n_bricks; b += (int64_t)gridDim.x * kBlock) {
        const uint64_t bits = bricks[b];
        if (!bits) continue;
        const uint32_t w = coarse[b >> 5];
        compact[prefix[b >> 5] + __popc(w & ((1u << (b & 31)) - 1u))] = bits;
    }
}

// ----------------------------------------------------------------------------------------
// K2: traversal
// ----------------------------------------------------------------------------------------
struct GridView {
    const uint64_t *__restrict__ bricks;     // dense, global
    const uint32_t *__restrict__ coarse;     // global copies of the sparse form
    const uint32_t *__restrict__ prefix;
    const uint64_t *__restrict__ compact;
    const int64_t *__restrict__ header;
    int res[3];
    int nbx, nby, nbz;
    int bricks_per_grid;
    int n_words;
    int lds_words;        // bitmap (and rank) words staged in LDS (0 = nothing staged)
    int lds_compact_cap;  // compact bricks staged in LDS (0 with lds_words > 0: bitmap only, bricks from L2)
};

struct BrickCache {
    int id;
    uint64_t bits;
};

// Where a voxel's occupancy comes from, per kernel variant:
//   LDS_OCC  the whole sparse form sits in LDS — bitmap of non-empty bricks, its rank prefix and the
//            compacted non-empty bricks (arrays addressed off the dynamic-LDS base so the compiler
//            emits ds_read, not flat loads).  Chosen when the non-empty bricks fit (make_view).
//   else     the dense brick array in global memory (L2-resident: 8 B per 64 voxels), ONE load per
//            brick the walk enters — a dependent bitmap -> rank -> brick chain through L2 costs
//            three latencies per brick.  When the bitmap alone fits in LDS it is staged and
//            answers the empty bricks without touching memory.
// LDS image layout: {coarse, prefix}[W4] as uint2 (ONE ds_read_b64 answers "is the brick empty" and gives its
// rank; as two arrays the rank was a second, dependent LDS latency per non-empty brick) | compact[cap] u64,
// W4 = lds_words rounded to 4 (bitmap-only: just coarse[W4] u32).
template <bool LDS_OCC>
struct Occ {
    const char *smem;
    int w4, cap;
    int bytes;      // LDS bytes of the image; the per-lane boundary lists start behind it (16-aligned)
};

template <bool LDS_OCC>
__device__ __forceinline__ Occ<LDS_OCC> stage_occupancy(const GridView &g, char *smem) {
    Occ<LDS_OCC> l;
    l.smem = smem;
    l.w4 = (g.lds_words + 3) & ~3;
    l.cap = 0;
    l.bytes = 0;
    uint32_t *lc = (uint32_t *)smem;
    if (LDS_OCC) {
        uint2 *lw = (uint2 *)smem;
        uint64_t *lb = (uint64_t *)(lc + 2 * l.w4);
        for (int i = threadIdx.x; i < g.lds_words; i += blockDim.x) lw[i] = make_uint2(g.coarse[i], g.prefix[i]);
        // make_view selects this variant only when the caller gave the number of non-empty bricks and sized the image for
        // exactly that many (args.n_nonempty_bricks "MUST be that value"): no dependent load of the header word in front of
        // the copy loop — one L2 round trip less at the start of every traversal workgroup
        const int n_compact = g.lds_compact_cap;
        for (int i = threadIdx.x; i < n_compact; i += blockDim.x) lb[i] = g.compact[i];
        l.cap = n_compact;
        l.bytes = (2 * l.w4 * 4 + g.lds_compact_cap * 8 + 15) & ~15;
        __syncthreads();
    } else if (g.lds_words > 0) {
        for (int i = threadIdx.x; i < g.lds_words; i += blockDim.x) lc[i] = g.coarse[i];
        l.bytes = (l.w4 * 4 + 15) & ~15;
        __syncthreads();
    }
    return l;
}

// occupancy of voxel (x,y,z); the brick is re-resolved only when the walk enters another brick
template <bool LDS_OCC>
__device__ __forceinline__ bool occupied(const GridView &g, const Occ<LDS_OCC> &l, BrickCache &c, int level, int x, int y, int z) {
    // brick counts are < 2^24 (checked on the host): full-rate 24-bit multiplies
    const int id = (int)__umul24(__umul24(x >> 2, g.nby) + (y >> 2), g.nbz) + (z >> 2) + level * g.bricks_per_grid;
    if (id != c.id) {
        c.id = id;
        const uint32_t bit = 1u << (id & 31);
        uint64_t bits = 0;
        const uint32_t *lc = (const uint32_t *)l.smem;
        if (LDS_OCC) {
            const uint2 wr = ((const uint2 *)l.smem)[id >> 5];
            const uint32_t w = wr.x;
            if (w & bit) {
                const int k = (int)wr.y + __popc(w & (bit - 1u));
                // the LDS image holds ALL non-empty bricks (make_view only selects this variant
                // when they fit), so there is no global fallback here: a "k < cap ? lds : global"
                // select is compiled into one flat load, which is what this code avoids
                bits = ((const uint64_t *)(lc + 2 * l.w4))[k];
            }
        } else if (l.bytes > 0) {                     // bitmap in LDS, bricks dense in L2 (wave-uniform branch)
            if (lc[id >> 5] & bit) bits = g.bricks[id];
        } else {
            bits = g.bricks[id];
        }
        c.bits = bits;
    }
    return (c.bits >> (((x & 3) << 4) | ((y & 3) << 2) | (z & 3))) & 1ull;
}

// sorted ray/grid events (grid.py:156-162): EV_PRE = the caller's arrays; EV_ONE = one level,
// slab test in registers (the Lego configuration: no sort, no arrays); EV_MANY = several levels,
// slab tests + an insertion sort of the 2G event times in per-lane scratch.
enum { EV_PRE = 0, EV_ONE = 1, EV_MANY = 2 };

template <int MODE>
struct Events;

template <>
struct Events<EV_PRE> {
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
struct Events<EV_ONE> {
    float t0, t1;
    bool hit;
    __device__ __forceinline__ void init(const nfa_traverse_args &a, int64_t, const float *o, const float *inv) {
        t0 = t1 = INFINITY;
        float x0 = 0.f, x1 = 0.f;
        hit = slab_test(o, inv, a.aabbs, -INFINITY, INFINITY, x0, x1);
        if (hit) { t0 = x0; t1 = x1; }
    }
    __device__ __forceinline__ bool hits(int) const { return hit; }
    __device__ __forceinline__ float time(int i) const { return i == 0 ? t0 : t1; }
    __device__ __forceinline__ int index(int i) const { return i; }
};

template <>
struct Events<EV_MANY> {
    float t[2 * NFA_MAX_GRID_LEVELS];
    int id[2 * NFA_MAX_GRID_LEVELS];
    bool hit[NFA_MAX_GRID_LEVELS];
    __device__ __forceinline__ void init(const nfa_traverse_args &a, int64_t, const float *o, const float *inv) {
        const int G = a.n_grids;
        for (int g = 0; g < G; ++g) {
            float x0 = 0.f, x1 = 0.f;
            const bool h = slab_test(o, inv, a.aabbs + 6 * g, -INFINITY, INFINITY, x0, x1);
            hit[g] = h;
            t[g] = h ? x0 : INFINITY;
            t[G + g] = h ? x1 : INFINITY;
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

// resolve the grid level and clipped [seg_lo, seg_hi) between sorted events i and i+1
// (grid.cu:131-150); false = nothing to traverse there
template <class Ev>
__device__ __forceinline__ bool segment_of(const Ev &ev, int i, int G, float near, float far, int &level, float &lo, float &hi) {
    const int e = ev.index(i);
    level = e % G;
    if (!ev.hits(level)) return false;
    if (e >= G) {                                   // leaving `level`: are we inside another grid?
        const int e1 = ev.index(i + 1);
        if (e1 < G) return false;
        level = e1 % G;
        if (!ev.hits(level)) return false;
    }
    lo = fmaxf(ev.time(i), near);
    hi = fminf(ev.time(i + 1), far);
    return lo < hi;
}

// ---- run-length records (pass 1 -> pass 2) -------------------------------------------------
// All samples of a ray sit on its marching lattice, so a maximal run of consecutive samples is
// described by the lattice point it starts at (the exact float) and the index of its first
// sample within the ray.  Up to max_runs runs per ray, laid out [run][ray] so a wave's lanes touch
// consecutive words; a ray with more runs (or step_size <= 0, where edges come from voxel faces)
// is flagged and re-traversed in pass 2.  max_runs is sized from the ray count (run_capacity):
// blob-like grids need a handful, a noise grid (tests/test_grid.py: rand > 0.5) a run per ~2 voxels.
constexpr int kMinRuns = 14, kMaxRunsCap = 512;
constexpr int kRunsOverflow = 0xffff;
constexpr int64_t kRunBudgetBytes = 256ll << 20;

inline int run_capacity(int64_t n_rays) {
    const int64_t fit = kRunBudgetBytes / (8 * (n_rays > 0 ? n_rays : 1));
    return (int)(fit < kMinRuns ? kMinRuns : (fit > kMaxRunsCap ? kMaxRunsCap : fit));
}

struct RunStore {
    float *t0;         // [max_runs][R] lattice point the run starts at
    int32_t *first;    // [max_runs][R] index of the run's first sample within its ray
    uint16_t *n_runs;  // [R]
    int max_runs;
};

struct CountSink {
    RunStore rs;
    int64_t r, R;
    int64_t n_iv = 0, n_sm = 0;
    int n_runs = 0;
    // k consecutive samples starting at lattice point t0
    __device__ __forceinline__ void run(float t0, int64_t k, bool continuous) {
        if (k <= 0) return;
        if (rs.t0 && !continuous) {
            if (n_runs < rs.max_runs) { rs.t0[(int64_t)n_runs * R + r] = t0; rs.first[(int64_t)n_runs * R + r] = (int32_t)n_sm; }
            n_runs += 1;
        }
        n_iv += continuous ? k : k + 1;
        n_sm += k;
    }
    __device__ __forceinline__ void sample(float t0, float, bool continuous) { run(t0, 1, continuous); }
    // returns true when the ray needs the re-traversal fallback
    __device__ __forceinline__ bool finish(bool replayable) {
        if (!rs.t0) return false;
        const bool overflow = (n_sm > 0) && (n_runs > rs.max_runs || !replayable || n_sm > 0x7fffffffll);
        rs.n_runs[r] = (uint16_t)(overflow ? kRunsOverflow : n_runs);
        return overflow;
    }
};

struct FillSink {
    const nfa_traverse_args &a;
    int64_t r, iv_base, sm_base;
    int64_t n_iv = 0, n_sm = 0;
    __device__ __forceinline__ void sample(float t0, float t1, bool continuous) {
        if (a.iv_vals) {
            const int64_t k = iv_base + n_iv;
            if (!continuous) {
                a.iv_vals[k] = t0;      a.iv_ray_indices[k] = r;     a.iv_is_left[k] = 1;
                a.iv_vals[k + 1] = t1;  a.iv_ray_indices[k + 1] = r; a.iv_is_right[k + 1] = 1;
            } else {
                a.iv_vals[k] = t1;      a.iv_ray_indices[k] = r;
                a.iv_is_left[k - 1] = 1; a.iv_is_right[k] = 1;
            }
        }
        const int64_t k = sm_base + n_sm;
        if (a.sm_vals) a.sm_vals[k] = (t1 + t0) * 0.5f;
        if (a.sm_ray_indices) a.sm_ray_indices[k] = r;
        if (a.sm_is_valid) a.sm_is_valid[k] = 1;
        if (a.t_starts) { a.t_starts[k] = t0; a.t_ends[k] = t1; }
        n_iv += continuous ? 1 : 2;
        n_sm += 1;
    }
};

// per-ray near / far plane as OccGridEstimator.sampling forms them (occ_grid.py:154-163) — the same float
// operations in the same order as the torch expressions, so the result is bit-identical to passing tensors:
//   near = full_like(near_plane); near = clamp(near, min=t_min); near += rand * render_step_size
//   far  = full_like(far_plane);  far  = clamp(far, max=t_max)
// Tensors (near_planes / far_planes) win when given; the scalar + t_min/t_max + jitter form saves the caller five
// elementwise launches per sampling call.
__device__ __forceinline__ float ray_near(const nfa_traverse_args &a, int64_t r) {
    float v = a.near_planes ? a.near_planes[r] : a.near_plane;
    if (a.t_min) { const float m = a.t_min[r]; v = (m != m) ? m : (v < m ? m : v); }        // torch.clamp(min=): NaN bound propagates
    if (a.jitter) v = v + a.jitter[r] * a.jitter_scale;                                        // -ffp-contract=off: mul, then add
    return v;
}
__device__ __forceinline__ float ray_far(const nfa_traverse_args &a, int64_t r) {
    float v = a.far_planes ? a.far_planes[r] : a.far_plane;
    if (a.t_max) { const float m = a.t_max[r]; v = (m != m) ? m : (v > m ? m : v); }
    return v;
}

// ---- general walk: any step_size / cone_angle; the reference's loop shape (grid.cu:95-281).
// Used for cone_angle != 0, step_size <= 0, the over-allocated test-time pass and as the
// pass-2 fallback of rays whose runs did not fit.
template <class Sink, int EV, bool LDS_OCC>
__device__ __forceinline__ void traverse_ray_general(const nfa_traverse_args &a, const GridView &gv, const Occ<LDS_OCC> &occ,
                                                     int64_t r, Sink &sink, float &t_term)
{
    const float o[3] = {a.rays_o[3 * r], a.rays_o[3 * r + 1], a.rays_o[3 * r + 2]};
    const float d[3] = {a.rays_d[3 * r], a.rays_d[3 * r + 1], a.rays_d[3 * r + 2]};
    const float inv[3] = {1.0f / d[0], 1.0f / d[1], 1.0f / d[2]};
    const float near = ray_near(a, r), far = ray_far(a, r);
    const float step_size = a.step_size, cone = a.cone_angle;
    const int limit = a.traverse_steps_limit;
    const int G = a.n_grids;

    Events<EV> ev;
    ev.init(a, r, o, inv);

    float t_last = near;
    bool continuous = false;
    BrickCache cache;
    cache.id = -1;
    cache.bits = 0;

    for (int i = 0; i + 1 < 2 * G; ++i) {
        int level;
        float seg_lo, seg_hi;
        if (!segment_of(ev, i, G, near, far, level, seg_lo, seg_hi)) continue;
        if (!continuous) {
            if (step_size <= 0.0f) t_last = seg_lo;
            else t_last = lattice_skip(t_last, march_dt(t_last, cone, step_size), seg_lo);
        }
        Dda s;
        dda_setup(s, o, d, inv, seg_lo, seg_hi, a.aabbs + 6 * level, gv.res);
        while (limit <= 0 || sink.n_sm < limit) {
            const float t_cell = fminf(fminf(s.tx, fminf(s.ty, s.tz)), seg_hi);
            if (!occupied(gv, occ, cache, level, s.cx, s.cy, s.cz)) {
                if (step_size <= 0.0f) t_last = t_cell;
                else t_last = lattice_skip(t_last, march_dt(t_last, cone, step_size), t_cell);
                continuous = false;
            } else {
                while (limit <= 0 || sink.n_sm < limit) {
                    float t_next;
                    if (step_size <= 0.0f) t_next = t_cell;
                    else {
                        const float dt = march_dt(t_last, cone, step_size);
                        if (t_last + dt * 0.5f >= t_cell) break;
                        t_next = t_last + dt;
                    }
                    sink.sample(t_last, t_next, continuous);
                    continuous = true;
                    t_last = t_next;
                    if (t_next >= t_cell) break;
                }
            }
            if (!dda_advance(s)) break;
        }
    }
    t_term = t_last;
}

// ---- lattice walk: step_size > 0 and cone_angle == 0 (the training configuration) ----------
// dt is one constant, so (a) consecutive empty voxels are ONE lattice jump to the far side of
// the last one and consecutive occupied voxels ONE batch of samples up to the exit of the last
// one (voxel exit times never decrease, and the reference's per-voxel conditions are monotone
// in t), and (b) jumps and batches have the closed forms of lattice.hpp.  The voxel loop only
// records run boundaries; they are consumed afterwards, all lanes in step.
#ifndef NFA_EVCAP
#define NFA_EVCAP 16
#endif
constexpr int kEvCap = NFA_EVCAP;   // run boundaries buffered per lane and round

template <int EV, bool LDS_OCC>
__device__ __forceinline__ void traverse_ray_lattice(const nfa_traverse_args &a, const GridView &gv, const Occ<LDS_OCC> &occ,
                                                     float *__restrict__ ev_lds /* [kEvCap][blockDim] */,
                                                     int64_t r, bool active, CountSink &sink, float &t_term)
{
    float o[3] = {0.f, 0.f, 0.f}, d[3] = {1.f, 1.f, 1.f};
    float near = 0.f, far = 0.f;
    if (active) {
        o[0] = a.rays_o[3 * r]; o[1] = a.rays_o[3 * r + 1]; o[2] = a.rays_o[3 * r + 2];
        d[0] = a.rays_d[3 * r]; d[1] = a.rays_d[3 * r + 1]; d[2] = a.rays_d[3 * r + 2];
        near = ray_near(a, r);
        far = ray_far(a, r);
    }
    const float inv[3] = {1.0f / d[0], 1.0f / d[1], 1.0f / d[2]};
    const float dt = march_dt(0.0f, 0.0f, a.step_size);     // = clamp(step_size, ., 1e10)
    const int limit = a.traverse_steps_limit;
    const int G = a.n_grids;
    const int nthr = blockDim.x, tid = threadIdx.x;

    Events<EV> ev;
    if (active) ev.init(a, r, o, inv);

    float t_last = near;
    bool continuous = false, finished = !active;
    BrickCache cache;
    cache.id = -1;
    cache.bits = 0;

    for (int i = 0; i + 1 < 2 * G; ++i) {
        int level = 0;
        float seg_lo = 0.f, seg_hi = 0.f;
        bool seg_live = !finished && segment_of(ev, i, G, near, far, level, seg_lo, seg_hi);
        if (seg_live && !continuous) {
            int64_t k; bool stuck;
            t_last = nfa_lattice_until(t_last, dt, seg_lo, &k, &stuck);
            if (stuck) t_last = seg_lo;
        }
        Dda s;
        if (seg_live) dda_setup(s, o, d, inv, seg_lo, seg_hi, a.aabbs + 6 * level, gv.res);
        bool have_run = false, run_occ = false;
        float run_exit = 0.f;
        while (__any(seg_live)) {
            // ---- A: voxel walk, boundaries only
            int n_ev = 0;
            unsigned ev_occ = 0;
            while (seg_live && n_ev < kEvCap - 1) {
                const float t_cell = fminf(fminf(s.tx, fminf(s.ty, s.tz)), seg_hi);
                const bool oc = occupied(gv, occ, cache, level, s.cx, s.cy, s.cz);
                if (have_run && oc != run_occ) {
                    ev_lds[n_ev * nthr + tid] = run_exit;
                    ev_occ |= (run_occ ? 1u : 0u) << n_ev;
                    ++n_ev;
                }
                have_run = true;
                run_occ = oc;
                run_exit = t_cell;
                if (!dda_advance(s)) seg_live = false;
            }
            if (!seg_live && have_run) {            // the segment's last run
                ev_lds[n_ev * nthr + tid] = run_exit;
                ev_occ |= (run_occ ? 1u : 0u) << n_ev;
                ++n_ev;
                have_run = false;
            }
            // ---- B: lattice arithmetic, one boundary per lane per iteration
            int n_max = n_ev;
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) n_max = max(n_max, __shfl_xor(n_max, off, 64));
            for (int j = 0; j < n_max; ++j) {
                if (j >= n_ev || finished) continue;
                const float bound = ev_lds[j * nthr + tid];
                int64_t k; bool stuck;
                const float t_new = nfa_lattice_until(t_last, dt, bound, &k, &stuck);
                if ((ev_occ >> j) & 1u) {
                    if (limit > 0 && sink.n_sm + k >= limit) {        // grid.cu:184,208
                        k = limit - sink.n_sm;
                        sink.run(t_last, k, continuous);
                        t_last = nfa_lattice_advance(t_last, dt, k, nullptr);
                        if (k > 0) continuous = true;
                        finished = true;
                        seg_live = false;
                    } else {
                        sink.run(t_last, k, continuous);
                        if (k > 0) continuous = true;
                        t_last = t_new;
                    }
                } else {
                    continuous = false;
                    t_last = stuck ? bound : t_new;
                }
            }
        }
    }
    t_term = t_last;
}

// per-WAVE reduction of per-ray {edges, samples, overflow rays} -> wave_sums[3 w + {0,1,2}], w = the
// wave's global index (lanes that do not own a ray pass zeros).  One triple per wave instead of
// per workgroup: no LDS, no __syncthreads, so a wave that finished its rays retires at once.
__device__ __forceinline__ void publish_wave_sums(int64_t n_iv, int64_t n_sm, int64_t n_ovf, int64_t *__restrict__ wave_sums) {
    const int64_t w_iv = wave_sum_i64(n_iv), w_sm = wave_sum_i64(n_sm), w_ov = wave_sum_i64(n_ovf);
    if (lane_id() == 0) {
        const int64_t w = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
        wave_sums[3 * w] = w_iv;
        wave_sums[3 * w + 1] = w_sm;
        wave_sums[3 * w + 2] = w_ov;
    }
}

// pass 1.  Block b owns rays [256 b, 256 b + 256).
template <int EV, bool LATTICE, bool LDS_OCC>
__global__ __launch_bounds__(kBlock) void traverse_count_kernel(nfa_traverse_args a, GridView gv,
                                                                int64_t *__restrict__ block_sums, RunStore rs)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const Occ<LDS_OCC> occ = stage_occupancy<LDS_OCC>(gv, smem);
    const int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const bool active = r < a.n_rays && !(a.rays_mask && !a.rays_mask[r]);
    CountSink sink{rs, r, a.n_rays};
    float t_term = 0.f;
    if (LATTICE) {
        // the boundary lists sit behind the occupancy image in LDS
        float *ev_lds = (float *)(smem + occ.bytes);
        traverse_ray_lattice<EV, LDS_OCC>(a, gv, occ, ev_lds, r, active, sink, t_term);
    } else if (active) {
        traverse_ray_general<CountSink, EV, LDS_OCC>(a, gv, occ, r, sink, t_term);
    }
    int64_t ovf = 0;
    if (r < a.n_rays) {
        if (active) {
            ovf = sink.finish(a.step_size > 0.0f) ? 1 : 0;
            if (a.terminate_planes) a.terminate_planes[r] = t_term;
        } else if (rs.n_runs) rs.n_runs[r] = 0;
        if (a.iv_cnts) a.iv_cnts[r] = sink.n_iv;
        a.sm_cnts[r] = sink.n_sm;
    }
    publish_wave_sums(sink.n_iv, sink.n_sm, ovf, block_sums);
}

// ---- split walk: P lanes per ray ---------------------------------------------------------
// With ~10^4 rays per training step a lane-per-ray walk fills only ~200 of the chip's 1024
// SIMDs, one wave each, and its time is the instruction latency of ONE ray's ~400-voxel walk.
// Plane crossings of each axis are chains t <- t + delta as well, so the DDA state after the
// j-th crossing of the ray's major axis has a closed form (lattice.hpp) — the walk can START
// anywhere.  Each ray is cut into P parts at major-axis crossings; a lane walks one part and
// lists its occupied<->empty boundaries; lattice positions of boundaries are absolute (counted
// from the segment start), so every lane resolves its own boundaries independently and the P
// lanes of a ray are stitched with a P-wide shuffle prefix.  One level, cone_angle == 0,
// no step limit (the training configuration); everything else uses the kernels above.

// (time, axis) order of plane crossings in the voxel walk: earlier time first, ties z, y, x
// (the strict '<' chain of utils_grid.cuh:119-141)
__device__ __forceinline__ bool crossing_precedes(float ta, int rank_a, float tb, int rank_b) {
    return (ta < tb) || (ta == tb && rank_a < rank_b);
}

// number of crossings of a chain (first at t0, then +d each) that precede (T, rank_T); also
// returns the time of the first crossing that does not (the pending tdist of that axis)
__device__ __forceinline__ int crossings_before(float t0, float d, int rank, float T, int rank_T, int n_max, float &pending) {
    pending = t0;
    if (n_max <= 0 || !crossing_precedes(t0, rank, T, rank_T)) return 0;
    int i = 1;                          // v = time of crossing i
    float v = t0;
    const float est = (T - t0) / d;
    if (est > 8.0f && est < 1.0e7f) {   // jump close, from below; verified
        int j = (int)est - 2;
        if (j > n_max) j = n_max;
        const float vj = nfa_lattice_advance(t0, d, j - 1, nullptr);
        if (crossing_precedes(vj, rank, T, rank_T)) { i = j; v = vj; }
    }
    while (i < n_max) {
        const float nv = v + d;
        if (!crossing_precedes(nv, rank, T, rank_T)) { pending = nv; return i; }
        v = nv;
        ++i;
    }
    pending = v + d;
    return i;
}

// serial walk of one ray with the lattice arithmetic done inline at every transition: the
// fallback of the split kernel for rays with too many transitions per part or a stuck lattice
template <int EV, bool LDS_OCC>
__device__ void traverse_ray_lattice_inline(const nfa_traverse_args &a, const GridView &gv, const Occ<LDS_OCC> &occ,
                                            int64_t r, CountSink &sink, float &t_term)
{
    const float o[3] = {a.rays_o[3 * r], a.rays_o[3 * r + 1], a.rays_o[3 * r + 2]};
    const float d[3] = {a.rays_d[3 * r], a.rays_d[3 * r + 1], a.rays_d[3 * r + 2]};
    const float inv[3] = {1.0f / d[0], 1.0f / d[1], 1.0f / d[2]};
    const float near = ray_near(a, r), far = ray_far(a, r);
    const float dt = march_dt(0.0f, 0.0f, a.step_size);
    const int G = a.n_grids;
    Events<EV> ev;
    ev.init(a, r, o, inv);
    float t_last = near;
    bool continuous = false;
    BrickCache cache;
    cache.id = -1;
    cache.bits = 0;
    for (int i = 0; i + 1 < 2 * G; ++i) {
        int level;
        float seg_lo, seg_hi;
        if (!segment_of(ev, i, G, near, far, level, seg_lo, seg_hi)) continue;
        int64_t k; bool stuck;
        if (!continuous) { t_last = nfa_lattice_until(t_last, dt, seg_lo, &k, &stuck); if (stuck) t_last = seg_lo; }
        Dda s;
        dda_setup(s, o, d, inv, seg_lo, seg_hi, a.aabbs + 6 * level, gv.res);
        bool have_run = false, run_occ = false, more = true;
        float run_exit = 0.f;
        while (more || have_run) {
            bool oc = false;
            float t_cell = 0.f;
            if (more) {
                t_cell = fminf(fminf(s.tx, fminf(s.ty, s.tz)), seg_hi);
                oc = occupied(gv, occ, cache, level, s.cx, s.cy, s.cz);
            }
            if (have_run && (!more || oc != run_occ)) {           // close the run that ends at run_exit
                const float t_new = nfa_lattice_until(t_last, dt, run_exit, &k, &stuck);
                if (run_occ) { sink.run(t_last, k, continuous); if (k > 0) continuous = true; t_last = t_new; }
                else { continuous = false; t_last = stuck ? run_exit : t_new; }
                have_run = false;
            }
            if (!more) break;
            have_run = true;
            run_occ = oc;
            run_exit = t_cell;
            more = dda_advance(s);
        }
    }
    t_term = t_last;
}

// Voxel walk of one part of a ray (split kernel): `on_boundary(t_exit, run_was_occupied)` is called
// for every occupied<->empty boundary and for the ray's last run; returning false stops the walk.
// `major_done` counts crossings of the ray's major axis; the part ends at crossing j_end.
template <bool LDS_OCC, class F>
__device__ __forceinline__ void walk_part(const GridView &gv, const Occ<LDS_OCC> &occ, BrickCache cache, Dda s, bool live,
                                          bool have_run, bool run_occ, float run_exit, int major_done, int j_end,
                                          int m_rank, float seg_hi, F &&on_boundary)
{
    while (live) {
        const float t_cell = fminf(fminf(s.tx, fminf(s.ty, s.tz)), seg_hi);
        const bool oc = occupied(gv, occ, cache, 0, s.cx, s.cy, s.cz);
        if (have_run && oc != run_occ) {
            if (!on_boundary(run_exit, run_occ)) break;
        }
        have_run = true;
        run_occ = oc;
        run_exit = t_cell;
        const int cm_before = m_rank == 2 ? s.cx : (m_rank == 1 ? s.cy : s.cz);
        const bool cont = dda_advance(s);
        const int cm_after = m_rank == 2 ? s.cx : (m_rank == 1 ? s.cy : s.cz);
        major_done += (cm_after != cm_before) ? 1 : 0;
        if (!cont) {                                   // end of the walk: the ray's last run
            on_boundary(run_exit, run_occ);
            live = false;
        } else if (major_done >= j_end) live = false;  // next part's seam
    }
}

// The same walk with the boundary list as its only consumer (phase A of the split kernel's L2 forms), written for the
// instruction count of its loop body: the candidate boundary is stored to the list's NEXT slot at every voxel — harmless when
// the voxel is no boundary: the slot stays free — and taken with selects; the ray's last run is appended after the loop.
// Same voxels, same list, same flags as walk_part with the list-appending callback.
template <bool LDS_OCC, int CAP, int BLK>
__device__ __forceinline__ void walk_part_list(const GridView &gv, const Occ<LDS_OCC> &occ, BrickCache cache, Dda s, bool live,
                                               bool have_run, bool run_occ, float run_exit, int major_done, int j_end,
                                               int m_rank, float seg_hi, float *__restrict__ ev_lane /* &ev_lds[tid] */,
                                               int &n_ev, unsigned &ev_occ, bool &overflow)
{
    bool ended = false;                       // the walk (not just the part) ended: the ray's last run is a boundary too
    const uint32_t *lc = (const uint32_t *)occ.smem;
    while (live) {
        const float t_cell = fminf(fminf(s.tx, fminf(s.ty, s.tz)), seg_hi);
        bool oc;
        if (LDS_OCC) {
            const int id = (int)__umul24(__umul24(s.cx >> 2, gv.nby) + (s.cy >> 2), gv.nbz) + (s.cz >> 2);
            if (id != cache.id) {
                cache.id = id;
                const uint2 wr = ((const uint2 *)occ.smem)[id >> 5];
                const uint32_t bit = 1u << (id & 31);
                const bool has = (wr.x & bit) != 0u;
                const int k = has ? (int)wr.y + __popc(wr.x & (bit - 1u)) : 0;
                const uint64_t b = ((const uint64_t *)(lc + 2 * occ.w4))[k];
                cache.bits = has ? b : 0ull;
            }
            oc = (cache.bits >> (((s.cx & 3) << 4) | ((s.cy & 3) << 2) | (s.cz & 3))) & 1ull;
        } else {
            oc = occupied(gv, occ, cache, 0, s.cx, s.cy, s.cz);
        }
        const bool is_b = have_run && oc != run_occ;
        const bool room = n_ev < CAP;
        const int slot = room ? n_ev : CAP - 1;
        if (room) ev_lane[slot * BLK] = run_exit;
        overflow = overflow || (is_b && !room);
        ev_occ |= ((is_b && room && run_occ) ? 1u : 0u) << slot;
        n_ev += (is_b && room) ? 1 : 0;
        have_run = true;
        run_occ = oc;
        run_exit = t_cell;
        const int cm_before = m_rank == 2 ? s.cx : (m_rank == 1 ? s.cy : s.cz);
        const bool cont = dda_advance(s);
        const int cm_after = m_rank == 2 ? s.cx : (m_rank == 1 ? s.cy : s.cz);
        major_done += (cm_after != cm_before) ? 1 : 0;
        ended = !cont;
        live = cont && major_done < j_end && !overflow;
    }
    if (ended && !overflow) {                 // end of the walk: the ray's last run
        if (n_ev < CAP) {
            ev_lane[n_ev * BLK] = run_exit;
            ev_occ |= (run_occ ? 1u : 0u) << n_ev;
            ++n_ev;
        } else {
            overflow = true;
        }
    }
}


#ifdef NFA_PHASE_CYCLES
// build-time instrumentation (tools/phase_cycles.py builds with -DNFA_PHASE_CYCLES): shader-clock
// stamps between the phases of the split kernel, kept in registers and stored once per wave at the
// end (one slot per wave, no atomics); read back with nfa_debug_phase_cycles
constexpr int kPhaseSlots = 16384;
__device__ unsigned long long g_phase_cycles[kPhaseSlots][16];
__device__ unsigned long long g_phase_max_wave = 0, g_phase_hist[16] = {0};     // slowest wave; histogram of wave totals in 16 k-cycle bins
__device__ unsigned long long g_phase_slow[16] = {0};                           // phase sums over the waves slower than 60 k cycles ([15] = how many)
#define NFA_PHASE_BEGIN() unsigned long long ph_[16] = {0}; unsigned long long phase_t_ = __builtin_readcyclecounter(); const unsigned long long phase_t0_ = phase_t_
#define NFA_PHASE_MARK(i) do { const unsigned long long now_ = __builtin_readcyclecounter(); ph_[i] = now_ - phase_t_; phase_t_ = now_; } while (0)
#define NFA_PHASE_END()                                                                        \
    do {                                                                                      \
        const int slot_ = (int)blockIdx.x * kWavesPerBlock + (int)(threadIdx.x >> 6);           \
        if (lane_id() == 0) {                                                                 \
            const unsigned long long tot_ = __builtin_readcyclecounter() - phase_t0_;          \
            atomicMax(&g_phase_max_wave, tot_);                                                \
            atomicAdd(&g_phase_hist[tot_ >> 14 > 15 ? 15 : tot_ >> 14], 1ull);                 \
            if (tot_ > 60000ull) { for (int i_ = 0; i_ < 14; ++i_) atomicAdd(&g_phase_slow[i_], ph_[i_]); atomicAdd(&g_phase_slow[15], 1ull); } \
        }                                                                                     \
        if (lane_id() == 0 && slot_ < kPhaseSlots) {                                          \
            ph_[14] = phase_t0_; ph_[15] = 1;                                                  \
            for (int i_ = 0; i_ < 16; ++i_) g_phase_cycles[slot_][i_] += ph_[i_];              \
        }                                                                                     \
    } while (0)
#else
#define NFA_PHASE_MARK(i) do {} while (0)
#define NFA_PHASE_BEGIN() do {} while (0)
#define NFA_PHASE_END() do {} while (0)
#endif

// ---- cross-lane moves inside the P adjacent lanes of a ray (P <= 16: one DPP row).  ds_bpermute shuffles go through
// the LDS crossbar and a lone wave waits for each batch; these stay in the VALU.
// value of the lane Q below (same row); only meaningful where part >= Q
template <int Q>
__device__ __forceinline__ int group_shr_i32(int v) { return __builtin_amdgcn_update_dpp(0, v, kDppRowShr + Q, 0xf, 0xf, false); }
template <int Q>
__device__ __forceinline__ int64_t group_shr_i64(int64_t v) { return dpp_i64<kDppRowShr + Q>(v); }
// inclusive prefix sums over the lanes of a group (part = lane index inside the group)
template <int P>
__device__ __forceinline__ int group_incl_sum_i32(int v, int part) {
    if (P > 1) { const int u = group_shr_i32<1>(v); if (part >= 1) v += u; }
    if (P > 2) { const int u = group_shr_i32<2>(v); if (part >= 2) v += u; }
    if (P > 4) { const int u = group_shr_i32<4>(v); if (part >= 4) v += u; }
    if (P > 8) { const int u = group_shr_i32<8>(v); if (part >= 8) v += u; }
    return v;
}
template <int P>
__device__ __forceinline__ int64_t group_incl_sum_i64(int64_t v, int part) {
    if (P > 1) { const int64_t u = group_shr_i64<1>(v); if (part >= 1) v += u; }
    if (P > 2) { const int64_t u = group_shr_i64<2>(v); if (part >= 2) v += u; }
    if (P > 4) { const int64_t u = group_shr_i64<4>(v); if (part >= 4) v += u; }
    if (P > 8) { const int64_t u = group_shr_i64<8>(v); if (part >= 8) v += u; }
    return v;
}
// the group's P bits of a wave-wide ballot
template <int P>
__device__ __forceinline__ unsigned group_bits(unsigned long long ballot, int group_base) {
    return (unsigned)((ballot >> group_base) & ((1ull << (P < 32 ? P : 32)) - 1ull));     // (groups wider than 32 lanes only use the low bits: <= 15 segments)
}

// XT: the plane-crossing times of the ray's three axes are written out in LDS (lanes 1..3 of the ray walk the x / y / z chains
// with plain adds — exact by construction — n + 1 values each); the walk's end times and every part's seam restart are then
// reads and two binary searches instead of closed forms (512-thread form only: the arrays need 1.5 KB per ray).
template <bool LDS_OCC, int P, int CAP, int BLK = kBlock, bool XT = false>
__global__ __launch_bounds__(BLK) void traverse_count_split_kernel(nfa_traverse_args a, GridView gv,
                                                                   int64_t *__restrict__ block_sums, RunStore rs)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    NFA_PHASE_BEGIN();
    const int tid = threadIdx.x, part = tid % P;
    const int64_t R = a.n_rays;
    const int64_t r = (int64_t)blockIdx.x * (BLK / P) + tid / P;
    const bool ray_ok = r < R;
    const int64_t rr = ray_ok ? r : 0;
    // the ray's loads are requested BEFORE the occupancy image is staged: their L2 round trip overlaps the image's
    const float o[3] = {a.rays_o[3 * rr], a.rays_o[3 * rr + 1], a.rays_o[3 * rr + 2]};
    const float d[3] = {a.rays_d[3 * rr], a.rays_d[3 * rr + 1], a.rays_d[3 * rr + 2]};
    const float near = ray_near(a, rr), far = ray_far(a, rr);
    const Occ<LDS_OCC> occ = stage_occupancy<LDS_OCC>(gv, smem);
    NFA_PHASE_MARK(0);
    float *ev_lds = (float *)(smem + occ.bytes);        // [CAP][BLK] times, then [CAP][BLK] indices
    const float inv[3] = {1.0f / d[0], 1.0f / d[1], 1.0f / d[2]};
    const float dt = march_dt(0.0f, 0.0f, a.step_size);

    // the single segment (grid.cu:129-150 with one level)
    float x0 = 0.f, x1 = 0.f;
    const bool hit = slab_test(o, inv, a.aabbs, -INFINITY, INFINITY, x0, x1);
    const float seg_lo = fmaxf(x0, near), seg_hi = fminf(x1, far);
    const bool live = ray_ok && hit && seg_lo < seg_hi;

    // quantities shared by the P lanes of a ray are computed once and passed around by shuffles
    const int group_base = lane_id() - part;
    int64_t k_tmp; bool stuck_any = false, stuck = false;
    float t_seg = near;

    Dda s;
    s.tx = s.ty = s.tz = 0.f; s.dx = s.dy = s.dz = 0.f;
    s.sx = s.sy = s.sz = 0; s.cx = s.cy = s.cz = 0; s.ox = s.oy = s.oz = 0;
    if (live) dda_setup(s, o, d, inv, seg_lo, seg_hi, a.aabbs, gv.res);
    NFA_PHASE_MARK(1);

    // crossings until each axis reaches its overflow index
    const int nx = s.sx ? (s.ox - s.cx) * s.sx : 1, ny = s.sy ? (s.oy - s.cy) * s.sy : 1, nz = s.sz ? (s.oz - s.cz) * s.sz : 1;
    float Tx, Ty, Tz;      // time of the last crossing of each axis: when the walk ends
    // XT layout of a ray: x crossings at [0, rx], y at [rx + 1, rx + ry + 1], z behind them (n + 1 values per axis)
    const int xt_oy = gv.res[0] + 1, xt_oz = gv.res[0] + gv.res[1] + 2;
    float *xt_ray = nullptr;
    if (XT) xt_ray = (float *)(smem + occ.bytes) + 2 * CAP * BLK + (tid / P) * (gv.res[0] + gv.res[1] + gv.res[2] + 3);
    if (XT) {
        static_assert(!XT || P == 16, "the crossing-time arrays are filled by lanes 1..15 of a ray's group");
        // ONE closed-form call per ray group: lane 0 jumps the lattice from `near` to the segment start, lanes 1..15 jump to
        // the first entry of their fifth of the x / y / z chain (5 lanes per axis); then short plain-add loops: lane 0's last
        // few lattice steps, the others' <= 26 chain entries (exact by construction).
        const float h = dt * 0.5f;
        float adv_t = near, adv_d = dt;
        int64_t adv_j = 0;
        bool jump = false;
        int i_lo = 0, i_hi = 0;
        float *dst = xt_ray;
        if (live && part == 0 && near + h < seg_lo) {          // nfa_lattice_until's verified under-estimate
            const float est = (seg_lo - h - near) / dt;
            if (est > 24.0f && est < 1.0e9f) { const int64_t guess = (int64_t)est; adv_j = guess - nfa_jump_margin(guess); jump = true; }
        }
        if (live && part >= 1) {
            const int ax = (part - 1) / 5, k = (part - 1) % 5;
            adv_t = ax == 0 ? s.tx : (ax == 1 ? s.ty : s.tz);
            adv_d = ax == 0 ? s.dx : (ax == 1 ? s.dy : s.dz);
            int n = ax == 0 ? nx : (ax == 1 ? ny : nz);
            const int cap_n = gv.res[ax];
            n = n < 0 ? 0 : (n > cap_n ? cap_n : n);
            const int L = (n + 1 + 4) / 5;
            i_lo = k * L;
            i_hi = (k + 1) * L < n + 1 ? (k + 1) * L : n + 1;
            adv_j = i_lo;
            jump = i_lo > 0 && i_lo < i_hi;
            dst = xt_ray + (ax == 0 ? 0 : (ax == 1 ? xt_oy : xt_oz));
        }
        NFA_PHASE_MARK(9);
        float adv_v = adv_t;
        if (jump) adv_v = nfa_lattice_advance(adv_t, adv_d, adv_j, nullptr);
        NFA_PHASE_MARK(10);
        if (live && part == 0) {
            float t = near;
            if (jump && adv_v + h < seg_lo) t = adv_v;
            while (t + h < seg_lo) {
                const float nt = t + dt;
                if (nt == t) { stuck_any = true; break; }
                t = nt;
            }
            t_seg = t;
        } else if (live) {
            float t = adv_v;
            for (int i = i_lo; i < i_hi; ++i) { dst[i] = t; t = t + adv_d; }
        }
        NFA_PHASE_MARK(11);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        t_seg = __shfl(t_seg, group_base, 64);
        const bool ok3 = live && nx > 0 && ny > 0 && nz > 0 && nx <= gv.res[0] && ny <= gv.res[1] && nz <= gv.res[2];
        Tx = ok3 ? xt_ray[nx - 1] : 0.0f;
        Ty = ok3 ? xt_ray[xt_oy + ny - 1] : 0.0f;
        Tz = ok3 ? xt_ray[xt_oz + nz - 1] : 0.0f;
    } else if (P >= 4) {
        // FOUR closed-form jumps per ray — the lattice from `near` to the segment start and the
        // last crossing of x, y, z — run as ONE call, on lanes 0..3 of the ray's group (a wave pays
        // for a call once, however many of its lanes are in it).  (Folding the segment-start jump
        // into phase B instead — every part starting its lattice at `near` — was measured slower:
        // all parts then pay the many short binades next to zero.)
        const float h = dt * 0.5f;
        float adv_t = near, adv_d = dt;
        int64_t adv_j = 0;
        bool jump = false;
        if (live && part == 0 && near + h < seg_lo) {          // nfa_lattice_until's verified under-estimate
            const float est = (seg_lo - h - near) / dt;
            if (est > 24.0f && est < 1.0e9f) { const int64_t guess = (int64_t)est; adv_j = guess - nfa_jump_margin(guess); jump = true; }
        }
        if (part >= 1 && part <= 3) {
            adv_t = part == 1 ? s.tx : (part == 2 ? s.ty : s.tz);
            adv_d = part == 1 ? s.dx : (part == 2 ? s.dy : s.dz);
            adv_j = (part == 1 ? nx : (part == 2 ? ny : nz)) - 1;
            jump = live;
        }
        float adv_v = adv_t;
        if (jump) adv_v = nfa_lattice_advance(adv_t, adv_d, adv_j, nullptr);
        if (live && part == 0) {
            float t = near;
            if (jump && adv_v + h < seg_lo) t = adv_v;
            while (t + h < seg_lo) {
                const float nt = t + dt;
                if (nt == t) { stuck_any = true; break; }
                t = nt;
            }
            t_seg = t;
        }
        t_seg = __shfl(t_seg, group_base, 64);
        Tx = __shfl(adv_v, group_base + 1, 64);
        Ty = __shfl(adv_v, group_base + 2, 64);
        Tz = __shfl(adv_v, group_base + 3, 64);
    } else {
        if (live && part == 0) { t_seg = nfa_lattice_until(near, dt, seg_lo, &k_tmp, &stuck); stuck_any = stuck; }
        t_seg = __shfl(t_seg, group_base, 64);
        Tx = nfa_lattice_advance(s.tx, s.dx, nx - 1, nullptr);
        Ty = nfa_lattice_advance(s.ty, s.dy, ny - 1, nullptr);
        Tz = nfa_lattice_advance(s.tz, s.dz, nz - 1, nullptr);
    }
    int end_rank = 2; float T_end = Tx;                                   // ranks: z 0, y 1, x 2
    if (crossing_precedes(Ty, 1, T_end, end_rank)) { T_end = Ty; end_rank = 1; }
    if (crossing_precedes(Tz, 0, T_end, end_rank)) { T_end = Tz; end_rank = 0; }
    // major axis: most crossings
    const int m_rank = (nx >= ny && nx >= nz) ? 2 : (ny >= nz ? 1 : 0);
    const int n_major = m_rank == 2 ? nx : (m_rank == 1 ? ny : nz);
    const int j_begin = (int)(((int64_t)part * n_major) / P);
    const int j_end = (part == P - 1) ? 0x7fffffff : (int)(((int64_t)(part + 1) * n_major) / P);

    // index bookkeeping the closed forms rely on; anything odd (a final voxel "behind" the first
    // one through float error) is left to the serial walk
    NFA_PHASE_MARK(2);
    const bool weird = live && (nx <= 0 || ny <= 0 || nz <= 0 || (XT && (nx > gv.res[0] || ny > gv.res[1] || nz > gv.res[2])));
    // parts whose range is empty do nothing; a part with j_begin == 0 starts at the segment start
    bool part_live = live && !weird && j_begin < j_end;
    bool have_run = false, run_occ = false;
    float run_exit = 0.f;
    BrickCache cache;
    cache.id = -1;
    cache.bits = 0;
    if (part_live && j_begin > 0) {
        const float t0m = m_rank == 2 ? s.tx : (m_rank == 1 ? s.ty : s.tz);
        const float dm = m_rank == 2 ? s.dx : (m_rank == 1 ? s.dy : s.dz);
        const int xt_om = m_rank == 2 ? 0 : (m_rank == 1 ? xt_oy : xt_oz);
        const float T_seam = XT ? xt_ray[xt_om + j_begin - 1]
                                : nfa_lattice_advance(t0m, dm, j_begin - 1, nullptr);    // time of major crossing j_begin
        if (m_rank != end_rank && !crossing_precedes(T_seam, m_rank, T_end, end_rank)) part_live = false;
        else if (XT) {
            // crossings of the two minor axes that precede the seam: lower bounds in their arrays (both searches in one
            // loop of 8 rounds: <= 129 entries), the pending crossing is the entry found
            const bool xm = m_rank == 2, zm = m_rank == 0;
            const float *A1 = xt_ray + (xm ? xt_oy : 0), *A2 = xt_ray + (zm ? xt_oy : xt_oz);
            const int r1 = xm ? 1 : 2, r2 = zm ? 1 : 0;
            const int n1 = xm ? ny : nx, n2 = zm ? ny : nz;
            int lo1 = 0, hi1 = n1, lo2 = 0, hi2 = n2;
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int m1 = (lo1 + hi1) >> 1, m2 = (lo2 + hi2) >> 1;
                const float v1 = A1[m1], v2 = A2[m2];
                if (lo1 < hi1) { if (crossing_precedes(v1, r1, T_seam, m_rank)) lo1 = m1 + 1; else hi1 = m1; }
                if (lo2 < hi2) { if (crossing_precedes(v2, r2, T_seam, m_rank)) lo2 = m2 + 1; else hi2 = m2; }
            }
            const int c1 = lo1, c2 = lo2;
            const float pend1 = A1[c1], pend2 = A2[c2];
            if (!xm) { s.cx += c1 * s.sx; s.tx = pend1; }
            if (xm) { s.cy += c1 * s.sy; s.ty = pend1; }
            if (zm) { s.cy += c2 * s.sy; s.ty = pend2; }
            if (!zm) { s.cz += c2 * s.sz; s.tz = pend2; }
            int px = s.cx, py = s.cy, pz = s.cz;
            if (m_rank == 2) { px += (j_begin - 1) * s.sx; s.cx += j_begin * s.sx; s.tx = T_seam + s.dx; }
            else if (m_rank == 1) { py += (j_begin - 1) * s.sy; s.cy += j_begin * s.sy; s.ty = T_seam + s.dy; }
            else { pz += (j_begin - 1) * s.sz; s.cz += j_begin * s.sz; s.tz = T_seam + s.dz; }
            have_run = true;
            run_occ = occupied(gv, occ, cache, 0, px, py, pz);
            run_exit = fminf(T_seam, seg_hi);
        } else {
            // the two minor axes, picked with selects so that every lane of the wave runs the SAME two
            // closed-form counts whatever its ray's major axis is (three `if (m_rank != k)` blocks made
            // a wave with mixed major axes execute all three): minor 1 is x (y for an x-major ray),
            // minor 2 is z (y for a z-major ray)
            const bool xm = m_rank == 2, zm = m_rank == 0;
            float pend1, pend2;
            const int c1 = crossings_before(xm ? s.ty : s.tx, xm ? s.dy : s.dx, xm ? 1 : 2, T_seam, m_rank, xm ? ny : nx, pend1);
            const int c2 = crossings_before(zm ? s.ty : s.tz, zm ? s.dy : s.dz, zm ? 1 : 0, T_seam, m_rank, zm ? ny : nz, pend2);
            if (!xm) { s.cx += c1 * s.sx; s.tx = pend1; }
            if (xm) { s.cy += c1 * s.sy; s.ty = pend1; }
            if (zm) { s.cy += c2 * s.sy; s.ty = pend2; }
            if (!zm) { s.cz += c2 * s.sz; s.tz = pend2; }
            // the voxel just before the seam: occupancy state the part inherits
            int px = s.cx, py = s.cy, pz = s.cz;
            if (m_rank == 2) { px += (j_begin - 1) * s.sx; s.cx += j_begin * s.sx; s.tx = T_seam + s.dx; }
            else if (m_rank == 1) { py += (j_begin - 1) * s.sy; s.cy += j_begin * s.sy; s.ty = T_seam + s.dy; }
            else { pz += (j_begin - 1) * s.sz; s.cz += j_begin * s.sz; s.tz = T_seam + s.dz; }
            have_run = true;
            run_occ = occupied(gv, occ, cache, 0, px, py, pz);
            run_exit = fminf(T_seam, seg_hi);
        }
    }

    NFA_PHASE_MARK(3);
    // ---- A: this part's voxels, boundaries only (times into the lane's LDS list)
    int n_ev = 0;
    unsigned ev_occ = 0;
    bool overflow = false;
    if (!LDS_OCC) {
        walk_part_list<LDS_OCC, CAP, BLK>(gv, occ, cache, s, part_live, have_run, run_occ, run_exit, j_begin, j_end, m_rank, seg_hi,
                                          ev_lds + tid, n_ev, ev_occ, overflow);
    } else
    walk_part<LDS_OCC>(gv, occ, cache, s, part_live, have_run, run_occ, run_exit, j_begin, j_end, m_rank, seg_hi,
                       [&](float t_exit, bool o) {
                           if (n_ev == CAP) { overflow = true; return false; }
                           ev_lds[n_ev * BLK + tid] = t_exit;
                           ev_occ |= (o ? 1u : 0u) << n_ev;
                           ++n_ev;
                           return true;
                       });
    // a part with more boundaries than its list holds puts its whole ray (all P lanes) into
    // streaming mode: boundaries are resolved as the walk finds them and only aggregates are kept
    // (S1); the run records are written by walking once more when the ray's prefixes are known (S2)
    const bool streaming = group_bits<P>(__ballot(overflow), group_base) != 0u;

    NFA_PHASE_MARK(4);
    // ---- B: absolute lattice position (T_j, K_j = steps from the segment start) of every own
    // boundary; the lists stay in LDS
    int32_t *ev_K = (int32_t *)(ev_lds + CAP * BLK);
    int64_t K_last = 0;
    float T_last = t_seg;
    // streaming aggregates: first boundary kept apart (its samples depend on the previous part)
    int64_t K_first = 0, sm_rest = 0;
    int fresh_rest = 0;
    bool occ_first = false;
    if (!streaming) {
        int64_t K = 0;
        float T = t_seg;
        for (int j = 0; j < n_ev; ++j) {
            const float bound = ev_lds[j * BLK + tid];
            T = nfa_lattice_until(T, dt, bound, &k_tmp, &stuck);
            stuck_any = stuck_any || stuck;
            K += k_tmp;
            ev_lds[j * BLK + tid] = T;
            ev_K[j * BLK + tid] = (int32_t)K;
        }
        K_last = K;
        T_last = T;
    } else {
        int64_t K = 0, K_prev = 0;
        float T = t_seg;
        n_ev = 0;
        auto on_boundary = [&](float t_exit, bool o) {
            int64_t k; bool st;
            T = nfa_lattice_until(T, dt, t_exit, &k, &st);
            stuck_any = stuck_any || st;
            K += k;
            if (n_ev == 0) { K_first = K; occ_first = o; }
            else if (o && K > K_prev) { sm_rest += K - K_prev; ++fresh_rest; }
            K_prev = K;
            ++n_ev;
            return true;
        };
        walk_part<LDS_OCC>(gv, occ, cache, s, part_live, have_run, run_occ, run_exit, j_begin, j_end, m_rank, seg_hi, on_boundary);
        K_last = K;
        T_last = T;
    }
    NFA_PHASE_MARK(5);
    // group-wide decisions (the P lanes of a ray are adjacent lanes of one wave)
    bool bad = stuck_any || weird || K_last > 0x7fffffffll;
#ifdef NFA_FORCE_SERIAL                          // test builds: every ray through the serial in-kernel walk
    bad = true;
#endif
    bad = group_bits<P>(__ballot(bad), group_base) != 0u;
#ifdef NFA_PHASE_CYCLES
    ph_[12] = __popcll(__ballot(bad && ray_ok && part == 0));          // rays of this wave that take the serial walk
    ph_[13] = __popcll(__ballot(streaming && ray_ok && part == 0));    // rays in streaming mode
#endif
    // boundary before this part's first one: the nearest earlier part that has boundaries
    const unsigned ev_parts = group_bits<P>(__ballot(n_ev > 0), group_base);     // bit p: part p of this ray has boundaries
    int64_t K_before = 0;
    float T_before = t_seg;
    {
        const unsigned earlier = ev_parts & ((1u << part) - 1u);
        const int src = group_base + (earlier ? 31 - __clz(earlier) : part);
        const int64_t Kq = __shfl(K_last, src, 64);
        const float Tq = __shfl(T_last, src, 64);
        if (earlier) { K_before = Kq; T_before = Tq; }
    }
    // samples of every occupied run that ends in this part; runs with samples are "fresh"
    // (each is preceded by an empty run or starts the ray: boundaries alternate)
    int64_t n_sm = 0;
    int n_fresh = 0;
    if (!streaming) {
        int64_t K_prev = K_before;
        for (int j = 0; j < n_ev; ++j) {
            const int64_t K = ev_K[j * BLK + tid];
            if (((ev_occ >> j) & 1u) && K > K_prev) { n_sm += K - K_prev; ++n_fresh; }
            K_prev = K;
        }
    } else if (n_ev > 0) {
        n_sm = sm_rest;
        n_fresh = fresh_rest;
        if (occ_first && K_first > K_before) { n_sm += K_first - K_before; ++n_fresh; }
    }
    // exclusive prefixes of fresh runs and samples over the ray's parts, and ray totals
    const int fresh_incl = group_incl_sum_i32<P>(n_fresh, part);
    const int64_t sm_incl = group_incl_sum_i64<P>(n_sm, part);
    const int fresh_before = fresh_incl - n_fresh;
    const int64_t sm_before = sm_incl - n_sm;
    const int fresh_total = __shfl(fresh_incl, group_base + P - 1, 64);
    const int64_t sm_total = __shfl(sm_incl, group_base + P - 1, 64);
    const int last_part_with_ev = ev_parts ? 31 - __clz(ev_parts) : -1;
    const float T_final = __shfl(T_last, group_base + max(last_part_with_ev, 0), 64);

    // run records of this part
    if (!bad && rs.t0 && n_fresh > 0 && fresh_total <= rs.max_runs) {
        if (!streaming) {
            int64_t K_prev = K_before, first = sm_before;
            float T_prev = T_before;
            int idx = fresh_before;
            for (int j = 0; j < n_ev; ++j) {
                const int64_t K = ev_K[j * BLK + tid];
                const float T = ev_lds[j * BLK + tid];
                if (((ev_occ >> j) & 1u) && K > K_prev) {
                    rs.t0[(int64_t)idx * R + r] = T_prev;
                    rs.first[(int64_t)idx * R + r] = (int32_t)first;
                    first += K - K_prev;
                    ++idx;
                }
                K_prev = K;
                T_prev = T;
            }
        } else {                                   // S2: the same walk again, now writing
            int64_t K = 0, K_prev = K_before, first = sm_before;
            float T = t_seg, T_prev = T_before;
            int idx = fresh_before;
            auto on_boundary = [&](float t_exit, bool o) {
                int64_t k; bool st;
                T = nfa_lattice_until(T, dt, t_exit, &k, &st);
                K += k;
                if (o && K > K_prev) {
                    rs.t0[(int64_t)idx * R + r] = T_prev;
                    rs.first[(int64_t)idx * R + r] = (int32_t)first;
                    first += K - K_prev;
                    ++idx;
                }
                K_prev = K;
                T_prev = T;
                return true;
            };
                walk_part<LDS_OCC>(gv, occ, cache, s, part_live, have_run, run_occ, run_exit, j_begin, j_end, m_rank, seg_hi, on_boundary);
        }
    }
    NFA_PHASE_MARK(6);
    int64_t out_iv = 0, out_sm = 0, out_ovf = 0;
    if (!bad) {
        if (ray_ok && part == 0) {
            const bool ovf = fresh_total > rs.max_runs;
            if (rs.n_runs) rs.n_runs[r] = (uint16_t)(ovf ? kRunsOverflow : fresh_total);
            out_iv = sm_total + fresh_total;
            out_sm = sm_total;
            out_ovf = ovf ? 1 : 0;
            if (a.terminate_planes) a.terminate_planes[r] = last_part_with_ev >= 0 ? T_final : t_seg;
        }
    } else if (ray_ok && part == 0) {
        CountSink sink{rs, r, R};
        float t_term = 0.f;
        traverse_ray_lattice_inline<EV_ONE, LDS_OCC>(a, gv, occ, r, sink, t_term);
        out_ovf = sink.finish(true) ? 1 : 0;
        out_iv = sink.n_iv;
        out_sm = sink.n_sm;
        if (a.terminate_planes) a.terminate_planes[r] = t_term;
    }
    if (ray_ok && part == 0) {
        if (a.iv_cnts) a.iv_cnts[r] = out_iv;
        a.sm_cnts[r] = out_sm;
    }
    NFA_PHASE_MARK(7);
    publish_wave_sums(out_iv, out_sm, out_ovf, block_sums);      // this wave's 64 / P rays
    NFA_PHASE_MARK(8);
    NFA_PHASE_END();
}

// ---- segment walk: several levels, one lane per LEVEL SEGMENT of a ray ---------------------------
// A ray through G nested grids is a sequence of up to 2 G - 1 segments, each inside one level (grid.cu:129-150).
// The lane-per-ray walk does them one after the other, every voxel a dependent brick load from L2: its time is
// one ray's ~250-voxel chain whatever the ray count.  Here the P >= 2 G - 1 adjacent lanes of a ray take ONE
// segment each and list its occupied<->empty boundaries.  The marching lattice is one chain t <- t + dt from the
// first live segment's start across all segments (a jump to a later segment's start is the same recurrence), so
// every lane resolves its own boundaries as absolute positions (T, K) on that chain and the lanes of a ray are
// stitched in order.  What a segment adds to the single-level stitch: entering a segment while not `continuous`
// jumps the lattice to its start (grid.cu:157-161) — a virtual empty boundary at seg_lo that only applies in that
// state; and a first occupied run that continues the previous segment's samples starts no new run record.
// cone_angle == 0, no step limit, no ray mask; anything odd (stuck lattice, a segment with more than CAP
// boundaries) goes through the serial walk of the whole ray by the group's first lane.
// K > 1 (round 3; P = 32 lanes per ray, K = 4 per segment slot, up to 4 levels and 4096 rays): a launch that small has lanes to
// spare, and a segment's walk — 100-190 voxels at ~1000 cycles each, more than half of this kernel — is cut into K PARTS at
// crossings of its major axis, as the single-level kernel cuts a ray: three lanes of the slot write the plane-crossing times of
// the segment's x / y / z chains into scratch (`xt`, plain adds: exact by construction), every part then finds its start state
// with reads and two binary searches and inherits the occupancy of the voxel before its seam.  A part's boundaries are positions on
// the ray's ONE chain like a segment's; only the first lane WITH boundaries of a slot applies the jump to the segment's start.
template <bool LDS_OCC, int P, int CAP, int KP = 1>
__global__ __launch_bounds__(kBlock) void traverse_count_segments_kernel(nfa_traverse_args a, GridView gv,
                                                                         int64_t *__restrict__ block_sums, RunStore rs, float *__restrict__ xt)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    NFA_PHASE_BEGIN();
    const Occ<LDS_OCC> occ = stage_occupancy<LDS_OCC>(gv, smem);
    NFA_PHASE_MARK(0);
    float *ev_lds = (float *)(smem + occ.bytes);        // [CAP][kBlock] times, then [CAP][kBlock] lattice indices
    int32_t *ev_K = (int32_t *)(ev_lds + CAP * kBlock);
    const int tid = threadIdx.x, part = tid % P;
    const int group_base = lane_id() - part;
    const int64_t R = a.n_rays;
    const int64_t r = (int64_t)blockIdx.x * (kBlock / P) + tid / P;
    const bool ray_ok = r < R;
    const int64_t rr = ray_ok ? r : 0;
    const int G = a.n_grids;

    const float o[3] = {a.rays_o[3 * rr], a.rays_o[3 * rr + 1], a.rays_o[3 * rr + 2]};
    const float d[3] = {a.rays_d[3 * rr], a.rays_d[3 * rr + 1], a.rays_d[3 * rr + 2]};
    const float inv[3] = {1.0f / d[0], 1.0f / d[1], 1.0f / d[2]};
    const float near = ray_near(a, rr), far = ray_far(a, rr);
    const float dt = march_dt(0.0f, 0.0f, a.step_size);

    Events<EV_MANY> ev;
    ev.init(a, rr, o, inv);
    int level = 0;
    float seg_lo = 0.f, seg_hi = 0.f;
    const int slot = part / KP, sub = part % KP;
    const bool live = ray_ok && slot + 1 < 2 * G && segment_of(ev, slot, G, near, far, level, seg_lo, seg_hi);

    NFA_PHASE_MARK(1);
    // the chain starts at the first live segment
    const unsigned live_parts = group_bits<P>(__ballot(live), group_base);
    const int first_part = live_parts ? __ffs((int)live_parts) - 1 : 0;
    const float lo_first = __shfl(seg_lo, group_base + first_part, 64);
    int64_t k_tmp = 0;
    bool stuck = false, stuck_any = false;
    float t_seg = near;
    if (live_parts) {
        t_seg = nfa_lattice_until(near, dt, lo_first, &k_tmp, &stuck);
        stuck_any = stuck;
    }

    NFA_PHASE_MARK(2);
    constexpr int kSegBatch = 4;
    // the segment's voxel walk: on_boundary(t_exit, run_was_occupied) for every occupied<->empty boundary and for the last run;
    // returning false stops the walk.  kSegBatch voxels per trip: the DDA does not depend on the occupancy, so the steps of a
    // batch run first, their brick words are requested together (one LDS / L2 latency per batch instead of one per voxel:
    // 126 k -> 93 k cycles per wave) and the boundaries are found afterwards, in order.
    // the lane's start state: the segment's first voxel, or (K > 1) the first voxel behind the part's seam
    Dda s0;
    s0.tx = s0.ty = s0.tz = 0.f; s0.dx = s0.dy = s0.dz = 0.f;
    s0.sx = s0.sy = s0.sz = 0; s0.cx = s0.cy = s0.cz = 0; s0.ox = s0.oy = s0.oz = 0;
    if (live) dda_setup(s0, o, d, inv, seg_lo, seg_hi, a.aabbs + 6 * level, gv.res);
    bool part_live = live, have_run0 = false, run_occ0 = false;
    float run_exit0 = 0.f;
    int major0 = 0, j_end = 0x7fffffff, m_rank = 0;
    if (KP > 1) {
        const int nx = s0.sx ? (s0.ox - s0.cx) * s0.sx : 1, ny = s0.sy ? (s0.oy - s0.cy) * s0.sy : 1, nz = s0.sz ? (s0.oz - s0.cz) * s0.sz : 1;
        const bool regular = live && nx > 0 && ny > 0 && nz > 0 && nx <= gv.res[0] && ny <= gv.res[1] && nz <= gv.res[2];
        const int oy_ = gv.res[0] + 1, oz_ = gv.res[0] + gv.res[1] + 2;
        float *const A = xt + ((int64_t)blockIdx.x * (kBlock / KP) + tid / KP) * (gv.res[0] + gv.res[1] + gv.res[2] + 3);
        if (regular && sub < 3) {                          // the chain of axis `sub`: entry i = time of its crossing i
            float t = sub == 0 ? s0.tx : (sub == 1 ? s0.ty : s0.tz);
            const float dd = sub == 0 ? s0.dx : (sub == 1 ? s0.dy : s0.dz);
            const int na = sub == 0 ? nx : (sub == 1 ? ny : nz);
            float *dst = A + (sub == 0 ? 0 : (sub == 1 ? oy_ : oz_));
            for (int i = 0; i <= na; ++i) { dst[i] = t; t = t + dd; }
        }
        __threadfence_block();
        __builtin_amdgcn_wave_barrier();                   // (the K lanes of a slot are lanes of one wave)
        if (regular) {
            const float Tx = A[nx - 1], Ty = A[oy_ + ny - 1], Tz = A[oz_ + nz - 1];       // the walk ends with the earliest of these
            int end_rank = 2; float T_end = Tx;                                       // ranks: z 0, y 1, x 2
            if (crossing_precedes(Ty, 1, T_end, end_rank)) { T_end = Ty; end_rank = 1; }
            if (crossing_precedes(Tz, 0, T_end, end_rank)) { T_end = Tz; end_rank = 0; }
            m_rank = (nx >= ny && nx >= nz) ? 2 : (ny >= nz ? 1 : 0);
            const int n_major = m_rank == 2 ? nx : (m_rank == 1 ? ny : nz);
            const int j_begin = (int)(((int64_t)sub * n_major) / KP);
            j_end = (sub == KP - 1) ? 0x7fffffff : (int)(((int64_t)(sub + 1) * n_major) / KP);
            major0 = j_begin;
            part_live = j_begin < j_end;
            if (part_live && j_begin > 0) {
                const float T_seam = A[(m_rank == 2 ? 0 : (m_rank == 1 ? oy_ : oz_)) + j_begin - 1];   // time of major crossing j_begin
                if (m_rank != end_rank && !crossing_precedes(T_seam, m_rank, T_end, end_rank)) part_live = false;   // the walk ends before this seam
                else {
                    const bool xm = m_rank == 2, zm = m_rank == 0;
                    const float *A1 = A + (xm ? oy_ : 0), *A2 = A + (zm ? oy_ : oz_);
                    const int r1 = xm ? 1 : 2, r2 = zm ? 1 : 0;
                    const int n1 = xm ? ny : nx, n2 = zm ? ny : nz;
                    int lo1 = 0, hi1 = n1, lo2 = 0, hi2 = n2;
#pragma unroll
                    for (int it = 0; it < 8; ++it) {
                        const int m1 = (lo1 + hi1) >> 1, m2 = (lo2 + hi2) >> 1;
                        const float v1 = A1[m1], v2 = A2[m2];
                        if (lo1 < hi1) { if (crossing_precedes(v1, r1, T_seam, m_rank)) lo1 = m1 + 1; else hi1 = m1; }
                        if (lo2 < hi2) { if (crossing_precedes(v2, r2, T_seam, m_rank)) lo2 = m2 + 1; else hi2 = m2; }
                    }
                    const float pend1 = A1[lo1], pend2 = A2[lo2];
                    if (!xm) { s0.cx += lo1 * s0.sx; s0.tx = pend1; }
                    if (xm) { s0.cy += lo1 * s0.sy; s0.ty = pend1; }
                    if (zm) { s0.cy += lo2 * s0.sy; s0.ty = pend2; }
                    if (!zm) { s0.cz += lo2 * s0.sz; s0.tz = pend2; }
                    // the voxel just before the seam: the run state the part inherits
                    int px = s0.cx, py = s0.cy, pz = s0.cz;
                    if (m_rank == 2) { px += (j_begin - 1) * s0.sx; s0.cx += j_begin * s0.sx; s0.tx = T_seam + s0.dx; }
                    else if (m_rank == 1) { py += (j_begin - 1) * s0.sy; s0.cy += j_begin * s0.sy; s0.ty = T_seam + s0.dy; }
                    else { pz += (j_begin - 1) * s0.sz; s0.cz += j_begin * s0.sz; s0.tz = T_seam + s0.dz; }
                    BrickCache cache;
                    cache.id = -1;
                    cache.bits = 0;
                    have_run0 = true;
                    run_occ0 = occupied(gv, occ, cache, level, px, py, pz);
                    run_exit0 = fminf(T_seam, seg_hi);
                }
            }
        } else {
            part_live = live && sub == 0;                  // odd index bookkeeping: the slot's first lane walks the whole segment
        }
    }
    auto walk = [&](auto &&on_boundary) {
        Dda s = s0;
        bool have_run = have_run0, run_occ = run_occ0, stop = false, ended = true;
        float run_exit = run_exit0;
        int major_done = major0;
        const uint32_t *lc = (const uint32_t *)occ.smem;
        for (bool more = true; more;) {
            bool valid[kSegBatch];
            float tc[kSegBatch];
            int id[kSegBatch], bp[kSegBatch];
#pragma unroll
            for (int k = 0; k < kSegBatch; ++k) {
                valid[k] = more;
                tc[k] = fminf(fminf(s.tx, fminf(s.ty, s.tz)), seg_hi);
                id[k] = (int)__umul24(__umul24(s.cx >> 2, gv.nby) + (s.cy >> 2), gv.nbz) + (s.cz >> 2) + level * gv.bricks_per_grid;
                bp[k] = ((s.cx & 3) << 4) | ((s.cy & 3) << 2) | (s.cz & 3);
                if (more) {
                    if (KP > 1) {
                        const int cm_before = m_rank == 2 ? s.cx : (m_rank == 1 ? s.cy : s.cz);
                        more = dda_advance(s);
                        const int cm_after = m_rank == 2 ? s.cx : (m_rank == 1 ? s.cy : s.cz);
                        major_done += (cm_after != cm_before) ? 1 : 0;
                        if (more && major_done >= j_end) { more = false; ended = false; }     // the next part's seam: the run stays open
                    } else {
                        more = dda_advance(s);
                    }
                }
            }
            uint64_t bits[kSegBatch];
            if (LDS_OCC) {
                uint2 wr[kSegBatch];
#pragma unroll
                for (int k = 0; k < kSegBatch; ++k) wr[k] = valid[k] ? ((const uint2 *)occ.smem)[id[k] >> 5] : make_uint2(0u, 0u);
#pragma unroll
                for (int k = 0; k < kSegBatch; ++k) {
                    const uint32_t bit = 1u << (id[k] & 31);
                    bits[k] = (wr[k].x & bit) ? ((const uint64_t *)(lc + 2 * occ.w4))[(int)wr[k].y + __popc(wr[k].x & (bit - 1u))] : 0ull;
                }
            } else if (occ.bytes > 0) {
                uint32_t w[kSegBatch];
#pragma unroll
                for (int k = 0; k < kSegBatch; ++k) w[k] = valid[k] ? lc[id[k] >> 5] : 0u;
#pragma unroll
                for (int k = 0; k < kSegBatch; ++k) bits[k] = (w[k] & (1u << (id[k] & 31))) ? gv.bricks[id[k]] : 0ull;
            } else {
#pragma unroll
                for (int k = 0; k < kSegBatch; ++k) bits[k] = valid[k] ? gv.bricks[id[k]] : 0ull;
            }
#pragma unroll
            for (int k = 0; k < kSegBatch; ++k) {
                if (valid[k] && !stop) {
                    const bool oc = (bits[k] >> bp[k]) & 1ull;
                    if (have_run && oc != run_occ) stop = !on_boundary(run_exit, run_occ);
                    have_run = true;
                    run_occ = oc;
                    run_exit = tc[k];
                }
            }
            if (stop) more = false;
        }
        if (!stop && ended) on_boundary(run_exit, run_occ);    // the segment's last run
    };

    // ---- A: this segment's boundaries into the lane's list; a segment with more than CAP of them is STREAMED: its boundaries
    // are resolved as a second walk finds them (aggregates only) and its run records written by a third one
    int n_ev = 0;
    unsigned ev_occ = 0;
    bool streaming = false;
    if (part_live)
        walk([&](float t_exit, bool oc) {
            if (n_ev == CAP) { streaming = true; return false; }
            ev_lds[n_ev * kBlock + tid] = t_exit;
            ev_occ |= (oc ? 1u : 0u) << n_ev;
            ++n_ev;
            return true;
        });

    NFA_PHASE_MARK(3);
    // ---- B: positions on the chain: the segment start (the virtual boundary), then the own boundaries
    float T_lo = t_seg, T_last = t_seg;
    int64_t K_lo = 0, K_last = 0;
    int64_t sm_rest = 0;                 // samples / fresh runs of boundaries 1.. (each preceded by an empty boundary of this segment)
    int fresh_rest = 0;
    int64_t K_first = 0;
    bool cont_rest = false, occ_first = false;
    if (part_live) {
        T_lo = nfa_lattice_until(t_seg, dt, seg_lo, &K_lo, &stuck);
        stuck_any = stuck_any || stuck;
        float T = T_lo;
        int64_t K = K_lo, K_prev = K_lo;
        int j = 0;
        auto resolve = [&](float bound, bool oj) {
            // A voxel exit BEFORE the segment's start (a ray lying in a bounding plane of a level: its slab test returns an
            // infinite exit, the level's segment outlasts the box, and the next segment's first voxel lies behind the ray) has
            // no position relative to this segment's start; where the chain stands then depends on the segments before it.
            // Such rays take the serial walk (tests/golden/k2_inplane.npz pins them).
            stuck_any = stuck_any || bound < seg_lo;
            T = nfa_lattice_until(T, dt, bound, &k_tmp, &stuck);
            stuck_any = stuck_any || stuck;
            K += k_tmp;
            if (j == 0) { K_first = K; occ_first = oj; }
            else if (oj && K > K_prev) { sm_rest += K - K_prev; ++fresh_rest; cont_rest = true; }
            else if (!oj) cont_rest = false;
            K_prev = K;
            ++j;
        };
        if (!streaming) {
            for (int q = 0; q < n_ev; ++q) {
                resolve(ev_lds[q * kBlock + tid], (ev_occ >> q) & 1u);
                ev_lds[q * kBlock + tid] = T;
                ev_K[q * kBlock + tid] = (int32_t)K;
            }
        } else {
            walk([&](float t_exit, bool oc) { resolve(t_exit, oc); return true; });
            n_ev = j;
        }
        K_last = K;
        T_last = T;
    }
    NFA_PHASE_MARK(4);
    bool bad = stuck_any || K_last > 0x7fffffffll;
#ifdef NFA_FORCE_SERIAL
    bad = true;
#endif
    bad = group_bits<P>(__ballot(bad), group_base) != 0u;
#ifdef NFA_PHASE_CYCLES
    ph_[12] = __popcll(__ballot(bad && ray_ok && part == 0));          // rays of this wave that take the serial walk
    ph_[13] = __popcll(__ballot(streaming));                           // streamed segments
#endif

    // ---- stitch, segment by segment: (position, continuous) before every part
    const bool has = part_live && n_ev > 0;
    // the jump to a segment's start (entering it while not continuous) belongs to the first lane WITH boundaries of its slot
    const unsigned has_lanes = group_bits<P>(__ballot(has), group_base);
    const unsigned slot_lanes = ((1u << KP) - 1u) << (slot * KP);
    const bool enters = has && (has_lanes & slot_lanes & ((1u << part) - 1u)) == 0u;
    int Kpos = 0;
    float Tpos = t_seg;
    bool cont = false, any_has = false;
    int64_t sm_acc = 0;
    int fresh_acc = 0;
    int my_K_start = 0, my_fresh_before = 0;
    float my_T_start = t_seg;
    bool my_cont_in = false;
    int64_t my_sm_before = 0;
#pragma unroll
    for (int p = 0; p < P; ++p) {
        const int src = group_base + p;
        const bool has_p = __shfl((int)has, src, 64) != 0;
        const int Klo_p = __shfl((int)K_lo, src, 64), Kf_p = __shfl((int)K_first, src, 64), Kl_p = __shfl((int)K_last, src, 64);
        const float Tlo_p = __shfl(T_lo, src, 64), Tl_p = __shfl(T_last, src, 64);
        const int flags_p = __shfl((occ_first ? 1 : 0) | (cont_rest ? 2 : 0) | (n_ev >= 2 ? 4 : 0) | (enters ? 8 : 0), src, 64);
        const int64_t smr_p = __shfl(sm_rest, src, 64);
        const int frr_p = __shfl(fresh_rest, src, 64);
        if (part == p) { my_sm_before = sm_acc; my_fresh_before = fresh_acc; my_cont_in = cont; }
        if (has_p) {
            int Ks = Kpos;
            float Ts = Tpos;
            if ((flags_p & 8) && !cont && Klo_p > Kpos) { Ks = Klo_p; Ts = Tlo_p; }      // entering the segment: jump to its start
            const bool of = flags_p & 1;
            const int k1 = of && Kf_p > Ks ? Kf_p - Ks : 0;
            const bool fresh1 = k1 > 0 && !cont;
            if (part == p) { my_K_start = Ks; my_T_start = Ts; }
            sm_acc += k1 + smr_p;
            fresh_acc += (fresh1 ? 1 : 0) + frr_p;
            if (of) { if (k1 > 0) cont = true; } else cont = false;
            if (flags_p & 4) cont = (flags_p & 2) != 0;
            Kpos = Kl_p;
            Tpos = Tl_p;
            any_has = true;
        }
    }
    const int64_t sm_total = sm_acc;
    const int fresh_total = fresh_acc;

    NFA_PHASE_MARK(5);
    // run records of this segment
    if (!bad && rs.t0 && has && fresh_total <= rs.max_runs) {
        int64_t first = my_sm_before;
        int64_t K_prev = my_K_start;
        float T_prev = my_T_start;
        int idx = my_fresh_before, j = 0;
        auto record = [&](int64_t K, float T, bool oj) {
            if (oj && K > K_prev) {
                if (j > 0 || !my_cont_in) {
                    rs.t0[(int64_t)idx * R + r] = T_prev;
                    rs.first[(int64_t)idx * R + r] = (int32_t)first;
                    ++idx;
                }
                first += K - K_prev;
            }
            K_prev = K > K_prev ? K : K_prev;
            T_prev = T;
            ++j;
        };
        if (!streaming) {
            for (int q = 0; q < n_ev; ++q) record(ev_K[q * kBlock + tid], ev_lds[q * kBlock + tid], (ev_occ >> q) & 1u);
        } else {
            float T = T_lo;
            int64_t K = K_lo;
            walk([&](float t_exit, bool oc) {
                int64_t k; bool st;
                T = nfa_lattice_until(T, dt, t_exit, &k, &st);
                K += k;
                record(K, T, oc);
                return true;
            });
        }
    }
    NFA_PHASE_MARK(6);
    int64_t out_iv = 0, out_sm = 0, out_ovf = 0;
    if (!bad) {
        if (ray_ok && part == 0) {
            const bool ovf = fresh_total > rs.max_runs || sm_total > 0x7fffffffll;
            if (rs.n_runs) rs.n_runs[r] = (uint16_t)(ovf ? kRunsOverflow : fresh_total);
            out_iv = sm_total + fresh_total;
            out_sm = sm_total;
            out_ovf = ovf && sm_total > 0 ? 1 : 0;
            if (a.terminate_planes) a.terminate_planes[r] = any_has ? Tpos : near;
        }
    } else if (ray_ok && part == 0) {
        CountSink sink{rs, r, R};
        float t_term = 0.f;
        traverse_ray_lattice_inline<EV_MANY, LDS_OCC>(a, gv, occ, r, sink, t_term);
        out_ovf = sink.finish(true) ? 1 : 0;
        out_iv = sink.n_iv;
        out_sm = sink.n_sm;
        if (a.terminate_planes) a.terminate_planes[r] = t_term;
    }
    if (ray_ok && part == 0) {
        if (a.iv_cnts) a.iv_cnts[r] = out_iv;
        a.sm_cnts[r] = out_sm;
    }
    NFA_PHASE_MARK(7);
    publish_wave_sums(out_iv, out_sm, out_ovf, block_sums);
    NFA_PHASE_MARK(8);
    NFA_PHASE_END();
}

#include "cone_walk.hpp"

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

// pass 1b: offsets.  The count kernels publish one {edges, samples, overflow} triple per
// workgroup of `rays_per_sum` rays; every block here first adds up the triples before its own
// 256 rays (<= R / rays_per_sum values, L2-resident), then scans its 256 counts.  The last
// block stores the totals.
__global__ __launch_bounds__(kBlock) void traverse_offsets_kernel(
    const int64_t *__restrict__ iv_cnts, int64_t *__restrict__ iv_starts,
    const int64_t *__restrict__ sm_cnts, int64_t *__restrict__ sm_starts,
    int64_t n_rays, const int64_t *__restrict__ block_sums, int sums_per_block, int64_t n_sums,
    int64_t *__restrict__ totals, int64_t *__restrict__ totals_dev, int64_t stamp)
{
    __shared__ int64_t lds[kWavesPerBlock];
    __shared__ int64_t base[2];
    const int b = blockIdx.x;
    const bool last = b == (int)gridDim.x - 1;
    // every load of the kernel is requested up front (the host polls for this kernel's stamp: its last block's chain of
    // dependent round trips is the sampling call's latency — three of them before, one now)
    const int64_t r = (int64_t)b * kBlock + threadIdx.x;
    const bool in = r < n_rays;
    const int64_t c_iv = (iv_cnts && in) ? iv_cnts[r] : 0;
    const int64_t c_sm = in ? sm_cnts[r] : 0;
    int64_t p0 = 0, p1 = 0, ov = 0;
    const int64_t before = (int64_t)b * sums_per_block;
    for (int64_t j = threadIdx.x; j < before; j += kBlock) { p0 += block_sums[3 * j]; p1 += block_sums[3 * j + 1]; }
    int64_t ed = 0;
    if (last)                                     // rays that need the pass-2 re-traversal; the call's edges (also without iv_cnts: the emit
        for (int64_t j = threadIdx.x; j < n_sums; j += kBlock) { ov += block_sums[3 * j + 2]; ed += block_sums[3 * j]; }      // pass reads runs = edges - samples)
    int64_t t0, t1;
    block_excl_scan_i64(p0, lds, t0);
    block_excl_scan_i64(p1, lds, t1);
    if (threadIdx.x == 0) { base[0] = t0; base[1] = t1; }
    __syncthreads();
    int64_t tot;
    if (iv_cnts) {
        const int64_t e = block_excl_scan_i64(c_iv, lds, tot);
        if (in) iv_starts[r] = base[0] + e;
    }
    {
        const int64_t e = block_excl_scan_i64(c_sm, lds, tot);
        if (in) sm_starts[r] = base[1] + e;
        if (last && threadIdx.x == 0) { totals[1] = base[1] + tot; totals_dev[1] = base[1] + tot; }
    }
    if (last) {
        int64_t tov, ted;
        block_excl_scan_i64(ov, lds, tov);
        block_excl_scan_i64(ed, lds, ted);
        if (threadIdx.x == 0) {
            totals[0] = ted; totals_dev[0] = ted;
            totals[2] = tov; totals_dev[2] = tov; totals_dev[3] = 0;
            // totals[3]: the caller's completion stamp, stored LAST and behind a system-scope fence — a host that polls this word
            // in (coherent) pinned memory may read the three totals as soon as it sees the stamp (nfa_traverse_offsets_stamped)
            __threadfence_system();
            totals[3] = stamp;
        }
    }
}

// pass 2, general form: walk the grid again and write.  Also the single pass of the
// over-allocated test-time mode (grid.cu:375).  only_overflow: just the rays pass 1 flagged.
template <int EV, bool LDS_OCC>
__global__ __launch_bounds__(kBlock) void traverse_fill_kernel(nfa_traverse_args a, GridView gv,
                                                               int skip_empty, int rewrite_counts,
                                                               const uint16_t *__restrict__ only_overflow)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const Occ<LDS_OCC> occ = stage_occupancy<LDS_OCC>(gv, smem);
    const int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (r >= a.n_rays) return;
    if (a.rays_mask && !a.rays_mask[r]) return;
    if (only_overflow && only_overflow[r] != kRunsOverflow) return;
    if (skip_empty) {
        if (a.iv_cnts && a.iv_cnts[r] == 0) return;
        if (a.sm_cnts[r] == 0) return;
    }
    FillSink sink{a, r, a.iv_starts ? a.iv_starts[r] : 0, a.sm_starts[r]};
    float t_term;
    traverse_ray_general<FillSink, EV, LDS_OCC>(a, gv, occ, r, sink, t_term);
    if (a.terminate_planes && !only_overflow) a.terminate_planes[r] = t_term;
    if (rewrite_counts) {
        if (a.iv_cnts) a.iv_cnts[r] = sink.n_iv;
        a.sm_cnts[r] = sink.n_sm;
    }
}

// pass 2, fast form: ONE LANE PER OUTPUT SAMPLE.  sample s -> ray (binary search in the
// exclusive offsets) -> run (binary search in the runs' first-sample indices) -> lattice point (closed form) -> coalesced
// stores of ray_indices / t_starts / t_ends (+ interval edges when asked for).
// n_dev != NULL: speculative launch (before the host knows the total): the total comes from n_dev[1] and a launch whose
// outputs (sized `n_samples` = the caller's guess) are too small does nothing — the caller launches again with the right size.
__device__ __forceinline__ void emit_by_samples(const nfa_traverse_args &a, const RunStore &rs, int64_t n_samples)
{
    const float step_size = a.step_size, cone = a.cone_angle;
    const int64_t R = a.n_rays;
    __shared__ int64_t s_span[2];
    for (int64_t s0 = (int64_t)blockIdx.x * kBlock; s0 < n_samples; s0 += (int64_t)gridDim.x * kBlock) {
        // the workgroup's 256 consecutive samples belong to a narrow range of rays: two lanes
        // search the whole offset array for the first and the last sample, everyone else only
        // that range (a dozen rays for NeRF-like rays: 4 dependent loads instead of log2 R)
        const int64_t s_last = (s0 + kBlock - 1 < n_samples ? s0 + kBlock - 1 : n_samples - 1);
        if (threadIdx.x < 128) {
            // waves 0 and 1 look for the ray of the first / last sample with a 64-ary search: every round is ONE memory
            // round trip for 64 probes (3 rounds for 10^5 rays) instead of the ~log2 R dependent loads of a bisection —
            // those were this kernel's critical path
            const int lane = lane_id();
            const int64_t target = threadIdx.x < 64 ? s0 : s_last;
            int64_t lo = 0, hi = R;                   // invariant: sm_starts[lo - 1] <= target (or lo == 0), sm_starts[hi] > target (or hi == R)
            while (hi - lo > 0) {
                const int64_t span = hi - lo;
                const int64_t stride = (span + 63) >> 6;
                const int64_t m = lo + (int64_t)lane * stride;                      // probes lo, lo + stride, ...
                const bool le = m < hi && a.sm_starts[m] <= target;
                const unsigned long long b = __ballot(le);                          // a prefix of ones (ascending offsets)
                const int k = __popcll(b);                                          // probes that are <= target
                if (k == 0) { hi = lo; break; }
                const int64_t base = lo + (int64_t)(k - 1) * stride;                // last probe <= target
                lo = base + 1;
                const int64_t nh = base + stride;
                if (nh < hi) hi = nh;
            }
            if (lane == 0) s_span[threadIdx.x >> 6] = lo - 1;
        }
        __syncthreads();
        const int64_t r_first = s_span[0], r_last = s_span[1];
        __syncthreads();
        const int64_t s = s0 + threadIdx.x;
        if (s >= n_samples) continue;
        int64_t lo = r_first, hi = r_last + 1;
        while (lo < hi) { const int64_t m = lo + ((hi - lo) >> 1); if (a.sm_starts[m] <= s) lo = m + 1; else hi = m; }
        const int64_t r = lo - 1;
        const int n_runs = rs.n_runs[r];
        if (n_runs == kRunsOverflow) continue;        // written by the fallback launch
        int64_t j = s - a.sm_starts[r];
        int qlo = 1, qhi = n_runs;                    // last run whose first sample is <= j (run 0 starts at 0)
        while (qlo < qhi) { const int m = qlo + ((qhi - qlo) >> 1); if ((int64_t)rs.first[(int64_t)m * R + r] <= j) qlo = m + 1; else qhi = m; }
        const int q = qlo - 1;
        if (q > 0) j -= rs.first[(int64_t)q * R + r];
        float t0 = rs.t0[(int64_t)q * R + r];
        if (cone == 0.0f) t0 = nfa_lattice_advance(t0, march_dt(t0, cone, step_size), j, nullptr);
        else for (int64_t k = 0; k < j; ++k) t0 = t0 + march_dt(t0, cone, step_size);
        const float t1 = t0 + march_dt(t0, cone, step_size);
        if (a.sm_vals) a.sm_vals[s] = (t1 + t0) * 0.5f;
        if (a.sm_ray_indices) a.sm_ray_indices[s] = r;
        if (a.sm_is_valid) a.sm_is_valid[s] = 1;
        if (a.t_starts) { a.t_starts[s] = t0; a.t_ends[s] = t1; }
        if (a.iv_vals) {
            // edge layout of a ray: every run contributes len + 1 edges (grid.cu:219-245)
            const int64_t e_right = a.iv_starts[r] + (s - a.sm_starts[r]) + q + 1;
            a.iv_vals[e_right] = t1; a.iv_ray_indices[e_right] = r; a.iv_is_right[e_right] = 1;
            a.iv_is_left[e_right - 1] = 1;
            if (j == 0) { a.iv_vals[e_right - 1] = t0; a.iv_ray_indices[e_right - 1] = r; }
        }
    }
}

// pass 2, ray-group form (round 3): 16 LANES PER RAY walk the ray's run records in order — no searches.  The sample-parallel
// form above is bound by its chain of ~14 dependent loads per sample (0.9 TB/s of stores at any size: 666 us for the 38 M
// samples of 10^6 rays); here a ray costs two round trips (its counts and offsets, then its runs) and every 16 samples one
// pass of 15 predicated adds: lane k of a group holds the run's lattice point after k steps, the next pass starts from lane
// 15's end — the same sequential float adds the reference performs, so exact for any cone angle (the sample-parallel form
// needs the closed form for cone_angle = 0 and j adds per sample otherwise).  Adjacent groups take adjacent rays: their
// loads coalesce and their stores fill one contiguous stretch of the outputs.
struct EmitRay {           // what a group of 16 lanes needs of a ray: ONE round trip (every load is independent of the others)
    int64_t cnt, S, E;
    int nr;
    float run_t0;          // lane k: run k of the ray (garbage beyond the ray's runs, never used)
    int run_first, run_next;
};
__device__ __forceinline__ EmitRay emit_ray_load(const nfa_traverse_args &a, const RunStore &rs, int64_t r, int gl) {
    EmitRay m;
    const int64_t R = a.n_rays;
    m.cnt = a.sm_cnts[r];
    m.nr = rs.n_runs[r];
    m.S = a.sm_starts[r];
    m.E = a.iv_vals ? a.iv_starts[r] : 0;
    m.run_t0 = 0.0f; m.run_first = 0; m.run_next = 0;
    if (gl < rs.max_runs) {
        m.run_t0 = rs.t0[(int64_t)gl * R + r];
        m.run_first = gl > 0 ? rs.first[(int64_t)gl * R + r] : 0;
        if (gl + 1 < rs.max_runs) m.run_next = rs.first[(int64_t)(gl + 1) * R + r];
    }
    return m;
}

__device__ __forceinline__ void emit_by_ray_groups(const nfa_traverse_args &a, const RunStore &rs)
{
    constexpr int G = 16;
    const float step_size = a.step_size, cone = a.cone_angle;
    const int64_t R = a.n_rays;
    const int lane = lane_id(), gl = lane & (G - 1), gbase = lane & ~(G - 1);
    const int64_t n_groups = (int64_t)gridDim.x * (kBlock / G);
    int64_t r = (int64_t)blockIdx.x * (kBlock / G) + threadIdx.x / G;
    if (r >= R) return;
    EmitRay m = emit_ray_load(a, rs, r, gl);
    for (; r < R; r += n_groups) {
        const EmitRay c = m;
        if (r + n_groups < R) m = emit_ray_load(a, rs, r + n_groups, gl);      // the next ray's round trip overlaps this ray's stores
        if (c.cnt <= 0) continue;                  // (a masked ray recorded nothing)
        const int nr = c.nr;
        if (nr == kRunsOverflow) continue;         // written by the fallback launch
        const int64_t S = c.S, E = c.E;
        for (int q0 = 0; q0 < nr; q0 += G) {
            const int q = q0 + gl;                 // the group's lanes hold 16 runs at once
            float run_t0 = c.run_t0;
            int run_first = c.run_first, run_end = q + 1 < nr ? c.run_next : (int)c.cnt;
            if (q0 > 0 && q < nr) {                // (a ray with more than 16 runs)
                run_t0 = rs.t0[(int64_t)q * R + r];
                run_first = rs.first[(int64_t)q * R + r];
                run_end = q + 1 < nr ? rs.first[(int64_t)(q + 1) * R + r] : (int)c.cnt;
            }
            const int nq = nr - q0 < G ? nr - q0 : G;
            for (int i = 0; i < nq; ++i) {
                float base = __shfl(run_t0, gbase + i, 64);
                const int first = __shfl(run_first, gbase + i, 64);
                const int len = __shfl(run_end, gbase + i, 64) - first;
                const float dt0 = march_dt(base, cone, step_size);
                // A pass of a group covers 64 samples, FOUR CONSECUTIVE ONES PER LANE: every lane runs the pass's sequential adds (the
                // only chain from one pass to the next: no shuffle) in blocks of 16, keeps the value after its own 4 gl steps and
                // takes four more steps for its own samples — 63 + 4 adds, 15 selects and four 16-byte stores per 64 samples
                // (the first form, one sample per lane and 16 per pass, spent 45 instructions per 16: 12.9 -> 7 us on the
                // bench's longest ray).  Blocks beyond the run's end are skipped (group-uniform).
                for (int j0 = 0; j0 < len; j0 += 4 * G) {
                    const int rem = len - j0;
                    float t = base, full = base;
                    float sv[5];
                    auto chain = [&](auto step) {            // (instantiated for the constant step and for the cone's clamp)
#pragma unroll
                        for (int blk = 0; blk < 4; ++blk) {
                            if (blk == 0 || 16 * blk < rem) {
#pragma unroll
                                for (int l = 4 * blk; l < 4 * blk + 4; ++l) {
                                    if (l > 0) { full = step(step(step(step(full)))); t = gl >= l ? full : t; }
                                }
                            }
                        }
                        if (rem > 4 * G) base = step(step(step(step(full))));
                        sv[0] = t;
#pragma unroll
                        for (int e = 0; e < 4; ++e) sv[e + 1] = step(sv[e]);
                    };
                    if (cone == 0.0f) chain([&](float x) { return x + dt0; });
                    else chain([&](float x) { return x + march_dt(x, cone, step_size); });
                    const int j = j0 + 4 * gl;              // this lane's first sample of the pass
                    const int nv = rem - 4 * gl;            // its samples that exist (>= 4: all)
                    if (nv <= 0) continue;
                    const int64_t s = S + first + j;
                    if (nv >= 4 && !a.iv_vals) {
                        typedef float vf4 __attribute__((ext_vector_type(4)));
                        if (a.t_starts) {
                            const vf4 v0 = {sv[0], sv[1], sv[2], sv[3]}, v1 = {sv[1], sv[2], sv[3], sv[4]};
                            __builtin_memcpy(a.t_starts + s, &v0, 16);
                            __builtin_memcpy(a.t_ends + s, &v1, 16);
                        }
                        if (a.sm_vals) {
                            const vf4 m = {(sv[1] + sv[0]) * 0.5f, (sv[2] + sv[1]) * 0.5f, (sv[3] + sv[2]) * 0.5f, (sv[4] + sv[3]) * 0.5f};
                            __builtin_memcpy(a.sm_vals + s, &m, 16);
                        }
                        if (a.sm_ray_indices) {
                            const int64_t rr[4] = {r, r, r, r};
                            __builtin_memcpy(a.sm_ray_indices + s, rr, 32);
                        }
                        if (a.sm_is_valid) { const uint32_t ones = 0x01010101u; __builtin_memcpy(a.sm_is_valid + s, &ones, 4); }
                        continue;
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if (e >= nv) break;
                        const float t0 = sv[e], t1 = sv[e + 1];
                        if (a.sm_vals) a.sm_vals[s + e] = (t1 + t0) * 0.5f;
                        if (a.sm_ray_indices) a.sm_ray_indices[s + e] = r;
                        if (a.sm_is_valid) a.sm_is_valid[s + e] = 1;
                        if (a.t_starts) { a.t_starts[s + e] = t0; a.t_ends[s + e] = t1; }
                        if (a.iv_vals) {
                            // edge layout of a ray: every run contributes len + 1 edges (grid.cu:219-245)
                            const int64_t e_right = E + first + j + e + (q0 + i) + 1;
                            a.iv_vals[e_right] = t1; a.iv_ray_indices[e_right] = r; a.iv_is_right[e_right] = 1;
                            a.iv_is_left[e_right - 1] = 1;
                            if (j + e == 0) { a.iv_vals[e_right - 1] = t0; a.iv_ray_indices[e_right - 1] = r; }
                        }
                    }
                }
            }
        }
    }
}

// pass 2: ONE launch, the form chosen ON THE DEVICE from the totals the offsets kernel left in the workspace (the speculative
// launch runs before the host has seen them).  16 lanes per ray pay per RUN (~25 instructions + a pass per 64 samples), a lane per
// sample pays ~14 dependent loads per SAMPLE: the ray groups win on long runs and lose on a grid of alternating voxels (the
// reference's `rand > 0.5` test grid: 271 samples per ray in ~130 runs — 113 vs 77 us at 4 k rays when the choice looked at the
// sample count alone, profiles/r03_count_pass.md).  runs = edges - samples.  With a cone angle the sample-parallel form re-runs a
// sample's chain from its run's start, so the ray groups take over at much shorter runs.
//   hint: 0 = choose, 1 = ray groups, 2 = lane per sample (NFA_EMIT).  speculative: outputs hold `capacity` samples — a launch whose
//   outputs are too small does nothing, the caller launches again with the right size.
__global__ __launch_bounds__(kBlock) void traverse_emit_kernel(nfa_traverse_args a, RunStore rs, int64_t capacity,
                                                               const int64_t *__restrict__ n_dev, int speculative, int hint)
{
    const int64_t n_ed = n_dev[0], n_sm = n_dev[1];
    if (speculative && n_sm > capacity) return;
    const int64_t runs = n_ed - n_sm > 0 ? n_ed - n_sm : 1;
    const bool long_runs = a.cone_angle != 0.0f ? n_sm >= 8 * runs : (n_sm >= 900000 && n_sm >= 20 * runs);
    if (hint == 1 || (hint == 0 && long_runs)) emit_by_ray_groups(a, rs);
    else emit_by_samples(a, rs, speculative ? n_sm : capacity);
}

// generic exclusive sum of int64 counts (data_spec.hpp:86-106), single workgroup of 1024:
// rounds of 1024 coalesced elements with a running carry.  Used for the per-ray count arrays
// of the over-allocated traversal mode and the tile counts of the visibility compaction.
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

// hierarchical form of the same scan for long arrays (nfa_exclusive_sum_i64): see the host function
constexpr int64_t kScanSingleMax = 8192;
__global__ __launch_bounds__(kBlock) void excl_sum_chunks_kernel(const int64_t *__restrict__ cnts, int64_t n, int64_t chunk,
                                                                 int64_t *__restrict__ starts)
{
    __shared__ int64_t lds[kWavesPerBlock];
    const int64_t lo = (int64_t)blockIdx.x * chunk, hi = lo + chunk < n ? lo + chunk : n;
    int64_t carry = 0;
    for (int64_t base = lo; base < hi; base += kBlock) {
        const int64_t i = base + threadIdx.x;
        const int64_t v = i < hi ? cnts[i] : 0;
        int64_t tot;
        const int64_t ex = block_excl_scan_i64(v, lds, tot);
        if (i < hi && i != lo) starts[i] = carry + ex;
        carry += tot;
    }
    if (threadIdx.x == 0) starts[lo] = carry;            // the chunk's total, parked in the slot whose local value is 0
}
__global__ __launch_bounds__(1024) void excl_sum_parked_kernel(int64_t *__restrict__ starts, int64_t chunk, int64_t nb,
                                                               int64_t *__restrict__ total)
{
    __shared__ int64_t wsum[16];
    __shared__ int64_t carry_s;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    const int lane = lane_id(), wave = threadIdx.x >> 6;
    for (int64_t base = 0; base < nb; base += 1024) {
        const int64_t b = base + threadIdx.x;
        const int64_t v = b < nb ? starts[b * chunk] : 0;
        int64_t inc = v;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int64_t u = __shfl_up(inc, off, 64);
            if (lane >= off) inc += u;
        }
        if (lane == 63) wsum[wave] = inc;
        __syncthreads();
        int64_t woff = 0, tot = 0;
        for (int w = 0; w < 16; ++w) { const int64_t x = wsum[w]; if (w < wave) woff += x; tot += x; }
        const int64_t carry = carry_s;
        if (b < nb) starts[b * chunk] = carry + woff + inc - v;
        __syncthreads();
        if (threadIdx.x == 0) carry_s = carry + tot;
        __syncthreads();
    }
    if (threadIdx.x == 0 && total) *total = carry_s;
}
__global__ __launch_bounds__(kBlock) void excl_sum_add_kernel(int64_t *__restrict__ starts, int64_t n, int64_t chunk)
{
    const int64_t lo = (int64_t)blockIdx.x * chunk, hi = lo + chunk < n ? lo + chunk : n;
    const int64_t off = starts[lo];
    for (int64_t i = lo + 1 + threadIdx.x; i < hi; i += kBlock) starts[i] += off;
}

int validate_traverse(const nfa_traverse_args *a) {
    NFA_REQUIRE(a != nullptr, "traverse: args is NULL");
    NFA_REQUIRE(a->n_rays >= 0, "traverse: n_rays < 0");
    NFA_REQUIRE(a->n_grids >= 1 && a->n_grids <= NFA_MAX_GRID_LEVELS, "traverse: n_grids=%d not in [1,%d]", a->n_grids, NFA_MAX_GRID_LEVELS);
    NFA_REQUIRE(a->res[0] > 0 && a->res[1] > 0 && a->res[2] > 0, "traverse: bad resolution");
    if (a->n_rays == 0) return NFA_OK;
    NFA_REQUIRE(a->rays_o && a->rays_d && a->bricks && a->aabbs, "traverse: NULL input");
    const int given = (a->hits != nullptr) + (a->t_sorted != nullptr) + (a->t_indices != nullptr);
    NFA_REQUIRE(given == 0 || given == 3, "traverse: hits/t_sorted/t_indices must be given together");
    NFA_REQUIRE(a->sm_cnts && a->sm_starts, "traverse: sm_cnts/sm_starts are required");
    NFA_REQUIRE((a->iv_cnts == nullptr) == (a->iv_starts == nullptr), "traverse: iv_cnts/iv_starts go together");
    return NFA_OK;
}

// LDS of the traversal kernels: occupancy image (bitmap + rank + compacted bricks) followed by
// the per-lane boundary lists.  The image is sized from the caller's count of non-empty bricks
// when known (args.n_nonempty_bricks >= 0, nfa_pack_binaries' header read back once per grid
// update) so that several workgroups fit per CU; otherwise from the budget.
constexpr int kLdsBudget = 96 * 1024;
constexpr int kLdsPerCU = 160 * 1024;
constexpr int kEvBytes = kEvCap * kBlock * 4 * 2;   // boundary times + lattice indices

GridView make_view(const nfa_traverse_args *a, int ev_bytes, int *lds_bytes, int budget = kLdsBudget) {
    GridView gv;
    const PackedLayout L = packed_layout(a->n_grids, a->res[0], a->res[1], a->res[2]);
    gv.bricks = a->bricks;
    gv.header = (const int64_t *)(a->bricks + L.off_header);
    gv.coarse = (const uint32_t *)(a->bricks + L.off_coarse);
    gv.prefix = (const uint32_t *)(a->bricks + L.off_prefix);
    gv.compact = a->bricks + L.off_compact;
    for (int k = 0; k < 3; ++k) gv.res[k] = a->res[k];
    gv.nbx = (a->res[0] + 3) / 4;
    gv.nby = (a->res[1] + 3) / 4;
    gv.nbz = (a->res[2] + 3) / 4;
    gv.bricks_per_grid = gv.nbx * gv.nby * gv.nbz;
    gv.n_words = (int)L.n_words;
    const int64_t w4 = ((int64_t)L.n_words + 3) & ~3ll;
    const int64_t words_bytes = 2 * w4 * 4;
    const int64_t room = (int64_t)budget - ev_bytes - 16 - words_bytes;
    // full LDS image only when the caller told us how many bricks are non-empty and they all fit
    const int64_t need = a->n_nonempty_bricks >= 0 ? (a->n_nonempty_bricks > 0 ? a->n_nonempty_bricks : 1) : -1;
    if (need > 0 && room >= need * 8) {
        gv.lds_words = (int)L.n_words;
        gv.lds_compact_cap = (int)need;
        *lds_bytes = (int)(((words_bytes + need * 8 + 15) & ~15ll) + ev_bytes);
    } else if (w4 * 4 <= 16 * 1024 && (int64_t)budget - ev_bytes - 16 >= w4 * 4) {
        // bricks from L2; a small bitmap of non-empty bricks (<= 16 KiB: up to 128 K bricks) is
        // still staged and answers the empty bricks
        gv.lds_words = (int)L.n_words;
        gv.lds_compact_cap = 0;
        *lds_bytes = (int)(((w4 * 4 + 15) & ~15ll) + ev_bytes);
    } else {                                      // everything from L2
        gv.lds_words = 0;
        gv.lds_compact_cap = 0;
        *lds_bytes = ev_bytes;
    }
    return gv;
}

// workspace layout (bytes): [ block_sums: 3 int64 per count workgroup ][ run t0: max_runs*R f32 ][ run first: max_runs*R i32 ][ n_runs: R u16 ]
// one triple per wave; the finest granularity any count kernel publishes is 1 ray per wave (the cone kernel's P = 64)
inline int64_t ws_block_sums_bytes(int64_t n_rays) { return 24 * ((n_rays > 0 ? n_rays : 1) + 2 * kWavesPerBlock); }
// [ block sums | run records | n_runs ] [ totals: 4 int64, the device copy of what nfa_traverse_offsets stores in args.totals ]
inline int64_t ws_totals_offset(int64_t n_rays) {
    const int64_t R = n_rays > 0 ? n_rays : 1;
    return ws_block_sums_bytes(R) + (int64_t)run_capacity(R) * R * 8 + ceil_div(2 * R, 16) * 16;
}
// cone_angle != 0 (cone_walk.hpp): [ ... totals ][ voxel records: count workgroups * (rx + ry + rz) * kBlock u32 ], 256-byte aligned
inline int64_t ws_voxels_offset(int64_t n_rays) { return (ws_totals_offset(n_rays) + 4 * (int64_t)sizeof(int64_t) + 255) & ~255ll; }
// lanes per ray: one per level segment (2 G - 1 of them) would do, but (i) the chain phase runs on ONE lane per ray and a wave
// pays the longest of its rays at every voxel, and (ii) with 32 / 64 lanes per ray a segment is walked by 4 / 8 lanes (parts,
// cone_walk.hpp) — so small launches get more lanes per ray, as long as the waves still find free SIMDs
// (measured, 4 x 128^3, cone 0.004, sample_occgrid end to end, us; profiles/r03_cone.md):
//   rays      8 lanes   16     32     64
//   1 024       -      314     -     192
//   2 048      341     309     -     185
//   4 096      344     328    229    236
//   8 192      373     305    299    382
//  16 384      347     393    499    636
inline int cone_lanes_for_levels(int n_grids, int64_t n_rays) {
    if (n_grids == 1) return 1;
    int P = 2 * n_grids - 1 <= 8 ? 8 : 16;
    if (n_rays <= 2048) P = 64;
    else if (n_rays <= 8192) P = 32;
    if (const char *e = getenv("NFA_CONE_P")) { const int v = atoi(e); if ((v == 8 && 2 * n_grids - 1 <= 8) || v == 16 || v == 32 || v == 64) P = v; }
    return P;
}
inline int64_t cone_voxel_bytes(const nfa_traverse_args *a) {
    const int P = cone_lanes_for_levels(a->n_grids, a->n_rays);
    const int64_t nb = ceil_div(a->n_rays > 0 ? a->n_rays : 1, kBlock / P);
    int64_t bytes = nb * (int64_t)(a->res[0] + a->res[1] + a->res[2] + 8 /* VoxelStore::kSlack */) * kBlock * 4;
    if (P >= 32) bytes += nb * (kBlock / 4) * (int64_t)(a->res[0] + a->res[1] + a->res[2] + 3) * 4 + 256;      // crossing-time arrays (K >= 4 lanes per slot)
    return bytes;
}
// lanes per ray of the cone count pass, 0 = the general lane-per-ray kernel.  Needs the larger workspace
// (nfa_traverse_workspace_bytes_for) announced through args.workspace_bytes.
static int cone_lanes_per_ray(const nfa_traverse_args *a) {
    if (!(a->step_size > 0.0f) || a->cone_angle == 0.0f) return 0;
    if (a->t_sorted || a->traverse_steps_limit > 0 || a->rays_mask) return 0;
    const int64_t max_rays = 32768;       // beyond, a lane per ray fills the chip (and the voxel planes grow with the ray count)
    if (const char *e = getenv("NFA_CONE")) { if (atoi(e) == 0) return 0; }
    if (a->n_rays > max_rays) return 0;
    if (a->workspace_bytes < ws_voxels_offset(a->n_rays) + cone_voxel_bytes(a)) return 0;
    return cone_lanes_for_levels(a->n_grids, a->n_rays);
}

RunStore make_runs(void *workspace, int64_t n_rays) {
    RunStore rs;
    uint8_t *p = (uint8_t *)workspace + ws_block_sums_bytes(n_rays);
    rs.max_runs = run_capacity(n_rays);
    rs.t0 = (float *)p;
    rs.first = (int32_t *)(p + (int64_t)rs.max_runs * n_rays * 4);
    rs.n_runs = (uint16_t *)(p + (int64_t)rs.max_runs * n_rays * 8);
    return rs;
}

// raise a kernel's dynamic-LDS limit once per (kernel, size) — it is a host-side driver call
template <class K>
int allow_lds(K kernel, int bytes) {
    if (bytes <= 48 * 1024) return NFA_OK;
    static std::mutex mu;
    static std::unordered_map<const void *, int> granted;
    std::lock_guard<std::mutex> lock(mu);
    int &g = granted[(const void *)kernel];
    if (bytes > g) {
        hipError_t e = hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        if (e != hipSuccess) return fail(NFA_ERR_LAUNCH, "hipFuncSetAttribute(%d B LDS): %s", bytes, hipGetErrorString(e));
        g = bytes;
    }
    return NFA_OK;
}

}  // namespace
}  // namespace nfa

using namespace nfa;

NFA_EXPORT const char *nfa_version(void) { return "nerfacc_hip 0.3.0 gfx950"; }
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
    return packed_layout(n_grids, rx, ry, rz).total_words;
}

static int rank_and_compact(uint64_t *bricks, const PackedLayout &L, hipStream_t s) {
    uint32_t *coarse = (uint32_t *)(bricks + L.off_coarse);
    uint32_t *prefix = (uint32_t *)(bricks + L.off_prefix);
    hipLaunchKernelGGL(rank_bricks_kernel, dim3(1), dim3(1024), 0, s, coarse, L.n_words, prefix, (int64_t *)(bricks + L.off_header));
    if (int rc = check_launch("rank_bricks_kernel")) return rc;
    hipLaunchKernelGGL(compact_bricks_kernel, dim3(blocks_for(L.n_bricks)), dim3(kBlock), 0, s,
                       bricks, L.n_bricks, coarse, prefix, bricks + L.off_compact);
    return check_launch("compact_bricks_kernel");
}

NFA_EXPORT int nfa_pack_binaries(const uint8_t *binaries, int32_t n_grids, int32_t rx, int32_t ry, int32_t rz,
                                 uint64_t *bricks, void *stream)
{
    NFA_REQUIRE(n_grids > 0 && rx > 0 && ry > 0 && rz > 0, "pack_binaries: empty grid");
    NFA_REQUIRE(n_grids <= NFA_MAX_GRID_LEVELS, "pack_binaries: n_grids=%d > %d (the header holds one count per level)", n_grids, NFA_MAX_GRID_LEVELS);
    NFA_REQUIRE(binaries && bricks, "pack_binaries: NULL pointer");
    const PackedLayout L = packed_layout(n_grids, rx, ry, rz);
    NFA_REQUIRE(L.n_bricks < (1ll << 24), "pack_binaries: grid too large (more than 2^24 bricks)");
    hipStream_t s = (hipStream_t)stream;
    uint32_t *coarse = (uint32_t *)(bricks + L.off_coarse);
    int64_t *header = (int64_t *)(bricks + L.off_header);
    if (hipMemsetAsync(header, 0, 12 * sizeof(int64_t), s) != hipSuccess) return fail(NFA_ERR_LAUNCH, "pack_binaries: memset failed");
    hipLaunchKernelGGL(pack_bricks_kernel<false>, dim3(blocks_for(L.n_bricks)), dim3(kBlock), 0, s,
                       binaries, n_grids, rx, ry, rz, (rx + 3) / 4, (ry + 3) / 4, (rz + 3) / 4, bricks, coarse, header + 1,
                       (const float *)nullptr, (const double *)nullptr, 0, 0.0f, (uint8_t *)nullptr, (float *)nullptr);
    if (int rc = check_launch("pack_bricks_kernel")) return rc;
    return rank_and_compact(bricks, L, s);
}

// nfa_grid_threshold + nfa_pack_binaries in four launches instead of five: the pass that compares the occupancies
// with the threshold writes the bool grid AND the bricks (the bool grid is not read back)
NFA_EXPORT int nfa_grid_threshold_packed(const float *occs, int32_t n_grids, int32_t rx, int32_t ry, int32_t rz, float occ_thre,
                                         void *workspace, uint8_t *binaries, float *threshold_out, uint64_t *bricks, void *stream)
{
    NFA_REQUIRE(n_grids > 0 && rx > 0 && ry > 0 && rz > 0, "grid_threshold_packed: empty grid");
    NFA_REQUIRE(n_grids <= NFA_MAX_GRID_LEVELS, "grid_threshold_packed: n_grids=%d > %d (the header holds one count per level)", n_grids, NFA_MAX_GRID_LEVELS);
    NFA_REQUIRE(occs && workspace && binaries && bricks, "grid_threshold_packed: NULL pointer");
    const PackedLayout L = packed_layout(n_grids, rx, ry, rz);
    NFA_REQUIRE(L.n_bricks < (1ll << 24), "grid_threshold_packed: grid too large (more than 2^24 bricks)");
    hipStream_t s = (hipStream_t)stream;
    uint32_t *coarse = (uint32_t *)(bricks + L.off_coarse);
    int64_t *header = (int64_t *)(bricks + L.off_header);
    if (hipMemsetAsync(header, 0, 12 * sizeof(int64_t), s) != hipSuccess) return fail(NFA_ERR_LAUNCH, "grid_threshold_packed: memset failed");
    const int64_t n_cells = (int64_t)n_grids * rx * ry * rz;
    const int nb = launch_grid_mean_partials(occs, n_cells, (double *)workspace, s);
    hipLaunchKernelGGL(pack_bricks_kernel<true>, dim3(blocks_for(L.n_bricks)), dim3(kBlock), 0, s,
                       (const uint8_t *)nullptr, n_grids, rx, ry, rz, (rx + 3) / 4, (ry + 3) / 4, (rz + 3) / 4, bricks, coarse, header + 1,
                       occs, (const double *)workspace, nb, occ_thre, binaries, threshold_out);
    if (int rc = check_launch("pack_bricks_kernel<occs>")) return rc;
    return rank_and_compact(bricks, L, s);
}

NFA_EXPORT int64_t nfa_traverse_workspace_bytes(int64_t n_rays) {
    const int64_t R = n_rays > 0 ? n_rays : 1;
    return ws_totals_offset(R) + 4 * (int64_t)sizeof(int64_t);
}

static int segment_lanes_per_ray(const nfa_traverse_args *a);
static int64_t seg_parts_bytes_fwd(const nfa_traverse_args *a);
NFA_EXPORT int64_t nfa_traverse_workspace_bytes_for(const nfa_traverse_args *a) {
    if (!a) return 0;
    const int64_t base = nfa_traverse_workspace_bytes(a->n_rays);
    if (a->n_rays <= 0 || a->n_grids < 1 || a->n_grids > NFA_MAX_GRID_LEVELS) return base;
    nfa_traverse_args probe = *a;
    probe.workspace_bytes = INT64_MAX;
    if (cone_lanes_per_ray(&probe)) return ws_voxels_offset(a->n_rays) + cone_voxel_bytes(a);
    if (segment_lanes_per_ray(&probe) == 32) return ws_voxels_offset(a->n_rays) + seg_parts_bytes_fwd(a);
    return base;
}

// lanes per ray of the count pass for this call (1 = lane-per-ray kernels).  `sparse`: the full
// occupancy image fits in LDS (few non-empty bricks: a blob-like grid, few occupied<->empty
// boundaries per ray).
static int count_lanes_per_ray(const nfa_traverse_args *a, bool sparse) {
    const bool lattice = a->step_size > 0.0f && a->cone_angle == 0.0f;
    const bool split = lattice && !a->t_sorted && a->n_grids == 1 && a->traverse_steps_limit <= 0 && a->rays_mask == nullptr;
    int P = 1;
    if (split) {
        // measured on MI355X (profiles/r01_split_sweep.md): splitting pays while the ray batch is
        // too small to fill the chip with one lane per ray; beyond ~40 k rays the lane-per-ray
        // walk does the same job in fewer instructions.  Dense / noisy grids have a boundary every
        // few voxels, so their parts are kept shorter (more lanes per ray).
        if (sparse) {
            // profiles/r02_split_sweep.md, r03_count_pass.md (bench steady state: ~190 voxels and 40 samples per ray): 16 lanes
            // per ray up to 16 k rays, 8 up to 96 k, lane-per-ray beyond; 4 and 2 lanes per ray never win on a sparse grid
            if (a->n_rays <= 16384) P = 16;
            else if (a->n_rays <= 98304) P = 8;
        } else {
            if (a->n_rays <= 8192) P = 16;
            else if (a->n_rays <= 16384) P = 8;
            else if (a->n_rays <= 36864) P = 4;
            else if (a->n_rays <= 65536) P = 2;
        }
        if (const char *e = getenv("NFA_SPLIT_P")) {          // tuning knob: 1, 2, 4, 8 or 16
            const int v = atoi(e);
            if (v == 1 || v == 2 || v == 4 || v == 8 || v == 16) P = v;
            if (sparse && (P == 2 || P == 4)) P = 8;          // (no 2- / 4-lane instances for sparse grids: they never won)
        }
    }
    return P;
}

// split kernels: a part keeps its boundaries in LDS, CAP of them (8 B each per lane).  With the
// sparse occupancy image in LDS (blob-like grid) 16 (8 at P = 16) is plenty; otherwise the grid may
// be dense or noisy — a boundary every other voxel for the reference's rand > 0.5 test grid — and
// LDS is free of the image, so the lists get 32 entries (the width of the lane's mask register).
struct SplitPlan { int P, cap, lds, blk, xt, seg, l2; GridView gv; };
// several levels: one lane per level segment (traverse_count_segments_kernel) while the batch is too small to fill the chip
// with a lane per ray — measured on 4 x 128^3 (profiles/r02_microbench.md): 125 vs 235 us at 1 k rays, 124 vs 267 at 4 k,
// 169 vs 291 at 16 k, 309 vs 320 at 32 k, 566 vs 387 at 65 k (count pass alone).  NFA_SEGMENTS = 0 switches it off
// crossing-time arrays of the segment kernel's parts: one (rx + ry + rz + 3)-float array per segment slot, 8 slots per ray
inline int64_t seg_parts_bytes(const nfa_traverse_args *a) {
    return ceil_div(a->n_rays > 0 ? a->n_rays : 1, kBlock / 32) * (kBlock / 4) * (int64_t)(a->res[0] + a->res[1] + a->res[2] + 3) * 4 + 256;
}
static int64_t seg_parts_bytes_fwd(const nfa_traverse_args *a) { return seg_parts_bytes(a); }
static int segment_lanes_per_ray(const nfa_traverse_args *a) {
    const bool lattice = a->step_size > 0.0f && a->cone_angle == 0.0f;
    if (!lattice || a->t_sorted || a->n_grids < 2 || a->traverse_steps_limit > 0 || a->rays_mask) return 0;
    const int64_t max_rays = 40960;     // (r03: with the ray-group emit pass behind it the call is 213 vs 285 us at 32 k rays, 383 vs 312 at 65 k)
    if (const char *e = getenv("NFA_SEGMENTS")) { if (atoi(e) == 0) return 0; }
    if (a->n_rays > max_rays) return 0;
    int P = 2 * a->n_grids - 1 <= 8 ? 8 : 16;
    // 32 lanes per ray = 4 per segment slot (parts) while the launch has lanes to spare and the caller's workspace holds the
    // crossing-time arrays (nfa_traverse_workspace_bytes_for); NFA_SEG_P = 8 | 32 overrides
    const bool room = a->workspace_bytes >= ws_voxels_offset(a->n_rays) + seg_parts_bytes(a);
    if (P == 8 && room && a->n_rays <= 4096) P = 32;
    if (const char *e = getenv("NFA_SEG_P")) { const int v = atoi(e); if (v == 8 && P == 32) P = 8; else if (v == 32 && P == 8 && room) P = 32; }
    return P;
}
static SplitPlan plan_split(const nfa_traverse_args *a) {
    SplitPlan p;
    p.seg = 0;
    p.l2 = 0;
    if (const int pc = cone_lanes_per_ray(a)) {
        p.P = pc;
        p.seg = 2;                  // cone_walk.hpp
        p.blk = kBlock;
        p.xt = 0;
        p.cap = 0;
        p.lds = 0;
        // the per-lane segment table (12 B per lane) sits behind the occupancy image; the image only when it leaves room for
        // several workgroups per CU (the walk is bound by its instructions, not by where the brick words come from)
        p.gv = make_view(a, kBlock * 12, &p.lds, 40 * 1024);
        return p;
    }
    if (const int ps = segment_lanes_per_ray(a)) {
        p.P = ps;
        p.seg = 1;
        p.blk = kBlock;
        p.xt = 0;
        p.lds = 0;
        // 32-entry boundary lists always (a segment with more is streamed: three walks); the brick image joins them in LDS when
        // it fits in what a workgroup alone on its CU can have — the lists alone already keep a second workgroup off the CU
        p.cap = 32;
        // (beyond one workgroup per CU the image stays in L2 so that two workgroups share a CU's 160 KB: the walk is as fast
        // from L2 — it is bound by its instructions — and 16 k rays take 145 instead of 208 us)
        const bool one_round = ceil_div(a->n_rays, kBlock / ps) <= kNumCU;
        p.gv = make_view(a, p.cap * kBlock * 8, &p.lds, one_round ? 156 * 1024 : 80 * 1024);
        return p;
    }
    p.P = count_lanes_per_ray(a, true);
    p.cap = 16;      // (8-entry lists at P = 16 overflow into the streaming mode on a trained scene: 80 us instead of 44)
    p.lds = 0;
    p.blk = kBlock;
    p.xt = 0;
    if (p.P <= 1) return p;
    // 512-thread workgroups for the 16-lane variant from 3 k rays (one workgroup per CU, 32 rays share one staged grid
    // image: 37.7 vs 39.6 us at 6.5 k rays; below ~3 k rays the 256-thread form spreads over more CUs and wins);
    // NFA_SPLIT_BLK = 256 | 512 overrides
    if (p.P == 16 && p.cap == 16 && a->n_rays >= 3072) p.blk = 512;
    if (const char *e = getenv("NFA_SPLIT_BLK")) { const int v = atoi(e); if (v == 256 || (v == 512 && p.P == 16 && p.cap == 16)) p.blk = v; }
    // crossing-time arrays (512-thread form only; NFA_SPLIT_XT = 0 switches them off)
    p.xt = p.blk == 512 && !(getenv("NFA_SPLIT_XT") && atoi(getenv("NFA_SPLIT_XT")) == 0);
    const int xt_bytes = p.xt ? (p.blk / p.P) * (a->res[0] + a->res[1] + a->res[2] + 3) * 4 : 0;
    // (the 512-thread form is alone on its CU: it may take the whole 160 KB)
    const int budget = p.blk == 512 ? 156 * 1024 : kLdsBudget;
    p.gv = make_view(a, p.cap * p.blk * 8 + xt_bytes, &p.lds, budget);
    if (p.gv.lds_compact_cap == 0 && p.xt) { p.xt = 0; p.gv = make_view(a, p.cap * p.blk * 8, &p.lds, budget); }
    if (p.gv.lds_compact_cap == 0 && p.blk != kBlock) { p.blk = kBlock; p.gv = make_view(a, p.cap * kBlock * 8, &p.lds); }
    if (p.gv.lds_compact_cap == 0) {
        p.P = count_lanes_per_ray(a, false);
        p.cap = 32;
        p.gv = make_view(a, p.cap * kBlock * 8, &p.lds);
        return p;
    }
    // sparse grid, but more rays than one round of LDS-image workgroups holds (> 8192): the image stays in L2 and the
    // workgroup's LDS is its lists only, 32 KB — five workgroups share a CU instead of one and the walk, bound by its
    // dependent instructions, overlaps five times as many of them (r03_count_pass.md: 74 -> 49 us at 20 k rays, 142 -> 87 at
    // 50 k; also staging the 4 KB bitmap of non-empty bricks is 2-3 % slower than leaving everything in L2).
    // NFA_SPLIT_L2 = 0 | 1 overrides
    bool l2 = a->n_rays > 8192;
    if (const char *e = getenv("NFA_SPLIT_L2")) l2 = atoi(e) != 0;
    if (p.P == 8) l2 = true;                       // (no 8-lane instance with the image in LDS any more: it never wins)
    if (l2) {
        p.l2 = 1;
        p.blk = kBlock;
        p.xt = 0;
        p.cap = 16;
        p.gv = make_view(a, p.cap * kBlock * 8, &p.lds, 0);
    }
    return p;
}

NFA_EXPORT int nfa_traverse_count(const nfa_traverse_args *a, void *workspace, void *stream)
{
    if (int rc = validate_traverse(a)) return rc;
    if (a->n_rays == 0) return NFA_OK;
    NFA_REQUIRE(workspace != nullptr, "traverse_count: workspace is NULL");
    hipStream_t s = (hipStream_t)stream;
    int64_t *block_sums = (int64_t *)workspace;
    const RunStore rs = make_runs(workspace, a->n_rays);
    const bool lattice = a->step_size > 0.0f && a->cone_angle == 0.0f;
    const int evm = a->t_sorted ? EV_PRE : (a->n_grids == 1 ? EV_ONE : EV_MANY);
    const SplitPlan plan = plan_split(a);
    const int P = plan.P;
    if (plan.seg == 2) {
        const int lds = plan.lds;
        const GridView &gv = plan.gv;
        const bool lds_occ = gv.lds_compact_cap > 0;
        const unsigned nbs = (unsigned)ceil_div(a->n_rays, kBlock / P);
        VoxelStore vs;
        vs.rec = (uint32_t *)((uint8_t *)workspace + ws_voxels_offset(a->n_rays));
        vs.cap = a->res[0] + a->res[1] + a->res[2];
        vs.xt = nullptr;
        if (P >= 32)
            vs.xt = (float *)((uint8_t *)vs.rec + ((nbs * (int64_t)(vs.cap + VoxelStore::kSlack) * kBlock * 4 + 255) & ~255ll));
#define NFA_LAUNCH_CONE(LDSO, PP)                                                                                               \
    do {                                                                                                                       \
        if (int rc = allow_lds(traverse_count_cone_kernel<LDSO, PP>, lds)) return rc;                                           \
        hipLaunchKernelGGL((traverse_count_cone_kernel<LDSO, PP>), dim3(nbs), dim3(kBlock), lds, s, *a, gv, block_sums, rs, vs); \
    } while (0)
        if (lds_occ) { if (P == 1) NFA_LAUNCH_CONE(true, 1); else if (P == 8) NFA_LAUNCH_CONE(true, 8); else if (P == 16) NFA_LAUNCH_CONE(true, 16); else if (P == 32) NFA_LAUNCH_CONE(true, 32); else NFA_LAUNCH_CONE(true, 64); }
        else { if (P == 1) NFA_LAUNCH_CONE(false, 1); else if (P == 8) NFA_LAUNCH_CONE(false, 8); else if (P == 16) NFA_LAUNCH_CONE(false, 16); else if (P == 32) NFA_LAUNCH_CONE(false, 32); else NFA_LAUNCH_CONE(false, 64); }
#undef NFA_LAUNCH_CONE
        return check_launch("traverse_count_cone_kernel");
    }
    if (P > 1) {
        const int lds = plan.lds;
        const GridView &gv = plan.gv;
        const bool lds_occ = gv.lds_compact_cap > 0;
        const unsigned nbs = (unsigned)ceil_div(a->n_rays, plan.blk / P);
        if (plan.seg) {
#define NFA_LAUNCH_SEG(LDSO, PP, CAP, KK)                                                                                       \
    do {                                                                                                                       \
        if (int rc = allow_lds(traverse_count_segments_kernel<LDSO, PP, CAP, KK>, lds)) return rc;                               \
        hipLaunchKernelGGL((traverse_count_segments_kernel<LDSO, PP, CAP, KK>), dim3(nbs), dim3(kBlock), lds, s, *a, gv, block_sums, rs, seg_xt); \
    } while (0)
            float *seg_xt = (float *)((uint8_t *)workspace + ws_voxels_offset(a->n_rays));     // crossing-time arrays (P = 32 only)
            if (lds_occ) { if (P == 8) NFA_LAUNCH_SEG(true, 8, 32, 1); else if (P == 16) NFA_LAUNCH_SEG(true, 16, 32, 1); else NFA_LAUNCH_SEG(true, 32, 32, 4); }
            else { if (P == 8) NFA_LAUNCH_SEG(false, 8, 32, 1); else if (P == 16) NFA_LAUNCH_SEG(false, 16, 32, 1); else NFA_LAUNCH_SEG(false, 32, 32, 4); }
#undef NFA_LAUNCH_SEG
            return check_launch("traverse_count_segments_kernel");
        }
#define NFA_LAUNCH_SPLIT(LDSO, PP, CAP)                                                                                        \
    do {                                                                                                                       \
        if (int rc = allow_lds(traverse_count_split_kernel<LDSO, PP, CAP>, lds)) return rc;                                     \
        hipLaunchKernelGGL((traverse_count_split_kernel<LDSO, PP, CAP>), dim3(nbs), dim3(kBlock), lds, s, *a, gv, block_sums, rs); \
    } while (0)
        if (plan.l2) {
            // grid image from L2: 16-entry lists, 32 KB of LDS per workgroup
            if (P == 8) NFA_LAUNCH_SPLIT(false, 8, 16); else NFA_LAUNCH_SPLIT(false, 16, 16);
        } else if (lds_occ && plan.blk == 512 && plan.xt) {
            if (int rc = allow_lds(traverse_count_split_kernel<true, 16, 16, 512, true>, lds)) return rc;
            hipLaunchKernelGGL((traverse_count_split_kernel<true, 16, 16, 512, true>), dim3(nbs), dim3(512), lds, s, *a, gv, block_sums, rs);
        } else if (lds_occ && plan.blk == 512) {
            if (int rc = allow_lds(traverse_count_split_kernel<true, 16, 16, 512>, lds)) return rc;
            hipLaunchKernelGGL((traverse_count_split_kernel<true, 16, 16, 512>), dim3(nbs), dim3(512), lds, s, *a, gv, block_sums, rs);
        } else if (lds_occ) {
            NFA_LAUNCH_SPLIT(true, 16, 16);
        } else {
            if (P == 2) NFA_LAUNCH_SPLIT(false, 2, 32); else if (P == 4) NFA_LAUNCH_SPLIT(false, 4, 32);
            else if (P == 8) NFA_LAUNCH_SPLIT(false, 8, 32); else NFA_LAUNCH_SPLIT(false, 16, 32);
        }
#undef NFA_LAUNCH_SPLIT
        return check_launch("traverse_count_split_kernel");
    }
    const unsigned nb = (unsigned)ceil_div(a->n_rays, kBlock);
    int lds = 0;
    GridView gv = make_view(a, kEvBytes, &lds);
    // the image in LDS only while one round of workgroups holds every ray: beyond, its copy per workgroup and the CU's
    // occupancy (one or two workgroups) cost more than L2 latency does (r03_count_pass.md: 937 -> 632 us at 1 M rays).
    // NFA_COUNT_L2 = 0 | 1 overrides
    {
        bool l2 = gv.lds_compact_cap > 0 && (int64_t)nb > (int64_t)kNumCU * (kLdsPerCU / (lds > 0 ? lds : 1));
        if (const char *e = getenv("NFA_COUNT_L2")) l2 = atoi(e) != 0;
        if (l2) gv = make_view(a, kEvBytes, &lds, 0);
    }
    const bool lds_occ = gv.lds_compact_cap > 0;
#define NFA_LAUNCH_COUNT(EVM, LAT, LDSO)                                                                                    \
    do {                                                                                                                    \
        if (int rc = allow_lds(traverse_count_kernel<EVM, LAT, LDSO>, lds)) return rc;                                       \
        hipLaunchKernelGGL((traverse_count_kernel<EVM, LAT, LDSO>), dim3(nb), dim3(kBlock), lds, s, *a, gv, block_sums, rs); \
    } while (0)
#define NFA_COUNT_EV(EVM)                                                              \
    do {                                                                               \
        if (lattice) { if (lds_occ) NFA_LAUNCH_COUNT(EVM, true, true); else NFA_LAUNCH_COUNT(EVM, true, false); }   \
        else         { if (lds_occ) NFA_LAUNCH_COUNT(EVM, false, true); else NFA_LAUNCH_COUNT(EVM, false, false); } \
    } while (0)
    if (evm == EV_PRE) NFA_COUNT_EV(EV_PRE);
    else if (evm == EV_ONE) NFA_COUNT_EV(EV_ONE);
    else NFA_COUNT_EV(EV_MANY);
#undef NFA_COUNT_EV
#undef NFA_LAUNCH_COUNT
    return check_launch("traverse_count_kernel");
}

NFA_EXPORT int nfa_traverse_offsets(const nfa_traverse_args *a, const void *workspace, void *stream)
{
    return nfa_traverse_offsets_stamped(a, workspace, 0, stream);
}

NFA_EXPORT int nfa_traverse_offsets_stamped(const nfa_traverse_args *a, const void *workspace, int64_t stamp, void *stream)
{
    if (int rc = validate_traverse(a)) return rc;
    NFA_REQUIRE(a->totals != nullptr, "traverse_offsets: totals is NULL");
    hipStream_t s = (hipStream_t)stream;
    if (a->n_rays == 0) { (void)hipMemsetAsync(a->totals, 0, 4 * sizeof(int64_t), s); return NFA_OK; }
    NFA_REQUIRE(workspace != nullptr, "traverse_offsets: workspace is NULL");
    const SplitPlan plan = plan_split(a);
    const int P = plan.P;
    const unsigned nb = (unsigned)ceil_div(a->n_rays, kBlock);
    const int64_t n_sums = ceil_div(a->n_rays, plan.blk / P) * (plan.blk / kWave);      // one triple per wave of the count launch
    hipLaunchKernelGGL(traverse_offsets_kernel, dim3(nb), dim3(kBlock), 0, s, a->iv_cnts, a->iv_starts, a->sm_cnts,
                       a->sm_starts, a->n_rays, (const int64_t *)workspace, P * kWavesPerBlock, n_sums, a->totals,
                       (int64_t *)((uint8_t *)const_cast<void *>(workspace) + ws_totals_offset(a->n_rays)), stamp);
    return check_launch("traverse_offsets_kernel");
}

static int launch_fill(const nfa_traverse_args *a, int skip_empty, int rewrite_counts, const uint16_t *only_overflow, hipStream_t s)
{
    int lds = 0;
    GridView gv = make_view(a, 0, &lds);       // no boundary lists in the general walk
    const unsigned nb = (unsigned)ceil_div(a->n_rays, kBlock);
    {   // (image in LDS only while one round of workgroups holds every ray, as in nfa_traverse_count)
        bool l2 = gv.lds_compact_cap > 0 && (int64_t)nb > (int64_t)kNumCU * (kLdsPerCU / (lds > 0 ? lds : 1));
        if (const char *e = getenv("NFA_COUNT_L2")) l2 = atoi(e) != 0;
        if (l2) gv = make_view(a, 0, &lds, 0);
    }
    const int evm = a->t_sorted ? EV_PRE : (a->n_grids == 1 ? EV_ONE : EV_MANY);
    const bool lds_occ = gv.lds_compact_cap > 0;
#define NFA_LAUNCH_FILL(EVM, LDSO)                                                                                           \
    do {                                                                                                                     \
        if (int rc = allow_lds(traverse_fill_kernel<EVM, LDSO>, lds)) return rc;                                              \
        hipLaunchKernelGGL((traverse_fill_kernel<EVM, LDSO>), dim3(nb), dim3(kBlock), lds, s, *a, gv, skip_empty, rewrite_counts, \
                           only_overflow);                                                                                   \
    } while (0)
    if (evm == EV_PRE) { if (lds_occ) NFA_LAUNCH_FILL(EV_PRE, true); else NFA_LAUNCH_FILL(EV_PRE, false); }
    else if (evm == EV_ONE) { if (lds_occ) NFA_LAUNCH_FILL(EV_ONE, true); else NFA_LAUNCH_FILL(EV_ONE, false); }
    else { if (lds_occ) NFA_LAUNCH_FILL(EV_MANY, true); else NFA_LAUNCH_FILL(EV_MANY, false); }
#undef NFA_LAUNCH_FILL
    return check_launch("traverse_fill_kernel");
}

// NFA_EMIT = rays | samples forces a form of the emit pass (traverse_emit_kernel chooses otherwise)
static int emit_hint() {
    if (const char *e = getenv("NFA_EMIT")) return e[0] == 'r' ? 1 : 2;
    return 0;
}
static unsigned emit_ray_blocks(int64_t n_rays) {
    const int64_t nb = ceil_div(n_rays, kBlock / 16), cap = (int64_t)kNumCU * 8;
    return (unsigned)(nb < cap ? nb : cap);
}

NFA_EXPORT int nfa_traverse_fill(const nfa_traverse_args *a, int32_t skip_empty, int32_t rewrite_counts,
                                 const void *workspace, int64_t n_samples, int64_t n_overflow, void *stream)
{
    if (int rc = validate_traverse(a)) return rc;
    if (a->n_rays == 0) return NFA_OK;
    if (a->iv_vals) NFA_REQUIRE(a->iv_ray_indices && a->iv_is_left && a->iv_is_right && a->iv_starts,
                                "traverse_fill: interval outputs must be given together");
    if (a->t_starts) NFA_REQUIRE(a->t_ends != nullptr, "traverse_fill: t_starts without t_ends");
    hipStream_t s = (hipStream_t)stream;
    if (!workspace) return launch_fill(a, skip_empty, rewrite_counts, nullptr, s);
    // runs recorded by nfa_traverse_count with the same args (two-pass mode only; skipped rays
    // of a rays_mask recorded no runs)
    NFA_REQUIRE(!rewrite_counts, "traverse_fill: replay is for the two-pass mode");
    NFA_REQUIRE(n_samples >= 0 && n_overflow >= 0, "traverse_fill: negative totals");
    const RunStore rs = make_runs(const_cast<void *>(workspace), a->n_rays);
    if (n_samples > 0) {
        const int64_t *n_dev = (const int64_t *)((const uint8_t *)workspace + ws_totals_offset(a->n_rays));
        const unsigned nb_s = blocks_for(n_samples), nb_r = emit_ray_blocks(a->n_rays);
        hipLaunchKernelGGL(traverse_emit_kernel, dim3(nb_s > nb_r ? nb_s : nb_r), dim3(kBlock), 0, s, *a, rs, n_samples, n_dev, 0, emit_hint());
        if (int rc = check_launch("traverse_emit_kernel")) return rc;
    }
    if (n_overflow > 0) return launch_fill(a, 1, 0, rs.n_runs, s);
    return NFA_OK;
}

NFA_EXPORT int nfa_traverse_emit_speculative(const nfa_traverse_args *a, const void *workspace, int64_t capacity, void *stream)
{
    if (int rc = validate_traverse(a)) return rc;
    if (a->n_rays == 0 || capacity <= 0) return NFA_OK;
    NFA_REQUIRE(workspace != nullptr, "traverse_emit_speculative: workspace is NULL");
    NFA_REQUIRE(!a->iv_vals, "traverse_emit_speculative: sampling outputs only");
    if (a->t_starts) NFA_REQUIRE(a->t_ends != nullptr, "traverse_emit_speculative: t_starts without t_ends");
    const RunStore rs = make_runs(const_cast<void *>(workspace), a->n_rays);
    const int64_t *n_dev = (const int64_t *)((const uint8_t *)workspace + ws_totals_offset(a->n_rays));
    const unsigned nb_s = blocks_for(capacity), nb_r = emit_ray_blocks(a->n_rays);
    hipLaunchKernelGGL(traverse_emit_kernel, dim3(nb_s > nb_r ? nb_s : nb_r), dim3(kBlock), 0, (hipStream_t)stream, *a, rs, capacity, n_dev, 1, emit_hint());
    return check_launch("traverse_emit_kernel");
}

NFA_EXPORT int nfa_exclusive_sum_i64(const int64_t *cnts, int64_t n, int64_t *starts, int64_t *total, void *stream)
{
    NFA_REQUIRE(n >= 0, "exclusive_sum_i64: n < 0");
    if (n == 0) {
        if (total) (void)hipMemsetAsync(total, 0, sizeof(int64_t), (hipStream_t)stream);
        return NFA_OK;
    }
    NFA_REQUIRE(cnts && starts, "exclusive_sum_i64: NULL pointer");
    hipStream_t s = (hipStream_t)stream;
    if (n <= kScanSingleMax) {
        hipLaunchKernelGGL(excl_sum_i64_kernel, dim3(1), dim3(1024), 0, s, cnts, n, starts, total);
        return check_launch("excl_sum_i64_kernel");
    }
    // three launches, in place, no scratch (the test-time marcher scans 640 000 counts four times per round: one workgroup
    // crawling through them took 905 us): chunks of >= 2048 counts scanned locally by a workgroup each, the chunk's total
    // parked in its FIRST slot (whose local value is always 0); the parked totals scanned in place by one workgroup — that IS
    // the final value of those slots; every chunk then adds its first slot to its other slots.  At most 4096 chunks.
    int64_t chunk = ceil_div(ceil_div(n, 4096), kBlock) * kBlock;
    if (chunk < 2048) chunk = 2048;
    const int64_t nb = ceil_div(n, chunk);
    hipLaunchKernelGGL(excl_sum_chunks_kernel, dim3((unsigned)nb), dim3(kBlock), 0, s, cnts, n, chunk, starts);
    hipLaunchKernelGGL(excl_sum_parked_kernel, dim3(1), dim3(1024), 0, s, starts, chunk, nb, total);
    hipLaunchKernelGGL(excl_sum_add_kernel, dim3((unsigned)nb), dim3(kBlock), 0, s, starts, n, chunk);
    return check_launch("excl_sum_chunks_kernel");
}

#ifdef NFA_PHASE_CYCLES
// instrumentation builds only: sum the per-wave slots into out16 (and optionally clear them)
extern "C" __attribute__((visibility("default"))) int nfa_debug_phase_cycles(unsigned long long *out16, int clear) {
    static unsigned long long host[nfa::kPhaseSlots][16];
    if (hipMemcpyFromSymbol(host, HIP_SYMBOL(nfa::g_phase_cycles), sizeof(host)) != hipSuccess) return 1;
    for (int i = 0; i < 16; ++i) out16[i] = 0;
    {
        unsigned long long mx = 0, hist[16];
        if (hipMemcpyFromSymbol(&mx, HIP_SYMBOL(nfa::g_phase_max_wave), sizeof(mx)) == hipSuccess &&
            hipMemcpyFromSymbol(hist, HIP_SYMBOL(nfa::g_phase_hist), sizeof(hist)) == hipSuccess) {
            fprintf(stderr, "[phase] slowest wave %llu cycles; waves per 16k-cycle bin:", mx);
            for (int i = 0; i < 16; ++i) fprintf(stderr, " %llu", hist[i]);
            fprintf(stderr, "\n");
            unsigned long long slow[16];
            if (hipMemcpyFromSymbol(slow, HIP_SYMBOL(nfa::g_phase_slow), sizeof(slow)) == hipSuccess && slow[15]) {
                fprintf(stderr, "[phase] %llu waves slower than 60 k cycles, average per phase:", slow[15]);
                for (int i = 0; i < 12; ++i) fprintf(stderr, " %llu", slow[i] / slow[15]);
                fprintf(stderr, " | serial rays %llu, streaming rays %llu in those waves", slow[12], slow[13]);
                fprintf(stderr, "\n");
            }
        }
        if (clear) {
            void *sym = nullptr;
            if (hipGetSymbolAddress(&sym, HIP_SYMBOL(nfa::g_phase_max_wave)) == hipSuccess) (void)hipMemset(sym, 0, sizeof(mx));
            if (hipGetSymbolAddress(&sym, HIP_SYMBOL(nfa::g_phase_hist)) == hipSuccess) (void)hipMemset(sym, 0, sizeof(hist));
            if (hipGetSymbolAddress(&sym, HIP_SYMBOL(nfa::g_phase_slow)) == hipSuccess) (void)hipMemset(sym, 0, 16 * sizeof(unsigned long long));
        }
    }
    unsigned long long t_min = ~0ull, t_max = 0;
    for (int w = 0; w < nfa::kPhaseSlots; ++w) {
        if (!host[w][15]) continue;
        for (int i = 0; i < 14; ++i) out16[i] += host[w][i];
        out16[15] += host[w][15];
    }
    if (clear) {
        if (hipMemset((void *)nullptr, 0, 0) != hipSuccess) {}
        void *sym = nullptr;
        if (hipGetSymbolAddress(&sym, HIP_SYMBOL(nfa::g_phase_cycles)) != hipSuccess) return 1;
        if (hipMemset(sym, 0, sizeof(host)) != hipSuccess) return 1;
    }
    (void)t_min; (void)t_max;
    return 0;
}
#endif
