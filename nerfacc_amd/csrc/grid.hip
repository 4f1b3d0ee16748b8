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
#include "dda_skip.hpp"       // struct Dda, dda_advance and the macro step dda_skip (host-testable)
#include "lookback.hpp"       // hand-offs between the workgroups of one launch (the fused sampling kernel)

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
//   the word), each padded to 8 bytes, then compact[n_bricks] (the non-empty bricks in order), then dist: one NIBBLE per brick
//   (brick b in byte b / 2, low nibble first) = Chebyshev distance in bricks, within its level, to the nearest non-empty brick,
//   capped at kDistCap (0 = the brick itself holds an occupied voxel).
// ----------------------------------------------------------------------------------------
struct PackedLayout {
    int64_t n_bricks, n_words, off_header, off_coarse, off_prefix, off_compact, off_dist, total_words;
};
__host__ __device__ inline PackedLayout packed_layout(int n_grids, int rx, int ry, int rz) {
    PackedLayout L;
    L.n_bricks = (int64_t)n_grids * ((rx + 3) / 4) * ((ry + 3) / 4) * ((rz + 3) / 4);
    L.n_words = (L.n_bricks + 31) / 32;
    L.off_header = L.n_bricks;
    L.off_coarse = L.off_header + 12;          // header: [0] non-empty bricks, [1 + g] occupied voxels of level g (g < 8), 3 spare
    L.off_prefix = L.off_coarse + (L.n_words + 1) / 2;
    L.off_compact = L.off_prefix + (L.n_words + 1) / 2;
    L.off_dist = L.off_compact + L.n_bricks;     // round 5: 4 bits per brick, distance to the nearest non-empty brick (brick_dist_kernel)
    L.total_words = L.off_dist + 2 * ((L.n_bricks + 31) / 32);      // (whole 16-byte units)
    return L;
}

// Sixteen lanes per brick — one per (dx, dy) row of its 4 x 4 x 4 voxels, ONE 16-byte (floats) or 4-byte (bools) load each — and a
// 512-thread workgroup per coarse word (32 bricks).  The sixteen nibbles of a brick meet by OR over DPP row rotations (a brick's
// lanes are one DPP row); the workgroup's eight waves put their four "non-empty" bits together in LDS.  (Rounds 1-5: one lane per
// brick with 64 scalar loads and 64 byte stores — 128 workgroups of dependent loads: 24 us per 128^3 grid update in the bench's
// trace, 8 MB in 24 us; round 6: ~2 k workgroups of one load per lane.)
// FROM_OCCS: the voxels come from the float occupancies (`occs > min(mean, occ_thre)`, occ_grid.py:392-404) and the
// bool grid is an OUTPUT — threshold and bit-pack in one pass (nfa_grid_threshold_packed).
constexpr int kPackBlock = 512;
constexpr int kPackWords = 4;      // coarse words (32 bricks each) per workgroup
template <bool FROM_OCCS>
__global__ __launch_bounds__(kPackBlock) void pack_bricks_kernel(
    const uint8_t *__restrict__ binaries, int n_grids, int rx, int ry, int rz,
    int nbx, int nby, int nbz, uint64_t *__restrict__ bricks, uint32_t *__restrict__ coarse, int64_t *__restrict__ level_counts,
    const float *__restrict__ occs, const double *__restrict__ partials, int n_partials, float occ_thre,
    uint8_t *__restrict__ binaries_out, float *__restrict__ thre_out)
{
    __shared__ uint32_t s_any[kPackWords][kPackBlock / 64];
    __shared__ int s_cnt[kPackWords][kPackBlock / 64];
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
    const int64_t n_words = (total + 31) / 32;
    const int lane = lane_id(), wv = (int)(threadIdx.x >> 6);
    const int lb = (int)(threadIdx.x >> 4), dx = (int)((threadIdx.x >> 2) & 3), dy = (int)(threadIdx.x & 3);
    // kPackWords consecutive coarse words per workgroup (round 6).  (1) ONE atomic per workgroup and level (thread 0 adds up): the
    // atomics on a level's counter serialise in the L2 — one per word, 1 024 of them for a 128^3 grid and 4 096 for four levels, were
    // a third of this kernel's 19 / 44 us.  (2) The words' loads are all requested before the first is looked at, and the workgroup
    // meets ONCE: a word per trip was a memory round trip and two barriers per word.
    const int64_t w_begin = (int64_t)blockIdx.x * kPackWords;
    uint32_t nibs[kPackWords];
    int64_t gs[kPackWords];
#pragma unroll
    for (int i = 0; i < kPackWords; ++i) {
        const int64_t b = (w_begin + i) * 32 + lb;
        uint32_t nib = 0;
        int64_t g = 0;
        if (b < total) {
            g = b / per_grid;
            int64_t rem = b - g * per_grid;
            const int bx = (int)(rem / ((int64_t)nby * nbz));
            rem -= (int64_t)bx * nby * nbz;
            const int by = (int)(rem / nbz), bz = (int)(rem - (int64_t)by * nbz);
            const int x = bx * 4 + dx, y = by * 4 + dy;
            if (x < rx && y < ry) {
                const int64_t row = g * (int64_t)rx * ry * rz + ((int64_t)x * ry + y) * rz + bz * 4;
                if (bz * 4 + 4 <= rz && (row & 3) == 0) {               // a whole, aligned row of four voxels: one load, one store
                    if (FROM_OCCS) {
                        const float4 v = *reinterpret_cast<const float4 *>(occs + row);
                        nib = (v.x > thre ? 1u : 0u) | (v.y > thre ? 2u : 0u) | (v.z > thre ? 4u : 0u) | (v.w > thre ? 8u : 0u);
                        *reinterpret_cast<uint32_t *>(binaries_out + row) = (nib & 1u) | ((nib & 2u) << 7) | ((nib & 4u) << 14) | ((nib & 8u) << 21);
                    } else {
                        const uint32_t v = *reinterpret_cast<const uint32_t *>(binaries + row);
                        nib = ((v & 0xffu) ? 1u : 0u) | ((v & 0xff00u) ? 2u : 0u) | ((v & 0xff0000u) ? 4u : 0u) | ((v & 0xff000000u) ? 8u : 0u);
                    }
                } else {
#pragma unroll
                    for (int dz = 0; dz < 4; ++dz) {
                        if (bz * 4 + dz >= rz) continue;
                        bool on;
                        if (FROM_OCCS) { on = occs[row + dz] > thre; binaries_out[row + dz] = on ? 1 : 0; }
                        else on = binaries[row + dz] != 0;
                        if (on) nib |= 1u << dz;
                    }
                }
            }
        }
        nibs[i] = nib;
        gs[i] = g;
    }
    const bool leader = (threadIdx.x & 15) == 0;
    const int sh = dx * 16 + dy * 4;
#pragma unroll
    for (int i = 0; i < kPackWords; ++i) {
        const int64_t w = w_begin + i, b = w * 32 + lb;
        // the brick's word: bit dx * 16 + dy * 4 + dz — OR over the brick's sixteen lanes (row rotations by 1, 2, 4, 8)
        uint32_t lo = sh < 32 ? nibs[i] << sh : 0u, hi = sh >= 32 ? nibs[i] << (sh - 32) : 0u;
#define NFA_ROW_OR(N)                                                                                          \
        lo |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)lo, 0x120 + N, 0xf, 0xf, false);                      \
        hi |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)hi, 0x120 + N, 0xf, 0xf, false);
        NFA_ROW_OR(1) NFA_ROW_OR(2) NFA_ROW_OR(4) NFA_ROW_OR(8)
#undef NFA_ROW_OR
        const uint64_t bits = ((uint64_t)hi << 32) | lo;
        if (leader && b < total) bricks[b] = bits;
        // the wave's four bricks: non-empty flags (bits 4 wv ... 4 wv + 3 of the coarse word) and occupied voxels
        const unsigned long long any = __ballot(leader && bits != 0ull);
        const uint32_t four = (uint32_t)((any & 1ull) | ((any >> 15) & 2ull) | ((any >> 30) & 4ull) | ((any >> 45) & 8ull));
        const int cnt = (int)wave_sum_i64(leader ? (int64_t)__popcll(bits) : (int64_t)0);
        if (lane == 0) { s_any[i][wv] = four; s_cnt[i][wv] = cnt; }
        // occupied voxels per level (integer atomics: deterministic): lets OccGridEstimator size its list of occupied cells without
        // a sync.  A word whose 32 bricks straddle two levels adds brick by brick.
        if (w < n_words) {
            const int64_t g_first = (w * 32) / per_grid, g_last = ((w * 32 + 31 < total ? w * 32 + 31 : total - 1)) / per_grid;
            if (g_first != g_last && leader && bits && b < total) atomicAdd((unsigned long long *)(level_counts + gs[i]), (unsigned long long)__popcll(bits));
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int64_t acc_cnt = 0, acc_g = -1;
        for (int i = 0; i < kPackWords; ++i) {
            const int64_t w = w_begin + i;
            if (w >= n_words) break;
            uint32_t word = 0;
            int tot = 0;
#pragma unroll
            for (int k = 0; k < kPackBlock / 64; ++k) { word |= s_any[i][k] << (4 * k); tot += s_cnt[i][k]; }
            coarse[w] = word;
            const int64_t g_first = (w * 32) / per_grid, g_last = ((w * 32 + 31 < total ? w * 32 + 31 : total - 1)) / per_grid;
            if (g_first == g_last && tot) {
                if (acc_g != g_first && acc_cnt) atomicAdd((unsigned long long *)(level_counts + acc_g), (unsigned long long)acc_cnt);
                if (acc_g != g_first) { acc_g = g_first; acc_cnt = 0; }
                acc_cnt += tot;
            }
        }
        if (acc_cnt) atomicAdd((unsigned long long *)(level_counts + acc_g), (unsigned long long)acc_cnt);
    }
}

// a workgroup per kPackWords coarse words
inline unsigned pack_blocks(int64_t n_bricks) {
    const int64_t w = (n_bricks + 31) / 32;
    return (unsigned)(w < 1 ? 1 : ceil_div(w, (int64_t)kPackWords));
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

// Empty-space skipping (round 5): per brick the Chebyshev distance D, in bricks and within its level, to the nearest non-empty
// brick, capped at kDistCap.  A walk standing in a brick with D >= 1 knows that the cube of (2 D - 1)^3 bricks around it is empty
// and leaves it in ONE macro step (dda_skip.hpp) — with D <= 4 that is at most 15 further plane crossings per axis, the range of
// the macro step's integer bisection.  One lane per PAIR of bricks (one byte of output), brute force over the 7^3 neighbourhood of
// the bitmap of non-empty bricks: 32 k bricks at 128^3, a few microseconds once per grid update.  Bricks beyond the grid count as
// empty (a walk never goes there: its overflow index stops it first).
constexpr int kDistCap = 4;
// One lane per brick (the two nibbles of an output byte meet through a lane shuffle); per (dx, dy) the 7 bits of the neighbouring
// z-row around bz come out of the bitmap with two word loads and a funnel shift, and the nearest set bit of the window is four mask
// tests — 49 windows instead of 343 single-bit probes (round 5, second form: 47 -> ~10 us at 128^3 in the bench's kernel trace).
// LDS_MAP (round 6): the whole bitmap (4 KB at 128^3, 32 KB at 256^3) is staged in LDS first — the ~100 window reads of a lane are a
// dependent chain (`best` prunes the loops), and from L2 that chain was the kernel's 11.5 us at 128^3 in the bench's trace.
template <bool LDS_MAP>
__global__ __launch_bounds__(kBlock) void brick_dist_kernel(const uint32_t *__restrict__ coarse_g, int n_grids, int nbx, int nby, int nbz,
                                                            uint8_t *__restrict__ dist)
{
    extern __shared__ uint32_t s_coarse[];
    const int64_t per_grid = (int64_t)nbx * nby * nbz, total = per_grid * n_grids;
    const int64_t rounded = (total + 63) / 64 * 64, n_words = (total + 31) / 32;
    if (LDS_MAP) {
        for (int64_t i = threadIdx.x; i < n_words; i += kBlock) s_coarse[i] = coarse_g[i];
        __syncthreads();
    }
    const uint32_t *const coarse = LDS_MAP ? (const uint32_t *)s_coarse : coarse_g;
    for (int64_t b = (int64_t)blockIdx.x * kBlock + threadIdx.x; b < rounded; b += (int64_t)gridDim.x * kBlock) {
        int best = kDistCap;
        if (b < total) {
            const int64_t g = b / per_grid;
            int64_t rem = b - g * per_grid;
            const int bx = (int)(rem / ((int64_t)nby * nbz));
            rem -= (int64_t)bx * nby * nbz;
            const int by = (int)(rem / nbz), bz = (int)(rem - (int64_t)by * nbz);
            // window bits k = 0 .. 6 <-> z = bz - 3 + k; the bits whose z lies outside the row are masked off
            const int lo_z = bz - 3, k_lo = lo_z < 0 ? -lo_z : 0, k_hi = (bz + 3 >= nbz) ? (nbz - 1 - lo_z) : 6;      // valid k range
            const unsigned valid = ((2u << k_hi) - 1u) & ~((1u << k_lo) - 1u);
            for (int dx = -(kDistCap - 1); dx <= kDistCap - 1; ++dx) {
                const int x = bx + dx, adx = dx < 0 ? -dx : dx;
                if (x < 0 || x >= nbx || adx >= best) continue;
                for (int dy = -(kDistCap - 1); dy <= kDistCap - 1; ++dy) {
                    const int y = by + dy, ady = dy < 0 ? -dy : dy;
                    const int axy = adx > ady ? adx : ady;
                    if (y < 0 || y >= nby || axy >= best) continue;
                    // bit position of (x, y, bz - 3) in the bitmap; may be "negative" at the very start of the array: shift instead
                    const int64_t pos = g * per_grid + ((int64_t)x * nby + y) * nbz + lo_z;
                    const int64_t p0 = pos < 0 ? 0 : pos;
                    const int64_t wi = p0 >> 5;
                    const uint64_t two = (uint64_t)coarse[wi] | ((wi + 1 < n_words) ? ((uint64_t)coarse[wi + 1] << 32) : 0ull);
                    unsigned w = (unsigned)(two >> (p0 & 31));
                    if (pos < 0) w <<= (unsigned)(-pos);              // (only the first row of the first level, bz < 3)
                    w &= valid & 0x7fu;
                    if (!w) continue;
                    const int dz = (w & 0x08u) ? 0 : (w & 0x1cu) ? 1 : (w & 0x3eu) ? 2 : 3;
                    const int r = axy > dz ? axy : dz;
                    best = r < best ? r : best;
                }
            }
        }
        const int other = __shfl_xor(best, 1, 64);                   // bricks 2 i and 2 i + 1 sit in adjacent lanes (b is even on even lanes)
        if (!(threadIdx.x & 1) && b < total) dist[b >> 1] = (uint8_t)(best | ((b + 1 < total ? other : kDistCap) << 4));
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
    const uint8_t *__restrict__ dist;   // one nibble per brick: distance to the nearest non-empty brick (brick_dist_kernel)
    int skip_auto;        // 1: a wave takes the empty-space macro steps only when its 64 rays are coherent (wave_rays_coherent)
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

// The same staging in two halves (round 5, the 512-thread crossing-time form of the split count pass): `stage_issue` requests the
// image's words into registers, `stage_commit` writes them to LDS and joins the workgroup.  What a kernel computes between the two
// — the ray's slab test, DDA setup and the closed-form jumps, none of which touches the image — runs under the L2 round trips of the
// copy instead of behind them (3.6 k of the mean wave's 45 k cycles, profiles/r04_count_pass.md section 4).  At most kStageWords
// bitmap words and kStageBricks compact bricks per thread are held; a larger image is copied by the plain loop in stage_commit.
constexpr int kStageWords = 3, kStageBricks = 8;
struct StagePending {
    uint2 w[kStageWords];
    uint64_t b[kStageBricks];
    bool in_regs;
};
__device__ __forceinline__ Occ<true> stage_layout(const GridView &g, char *smem) {
    Occ<true> l;
    l.smem = smem;
    l.w4 = (g.lds_words + 3) & ~3;
    l.cap = g.lds_compact_cap;
    l.bytes = (2 * l.w4 * 4 + g.lds_compact_cap * 8 + 15) & ~15;
    return l;
}
template <int BLK_UNUSED>
__device__ __forceinline__ void stage_issue(const GridView &g, StagePending &sp) {
    const int BLK = (int)blockDim.x;                   // (the launch may hold fewer threads than the layout's stride: split_launch_threads)
    sp.in_regs = g.lds_words <= kStageWords * BLK && g.lds_compact_cap <= kStageBricks * BLK;      // (workgroup-uniform)
    // EVERY thread issues the same loads on one straight path (indices clamped, not predicated): with a branch around them the
    // compiler's wait-count pass has to assume the path without loads at the join and waits for vmcnt(0) at the first use of the
    // ray's values — the copy then completes before the arithmetic it was meant to run under.
    const int tid = threadIdx.x;
    const int w_last = g.lds_words > 0 ? g.lds_words - 1 : 0, b_last = g.lds_compact_cap > 0 ? g.lds_compact_cap - 1 : 0;
#pragma unroll
    for (int k = 0; k < kStageWords; ++k) {
        const int i = min(tid + k * BLK, w_last);
        sp.w[k] = make_uint2(g.coarse[i], g.prefix[i]);
    }
#pragma unroll
    for (int k = 0; k < kStageBricks; ++k) {
        const int i = min(tid + k * BLK, b_last);
        sp.b[k] = g.compact[i];
    }
}
template <int BLK_UNUSED>
__device__ __forceinline__ void stage_commit(const GridView &g, const Occ<true> &l, const StagePending &sp) {
    const int BLK = (int)blockDim.x;
    uint2 *lw = (uint2 *)l.smem;
    uint64_t *lb = (uint64_t *)((uint32_t *)l.smem + 2 * l.w4);
    const int tid = threadIdx.x;
    if (sp.in_regs) {
#pragma unroll
        for (int k = 0; k < kStageWords; ++k) { const int i = tid + k * BLK; if (i < g.lds_words) lw[i] = sp.w[k]; }
#pragma unroll
        for (int k = 0; k < kStageBricks; ++k) { const int i = tid + k * BLK; if (i < g.lds_compact_cap) lb[i] = sp.b[k]; }
    } else {
        for (int i = tid; i < g.lds_words; i += BLK) lw[i] = make_uint2(g.coarse[i], g.prefix[i]);
        for (int i = tid; i < g.lds_compact_cap; i += BLK) lb[i] = g.compact[i];
    }
    __syncthreads();
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
            // (read as ONE 64-bit word: as a uint2 the compiler fetched .y in a second, dependent ds_read inside the branch — three LDS
            //  round trips per non-empty brick instead of two)
            const uint64_t wr64 = ((const uint64_t *)l.smem)[id >> 5];
            const uint32_t w = (uint32_t)wr64, wr_y = (uint32_t)(wr64 >> 32);
            if (w & bit) {
                const int k = (int)wr_y + __popc(w & (bit - 1u));
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

// ---- empty-space skipping (round 5) -------------------------------------------------------------------------------------
// the 64 voxels of brick `id` (same sources as `occupied`)
template <bool LDS_OCC>
__device__ __forceinline__ uint64_t brick_bits(const GridView &g, const Occ<LDS_OCC> &l, int id) {
    if (LDS_OCC) {
        const uint64_t wr64 = ((const uint64_t *)l.smem)[id >> 5];          // {bitmap word, rank prefix} in one ds_read_b64
        const uint32_t wx = (uint32_t)wr64, wy = (uint32_t)(wr64 >> 32);
        const uint32_t bit = 1u << (id & 31);
        const int k = (wx & bit) ? (int)wy + __popc(wx & (bit - 1u)) : 0;
        const uint64_t b = ((const uint64_t *)((const uint32_t *)l.smem + 2 * l.w4))[k];
        return (wx & bit) ? b : 0ull;
    }
    return g.bricks[id];
}
// distance (in bricks, capped at kDistCap) from brick `id` to the nearest non-empty brick of its level; 0 = non-empty
__device__ __forceinline__ int brick_dist(const GridView &g, int id) {
    const unsigned v = (unsigned)g.dist[id >> 1];
    return (int)((v >> ((id & 1) << 2)) & 15u);
}
// plane crossings a macro step may take per axis BEYOND the pending one (dda_skip's k?1) from voxel c of a brick at distance D >= 1:
// to the face of the cube of (2 D - 1)^3 empty bricks around the brick, never past the axis' overflow index
__device__ __forceinline__ int skip_reach(int c, int step, int overflow, int D) {
    const int to_face = 4 * (D - 1) + ((c & 3) ^ (step > 0 ? 3 : 0));
    const int left = (overflow - c) * step;                 // crossings until the walk ends on this axis (<= 0: never, or step 0)
    return left >= 1 ? min(to_face, left - 1) : 0;
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
// Round 5: every optional array is read through a SELECTED POINTER (the array's element, or the ray's origin when the array is absent —
// always a valid address) on one straight path, and the value is selected afterwards.  With `if (a.t_min) { ... a.t_min[r] ... }` the
// compiler put each load in its own block with its own s_waitcnt vmcnt(0): three to four SERIALISED memory round trips at the head of
// every traversal kernel (t_min, jitter, t_max of OccGridEstimator.sampling) before the slab test could start.
__device__ __forceinline__ float ray_near(const nfa_traverse_args &a, int64_t r) {
    const float *fallback = a.rays_o + 3 * r;
    const float ld_n = *(a.near_planes ? a.near_planes + r : fallback);
    const float ld_m = *(a.t_min ? a.t_min + r : fallback);
    const float ld_j = *(a.jitter ? a.jitter + r : fallback);
    float v = a.near_planes ? ld_n : a.near_plane;
    if (a.t_min) { const float m = ld_m; v = (m != m) ? m : (v < m ? m : v); }                 // torch.clamp(min=): NaN bound propagates
    if (a.jitter) v = v + ld_j * a.jitter_scale;                                               // -ffp-contract=off: mul, then add
    return v;
}
__device__ __forceinline__ float ray_far(const nfa_traverse_args &a, int64_t r) {
    const float *fallback = a.rays_o + 3 * r;
    const float ld_f = *(a.far_planes ? a.far_planes + r : fallback);
    const float ld_m = *(a.t_max ? a.t_max + r : fallback);
    float v = a.far_planes ? ld_f : a.far_plane;
    if (a.t_max) { const float m = ld_m; v = (m != m) ? m : (v > m ? m : v); }
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

// The same lattice walk with EMPTY-SPACE SKIPPING (round 5).  Phase A alternates between two tight loops — the
// persistent "while-while" shape of GPU ray traversal:
//   far:   the voxel's brick is empty and D bricks from the nearest non-empty one (brick_dist_kernel): ONE macro step (dda_skip)
//          leaves the cube of (2 D - 1)^3 empty bricks around it — up to 15 plane crossings per axis, the state that the voxel walk
//          would reach, bit for bit — and the brick under the landing voxel is looked up;
//   near:  the brick holds an occupied voxel: the voxel walk of rounds 1-4 (boundary list, one DDA step), brick bits in a register.
// A lane leaves a loop when its walk changes regime; the wave re-converges between the loops, so a lane in open space never pays
// for its neighbour's voxel steps more than once per regime change (a flat "macro step or voxel step" loop body would cost the sum
// of both at every iteration: P(the 64 lanes of a wave agree) is ~0).  On the bench scene a ray visits 188 voxels, 16 of them in
// non-empty bricks, and takes ~11 macro steps (tools/experiments/r05_skip_stats.py).  Phase B is unchanged.
template <int EV, bool LDS_OCC>
__device__ __forceinline__ void traverse_ray_lattice_skip(const nfa_traverse_args &a, const GridView &gv, const Occ<LDS_OCC> &occ,
                                                          const char *smem, float *__restrict__ ev_lds /* [kEvCap][blockDim] */,
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
        s.tx = s.ty = s.tz = 0.f; s.dx = s.dy = s.dz = 0.f;
        s.sx = s.sy = s.sz = 0; s.cx = s.cy = s.cz = 0; s.ox = s.oy = s.oz = 0;
        if (seg_live) dda_setup(s, o, d, inv, seg_lo, seg_hi, a.aabbs + 6 * level, gv.res);
        bool have_run = false, run_occ = false;
        float run_exit = 0.f;
        const int id_base = level * gv.bricks_per_grid;
        // macro steps run in the integer domain (dda_skip.hpp: IDda), which needs positive, ordered crossing times; a ray without
        // them simply never leaves the voxel loop
        const bool sane = seg_live && idda_sane(s);
        // brick under the current voxel: its distance nibble and its 64 voxels (both requested together: one round trip)
        int cur_id = -1, cur_dist = 0;
        uint64_t cur_bits = 0;
        while (__any(seg_live)) {
            // ---- A: boundaries only
            int n_ev = 0;
            unsigned ev_occ = 0;
            bool go = seg_live;
            while (go) {
                {
                    const int id = (int)__umul24(__umul24(s.cx >> 2, gv.nby) + (s.cy >> 2), gv.nbz) + (s.cz >> 2) + id_base;
                    if (id != cur_id) {
                        cur_id = id;
                        cur_dist = brick_dist(gv, id);
                        cur_bits = brick_bits<LDS_OCC>(gv, occ, id);
                    }
                }
                if (sane && cur_dist > 0) {
                    // far: the current voxel is empty — the occupied run before it, if any, ends here
                    if (have_run && run_occ) {
                        ev_lds[n_ev * nthr + tid] = run_exit;
                        ev_occ |= 1u << n_ev;
                        ++n_ev;
                    }
                    have_run = true;
                    run_occ = false;
                    IDda w = idda_init(s);
                    bool far_ = true;
                    while (far_) {
                        const DdaSkip k = idda_skip(w, skip_reach(w.cx, w.sx, w.ox, cur_dist), skip_reach(w.cy, w.sy, w.oy, cur_dist),
                                                    skip_reach(w.cz, w.sz, w.oz, cur_dist));
                        run_exit = fminf(k.t_exit, seg_hi);
                        if (!k.cont) {
                            seg_live = false;
                            far_ = false;
                        } else {
                            const int id = (int)__umul24(__umul24(w.cx >> 2, gv.nby) + (w.cy >> 2), gv.nbz) + (w.cz >> 2) + id_base;
                            if (id != cur_id) { cur_id = id; cur_dist = brick_dist(gv, id); }
                            far_ = cur_dist > 0;
                        }
                    }
                    s.tx = nfa_u2f(w.x.tb); s.ty = nfa_u2f(w.y.tb); s.tz = nfa_u2f(w.z.tb);
                    s.cx = w.cx; s.cy = w.cy; s.cz = w.cz;
                    if (seg_live) cur_bits = brick_bits<LDS_OCC>(gv, occ, cur_id);
                } else {
                    // near: voxel by voxel inside non-empty bricks
                    bool near_ = true;
                    while (near_) {
                        const float t_cell = fminf(fminf(s.tx, fminf(s.ty, s.tz)), seg_hi);
                        const bool oc = (cur_bits >> (((s.cx & 3) << 4) | ((s.cy & 3) << 2) | (s.cz & 3))) & 1ull;
                        if (have_run && oc != run_occ) {
                            ev_lds[n_ev * nthr + tid] = run_exit;
                            ev_occ |= (run_occ ? 1u : 0u) << n_ev;
                            ++n_ev;
                        }
                        have_run = true;
                        run_occ = oc;
                        run_exit = t_cell;
                        if (!dda_advance(s)) {
                            seg_live = false;
                            near_ = false;
                        } else {
                            const int id = (int)__umul24(__umul24(s.cx >> 2, gv.nby) + (s.cy >> 2), gv.nbz) + (s.cz >> 2) + id_base;
                            if (id != cur_id) {
                                cur_id = id;
                                cur_dist = brick_dist(gv, id);
                                cur_bits = brick_bits<LDS_OCC>(gv, occ, id);
                            }
                            near_ = !(sane && cur_dist > 0) && n_ev < kEvCap - 1;
                        }
                    }
                }
                go = seg_live && n_ev < kEvCap - 1;
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
// Are the rays of this wave (one per lane) a coherent bundle — neighbouring pixels of one camera?  Wave-uniform answer: every active
// lane's direction within ~8 degrees of the first active lane's (normalised dot product > 0.99) and its origin within 5 % of the first
// level's box of that lane's.  Random training rays practically never pass; rows of an image do.
__device__ __forceinline__ bool wave_rays_coherent(const nfa_traverse_args &a, int64_t r, bool active) {
    const unsigned long long am = __ballot(active);
    if (am == 0ull) return false;
    const int l0 = __ffsll((long long)am) - 1;
    float o[3] = {0.f, 0.f, 0.f}, d[3] = {0.f, 0.f, 1.f};
    if (active) {
        o[0] = a.rays_o[3 * r]; o[1] = a.rays_o[3 * r + 1]; o[2] = a.rays_o[3 * r + 2];
        d[0] = a.rays_d[3 * r]; d[1] = a.rays_d[3 * r + 1]; d[2] = a.rays_d[3 * r + 2];
    }
    const float o0[3] = {__shfl(o[0], l0, 64), __shfl(o[1], l0, 64), __shfl(o[2], l0, 64)};
    const float d0[3] = {__shfl(d[0], l0, 64), __shfl(d[1], l0, 64), __shfl(d[2], l0, 64)};
    const float dd = d[0] * d[0] + d[1] * d[1] + d[2] * d[2], d00 = d0[0] * d0[0] + d0[1] * d0[1] + d0[2] * d0[2];
    const float dot = d[0] * d0[0] + d[1] * d0[1] + d[2] * d0[2];
    const float ext = fmaxf(a.aabbs[3] - a.aabbs[0], fmaxf(a.aabbs[4] - a.aabbs[1], a.aabbs[5] - a.aabbs[2]));
    const float od = fmaxf(fabsf(o[0] - o0[0]), fmaxf(fabsf(o[1] - o0[1]), fabsf(o[2] - o0[2])));
    const bool ok = dot > 0.0f && dot * dot > 0.9801f * dd * d00 && od <= 0.05f * ext;
    return __ballot(active && !ok) == 0ull;
}

// SKIP (lattice form only): 0 = voxel by voxel, 1 = empty-space macro steps (brick distances read from L2)
template <int EV, bool LATTICE, bool LDS_OCC, int SKIP = 0>
__global__ __launch_bounds__(kBlock) void traverse_count_kernel(nfa_traverse_args a, GridView gv,
                                                                int64_t *__restrict__ block_sums, RunStore rs)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const Occ<LDS_OCC> occ = stage_occupancy<LDS_OCC>(gv, smem);
    const int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const bool active = r < a.n_rays && !(a.rays_mask && !a.rays_mask[r]);
    CountSink sink{rs, r, a.n_rays};
    float t_term = 0.f;
    if (LATTICE && SKIP) {
        float *ev_lds = (float *)(smem + occ.bytes);
        // The macro steps pay off when the lanes of a wave change regime TOGETHER (pixel-ordered rays of a frame: 276 -> 189 us at
        // 10^6 rays, 268 -> 167 for a round of the test-time marcher) and cost 17-30 % when they do not (a training batch: every
        // regime change is paid at the slowest lane's length in both loops; profiles/r05_count_pass.md).  Which of the two a wave is
        // can be read off its rays: same origin region, directions within a few degrees of lane 0's.
        const bool use_skip = !gv.skip_auto || wave_rays_coherent(a, r, active);
        if (use_skip) traverse_ray_lattice_skip<EV, LDS_OCC>(a, gv, occ, smem, ev_lds, r, active, sink, t_term);
        else traverse_ray_lattice<EV, LDS_OCC>(a, gv, occ, ev_lds, r, active, sink, t_term);
    } else if (LATTICE) {
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

#include "emit_pass.hpp"
#include "sample_fused.hpp"
#include "split_walk.hpp"
#include "segments_walk.hpp"
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
    // (four strides per trip, every load of a trip requested before the first is added: the last block walks ~1 600 triples at the bench
    //  size — seven DEPENDENT-looking trips of one load pair each were most of this kernel's 6.6 us, and the host waits for its stamp)
    for (int64_t j = threadIdx.x; j < before; j += 4 * kBlock) {
        int64_t a0[4], a1[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t k = j + u * kBlock;
            const bool in_k = k < before;
            a0[u] = in_k ? block_sums[3 * k] : 0;
            a1[u] = in_k ? block_sums[3 * k + 1] : 0;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) { p0 += a0[u]; p1 += a1[u]; }
    }
    int64_t ed = 0;
    if (last)                                     // rays that need the pass-2 re-traversal; the call's edges (also without iv_cnts: the emit
        for (int64_t j = threadIdx.x; j < n_sums; j += 4 * kBlock) {                                                         // pass reads runs = edges - samples)
            int64_t b0[4], b2[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int64_t k = j + u * kBlock;
                const bool in_k = k < n_sums;
                b2[u] = in_k ? block_sums[3 * k + 2] : 0;
                b0[u] = in_k ? block_sums[3 * k] : 0;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) { ov += b2[u]; ed += b0[u]; }
        }
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
    gv.dist = (const uint8_t *)(a->bricks + L.off_dist);
    gv.skip_auto = 0;
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
    { const int v = (int)opt(OPT_CONE_P, 0); if ((v == 8 && 2 * n_grids - 1 <= 8) || v == 16 || v == 32 || v == 64) P = v; }
    return P;
}
inline int64_t cone_voxel_bytes(const nfa_traverse_args *a) {
    const int P = cone_lanes_for_levels(a->n_grids, a->n_rays);
    const int64_t nb = ceil_div(a->n_rays > 0 ? a->n_rays : 1, kBlock / P);
    int64_t bytes = nb * (int64_t)(a->res[0] + a->res[1] + a->res[2] + 8 /* VoxelStore::kSlack */) * kBlock * 4;
    if (P >= 32) bytes += nb * (kBlock / 4) * (int64_t)(a->res[0] + a->res[1] + a->res[2] + 3) * 4 + 256;      // crossing-time arrays (K >= 4 lanes per slot)
    return bytes;
}
constexpr int64_t kConeVoxelBudgetBytes = 1ll << 30;
// lanes per ray of the cone count pass, 0 = the general lane-per-ray kernel.  Needs the larger workspace
// (nfa_traverse_workspace_bytes_for) announced through args.workspace_bytes.
static int cone_lanes_per_ray(const nfa_traverse_args *a) {
    if (!(a->step_size > 0.0f) || a->cone_angle == 0.0f) return 0;
    if (a->t_sorted || a->traverse_steps_limit > 0 || a->rays_mask) return 0;
    const int64_t max_rays = 32768;       // beyond, a lane per ray fills the chip (and the voxel planes grow with the ray count)
    if (opt(OPT_CONE, 1) == 0) return 0;
    if (a->n_rays > max_rays) return 0;
    // the per-voxel records grow with rays x (rx + ry + rz): 410 MB at 32 k rays of 4 x 128^3, twice that at 256^3.  Beyond a
    // budget of 1 GiB the general kernel serves the call (the workspace was unbounded and, cached by the caller's
    // allocator, stayed)
    if (cone_voxel_bytes(a) > kConeVoxelBudgetBytes) return 0;
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

static int rank_and_compact(uint64_t *bricks, const PackedLayout &L, hipStream_t s, int n_grids, int rx, int ry, int rz) {
    uint32_t *coarse = (uint32_t *)(bricks + L.off_coarse);
    uint32_t *prefix = (uint32_t *)(bricks + L.off_prefix);
    hipLaunchKernelGGL(rank_bricks_kernel, dim3(1), dim3(1024), 0, s, coarse, L.n_words, prefix, (int64_t *)(bricks + L.off_header));
    if (int rc = check_launch("rank_bricks_kernel")) return rc;
    hipLaunchKernelGGL(compact_bricks_kernel, dim3(blocks_for(L.n_bricks)), dim3(kBlock), 0, s,
                       bricks, L.n_bricks, coarse, prefix, bricks + L.off_compact);
    if (int rc = check_launch("compact_bricks_kernel")) return rc;
    const size_t map_bytes = (size_t)L.n_words * 4;
    if (map_bytes <= 64 * 1024)
        hipLaunchKernelGGL(brick_dist_kernel<true>, dim3(blocks_for(L.n_bricks)), dim3(kBlock), map_bytes, s, coarse, n_grids, (rx + 3) / 4,
                           (ry + 3) / 4, (rz + 3) / 4, (uint8_t *)(bricks + L.off_dist));
    else
        hipLaunchKernelGGL(brick_dist_kernel<false>, dim3(blocks_for(L.n_bricks)), dim3(kBlock), 0, s, coarse, n_grids, (rx + 3) / 4,
                           (ry + 3) / 4, (rz + 3) / 4, (uint8_t *)(bricks + L.off_dist));
    return check_launch("brick_dist_kernel");
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
    hipLaunchKernelGGL(pack_bricks_kernel<false>, dim3(pack_blocks(L.n_bricks)), dim3(kPackBlock), 0, s,
                       binaries, n_grids, rx, ry, rz, (rx + 3) / 4, (ry + 3) / 4, (rz + 3) / 4, bricks, coarse, header + 1,
                       (const float *)nullptr, (const double *)nullptr, 0, 0.0f, (uint8_t *)nullptr, (float *)nullptr);
    if (int rc = check_launch("pack_bricks_kernel")) return rc;
    return rank_and_compact(bricks, L, s, n_grids, rx, ry, rz);
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
    hipLaunchKernelGGL(pack_bricks_kernel<true>, dim3(pack_blocks(L.n_bricks)), dim3(kPackBlock), 0, s,
                       (const uint8_t *)nullptr, n_grids, rx, ry, rz, (rx + 3) / 4, (ry + 3) / 4, (rz + 3) / 4, bricks, coarse, header + 1,
                       occs, (const double *)workspace, nb, occ_thre, binaries, threshold_out);
    if (int rc = check_launch("pack_bricks_kernel<occs>")) return rc;
    return rank_and_compact(bricks, L, s, n_grids, rx, ry, rz);
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
// half of the bricks or more hold an occupied voxel (the reference's rand > 0.5 test grid: all of them), or nothing is known
static bool grid_is_noisy(const nfa_traverse_args *a) {
    const int64_t n_bricks = (int64_t)a->n_grids * ceil_div(a->res[0], 4) * ceil_div(a->res[1], 4) * ceil_div(a->res[2], 4);
    return a->n_nonempty_bricks < 0 || 2 * a->n_nonempty_bricks >= n_bricks;
}
// fewer than one brick in fifty holds an occupied voxel (thin structures, a small object in a large box): the rays do little but
// walk, and hardly ever list more than a few boundaries per part
static bool grid_is_near_empty(const nfa_traverse_args *a) {
    const int64_t n_bricks = (int64_t)a->n_grids * ceil_div(a->res[0], 4) * ceil_div(a->res[1], 4) * ceil_div(a->res[2], 4);
    return a->n_nonempty_bricks >= 0 && 50 * a->n_nonempty_bricks <= n_bricks;
}
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
            // Grids read from L2 (256^3, or dense at 128^3).  Round 4, eight scenes x six ray counts x two resolutions
            // (tools/experiments/r04_count_grid.py, tools/scene_sweep.py; profiles/r04_count_pass.md): the ray count alone cannot
            // choose.  At 48 k rays a noise grid wants 16 lanes per ray (8 overflow their boundary lists: 683 vs 380 us) and so does
            // a near-empty one (nothing but walking: 147 vs 121 us), while an object that fills 3-30 % of the bricks wants 8 (152
            // vs 192 us: every lane of a ray pays for resolving and stitching its boundaries), and 4 between 48 k and 96 k rays.
            // The share of non-empty bricks — which the caller read back with the packed grid's header — separates the three.
            // Round 3's rule (8 / 4 / 2 lanes from 8 k / 16 k / 36 k rays) was tuned on one object and cost 2-2.8x on the noise
            // and the thin-structure scenes; 2 lanes per ray never won.
            // With 24-entry boundary lists (plan_split; tools/experiments/r04_cap_sweep.py, last table of r04_count_pass.md) the
            // objects in between take 16 lanes up to 16 k rays and 8 up to 98 k: 43-48 us at 10 k rays of five scenes (57-61 with
            // 32-entry lists), 62-68 at 24 k (96-99), 110-148 at 48 k (150-155), 205-275 at 96 k (240-255 with 4 lanes).
            if (a->n_nonempty_bricks < 0) {                 // nothing known about the grid: 32-entry lists
                if (a->n_rays <= 8192) P = 16;
                else if (a->n_rays <= 49152) P = 8;
                else if (a->n_rays <= 98304) P = 4;
            } else if (grid_is_noisy(a) || grid_is_near_empty(a)) P = a->n_rays <= 98304 ? 16 : 1;
            else P = a->n_rays <= 16384 ? 16 : a->n_rays <= 98304 ? 8 : 1;
        }
        if (opt_is_set(OPT_SPLIT_P)) {                        // tuning knob: 1, 2, 4, 8 or 16
            P = (int)opt(OPT_SPLIT_P, P);
            if (sparse && (P == 2 || P == 4)) P = 8;          // (no 2- / 4-lane instances for sparse grids: they never won)
        }
    }
    return P;
}

// split kernels: a part keeps its boundaries in LDS, CAP of them (8 B each per lane).  With the
// sparse occupancy image in LDS (blob-like grid) 16 (8 at P = 16) is plenty; otherwise the grid may
// be dense or noisy — a boundary every other voxel for the reference's rand > 0.5 test grid — and
// LDS is free of the image, so the lists get 32 entries (the width of the lane's mask register).
struct SplitPlan { int P, cap, lds, blk, xt, seg, l2; GridView gv; int thr = 0; };     // thr: threads LAUNCHED per workgroup (0 = blk)
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
    if (opt(OPT_SEGMENTS, 1) == 0) return 0;
    if (a->n_rays > max_rays) return 0;
    int P = 2 * a->n_grids - 1 <= 8 ? 8 : 16;
    // 32 lanes per ray = 4 per segment slot (parts) while the launch has lanes to spare and the caller's workspace holds the
    // crossing-time arrays (nfa_traverse_workspace_bytes_for); NFA_SEG_P = 8 | 32 overrides
    const bool room = a->workspace_bytes >= ws_voxels_offset(a->n_rays) + seg_parts_bytes(a);
    if (P == 8 && room && a->n_rays <= 4096) P = 32;
    { const int v = (int)opt(OPT_SEG_P, 0); if (v == 8 && P == 32) P = 8; else if (v == 32 && P == 8 && room) P = 32; }
    return P;
}
static SplitPlan plan_split_(const nfa_traverse_args *a);
static SplitPlan plan_split(const nfa_traverse_args *a) {
    // (rounds 4-5 tried two more shapes of the crossing-time form here — workgroups launched narrower than their LDS stride,
    //  `split_thr`, and 32 lanes per ray in 1024-thread workgroups, `split_p = 32`: neither was ever faster than this one at any ray
    //  count (profiles/r05_count_pass.md section 5: the launch's time is one wave's dependent chain), removed in round 6)
    SplitPlan p = plan_split_(a);
    p.thr = p.blk;
    return p;
}
static SplitPlan plan_split_(const nfa_traverse_args *a) {
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
    { const int v = (int)opt(OPT_SPLIT_BLK, 0); if (v == 256 || (v == 512 && p.P == 16 && p.cap == 16)) p.blk = v; }
    // crossing-time arrays (512-thread form only; NFA_SPLIT_XT = 0 switches them off)
    p.xt = p.blk == 512 && opt(OPT_SPLIT_XT, 1) != 0;
    const int xt_bytes = p.xt ? (p.blk / p.P) * (a->res[0] + a->res[1] + a->res[2] + 3) * 4 : 0;
    // (the 512-thread form is alone on its CU: it may take the whole 160 KB)
    const int budget = p.blk == 512 ? 156 * 1024 : kLdsBudget;
    p.gv = make_view(a, p.cap * p.blk * 8 + xt_bytes, &p.lds, budget);
    if (p.gv.lds_compact_cap == 0 && p.xt) { p.xt = 0; p.gv = make_view(a, p.cap * p.blk * 8, &p.lds, budget); }
    if (p.gv.lds_compact_cap == 0 && p.blk != kBlock) { p.blk = kBlock; p.gv = make_view(a, p.cap * kBlock * 8, &p.lds); }
    if (p.gv.lds_compact_cap == 0) {
        // the image does not fit beside the lists: read from L2, and the workgroup's LDS is its boundary lists.  32 entries per
        // part = 64 KB, two workgroups per CU: what a noise grid needs (a boundary every other voxel).  16 entries = 32 KB, five
        // per CU (the form `l2` of the sparse grids): 10-40 % faster on box-like objects, but rays that graze a curved, voxelised
        // surface overflow them into the streaming mode — a hollow sphere at 8 k rays 68 vs 45 us, six of them 79 vs 58
        // (profiles/r04_count_pass.md) — so only a near-empty grid gets them (thin structures at 92 k rays: 228 vs 348 us).
        // 24 entries = 48 KB, three per CU, hold the curved scenes' parts AND keep most of the occupancy: the default in between.
        p.P = count_lanes_per_ray(a, false);
        const int cap_auto = a->n_nonempty_bricks < 0 || grid_is_noisy(a) ? 32 : (grid_is_near_empty(a) ? 16 : 24);
        int cap = (int)opt(OPT_SPLIT_CAP, cap_auto);
        if (p.P != 8 && p.P != 16) cap = 32;           // (2 and 4 lanes per ray exist with 32-entry lists only)
        if (cap == 24) {
            p.cap = 24;
            p.gv = make_view(a, p.cap * kBlock * 8, &p.lds, 0);
            return p;
        }
        if (cap == 16) {
            p.l2 = 1;
            p.cap = 16;
            p.gv = make_view(a, p.cap * kBlock * 8, &p.lds, 0);
            return p;
        }
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
    l2 = opt(OPT_SPLIT_L2, l2) != 0;
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
        const unsigned nbs = (unsigned)ceil_div(a->n_rays, plan.thr / P);
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
        hipLaunchKernelGGL((traverse_count_split_kernel<LDSO, PP, CAP>), dim3(nbs), dim3(kBlock), lds, s, *a, gv, block_sums, rs, FuseArgs{}); \
    } while (0)
        if (plan.l2) {
            // grid image from L2: 16-entry lists, 32 KB of LDS per workgroup
            if (P == 8) NFA_LAUNCH_SPLIT(false, 8, 16); else NFA_LAUNCH_SPLIT(false, 16, 16);
        } else if (lds_occ && plan.blk == 512 && plan.xt) {
            if (int rc = allow_lds(traverse_count_split_kernel<true, 16, 16, 512, true>, lds)) return rc;
            hipLaunchKernelGGL((traverse_count_split_kernel<true, 16, 16, 512, true>), dim3(nbs), dim3(plan.thr), lds, s, *a, gv, block_sums, rs, FuseArgs{});
        } else if (lds_occ && plan.blk == 512) {
            if (int rc = allow_lds(traverse_count_split_kernel<true, 16, 16, 512>, lds)) return rc;
            hipLaunchKernelGGL((traverse_count_split_kernel<true, 16, 16, 512>), dim3(nbs), dim3(512), lds, s, *a, gv, block_sums, rs, FuseArgs{});
        } else if (lds_occ) {
            NFA_LAUNCH_SPLIT(true, 16, 16);
        } else if (plan.cap == 24) {
            if (P == 8) NFA_LAUNCH_SPLIT(false, 8, 24); else NFA_LAUNCH_SPLIT(false, 16, 24);
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
        l2 = opt(OPT_COUNT_L2, l2) != 0;
        if (l2) gv = make_view(a, kEvBytes, &lds, 0);
    }
    const bool lds_occ = gv.lds_compact_cap > 0;
    // empty-space skipping (lattice form), the brick distances (half a byte per brick) read from L2 next to the brick word they
    // replace for every empty brick.  Measured (profiles/r05_count_pass.md): on incoherent rays — training batches, the replay's tiled
    // batch — the wave pays the longest far phase AND the longest near phase of its 64 lanes at every regime change and the macro steps
    // lose: 10^6 rays 621 us voxel by voxel, 728 with them; on the pixel-ordered rays of a frame they win, 276 -> 189.  So: unset =
    // every WAVE decides from its rays (wave_rays_coherent); `skip` = 0 never, 1 always.  (Round 5 also staged the distances in LDS,
    // `skip = 2`: 208 us on the frame, 828 on the tiled batch — slower than L2 in both; removed in round 6.)
    int skip = 0;
    if (lattice) {
        skip = (int)opt(OPT_SKIP, 1);
        gv.skip_auto = opt_is_set(OPT_SKIP) ? 0 : 1;
    }
#define NFA_LAUNCH_COUNT(EVM, LAT, LDSO, SK)                                                                                    \
    do {                                                                                                                        \
        if (int rc = allow_lds(traverse_count_kernel<EVM, LAT, LDSO, SK>, lds)) return rc;                                       \
        hipLaunchKernelGGL((traverse_count_kernel<EVM, LAT, LDSO, SK>), dim3(nb), dim3(kBlock), lds, s, *a, gv, block_sums, rs); \
    } while (0)
#define NFA_COUNT_LAT(EVM, LDSO)                                                      \
    do {                                                                              \
        if (skip == 1) NFA_LAUNCH_COUNT(EVM, true, LDSO, 1);                          \
        else NFA_LAUNCH_COUNT(EVM, true, LDSO, 0);                                    \
    } while (0)
#define NFA_COUNT_EV(EVM)                                                              \
    do {                                                                               \
        if (lattice) { if (lds_occ) NFA_COUNT_LAT(EVM, true); else NFA_COUNT_LAT(EVM, false); }   \
        else         { if (lds_occ) NFA_LAUNCH_COUNT(EVM, false, true, 0); else NFA_LAUNCH_COUNT(EVM, false, false, 0); } \
    } while (0)
    if (evm == EV_PRE) NFA_COUNT_EV(EV_PRE);
    else if (evm == EV_ONE) NFA_COUNT_EV(EV_ONE);
    else NFA_COUNT_EV(EV_MANY);
#undef NFA_COUNT_EV
#undef NFA_COUNT_LAT
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
    const int64_t n_sums = ceil_div(a->n_rays, plan.thr / P) * (plan.thr / kWave);      // one triple per wave of the count launch
    hipLaunchKernelGGL(traverse_offsets_kernel, dim3(nb), dim3(kBlock), 0, s, a->iv_cnts, a->iv_starts, a->sm_cnts,
                       a->sm_starts, a->n_rays, (const int64_t *)workspace, P * kWavesPerBlock, n_sums, a->totals,
                       (int64_t *)((uint8_t *)const_cast<void *>(workspace) + ws_totals_offset(a->n_rays)), stamp);
    return check_launch("traverse_offsets_kernel");
}

// ---- the whole sampling call as ONE launch (sample_fused.hpp) ------------------------------------------------------------------
// Exists for the count pass the training step and the eval loop's 8192-ray slices take: one level, constant step, no interval
// outputs, the 512-thread crossing-time split kernel at full width, one round of workgroups (<= one per CU: every workgroup a
// look-back could wait for is resident).  `fused_sample` = 0 switches it off.
static bool sample_fusable(const nfa_traverse_args *a, const SplitPlan &plan) {
    if (opt(OPT_FUSED_SAMPLE, 1) == 0) return false;
    if (plan.seg || plan.P != 16 || plan.blk != 512 || !plan.xt || plan.l2 || plan.thr != 512) return false;
    if (plan.gv.lds_compact_cap <= 0) return false;
    if (a->iv_cnts || a->iv_vals || !a->totals) return false;
    const int64_t nb = ceil_div(a->n_rays, 512 / 16);
    return nb <= kNumCU && nb <= kSyncMaxBlocks;
}

NFA_EXPORT int nfa_traverse_sample_fused(const nfa_traverse_args *a)
{
    if (!a || validate_traverse(a) != NFA_OK || a->n_rays == 0) return 0;
    return sample_fusable(a, plan_split(a)) ? 1 : 0;
}

NFA_EXPORT int nfa_traverse_sample(const nfa_traverse_args *a, void *workspace, int64_t capacity, int64_t stamp, void *sync,
                                   int32_t *fused, void *stream)
{
    if (fused) *fused = 0;
    if (int rc = validate_traverse(a)) return rc;
    NFA_REQUIRE(a->totals != nullptr, "traverse_sample: totals is NULL");
    NFA_REQUIRE(capacity >= 0, "traverse_sample: capacity < 0");
    const bool outputs = a->sm_ray_indices || a->t_starts || a->sm_vals;
    if (a->t_starts) NFA_REQUIRE(a->t_ends != nullptr, "traverse_sample: t_starts without t_ends");
    hipStream_t s = (hipStream_t)stream;
    if (a->n_rays > 0 && sync) {
        NFA_REQUIRE(workspace != nullptr, "traverse_sample: workspace is NULL");
        const SplitPlan plan = plan_split(a);
        // (long rays: the emit half of the single launch is ONE wave per four rays, and beyond ~64 samples per ray — a capacity, the
        //  caller's guess with its margin, of more than 80 per ray — the emit kernel's spread over the whole chip wins: SURVEY 8d's M1
        //  sphere, 4096 rays x 84 samples, 49.5 us fused against 47.2 in three launches; the training step's 39 per ray: 36 against 42)
        const bool long_rays = outputs && capacity > 80 * a->n_rays && opt(OPT_FUSED_SAMPLE, 1) != 2;      // (`fused_sample` = 2: whatever the rays' length)
        if (!long_rays && sample_fusable(a, plan)) {
            const RunStore rs = make_runs(workspace, a->n_rays);
            FuseArgs fz;
            fz.sync = (uint64_t *)sync;
            fz.capacity = outputs ? capacity : 0;
            fz.stamp = stamp;
            fz.totals_dev = (int64_t *)((uint8_t *)workspace + ws_totals_offset(a->n_rays));
            fz.spin = sync_spin_ticks();
            fz.seg_cap = 128;
            while (fz.seg_cap < rs.max_runs) fz.seg_cap *= 2;
            // the emit tail's segment lists take over the workgroup's LDS once its waves have counted
            const int emit_lds = (512 / kWave) * emit_lds_per_wave(fz.seg_cap, false);
            const int lds = plan.lds > emit_lds ? plan.lds : emit_lds;
            if (lds + 256 <= kLdsPerCU) {
                const unsigned nbs = (unsigned)ceil_div(a->n_rays, 512 / 16);
                if (int rc = allow_lds(traverse_count_split_kernel<true, 16, 16, 512, true, true>, lds)) return rc;
                hipLaunchKernelGGL((traverse_count_split_kernel<true, 16, 16, 512, true, true>), dim3(nbs), dim3(512), lds, s, *a, plan.gv,
                                   (int64_t *)workspace, rs, fz);
                if (fused) *fused = 1;
                return check_launch("traverse_count_split_kernel<fused>");
            }
        }
    }
    if (int rc = nfa_traverse_count(a, workspace, stream)) return rc;
    if (int rc = nfa_traverse_offsets_stamped(a, workspace, stamp, stream)) return rc;
    if (a->n_rays > 0 && outputs && capacity > 0) return nfa_traverse_emit_speculative(a, workspace, capacity, stream);
    return NFA_OK;
}

static int launch_fill(const nfa_traverse_args *a, int skip_empty, int rewrite_counts, const uint16_t *only_overflow, hipStream_t s)
{
    int lds = 0;
    GridView gv = make_view(a, 0, &lds);       // no boundary lists in the general walk
    const unsigned nb = (unsigned)ceil_div(a->n_rays, kBlock);
    {   // (image in LDS only while one round of workgroups holds every ray, as in nfa_traverse_count)
        bool l2 = gv.lds_compact_cap > 0 && (int64_t)nb > (int64_t)kNumCU * (kLdsPerCU / (lds > 0 ? lds : 1));
        l2 = opt(OPT_COUNT_L2, l2) != 0;
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
    return (int)opt(OPT_EMIT, 0);
}
// The emit pass of a call: cone_angle == 0 takes the tile form (its own kernel: a wave expands a block of consecutive rays,
// emit_pass.hpp) unless the `emit` option forces one of the older forms; cone_angle != 0 takes traverse_emit_kernel (ray groups
// or a lane per sample, chosen on the device).  `speculative`: see nfa_traverse_emit_speculative.
static int launch_emit(const nfa_traverse_args *a, const RunStore &rs, int64_t capacity, const int64_t *n_dev, int speculative, hipStream_t s);
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
        if (int rc = launch_emit(a, rs, n_samples, n_dev, 0, s)) return rc;
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
    return launch_emit(a, rs, capacity, n_dev, 1, (hipStream_t)stream);
}

static int launch_emit(const nfa_traverse_args *a, const RunStore &rs, int64_t capacity, const int64_t *n_dev, int speculative, hipStream_t s)
{
    const int hint = emit_hint();
    if (a->cone_angle == 0.0f && a->step_size > 0.0f && (hint == 0 || hint == 3)) {
        const int64_t R = a->n_rays;
        // rays per wave: enough waves for every SIMD (1024) to hold two or more; segment-list capacity: a ray's runs must fit
        // (run_capacity), 128 at least — a 64-ray block of a NeRF-like scene has 40-80 runs
        const int rb_log2 = (int)opt(OPT_EMIT_RB, R <= 2048 ? 0 : R <= 4096 ? 1 : R <= 8192 ? 2 : R <= 32768 ? 3 : R <= 131072 ? 4 : R <= 262144 ? 5 : 6);      // (tools/experiments/r04_emit_rb*.{sh,py})
        // the kernel may take FEWER rays per wave than that when the call turns out to have many runs per ray (emit_rays_per_wave_log2:
        // it sees the totals, the host at a speculative launch does not) — down to rb_min, which the grid is sized for; a forced
        // `emit_rb` is taken as it is
        const int rb_min = opt_is_set(OPT_EMIT_RB) ? rb_log2 : (rb_log2 > 3 ? rb_log2 - 3 : 0);
        int seg_cap = 128;
        while (seg_cap < rs.max_runs) seg_cap *= 2;
        const int words = kEmitSegWords + (a->iv_vals ? 2 : 0);
        const unsigned lds = (unsigned)(kWavesPerBlock * (seg_cap * 4 * words + kEmitRayBaseBytes));
        const int64_t n_rb = ceil_div(R, (int64_t)1 << rb_min), cap = (int64_t)kNumCU * 16;
        const int64_t nb = ceil_div(n_rb, kWavesPerBlock);
        const dim3 g((unsigned)(nb < cap ? nb : cap)), b(kBlock);
        if (a->iv_vals) hipLaunchKernelGGL(traverse_emit_tiles_kernel<true>, g, b, lds, s, *a, rs, capacity, n_dev, speculative, rb_log2, rb_min, seg_cap);
        else hipLaunchKernelGGL(traverse_emit_tiles_kernel<false>, g, b, lds, s, *a, rs, capacity, n_dev, speculative, rb_log2, rb_min, seg_cap);
        return check_launch("traverse_emit_tiles_kernel");
    }
    const unsigned nb_s = blocks_for(capacity), nb_r = emit_ray_blocks(a->n_rays);
    hipLaunchKernelGGL(traverse_emit_kernel, dim3(nb_s > nb_r ? nb_s : nb_r), dim3(kBlock), 0, s, *a, rs, capacity, n_dev, speculative, hint == 3 ? 0 : hint);
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

#ifdef NFA_FUSE_TRACE
// instrumentation builds only (tools/fuse_trace.py): the fused sampling kernel's per-workgroup stamps, then cleared
extern "C" __attribute__((visibility("default"))) int nfa_debug_fuse_trace(unsigned long long *out /* [512][4] */) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(nfa::g_fuse_trace), 512 * 4 * sizeof(unsigned long long)) != hipSuccess) return 1;
    void *sym = nullptr;
    if (hipGetSymbolAddress(&sym, HIP_SYMBOL(nfa::g_fuse_trace)) != hipSuccess) return 1;
    return hipMemset(sym, 0, 512 * 4 * sizeof(unsigned long long)) == hipSuccess ? 0 : 1;
}
#endif

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
