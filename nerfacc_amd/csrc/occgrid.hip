// occgrid.hip — occupancy-grid maintenance for gfx950.
//
// The reference keeps its grid with ~25 small ATen ops per level and update
// (OccGridEstimator._update, occ_grid.py:366-404): gather voxel coordinates, jitter, scale to
// world space, query the field, gather / decay / maximum / scatter into `occs`, masked mean,
// threshold.  Here everything either side of the user's occ_eval_fn is three launches:
//   cell points   voxel ids (+ the caller's U[0,1) jitter) -> world positions      :377-384
//   EMA-max       occs[id] = max(occs[id] * decay, occ)                              :388-390
//   threshold     mean of the non-negative occs -> binaries = occs > min(mean, thre) :392-404
// Random numbers stay with the caller (torch's generator, drawn in the reference's order), so
// the result is a function of the same inputs as the reference's.  All HBM-streaming, one
// element per lane; arithmetic order follows the torch expressions (no contraction: the
// library is built with -ffp-contract=off).
#include "common.hpp"
#include "lookback.hpp"

namespace nfa {
namespace {

// occ_grid.py:377-384: x = (coords + rand) / resolution ; x = lo + x * (hi - lo)
__global__ __launch_bounds__(kBlock) void grid_cell_points_kernel(
    const int64_t *__restrict__ cell_ids, int64_t n, const float *__restrict__ jitter,
    int rx, int ry, int rz, const float *__restrict__ aabb, float *__restrict__ points)
{
    const float lo[3] = {aabb[0], aabb[1], aabb[2]};
    const float ext[3] = {aabb[3] - lo[0], aabb[4] - lo[1], aabb[5] - lo[2]};
    const float resf[3] = {(float)rx, (float)ry, (float)rz};
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
        const int64_t id = cell_ids ? cell_ids[i] : i;
        const int64_t yz = (int64_t)ry * rz;
        const int c[3] = {(int)(id / yz), (int)((id / rz) % ry), (int)(id % rz)};   // x-major voxel order (grid.cu:187-192)
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const float unit = ((float)c[a] + jitter[3 * i + a]) / resf[a];
            points[3 * i + a] = lo[a] + unit * ext[a];
        }
    }
}

// torch.maximum semantics (NaN wins)
__device__ __forceinline__ float max_nan(float a, float b) { return (a != a || b != b) ? __builtin_nanf("") : fmaxf(a, b); }

// EMA-max in two launches: every candidate is formed from the OLD value of its cell before any
// cell is written (ids may repeat — uniform draws with replacement, :350-352 — and the reference's
// gather-then-scatter never sees a half-updated grid); with repeats one of the candidates wins,
// as with index_put_.
__global__ __launch_bounds__(kBlock) void grid_ema_candidates_kernel(
    const float *__restrict__ occs, const int64_t *__restrict__ cell_ids, int64_t n,
    const float *__restrict__ occ_new, float decay, float *__restrict__ cand)
{
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
        const int64_t id = cell_ids ? cell_ids[i] : i;
        cand[i] = max_nan(occs[id] * decay, occ_new[i]);
    }
}
__global__ __launch_bounds__(kBlock) void grid_scatter_kernel(
    float *__restrict__ occs, const int64_t *__restrict__ cell_ids, int64_t n, const float *__restrict__ cand)
{
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock)
        occs[cell_ids ? cell_ids[i] : i] = cand[i];
}

// sum and count of the visible cells (occs >= 0; invisible ones hold -1, :330-332); one
// {sum, count} pair of doubles per workgroup, combined in a fixed order by the threshold kernel
__global__ __launch_bounds__(kBlock) void grid_mean_partials_kernel(const float *__restrict__ occs, int64_t n, double *__restrict__ partials)
{
    __shared__ double s_sum[kWavesPerBlock], s_cnt[kWavesPerBlock];
    double sum = 0.0, cnt = 0.0;
    // four cells per load (the grid is a whole tensor: 16-byte aligned), the n % 4 cells at the end by the first threads
    const int64_t n4 = (reinterpret_cast<uintptr_t>(occs) & 15u) == 0 ? n >> 2 : 0;
    const float4 *occs4 = reinterpret_cast<const float4 *>(occs);
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n4; i += (int64_t)gridDim.x * kBlock) {
        const float4 v = occs4[i];
        if (v.x >= 0.0f) { sum += (double)v.x; cnt += 1.0; }
        if (v.y >= 0.0f) { sum += (double)v.y; cnt += 1.0; }
        if (v.z >= 0.0f) { sum += (double)v.z; cnt += 1.0; }
        if (v.w >= 0.0f) { sum += (double)v.w; cnt += 1.0; }
    }
    for (int64_t i = 4 * n4 + (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
        const float v = occs[i];
        if (v >= 0.0f) { sum += (double)v; cnt += 1.0; }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) { sum += __shfl_down(sum, off, 64); cnt += __shfl_down(cnt, off, 64); }
    if (lane_id() == 0) { s_sum[threadIdx.x >> 6] = sum; s_cnt[threadIdx.x >> 6] = cnt; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double a = 0.0, b = 0.0;
        for (int w = 0; w < kWavesPerBlock; ++w) { a += s_sum[w]; b += s_cnt[w]; }
        partials[2 * blockIdx.x] = a;
        partials[2 * blockIdx.x + 1] = b;
    }
}
// binaries = occs > min(mean, occ_thre)  (:392-404; an all-invisible grid has a NaN mean: nothing passes)
__global__ __launch_bounds__(kBlock) void grid_threshold_kernel(
    const float *__restrict__ occs, int64_t n, const double *__restrict__ partials, int n_partials, float occ_thre,
    uint8_t *__restrict__ binaries, float *__restrict__ thre_out)
{
    __shared__ float s_thre;
    if (threadIdx.x < 64) {                       // first wave: fixed-order tree over the <= 128 partial pairs
        const float th = threshold_from_partials(partials, n_partials, occ_thre);
        if (threadIdx.x == 0) {
            s_thre = th;
            if (blockIdx.x == 0 && thre_out) *thre_out = th;
        }
    }
    __syncthreads();
    const float thre = s_thre;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock)
        binaries[i] = occs[i] > thre ? 1 : 0;
}

// OccGridEstimator.mark_invisible_cells (occ_grid.py:262-332; ~15 ATen ops per 32^3-cell chunk there, incl. two
// batched matmuls that materialise [n_cams, 3, chunk] tensors): one lane per cell, cameras in a loop.
//   x = coords / (res - 1); xyz = lo + x * (hi - lo); p = R_c xyz + T_c; uvd = K_c p; uv = uvd.xy / uvd.z
//   in_image = uvd.z >= 0 & 0 <= u < width & 0 <= v < height
//   valid = any_c(uvd.z >= near & in_image) & !any_c(uvd.z < near & in_image);  occs[cell] = valid ? 0 : -1
// Dot products are accumulated left to right without contraction; a BLAS may order / fuse them differently, which
// can only matter for a cell whose projection lies within an ulp of an image border.
__global__ __launch_bounds__(kBlock) void grid_mark_invisible_kernel(
    float *__restrict__ occs, const int64_t *__restrict__ cell_ids, int64_t n, int rx, int ry, int rz,
    const float *__restrict__ aabb, const float *__restrict__ w2c_R, const float *__restrict__ w2c_T,
    const float *__restrict__ K, int n_cams, int k_stride, float width, float height, float near_plane)
{
    const float lo[3] = {aabb[0], aabb[1], aabb[2]};
    const float ext[3] = {aabb[3] - lo[0], aabb[4] - lo[1], aabb[5] - lo[2]};
    const float den[3] = {(float)(rx - 1), (float)(ry - 1), (float)(rz - 1)};
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
        const int64_t id = cell_ids ? cell_ids[i] : i;
        const int64_t yz = (int64_t)ry * rz;
        const int c[3] = {(int)(id / yz), (int)((id / rz) % ry), (int)(id % rz)};
        float w[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) w[a] = lo[a] + ((float)c[a] / den[a]) * ext[a];
        bool covered = false, too_near = false;
        for (int cam = 0; cam < n_cams; ++cam) {          // camera data: wave-uniform addresses (scalar loads)
            const float *R = w2c_R + 9 * cam, *T = w2c_T + 3 * cam, *Kc = K + (int64_t)k_stride * cam;
            float p[3], u[3];
#pragma unroll
            for (int r = 0; r < 3; ++r) p[r] = ((R[3 * r] * w[0] + R[3 * r + 1] * w[1]) + R[3 * r + 2] * w[2]) + T[r];
#pragma unroll
            for (int r = 0; r < 3; ++r) u[r] = (Kc[3 * r] * p[0] + Kc[3 * r + 1] * p[1]) + Kc[3 * r + 2] * p[2];
            const float px = u[0] / u[2], py = u[1] / u[2];
            const bool in_image = (u[2] >= 0.0f) && (px >= 0.0f) && (px < width) && (py >= 0.0f) && (py < height);
            covered = covered || (in_image && u[2] >= near_plane);
            too_near = too_near || (in_image && u[2] < near_plane);
        }
        occs[id] = (covered && !too_near) ? 0.0f : -1.0f;
    }
}


// ----------------------------------------------------------------------------------------
// The occupied cells of one level, ascending — what `torch.nonzero(binaries[lvl].flatten())[:, 0]` returns (occ_grid.py:356) — as ONE
// launch.  A thread takes CPT consecutive cells (16-byte loads of the bool bytes), the wave and the workgroup rank them with ballots'
// cousins (popcounts + the DPP sum), the workgroups hand their counts on through the caller's sync block (lookback.hpp: static ids,
// every workgroup resident, bounded waits) and every thread stores the indices of its cells.  A look-back that gives up is not
// reported to the host (this path has no read-back: the count comes from the packed grid's header): the workgroup counts the cells
// before its own itself — slow, correct, and it has never been seen to happen.
// ----------------------------------------------------------------------------------------
template <int CPT>
__global__ __launch_bounds__(kBlock) void occupied_cells_kernel(const uint8_t *__restrict__ cells, int64_t n_cells, int64_t *__restrict__ out,
                                                                int64_t capacity, uint64_t *__restrict__ sync, uint64_t spin)
{
    static_assert(CPT % 16 == 0, "whole 16-byte loads");
    __shared__ int64_t s_w[kWavesPerBlock];
    __shared__ int64_t s_pre;
    const int lane = lane_id(), wv = wave_in_block();
    const int64_t b = blockIdx.x, nb = gridDim.x;
    const int64_t c0 = (b * kBlock + threadIdx.x) * CPT;
    uint32_t w[CPT / 4];
#pragma unroll
    for (int k = 0; k < CPT / 16; ++k) {
        uint4 v = make_uint4(0, 0, 0, 0);
        const int64_t c = c0 + 16 * k;
        if (c + 16 <= n_cells) v = *reinterpret_cast<const uint4 *>(cells + c);
        else {
            uint32_t t[4] = {0, 0, 0, 0};
            for (int j = 0; j < 16; ++j) if (c + j < n_cells && cells[c + j]) t[j >> 2] |= 1u << (8 * (j & 3));
            v = make_uint4(t[0], t[1], t[2], t[3]);
        }
        w[4 * k] = v.x; w[4 * k + 1] = v.y; w[4 * k + 2] = v.z; w[4 * k + 3] = v.w;
    }
    int mine = 0;
#pragma unroll
    for (int k = 0; k < CPT / 4; ++k) {
        // a bool byte is 0 or 1 (torch): the four bytes of a word add up in its popcount
        mine += __popc(w[k] & 0x01010101u);
    }
    // exclusive rank inside the wave (DPP scan of the lane counts), wave totals through LDS
    int64_t incl = mine;
    { const int64_t u = dpp_i64<kDppRowShr + 1>(incl); if ((lane & 15) >= 1) incl += u; }
    { const int64_t u = dpp_i64<kDppRowShr + 2>(incl); if ((lane & 15) >= 2) incl += u; }
    { const int64_t u = dpp_i64<kDppRowShr + 4>(incl); if ((lane & 15) >= 4) incl += u; }
    { const int64_t u = dpp_i64<kDppRowShr + 8>(incl); if ((lane & 15) >= 8) incl += u; }
    { const int64_t u = dpp_i64<kDppRowBcast15>(incl); if (lane & 16) incl += u; }
    { const int64_t u = dpp_i64<kDppRowBcast31>(incl); if (lane & 32) incl += u; }
    if (lane == 63) s_w[wv] = incl;
    __syncthreads();
    if (wv == 0) {
        const int64_t tot = s_w[0] + s_w[1] + s_w[2] + s_w[3];
        int64_t excl = sync_publish_and_lookback(sync, b, tot, 0, 0, lane, spin);
        if (excl < 0) {                                   // (bounded wait ran out: count the cells before this workgroup directly)
            int64_t cnt = 0;
            const int64_t end = b * kBlock * CPT;
            for (int64_t c = (int64_t)lane * 16; c < end; c += 64 * 16) {
                const uint4 v = *reinterpret_cast<const uint4 *>(cells + c);        // (end is a multiple of 16 and <= n_cells here)
                cnt += __popc(v.x & 0x01010101u) + __popc(v.y & 0x01010101u) + __popc(v.z & 0x01010101u) + __popc(v.w & 0x01010101u);
            }
            excl = wave_sum_i64(cnt);
        }
        if (lane == 0) s_pre = excl;
        sync_leave(sync, nb, lane);
    }
    __syncthreads();
    int64_t dst = s_pre + (incl - mine);
#pragma unroll
    for (int k = 0; k < kWavesPerBlock - 1; ++k) dst += k < wv ? s_w[k] : 0;
#pragma unroll
    for (int k = 0; k < CPT / 4; ++k) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if ((w[k] >> (8 * j)) & 1u) {
                if (dst < capacity) out[dst] = c0 + 4 * k + j;
                ++dst;
            }
        }
    }
}

}  // namespace
}  // namespace nfa

namespace nfa {
int launch_grid_mean_partials(const float *occs, int64_t n_cells, double *partials, hipStream_t s) {
    int nb = (int)blocks_for(ceil_div(n_cells, 4));
    if (nb > kReduceBlocks) nb = kReduceBlocks;
    hipLaunchKernelGGL(grid_mean_partials_kernel, dim3(nb), dim3(kBlock), 0, s, occs, n_cells, partials);
    return nb;
}
}  // namespace nfa

using namespace nfa;

NFA_EXPORT int nfa_grid_mark_invisible(float *occs, const int64_t *cell_ids, int64_t n, int32_t rx, int32_t ry, int32_t rz,
                                       const float *aabb, const float *w2c_R, const float *w2c_T, const float *K,
                                       int32_t n_cams, int32_t k_shared, float width, float height, float near_plane, void *stream)
{
    NFA_REQUIRE(n >= 0 && n_cams >= 0, "grid_mark_invisible: negative size");
    NFA_REQUIRE(rx > 0 && ry > 0 && rz > 0, "grid_mark_invisible: bad resolution %d x %d x %d", rx, ry, rz);
    if (n == 0) return NFA_OK;
    NFA_REQUIRE(occs && aabb && (n_cams == 0 || (w2c_R && w2c_T && K)), "grid_mark_invisible: NULL pointer");
    NFA_REQUIRE(cell_ids || n <= (int64_t)rx * ry * rz, "grid_mark_invisible: n exceeds the cells of one level");
    hipLaunchKernelGGL(grid_mark_invisible_kernel, dim3(blocks_for(n)), dim3(kBlock), 0, (hipStream_t)stream, occs, cell_ids, n,
                       rx, ry, rz, aabb, w2c_R, w2c_T, K, n_cams, k_shared ? 0 : 9, width, height, near_plane);
    return check_launch("grid_mark_invisible_kernel");
}

NFA_EXPORT int nfa_grid_cell_points(const int64_t *cell_ids, int64_t n, const float *jitter,
                                    int32_t rx, int32_t ry, int32_t rz, const float *aabb,
                                    float *points, void *stream)
{
    NFA_REQUIRE(n >= 0, "grid_cell_points: n < 0");
    NFA_REQUIRE(rx > 0 && ry > 0 && rz > 0, "grid_cell_points: bad resolution %d x %d x %d", rx, ry, rz);
    if (n == 0) return NFA_OK;
    NFA_REQUIRE(jitter && aabb && points, "grid_cell_points: NULL pointer");
    NFA_REQUIRE(cell_ids || n <= (int64_t)rx * ry * rz, "grid_cell_points: n exceeds the cells of one level");
    hipLaunchKernelGGL(grid_cell_points_kernel, dim3(blocks_for(n)), dim3(kBlock), 0, (hipStream_t)stream,
                       cell_ids, n, jitter, rx, ry, rz, aabb, points);
    return check_launch("grid_cell_points_kernel");
}

NFA_EXPORT int nfa_grid_ema_update(float *occs, const int64_t *cell_ids, int64_t n, const float *occ_new,
                                   float ema_decay, float *scratch, void *stream)
{
    NFA_REQUIRE(n >= 0, "grid_ema_update: n < 0");
    if (n == 0) return NFA_OK;
    NFA_REQUIRE(occs && occ_new && scratch, "grid_ema_update: NULL pointer");
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(grid_ema_candidates_kernel, dim3(blocks_for(n)), dim3(kBlock), 0, s, occs, cell_ids, n, occ_new, ema_decay, scratch);
    hipLaunchKernelGGL(grid_scatter_kernel, dim3(blocks_for(n)), dim3(kBlock), 0, s, occs, cell_ids, n, scratch);
    return check_launch("grid_ema_update");
}

NFA_EXPORT int64_t nfa_grid_threshold_workspace_bytes(void) { return (int64_t)kReduceBlocks * 2 * sizeof(double); }

NFA_EXPORT int nfa_grid_threshold(const float *occs, int64_t n_cells, float occ_thre, void *workspace,
                                  uint8_t *binaries, float *threshold_out, void *stream)
{
    NFA_REQUIRE(n_cells >= 0, "grid_threshold: n_cells < 0");
    if (n_cells == 0) return NFA_OK;
    NFA_REQUIRE(occs && workspace && binaries, "grid_threshold: NULL pointer");
    hipStream_t s = (hipStream_t)stream;
    const int nb = launch_grid_mean_partials(occs, n_cells, (double *)workspace, s);
    hipLaunchKernelGGL(grid_threshold_kernel, dim3(blocks_for(n_cells)), dim3(kBlock), 0, s, occs, n_cells,
                       (const double *)workspace, nb, occ_thre, binaries, threshold_out);
    return check_launch("grid_threshold");
}

// occupied cells of one level in ascending order: out[0 .. count) = the flat indices c with cells[c] != 0 (what torch.nonzero
// returns for the flattened level, occ_grid.py:356).  `capacity`: entries `out` holds (the caller knows the count from the packed
// grid's header; cells beyond it are not stored).  `sync`: NFA_SYNC_BYTES, zero, left zero (include/nerfacc_hip.h).
NFA_EXPORT int nfa_grid_occupied_cells(const uint8_t *cells, int64_t n_cells, int64_t *out, int64_t capacity, void *sync, void *stream)
{
    NFA_REQUIRE(n_cells >= 0 && capacity >= 0, "grid_occupied_cells: negative size");
    if (n_cells == 0 || capacity == 0) return NFA_OK;
    NFA_REQUIRE(cells && out && sync, "grid_occupied_cells: NULL pointer");
    NFA_REQUIRE((reinterpret_cast<uintptr_t>(cells) & 15u) == 0, "grid_occupied_cells: cells must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    // cells per thread: 16 while that keeps the launch inside the sync block's states and on the chip at once, else 64 / 256
    const int64_t per16 = ceil_div(n_cells, (int64_t)kBlock * 16), per64 = ceil_div(n_cells, (int64_t)kBlock * 64), per256 = ceil_div(n_cells, (int64_t)kBlock * 256);
    const int64_t limit = std::min<int64_t>(kSyncMaxBlocks, (int64_t)kNumCU * 4);
    if (per16 <= limit) hipLaunchKernelGGL((occupied_cells_kernel<16>), dim3((unsigned)per16), dim3(kBlock), 0, s, cells, n_cells, out, capacity, (uint64_t *)sync, sync_spin_ticks());
    else if (per64 <= limit) hipLaunchKernelGGL((occupied_cells_kernel<64>), dim3((unsigned)per64), dim3(kBlock), 0, s, cells, n_cells, out, capacity, (uint64_t *)sync, sync_spin_ticks());
    else {
        NFA_REQUIRE(per256 <= limit, "grid_occupied_cells: %lld cells are more than one launch ranks (%lld)", (long long)n_cells, (long long)(limit * kBlock * 256));
        hipLaunchKernelGGL((occupied_cells_kernel<256>), dim3((unsigned)per256), dim3(kBlock), 0, s, cells, n_cells, out, capacity, (uint64_t *)sync, sync_spin_ticks());
    }
    return check_launch("occupied_cells_kernel");
}
