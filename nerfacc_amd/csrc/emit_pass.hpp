// pass 2 of the two-pass traversal: the emit kernel and its two forms (traverse_emit_kernel) — part of grid.hip (included there, inside namespace nfa::{anonymous}, after the helpers it uses); moved out
// of grid.hip in round 3 for size only: the text is unchanged.
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

