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
    // the constant step of cone_angle == 0 as the lattice's integer increment (lattice.hpp): mantissa with the hidden bit, exponent
    const uint32_t lat_db = nfa_f2u(march_dt(0.0f, 0.0f, step_size));
    const int lat_ed = (int)((lat_db >> 23) & 0xffu);
    const uint32_t lat_D = (lat_db & 0x7fffffu) | 0x800000u;
    const bool lat_ok = lat_ed >= 1 && lat_ed < 255 && (lat_db >> 31) == 0;
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
                    // cone_angle == 0, round 4: the pass's 64 lattice points IN CLOSED FORM when they share base's binade.  Inside a
                    // binade a step adds the same integer number of ulps c = RN(dt / ulp) to the mantissa (lattice.hpp: exact while
                    // the sum stays below 2^(e+1) and dt / ulp is not exactly half-way), and while the mantissa does not carry the
                    // float's BIT PATTERN advances by c too: lane gl's samples are bits(base) + (4 gl + e) c — five integer adds
                    // instead of the 63-add chain every lane ran before (331 us for the 38 M samples of 10^6 rays, 1.7 TB/s of
                    // stores, instruction-bound).  A pass that crosses a binade edge (about a dozen between dt and t = 8), a tie or
                    // an irregular value takes the chain below: the same adds the reference performs.
                    if (cone == 0.0f) {
                        const uint32_t tb = nfa_f2u(base);
                        const int e = (int)((tb >> 23) & 0xffu), sh = e - lat_ed;
                        const bool ok = lat_ok && (tb >> 31) == 0 && e >= 1 && e < 254 && sh >= 1 && sh <= 24;
                        const uint32_t shc = ok ? (uint32_t)sh : 1u;
                        const uint32_t c0 = lat_D >> shc, frac = lat_D & ((1u << shc) - 1u), half = (1u << shc) >> 1;
                        const uint32_t inc = c0 + (frac > half ? 1u : 0u);
                        const uint32_t mant = (tb & 0x7fffffu) | 0x800000u;
                        if (ok && frac != half && mant + 64u * inc < (1u << 24)) {
                            const uint32_t b0 = tb + (uint32_t)(4 * gl) * inc;
#pragma unroll
                            for (int e2 = 0; e2 < 5; ++e2) sv[e2] = nfa_u2f(b0 + (uint32_t)e2 * inc);
                            base = nfa_u2f(tb + 64u * inc);
                            goto pass_ready;
                        }
                    }
                    {
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
                    }
                pass_ready:
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

// pass 2, tile form (round 4, cone_angle == 0): A WAVE TAKES A BLOCK OF CONSECUTIVE RAYS AND DEALS ITS RUNS' SAMPLES TO ITS LANES.
// What the two forms above leave on the table at frame scale (38 M samples of 10^6 rays: 1.65 TB/s of stores, a quarter of what
// this part writes): the lane-per-sample form pays ~14 dependent loads per sample (two searches in global memory), the ray groups
// leave most lanes idle — a ray has 38 samples on average and a group's pass holds 64, rays without samples hold a group for
// nothing, and a wave waits for the longest of its four rays.  Here the unit of work is a block of RB consecutive rays (4 ... 64,
// by the ray count: enough waves to fill the chip) whose samples are ONE contiguous stretch of the outputs:
//   1. W = 64 / RB lanes share a ray: lane (ray, w) loads the ray's count, offset, number of runs AND run record w in one round
//      trip (the records' addresses do not depend on the counts);
//   2. the block's runs become a SEGMENT LIST in the wave's LDS, ray-major — first sample (relative to the block), lattice start
//      t0, and, for runs that leave t0's binade, the index j1 of the first lattice point beyond the edge (one division per run, so
//      that the lanes below almost never need the general closed form).  Every run is cut into QUADS of four consecutive samples
//      counted from the run's start (the last one partial); quad positions come from a segmented DPP scan inside the ray's lanes,
//      a scan over the rays, and one LDS pass that adds a ray's base to its entries.  A ray with more than W runs takes more
//      rounds (a noise grid's 300 runs per ray: 19 rounds of 16 lanes instead of 300 dependent loads of one lane); blocks with
//      more runs than the list holds are taken in sub-blocks of whole rays;
//   3. the quads are dealt to the lanes 64 at a time: a lane finds its quad's run by bisection in LDS (at most 65 candidates),
//      takes its lattice point as bits(t) + j c — exact inside a binade (lattice.hpp; nfa_lattice_advance when the check fails),
//      three more points the same way, and writes 16-byte vectors (a partial quad: single elements).  A lane never straddles
//      runs, so there is no per-sample path, and consecutive lanes write consecutive addresses.
// Rays flagged by the count pass (run records did not fit) become one "skip" segment: the fallback launch writes them.
constexpr int kSegCap = 512;                // >= kMaxRunsCap: a single ray always fits a sub-block
template <int CTRL>
__device__ __forceinline__ int dpp_i32(int old, int v) { return __builtin_amdgcn_update_dpp(old, v, CTRL, 0xf, 0xf, false); }
__device__ __forceinline__ int wave_incl_scan_i32(int v) {
    const int lane = lane_id();
    v += dpp_i32<kDppRowShr + 1>(0, v);
    v += dpp_i32<kDppRowShr + 2>(0, v);
    v += dpp_i32<kDppRowShr + 4>(0, v);
    v += dpp_i32<kDppRowShr + 8>(0, v);
    int u = dpp_i32<kDppRowBcast15>(0, v); if (lane & 16) v += u;
    u = dpp_i32<kDppRowBcast31>(0, v); if (lane & 32) v += u;
    return v;
}
__device__ __forceinline__ int readlane_dyn_i32(int v, int src) { return __builtin_amdgcn_readlane(v, src); }
__device__ __forceinline__ int64_t readlane_dyn_i64(int64_t v, int src) {
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(uint64_t)v, src);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)((uint64_t)v >> 32), src);
    return (int64_t)(((uint64_t)hi << 32) | lo);
}
// the constant step of cone_angle == 0 as the lattice's integer increment (lattice.hpp)
struct LatStep {
    float dt;
    uint32_t D;       // dt's mantissa with the hidden bit
    int ed;           // dt's biased exponent
    bool ok;          // dt is a positive normal number
    __device__ __forceinline__ explicit LatStep(float d) : dt(d) {
        const uint32_t db = nfa_f2u(d);
        ed = (int)((db >> 23) & 0xffu);
        D = (db & 0x7fffffu) | 0x800000u;
        ok = ed >= 1 && ed < 255 && (db >> 31) == 0;
    }
    // ulps per step in t's binade (0: no closed form here — a tie, t below dt's binade, an irregular value)
    __device__ __forceinline__ uint32_t inc_at(uint32_t tb) const {
        const int e = (int)((tb >> 23) & 0xffu), sh = e - ed;
        const bool reg = ok && (tb >> 31) == 0 && e >= 1 && e < 254 && sh >= 1 && sh <= 24;
        const uint32_t shc = reg ? (uint32_t)sh : 1u;
        const uint32_t c0 = D >> shc, frac = D & ((1u << shc) - 1u), half = (1u << shc) >> 1;
        return (reg && frac != half) ? c0 + (frac > half ? 1u : 0u) : 0u;
    }
};
constexpr int kEmitSpeculate = 4;           // run records of a ray requested together with its counts
constexpr int kEmitSegWords = 5;            // qpos, spos, t0, j1, ray (+ 2 words of edge offset with interval outputs)
constexpr int kEmitRayBaseBytes = 64 * 4 + 64;   // per wave, behind the segment list: first quad of every ray of the block, then 64 flag bytes
                                                  // (which quad slots of the current chunk open a run: emit_by_tiles, round 5)
#ifndef NFA_EMIT_BISECT
#define NFA_EMIT_BISECT 1             // 1: per-quad bisection over the segment list; 0: the run of every quad slot from a ballot of "a run opens
#endif                                // here" flags (round 5: two LDS round trips per chunk instead of seven per quad — measured NO faster: 207-209 us
                                      // both ways at 10^6 rays, 12.9 vs 12.6 us at 6.5 k; the searches are not what the kernel waits for)
constexpr int kSegSkip = 64;                // seg_ray flag: the run belongs to a ray the fallback launch writes

// inclusive scan over the lanes of a wave, restarting at every segment head; `dist` = lanes between this lane and its segment's head
__device__ __forceinline__ int wave_seg_incl_scan_i32(int v, int dist) {
    const int lane = lane_id();
    const int rl = lane & 15;
    const int dr = dist < rl ? dist : rl;
    int u;
    u = dpp_i32<kDppRowShr + 1>(0, v); if (dr >= 1) v += u;
    u = dpp_i32<kDppRowShr + 2>(0, v); if (dr >= 2) v += u;
    u = dpp_i32<kDppRowShr + 4>(0, v); if (dr >= 4) v += u;
    u = dpp_i32<kDppRowShr + 8>(0, v); if (dr >= 8) v += u;
    u = dpp_i32<kDppRowBcast15>(0, v); if ((lane & 16) && dist > rl) v += u;
    u = dpp_i32<kDppRowBcast31>(0, v); if ((lane & 32) && dist > (lane & 31)) v += u;
    return v;
}

// Rays per wave, decided ON THE DEVICE (round 5).  The host's choice goes by the ray count alone (enough waves for
// every SIMD); what a wave's block costs, though, is its RUNS — every run is a segment to list, scan and bisect — and a grid of
// alternating voxels has 30-60 runs per ray where an object has 2-5: at 65 k rays of the 256^3 noise grid 16 rays per wave took 419 us
// and 2 rays per wave 194 (profiles/r04_emit.md).  The call's runs = edges - samples are in the workspace when this kernel starts
// (the offsets kernel's totals), so the block is halved until it holds at most kEmitRunsPerBlock runs on average — never below
// `rb_min` (the grid was sized for that many blocks).
constexpr int64_t kEmitRunsPerBlock = 288;
__device__ __forceinline__ int emit_rays_per_wave_log2(int rb_log2, int rb_min, int64_t n_rays, const int64_t *__restrict__ n_dev) {
    if (rb_min >= rb_log2) return rb_log2;
    const int64_t runs = n_dev[0] - n_dev[1];
    int rb = rb_log2;
    while (rb > rb_min && (runs << rb) > kEmitRunsPerBlock * n_rays) --rb;
    return rb;
}

// LDS bytes of one wave's segment list (+ its ray bases and run flags)
__host__ __device__ constexpr int emit_lds_per_wave(int seg_cap, bool iv) { return seg_cap * 4 * (kEmitSegWords + (iv ? 2 : 0)) + kEmitRayBaseBytes; }

// How a wave of the emit pass learns where its rays' samples go.  EmitPlain: the offsets kernel has run, sm_starts / sm_cnts are in
// memory.  A hook with kDeferred = true (sample_fused.hpp: the single-launch sampling call) hands over the counts it holds in
// registers and starts RELATIVE to the wave's first sample (`counts`), and `wait()` — called exactly once per wave, after the first
// sub-block's segment list has been built and before anything is stored — returns what to add to them (or -1: store nothing): the
// list is built while the offset of the wave's first sample is still being worked out.
struct EmitPlain {
    static constexpr bool kDeferred = false;
    __device__ __forceinline__ void counts(int64_t, bool, int64_t &, int64_t &) const {}
    __device__ __forceinline__ int64_t wait() const { return 0; }
};

// The wave expands ray blocks blk_first, blk_first + blk_step, ... (< n_rb) of 2^rb_log2 rays each.  `lds_wave`: this wave's
// emit_lds_per_wave bytes.  `n_total`: what has to fit into `capacity` for anything to be stored (a speculative launch: the call's
// total, requested by the caller and looked at here after the first block's loads have been issued; 0 = no check).
template <bool IV, class Hook = EmitPlain>
__device__ __forceinline__ void emit_by_tiles(const nfa_traverse_args &a, const RunStore &rs, int rb_log2, int seg_cap, unsigned char *lds_wave,
                                              int64_t n_total, int64_t capacity, int64_t blk_first, int64_t blk_step, int64_t blk_end = -1,
                                              Hook hook = Hook())
{
    bool checked = false;
    bool waited = !Hook::kDeferred;
    int64_t gbase = 0;                                            // (deferred: added to every sample position at the store; -1: nothing is stored)
    const int lane = lane_id();
    const int64_t R = a.n_rays;
    const LatStep L(march_dt(0.0f, 0.0f, a.step_size));
    const float dt = L.dt;
    int32_t *seg_qpos = (int32_t *)lds_wave;                      // first quad of the run (relative to the sub-block)
    int32_t *seg_spos = seg_qpos + seg_cap;                       // first sample of the run (relative to the sub-block)
    float *seg_t0 = (float *)(seg_spos + seg_cap);                // lattice point of the run's first sample
    int32_t *seg_j1 = (int32_t *)(seg_t0 + seg_cap);              // index of the first lattice point past t0's binade (INT_MAX: none in the run)
    int32_t *seg_ray = seg_j1 + seg_cap;                          // the run's ray (its index in the block; + kSegSkip: not written here)
    int32_t *ray_base = seg_ray + seg_cap;                        // [64] first quad of every ray
    unsigned char *run_flag = (unsigned char *)(ray_base + 64);   // [64] quad slot of the current chunk opens a run (zero between chunks)
    int64_t *seg_eoff = (int64_t *)(ray_base + 64 + 16);          // (IV only) edge index of a sample = its sample index + this
    run_flag[lane] = 0;                                           // (LDS comes uninitialised; every reader clears its flag again)
    // W = 64 / RB lanes share a ray while the segment list is built: lane (ray_l, w) takes the ray's runs w, w + W, ...
    const int RB = 1 << rb_log2, wshift = 6 - rb_log2, W = 1 << wshift;
    const int ray_l = lane >> wshift, w = lane & (W - 1);
    const bool head = w == 0;
    int64_t n_rb = (R + RB - 1) >> rb_log2;
    if (Hook::kDeferred && blk_end >= 0 && blk_end < n_rb) n_rb = blk_end;      // (only the single-launch call bounds the blocks of a wave)
    for (int64_t blk = blk_first; blk < n_rb; blk += blk_step) {
        const int64_t r0 = blk << rb_log2, r = r0 + ray_l;
        const bool own = r < R;
        int64_t S = 0, cnt = 0, E = 0;
        int nr = 0;
        float t0_q = 0.0f, t0_q2 = 0.0f;
        int32_t first_q = 0, next_q = 0, first_q2 = 0, next_q2 = 0;
        if (own) {
            // ONE round trip: the ray's counts and its first kEmitSpeculate run records (their addresses do not depend on the
            // counts; almost every ray of a NeRF-like scene has fewer runs — requesting all 2 W of the first two rounds read 5 MB
            // of never-written record slots per call at 6.5 k rays, counters of profiles/r04_pmc_traverse.json's first version)
            if (Hook::kDeferred) hook.counts(r, own, cnt, S);
            else { cnt = a.sm_cnts[r]; S = a.sm_starts[r]; }
            nr = rs.n_runs[r];
            if (IV) E = a.iv_starts[r];
            const int lim = rs.max_runs < kEmitSpeculate ? rs.max_runs : kEmitSpeculate;
            if (w < lim) { t0_q = rs.t0[(int64_t)w * R + r]; if (w > 0) first_q = rs.first[(int64_t)w * R + r]; }
            if (w + 1 < rs.max_runs && w < lim) next_q = rs.first[(int64_t)(w + 1) * R + r];
            if (W + w < lim) { t0_q2 = rs.t0[(int64_t)(W + w) * R + r]; first_q2 = rs.first[(int64_t)(W + w) * R + r]; }
            if (W + w + 1 < rs.max_runs && W + w < lim) next_q2 = rs.first[(int64_t)(W + w + 1) * R + r];
        }
        if (!checked) {
            if (n_total > capacity || n_total < 0) return;          // outputs too small: the caller launches again after its read-back
            checked = true;                                         // (a negative total cannot be: see below)
        }
        // Defence, a handful of instructions per RAY: counts, offsets or run counts that cannot be are not expanded.  They were seen
        // once in ~60 runs of eight processes sharing one GPU — every eighth workgroup of the count launch (one XCD's share) had left
        // none of its stores in memory, profiles/r06_oversubscription.md — and the samples of a ray placed by such an offset are wild
        // stores.  The call's totals are inconsistent in the same way; the host raises on them.
        if (!Hook::kDeferred && capacity > 0 && (S < 0 || S > capacity || cnt > capacity - S)) cnt = 0;
        if (nr != kRunsOverflow && nr > rs.max_runs) cnt = 0;
#ifdef NFA_EMIT_CHECK
        // (debug build: what the kernel was handed, validated before anything depends on it)
        {
            const bool bad_counts = own && (cnt < 0 || S < 0 || (n_total > 0 && S + cnt > n_total));
            const bool bad_runs = own && cnt > 0 && nr != kRunsOverflow && (nr < 1 || nr > rs.max_runs);
            const bool bad_rec = own && cnt > 0 && nr != kRunsOverflow && !bad_runs &&
                                 ((w > 0 && w < nr && w < kEmitSpeculate && (first_q <= 0 || first_q >= cnt)) ||
                                  (w + 1 < nr && w < kEmitSpeculate && (next_q <= first_q || next_q >= cnt)));
            if (bad_counts || bad_runs || bad_rec)
                printf("EMIT CHECK: block %d wave %d lane %d ray %lld of %lld: cnt %lld S %lld nr %d max_runs %d first %d next %d n_total %lld capacity %lld rb %d seg_cap %d (%d%d%d)\n",
                       (int)blockIdx.x, (int)(threadIdx.x >> 6), lane, (long long)r, (long long)R, (long long)cnt, (long long)S, nr, rs.max_runs, first_q, next_q,
                       (long long)n_total, (long long)capacity, rb_log2, seg_cap, (int)bad_counts, (int)bad_runs, (int)bad_rec);
            if (__ballot(bad_counts || bad_runs || bad_rec)) return;
        }
#endif
        if (cnt < 0) cnt = 0;
        const bool skip = nr == kRunsOverflow;
        const int nseg = cnt > 0 ? (skip ? 1 : nr) : 0;
        const int P = wave_incl_scan_i32(head ? nseg : 0);          // inclusive prefix of the rays' segment counts (the same in a ray's W lanes)
        if (readlane_dyn_i32(P, 63) == 0) continue;                 // a block of rays without samples
        // rays beyond the last one inherit its end (monotone ends for the sub-block search below)
        const int n_own = (int)((R - r0) < RB ? (R - r0) : RB);
        int64_t End = S + cnt;
        {
            const int64_t last_end = readlane_dyn_i64(End, (n_own << wshift) - 1);
            if (ray_l >= n_own) { S = last_end; End = last_end; }
        }
        int a_ray = 0, seg_off = 0;
        while (a_ray < n_own) {
            const int64_t S_a = readlane_dyn_i64(S, a_ray << wshift);
            // the sub-block: rays [a_ray, b_ray) — as many whole rays as the segment list holds (and 2^31 samples at most)
            const bool fits = ray_l >= a_ray && ray_l < n_own && (P - seg_off) <= seg_cap && (End - S_a) < (1ll << 31);
            const unsigned long long fm = __ballot(fits) >> (a_ray << wshift);
            int n_take = (fm == ~0ull ? 64 : __ffsll((long long)~fm) - 1) >> wshift;     // leading run of ones, in rays
            if (n_take < 1) n_take = 1;                              // (one ray's runs always fit; one ray of >= 2^31 samples: below)
            const int b_ray = a_ray + n_take;
            const bool in_sub = ray_l >= a_ray && ray_l < b_ray;
            const int my_base = P - nseg - seg_off;
            const int n_sub = readlane_dyn_i32(P, (b_ray << wshift) - 1) - seg_off;
            const int64_t sub_end = readlane_dyn_i64(End, (b_ray << wshift) - 1);
            const bool too_long = sub_end - S_a >= (1ll << 31);      // (one ray of >= 2^31 samples: flagged by the count pass, the fallback launch writes it)
            // the segment list, quad positions counted from the RAY's first quad; a round takes W runs of every ray
            int carry = 0;                                           // quads of the ray's earlier rounds
            for (int round = 0;; ++round) {
                const int q = (round << wshift) + w;
                const bool valid = in_sub && q < nseg;
                if (__ballot(valid) == 0ull) break;
                // the records of this round were requested a round ago (the first kEmitSpeculate of a ray: with its counts); request
                // the next round's (only the rays of THIS sub-block rotate their registers: a later sub-block's rays still hold
                // their first records)
                if (in_sub) {
                    if (round >= 1) { t0_q = t0_q2; first_q = first_q2; next_q = next_q2; }
                    if (valid && q >= kEmitSpeculate && round == 0) {      // a run beyond the speculative ones: fetched now
                        t0_q = rs.t0[(int64_t)q * R + r];
                        first_q = rs.first[(int64_t)q * R + r];
                        next_q = q + 1 < nseg ? rs.first[(int64_t)(q + 1) * R + r] : 0;
                    }
                    const int qn = q + W;
                    if (qn < nseg && (round >= 1 || qn >= kEmitSpeculate)) {
                        t0_q2 = rs.t0[(int64_t)qn * R + r];
                        first_q2 = rs.first[(int64_t)qn * R + r];
                        next_q2 = qn + 1 < nseg ? rs.first[(int64_t)(qn + 1) * R + r] : 0;
                    }
                }
                const int32_t first = (skip || q == 0) ? 0 : first_q;
                const int32_t next = (skip || q + 1 >= nseg) ? (int32_t)cnt : next_q;
                const int len = valid ? next - first : 0;
                const int quads = (len + 3) >> 2;
                const int incl = wave_seg_incl_scan_i32(quads, w);
                if (valid) {
                    const int k = my_base + q;
                    seg_qpos[k] = carry + incl - quads;
                    seg_spos[k] = (int32_t)(S - S_a) + first;
                    seg_t0[k] = t0_q;
                    seg_ray[k] = ray_l + ((skip || too_long) ? kSegSkip : 0);
                    // index of the first lattice point past t0's binade: the steps that stay inside, plus the one that crosses
                    const uint32_t tb = nfa_f2u(t0_q);
                    const uint32_t inc = L.inc_at(tb);
                    int32_t j1 = 0x7fffffff;
                    if (inc != 0u) {
                        const uint32_t x = (1u << 24) - 1u - ((tb & 0x7fffffu) | 0x800000u);
                        uint32_t n_in = (uint32_t)((float)x / (float)inc);         // both < 2^24: exact operands, the correctly
                        n_in -= ((uint64_t)n_in * inc > x) ? 1u : 0u;              // rounded quotient truncates to floor or floor + 1
                        n_in += ((uint64_t)(n_in + 1u) * inc <= x) ? 1u : 0u;
                        if ((int)n_in + 1 < len) j1 = (int32_t)n_in + 1;
                    }
                    seg_j1[k] = j1;
                    if (IV) seg_eoff[k] = E - S + q + 1;
                }
                carry += __shfl(incl, lane | (W - 1), 64);           // the round's quads of this ray (its last lane holds the sum)
            }
            // the ray's first quad = exclusive prefix of the rays' quad counts; every entry then moves by its ray's base
            const int QP = wave_incl_scan_i32(head && in_sub ? carry : 0);
            const int n_quads = too_long ? 0 : readlane_dyn_i32(QP, 63);
            const int span = (int)(sub_end - S_a);
            if (head && in_sub) ray_base[ray_l] = QP - carry;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            for (int k = lane; k < n_sub; k += 64) seg_qpos[k] += ray_base[seg_ray[k] & 63];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            // One chunk of 64 quad slots in two halves: `quad_compute` — which run a slot lies in, its lattice points — needs nothing
            // but the segment list; `quad_store` needs the offset of the wave's first sample.  A deferred hook's wait sits between
            // the two halves of the FIRST TWO chunks (round 6: a training wave has ~40 quads; the look-back's last microseconds are
            // spent on LDS reads and lattice arithmetic instead of waiting), later chunks run back to back.
            struct QuadOut {
                bool live = false;
                int nv = 0, seg = 0, j = 0;
                float sv[5];
                int64_t s_rel = 0, rr = 0;      // first sample (without the deferred offset), ray
            };
            int seg_lo = 0;                                          // wave-uniform: the run of the chunk's first quad
            auto quad_compute = [&](int qc) {
                QuadOut q;
                const int slot = qc + lane;
                int seg = seg_lo;
#if !NFA_EMIT_BISECT
                // The run of every quad slot of this chunk WITHOUT a search (round 5): the runs that open inside the chunk are
                // seg_lo + 1, seg_lo + 2, ... (at most 64: a run has at least one quad; seg_lo is the run of the quad BEFORE the chunk, so a
                // run may open on the chunk's very first slot); lane i looks at run seg_lo + 1 + i and, when it opens at slot qc + st with
                // 0 <= st < 64, raises flag st; a ballot of the flags is the chunk's map of run starts and a slot's run is seg_lo + the
                // starts at or below it.  Two LDS round trips per chunk instead of the bisection's seven
                // dependent ones per quad.
                {
                    const int c = seg_lo + 1 + lane;
                    if (c < n_sub) {
                        const int st = seg_qpos[c] - qc;
                        if (st >= 0 && st < 64) run_flag[st] = 1;
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    const bool opens = run_flag[lane] != 0;
                    run_flag[lane] = 0;
                    const unsigned long long starts = __ballot(opens);
                    seg = seg_lo + __popcll(starts & lanes_le(lane));
                }
#endif
                if (slot < n_quads) {
#if NFA_EMIT_BISECT
                    int lo = seg_lo, hi = seg_lo + 65 < n_sub ? seg_lo + 65 : n_sub;
                    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (seg_qpos[mid] <= slot) lo = mid; else hi = mid; }
                    seg = lo;
#endif
                    const int ray_s = seg_ray[seg];
                    if (ray_s < kSegSkip) {
                        const int sp = seg_spos[seg];
                        const int j = 4 * (slot - seg_qpos[seg]);                   // the quad's first sample within its run
                        const int len = (seg + 1 < n_sub ? seg_spos[seg + 1] : span) - sp;
                        const int j1 = seg_j1[seg];
                        float tbase = seg_t0[seg];
                        uint32_t tb = nfa_f2u(tbase);
                        uint32_t inc = L.inc_at(tb);
                        int jj = j;
                        if (j >= j1) {                                              // past t0's binade: restart from the first point beyond
                            tbase = nfa_u2f(tb + (uint32_t)(j1 - 1) * inc) + dt;    // (j1 was formed with this inc: exact inside the binade, one real add)
                            tb = nfa_f2u(tbase);
                            inc = L.inc_at(tb);
                            jj = j - j1;
                        }
                        const uint64_t top = (uint64_t)((tb & 0x7fffffu) | 0x800000u) + (uint64_t)(uint32_t)(jj + 4) * inc;
                        if (inc != 0u && top < (1ull << 24)) {
                            const uint32_t b0 = tb + (uint32_t)jj * inc;
#pragma unroll
                            for (int k = 0; k < 5; ++k) q.sv[k] = nfa_u2f(b0 + (uint32_t)k * inc);
                        } else {                                                    // (a quad on a binade edge, a second edge in one run)
                            q.sv[0] = nfa_lattice_advance(tbase, dt, (int64_t)jj, nullptr);
#pragma unroll
                            for (int k = 0; k < 4; ++k) q.sv[k + 1] = q.sv[k] + dt;
                        }
                        q.live = true;
                        q.nv = len - j < 4 ? len - j : 4;                           // samples of this quad (>= 1)
                        q.seg = seg;
                        q.j = j;
                        q.s_rel = S_a + sp + j;
                        q.rr = r0 + ray_s;
                    }
                }
                seg_lo = readlane_dyn_i32(seg, 63);                  // (lane 63 idle = the last chunk)
                return q;
            };
            auto quad_store = [&](const QuadOut &q) {
                if (!q.live) return;
                const float (&sv)[5] = q.sv;
                const int nv = q.nv;
                const int64_t s = gbase + q.s_rel, rr = q.rr;
                if (!IV && nv == 4) {
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
                        const int64_t rv[4] = {rr, rr, rr, rr};
                        __builtin_memcpy(a.sm_ray_indices + s, rv, 32);
                    }
                    if (a.sm_is_valid) { const uint32_t ones = 0x01010101u; __builtin_memcpy(a.sm_is_valid + s, &ones, 4); }
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if (e >= nv) break;
                        const float ta = sv[e], tb2 = sv[e + 1];
                        if (a.sm_vals) a.sm_vals[s + e] = (tb2 + ta) * 0.5f;
                        if (a.sm_ray_indices) a.sm_ray_indices[s + e] = rr;
                        if (a.sm_is_valid) a.sm_is_valid[s + e] = 1;
                        if (a.t_starts) { a.t_starts[s + e] = ta; a.t_ends[s + e] = tb2; }
                        if (IV && a.iv_vals) {
                            // edge layout of a ray: every run contributes len + 1 edges (grid.cu:219-245)
                            const int64_t e_right = s + e + seg_eoff[q.seg];
                            a.iv_vals[e_right] = tb2; a.iv_ray_indices[e_right] = rr; a.iv_is_right[e_right] = 1;
                            a.iv_is_left[e_right - 1] = 1;
                            if (q.j + e == 0) { a.iv_vals[e_right - 1] = ta; a.iv_ray_indices[e_right - 1] = rr; }
                        }
                    }
                }
            };
            int qc = 0;
            if (Hook::kDeferred && !waited && n_quads > 0) {           // (wave-uniform) the first two chunks' arithmetic, then the wait, then their stores
                const QuadOut q0 = quad_compute(0);
                QuadOut q1;
                if (n_quads > 64) q1 = quad_compute(64);
                gbase = hook.wait();
                waited = true;
                if (gbase >= 0) { quad_store(q0); quad_store(q1); }
                qc = n_quads > 64 ? 128 : 64;
            }
            if (!waited) { gbase = hook.wait(); waited = true; }
            if (gbase >= 0) {
                for (; qc < n_quads; qc += 64) {
                    const QuadOut q = quad_compute(qc);
                    quad_store(q);
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();                          // the list is rewritten by the next sub-block
            seg_off = readlane_dyn_i32(P, (b_ray << wshift) - 1);
            a_ray = b_ray;
        }
    }
    if (!waited) (void)hook.wait();                                  // (a wave without a sample: the hook still runs once)
}

// the tile form's own kernel: the host launches it for every cone_angle == 0 call (`emit` option: auto or tiles)
#ifndef NFA_EMIT_MINBLOCKS
#define NFA_EMIT_MINBLOCKS 5          // workgroups the compiler has to fit per CU (x 4 waves = waves per SIMD): bounds the VGPR count
#endif
template <bool IV>
__global__ __launch_bounds__(kBlock, NFA_EMIT_MINBLOCKS) void traverse_emit_tiles_kernel(nfa_traverse_args a, RunStore rs, int64_t capacity,
                                                                     const int64_t *__restrict__ n_dev, int speculative, int rb_log2, int rb_min, int seg_cap)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char emit_lds[];
    // speculative launch: the true total is requested here and looked at after the first block's loads have been issued — nothing
    // is stored before the check
    const int64_t n_total = speculative ? n_dev[1] : 0;
    const int wib = (int)(threadIdx.x >> 6);
    emit_by_tiles<IV>(a, rs, emit_rays_per_wave_log2(rb_log2, rb_min, a.n_rays, n_dev), seg_cap, emit_lds + wib * emit_lds_per_wave(seg_cap, IV),
                      n_total, capacity, (int64_t)blockIdx.x * kWavesPerBlock + wib, (int64_t)gridDim.x * kWavesPerBlock);
}

// pass 2: ONE launch, the form chosen ON THE DEVICE from the totals the offsets kernel left in the workspace (the speculative
// launch runs before the host has seen them).  16 lanes per ray pay per RUN (~25 instructions + a pass per 64 samples), a lane per
// sample pays ~14 dependent loads per SAMPLE: the ray groups win on long runs and lose on a grid of alternating voxels (the
// reference's `rand > 0.5` test grid: 271 samples per ray in ~130 runs — 113 vs 77 us at 4 k rays when the choice looked at the
// sample count alone, profiles/r03_count_pass.md).  runs = edges - samples.  With a cone angle the sample-parallel form re-runs a
// sample's chain from its run's start, so the ray groups take over at much shorter runs.
//   hint: 0 = choose, 1 = ray groups, 2 = lane per sample (the `emit` option; tiles: traverse_emit_tiles_kernel, cone_angle == 0 only).
//   speculative: outputs hold `capacity` samples — a launch whose outputs are too small does nothing, the caller launches again
//   with the right size.
__global__ __launch_bounds__(kBlock) void traverse_emit_kernel(nfa_traverse_args a, RunStore rs, int64_t capacity,
                                                               const int64_t *__restrict__ n_dev, int speculative, int hint)
{
    const int64_t n_ed = n_dev[0], n_sm = n_dev[1];
    if (speculative && (n_sm > capacity || n_sm < 0)) return;        // (a negative total cannot be: emit_by_tiles has the story)
    const int64_t runs = n_ed - n_sm > 0 ? n_ed - n_sm : 1;
    const bool long_runs = a.cone_angle != 0.0f ? n_sm >= 8 * runs : (n_sm >= 900000 && n_sm >= 20 * runs);
    if (hint == 1 || (hint != 2 && long_runs)) emit_by_ray_groups(a, rs);
    else emit_by_samples(a, rs, speculative ? n_sm : capacity);
}
