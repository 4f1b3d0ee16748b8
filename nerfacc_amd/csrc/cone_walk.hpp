// cone_walk.hpp — the count pass for cone_angle != 0 (the reference's unbounded-scene setting,
// examples/train_ngp_nerf_occ.py:53), included by grid.hip inside namespace nfa::{anonymous}.
//
// Reference shape (grid.cu:95-281): one thread per ray; per voxel either `dt = calc_dt(t); while (t + dt/2 < t_exit) t += dt`
// (empty) or a sample loop that re-evaluates dt per sample (occupied).  dt = clamp(t * cone_angle, step, 1e10) depends on t, so
// a ray's samples form ONE chain through every voxel before them — that part is serial per ray.  The voxel walk itself (which
// voxels, their exit times, their occupancy) does not depend on the chain.  So:
//
//   phase 1  one lane per LEVEL SEGMENT of a ray (up to 2 G - 1 segments, grid.cu:129-150; P = 8 lanes per ray up to 4 levels,
//            16 up to 8, 1 for a single level): DDA + brick lookups, four voxels per trip with their brick words requested
//            together; every voxel leaves ONE 4-byte record — its exit time, occupancy in the sign bit — in a per-workgroup
//            scratch plane [voxel][lane] (coalesced; the plane is sized for the longest possible walk, rx + ry + rz voxels).
//   phase 2  the ray's first lane runs the chain over the records, segment after segment, as a FLAT loop: one iteration is
//            either "take the next record" or "one lattice step" (an add, plus a sample when the voxel is occupied).  The
//            rays of a wave are never in the same kind of voxel, and a loop nest (voxels outside, steps inside) pays the
//            slowest lane's trip count at every voxel; the flat loop pays max over the wave's rays of (voxels + steps) once.
//            Samples are recorded as runs (start, first index) exactly like the other count passes, so offsets and the
//            sample-parallel emit pass are shared.
//
// Rays whose records cannot be encoded (a negative voxel exit time: the sign bit is taken) take the serial general walk by
// their first lane, inside this kernel.  traverse_steps_limit > 0 (the test-time marcher: a handful of samples per round, so
// walking whole segments ahead of the chain would be wasted) and rays_mask keep the general kernel.
#pragma once

struct VoxelStore {
    uint32_t *rec;   // [count workgroups][cap + kSlack][kBlock]: record v of the lane tid of workgroup b at ((b * (cap + kSlack) + v) * kBlock + tid)
    int cap;         // rx + ry + rz: a DDA walk changes one index by one per step and never comes back
    static constexpr int kSlack = 8;
};

template <bool LDS_OCC, int P>
__global__ __launch_bounds__(kBlock) void traverse_count_cone_kernel(nfa_traverse_args a, GridView gv,
                                                                     int64_t *__restrict__ block_sums, RunStore rs, VoxelStore vs)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const Occ<LDS_OCC> occ = stage_occupancy<LDS_OCC>(gv, smem);
    int32_t *seg_n = (int32_t *)(smem + occ.bytes);      // [kBlock] records of every lane's segment; -1: the lane has no segment
    float *seg_start = (float *)(seg_n + kBlock);        // [kBlock] the segment's clipped start (grid.cu:148)
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
    const float step_size = a.step_size, cone = a.cone_angle;

    Events<EV_MANY> ev;
    ev.init(a, rr, o, inv);
    int level = 0;
    float seg_lo = 0.f, seg_hi = 0.f;
    const bool live = ray_ok && part + 1 < 2 * G && segment_of(ev, part, G, near, far, level, seg_lo, seg_hi);

    // ---- phase 1: the segment's voxels -> records
    uint32_t *const my = vs.rec + (int64_t)blockIdx.x * (vs.cap + VoxelStore::kSlack) * kBlock + tid;
    int n = 0;
    bool bad = false;
    if (live) {
        Dda s;
        dda_setup(s, o, d, inv, seg_lo, seg_hi, a.aabbs + 6 * level, gv.res);
        const uint32_t *lc = (const uint32_t *)occ.smem;
        constexpr int B = 4;
        for (bool more = true; more;) {
            bool valid[B];
            float tc[B];
            int id[B], bp[B];
#pragma unroll
            for (int k = 0; k < B; ++k) {
                valid[k] = more;
                tc[k] = fminf(fminf(s.tx, fminf(s.ty, s.tz)), seg_hi);
                id[k] = (int)__umul24(__umul24(s.cx >> 2, gv.nby) + (s.cy >> 2), gv.nbz) + (s.cz >> 2) + level * gv.bricks_per_grid;
                bp[k] = ((s.cx & 3) << 4) | ((s.cy & 3) << 2) | (s.cz & 3);
                if (more) more = dda_advance(s);
            }
            uint64_t bits[B];
            if (LDS_OCC) {
                uint2 wr[B];
#pragma unroll
                for (int k = 0; k < B; ++k) wr[k] = valid[k] ? ((const uint2 *)occ.smem)[id[k] >> 5] : make_uint2(0u, 0u);
#pragma unroll
                for (int k = 0; k < B; ++k) {
                    const uint32_t bit = 1u << (id[k] & 31);
                    bits[k] = (wr[k].x & bit) ? ((const uint64_t *)(lc + 2 * occ.w4))[(int)wr[k].y + __popc(wr[k].x & (bit - 1u))] : 0ull;
                }
            } else if (occ.bytes > 0) {
                uint32_t w[B];
#pragma unroll
                for (int k = 0; k < B; ++k) w[k] = valid[k] ? lc[id[k] >> 5] : 0u;
#pragma unroll
                for (int k = 0; k < B; ++k) bits[k] = (w[k] & (1u << (id[k] & 31))) ? gv.bricks[id[k]] : 0ull;
            } else {
#pragma unroll
                for (int k = 0; k < B; ++k) bits[k] = valid[k] ? gv.bricks[id[k]] : 0ull;
            }
            // the batch's records: stored unconditionally (valid[] is a prefix; what lies behind a segment's last record is never
            // read — the plane has B rows of slack), so that the four stores do not each sit in their own exec-masked region
#pragma unroll
            for (int k = 0; k < B; ++k) {
                const uint32_t tb = __float_as_uint(tc[k]);
                bad = bad || (valid[k] && (tb >> 31) != 0u);
                my[(int64_t)(n + k) * kBlock] = tb | ((uint32_t)((bits[k] >> bp[k]) & 1ull) << 31);
            }
#pragma unroll
            for (int k = 0; k < B; ++k) n += valid[k] ? 1 : 0;
            if (n > vs.cap) { bad = true; more = false; }        // cannot happen (cap = rx + ry + rz); keeps a broken walk inside its plane
        }
    }
    seg_n[tid] = live ? n : -1;
    seg_start[tid] = seg_lo;
    bad = group_bits<P>(__ballot(bad), group_base) != 0u;
    const unsigned live_parts = group_bits<P>(__ballot(live), group_base);
    __threadfence_block();
    __syncthreads();

    // ---- phase 2: the chain, one lane per ray
    int64_t out_iv = 0, out_sm = 0, out_ovf = 0;
    if (ray_ok && part == 0) {
        if (!bad) {
            // segments in order; per segment its records in order; per record ONE tight loop of lattice steps (an add, for an
            // occupied voxel a sample and dt re-evaluated).  Everything that happens once per voxel or per run — the record
            // fetch (requested one voxel ahead), dt at the voxel's entry, opening a run — stays outside the step loop: at one or
            // two waves per SIMD a wave's time is its dependent instruction count, and the steps outnumber the voxels 4 : 1.
            float t = near;
            bool cont = false, dead = false;
            int64_t n_sm = 0;
            int n_runs = 0;
            const bool store_runs = rs.t0 != nullptr;
            const bool fast_chain = near >= 0.0f && cone >= 0.0f && step_size >= 1.0e-30f;     // t >= near; halves of dt are exact
            unsigned rem = live_parts;
#if defined(NFA_CONE_DBG) && NFA_CONE_DBG == 1
            rem = 0u;
#endif
            while (rem != 0u && !dead) {
                const int sg = __ffs((int)rem) - 1;
                rem &= rem - 1u;
                const int nv = seg_n[tid + sg];
                const uint32_t *p = my + sg;
                uint32_t nxt = p[0];
                if (!cont) {                                        // march to the segment's start (grid.cu:153-163)
                    const float lo = seg_start[tid + sg];
                    const float dt = march_dt(t, cone, step_size), h = dt * 0.5f;
                    while (t + h < lo) {
                        const float nt = t + dt;
                        if (nt == t) { t = lo; break; }             // stuck lattice: as the oracle's lattice_skip
                        t = nt;
                    }
                }
                for (int v = 0; v < nv && !dead; ++v) {
                    const uint32_t rec = nxt;
                    p += kBlock;
                    nxt = *p;                                       // (the row behind a segment's last record exists: slack rows)
                    const bool oc = (rec >> 31) != 0u;
                    const float e = __uint_as_float(rec & 0x7fffffffu);
                    float dt = march_dt(t, cone, step_size), h = dt * 0.5f;
                    bool go = t + h < e;
                    if (oc && go && !cont) {                        // a run of samples starts here
                        if (store_runs && n_runs < rs.max_runs) {
                            rs.t0[(int64_t)n_runs * R + r] = t;
                            rs.first[(int64_t)n_runs * R + r] = (int32_t)n_sm;
                        }
                        ++n_runs;
                    }
                    cont = oc ? (cont || go) : false;               // grid.cu:205, 256
                    if (go) {
                        // The step loop, written for its dependent instruction count (the steps are what this phase's time
                        // is made of).  (1) No stuck test: the lattice can only get stuck (t + dt == t) when dt is at most half
                        // an ulp of t; t stays below e in this voxel and dt does not shrink (cone >= 0), so if HALF of dt moves
                        // e (dt >= ulp(e)), dt moves every t in [0, e].  (2) No upper clamp: t < e, so t * cone <= e * cone
                        // < 1e10 when checked once.  (3) No select between "dt re-evaluated" (occupied) and "dt as at the
                        // voxel's entry" (empty): dt = max(t * cA, floor) with cA = cone / 0 and floor = step / dt.  (4) the
                        // test t + dt/2 < e as ONE fma: dt * 0.5 is exact, so fma(dt, 0.5, t) rounds the same sum once.
                        // add, mul, max, fma, compare: 8 instructions per step with the loop's own three (13 before, 20 with the
                        // stuck test).
                        if (fast_chain && e + h != e && e * cone < 1.0e10f) {
                            const float cA = oc ? cone : 0.0f, fl = oc ? step_size : dt;
                            int k = 0;
                            do {
                                t = t + dt;
                                dt = fmaxf(t * cA, fl);
                                ++k;
                            } while (fmaf(dt, 0.5f, t) < e);
                            n_sm += oc ? k : 0;
                        } else {
                            while (go) {
                                const float nt = t + dt;
                                const bool stuck = nt == t;
                                n_sm += oc ? 1 : 0;
                                t = (stuck && !oc) ? e : nt;                // stuck lattice in an empty voxel: as the oracle's lattice_skip
                                dead = stuck && oc;                         // (the reference would spin here forever)
                                dt = oc ? march_dt(nt, cone, step_size) : dt;
                                go = !stuck && t + dt * 0.5f < e;
                            }
                        }
                    }
                }
            }
            const int64_t n_iv = n_sm + n_runs;                     // every run has one edge more than samples (grid.cu:219-245)
            const bool ovf = n_sm > 0 && rs.t0 && (n_runs > rs.max_runs || n_sm > 0x7fffffffll);
            if (rs.n_runs) rs.n_runs[r] = (uint16_t)(ovf ? kRunsOverflow : n_runs);
            out_iv = n_iv;
            out_sm = n_sm;
            out_ovf = ovf ? 1 : 0;
            if (a.terminate_planes) a.terminate_planes[r] = t;
        } else {
            CountSink sink{rs, r, R};
            float t_term = 0.f;
            traverse_ray_general<CountSink, EV_MANY, LDS_OCC>(a, gv, occ, r, sink, t_term);
            out_ovf = sink.finish(true) ? 1 : 0;
            out_iv = sink.n_iv;
            out_sm = sink.n_sm;
            if (a.terminate_planes) a.terminate_planes[r] = t_term;
        }
        if (a.iv_cnts) a.iv_cnts[r] = out_iv;
        a.sm_cnts[r] = out_sm;
    }
    publish_wave_sums(out_iv, out_sm, out_ovf, block_sums);
}
