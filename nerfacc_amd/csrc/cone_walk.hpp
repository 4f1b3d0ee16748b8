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
// Small launches (<= 4096 rays) have lanes to spare — they run 32 or 64 lanes per ray so that the chain phase has few rays per
// wave — and spend them on the walk: a level segment is cut into K = 4 or 8 PARTS at crossings of its major axis, as the
// single-level count pass cuts a ray.  Three lanes of the segment first write out the plane-crossing times of its x / y / z chains
// (plain sequential adds, exact by construction; n + 1 values per axis, in scratch: L2), then every part finds its start state
// with reads and two binary searches: the seam is the major axis' crossing j_begin, the minor axes' states are the counts of
// their crossings that precede it (ties z, y, x as the reference's if-chain picks them) and the first one that does not.
// Measured end to end (4 x 128^3): 1 024 rays 246 -> 192 us, 2 048 rays 239 -> 185, 4 096 rays 276 -> 231 (profiles/r03_cone.md).
//
// Rays whose records cannot be encoded (a negative voxel exit time: the sign bit is taken) take the serial general walk by
// their first lane, inside this kernel.  traverse_steps_limit > 0 (the test-time marcher: a handful of samples per round, so
// walking whole segments ahead of the chain would be wasted) and rays_mask keep the general kernel.
#pragma once

struct VoxelStore {
    uint32_t *rec;   // [count workgroups][cap + kSlack][kBlock]: record v of the lane tid of workgroup b at ((b * (cap + kSlack) + v) * kBlock + tid)
    int cap;         // rx + ry + rz: a DDA walk changes one index by one per step and never comes back
    static constexpr int kSlack = 8;
    float *xt;       // [count workgroups][kBlock / K segment slots][rx + ry + rz + 3] plane-crossing times (P >= 32 only, else NULL)
};

template <bool LDS_OCC, int P>
__global__ __launch_bounds__(kBlock) void traverse_count_cone_kernel(nfa_traverse_args a, GridView gv,
                                                                     int64_t *__restrict__ block_sums, RunStore rs, VoxelStore vs)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    NFA_PHASE_BEGIN();
    const Occ<LDS_OCC> occ = stage_occupancy<LDS_OCC>(gv, smem);
    NFA_PHASE_MARK(0);
    int32_t *seg_n = (int32_t *)(smem + occ.bytes);      // [kBlock] records of every lane's segment; -1: the lane has no segment
    float *seg_start = (float *)(seg_n + kBlock);        // [kBlock] the segment's clipped start (grid.cu:148)
    float *seg_end = seg_start + kBlock;                 // [kBlock] and end: no voxel of the segment exits later
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
    // lanes per segment slot: the 2 G - 1 segments sit in 8 (16) slots; with 32 / 64 lanes per ray a slot has K = 4 / 8 lanes
    const int slots = 2 * G - 1 <= 8 ? 8 : 16;
    const int K = (vs.xt && P / slots >= 4) ? P / slots : 1;
    const int slot = K > 1 ? part / K : part, sub = K > 1 ? part % K : 0;
    int level = 0;
    float seg_lo = 0.f, seg_hi = 0.f;
    const bool live = ray_ok && slot + 1 < 2 * G && segment_of(ev, slot, G, near, far, level, seg_lo, seg_hi);

    NFA_PHASE_MARK(1);
    // ---- phase 1: the segment's (the part's) voxels -> records
    uint32_t *const my = vs.rec + (int64_t)blockIdx.x * (vs.cap + VoxelStore::kSlack) * kBlock + tid;
    int n = 0;
    bool bad = false;
    Dda s;
    s.tx = s.ty = s.tz = 0.f; s.dx = s.dy = s.dz = 0.f;
    s.sx = s.sy = s.sz = 0; s.cx = s.cy = s.cz = 0; s.ox = s.oy = s.oz = 0;
    if (live) dda_setup(s, o, d, inv, seg_lo, seg_hi, a.aabbs + 6 * level, gv.res);
    bool part_live = live;
    int major_done = 0, j_end = 0x7fffffff, m_rank = 0;
    if (K > 1) {                                           // wave-uniform
        // crossings until each axis reaches its overflow index (an axis the ray does not move along has ONE pseudo-crossing,
        // at the segment's end: taking it ends the walk, utils_grid.cuh:95-112)
        const int nx = s.sx ? (s.ox - s.cx) * s.sx : 1, ny = s.sy ? (s.oy - s.cy) * s.sy : 1, nz = s.sz ? (s.oz - s.cz) * s.sz : 1;
        const bool regular = live && nx > 0 && ny > 0 && nz > 0 && nx <= gv.res[0] && ny <= gv.res[1] && nz <= gv.res[2];
        const int oy_ = gv.res[0] + 1, oz_ = gv.res[0] + gv.res[1] + 2;
        float *const A = vs.xt + ((int64_t)blockIdx.x * (kBlock / K) + tid / K) * (gv.res[0] + gv.res[1] + gv.res[2] + 3);
        if (regular && sub < 3) {                          // the chain of axis `sub`: entry i = time of its crossing i
            float t = sub == 0 ? s.tx : (sub == 1 ? s.ty : s.tz);
            const float dd = sub == 0 ? s.dx : (sub == 1 ? s.dy : s.dz);
            const int na = sub == 0 ? nx : (sub == 1 ? ny : nz);
            float *dst = A + (sub == 0 ? 0 : (sub == 1 ? oy_ : oz_));
            for (int i = 0; i <= na; ++i) { dst[i] = t; t = t + dd; }
        }
        __threadfence_block();
        __builtin_amdgcn_wave_barrier();                   // (the K lanes of a segment are lanes of one wave)
        if (regular) {
            const float Tx = A[nx - 1], Ty = A[oy_ + ny - 1], Tz = A[oz_ + nz - 1];       // the walk ends with the earliest of these
            int end_rank = 2; float T_end = Tx;                                       // ranks: z 0, y 1, x 2
            if (crossing_precedes(Ty, 1, T_end, end_rank)) { T_end = Ty; end_rank = 1; }
            if (crossing_precedes(Tz, 0, T_end, end_rank)) { T_end = Tz; end_rank = 0; }
            m_rank = (nx >= ny && nx >= nz) ? 2 : (ny >= nz ? 1 : 0);
            const int n_major = m_rank == 2 ? nx : (m_rank == 1 ? ny : nz);
            const int j_begin = (int)(((int64_t)sub * n_major) / K);
            j_end = (sub == K - 1) ? 0x7fffffff : (int)(((int64_t)(sub + 1) * n_major) / K);
            major_done = j_begin;
            part_live = j_begin < j_end;
            if (part_live && j_begin > 0) {
                const float T_seam = A[(m_rank == 2 ? 0 : (m_rank == 1 ? oy_ : oz_)) + j_begin - 1];   // time of major crossing j_begin
                if (m_rank != end_rank && !crossing_precedes(T_seam, m_rank, T_end, end_rank)) part_live = false;   // the walk ends before this seam
                else {
                    // crossings of the two minor axes that precede the seam (lower bounds, both searches in one loop of 8 rounds:
                    // <= 129 entries); the pending crossing of each is the entry found
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
                    if (!xm) { s.cx += lo1 * s.sx; s.tx = pend1; }
                    if (xm) { s.cy += lo1 * s.sy; s.ty = pend1; }
                    if (zm) { s.cy += lo2 * s.sy; s.ty = pend2; }
                    if (!zm) { s.cz += lo2 * s.sz; s.tz = pend2; }
                    if (m_rank == 2) { s.cx += j_begin * s.sx; s.tx = T_seam + s.dx; }
                    else if (m_rank == 1) { s.cy += j_begin * s.sy; s.ty = T_seam + s.dy; }
                    else { s.cz += j_begin * s.sz; s.tz = T_seam + s.dz; }
                }
            }
        } else {
            part_live = live && sub == 0;                  // odd index bookkeeping: the segment's first lane walks all of it
        }
    }
    NFA_PHASE_MARK(2);
    if (part_live) {
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
                if (more) {
                    const int cm_before = m_rank == 2 ? s.cx : (m_rank == 1 ? s.cy : s.cz);
                    more = dda_advance(s);
                    const int cm_after = m_rank == 2 ? s.cx : (m_rank == 1 ? s.cy : s.cz);
                    major_done += (cm_after != cm_before) ? 1 : 0;
                    more = more && major_done < j_end;     // the next part's seam
                }
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
    NFA_PHASE_MARK(3);
    seg_n[tid] = n;                                        // 0: no segment in this lane's slot, or an empty / dead part
    seg_start[tid] = seg_lo;
    seg_end[tid] = seg_hi;
    const unsigned long long group_mask = P >= 64 ? ~0ull : ((1ull << (P & 63)) - 1ull);
    bad = ((__ballot(bad) >> group_base) & group_mask) != 0ull;
    const unsigned long long live_lists = (__ballot(n > 0) >> group_base) & group_mask;     // bit l: lane l of the ray has records
    __threadfence_block();
    __syncthreads();
    NFA_PHASE_MARK(4);

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
            unsigned long long rem = live_lists;
            int last_slot = -1;
            while (rem != 0ull && !dead) {
                const int sg = __ffsll((long long)rem) - 1;          // the next lane with records: parts of a segment in order, segments in order
                rem &= rem - 1ull;
                const int sg_slot = K > 1 ? sg / K : sg;
                const bool seg_begins = sg_slot != last_slot;         // the first part with records of a segment
                last_slot = sg_slot;
                const int nv = seg_n[tid + sg];
                const uint32_t *p = my + sg;
                uint32_t nxt = p[0];
                // The step loop below has no stuck test and no upper clamp.  Both are decided once per SEGMENT from its end
                // time (no voxel of it exits later, t stays below the voxel's exit inside the loop, dt >= step and — cone >= 0 —
                // does not shrink):  a quarter step moves hi  =>  step / 2 >= ulp(hi) >= ulp(t)  =>  t + dt != t;
                // hi * cone < 1e10  =>  t * cone < 1e10.  An infinite end (a ray in a bounding plane, far = inf) fails both.
                const float hi_s = seg_end[tid + sg];
                const bool fast_seg = fast_chain && hi_s + 0.25f * step_size != hi_s && hi_s * cone < 1.0e10f;
                int k_oc = 0;                                       // samples of this segment (32-bit inside, added once)
                if (seg_begins && !cont) {                          // march to the segment's start (grid.cu:153-163)
                    const float lo = seg_start[tid + sg];
                    const float dt = march_dt(t, cone, step_size), h = dt * 0.5f;
                    while (t + h < lo) {
                        const float nt = t + dt;
                        if (nt == t) { t = lo; break; }             // stuck lattice: as the oracle's lattice_skip
                        t = nt;
#ifdef NFA_PHASE_CYCLES
                        ph_[13] += 1ull;
#endif
                    }
                }
#ifdef NFA_PHASE_CYCLES
                ph_[10] += (unsigned long long)nv; ph_[12] += 1ull;
#endif
                // The list's voxels, in two versions (round 6): the FAST one — fast_seg holds for the whole list, decided above — has no
                // stuck lattice, hence no `dead` ray, no careful step loop and none of the mask bookkeeping the compiler keeps for
                // them between two records (tools/cone_phases.py: the code between two records was ~650 of the chain's cycles
                // per voxel, the steps ~130); the careful one is the loop as it was.
                auto voxels = [&](auto fast) {
                    constexpr bool kFast = decltype(fast)::value;
                    for (int v = 0; v < nv && (kFast || !dead); ++v) {
                        const uint32_t rec = nxt;
                        p += kBlock;
                        nxt = *p;                                   // (the row behind a segment's last record exists: slack rows)
                        const bool oc = (rec >> 31) != 0u;
                        const float e = __uint_as_float(rec & 0x7fffffffu);
                        float dt = march_dt(t, cone, step_size), h = dt * 0.5f;
                        bool go = t + h < e;
                        if (oc && go && !cont) {                    // a run of samples starts here
                            if (store_runs && n_runs < rs.max_runs) {
                                rs.t0[(int64_t)n_runs * R + r] = t;
                                rs.first[(int64_t)n_runs * R + r] = (int32_t)(n_sm + k_oc);
                            }
                            ++n_runs;
                        }
                        cont = oc ? (cont || go) : false;           // grid.cu:205, 256
                        if (go) {
                            // The step loop, written for its dependent instruction count.  (1) No stuck test and (2) no upper
                            // clamp: see fast_seg above.  (3) No select between "dt re-evaluated" (occupied) and "dt as at the
                            // voxel's entry" (empty): dt = max(t * cA, floor) with cA = cone / 0 and floor = step / dt.  (4) the
                            // test t + dt/2 < e as ONE fma: dt * 0.5 is exact, so fma(dt, 0.5, t) rounds the same sum once.
                            // add, mul, max, fma, compare: 8 instructions per step with the loop's own three (13 before, 20 with
                            // the stuck test).
                            if constexpr (kFast) {
                                const float cA = oc ? cone : 0.0f, fl = oc ? step_size : dt;
                                int k = 0;
                                do {
                                    t = t + dt;
                                    dt = fmaxf(t * cA, fl);
                                    ++k;
                                } while (fmaf(dt, 0.5f, t) < e);
                                k_oc += oc ? k : 0;
#ifdef NFA_PHASE_CYCLES
                                ph_[11] += (unsigned long long)k;
#endif
                            } else {
                                while (go) {
                                    const float nt = t + dt;
                                    const bool stuck = nt == t;
                                    n_sm += oc ? 1 : 0;
                                    t = (stuck && !oc) ? e : nt;            // stuck lattice in an empty voxel: as the oracle's lattice_skip
                                    dead = stuck && oc;                     // (the reference would spin here forever)
                                    dt = oc ? march_dt(nt, cone, step_size) : dt;
                                    go = !stuck && t + dt * 0.5f < e;
#ifdef NFA_PHASE_CYCLES
                                    ph_[8] += 1ull;
#endif
                                }
                            }
                        }
                    }
                };
                if (fast_seg) voxels(std::true_type());
                else voxels(std::false_type());
                n_sm += k_oc;
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
            __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): the general walk's loads end HERE, not at the join (split_walk.hpp has the story)
        }
        if (a.iv_cnts) a.iv_cnts[r] = out_iv;
        a.sm_cnts[r] = out_sm;
    }
    NFA_PHASE_MARK(5);
    publish_wave_sums(out_iv, out_sm, out_ovf, block_sums);
    NFA_PHASE_MARK(6);
    NFA_PHASE_END();
}
