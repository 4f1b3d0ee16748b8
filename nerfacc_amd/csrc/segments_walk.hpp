// segment walk of the count pass: several levels, a lane per level segment (traverse_count_segments_kernel) — part of grid.hip (included there, inside namespace nfa::{anonymous}, after the helpers it uses); moved out
// of grid.hip in round 3 for size only: the text is unchanged.
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
        __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0): the serial walk's loads end HERE, not at the join (split_walk.hpp has the story)
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

