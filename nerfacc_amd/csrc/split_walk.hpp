// split walk of the count pass: P lanes per ray, one level (traverse_count_split_kernel) — part of grid.hip (included there, inside namespace nfa::{anonymous}, after the helpers it uses); moved out
// of grid.hip in round 3 for size only: the text is unchanged.
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
        const int slot_ = (int)blockIdx.x * (int)(blockDim.x >> 6) + (int)(threadIdx.x >> 6);           \
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
    // P = 32: a ray's group is two DPP rows; the upper row adds the lower row's total (row_bcast:15 hands lane 15 of every row to the next)
    if (P > 16) { const int u = __builtin_amdgcn_update_dpp(0, v, kDppRowBcast15, 0xf, 0xf, false); if (part >= 16) v += u; }
    return v;
}
template <int P>
__device__ __forceinline__ int64_t group_incl_sum_i64(int64_t v, int part) {
    if (P > 1) { const int64_t u = group_shr_i64<1>(v); if (part >= 1) v += u; }
    if (P > 2) { const int64_t u = group_shr_i64<2>(v); if (part >= 2) v += u; }
    if (P > 4) { const int64_t u = group_shr_i64<4>(v); if (part >= 4) v += u; }
    if (P > 8) { const int64_t u = group_shr_i64<8>(v); if (part >= 8) v += u; }
    if (P > 16) { const int64_t u = dpp_i64<kDppRowBcast15>(v); if (part >= 16) v += u; }
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
#ifdef NFA_NO_LATE_STAGE                       // A/B builds: the image staged before anything else, as in rounds 2-4
constexpr bool kNoLateStage = true;
#else
constexpr bool kNoLateStage = false;
#endif
// FUSE: the kernel is the whole sampling call — offsets by look-back over the workgroups and the emit pass of every wave's own rays
// behind the count (sample_fused.hpp; `fz` is only read then).
template <bool LDS_OCC, int P, int CAP, int BLK = kBlock, bool XT = false, bool FUSE = false>
__global__ __launch_bounds__(BLK) void traverse_count_split_kernel(nfa_traverse_args a, GridView gv,
                                                                   int64_t *__restrict__ block_sums, RunStore rs, FuseArgs fz)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    NFA_PHASE_BEGIN();
    if constexpr (FUSE) { NFA_FUSE_STAMP(0); }
    const int tid = threadIdx.x, part = tid % P;
    const int64_t R = a.n_rays;
    const int64_t r = (int64_t)blockIdx.x * ((int)blockDim.x / P) + tid / P;      // (blockDim.x <= BLK: the crossing-time form may be launched narrower, grid.hip: split_launch_threads)
    const bool ray_ok = r < R;
    const int64_t rr = ray_ok ? r : 0;
    // the ray's loads are requested BEFORE the occupancy image is staged: their L2 round trip overlaps the image's
    const float o[3] = {a.rays_o[3 * rr], a.rays_o[3 * rr + 1], a.rays_o[3 * rr + 2]};
    const float d[3] = {a.rays_d[3 * rr], a.rays_d[3 * rr + 1], a.rays_d[3 * rr + 2]};
    const float near = ray_near(a, rr), far = ray_far(a, rr);
    // crossing-time form with the image in LDS: the image's words are only REQUESTED here and written to LDS behind the closed-form
    // call below (stage_issue / stage_commit, grid.hip) — everything up to the seam restart works on the ray alone
    constexpr bool kLateStage = XT && LDS_OCC && !kNoLateStage;
    StagePending sp;
    Occ<LDS_OCC> occ;
    if constexpr (kLateStage) {
        occ = stage_layout(gv, smem);
        stage_issue<BLK>(gv, sp);
    } else {
        occ = stage_occupancy<LDS_OCC>(gv, smem);
    }
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
        static_assert(!XT || P == 16 || P == 32, "the crossing-time arrays are filled by lanes 1 .. 3 KPA of a ray's group");
        constexpr int KPA = (P - 1) / 3;          // lanes per axis: 5 of 16, 10 of 32 (lane 31 idles)
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
        if (live && part >= 1 && part <= 3 * KPA) {
            const int ax = (part - 1) / KPA, k = (part - 1) % KPA;
            adv_t = ax == 0 ? s.tx : (ax == 1 ? s.ty : s.tz);
            adv_d = ax == 0 ? s.dx : (ax == 1 ? s.dy : s.dz);
            int n = ax == 0 ? nx : (ax == 1 ? ny : nz);
            const int cap_n = ax == 0 ? gv.res[0] : (ax == 1 ? gv.res[1] : gv.res[2]);     // (gv.res[ax] is a memory load from the kernel arguments + a wait for EVERYTHING in flight)
            n = n < 0 ? 0 : (n > cap_n ? cap_n : n);
            const int L = (n + 1 + KPA - 1) / KPA;
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
        } else if (live && part <= 3 * KPA) {
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
    if constexpr (kLateStage) stage_commit<BLK>(gv, occ, sp);
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
#ifndef NFA_EXP_NO_RECORDS                         // (experiment builds: what do the scattered record stores cost?)
                    rs.t0[(int64_t)idx * R + r] = T_prev;
                    rs.first[(int64_t)idx * R + r] = (int32_t)first;
#endif
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
        // The serial walk's loads are waited for HERE, inside its branch.  The compiler's wait-count pass merges the branches' pending
        // loads at the join below; with a load of this (never taken, in a training step) branch still "in flight" there, the main path
        // got an s_waitcnt vmcnt(0) in front of its last stores — and on gfx9 vmcnt counts STORES too: every wave sat out the write
        // latency of its run records before it could retire (6 k of the mean wave's 48 k cycles, 20 k in the waves with the most runs;
        // profiles/r05_count_pass.md).  0x0F70 = vmcnt(0), the other counters untouched.
        __builtin_amdgcn_s_waitcnt(0x0F70);
    }
    if (ray_ok && part == 0) {
        if (a.iv_cnts) a.iv_cnts[r] = out_iv;
        a.sm_cnts[r] = out_sm;
    }
    NFA_PHASE_MARK(7);
    if constexpr (FUSE) fused_sample_tail<BLK, P>(a, rs, fz, smem, block_sums, r, ray_ok && part == 0, out_iv, out_sm, out_ovf);
    else publish_wave_sums(out_iv, out_sm, out_ovf, block_sums);      // this wave's 64 / P rays
    NFA_PHASE_MARK(8);
    NFA_PHASE_END();
}

