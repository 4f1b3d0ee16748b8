// sample_fused.hpp — the tail that turns a count kernel into the WHOLE sampling call: offsets and emit pass in the launch that
// counted (part of grid.hip: included there, inside namespace nfa::{anonymous}, behind emit_pass.hpp).
//
// The reference does count -> cumsum + .item() -> allocate -> fill (grid.cu:413-450, data_spec.hpp:86-96); rounds 1-5 here did
// count kernel -> offsets kernel -> emit kernel: at the training size (6.5 k rays) 26.7 + 5.9 + 10.4 us of kernels plus two kernel
// boundaries for what is, between the kernels, 52 KB of counts.  With this tail a workgroup of the count kernel
//   1. adds up the samples / edges / overflowed rays of its rays (LDS, one barrier: its waves have finished counting),
//   2. publishes the sum and looks back over the workgroups before it (lookback.hpp) — the exclusive sum of data_spec.hpp:90,
//   3. writes `sm_starts` of its rays, and
//   4. every WAVE expands its own rays' run records into (ray_indices, t_starts, t_ends) with the tile form of the emit pass
//      (emit_pass.hpp: the wave's 64 / P rays are one ray block, the records it reads were stored by its own workgroup a moment
//      ago), into outputs the caller sized from its previous call (`capacity`; a wave whose samples end beyond it stores nothing).
// The workgroup that is through with its look-back LAST stores the totals and the caller's stamp — the host is released when every
// ray has been COUNTED and every hand-off decided, not when the last sample is written.  totals[1] = -1 says the look-back gave up (bounded
// wait): counts, run records and wave sums are complete as after nfa_traverse_count, and the caller goes on with
// nfa_traverse_offsets.
#pragma once

#ifdef NFA_FUSE_TRACE
// instrumentation builds (tools/fuse_trace.py): 100 MHz wall-clock stamps per workgroup — kernel entry is stamped by the kernel,
// [1] every wave has counted (wave 0 past the first barrier), [2] look-back through, [3] the last wave's emit has ended
__device__ unsigned long long g_fuse_trace[512][4];
#define NFA_FUSE_STAMP(i) do { if (threadIdx.x == 0 && blockIdx.x < 512) g_fuse_trace[blockIdx.x][i] = wall_clock64(); } while (0)
#define NFA_FUSE_STAMP_MAX(i) do { sync_drain(); if (lane_id() == 0 && blockIdx.x < 512) atomicMax(&g_fuse_trace[blockIdx.x][i], (unsigned long long)wall_clock64()); } while (0)
#else
#define NFA_FUSE_STAMP(i) do {} while (0)
#define NFA_FUSE_STAMP_MAX(i) do {} while (0)
#endif

struct FuseArgs {
    uint64_t *sync;        // NFA_SYNC_BYTES, zero on entry, zero on exit
    int64_t capacity;      // samples the outputs hold; 0: count + offsets only
    int64_t stamp;
    int64_t *totals_dev;   // the workspace's copy of the totals (what nfa_traverse_offsets leaves there)
    int seg_cap;           // emit pass: segment-list entries per wave
    uint64_t spin;         // bound of the look-back's wait in ticks of the 100 MHz clock (lookback.hpp: sync_spin_ticks)
};

// The emit pass's hook of the single launch (emit_pass.hpp: Hook): counts and wave-relative starts from the count kernel's registers;
// wait() = the rest of the hand-off — wave 0 finishes the workgroup's look-back, the workgroup meets, every wave learns where its first
// sample goes and stores its rays' starts.  Runs exactly once per wave.
template <int NW>
struct FusedEmitHook {
    static constexpr bool kDeferred = true;
    const nfa_traverse_args &a;
    const FuseArgs &f;
    int64_t (*s_w)[3];            // [NW][3] in LDS: samples, edges, overflowed rays of every wave
    int64_t *s_pre;               // LDS: the workgroup's exclusive prefix (-1: the look-back gave up)
    int64_t r;                    // this lane's ray
    bool own_first;               // this lane is the first of its ray's lanes and the ray is inside the batch
    int64_t cnt_ray, rel_ray;     // the lane's ray: samples, first sample relative to the wave's first (the same in all lanes of a ray)
    int64_t wave_total, b_sm;     // samples of the wave / of the workgroup
    int lane, wv;

    __device__ __forceinline__ void counts(int64_t, bool own, int64_t &cnt, int64_t &S) const {
        cnt = own ? cnt_ray : 0;
        S = own ? rel_ray : 0;
    }
    __device__ __forceinline__ int64_t wait() {
        const int64_t b = blockIdx.x, nb = gridDim.x;
        int64_t excl = 0;
        if (wv == 0) {
            excl = sync_finish_lookback(f.sync, b, b_sm, lane, f.spin);
            if (lane == 0) *s_pre = excl;
            NFA_FUSE_STAMP(2);
        }
        __syncthreads();
        // behind the barrier (the workgroup's other waves are off to their stores): this workgroup's leave, and — on the workgroup that
        // leaves LAST, the only one that knows every look-back of the launch has been decided (lookback.hpp) — the call's totals and
        // the host's stamp
        if (wv == 0) {
            sync_deposit(f.sync, excl, excl + b_sm, b == nb - 1, lane);
            sync_leave(f.sync, nb, lane, [&](bool gave_up, int64_t total) {
                if (lane != 0) return;
                // (every workgroup added its edges / overflowed rays before its state became visible)
                const int64_t ed = (int64_t)sync_load(f.sync + 1), ov = (int64_t)sync_load(f.sync + 2);
                const int64_t n = gave_up ? -1 : total;
                f.totals_dev[0] = ed; f.totals_dev[1] = n; f.totals_dev[2] = ov; f.totals_dev[3] = 0;
                a.totals[0] = ed; a.totals[1] = n; a.totals[2] = ov;
                __threadfence_system();
                a.totals[3] = f.stamp;
            });
        }
        const int64_t pre = *s_pre;
        if (pre < 0) return -1;
        int64_t base = pre;
#pragma unroll
        for (int w = 0; w < NW; ++w) base += w < wv ? s_w[w][0] : 0;
        if (own_first) a.sm_starts[r] = base + rel_ray;
        if (f.capacity <= 0 || !(a.sm_ray_indices || a.t_starts || a.sm_vals)) return -1;
        if (base + wave_total > f.capacity) return -1;      // (wave-uniform) the caller's guess was too small for this wave's samples: it launches the emit pass itself
        return base;
    }
};

// `own`: this lane holds a ray's results (the first of the ray's P lanes, ray inside the batch).  All threads of the workgroup call.
template <int BLK, int P>
__device__ __forceinline__ void fused_sample_tail(const nfa_traverse_args &a, const RunStore &rs, const FuseArgs &f, char *smem,
                                                  int64_t *__restrict__ wave_sums, int64_t r, bool own, int64_t out_iv, int64_t out_sm, int64_t out_ovf)
{
    constexpr int NW = BLK / 64, RPW = 64 / P;
    static_assert(RPW >= 1 && (RPW & (RPW - 1)) == 0, "rays per wave must be a power of two");
    __shared__ int64_t s_w[NW][3];
    __shared__ int64_t s_pre;
    const int lane = lane_id(), wv = wave_in_block();
    const int64_t w_iv = wave_sum_i64(own ? out_iv : 0), w_sm = wave_sum_i64(own ? out_sm : 0), w_ov = wave_sum_i64(own ? out_ovf : 0);
    if (lane == 0) {
        // (the per-wave triples of the plain count kernel: what nfa_traverse_offsets reads if the look-back gives up)
        const int64_t w = (int64_t)blockIdx.x * NW + wv;
        wave_sums[3 * w] = w_iv; wave_sums[3 * w + 1] = w_sm; wave_sums[3 * w + 2] = w_ov;
        s_w[wv][0] = w_sm; s_w[wv][1] = w_iv; s_w[wv][2] = w_ov;
    }
    __syncthreads();                                    // every wave of the workgroup has counted; its stores are on their way to L2
    NFA_FUSE_STAMP(1);
    int64_t b_sm = 0;
    if (wv == 0) {
        int64_t b_iv = 0, b_ov = 0;
#pragma unroll
        for (int w = 0; w < NW; ++w) { b_sm += s_w[w][0]; b_iv += s_w[w][1]; b_ov += s_w[w][2]; }
        sync_publish(f.sync, (int64_t)blockIdx.x, b_sm, b_iv, b_ov, lane);      // the others can count on it from here on
    }
    // the rays of this wave: counts to all lanes of a ray, starts relative to the wave's first sample
    const int64_t mine = own ? out_sm : 0;
    const int ray_l = lane / P;
    int64_t rel = 0;
#pragma unroll
    for (int k = 0; k < RPW; ++k) {
        const int64_t v = __shfl(mine, k * P, 64);
        rel += k < ray_l ? v : 0;
    }
    const int64_t cnt_ray = __shfl(mine, ray_l * P, 64);
    FusedEmitHook<NW> hook{a, f, s_w, &s_pre, r, own, cnt_ray, rel, w_sm, b_sm, lane, wv};
    if (f.capacity <= 0 || !(a.sm_ray_indices || a.t_starts || a.sm_vals)) {      // count + offsets only
        (void)hook.wait();
        return;
    }
    // the run records and run counts the emit code reads were stored by this workgroup's waves before the barrier above:
    // workgroup-scope release / acquire = the stores have reached L2 before the loads are issued
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    int rb_log2 = 0;
    while ((1 << rb_log2) < RPW) ++rb_log2;
    const int64_t blk = (int64_t)blockIdx.x * NW + wv;  // ray block of 2^rb_log2 = RPW rays: exactly this wave's rays
    // the segment lists are built while the look-back is still waiting for the slowest workgroups; only the stores need its result
    emit_by_tiles<false>(a, rs, rb_log2, f.seg_cap, (unsigned char *)smem + wv * emit_lds_per_wave(f.seg_cap, false), (int64_t)0, f.capacity,
                         blk, (int64_t)1, blk + 1, hook);
    NFA_FUSE_STAMP_MAX(3);
}
