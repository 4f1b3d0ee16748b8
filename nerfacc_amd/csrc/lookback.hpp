// lookback.hpp — hand-offs between the workgroups of ONE launch: block aggregates + decoupled look-back over a caller-owned
// "sync block" (NFA_SYNC_BYTES of device memory, zero before its first use, left zero by every kernel that uses it).
//
// Why it exists (round 6): at the training size (6.5 k rays, 2.5e5 samples) every kernel of the path is
// latency-bound and the step pays a kernel boundary + a launch for each "a few KB between two kernels" hand-off — the exclusive
// sum of the per-ray counts between the count and the emit pass (the reference: cumsum + .item(), data_spec.hpp:86-96), the
// prefix of the per-tile survivor counts between the mask pass and the compaction (the reference: three boolean-mask gathers,
// occ_grid.py:180-220).  With these helpers a workgroup publishes its aggregate, sums the aggregates of the workgroups BEFORE it
// (static ids: workgroup b only ever waits for workgroups < b) and goes on with the second half of the work in the same launch.
//
// Visibility between CUs / XCDs (MI355X_MICROARCH.md, "inter-workgroup visibility"): the states are single 64-bit words
// [status:2 | value:62] moved with agent-scope relaxed atomics (sc1 stores / loads: write-through, L1-bypassing) — the value
// travels IN the word that says it is there, so no fence orders anything.  Totals that only the last workgroup needs are
// accumulated with agent-scope atomic adds, drained (s_waitcnt vmcnt(0)) before the state word is stored.
//
// Forward progress: the hardware dispatches workgroups in id order in practice but HIP promises nothing, and two such kernels on
// two streams could in principle fill the chip with waiters.  So every wait is BOUNDED (kSpinLimitTicks of the 100 MHz wall
// clock): a workgroup that gives up publishes ABORT, which every later workgroup inherits; the kernel's host-visible result then
// says "not done here" and the caller runs the unfused kernels — never a hang, and the results are the same either way.
#pragma once

#include "common.hpp"

namespace nfa {

constexpr int kSyncHeaderWords = 32;                                       // done | edges | overflow rays | (spare)
constexpr int kSyncMaxBlocks = NFA_SYNC_BYTES / 8 - kSyncHeaderWords;      // states behind the header
constexpr uint64_t kStValueMask = (1ull << 62) - 1ull;
constexpr uint64_t kStAgg = 1, kStPrefix = 2, kStAbort = 3;                // 0 = nothing yet
constexpr uint64_t kSpinLimitTicks = 200000;                               // 2 ms of s_memrealtime (100 MHz)
// what the host hands the kernels: option `sync_spin_us` (unset: 2 ms; 0: a look-back gives up at the first state that is not there yet —
// how the tests reach the callers' fallback paths)
inline uint64_t sync_spin_ticks() { return (uint64_t)opt(OPT_SYNC_SPIN_US, (int64_t)(kSpinLimitTicks / 100)) * 100ull; }

__device__ __forceinline__ uint64_t sync_load(const uint64_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void sync_store(uint64_t *p, uint64_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void sync_add(uint64_t *p, uint64_t v) { (void)__hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// every memory operation this wave has issued is complete (inline asm: the compiler's wait-count pass cannot drop it)
__device__ __forceinline__ void sync_drain() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// Sum of the values of states [0, b) — all 64 lanes of a wave call it and get the same answer; -1 = a state before b says ABORT,
// or the wait ran out.  256 states per round trip, lane l looking at b - 1 - l, b - 65 - l, ... (position 0 = the nearest); the
// walk stops at the nearest state that carries a PREFIX (the sum of everything up to and including its workgroup).
__device__ __forceinline__ int64_t sync_lookback(const uint64_t *__restrict__ st, int64_t b, int lane, uint64_t spin_limit = kSpinLimitTicks) {
    constexpr int U = 4;
    int64_t excl = 0;
    const uint64_t t_begin = wall_clock64();
    int spins = 0;
    for (int64_t j = b - 1; j >= 0; j -= 64 * U) {
        uint64_t v[U];
        unsigned long long take[U];
        bool has_prefix;
        for (;;) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int64_t idx = j - u * 64 - lane;
                v[u] = idx >= 0 ? sync_load(st + idx) : (kStPrefix << 62);         // before the first workgroup: prefix 0
            }
            bool ok = true, aborted = false;
            has_prefix = false;
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const unsigned long long ready = __ballot((v[u] >> 62) != 0);
                const unsigned long long pre = __ballot((v[u] >> 62) == kStPrefix);
                const unsigned long long ab = __ballot((v[u] >> 62) == kStAbort);
                // positions up to the nearest published prefix are summed; everything nearer must be ready
                unsigned long long need = has_prefix ? 0ull : ~0ull;
                if (!has_prefix && pre) { need = ((pre & (0ull - pre)) << 1) - 1ull; has_prefix = true; }
                take[u] = need;
                ok = ok && ((ready & need) == need);
                aborted = aborted || (ab & need) != 0ull;
            }
            if (aborted) return -1;
            if (ok) break;
            // (the clock is a memory-path read: looked at every 16th round; no sleep between rounds — the loads' own round trip is the
            //  polling period, and one wave per workgroup polls)
            // `spin_limit` = 0 (option `sync_spin_us` = 0, tests): give up at the first state that is not there yet
            if (spin_limit == 0 || ((++spins & 15) == 0 && wall_clock64() - t_begin > spin_limit)) return -1;
        }
        int64_t sum = 0;
#pragma unroll
        for (int u = 0; u < U; ++u) sum += ((take[u] >> lane) & 1ull) ? (int64_t)(v[u] & kStValueMask) : (int64_t)0;
        excl += wave_sum_i64(sum);
        if (has_prefix) break;
    }
    return excl;
}

// One wave of workgroup b (all 64 lanes): publish the workgroup's aggregate `agg` (>= 0), wait for the sum of the aggregates before
// it, publish the inclusive prefix.  `extra0` / `extra1` are added to header words 1 / 2 BEFORE the aggregate becomes visible (totals only
// the last workgroup reads).  Returns the exclusive prefix, or -1 (ABORT published).  The LAST workgroup to leave — whichever it is
// — zeroes the states and the header again (`done` counts the workgroups that are through with the states).
__device__ __forceinline__ void sync_publish(uint64_t *__restrict__ sync, int64_t b, int64_t agg, int64_t extra0, int64_t extra1, int lane) {
    uint64_t *st = sync + kSyncHeaderWords;
    if (lane == 0) {
        if (extra0) sync_add(sync + 1, (uint64_t)extra0);
        if (extra1) sync_add(sync + 2, (uint64_t)extra1);
        sync_drain();
        sync_store(st + b, (kStAgg << 62) | (uint64_t)agg);
    }
}
// (the two halves of the hand-off, for a caller that has work to do between publishing its aggregate and needing the prefix)
__device__ __forceinline__ int64_t sync_finish_lookback(uint64_t *__restrict__ sync, int64_t b, int64_t agg, int lane, uint64_t spin_limit = kSpinLimitTicks) {
    uint64_t *st = sync + kSyncHeaderWords;
    const int64_t excl = sync_lookback(st, b, lane, spin_limit);
    if (lane == 0) sync_store(st + b, excl < 0 ? (kStAbort << 62) : ((kStPrefix << 62) | (uint64_t)(excl + agg)));
    return excl;
}
__device__ __forceinline__ int64_t sync_publish_and_lookback(uint64_t *__restrict__ sync, int64_t b, int64_t agg, int64_t extra0, int64_t extra1, int lane,
                                                             uint64_t spin_limit = kSpinLimitTicks) {
    sync_publish(sync, b, agg, extra0, extra1, lane);
    return sync_finish_lookback(sync, b, agg, lane, spin_limit);
}
// Called by the same wave when it no longer needs the states or the header: the last caller of the launch resets them — after
// `verdict(aborted, total)` has run on it (all 64 lanes; `aborted`: some workgroup's look-back gave up; `total`: what the highest
// workgroup deposited, see sync_deposit).  The VERDICT of a launch — its total, or "a hand-off gave up: run the separate kernels" —
// has to come from the workgroup that leaves LAST: only it knows that every other workgroup has decided.  (The highest workgroup can
// finish its own look-back while a lower one, which started to wait earlier, is about to give up on a state that has just arrived:
// reporting from there would call a launch complete whose lower workgroup stored nothing.)
//   header words: [0] workgroups that have left, [1] / [2] the caller's extras, [3] workgroups that gave up, [4] total + 1
__device__ __forceinline__ void sync_deposit(uint64_t *__restrict__ sync, int64_t excl, int64_t total_if_highest, bool highest, int lane) {
    if (lane != 0) return;
    if (excl < 0) sync_add(sync + 3, 1ull);
    else if (highest) sync_store(sync + 4, (uint64_t)total_if_highest + 1ull);
}
template <class Verdict>
__device__ __forceinline__ void sync_leave(uint64_t *__restrict__ sync, int64_t n_blocks, int lane, Verdict verdict) {
    unsigned long long d = 0;
    if (lane == 0) {
        sync_drain();                                      // this workgroup's state stores, its deposit, have reached memory
        d = __hip_atomic_fetch_add((unsigned long long *)sync, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    d = (unsigned long long)readfirstlane_i64((int64_t)d);
    if ((int64_t)d != n_blocks - 1) return;
    const uint64_t gave_up = sync_load(sync + 3), total1 = sync_load(sync + 4);
    verdict(gave_up != 0ull || total1 == 0ull, (int64_t)total1 - 1);
    uint64_t *st = sync + kSyncHeaderWords;
    for (int64_t i = lane; i < n_blocks; i += 64) sync_store(st + i, 0ull);
    if (lane < 8) sync_store(sync + lane, 0ull);
}
__device__ __forceinline__ void sync_leave(uint64_t *__restrict__ sync, int64_t n_blocks, int lane) {
    sync_leave(sync, n_blocks, lane, [](bool, int64_t) {});
}

}  // namespace nfa
