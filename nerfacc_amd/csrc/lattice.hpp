// lattice.hpp — exact closed forms for the marching lattice  t_{k+1} = RN(t_k + d).
//
// The reference advances its sampling lattice with one fp32 add per step
// (`t_last += dt`, grid.cu:160,202,215) and decides sample counts by comparing those sums, so
// bit-exact ray_indices / pack offsets require the SAME rounding chain.  A chain of ~10^3
// dependent adds per ray is what makes a lane-per-ray traversal slow, and what forbids
// random access to the k-th lattice point.  These helpers remove both limits without changing
// a single bit:
//
//   Inside one binade [2^e, 2^(e+1)) every t is m * u (u = 2^(e-23), m a 24-bit integer) and
//   RN(m*u + d) = (m + c) * u with the SAME integer c for every m, as long as the exact sum
//   stays <= 2^(e+1):  c = RN(d / u), and when d / u is exactly half-way, round-half-even makes
//   the increment c0 + (c0 & 1) once m is even.  So j steps are one integer multiply-add.
//   Steps that leave the binade (or start from zero / below d's binade) are taken as real
//   fp32 adds.  A walk over ~10^3 lattice points costs one trip of the loop below — an integer jump to the
//   binade's edge and the real add that crosses it — per binade (about a dozen between dt = 5e-3 and t = 8).
//
// Compiled for both the device (HIP) and the host (oracle/test_lattice.c checks it against the
// plain sequential loop on millions of random cases; tests/test_lattice.py).
#pragma once

#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#define NFA_HD __host__ __device__ __forceinline__
#else
#define NFA_HD static inline
#endif

NFA_HD uint32_t nfa_f2u(float x) { uint32_t u; memcpy(&u, &x, 4); return u; }
NFA_HD float nfa_u2f(uint32_t u) { float x; memcpy(&x, &u, 4); return x; }

// Exact value after j sequential steps t = RN(t + d) (j >= 0, d > 0).  If the sequence gets
// stuck (t + d == t) the stuck value is returned.  `*taken` (optional) = steps really taken.
//
// Round 5: the trip is ONE basic block.  Round 2-4's body had three data-dependent `break`s and 64-bit step counts inside; the
// compiler cut it into a dozen exec-masked blocks (~110 instructions and ~12 branches per binade: profiles/r04_count_pass.md
// section 4 — lane 0's jump from the near plane, ~10 binades, was 11 k cycles of every count-pass wave).  Now every quantity is
// computed unconditionally in 32 bits and selected, the loop's only exit is its condition, and the plain-add prologue stops at
// 8 d instead of 32 d (a trip is cheap enough to take the binades [8 d, 32 d) as well).
NFA_HD float nfa_lattice_advance(float t, float d, int64_t j, int64_t *taken)
{
    const int64_t j0 = j;
    const uint32_t db = nfa_f2u(d);
    const int ed = (int)((db >> 23) & 0xffu);
    const uint32_t D = (db & 0x7fffffu) | 0x800000u;
    const bool d_ok = ed >= 1 && ed < 255 && (db >> 31) == 0;     // positive normal
    // Close to zero the binades are short (a binade below 8 d holds < 8 lattice points): plain
    // adds are cheaper there than the per-binade bookkeeping below.
    if (d_ok) {
        const float small = d * 8.0f;
        if (t > -small) {                               // (t only grows: checked once; a 32-bit counter keeps the loop short)
            const int lim = j < 24 ? (int)j : 24;
            int cnt = 0;
            while (cnt < lim && t < small) { t = t + d; ++cnt; }
            j -= cnt;
        }
    }
    // One trip = either n >= 1 steps inside the current binade (integer multiply-add) followed by the real add that leaves
    // it, or one real fp32 add (ties on an odd mantissa, anything irregular).  Both candidates are computed and one is
    // selected: on the GPU a wave whose lanes disagree would execute both sides of a branch anyway, and straight-line
    // code keeps a lone wave's pipeline full.
    bool live = j > 0;
    while (live) {
        const uint32_t tb = nfa_f2u(t);
        const int e = (int)((tb >> 23) & 0xffu);
        const int sh = e - ed;
        const bool ok = d_ok && (tb >> 31) == 0 && e >= 1 && e < 254 && sh >= 0 && sh <= 24;
        const uint32_t shc = ok ? (uint32_t)sh : 0u;
        const uint32_t m = (tb & 0x7fffffu) | 0x800000u;
        const uint32_t c0 = D >> shc;
        const uint32_t rem = D & ((1u << shc) - 1u);
        const uint32_t half = (1u << shc) >> 1;
        const bool tie = rem == half && rem != 0u;               // d / u exactly half-way
        const bool tie_odd = tie && (m & 1u);                     // one real step makes m even
        // round to nearest; on a tie round-half-even makes the increment c0 + (c0 & 1) once m is even
        const uint32_t c = c0 + (rem > half ? 1u : 0u) + ((tie && !(m & 1u)) ? (c0 & 1u) : 0u);
        const uint32_t lim = (1u << 24) - c0 - 1u;                // next sum may pass 2^(e+1) beyond this
        // (c == 0: t + d rounds back to t — the real add below finds the walk stuck)
        const bool fast = ok && !tie_odd && c != 0u && m <= lim;
        const float nt = t + d;
        const bool stuck = !fast && nt == t;
        // steps that provably stay inside the binade: floor((lim - m) / c) + 1.  The quotient is
        // taken in fp32 (both operands < 2^24: exact, and the correctly rounded quotient truncates
        // to floor or floor + 1) and corrected once — an integer divide is ~40 instructions on the GPU.
        const uint32_t cd = fast ? c : 1u;
        const uint32_t x = fast ? lim - m : 0u;
        uint32_t q = (uint32_t)((float)x / (float)cd);
        q -= (q * cd > x) ? 1u : 0u;                               // (q cd <= x + cd < 2^25)
        const uint32_t jmax = q + 1u;                              // <= 2^24
        const uint32_t jc = j > (int64_t)0x40000000 ? 0x40000000u : (uint32_t)j;
        const uint32_t n = fast ? (jmax < jc ? jmax : jc) : 1u;
        const uint32_t m2 = m + n * cd;                            // <= 2^24
        const float tf = (m2 >> 24) ? nfa_u2f((uint32_t)(e + 1) << 23) : nfa_u2f(((uint32_t)e << 23) | (m2 & 0x7fffffu));
        const float t1 = fast ? tf : nt;
        const int64_t j1 = j - (int64_t)n;
        // A jump that stopped at the binade's edge is followed by a real add (the crossing): taken here instead of paying
        // another trip's bookkeeping for it — the usual walk is then ONE trip per binade instead of two.
        const bool cross = fast && n == jmax && j1 > 0;
        const float t2 = t1 + d;
        const bool stuck2 = cross && t2 == t1;
        const bool crossed = cross && !stuck2;
        if (!stuck) {
            t = crossed ? t2 : t1;
            j = j1 - (crossed ? 1 : 0);
        }
        live = !stuck && !stuck2 && j > 0;
    }
    if (taken) *taken = j0 - j;
    return t;
}

// How far below a real-number estimate of the step count a jump has to aim so that it (almost) never overshoots: inside a
// binade every step makes the same rounding error (<= half an ulp), so k steps drift by at most k/2 ulps = (k 2^-12)^2 steps.
// Only speed depends on it: a jump is used only if the value it lands on is verified to be short of the target.
NFA_HD int64_t nfa_jump_margin(int64_t guess)
{
    const int64_t m = (guess >> 12) + 1;
    return 2 + m * m;
}

// The loop  `while (t + d/2 < target) { nt = t + d; if (nt == t) {stuck} t = nt; ++k; }`
// (grid.cu:157-161, 199-203, 208-216 with constant dt): returns the final t, the number of
// steps in *steps, and whether the walk got stuck before reaching the target.
NFA_HD float nfa_lattice_until(float t, float d, float target, int64_t *steps, bool *stuck)
{
    const float h = d * 0.5f;
    int64_t k = 0;
    *stuck = false;
    // Same-binade shortcut (most boundaries of a ray lie in the binade the walk is in): inside [2^e, 2^(e+1)) the chain is
    // m -> m + c (integer ulps, c = RN(d / ulp)) and the test compares integers too, so the step count is ONE division.
    // Nothing is assumed: the candidate is accepted only if the real fp32 test fails one step earlier and holds at the
    // candidate, for chain values that are exact by construction (no tie, no binade crossing); anything else takes the
    // general path below.
    {
        const uint32_t tb = nfa_f2u(t), db = nfa_f2u(d), gb = nfa_f2u(target);
        const int e = (int)((tb >> 23) & 0xffu), ed = (int)((db >> 23) & 0xffu);
        const int sh = e - ed;
        if ((tb >> 31) == 0 && (db >> 31) == 0 && (gb >> 31) == 0 && e >= 1 && e < 254 && ed >= 1 && sh >= 1 && sh <= 23 &&
            (int)((gb >> 23) & 0xffu) == e && t + h < target) {
            const uint32_t D = (db & 0x7fffffu) | 0x800000u;
            const uint32_t c0 = D >> sh, rem = D & ((1u << sh) - 1u), half = (1u << sh) >> 1;
            const uint32_t c = c0 + (rem > half ? 1u : 0u);
            const uint32_t m = (tb & 0x7fffffu) | 0x800000u, g = (gb & 0x7fffffu) | 0x800000u;
            if (rem != half && c != 0u && g > m) {
                // smallest n with m + n c + (h in ulps, about c / 2) >= g: estimate from the real-number inequality, then verify
                const float need = (float)(g - m) - (float)c * 0.5f;
                int64_t n = need > 0.0f ? (int64_t)(need / (float)c) : 0;
                if (n < 1) n = 1;
                const uint64_t m_prev = (uint64_t)m + (uint64_t)(n - 1) * c, m_n = m_prev + c;
                if (m_n < (1u << 24)) {                              // both values inside the binade: exact chain values
                    const float t_prev = nfa_u2f(((uint32_t)e << 23) | ((uint32_t)m_prev & 0x7fffffu));
                    const float t_n = nfa_u2f(((uint32_t)e << 23) | ((uint32_t)m_n & 0x7fffffu));
                    if (t_prev + h < target && !(t_n + h < target)) { *steps = n; return t_n; }
                    const uint64_t m_n1 = m_n + c;                   // the estimate is one short about half of the time
                    if (m_n1 < (1u << 24)) {
                        const float t_n1 = nfa_u2f(((uint32_t)e << 23) | ((uint32_t)m_n1 & 0x7fffffu));
                        if (t_n + h < target && !(t_n1 + h < target)) { *steps = n + 1; return t_n1; }
                    }
                }
            }
        }
    }
    if (t + h < target) {
        // jump most of the way: an under-estimate of the step count, verified after the jump
        // (the walk is monotone, so "still short of the target" proves no overshoot)
        const float est = (target - h - t) / d;
        if (est > 24.0f && est < 1.0e9f) {
            const int64_t guess = (int64_t)est;
            const int64_t j = guess - nfa_jump_margin(guess);
            int64_t took = 0;
            const float tj = nfa_lattice_advance(t, d, j, &took);
            if (tj + h < target) { t = tj; k = took; }
        }
        while (t + h < target) {
            const float nt = t + d;
            if (nt == t) { *stuck = true; break; }
            t = nt;
            ++k;
        }
    }
    *steps = k;
    return t;
}
