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
//   fp32 adds.  A walk over ~10^3 lattice points costs ~2 real adds + 1 integer jump per
//   binade crossed (about a dozen binades between dt = 5e-3 and t = 8).
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
NFA_HD float nfa_lattice_advance(float t, float d, int64_t j, int64_t *taken)
{
    const int64_t j0 = j;
    const uint32_t db = nfa_f2u(d);
    const int ed = (int)((db >> 23) & 0xffu);
    const uint32_t D = (db & 0x7fffffu) | 0x800000u;
    const bool d_ok = ed >= 1 && ed < 255 && (db >> 31) == 0;     // positive normal
    // Close to zero the binades are short (a binade below 32 d holds < 32 lattice points): plain
    // adds are cheaper there than the per-binade bookkeeping below.
    if (d_ok) {
        const float small = d * 32.0f;
        while (j > 0 && t < small && t > -small) { t = t + d; --j; }
    }
    while (j > 0) {
        const uint32_t tb = nfa_f2u(t);
        const int e = (int)((tb >> 23) & 0xffu);
        const int sh = e - ed;
        bool fast = d_ok && (tb >> 31) == 0 && e >= 1 && e < 254 && sh >= 0 && sh <= 24;
        uint32_t c = 0, c0 = 0, m = 0;
        if (fast) {
            m = (tb & 0x7fffffu) | 0x800000u;
            c0 = D >> sh;
            const uint32_t rem = D & ((1u << sh) - 1u);
            const uint32_t half = sh ? (1u << (sh - 1)) : 0u;
            if (rem == 0u || rem < half) c = c0;
            else if (rem > half) c = c0 + 1u;
            else if (m & 1u) fast = false;             // tie with odd m: one real step makes m even
            else c = c0 + (c0 & 1u);
            if (fast && c == 0u) break;                // t + d rounds back to t: stuck
            if (fast && m > (1u << 24) - c0 - 1u) fast = false;   // next sum may pass 2^(e+1)
        }
        if (!fast) {
            const float nt = t + d;
            if (nt == t) break;
            t = nt;
            --j;
            continue;
        }
        // steps that provably stay inside the binade: floor((lim - m) / c) + 1.  The quotient is
        // taken in fp32 and corrected downwards (never upwards: a smaller count is always safe,
        // it only costs one more trip through this loop) — an integer divide is ~40 instructions
        // on the GPU.
        const uint32_t lim = (1u << 24) - c0 - 1u;
        const uint32_t x = lim - m;
        uint32_t q = (uint32_t)((float)x / (float)c);
        while ((uint64_t)q * c > x) --q;
        const uint64_t jmax = (uint64_t)q + 1u;
        const uint64_t n = jmax < (uint64_t)j ? jmax : (uint64_t)j;
        m += (uint32_t)n * c;                          // <= 2^24
        j -= (int64_t)n;
        t = (m >> 24) ? nfa_u2f((uint32_t)(e + 1) << 23) : nfa_u2f(((uint32_t)e << 23) | (m & 0x7fffffu));
    }
    if (taken) *taken = j0 - j;
    return t;
}

// The loop  `while (t + d/2 < target) { nt = t + d; if (nt == t) {stuck} t = nt; ++k; }`
// (grid.cu:157-161, 199-203, 208-216 with constant dt): returns the final t, the number of
// steps in *steps, and whether the walk got stuck before reaching the target.
NFA_HD float nfa_lattice_until(float t, float d, float target, int64_t *steps, bool *stuck)
{
    const float h = d * 0.5f;
    int64_t k = 0;
    *stuck = false;
    if (t + h < target) {
        // jump most of the way: an under-estimate of the step count, verified after the jump
        // (the walk is monotone, so "still short of the target" proves no overshoot)
        const float est = (target - h - t) / d;
        if (est > 24.0f && est < 1.0e9f) {
            const int64_t guess = (int64_t)est;
            const int64_t j = guess - 2 - (guess >> 6);
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
