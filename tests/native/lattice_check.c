/* tests/native/lattice_check.c — host-side check of nerfacc_amd/csrc/lattice.hpp (the closed-form
 * lattice walk used by the HIP traversal kernels) against the plain sequential fp32 loop.
 * Built and driven by tests/test_lattice.py. */
#include <stdbool.h>
#include <stdint.h>
#include "../../nerfacc_amd/csrc/lattice.hpp"

#define API __attribute__((visibility("default")))

static float seq_advance(float t, float d, int64_t j) {
    for (int64_t k = 0; k < j; ++k) { const float nt = t + d; if (nt == t) break; t = nt; }
    return t;
}
static float seq_until(float t, float d, float target, int64_t *steps, bool *stuck) {
    const float h = d * 0.5f;
    int64_t k = 0;
    *stuck = false;
    while (t + h < target) { const float nt = t + d; if (nt == t) { *stuck = true; break; } t = nt; ++k; }
    *steps = k;
    return t;
}

/* returns the number of mismatches over n cases; first mismatch index in *first */
API int64_t check_advance(int64_t n, const float *t, const float *d, const int64_t *j, int64_t *first) {
    int64_t bad = 0;
    for (int64_t i = 0; i < n; ++i) {
        const float a = nfa_lattice_advance(t[i], d[i], j[i], 0), b = seq_advance(t[i], d[i], j[i]);
        if (nfa_f2u(a) != nfa_f2u(b)) { if (!bad) *first = i; ++bad; }
    }
    return bad;
}
API int64_t check_until(int64_t n, const float *t, const float *d, const float *target, int64_t *first) {
    int64_t bad = 0;
    for (int64_t i = 0; i < n; ++i) {
        int64_t k1, k2; bool s1, s2;
        const float a = nfa_lattice_until(t[i], d[i], target[i], &k1, &s1);
        const float b = seq_until(t[i], d[i], target[i], &k2, &s2);
        if (nfa_f2u(a) != nfa_f2u(b) || k1 != k2 || s1 != s2) { if (!bad) *first = i; ++bad; }
    }
    return bad;
}
