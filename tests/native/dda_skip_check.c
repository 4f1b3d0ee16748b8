/* tests/native/dda_skip_check.c — host-side check of nerfacc_amd/csrc/dda_skip.hpp: the macro step of the voxel DDA must leave exactly
 * the state that the same number of single steps (dda_advance, utils_grid.cuh:116-142) leaves.  Built and driven by tests/test_dda_skip.py
 * (compiled as C++: the header uses references). */
#include <stdint.h>
#include <string.h>
#include "../../nerfacc_amd/csrc/dda_skip.hpp"

#define API extern "C" __attribute__((visibility("default")))

static inline uint32_t fb(float x) { uint32_t u; memcpy(&u, &x, 4); return u; }

/* state arrays: t[3n], d[3n], s[3n], c[3n], o[3n], k1[3n].  Returns mismatches; stats[0] = single steps replaced in total,
 * stats[1] = macro steps that took everything asked for on their final axis, stats[2] = calls */
API int64_t check_skip(int64_t n, const float *t, const float *d, const int32_t *sg, const int32_t *c, const int32_t *o, const int32_t *k1,
                       int64_t *first, int64_t *stats) {
    int64_t bad = 0;
    stats[0] = stats[1] = stats[2] = 0;
    for (int64_t i = 0; i < n; ++i) {
        Dda a;
        a.tx = t[3 * i]; a.ty = t[3 * i + 1]; a.tz = t[3 * i + 2];
        a.dx = d[3 * i]; a.dy = d[3 * i + 1]; a.dz = d[3 * i + 2];
        a.sx = sg[3 * i]; a.sy = sg[3 * i + 1]; a.sz = sg[3 * i + 2];
        a.cx = c[3 * i]; a.cy = c[3 * i + 1]; a.cz = c[3 * i + 2];
        a.ox = o[3 * i]; a.oy = o[3 * i + 1]; a.oz = o[3 * i + 2];
        Dda b = a;
        const DdaSkip r = dda_skip(a, k1[3 * i], k1[3 * i + 1], k1[3 * i + 2]);
        const int total = r.nx + r.ny + r.nz;
        int cnt[3] = {0, 0, 0}, last_axis = -1;
        bool cont = true, ok = total >= 1 && r.nx >= 0 && r.ny >= 0 && r.nz >= 0 && r.nx <= k1[3 * i] + 1 && r.ny <= k1[3 * i + 1] + 1 &&
                               r.nz <= k1[3 * i + 2] + 1;
        float t_cell = 0.f;
        for (int st = 0; st < total && ok; ++st) {
            if (!cont) { ok = false; break; }                       /* an intermediate crossing ended the walk */
            t_cell = fminf(b.tx, fminf(b.ty, b.tz));
            const bool ax = (b.tx < b.ty) && (b.tx < b.tz), ay = !ax && (b.ty < b.tz);
            last_axis = ax ? 0 : (ay ? 1 : 2);
            cnt[last_axis] += 1;
            cont = dda_advance(b);
        }
        ok = ok && cnt[0] == r.nx && cnt[1] == r.ny && cnt[2] == r.nz && cont == r.cont && fb(t_cell) == fb(r.t_exit) &&
             fb(a.tx) == fb(b.tx) && fb(a.ty) == fb(b.ty) && fb(a.tz) == fb(b.tz) && a.cx == b.cx && a.cy == b.cy && a.cz == b.cz;
        /* the final crossing is the only one that may complete an axis' request */
        if (ok && last_axis >= 0) {
            const int other1 = (last_axis + 1) % 3, other2 = (last_axis + 2) % 3;
            ok = cnt[other1] <= k1[3 * i + other1] && cnt[other2] <= k1[3 * i + other2];
            if (cnt[last_axis] == k1[3 * i + last_axis] + 1) stats[1] += 1;
        }
        stats[0] += total;
        stats[2] += 1;
        if (!ok) { if (!bad) *first = i; ++bad; }
    }
    return bad;
}

/* whole walks: from each start state, macro steps with pseudo-random requests (capped by the crossings left to each overflow index)
 * until the walk ends, checked against single steps at every landing point.  Returns mismatching walks. */
API int64_t check_walk(int64_t n, const float *t, const float *d, const int32_t *sg, const int32_t *c, const int32_t *o, uint32_t seed,
                       int64_t *first, int64_t *stats) {
    int64_t bad = 0;
    stats[0] = stats[1] = stats[2] = 0;
    uint32_t rng = seed * 2654435761u + 12345u;
    for (int64_t i = 0; i < n; ++i) {
        Dda a;
        a.tx = t[3 * i]; a.ty = t[3 * i + 1]; a.tz = t[3 * i + 2];
        a.dx = d[3 * i]; a.dy = d[3 * i + 1]; a.dz = d[3 * i + 2];
        a.sx = sg[3 * i]; a.sy = sg[3 * i + 1]; a.sz = sg[3 * i + 2];
        a.cx = c[3 * i]; a.cy = c[3 * i + 1]; a.cz = c[3 * i + 2];
        a.ox = o[3 * i]; a.oy = o[3 * i + 1]; a.oz = o[3 * i + 2];
        Dda b = a;
        bool ok = true, cont = true;
        for (int guard = 0; guard < 4096 && cont && ok; ++guard) {
            int k1[3];
            const int left[3] = {(a.ox - a.cx) * a.sx, (a.oy - a.cy) * a.sy, (a.oz - a.cz) * a.sz};
            for (int ax = 0; ax < 3; ++ax) {
                rng = rng * 1664525u + 1013904223u;
                int k = (int)((rng >> 20) & 15u);
                if ((rng >> 28) < 3u) k = 0;
                k1[ax] = left[ax] >= 1 ? (k < left[ax] - 1 ? k : left[ax] - 1) : 0;
            }
            const DdaSkip r = dda_skip(a, k1[0], k1[1], k1[2]);
            const int total = r.nx + r.ny + r.nz;
            ok = total >= 1;
            float t_cell = 0.f;
            bool bc = true;
            for (int st = 0; st < total && ok; ++st) {
                if (!bc) { ok = false; break; }
                t_cell = fminf(b.tx, fminf(b.ty, b.tz));
                bc = dda_advance(b);
            }
            ok = ok && bc == r.cont && fb(t_cell) == fb(r.t_exit) && fb(a.tx) == fb(b.tx) && fb(a.ty) == fb(b.ty) && fb(a.tz) == fb(b.tz) &&
                 a.cx == b.cx && a.cy == b.cy && a.cz == b.cz;
            cont = r.cont;
            stats[0] += total;
            stats[2] += 1;
        }
        if (!ok) { if (!bad) *first = i; ++bad; }
    }
    return bad;
}

/* the integer-domain walk (IDda): same whole-walk check; states that are not `idda_sane` are skipped (the kernels walk those with
 * dda_advance).  mode 0: idda_advance only (voxel steps), 1: random macro steps */
API int64_t check_iwalk(int64_t n, const float *t, const float *d, const int32_t *sg, const int32_t *c, const int32_t *o, uint32_t seed,
                        int mode, int64_t *first, int64_t *stats) {
    int64_t bad = 0;
    stats[0] = stats[1] = stats[2] = 0;
    uint32_t rng = seed * 2654435761u + 12345u;
    for (int64_t i = 0; i < n; ++i) {
        Dda b;
        b.tx = t[3 * i]; b.ty = t[3 * i + 1]; b.tz = t[3 * i + 2];
        b.dx = d[3 * i]; b.dy = d[3 * i + 1]; b.dz = d[3 * i + 2];
        b.sx = sg[3 * i]; b.sy = sg[3 * i + 1]; b.sz = sg[3 * i + 2];
        b.cx = c[3 * i]; b.cy = c[3 * i + 1]; b.cz = c[3 * i + 2];
        b.ox = o[3 * i]; b.oy = o[3 * i + 1]; b.oz = o[3 * i + 2];
        if (!idda_sane(b)) continue;
        IDda a = idda_init(b);
        bool ok = true, cont = true;
        for (int guard = 0; guard < 4096 && cont && ok; ++guard) {
            int total = 1;
            bool rc;
            float t_exit;
            if (mode == 0) {
                t_exit = idda_t_cell(a);
                rc = idda_advance(a);
            } else {
                int k1[3];
                const int left[3] = {(a.ox - a.cx) * a.sx, (a.oy - a.cy) * a.sy, (a.oz - a.cz) * a.sz};
                for (int ax = 0; ax < 3; ++ax) {
                    rng = rng * 1664525u + 1013904223u;
                    int k = (int)((rng >> 20) & 15u);
                    if ((rng >> 28) < 3u) k = 0;
                    k1[ax] = left[ax] >= 1 ? (k < left[ax] - 1 ? k : left[ax] - 1) : 0;
                }
                const DdaSkip r = idda_skip(a, k1[0], k1[1], k1[2]);
                total = r.nx + r.ny + r.nz;
                rc = r.cont;
                t_exit = r.t_exit;
                ok = total >= 1;
            }
            float t_cell = 0.f;
            bool bc = true;
            for (int st = 0; st < total && ok; ++st) {
                if (!bc) { ok = false; break; }
                t_cell = fminf(b.tx, fminf(b.ty, b.tz));
                bc = dda_advance(b);
            }
            ok = ok && bc == rc && fb(t_cell) == fb(t_exit) && a.x.tb == fb(b.tx) && a.y.tb == fb(b.ty) && a.z.tb == fb(b.tz) &&
                 a.cx == b.cx && a.cy == b.cy && a.cz == b.cz;
            cont = rc;
            stats[0] += total;
            stats[2] += 1;
        }
        if (!ok) { if (!bad) *first = i; ++bad; }
    }
    return bad;
}
