// dda_skip.hpp — the voxel DDA of the traversal (utils_grid.cuh:116-142) and its MACRO STEP: many plane crossings in one go,
// with exactly the state the voxel-by-voxel walk would reach.
//
// Why (round 5): 91 % of the voxels a training ray of the bench scene visits lie in EMPTY 4^3 bricks and 83 % in
// empty 16^3 blocks (188 voxels per ray, 27 "steps" if every empty region is left in one go — tools/experiments/r05_skip_stats.py),
// yet every walk paid one DDA step + one occupancy lookup per voxel.  Samples depend only on the occupied <-> empty boundaries and
// their times, so a walk may cross an empty box in one step PROVIDED it comes out with the same voxel, the same three pending
// crossing times (bit for bit — they are chains t <- RN(t + delta), and later boundary times are their values) and the same count of
// major-axis crossings.  Nothing is approximated:
//   * the crossings of one axis inside a binade are  bits(t) + i * c  with one integer c (lattice.hpp's identity: RN(m u + d) =
//     (m + c) u while the exact sum stays inside the binade), so "the time of the k-th next crossing" is one multiply-add on the bit
//     pattern and "how many crossings precede time X" is a 4-round bisection over integers (k <= 15);
//   * a jump is only taken as far as that identity PROVABLY holds for the axis (same binade, no half-way tie on an odd mantissa,
//     positive normal operands); otherwise the axis asks for fewer crossings — down to one, which is the plain DDA step.  Leaving a
//     box early is always allowed: the box is empty wherever the walk stops inside it;
//   * the order of crossings is the reference's: earlier time first, ties z, then y, then x (the strict '<' chain).
// Host-compiled and checked against repeated single steps on millions of random and adversarial states (tests/test_dda_skip.py).
#pragma once

#include <math.h>

#include "lattice.hpp"

struct Dda {
    float tx, ty, tz;   // t at which the ray crosses the next x / y / z voxel plane
    float dx, dy, dz;   // t between successive planes per axis
    int sx, sy, sz;     // index step per axis (-1, 0, +1)
    int cx, cy, cz;     // current voxel
    int ox, oy, oz;     // first out-of-segment index per axis (final + step)
};

// utils_grid.cuh:116-142: step along the axis whose next plane is strictly nearest (ties go
// z, then y, then x by the strict '<' chain).  Written with selects: the three-way branch of
// the reference makes a wave execute all three arms at almost every voxel.
NFA_HD bool dda_advance(Dda &s) {
    const bool ax = (s.tx < s.ty) && (s.tx < s.tz);
    const bool ay = !ax && (s.ty < s.tz);
    const bool az = !ax && !ay;
    s.cx += ax ? s.sx : 0;
    s.cy += ay ? s.sy : 0;
    s.cz += az ? s.sz : 0;
    s.tx = ax ? s.tx + s.dx : s.tx;
    s.ty = ay ? s.ty + s.dy : s.ty;
    s.tz = az ? s.tz + s.dz : s.tz;
    // (bitwise on purpose: a select between the three overflow indices makes the compiler
    //  spill them to scratch and index them, one scratch load per voxel)
    const bool hit_x = s.cx == s.ox, hit_y = s.cy == s.oy, hit_z = s.cz == s.oz;
    return !((ax & hit_x) | (ay & hit_y) | (az & hit_z));
}

// One axis' chain in the binade of its pending crossing: the integer increment c per crossing and how many further crossings (beyond
// the pending one) the identity covers.  want = crossings asked for beyond the pending one (0 .. 15); the result is <= want.
struct DdaChain {
    uint32_t tb;   // bits of the pending crossing time
    uint32_t c;    // ulps per crossing (valid when n > 0)
    int n;         // further crossings granted
};
NFA_HD DdaChain dda_chain(float t, float d, int want) {
    DdaChain ch;
    const uint32_t tb = nfa_f2u(t), db = nfa_f2u(d);
    ch.tb = tb;
    const int e = (int)(tb >> 23), ed = (int)(db >> 23);                     // (sign bits must be 0: checked below)
    const int sh = e - ed;
    const bool ok = (tb >> 31) == 0u && (db >> 31) == 0u && e >= 1 && e < 254 && ed >= 1 && sh >= 0 && sh <= 24;
    const uint32_t shc = ok ? (uint32_t)sh : 0u;
    const uint32_t D = (db & 0x7fffffu) | 0x800000u, m = (tb & 0x7fffffu) | 0x800000u;
    const uint32_t c0 = D >> shc, rem = D & ((1u << shc) - 1u), half = (1u << shc) >> 1;
    const bool tie = rem == half && rem != 0u;
    // half-way ties: round-half-even makes the increment c0 + (c0 & 1) from an EVEN mantissa on (and keeps it even); from an odd one
    // the next crossing is irregular: no jump this time
    const uint32_t c = c0 + (rem > half ? 1u : 0u) + (tie ? (c0 & 1u) : 0u);
    const uint32_t lim = (1u << 24) - c0 - 1u;                                 // a step from a mantissa beyond this may leave the binade
    const bool regular = ok && c != 0u && !(tie && (m & 1u));
    // crossing i (1 <= i <= n) starts from mantissa m + (i - 1) c, which has to be <= lim: the largest such n, by halving the request
    // (two halvings: a request of 15 becomes 7, then 3, then 0 — no division)
    int n = want;
    n = (m + (uint32_t)(n > 0 ? n - 1 : 0) * c <= lim) ? n : (n >> 1);
    n = (m + (uint32_t)(n > 0 ? n - 1 : 0) * c <= lim) ? n : (n >> 1);
    n = (m + (uint32_t)(n > 0 ? n - 1 : 0) * c <= lim) ? n : 0;
    n = (regular && m <= lim) ? n : 0;
    ch.c = c;
    ch.n = n;
    return ch;
}

// how many of the chain's entries 0 .. n-1 (bits tb + i c) are < lim_bits (integer order == float order: all operands are positive floats)
NFA_HD int dda_chain_count_below(const DdaChain &ch, uint32_t lim_bits) {
    int cnt = 0;
#define NFA_DDA_TRY(B)                                                                             \
    {                                                                                              \
        const int trial = cnt + (B);                                                               \
        const bool take = trial <= ch.n && ch.tb + (uint32_t)(trial - 1) * ch.c < lim_bits;         \
        cnt = take ? trial : cnt;                                                                  \
    }
    NFA_DDA_TRY(8) NFA_DDA_TRY(4) NFA_DDA_TRY(2) NFA_DDA_TRY(1)
#undef NFA_DDA_TRY
    return cnt;
}

struct DdaSkip {
    float t_exit;        // the crossing that ended the step: min(tx, ty, tz) at the LAST voxel before it (that voxel's exit time, unclamped)
    int nx, ny, nz;      // crossings taken per axis (their sum = voxel steps replaced)
    bool cont;           // false: the final crossing reached its axis' overflow index (the walk is over), as dda_advance returns
};

// Generalised dda_advance: take plane crossings in the walk's order up to AND INCLUDING the first crossing that is the (kx1 + 1)-th
// of x, the (ky1 + 1)-th of y or the (kz1 + 1)-th of z — or an earlier one where a chain's closed form is not provable (see top).
// k?1 in 0 .. 15.  With all three 0 this is dda_advance.  The caller guarantees that no crossing BEFORE the final one ends the walk
// (k?1 < crossings left to the axis' overflow index) and that every voxel entered before the final crossing may be skipped.
NFA_HD DdaSkip dda_skip(Dda &s, int kx1, int ky1, int kz1) {
    DdaSkip r;
    const float t_min3 = fminf(s.tx, fminf(s.ty, s.tz));          // the current voxel's exit as the voxel walk forms it (NaN rules of fminf)
    // all three pending times positive and ordered (NaN-free), deltas non-negative: otherwise plain single step
    const bool sane = s.tx > 0.0f && s.ty > 0.0f && s.tz > 0.0f && s.dx >= 0.0f && s.dy >= 0.0f && s.dz >= 0.0f;
    const DdaChain X = dda_chain(s.tx, s.dx, sane ? kx1 : 0), Y = dda_chain(s.ty, s.dy, sane ? ky1 : 0), Z = dda_chain(s.tz, s.dz, sane ? kz1 : 0);
    // candidate final crossing per axis
    const float ex = nfa_u2f(X.tb + (uint32_t)X.n * X.c), ey = nfa_u2f(Y.tb + (uint32_t)Y.n * Y.c), ez = nfa_u2f(Z.tb + (uint32_t)Z.n * Z.c);
    const bool fx = (ex < ey) && (ex < ez);
    const bool fy = !fx && (ey < ez);
    const bool fz = !fx && !fy;
    const float te = fx ? ex : (fy ? ey : ez);
    const uint32_t eb = nfa_f2u(te);
    // crossings of the other axes that precede the final one: time strictly earlier, or equal with the lower rank (z < y < x)
    //   x never wins a tie; y wins a tie against x only; z wins every tie
    const int cx_ = dda_chain_count_below(X, eb);
    const int cy_ = dda_chain_count_below(Y, eb + (fx ? 1u : 0u));
    const int cz_ = dda_chain_count_below(Z, eb + (fz ? 0u : 1u));
    const int nx = fx ? X.n + 1 : cx_, ny = fy ? Y.n + 1 : cy_, nz = fz ? Z.n + 1 : cz_;
    const float px = nfa_u2f(X.tb + (uint32_t)cx_ * X.c), py = nfa_u2f(Y.tb + (uint32_t)cy_ * Y.c), pz = nfa_u2f(Z.tb + (uint32_t)cz_ * Z.c);
    s.tx = fx ? ex + s.dx : px;
    s.ty = fy ? ey + s.dy : py;
    s.tz = fz ? ez + s.dz : pz;
    s.cx += nx * s.sx;
    s.cy += ny * s.sy;
    s.cz += nz * s.sz;
    const bool hit_x = s.cx == s.ox, hit_y = s.cy == s.oy, hit_z = s.cz == s.oz;
    r.t_exit = (nx + ny + nz == 1) ? t_min3 : te;                 // (equal whenever the operands are ordered)
    r.nx = nx; r.ny = ny; r.nz = nz;
    r.cont = !((fx & hit_x) | (fy & hit_y) | (fz & hit_z));
    return r;
}

// ---------------------------------------------------------------------------------------------------------------------------------
// The same walk in the INTEGER domain (round 5, second form).  The macro step above re-derives every axis' binade constants at
// every call (~380 instructions on gfx950 with its halvings and bisections: measured SLOWER than the voxel walk it replaces).  Here
// the constants live in the walk's state and are refreshed only when a chain leaves its binade (once or twice per ray and axis):
//   tb   bits of the pending crossing time (a positive float: integer order == float order)
//   c    ulps per crossing in the pending time's binade
//   n    crossings after the pending one whose times are provably tb + i c  (0: the next one is taken as a real fp32 add)
//   rc   1 / c as a float (for the count of crossings below a bound)
// A voxel step is then three integer compares and one integer add; a macro step is a multiply-add per axis, the same compares, and
// for the two other axes ONE float multiply + a fix-up instead of a bisection.  Valid while all three pending times are positive and
// non-NaN and the deltas non-negative (`idda_sane`) — anything else walks with dda_advance.
// 1 / x, possibly off by an ulp (v_rcp_f32 on the device): only ever used for an estimate that is corrected
#if defined(__HIP_DEVICE_COMPILE__)
NFA_HD float nfa_rcp_approx(float x) { return __builtin_amdgcn_rcpf(x); }
#elif defined(NFA_RCP_PERTURB)      /* host test builds: a reciprocal that is off by a few ulps, as v_rcp_f32 may be */
NFA_HD float nfa_rcp_approx(float x) { return (1.0f / x) * (1.0f + (float)(NFA_RCP_PERTURB) * 1.1920929e-7f); }
#else
NFA_HD float nfa_rcp_approx(float x) { return 1.0f / x; }
#endif
struct IAxis {
    uint32_t tb, c;
    int n;
    float rc;
};
struct IDda {
    IAxis x, y, z;
    float dx, dy, dz;
    int sx, sy, sz, cx, cy, cz, ox, oy, oz;
};
NFA_HD bool idda_sane(const Dda &s) {
    return s.tx > 0.0f && s.ty > 0.0f && s.tz > 0.0f && s.dx >= 0.0f && s.dy >= 0.0f && s.dz >= 0.0f;
}
// constants of the binade the pending time t lies in (exact count of regular crossings: one division, once per binade)
NFA_HD IAxis iaxis_refresh(float t, float d) {
    IAxis a;
    const uint32_t tb = nfa_f2u(t), db = nfa_f2u(d);
    a.tb = tb;
    const int e = (int)(tb >> 23), ed = (int)(db >> 23);
    const int sh = e - ed;
    const bool ok = (tb >> 31) == 0u && (db >> 31) == 0u && e >= 1 && e < 254 && ed >= 1 && sh >= 0 && sh <= 24;
    const uint32_t shc = ok ? (uint32_t)sh : 0u;
    const uint32_t D = (db & 0x7fffffu) | 0x800000u, m = (tb & 0x7fffffu) | 0x800000u;
    const uint32_t c0 = D >> shc, rem = D & ((1u << shc) - 1u), half = (1u << shc) >> 1;
    const bool tie = rem == half && rem != 0u;
    const uint32_t c = c0 + (rem > half ? 1u : 0u) + (tie ? (c0 & 1u) : 0u);
    const uint32_t lim = (1u << 24) - c0 - 1u;
    const bool regular = ok && c != 0u && !(tie && (m & 1u)) && m <= lim;
    // crossing i (1 .. n) starts from mantissa m + (i - 1) c <= lim:  n = floor((lim - m) / c) + 1
    const uint32_t cd = regular ? c : 1u, x = regular ? lim - m : 0u;
    uint32_t q = (uint32_t)((float)x / (float)cd);
    q -= (q * cd > x) ? 1u : 0u;
    a.c = c;
    a.n = regular ? (int)(q + 1u) : 0;
    a.rc = nfa_rcp_approx((float)cd);
    return a;
}
NFA_HD IDda idda_init(const Dda &s) {
    IDda w;
    w.x = iaxis_refresh(s.tx, s.dx); w.y = iaxis_refresh(s.ty, s.dy); w.z = iaxis_refresh(s.tz, s.dz);
    w.dx = s.dx; w.dy = s.dy; w.dz = s.dz;
    w.sx = s.sx; w.sy = s.sy; w.sz = s.sz; w.cx = s.cx; w.cy = s.cy; w.cz = s.cz; w.ox = s.ox; w.oy = s.oy; w.oz = s.oz;
    return w;
}
// the pending crossing of an axis has been taken: its next one
NFA_HD void iaxis_next(IAxis &a, float d) {
    if (a.n > 0) { a.tb += a.c; a.n -= 1; }
    else a = iaxis_refresh(nfa_u2f(a.tb) + d, d);
}
// exit time of the current voxel (min of the three pending times), as the float the voxel walk forms
NFA_HD float idda_t_cell(const IDda &w) {
    const uint32_t m = w.x.tb < w.y.tb ? w.x.tb : w.y.tb;
    return nfa_u2f(m < w.z.tb ? m : w.z.tb);
}
// dda_advance
NFA_HD bool idda_advance(IDda &w) {
    const bool ax = (w.x.tb < w.y.tb) && (w.x.tb < w.z.tb);
    const bool ay = !ax && (w.y.tb < w.z.tb);
    const bool az = !ax && !ay;
    if (ax) { w.cx += w.sx; iaxis_next(w.x, w.dx); }
    if (ay) { w.cy += w.sy; iaxis_next(w.y, w.dy); }
    if (az) { w.cz += w.sz; iaxis_next(w.z, w.dz); }
    const bool hit_x = w.cx == w.ox, hit_y = w.cy == w.oy, hit_z = w.cz == w.oz;
    return !((ax & hit_x) | (ay & hit_y) | (az & hit_z));
}
// entries 0 .. k-1 of an axis' chain (tb + i c, all regular: k <= n) that are < bound, as an integer in 0 .. k.  Straight-line.
NFA_HD int iaxis_count_below(const IAxis &a, int k, uint32_t bound) {
    const uint32_t diff = bound > a.tb ? bound - a.tb : 0u;      // entries i with i c < diff:  ceil(diff / c), capped at k
    const bool all = diff > (uint32_t)k * a.c;
    // otherwise diff <= 15 c: the quotient is < 16 and the float product (rc may be an approximate reciprocal) within one of it
    const uint32_t dd = all ? 0u : diff;
    uint32_t q = (uint32_t)((float)dd * a.rc);
    int32_t r = (int32_t)(dd - q * a.c);
    const bool lo = r < 0;
    q -= lo ? 1u : 0u; r += lo ? (int32_t)a.c : 0;
    const bool hi = r >= (int32_t)a.c;
    q += hi ? 1u : 0u; r -= hi ? (int32_t)a.c : 0;
    const int cnt = (int)q + (r > 0 ? 1 : 0);
    return (all || cnt > k) ? k : cnt;
}
// dda_skip: crossings in the walk's order up to and including the first that is the (k?1 + 1)-th of its axis (fewer where an axis'
// regular range ends sooner).  Same contract and result as dda_skip.
NFA_HD DdaSkip idda_skip(IDda &w, int kx1, int ky1, int kz1) {
    DdaSkip r;
    const int kx = kx1 < w.x.n ? kx1 : w.x.n, ky = ky1 < w.y.n ? ky1 : w.y.n, kz = kz1 < w.z.n ? kz1 : w.z.n;
    const uint32_t ex = w.x.tb + (uint32_t)kx * w.x.c, ey = w.y.tb + (uint32_t)ky * w.y.c, ez = w.z.tb + (uint32_t)kz * w.z.c;
    const bool fx = (ex < ey) && (ex < ez);
    const bool fy = !fx && (ey < ez);
    const bool fz = !fx && !fy;
    const uint32_t eb = fx ? ex : (fy ? ey : ez);
    const int cx_ = iaxis_count_below(w.x, kx, eb);
    const int cy_ = iaxis_count_below(w.y, ky, eb + (fx ? 1u : 0u));
    const int cz_ = iaxis_count_below(w.z, kz, eb + (fz ? 0u : 1u));
    const int nx = fx ? kx + 1 : cx_, ny = fy ? ky + 1 : cy_, nz = fz ? kz + 1 : cz_;
    // the other axes' pending crossings move cnt entries along their chains (still regular); the final axis takes one more step
    w.x.tb += (uint32_t)(fx ? kx : cx_) * w.x.c; w.x.n -= fx ? kx : cx_;
    w.y.tb += (uint32_t)(fy ? ky : cy_) * w.y.c; w.y.n -= fy ? ky : cy_;
    w.z.tb += (uint32_t)(fz ? kz : cz_) * w.z.c; w.z.n -= fz ? kz : cz_;
    if (fx) iaxis_next(w.x, w.dx);
    if (fy) iaxis_next(w.y, w.dy);
    if (fz) iaxis_next(w.z, w.dz);
    w.cx += nx * w.sx; w.cy += ny * w.sy; w.cz += nz * w.sz;
    const bool hit_x = w.cx == w.ox, hit_y = w.cy == w.oy, hit_z = w.cz == w.oz;
    r.t_exit = nfa_u2f(eb);
    r.nx = nx; r.ny = ny; r.nz = nz;
    r.cont = !((fx & hit_x) | (fy & hit_y) | (fz & hit_z));
    return r;
}
