"""The macro step of the voxel DDA (nerfacc_amd/csrc/dda_skip.hpp: many plane crossings at once, used by the traversal kernels to leave
empty regions in one go) must leave exactly the state — voxel, the three pending crossing times bit for bit, the walk-over flag, the exit
time of the last voxel — that the same number of single steps of the reference's DDA (utils_grid.cuh:116-142) leaves.  Host-compiled
check over random and adversarial states.  CPU only."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "native", "dda_skip_check.c")
HDRS = [os.path.join(HERE, "..", "nerfacc_amd", "csrc", h) for h in ("dda_skip.hpp", "lattice.hpp")]


@pytest.fixture(scope="module", params=[0, 3, -3], ids=["rcp-exact", "rcp+3ulp", "rcp-3ulp"])
def lib(request):
    """three builds: the reciprocal that estimates a quotient in the integer walk exact, and off by +-3 ulps (the device's v_rcp_f32 is
    only good to an ulp; the estimate is corrected, so the result must not depend on it)"""
    LIB = os.path.join(HERE, "native", f"libdda_skip_check_{request.param + 3}.so")
    if not os.path.exists(LIB) or os.path.getmtime(LIB) < max(os.path.getmtime(f) for f in [SRC] + HDRS):
        subprocess.check_call(["g++", "-O2", "-x", "c++", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-fvisibility=hidden"]
                              + ([f"-DNFA_RCP_PERTURB={request.param}"] if request.param else []) + [SRC, "-o", LIB])
    L = ctypes.CDLL(LIB)
    for f in (L.check_skip, L.check_walk, L.check_iwalk):
        f.restype = ctypes.c_int64
    return L


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _states(rng, n, kind):
    d = np.exp(rng.uniform(np.log(2e-3), np.log(0.4), (n, 3))).astype(np.float32)
    base = np.exp(rng.uniform(np.log(0.02), np.log(12.0), (n, 1))).astype(np.float32)
    if kind == "binade":                                   # pending crossings just below a power of two: jumps must stop at the edge
        base = (np.float32(2.0) ** rng.integers(-4, 4, (n, 1))).astype(np.float32) * (1 - rng.random((n, 1)).astype(np.float32) * 0.05)
    t = (base + rng.random((n, 3)).astype(np.float32) * d).astype(np.float32)
    if kind == "ties":                                     # equal times / equal deltas between axes, deltas with few mantissa bits
        sel = rng.random(n)
        t[sel < 0.5, 1] = t[sel < 0.5, 0]
        t[sel < 0.25, 2] = t[sel < 0.25, 0]
        d[sel > 0.3, 1] = d[sel > 0.3, 0]
        d[sel > 0.6, 2] = d[sel > 0.6, 0]
        k = rng.integers(8, 20, (n, 3)).astype(np.uint32)
        bits = d.view(np.uint32)
        d = ((bits >> k << k) | (np.uint32(1) << (k - 1))).view(np.float32)      # ...1000 tails: half-way ties in higher binades
        kt = rng.integers(6, 16, (n, 3)).astype(np.uint32)
        t = (t.view(np.uint32) >> kt << kt).view(np.float32)
    if kind == "pow2":
        d = (np.float32(2.0) ** rng.integers(-9, -1, (n, 3))).astype(np.float32)
        t = (np.round(t / d) * d).astype(np.float32)
        t[t <= 0] = d[t <= 0]
    s = rng.choice(np.array([-1, 1], np.int32), (n, 3))
    if kind == "flat":                                     # a zero direction component: step 0, tdist = delta = tmax (utils_grid.cuh:100-103)
        ax = rng.integers(0, 3, n)
        tmax = (base[:, 0] + 3.0).astype(np.float32)
        s[np.arange(n), ax] = 0
        t[np.arange(n), ax] = tmax
        d[np.arange(n), ax] = tmax
    if kind == "weird":                                    # negative / zero / non-finite operands: must degrade to the single step
        sel = rng.integers(0, 6, n)
        t[sel == 0, 0] = -t[sel == 0, 0]
        t[sel == 1, 1] = np.nan
        t[sel == 2, 2] = np.inf
        d[sel == 3, 0] = 0.0
        d[sel == 4, 1] = np.nan
        t[sel == 5] = 0.0
    c = rng.integers(20, 100, (n, 3)).astype(np.int32)
    left = rng.integers(1, 40, (n, 3)).astype(np.int32)    # crossings left until the overflow index
    o = (c + s * left).astype(np.int32)
    o[s == 0] = c[s == 0]
    k1 = rng.integers(0, 16, (n, 3)).astype(np.int32)
    k1 = np.minimum(k1, left - 1)
    k1[s == 0] = 0
    k1[rng.random(n) < 0.1] = 0                            # plain single steps
    return t, d, s.astype(np.int32), c, o, k1.astype(np.int32)


@pytest.mark.parametrize("kind", ["random", "binade", "ties", "pow2", "flat", "weird"])
def test_macro_step_equals_single_steps(lib, kind):
    rng = np.random.default_rng({"random": 0, "binade": 1, "ties": 2, "pow2": 3, "flat": 4, "weird": 5}[kind])
    n = 400000
    t, d, s, c, o, k1 = (np.ascontiguousarray(a) for a in _states(rng, n, kind))
    first = ctypes.c_int64(-1)
    stats = (ctypes.c_int64 * 3)()
    bad = lib.check_skip(ctypes.c_int64(n), _p(t), _p(d), _p(s), _p(c), _p(o), _p(k1), ctypes.byref(first), stats)
    i = first.value
    assert bad == 0, (kind, bad, i, t[i], d[i], s[i], c[i], o[i], k1[i])
    if kind == "random":                                   # the jumps really happen: most calls take all they asked for
        assert stats[1] > 0.5 * stats[2] and stats[0] > 4 * stats[2], list(stats)
    if kind == "weird":
        pass


def test_whole_walks_through_128_cubed(lib):
    """rays through a 128^3 grid set up as utils_grid.cuh:58-114 does (float32 throughout), walked to the end with random macro steps"""
    rng = np.random.default_rng(7)
    n = 60000
    o = rng.standard_normal((n, 3)); o = (4.0 * o / np.linalg.norm(o, axis=1, keepdims=True)).astype(np.float32)
    tgt = (rng.random((n, 3)) * 2.4 - 1.2).astype(np.float32)
    dirs = tgt - o
    dirs = (dirs / np.linalg.norm(dirs, axis=1, keepdims=True)).astype(np.float32)
    dirs[rng.random(n) < 0.05, 0] = 0.0                      # some axis-parallel components
    inv = (np.float32(1.0) / dirs).astype(np.float32)
    lo, hi, res = np.float32(-1.5), np.float32(1.5), 128
    with np.errstate(invalid="ignore", divide="ignore", over="ignore"):
        a = np.where(inv >= 0, (lo - o) * inv, (hi - o) * inv).astype(np.float32)
        b = np.where(inv >= 0, (hi - o) * inv, (lo - o) * inv).astype(np.float32)
        tmin, tmax = np.nanmax(a, axis=1).astype(np.float32), np.nanmin(b, axis=1).astype(np.float32)
        hit = (tmin < tmax) & (tmax > 0) & np.isfinite(tmin) & np.isfinite(tmax)
        o, dirs, inv, tmin, tmax = o[hit], dirs[hit], inv[hit], np.maximum(tmin[hit], 0).astype(np.float32), tmax[hit]
        n = o.shape[0]
        eps = np.float32(1e-6)
        vox = np.float32((hi - lo) / np.float32(res))
        p_in = (dirs * (tmin + eps)[:, None] + o).astype(np.float32)
        p_out = (dirs * (tmax - eps)[:, None] + o).astype(np.float32)
        cur = np.clip((((p_in - lo) / (hi - lo)) * np.float32(res)).astype(np.int32), 0, res - 1)
        fin = np.clip((((p_out - lo) / (hi - lo)) * np.float32(res)).astype(np.int32), 0, res - 1)
        first_plane = cur + (dirs > 0)
        inner = (first_plane.astype(np.float32) * vox - p_in).astype(np.float32)
        t_plane = ((lo + inner) * inv + tmin[:, None]).astype(np.float32)
        sgn = np.sign(dirs).astype(np.int32)
        t = np.where(dirs == 0, tmax[:, None], t_plane).astype(np.float32)
        d = np.where(dirs == 0, tmax[:, None], (vox * inv) * sgn).astype(np.float32)
    ovf = (fin + sgn).astype(np.int32)
    first = ctypes.c_int64(-1)
    stats = (ctypes.c_int64 * 3)()
    arrs = [np.ascontiguousarray(x) for x in (t, d, sgn, cur.astype(np.int32), ovf)]
    bad = lib.check_walk(ctypes.c_int64(n), *[_p(x) for x in arrs], ctypes.c_uint32(3), ctypes.byref(first), stats)
    assert bad == 0, (bad, first.value)
    assert stats[0] > 100 * n and stats[0] > 3 * stats[2], list(stats)      # ~190 voxels per ray, several per macro step
    # the integer-domain form of the same walk (what the kernels run): voxel steps only, then random macro steps
    for mode in (0, 1):
        bad = lib.check_iwalk(ctypes.c_int64(n), *[_p(x) for x in arrs], ctypes.c_uint32(5), ctypes.c_int(mode), ctypes.byref(first), stats)
        assert bad == 0, (mode, bad, first.value)
        assert stats[0] > 100 * n and (mode == 0 or stats[0] > 3 * stats[2]), list(stats)


@pytest.mark.parametrize("kind", ["random", "binade", "ties", "pow2", "flat"])
def test_integer_walk_from_adversarial_states(lib, kind):
    """the integer-domain walk started from the adversarial single-step states (ties between axes, deltas with few mantissa bits,
    pending times just below a power of two, zero direction components), walked to its end with voxel steps and with macro steps"""
    rng = np.random.default_rng({"random": 10, "binade": 11, "ties": 12, "pow2": 13, "flat": 14}[kind])
    n = 60000
    t, d, s, c, o, _ = (np.ascontiguousarray(a) for a in _states(rng, n, kind))
    for mode in (0, 1):
        first = ctypes.c_int64(-1)
        stats = (ctypes.c_int64 * 3)()
        bad = lib.check_iwalk(ctypes.c_int64(n), _p(t), _p(d), _p(s), _p(c), _p(o), ctypes.c_uint32(9), ctypes.c_int(mode), ctypes.byref(first), stats)
        i = first.value
        assert bad == 0, (kind, mode, bad, i, t[i], d[i], s[i], c[i], o[i])
        assert stats[2] > n
