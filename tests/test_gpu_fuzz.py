"""Adversarial traversal inputs, HIP vs oracle, bit-exact: axis-aligned and zero direction
components, origins inside / on the faces of the grid, rays that miss, near > far, tiny and huge
steps, non-cubic grids, all-empty / all-full grids, both count-pass variants (split and
lane-per-ray) and the general walk (cone angle)."""
import numpy as np
import pytest
import torch

import oracle
from gpu_utils import DEV, n, t

pytestmark = pytest.mark.gpu


def _rays(rng, R):
    o = (rng.random((R, 3)) * 4 - 2).astype(np.float32)
    d = rng.standard_normal((R, 3)).astype(np.float32)
    kind = rng.integers(0, 8, R)
    for ax in range(3):                                  # zero one or two direction components
        d[(kind == 1 + ax), ax] = 0.0
    d[kind == 4, 0] = 0.0
    d[kind == 4, 1] = 0.0
    inside = kind == 5                                   # origin inside the base box
    o[inside] = (rng.random((inside.sum(), 3)) * 1.8 - 0.9).astype(np.float32)
    onface = kind == 6                                   # origin exactly on a face, grazing direction
    o[onface, 0] = -1.0
    d[onface, 0] = np.abs(d[onface, 0]) * 1e-3
    snap = kind == 7                                     # origin on voxel corners
    o[snap] = np.round(o[snap] * 8) / 8
    nrm = np.linalg.norm(d, axis=-1, keepdims=True)
    nrm[nrm == 0] = 1
    return o, (d / nrm).astype(np.float32)


def _check(o, d, binaries, aabbs, **kw):
    from nerfacc_amd.grid import traverse_grids

    extra = {k: v for k, v in kw.items() if k in ("near_planes", "far_planes")}
    opts = {k: v for k, v in kw.items() if k not in extra}
    r_iv, r_sm, r_term = oracle.traverse_grids(o, d, binaries, aabbs, **extra, **opts)
    iv, sm, term = traverse_grids(t(o), t(d), t(binaries), t(aabbs), **{k: t(v) for k, v in extra.items()}, **opts)
    assert np.array_equal(n(sm.packed_info), r_sm["packed_info"])
    assert np.array_equal(n(iv.packed_info), r_iv["packed_info"])
    assert np.array_equal(n(sm.ray_indices), r_sm["ray_indices"])
    assert np.array_equal(n(iv.vals), r_iv["vals"]) and np.array_equal(n(sm.vals), r_sm["vals"])
    assert np.array_equal(n(iv.is_left), r_iv["is_left"]) and np.array_equal(n(iv.is_right), r_iv["is_right"])
    live = r_sm["packed_info"][:, 1] > 0
    assert np.array_equal(n(term)[live], r_term[live])
    return int(r_sm["packed_info"][:, 1].sum())


@pytest.mark.parametrize("seed", range(6))
@pytest.mark.parametrize("n_rays", [700, 40000])          # split walk / lane-per-ray walk
def test_fuzz_single_level(seed, n_rays):
    rng = np.random.default_rng(100 + seed)
    res = [(32, 32, 32), (16, 40, 24), (64, 64, 64), (5, 7, 3), (128, 128, 128), (32, 32, 32)][seed]
    occ_p = [0.3, 0.05, 0.02, 0.5, 0.01, 1.0][seed]
    binaries = (rng.random((1,) + res) < occ_p)
    aabbs = np.array([[-1, -1, -1, 1, 1, 1]], np.float32)
    o, d = _rays(rng, n_rays)
    step = [1e-2, 3e-3, 7.3e-3, 0.05, 5e-3, 0.11][seed]
    near = (rng.random(n_rays) * [0.0, 0.5, 0.0, 2.0, 0.005, 0.0][seed]).astype(np.float32)
    far = np.full(n_rays, 1e10, np.float32)
    if seed == 3:
        far = (near + rng.random(n_rays) * 2 - 0.3).astype(np.float32)      # some far < near
    total = _check(o, d, binaries, aabbs, near_planes=near, far_planes=far, step_size=step)
    assert total > 0


def test_fuzz_degenerate_grids():
    rng = np.random.default_rng(5)
    o, d = _rays(rng, 500)
    aabbs = np.array([[-1, -1, -1, 1, 1, 1]], np.float32)
    assert _check(o, d, np.zeros((1, 16, 16, 16), bool), aabbs, step_size=1e-2) == 0          # nothing occupied
    assert _check(o, d, np.ones((1, 16, 16, 16), bool), aabbs, step_size=1e-2) > 0            # everything occupied
    assert _check(o, d, np.ones((1, 1, 1, 1), bool), aabbs, step_size=0.3) > 0                # one voxel
    one = np.zeros((1, 16, 16, 16), bool)
    one[0, 7, 8, 9] = True
    _check(o, d, one, aabbs, step_size=1e-3)                                                   # one occupied voxel, fine lattice


@pytest.mark.parametrize("seed", range(3))
def test_fuzz_multi_level_and_cone(seed):
    rng = np.random.default_rng(200 + seed)
    levels = [2, 4, 3][seed]
    res = [(24, 24, 24), (16, 16, 16), (8, 20, 12)][seed]
    binaries = rng.random((levels,) + res) < 0.25
    aabbs = np.stack([np.array([-1, -1, -1, 1, 1, 1], np.float32) * 2**i for i in range(levels)])
    o, d = _rays(rng, 600)
    o *= 2.5
    assert _check(o, d, binaries, aabbs, step_size=8e-3) > 0
    assert _check(o, d, binaries, aabbs, step_size=4e-3, cone_angle=0.004) > 0
    assert _check(o, d, binaries, aabbs, step_size=-1.0) > 0
    assert _check(o, d, binaries, aabbs, step_size=8e-3, traverse_steps_limit=5) > 0


def test_many_transitions_overflow_paths():
    """checkerboard grid + fine lattice: tens of runs per ray and more boundaries per part than a
    lane's LDS list holds (the split kernel's streaming mode: two more walks, aggregates only).
    NFA_SPLIT_P pins the lanes per ray so that every list size / part length is exercised."""
    import os

    rng = np.random.default_rng(9)
    g = np.indices((32, 32, 32)).sum(0) % 2 == 0
    aabbs = np.array([[-1, -1, -1, 1, 1, 1]], np.float32)
    o, d = _rays(rng, 900)
    assert _check(o, d, g[None], aabbs, step_size=2e-3) > 0
    noise = rng.random((1, 128, 128, 128)) > 0.5           # boundary every other voxel
    box = np.array([[0, 0, 0, 1, 1, 1]], np.float32)
    o2 = (0.5 + 1.5 * d).astype(np.float32)
    d2 = (rng.random((900, 3)).astype(np.float32) - o2)
    d2 /= np.linalg.norm(d2, axis=1, keepdims=True)
    from nerfacc_amd import cuda as C

    def fused(o, d, grid, aabb, step):               # the fused sampling call is what runs the split kernels
        R = o.shape[0]
        near, far = np.zeros(R, np.float32), np.full(R, 1e10, np.float32)
        ri, ts, te, pk = C.sample_occgrid(t(o), t(d), t(grid), t(aabb), t(near), t(far), step, 0.0)
        r_ri, r_ts, r_te, r_pk = oracle.sampling(o, d, grid, aabb, render_step_size=step)
        assert np.array_equal(n(ri), r_ri) and np.array_equal(n(ts), r_ts) and np.array_equal(n(te), r_te)
        assert np.array_equal(n(pk), r_pk)
        return len(r_ri)

    import nerfacc_amd

    for p in (2, 4, 8, 16, 1):
        with nerfacc_amd.options(split_p=p):
            assert fused(o2, d2.astype(np.float32), noise, box, float(np.float32(5e-3 / 3))) > 100000
            assert fused(o, d, g[None], aabbs, 2e-3) > 0


@pytest.mark.randomised
def test_randomised_fuzz_time_boxed():
    """A fresh seed from the clock on every run (printed on failure; NFA_FUZZ_SEED replays it), ~24 s of the long campaigns'
    generators (tests/fuzz_cases.py): one-level grids under every lanes-per-ray form of the count pass, several levels under
    both count passes with and without a cone angle, and the reference-API call with per-voxel mode / step limits /
    over-allocation / ray masks — HIP vs oracle, bit for bit."""
    import os
    import time

    import fuzz_cases as F

    seed = int(os.environ.get("NFA_FUZZ_SEED", time.time_ns() % (1 << 31)))
    budget = float(os.environ.get("NFA_FUZZ_SECONDS", 24.0))
    g = np.random.default_rng(seed)
    bad, total, cases = [], 0, 0
    t0 = time.time()
    while time.time() - t0 < budget and not bad:
        which = cases % 10
        if which == 9:      # round 6: ray counts inside the single-launch sampling call's window (3 072 ... 8 192), the call fused / in three launches
            b, k = F.check_fused(F.fused_single_case(g, ray_counts=(3072, 3105, 4097, 6564, 8191, 8192)), "NFA_FUSED_SAMPLE", F.FUSED_FORMS)
        elif which == 8:      # both emit kernels, with and without a cone angle
            b, k = F.check_fused(F.fused_single_case(g, ray_counts=(1, 7, 64, 500, 3000), cones=(0.0, 0.004, 0.3)), "NFA_EMIT", F.EMIT_FORMS)
        elif which == 7:      # one level, grid image read from LDS / L2 / L2 + staged bitmap
            b, k = F.check_fused(F.fused_single_case(g, ray_counts=(1, 7, 64, 500, 3000, 9000)), None, F.IMAGE_FORMS)
        elif which == 6:      # ... under every lanes-per-ray form of the two-phase kernel
            b, k = F.check_fused(F.fused_levels_case(g, ray_counts=(1, 64, 700, 2048), cones=(0.004, 0.02, 0.1)), "NFA_CONE_P", F.CONE_P_FORMS)
        elif which == 4:    # (renumbered below)
            b, k = F.check_fused(F.fused_single_case(g, ray_counts=(1, 64, 500, 3000), cones=(0.004, 0.05)), "NFA_CONE", F.CONE_FORMS)
        elif which == 5:    # ... and several (a lane per level segment walks, the ray's first lane chains)
            b, k = F.check_fused(F.fused_levels_case(g, ray_counts=(1, 64, 700, 2048), cones=(0.004, 0.02, 0.1)), "NFA_CONE", F.CONE_FORMS)
        elif which == 0:
            b, k = F.check_fused(F.fused_single_case(g, ray_counts=(1, 7, 64, 500, 3000, 9000)), "NFA_SPLIT_P", F.SPLIT_P_FORMS)
        elif which == 1:
            c = F.fused_levels_case(g, ray_counts=(1, 5, 64, 700, 4096))
            b, k = F.check_fused(c, "NFA_SEGMENTS", F.SEGMENT_FORMS)
            b += F.check_fused(c, "NFA_SEG_P", F.SEG_P_FORMS)[0]
        elif which == 2:
            b, k = F.check_fused(F.fused_levels_case(g, ray_counts=(1, 64, 700, 2048), cones=(0.004, 0.02, 0.1)), "NFA_SEGMENTS", F.SEGMENT_FORMS)
        else:
            b, k = F.check_api(F.api_case(g, ray_counts=(3, 100, 2000)))
        bad += b
        total += k
        cases += 1
    assert not bad, f"seed {seed} (NFA_FUZZ_SEED={seed} replays), case {cases - 1}: " + "; ".join(bad[:3])
    assert cases >= 4 and total > 0, (cases, total)
    print(f"fuzz seed {seed}: {cases} scenes, {total} oracle samples, 0 mismatches")
