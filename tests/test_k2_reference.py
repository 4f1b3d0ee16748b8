"""K2 (traverse_grids) exact sample lists pinned by the REFERENCE'S OWN CODE.

tests/golden/k2_reference.npz was produced by /root/reference/nerfacc/cuda/csrc/grid.cu compiled for the
host (oracle/ref_shim/Makefile -> oracle/_ref/) and driven through the reference's Python layer
(tests/golden/make_k2_golden.py).  Here:
  * CPU (-m "not gpu"): the restated oracle must reproduce the fixture bit for bit — counts, pack offsets,
    ray_indices, flags AND the float edges / midpoints / terminate planes;
  * GPU (-m gpu): the HIP kernels, through the public API and the C ABI, must reproduce it too;
  * where oracle/_ref is present (the build container), the reference build itself is re-run against the
    fixture, so a stale fixture cannot go unnoticed.
"""
import importlib
import json
import os
import sys

import numpy as np
import pytest

import k2_cases as K

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="module")
def k2():
    return dict(np.load(os.path.join(GOLD, "k2_reference.npz")))


def _inputs(name, k2):
    c = K.build_case(name, k2)
    assert K.input_digest(c) == str(k2[f"{name}/input_sha"]), f"{name}: regenerated inputs differ from the fixture's"
    return c


def _live(c):
    return c["extra"].get("rays_mask") if c["kw"].get("over_allocate") else None


@pytest.mark.parametrize("name", K.ALL)
def test_oracle_reproduces_reference_k2(name, k2):
    import oracle

    c = _inputs(name, k2)
    iv, sm, term = oracle.traverse_grids(c["rays_o"], c["rays_d"], c["binaries"], c["aabbs"], **c["extra"], **c["kw"])
    n = K.check_against_fixture(name, K.pack_outputs(iv, sm, term, _live(c)), k2)
    assert n > 1000


def test_sensitivity_report_matches_fixture(k2):
    """the FMA-model numbers quoted in DESIGN.md §3.4 come from this file"""
    rep = json.load(open(os.path.join(GOLD, "k2_sensitivity.json")))
    assert sorted(rep) == sorted(K.ALL + ["k1_ray_aabb"])
    for name, r in rep.items():
        if name == "k1_ray_aabb":
            assert r["values_differing_oracle_vs_fma"] == 0 and r["values_differing_oracle_vs_off"] == 0 and r["hits_differing_oracle"] == 0
            continue
        assert r["fixture_build"] == "fma"
        assert r["rays_differing_oracle_vs_fma"] == 0, name


def _check_k1(k2, t_mins, t_maxs, hits, tag=""):
    assert K.sha(np.ascontiguousarray(t_mins, np.float32)) == str(k2[f"k1/sha/t_mins{tag}"])
    assert K.sha(np.ascontiguousarray(t_maxs, np.float32)) == str(k2[f"k1/sha/t_maxs{tag}"])
    assert np.array_equal(np.packbits(np.asarray(hits, bool).ravel()), k2[f"k1/hits_bits{tag}"])


def test_oracle_reproduces_reference_k1(k2):
    """ray_aabb_intersect through the reference's own kernel (grid.cu:284-313, tests/test_grid.py:7-35 generator)"""
    import oracle

    o, d, boxes = k2["k1/rays_o"], k2["k1/rays_d"], k2["k1/aabbs"]
    _check_k1(k2, *oracle.ray_aabb_intersect(o, d, boxes))
    _check_k1(k2, *oracle.ray_aabb_intersect(o, d, boxes, 0.1, 0.7, -1.0), tag="_nf")


@pytest.mark.gpu
def test_hip_reproduces_reference_k1(k2):
    from gpu_utils import n, t
    from nerfacc_amd.grid import ray_aabb_intersect

    o, d, boxes = t(k2["k1/rays_o"]), t(k2["k1/rays_d"]), t(k2["k1/aabbs"])
    _check_k1(k2, *[n(x) for x in ray_aabb_intersect(o, d, boxes)])
    _check_k1(k2, *[n(x) for x in ray_aabb_intersect(o, d, boxes, 0.1, 0.7, -1.0)], tag="_nf")


@pytest.mark.skipif(not (os.path.isdir("/root/reference/nerfacc") and os.path.isdir(os.path.join(ROOT, "oracle", "_ref"))
                         and any(f.startswith("nerfacc_ref_fma") for f in os.listdir(os.path.join(ROOT, "oracle", "_ref")))),
                    reason="oracle/_ref (host build of the reference) only exists in the build container")
@pytest.mark.parametrize("name", ["ref_test_grid", "m1_sphere", "per_voxel", "over_allocate"])
def test_reference_build_reproduces_fixture(name, k2):
    """re-runs the reference's compiled traverse_grids (C++ host wrapper, no Python layer of the reference, which
    cannot be imported next to this repo's `nerfacc` alias) and compares with the committed fixture"""
    import torch

    sys.path.insert(0, os.path.join(ROOT, "oracle", "_ref"))
    try:
        ref = importlib.import_module("nerfacc_ref_fma")
    finally:
        sys.path.pop(0)
    c = _inputs(name, k2)
    tt = lambda a: torch.from_numpy(np.ascontiguousarray(a))
    o, d = tt(c["rays_o"]), tt(c["rays_d"])
    R = o.shape[0]
    tmin, tmax, hits = ref.ray_aabb_intersect(o, d, tt(c["aabbs"]), -float("inf"), float("inf"), float("inf"))
    ts, ti = torch.sort(torch.cat([tmin, tmax], -1), -1)            # grid.py:156-162
    ex, kw = c["extra"], c["kw"]
    mask = tt(ex["rays_mask"]) if "rays_mask" in ex else torch.ones(R, dtype=torch.bool)
    near = tt(ex["near_planes"]) if "near_planes" in ex else torch.zeros(R)
    far = tt(ex["far_planes"]) if "far_planes" in ex else torch.full((R,), float("inf"))
    iv, sm, term = ref.traverse_grids(o, d, mask, tt(c["binaries"]), tt(c["aabbs"]), ts, ti, hits, near, far,
                                      kw.get("step_size", 1e-3), kw.get("cone_angle", 0.0), True, True, True,
                                      kw.get("traverse_steps_limit", -1), kw.get("over_allocate", False))
    m = lambda s, keys: {k: getattr(s, k).numpy() for k in keys}
    out = K.pack_outputs(m(iv, ("vals", "ray_indices", "is_left", "is_right", "chunk_starts", "chunk_cnts")),
                         m(sm, ("vals", "ray_indices", "is_valid", "chunk_starts", "chunk_cnts")), term.numpy(), _live(c))
    K.check_against_fixture(name, out, k2)


@pytest.mark.gpu
@pytest.mark.parametrize("name", K.ALL)
def test_hip_reproduces_reference_k2(name, k2):
    import torch

    from gpu_utils import n, t
    from nerfacc_amd.grid import traverse_grids

    c = _inputs(name, k2)
    extra = {k: t(v) for k, v in c["extra"].items()}
    iv, sm, term = traverse_grids(t(c["rays_o"]), t(c["rays_d"]), t(c["binaries"]), t(c["aabbs"]), **extra, **c["kw"])
    torch.cuda.synchronize()
    as_map = lambda s, flags: dict(vals=n(s.vals), ray_indices=n(s.ray_indices), chunk_starts=n(s.packed_info[:, 0]),
                                   chunk_cnts=n(s.packed_info[:, 1]), **{f: n(getattr(s, f)) for f in flags})
    out = K.pack_outputs(as_map(iv, ("is_left", "is_right")), as_map(sm, ("is_valid",)), n(term), _live(c))
    K.check_against_fixture(name, out, k2)


# Every kernel FORM that can serve a case, forced through the library's options (nerfacc_amd.set_option), against the reference's
# own output — not only against the oracle.  {} = the automatic choice.
_ONE_LEVEL = [{}, {"split_p": 16, "split_l2": 0}, {"split_p": 16, "split_l2": 1}, {"split_p": 8}, {"split_p": 1, "count_l2": 0},
              {"split_p": 1, "count_l2": 1}, {"emit": "rays"}, {"emit": "samples"}, {"emit": "tiles"},
              # boundary-list capacities of the forms that read the grid from L2 (lego_256, m1_noise; ignored where the image fits LDS)
              {"split_p": 16, "split_l2": 1, "split_cap": 16}, {"split_p": 16, "split_l2": 1, "split_cap": 24},
              {"split_p": 16, "split_l2": 1, "split_cap": 32}, {"split_p": 8, "split_cap": 24},
              # round 5: the lane-per-ray walk with and without empty-space macro steps
              {"split_p": 1, "count_l2": 0, "skip": 0}, {"split_p": 1, "count_l2": 1, "skip": 0}, {"split_p": 1, "count_l2": 0, "skip": 1},
              {"split_p": 1, "count_l2": 1, "skip": 1},
              # round 6: the single-launch sampling call switched off (m1_sphere is inside its window)
              {"fused_sample": 0}, {"fused_sample": 2}]
_LEVELS = [{}, {"segments": 0}, {"segments": 1, "seg_p": 8}, {"segments": 1, "seg_p": 32}, {"emit": "rays"}, {"emit": "samples"}, {"emit": "tiles"},
           {"segments": 0, "skip": 0}, {"segments": 0, "skip": 1}]
_CONE_ONE = [{}, {"cone": 0}, {"cone": 1, "emit": "rays"}, {"cone": 1, "emit": "samples"}, {"cone": 0, "emit": "rays"}]
_CONE_LEVELS = _CONE_ONE + [{"cone": 1, "cone_p": p} for p in (8, 16, 32, 64)] + [{"cone_p": 64, "emit": "samples"}]
SAMPLING_FORMS = {
    "m1_sphere": _ONE_LEVEL, "m1_noise": _ONE_LEVEL, "lego_4k": _ONE_LEVEL, "lego_12k": _ONE_LEVEL, "lego_256": _ONE_LEVEL,
    "lego_70k": [{}, {"split_p": 8}, {"split_p": 1, "count_l2": 1}, {"emit": "rays"}, {"emit": "samples"}, {"emit": "tiles"},
                 {"split_p": 1, "count_l2": 1, "skip": 0}, {"split_p": 1, "count_l2": 1, "skip": 1}, {"split_p": 1, "count_l2": 0, "skip": 1}],
    "lego_160k": [{}, {"split_p": 8}, {"split_p": 1, "count_l2": 0}, {"emit": "rays"}, {"emit": "samples"}, {"emit": "tiles"},
                  {"skip": 0}, {"skip": 1}],
    "near_far": _LEVELS, "degenerate": _LEVELS, "two_level_256": _LEVELS, "non_cubic": _LEVELS, "levels4_inside": _LEVELS,
    "cone_angle": _CONE_ONE, "cone_angle_levels": _CONE_LEVELS,
}
_FORM_ID = lambda f: ",".join(f"{k}={v}" for k, v in f.items()) or "auto"
SAMPLING_PARAMS = [pytest.param(name, form, id=f"{name}-{_FORM_ID(form)}") for name, forms in SAMPLING_FORMS.items() for form in forms]


@pytest.mark.gpu
@pytest.mark.parametrize("name,form", SAMPLING_PARAMS)
def test_hip_sampling_reproduces_reference_k2(name, form, k2):
    """OccGridEstimator.sampling (the fused count/emit kernels, not the general fill kernel) against the same
    fixture: ray_indices, t_starts = vals[is_left], t_ends = vals[is_right] (occ_grid.py:166-176) — under every
    form of the count and emit passes that can serve the case (grid.cu:23-28, 196-216 for the cone-angle ones:
    the reference's unbounded-scene setting, examples/train_ngp_nerf_occ.py:48-53)"""
    import torch

    import nerfacc_amd

    with nerfacc_amd.options(**form):
        _sampling_vs_fixture(name, k2)


def _sampling_vs_fixture(name, k2):
    import torch

    from gpu_utils import n, t
    from nerfacc_amd import OccGridEstimator

    c = _inputs(name, k2)
    lvl, res = c["binaries"].shape[0], c["binaries"].shape[1:]
    est = OccGridEstimator(roi_aabb=c["aabbs"][0].tolist(), resolution=list(res), levels=lvl).to("cuda:0")
    assert np.array_equal(n(est.aabbs), c["aabbs"])
    est.binaries = t(c["binaries"])
    kw = c["kw"]
    ex = c["extra"]
    near = t(ex["near_planes"]) if "near_planes" in ex else None
    far = t(ex["far_planes"]) if "far_planes" in ex else None
    # near_plane=0 / far_plane=inf are grid.py's defaults, which the fixture call used
    ri, ts, te = est.sampling(t(c["rays_o"]), t(c["rays_d"]), near_plane=0.0, far_plane=float("inf"),
                              t_min=near, t_max=far, render_step_size=kw["step_size"], cone_angle=kw.get("cone_angle", 0.0))
    torch.cuda.synchronize()
    ri, ts, te = n(ri), n(ts), n(te)
    assert K.sha(ri.astype(np.int64)) == str(k2[f"{name}/sha/sm_ray_indices"])
    cnts = np.bincount(ri, minlength=c["rays_o"].shape[0])
    ref_cnts = k2[f"{name}/cnts/sm_chunk_cnts"] if f"{name}/cnts/sm_chunk_cnts" in k2 else k2[f"{name}/full/sm_chunk_cnts"]
    assert np.array_equal(cnts, ref_cnts)
    # midpoints of the fixture = (t_start + t_end) * 0.5 with the same rounding (grid.cu:252)
    mids = ((te + ts) * np.float32(0.5)).astype(np.float32)
    assert K.sha(mids) == str(k2[f"{name}/sha/sm_vals"])
    # and the edges themselves: vals[is_left] / vals[is_right] of the reference's intervals (occ_grid.py:174-175)
    assert K.sha(ts.astype(np.float32)) == str(k2[f"{name}/sha/t_starts"])
    assert K.sha(te.astype(np.float32)) == str(k2[f"{name}/sha/t_ends"])


# ---------------------------------------------------------------------------------------------------------------------
# rays lying IN a bounding plane of a level: only defined under the GPU's float -> int conversion rule (saturating, NaN -> 0).
# tests/golden/k2_inplane.npz comes from the reference's grid.cu built with that rule (oracle/ref_shim `gpu` build:
# prelude.h re-routes the sources' `int(float)` casts; the x86-rule builds read outside the grid on these rays and crash).
# ---------------------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def k2_inplane():
    return dict(np.load(os.path.join(GOLD, "k2_inplane.npz")))


@pytest.mark.parametrize("name", K.GPU_RULE)
def test_oracle_reproduces_reference_inplane(name, k2_inplane):
    import oracle

    c = _inputs(name, k2_inplane)
    with np.errstate(all="ignore"):
        slab = (c["aabbs"].reshape(1, -1, 2, 3) - c["rays_o"][:, None, None, :]) * (np.float32(1) / c["rays_d"])[:, None, None, :]
    assert np.isnan(slab).any(axis=(1, 2, 3)).mean() > 0.4          # the case is what it claims to be
    iv, sm, term = oracle.traverse_grids(c["rays_o"], c["rays_d"], c["binaries"], c["aabbs"], **c["extra"], **c["kw"])
    assert K.check_against_fixture(name, K.pack_outputs(iv, sm, term, None), k2_inplane) > 50000
    rep = json.load(open(os.path.join(GOLD, "k2_inplane.json")))
    assert rep[name]["fixture_build"] == "gpu" and rep[name]["rays_differing_oracle_vs_gpu_rule"] == 0


@pytest.mark.skipif(not (os.path.isdir("/root/reference/nerfacc") and os.path.isdir(os.path.join(ROOT, "oracle", "_ref"))
                         and any(f.startswith("nerfacc_ref_gpu") for f in os.listdir(os.path.join(ROOT, "oracle", "_ref")))),
                    reason="oracle/_ref (host build of the reference) only exists in the build container")
@pytest.mark.parametrize("name", K.GPU_RULE)
def test_reference_gpu_rule_build_reproduces_inplane_fixture(name, k2_inplane):
    import torch

    sys.path.insert(0, os.path.join(ROOT, "oracle", "_ref"))
    try:
        ref = importlib.import_module("nerfacc_ref_gpu")
    finally:
        sys.path.pop(0)
    c = _inputs(name, k2_inplane)
    tt = lambda a: torch.from_numpy(np.ascontiguousarray(a))
    o, d = tt(c["rays_o"]), tt(c["rays_d"])
    R = o.shape[0]
    tmin, tmax, hits = ref.ray_aabb_intersect(o, d, tt(c["aabbs"]), -float("inf"), float("inf"), float("inf"))
    ts, ti = torch.sort(torch.cat([tmin, tmax], -1), -1)            # grid.py:156-162
    iv, sm, term = ref.traverse_grids(o, d, torch.ones(R, dtype=torch.bool), tt(c["binaries"]), tt(c["aabbs"]), ts, ti, hits,
                                      torch.zeros(R), torch.full((R,), float("inf")), c["kw"]["step_size"], 0.0, True, True, True, -1, False)
    m = lambda s, keys: {k: getattr(s, k).numpy() for k in keys}
    out = K.pack_outputs(m(iv, ("vals", "ray_indices", "is_left", "is_right", "chunk_starts", "chunk_cnts")),
                         m(sm, ("vals", "ray_indices", "is_valid", "chunk_starts", "chunk_cnts")), term.numpy(), None)
    K.check_against_fixture(name, out, k2_inplane)


@pytest.mark.gpu
@pytest.mark.parametrize("name", K.GPU_RULE)
def test_hip_reproduces_reference_inplane(name, k2_inplane):
    """both routes: the reference-API call (general fill kernel) and OccGridEstimator.sampling (segment count pass + emit)"""
    import torch

    from gpu_utils import n, t
    from nerfacc_amd import OccGridEstimator
    from nerfacc_amd.grid import traverse_grids

    c = _inputs(name, k2_inplane)
    iv, sm, term = traverse_grids(t(c["rays_o"]), t(c["rays_d"]), t(c["binaries"]), t(c["aabbs"]), **c["kw"])
    torch.cuda.synchronize()
    as_map = lambda s, flags: dict(vals=n(s.vals), ray_indices=n(s.ray_indices), chunk_starts=n(s.packed_info[:, 0]),
                                   chunk_cnts=n(s.packed_info[:, 1]), **{f: n(getattr(s, f)) for f in flags})
    K.check_against_fixture(name, K.pack_outputs(as_map(iv, ("is_left", "is_right")), as_map(sm, ("is_valid",)), n(term), None), k2_inplane)
    est = OccGridEstimator(roi_aabb=c["aabbs"][0].tolist(), resolution=list(c["binaries"].shape[1:]), levels=c["binaries"].shape[0]).to("cuda:0")
    est.binaries = t(c["binaries"])
    import nerfacc_amd

    for seg in (1, 0):
        with nerfacc_amd.options(segments=seg):
            ri, ts, te = est.sampling(t(c["rays_o"]), t(c["rays_d"]), near_plane=0.0, far_plane=float("inf"),
                                      render_step_size=c["kw"]["step_size"])
        ri, ts, te = n(ri), n(ts), n(te)
        assert K.sha(ri.astype(np.int64)) == str(k2_inplane[f"{name}/sha/sm_ray_indices"])
        assert np.array_equal(np.bincount(ri, minlength=c["rays_o"].shape[0]), k2_inplane[f"{name}/cnts/sm_chunk_cnts"])
        assert K.sha(((te + ts) * np.float32(0.5)).astype(np.float32)) == str(k2_inplane[f"{name}/sha/sm_vals"])
