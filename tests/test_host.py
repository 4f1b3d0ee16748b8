"""Host-side logic on CPU: the C-ABI library loads and exports what include/nerfacc_hip.h
declares; the pure-torch parts of the package (estimator maintenance, batched rendering paths,
twins) reproduce the reference (golden fixtures); native entry points refuse CPU tensors."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# ------------------------------------------------------------------ C ABI
def _header_functions():
    txt = open(os.path.join(ROOT, "include", "nerfacc_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(nfa_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    from nerfacc_amd.cuda._backend import EXPORTED_SYMBOLS, LIB_PATH, load_library

    assert os.path.exists(LIB_PATH), "run `python -m nerfacc_amd.build` (or __graft_entry__.build()) first"
    lib = load_library()
    declared = _header_functions()
    assert len(declared) >= 24
    for name in declared:
        assert hasattr(lib, name), f"{name} is declared in include/nerfacc_hip.h but not exported"
    assert set(EXPORTED_SYMBOLS) == set(declared), set(EXPORTED_SYMBOLS) ^ set(declared)
    assert lib.nfa_version().decode().startswith("nerfacc_hip ")


def test_ctypes_prototypes_are_generated_from_the_header():
    """the ctypes face reads its argument types and struct layouts from include/nerfacc_hip.h (nerfacc_amd/cuda/_cabi.py):
    spot-check the parser against prototypes and a struct whose layout is known"""
    from nerfacc_amd.cuda._cabi import parse_header

    structs, fns = parse_header()
    args = structs["nfa_traverse_args"]
    assert ctypes.sizeof(args) == 296 and args.res.offset == 36 and args.res.size == 12 and args.workspace_bytes.offset == 288
    assert ctypes.sizeof(structs["nfa_ray_segments"]) == 56
    assert fns["nfa_version"] == (ctypes.c_char_p, [])
    assert fns["nfa_packed_grid_words"] == (ctypes.c_int64, [ctypes.c_int32] * 4)
    res, argt = fns["nfa_traverse_fill"]
    assert res is ctypes.c_int and argt[0]._type_ is args and argt[1:] == [ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p]
    assert fns["nfa_transform_stot"][1][2:5] == [ctypes.c_double, ctypes.c_double, ctypes.c_int32]
    assert fns["nfa_reset_options"] == (None, [])


def test_options_table_and_environment_seeding():
    """nfa_set_option / nfa_get_option (include/nerfacc_hip.h): names with or without the NFA_ prefix, values validated, unknown
    names refused, `with options(...)` restores; the environment seeds the table ONCE at load (a later setenv changes nothing)"""
    import subprocess
    import sys

    import nerfacc_amd as na

    names = na.list_options()
    assert {"e", "tile", "split_p", "seg_p", "cone_p", "cone", "split_l2", "count_l2", "emit", "scan_rw", "split_blk", "split_xt",
            "segments", "speculative_emit"} <= set(names) and all(names.values())
    na.reset_options()
    try:
        assert na.get_option("split_p") is None
        na.set_option("NFA_SPLIT_P", 8)
        assert na.get_option("split_p") == 8 and na.get_option("Split_P") == 8
        with na.options(split_p=16, emit="rays"):
            assert na.get_option("split_p") == 16 and na.get_option("emit") == "rays"
        assert na.get_option("split_p") == 8 and na.get_option("emit") is None
        assert {"skip", "emit_rb", "split_cap", "chunk_prefetch"} <= set(names)      # rounds 4-5
        assert {"fused_sample", "fused_vis", "fold_fill"} <= set(names)                                                # round 6
        assert not {"split_thr", "vis_onepass", "vis_chunks"} & set(names)                                             # (pruned in round 6)
        na.set_option("fused_vis", 1); na.set_option("emit_rb", 4)
        assert na.get_option("fused_vis") == 1 and na.get_option("NFA_EMIT_RB") == 4
        na.set_option("fused_vis", None); na.set_option("emit_rb", None)
        for name, value in (("split_p", 3), ("emit", "r"), ("emit", ""), ("tile", 100), ("no_such_option", 1), ("vis_onepass", 1),
                            ("skip", 2), ("split_p", 32), ("fused_vis", 2), ("fused_sample", 3)):
            if value == "":
                na.set_option(name, value)          # "" = auto
                assert na.get_option(name) is None
                continue
            with pytest.raises(ValueError):
                na.set_option(name, value)
        assert na.get_option("split_p") == 8        # a refused value changes nothing
    finally:
        na.reset_options()
    code = ("import os, nerfacc_amd as na; a = na.get_option('emit'), na.get_option('split_p'), na.get_option('speculative_emit');"
            "os.environ['NFA_EMIT'] = 'samples'; os.environ['NFA_SCAN_RW'] = '4';"
            "print(a, na.get_option('emit'), na.get_option('scan_rw'))")
    env = dict(os.environ, NFA_EMIT="rays", NFA_SPLIT_P="5", NFA_NO_SPECULATIVE_EMIT="1", PYTHONPATH=ROOT)
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, check=True).stdout
    assert out.strip() == "('rays', None, 0) rays None", out      # (5 is not a lanes-per-ray value: ignored, as before)


def test_argument_validation_happens_before_any_launch():
    # no GPU here: these calls must fail in validation, with a message, not crash
    from nerfacc_amd.cuda._backend import load_library

    lib = load_library()
    assert lib.nfa_scan_keyed(None, None, None, -1, 0, 1, 0, None) == 1
    assert b"n < 0" in lib.nfa_last_error()
    assert lib.nfa_scan_keyed(None, None, None, 5, 7, 1, 0, None) == 1
    assert b"bad op" in lib.nfa_last_error()
    assert lib.nfa_scan_keyed(None, None, None, 0, 0, 1, 0, None) == 0          # n == 0 is legal and launches nothing
    assert lib.nfa_packed_grid_words(1, 128, 128, 128) == 32768 + 12 + 512 + 512 + 32768 + 32768 // 16      # (+ one nibble per brick: round 5)
    assert lib.nfa_traverse_workspace_bytes(1000) > 0 and lib.nfa_visibility_workspace_bytes(1000) > 1000
    # the filter's workspace holds its bit planes (two slots of 2 E words per chunk of 64 E samples, E <= 4) plus four words per
    # wave tile of the smallest plan, or the n keep bytes of the one-pass form's overflowing tiles — for every n
    for n in (0, 1, 63, 64, 65, 1000, 4097, (1 << 17) - 1, 1 << 17, (1 << 20) + 3, 1 << 24):
        ws = lib.nfa_visibility_workspace_bytes(n)
        m = max(n, 1)
        planes = max(-(-m // (64 * e)) * 2 * 2 * e * 8 for e in (1, 2, 4))
        assert ws % 8 == 0 and ws >= planes + 32 * -(-m // 576) and ws >= m


def test_native_paths_refuse_cpu_tensors():
    import nerfacc_amd as nerfacc

    ri = torch.tensor([0, 0, 1])
    x = torch.rand(3)
    with pytest.raises((RuntimeError, NotImplementedError)):
        nerfacc.pack_info(ri, 2)
    with pytest.raises(RuntimeError):
        nerfacc.render_weight_from_density(x, x + 1, x, ray_indices=ri)
    # ... except where the reference's function is a device-agnostic torch composition: flattened accumulate is
    # `index_add_` on host tensors (reference volrend.py:549-558, 582-584)
    v = torch.rand(3, 2, requires_grad=True)
    out = nerfacc.accumulate_along_rays(x, v, ri, 2)
    assert torch.allclose(out, torch.stack([x[0] * v[0] + x[1] * v[1], x[2] * v[2]]))
    out.sum().backward()
    assert torch.allclose(v.grad, x[:, None].expand(3, 2))
    acc = torch.ones(2, 1)
    from nerfacc_amd.volrend import accumulate_along_rays_
    accumulate_along_rays_(x, None, ri, acc)
    assert torch.allclose(acc[:, 0], 1 + torch.stack([x[0] + x[1], x[2]]))
    with pytest.raises(RuntimeError):
        nerfacc.exclusive_sum(x, indices=ri)
    est = nerfacc.OccGridEstimator([-1.0, -1, -1, 1, 1, 1], 8)
    with pytest.raises(RuntimeError):
        est.sampling(torch.rand(4, 3), torch.rand(4, 3))


def test_public_api_matches_reference_names():
    import nerfacc_amd as nerfacc

    ref = ["__version__", "inclusive_prod", "exclusive_prod", "inclusive_sum", "exclusive_sum", "pack_info",
           "render_visibility_from_alpha", "render_visibility_from_density", "render_weight_from_alpha",
           "render_weight_from_density", "render_transmittance_from_alpha", "render_transmittance_from_density",
           "accumulate_along_rays", "rendering", "importance_sampling", "searchsorted", "RayIntervals", "RaySamples",
           "ray_aabb_intersect", "traverse_grids", "OccGridEstimator", "PropNetEstimator", "distortion"]
    for name in ref:                       # nerfacc/__init__.py:26-56 minus the fVDB-only entries
        assert hasattr(nerfacc, name), name
    from nerfacc_amd import cuda as C

    for name in ["RaySegmentsSpec", "ray_aabb_intersect", "traverse_grids", "inclusive_sum", "exclusive_sum",
                 "inclusive_prod_forward", "inclusive_prod_backward", "exclusive_prod_forward", "exclusive_prod_backward",
                 "is_cub_available", "inclusive_sum_cub", "exclusive_sum_cub", "inclusive_prod_cub_forward",
                 "inclusive_prod_cub_backward", "exclusive_prod_cub_forward", "exclusive_prod_cub_backward",
                 "importance_sampling", "searchsorted"]:          # nerfacc/cuda/__init__.py:19-53
        assert callable(getattr(C, name)), name
    from nerfacc_amd.volrend import accumulate_along_rays_  # noqa: F401  (examples/utils.py:21-25)
    from nerfacc_amd.estimators.prop_net import get_proposal_requires_grad_fn  # noqa: F401


# ------------------------------------------------------------------ pure-torch parts vs the reference (golden)
def test_twins_vs_reference(golden):
    from nerfacc_amd.grid import _enlarge_aabb, _query, _ray_aabb_intersect

    g = golden
    tmin, tmax, hits = _ray_aabb_intersect(torch.from_numpy(g["k1_rays_o"]), torch.from_numpy(g["k1_rays_d"]),
                                           torch.from_numpy(g["k1_aabbs"]))
    assert np.array_equal(hits.numpy(), g["k1_hits"])
    np.testing.assert_allclose(tmin.numpy(), g["k1_tmin"], rtol=1e-6)
    np.testing.assert_allclose(tmax.numpy(), g["k1_tmax"], rtol=1e-6)
    base = torch.tensor([-1.0, -1, -1, 1, 1, 1])
    binaries = torch.from_numpy(np.unpackbits(g["q_binaries"]).astype(bool).reshape(4, 32, 32, 32))
    occ, sel = _query(torch.from_numpy(g["q_pts"]), binaries, base)
    assert np.array_equal(sel.numpy(), g["q_sel"]) and np.array_equal(occ.numpy().astype(bool), g["q_occ"].astype(bool))
    enl = torch.stack([_enlarge_aabb(base, 2**i) for i in range(4)])
    assert np.array_equal(enl.numpy(), g["enlarge"])


def test_batched_paths_vs_reference(golden):
    """[n_rays, n_samples] inputs without ray_indices are plain torch, as in the reference"""
    import nerfacc_amd as nerfacc
    import nerfacc_amd.scan as S

    g = golden
    ts, te = torch.from_numpy(g["v_ts"]), torch.from_numpy(g["v_te"])
    sig = torch.from_numpy(g["v_sig"]).requires_grad_(True)
    w, T, a = nerfacc.render_weight_from_density(ts, te, sig)
    np.testing.assert_allclose(w.detach().numpy(), g["v_w"], atol=1e-6)
    (w * torch.from_numpy(g["v_gw"]) + T * torch.from_numpy(g["v_gT"]) + a * torch.from_numpy(g["v_ga"])).sum().backward()
    np.testing.assert_allclose(sig.grad.numpy(), g["v_gsig"], atol=1e-5, rtol=1e-5)
    rgb = torch.from_numpy(g["r_rgb"])
    col, opa, dep, ex = nerfacc.rendering(ts, te, rgb_sigma_fn=lambda *_: (rgb, sig.detach()), render_bkgd=torch.from_numpy(g["r_bk"]))
    np.testing.assert_allclose(col.numpy(), g["r_col"], atol=1e-6)
    np.testing.assert_allclose(dep.numpy(), g["r_dep"], atol=1e-5)
    al = torch.from_numpy(g["a_al"])
    wa, Ta = nerfacc.render_weight_from_alpha(al)
    np.testing.assert_allclose(wa.numpy(), g["a_w"], atol=1e-6)
    assert np.array_equal(nerfacc.render_visibility_from_alpha(al, early_stop_eps=0.05, alpha_thre=0.35).numpy(), g["a_vis"])
    x = torch.from_numpy(g["s_in"])
    np.testing.assert_allclose(S.inclusive_sum(x).numpy(), g["s_isum"], rtol=1e-6)
    np.testing.assert_allclose(S.exclusive_sum(x).numpy(), g["s_esum"], rtol=1e-6)
    np.testing.assert_allclose(S.inclusive_prod(x * 0.2 + 0.9).numpy(), g["s_iprod"], rtol=1e-5)
    np.testing.assert_allclose(S.exclusive_prod(x * 0.2 + 0.9).numpy(), g["s_eprod"], rtol=1e-5)


def test_pdf_twins_and_schedules_vs_reference(golden):
    from nerfacc_amd.estimators.prop_net import _transform_stot, get_proposal_requires_grad_fn
    from nerfacc_amd.pdf import _sample_from_weighted

    g = golden
    vals, cdfs = torch.from_numpy(g["p_vals"]), torch.from_numpy(g["p_cdfs"])
    for i in range(5):
        e, m = _sample_from_weighted(vals[i:i + 1], cdfs[i:i + 1, 1:] - cdfs[i:i + 1, :-1], 100, False, vals[i].min(), vals[i].max())
        np.testing.assert_allclose(e.numpy()[0], g["p_edges"][i], atol=1e-6)
        np.testing.assert_allclose(m.numpy()[0], g["p_mids"][i], atol=1e-6)
    fn = get_proposal_requires_grad_fn()
    assert np.array_equal(np.array([fn(i) for i in range(3000)]), g["prop_sched"])
    sv = torch.linspace(0, 1, 17)[None]
    np.testing.assert_allclose(_transform_stot("uniform", sv, 0.2, 1000.0).numpy(), g["stot_uni"], rtol=1e-6)
    np.testing.assert_allclose(_transform_stot("lindisp", sv, 0.2, 1000.0).numpy(), g["stot_lin"], rtol=1e-6)
    with pytest.raises(ValueError):
        _transform_stot("nope", sv, 0.2, 1.0)


def test_estimator_maintenance_vs_reference(golden):
    """mark_invisible_cells known answer (tests/test_grid.py:232-233) and a seeded _update run
    that must evolve exactly like the reference's (same torch ops, same RNG call order)"""
    from nerfacc_amd import OccGridEstimator

    base = torch.tensor([-1.0, -1, -1, 1, 1, 1])
    est = OccGridEstimator(roi_aabb=base, resolution=32, levels=4)
    K = torch.tensor([[[100.0, 0, 50.0], [0, 100.0, 50.0], [0, 0, 1]]])
    pose = torch.tensor([[[-1.0, 0.0, 0.0, 0.0], [0.0, 1.0, 0.0, 0.0], [0.0, 0.0, -1.0, 2.5]]])
    est.mark_invisible_cells(K, pose, 100, 100)
    assert (est.occs == -1).sum() == 77660 and (est.occs == 0).sum() == 53412
    assert np.array_equal(np.packbits((est.occs == -1).numpy()), golden["mic_occs_bits"])
    assert list(est.state_dict()) == ["resolution", "aabbs", "occs", "binaries"]
    assert est.binaries.shape == (4, 32, 32, 32) and est.occs.shape == (4 * 32**3,) and est.resolution.dtype == torch.int32

    torch.manual_seed(123)
    est2 = OccGridEstimator(roi_aabb=base, resolution=16, levels=2)
    occ_fn = lambda x: torch.exp(-2.0 * (x**2).sum(-1, keepdim=True)) * 0.05
    for step in (0, 16, 256, 272):
        est2._update(step=step, occ_eval_fn=occ_fn, occ_thre=0.01)
    np.testing.assert_array_equal(est2.occs.numpy(), golden["upd_occs"])
    assert np.array_equal(np.packbits(est2.binaries.numpy().ravel()), golden["upd_bin"])
    with pytest.raises(ValueError):
        OccGridEstimator(roi_aabb=base, contraction_type=1)


def test_data_specs_roundtrip():
    from nerfacc_amd.cuda._backend import RaySegmentsSpec
    from nerfacc_amd.data_specs import RayIntervals, RaySamples

    pk = torch.tensor([[0, 2], [2, 0], [2, 4]])
    iv = RayIntervals(torch.rand(6), packed_info=pk, is_left=torch.ones(6, dtype=torch.bool), is_right=torch.ones(6, dtype=torch.bool))
    spec = iv._to_cpp()
    assert isinstance(spec, RaySegmentsSpec) and spec.is_valid is None
    back = RayIntervals._from_cpp(spec)
    assert torch.equal(back.packed_info, pk) and torch.equal(back.vals, iv.vals)
    sm = RaySamples(torch.rand(6), packed_info=pk, ray_indices=torch.tensor([0, 0, 2, 2, 2, 2]))
    back = RaySamples._from_cpp(sm._to_cpp())          # (the reference's RaySamples._to_cpp raises: data_specs.py:57)
    assert torch.equal(back.packed_info, pk) and torch.equal(back.ray_indices, sm.ray_indices)
    assert RaySamples._from_cpp(RaySamples(torch.rand(3, 4))._to_cpp()).packed_info is None


def test_torch_extension_is_the_default_backend_and_has_the_reference_names():
    """nerfacc_amd/_hip*.so (csrc/torch_ext.cpp) stands where the reference's pybind module `nerfacc.csrc` does
    (nerfacc/cuda/csrc/nerfacc.cpp:126-163): same 21 names, plus this implementation's fused entry points"""
    import os

    from nerfacc_amd import cuda as C
    from nerfacc_amd.cuda import _backend

    if os.environ.get("NERFACC_AMD_BACKEND", "ext") != "ext":
        pytest.skip("ctypes backend forced")
    assert _backend.BACKEND == "ext"
    ext = _backend._C
    assert ext.__name__ == "nerfacc_amd._hip" and ext.version().startswith("nerfacc_hip ")
    for name in C._REFERENCE_NAMES + C._FUSED_NAMES:
        assert hasattr(ext, name), name
        assert hasattr(_backend._CtypesC, name), name            # the fallback offers the same surface
    spec = ext.RaySegmentsSpec()
    assert all(getattr(spec, k) is None for k in ("vals", "is_left", "is_right", "is_valid", "chunk_starts", "chunk_cnts", "ray_indices"))
    with pytest.raises(NotImplementedError):
        ext.opencv_lens_undistortion(1, 2)
    with pytest.raises(RuntimeError):                                  # CHECK_INPUT: host tensors are refused
        ext.ray_aabb_intersect(torch.rand(2, 3), torch.rand(2, 3), torch.rand(1, 6), 0.0, 1.0, -1.0)


def test_propnet_glue_falls_back_to_torch_on_host_tensors():
    """the one-launch forms of PropNetEstimator's per-level glue (s -> t map, edge cdfs, histogram loss) apply to device tensors only:
    on host tensors the same functions run the reference's torch compositions (prop_net.py:99-112, 215-229, 232-256)"""
    from nerfacc_amd.data_specs import RayIntervals
    from nerfacc_amd.estimators.prop_net import _edge_cdfs, _level_cdfs, _pdf_loss, _transform_stot

    torch.manual_seed(0)
    s = torch.rand(5, 9)
    assert torch.equal(_transform_stot("lindisp", s, 0.2, 1e3), 1 / (s * (1 / 1e3) + (1 - s) * (1 / 0.2)))
    assert torch.equal(_transform_stot("uniform", s, 0.2, 1e3), s * 1e3 + (1 - s) * 0.2)
    t_vals = torch.sort(torch.rand(5, 9) * 4, -1)[0]
    sig = torch.rand(5, 8, requires_grad=True)
    cdfs = _level_cdfs(t_vals, sig)
    x = sig * (t_vals[:, 1:] - t_vals[:, :-1])
    trans = torch.exp(-torch.cumsum(torch.cat([torch.zeros_like(x[:, :1]), x[:, :-1]], -1), -1))
    assert torch.allclose(cdfs, _edge_cdfs(trans)) and cdfs.shape == (5, 9) and cdfs.requires_grad
    cdfs.sum().backward()
    assert sig.grad is not None and torch.isfinite(sig.grad).all()
    q, k = RayIntervals(vals=torch.sort(torch.rand(5, 4), -1)[0]), RayIntervals(vals=t_vals / 4)
    cq = torch.sort(torch.rand(5, 4), -1)[0]
    with pytest.raises(Exception):          # searchsorted is native: host tensors are refused loudly, not silently emulated
        _pdf_loss(q, cq, k, cdfs.detach())
