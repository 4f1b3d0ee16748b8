"""The CPU oracle against the reference: golden fixtures (tests/golden, made from
the Python reference) and the hand-computed answers of the reference's own tests.
CPU only."""
import numpy as np
import pytest

import oracle
from conftest import flatten_rows


# ------------------------------------------------------------------ K1
def test_ray_aabb_intersect_vs_reference_twin(golden):
    # reference: tests/test_grid.py:7-35 (ray_aabb_intersect vs _ray_aabb_intersect, torch.allclose)
    tmin, tmax, hits = oracle.ray_aabb_intersect(golden["k1_rays_o"], golden["k1_rays_d"], golden["k1_aabbs"])
    assert (hits == golden["k1_hits"]).all()
    np.testing.assert_allclose(tmin, golden["k1_tmin"], rtol=1e-5, atol=1e-8)
    np.testing.assert_allclose(tmax, golden["k1_tmax"], rtol=1e-5, atol=1e-8)


# ------------------------------------------------------------------ volrend known answers
RI = np.array([0, 2, 2, 2, 2], np.int64)


def test_known_weights_and_grads():
    # reference: tests/test_rendering.py:117-193
    sig = np.array([0.4, 0.8, 0.1, 0.8, 0.1], np.float32)
    ts = np.random.default_rng(0).random(5).astype(np.float32)
    te = ts + 1.0
    w, T, a = oracle.render_weight_from_density(ts, te, sig, RI)
    np.testing.assert_allclose(w, [0.3297, 0.5507, 0.0428, 0.2239, 0.0174], atol=1e-4)
    g = oracle.render_weight_from_density_bwd(ts, te, sig, RI, g_w=np.ones(5, np.float32))
    np.testing.assert_allclose(g, [0.6703, 0.1653, 0.1653, 0.1653, 0.1653], atol=1e-4)
    # alpha route gives the same weights (tests/test_rendering.py:61-83)
    w2, _ = oracle.render_weight_from_alpha(a, RI)
    np.testing.assert_allclose(w2, w, atol=1e-6)


def test_known_visibility_and_alpha_weights():
    # reference: tests/test_rendering.py:11-57
    al = np.array([0.4, 0.3, 0.8, 0.8, 0.5], np.float32)
    w, T = oracle.render_weight_from_alpha(al, RI)
    np.testing.assert_allclose(w, [0.4, 0.3, 0.56, 0.112, 0.014], atol=1e-6)
    assert oracle.visibility(T, al, 0.03, 0.0).tolist() == [True, True, True, True, False]
    assert oracle.visibility(T, al, 0.05, 0.35).tolist() == [True, False, True, True, False]


def test_known_pack_info_and_accumulate():
    # reference: tests/test_pack.py:11-18, tests/test_rendering.py:87-106
    assert oracle.pack_info(RI, 3).tolist() == [[0, 1], [1, 0], [1, 4]]
    assert oracle.pack_info([0, 0, 1, 1, 1, 2, 2, 2, 2], 3).tolist() == [[0, 2], [2, 3], [5, 4]]
    w = np.array([0.4, 0.3, 0.8, 0.8, 0.5], np.float32)
    v = np.random.default_rng(1).random((5, 2)).astype(np.float32)
    out = oracle.accumulate_along_rays(w, v, RI, 3)
    np.testing.assert_allclose(out[0], w[0] * v[0], rtol=1e-6)
    assert (out[1] == 0).all()
    np.testing.assert_allclose(out[2], (w[1:, None] * v[1:]).sum(0), rtol=1e-6)


def test_docstring_examples():
    # reference: nerfacc/scan.py:41-44, 105-108, 175-178, 240-243; volrend.py:256-263, 361-369, 475-485
    x = np.arange(1, 10, dtype=np.float32)
    pk = np.array([[0, 2], [2, 3], [5, 4]], np.int64)
    idx = np.array([0, 0, 1, 1, 1, 2, 2, 2, 2], np.int64)
    want = {
        ("sum", True): [1, 3, 3, 7, 12, 6, 13, 21, 30],
        ("sum", False): [0, 1, 0, 3, 7, 0, 6, 13, 21],
        ("prod", True): [1, 2, 3, 12, 60, 6, 42, 336, 3024],
        ("prod", False): [1, 1, 1, 3, 12, 1, 6, 42, 336],
    }
    for (op, inc), ref in want.items():
        assert oracle.scan_packed(x, pk, op, inc).tolist() == ref
        assert oracle.scan_keyed(x, idx, op, inc).tolist() == ref
    ts = np.arange(7, dtype=np.float32)
    sig = np.array([0.4, 0.8, 0.1, 0.8, 0.1, 0.0, 0.9], np.float32)
    ri = np.array([0, 0, 0, 1, 1, 2, 2], np.int64)
    w, T, a = oracle.render_weight_from_density(ts, ts + 1, sig, ri)
    np.testing.assert_allclose(T, [1.00, 0.67, 0.30, 1.00, 0.45, 1.00, 1.00], atol=5e-3)
    np.testing.assert_allclose(a, [0.33, 0.55, 0.095, 0.55, 0.095, 0.00, 0.59], atol=5e-3)
    np.testing.assert_allclose(w, [0.33, 0.37, 0.03, 0.55, 0.04, 0.00, 0.59], atol=5e-3)
    assert oracle.visibility(T, a, 0.3, 0.2).tolist() == [True, True, False, True, False, False, True]


# ------------------------------------------------------------------ golden: volrend / rendering / scans / pdf
def test_weights_and_grads_vs_reference_batched(golden):
    g = golden
    ts, ri, _ = flatten_rows(g["v_ts"])
    te, sig = g["v_te"].ravel(), g["v_sig"].ravel()
    w, T, a = oracle.render_weight_from_density(ts, te, sig, ri)
    np.testing.assert_allclose(w, g["v_w"].ravel(), atol=1e-5)
    np.testing.assert_allclose(T, g["v_T"].ravel(), atol=1e-5)
    np.testing.assert_allclose(a, g["v_a"].ravel(), atol=1e-6)
    gs = oracle.render_weight_from_density_bwd(ts, te, sig, ri, g["v_gw"].ravel(), g["v_gT"].ravel(), g["v_ga"].ravel())
    np.testing.assert_allclose(gs, g["v_gsig"].ravel(), atol=2e-5, rtol=1e-4)
    wa, Ta = oracle.render_weight_from_alpha(g["a_al"].ravel(), ri)
    np.testing.assert_allclose(wa, g["a_w"].ravel(), atol=1e-6)
    np.testing.assert_allclose(Ta, g["a_T"].ravel(), atol=1e-6)
    vis = oracle.visibility(Ta, g["a_al"].ravel(), 0.05, 0.35)
    assert (vis == g["a_vis"].ravel()).mean() > 0.999  # threshold ties may flip 1 ulp cases
    visd = oracle.visibility(T, a, 1e-2, 0.01)
    assert (visd == g["a_visd"].ravel()).mean() > 0.999


def test_rendering_vs_reference_batched(golden):
    g = golden
    ts, ri, _ = flatten_rows(g["v_ts"])
    R = g["v_ts"].shape[0]
    col, opa, dep, ex = oracle.rendering(ts, g["v_te"].ravel(), ri, R, g["v_sig"].ravel(),
                                         g["r_rgb"].reshape(-1, 3), g["r_bk"])
    np.testing.assert_allclose(col, g["r_col"], atol=1e-5)
    np.testing.assert_allclose(opa, g["r_opa"], atol=1e-5)
    np.testing.assert_allclose(dep, g["r_dep"], atol=2e-5)


def test_accumulate_vs_reference(golden):
    g = golden
    np.testing.assert_allclose(oracle.accumulate_along_rays(g["acc_w"], g["acc_v"], g["acc_idx"], 40), g["acc_out3"], atol=1e-5)
    np.testing.assert_allclose(oracle.accumulate_along_rays(g["acc_w"], None, g["acc_idx"], 40), g["acc_out1"], atol=1e-5)


@pytest.mark.parametrize("name,op,inc", [("isum", "sum", True), ("esum", "sum", False), ("iprod", "prod", True), ("eprod", "prod", False)])
def test_scans_vs_torch_cumsum(golden, name, op, inc):
    # reference: tests/test_scan.py:7-172 (seed 42, [5,1000]); its own tolerance for exclusive_sum is 3e-4
    x = golden["s_in"] if op == "sum" else golden["s_in"] * np.float32(0.2) + np.float32(0.9)
    flat, ri, pk = flatten_rows(x)
    ref = golden[f"s_{name}"].ravel()
    for out in (oracle.scan_packed(flat, pk, op, inc), oracle.scan_keyed(flat, ri, op, inc)):
        np.testing.assert_allclose(out, ref, rtol=2e-5, atol=3e-4)
    gref = golden[f"s_{name}_grad"].ravel()
    ones = np.ones_like(flat)
    if op == "sum":
        grad = oracle.scan_keyed(ones, ri, "sum", inc, reverse=True)
        grad2 = oracle.scan_packed(ones, pk, "sum", inc, reverse=True)
        np.testing.assert_allclose(grad2, grad)
    else:
        grad = oracle.prod_backward(flat, oracle.scan_keyed(flat, ri, "prod", inc), ones, ri, inc)
    np.testing.assert_allclose(grad, gref, rtol=2e-4, atol=1e-3)


def test_pdf_vs_reference(golden):
    g = golden
    # reference: tests/test_pdf.py:65-94 (atol 1e-4), :45-62
    edges, mids = oracle.importance_sampling(g["p_vals"], g["p_cdfs"], 100)
    np.testing.assert_allclose(mids, g["p_mids"], atol=1e-4)
    np.testing.assert_allclose(edges, g["p_edges"], atol=1e-4)
    _, right = oracle.searchsorted(g["p_vals"], g["p_q"])
    assert (right == g["p_ss"]).all()
    # docstring example pdf.py:108-120
    e, m = oracle.importance_sampling([[0.0, 0.5, 1.0]], [[0.0, 0.5, 1.0]], 2)
    e, m = oracle.importance_sampling(np.array([[0.0, 1.0, 2.0]]), np.array([[0.0, 0.5, 1.0]]), 2)
    np.testing.assert_allclose(e, [[0.0, 1.0, 2.0]])
    np.testing.assert_allclose(m, [[0.5, 1.5]])


# ------------------------------------------------------------------ K2 properties (no reference fixture exists)
def _rand_scene(seed, n_rays=10, levels=4, res=32):
    rng = np.random.default_rng(seed)
    o = rng.standard_normal((n_rays, 3)).astype(np.float32)
    d = rng.standard_normal((n_rays, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=-1, keepdims=True)
    base = np.array([-1, -1, -1, 1, 1, 1], np.float32)
    c, e = (base[:3] + base[3:]) / 2, (base[3:] - base[:3]) / 2
    aabbs = np.stack([np.concatenate([c - e * 2**i, c + e * 2**i]) for i in range(levels)]).astype(np.float32)
    binaries = rng.random((levels, res, res, res)) > 0.5
    return o, d, aabbs, binaries, base


def _query_np(x, data, base):
    """numpy twin of nerfacc/grid.py:201-237 (_query); golden-checked below."""
    mn, mx = base[:3], base[3:]
    xn = (x - mn) / (mx - mn)
    maxval = np.maximum(np.abs(xn - 0.5).max(-1), 0.1)
    mip = np.maximum(np.frexp(maxval)[1].astype(np.int64) + 1, 0)
    sel = mip < data.shape[0]
    xu = (xn - 0.5) / (2.0**mip)[:, None] + 0.5
    res = np.array(data.shape[1:])
    ix = np.minimum((xu * res).astype(np.int64), res - 1)
    mip = np.minimum(mip, data.shape[0] - 1)
    return data[mip, ix[:, 0], ix[:, 1], ix[:, 2]] * sel, sel


def test_query_twin_vs_reference(golden):
    binaries = np.unpackbits(golden["q_binaries"]).astype(bool).reshape(4, 32, 32, 32)
    occ, sel = _query_np(golden["q_pts"].astype(np.float32), binaries, np.array([-1, -1, -1, 1, 1, 1], np.float32))
    assert (sel == golden["q_sel"]).all()
    assert (occ.astype(bool) == golden["q_occ"].astype(bool)).all()


@pytest.mark.parametrize("seed", [42, 1, 2])
def test_traverse_samples_lie_in_occupied_cells(seed):
    # reference: tests/test_grid.py:38-68
    o, d, aabbs, binaries, base = _rand_scene(seed, n_rays=32)
    iv, sm, _ = oracle.traverse_grids(o, d, binaries, aabbs)
    ts, te = iv["vals"][iv["is_left"]], iv["vals"][iv["is_right"]]
    ri = sm["ray_indices"]
    assert ts.shape == te.shape == ri.shape and ri.shape[0] > 100
    np.testing.assert_allclose(sm["vals"], (ts + te) * np.float32(0.5), rtol=0, atol=0)
    pos = o[ri] + d[ri] * ((ts + te)[:, None] / 2.0)
    occ, sel = _query_np(pos, binaries, base)
    assert sel.all()
    if seed == 42:  # the reference's own case is strict
        assert occ.all()
    # a midpoint that lands within float error of a cell face may be attributed to either
    # neighbour (the DDA decides by t, _query by position): allow those, nothing else
    assert occ.mean() > 0.9999
    # packing invariants: sorted by ray, offsets = exclusive sum of counts
    assert (np.diff(ri) >= 0).all()
    assert (oracle.pack_info(ri, o.shape[0]) == sm["packed_info"]).all()
    assert (iv["packed_info"][:, 0] == np.cumsum(iv["packed_info"][:, 1]) - iv["packed_info"][:, 1]).all()
    assert iv["is_left"].sum() == iv["is_right"].sum() == ri.shape[0]


def test_traverse_near_far_and_test_mode():
    # reference: tests/test_grid.py:134-160 and :71-131
    o = np.array([[-1.0, 0.0, 0.0]], np.float32)
    d = np.array([[1.0, 0.01, 0.01]], np.float32)
    d /= np.linalg.norm(d)
    iv, sm, _ = oracle.traverse_grids(o, d, np.ones((1, 1, 1, 1), bool), np.array([[0, 0, 0, 1, 1, 1]], np.float32),
                                      near_planes=[1.2], far_planes=[1.5], step_size=0.05)
    assert iv["vals"].size > 0
    assert (iv["vals"] >= 1.2 - 0.025).all() and (iv["vals"] <= 1.5 + 0.025).all()

    o, d, aabbs, binaries, _ = _rand_scene(42)
    iv, sm, _ = oracle.traverse_grids(o, d, binaries, aabbs)
    ts, te = iv["vals"][iv["is_left"]], iv["vals"][iv["is_right"]]
    acc_s = oracle.accumulate_along_rays(ts, None, sm["ray_indices"], 10)
    acc_e = oracle.accumulate_along_rays(te, None, sm["ray_indices"], 10)
    a_s, a_e, near, mask = 0.0, 0.0, None, None
    for _ in range(2):
        iv2, sm2, near = oracle.traverse_grids(o, d, binaries, aabbs, near_planes=near, traverse_steps_limit=4000,
                                               over_allocate=True, rays_mask=mask)
        mask = sm2["packed_info"][:, 1] == 4000
        ri2 = sm2["ray_indices"][sm2["is_valid"]]
        a_s = a_s + oracle.accumulate_along_rays(iv2["vals"][iv2["is_left"]], None, ri2, 10)
        a_e = a_e + oracle.accumulate_along_rays(iv2["vals"][iv2["is_right"]], None, ri2, 10)
    assert (~mask).all()
    np.testing.assert_allclose(a_s, acc_s, atol=1e-1)
    np.testing.assert_allclose(a_e, acc_e, atol=1e-1)


def test_sampling_min_max_distances():
    # reference: tests/test_grid.py:163-204
    rng = np.random.default_rng(42)
    n = 64
    o = (rng.random((n, 3)) * 2 - 1).astype(np.float32)
    d = rng.random((n, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=-1, keepdims=True)
    _, _, aabbs, binaries, _ = _rand_scene(42)
    tmin = rng.random(n).astype(np.float32)
    tmax = tmin + rng.random(n).astype(np.float32)
    ri, ts, te, _ = oracle.sampling(o, d, binaries, aabbs, 0.15, 0.85, tmin, tmax, 0.01)
    assert ri.size > 0
    assert (ts >= tmin[ri] - 0.005).all() and (te <= tmax[ri] + 0.005).all()


def test_grid_maintenance_replays_reference_update(golden):
    """oracle.grid_* driven with the reference's RNG calls (torch CPU generator, seed 123, the
    call order of occ_grid.py:345-404) must reproduce the reference's `_update` evolution that
    tests/golden/make_golden.py recorded from nerfacc itself."""
    import torch

    torch.manual_seed(123)
    res, levels = 16, 2
    cells = res**3
    base = np.array([-1, -1, -1, 1, 1, 1], np.float32)
    ctr, half = (base[:3] + base[3:]) / 2, (base[3:] - base[:3]) / 2
    aabbs = np.stack([np.concatenate([ctr - half * 2**i, ctr + half * 2**i]) for i in range(levels)]).astype(np.float32)
    occs = np.zeros(levels * cells, np.float32)
    binaries = np.zeros(levels * cells, bool)
    occ_fn = lambda x: (np.exp(np.float32(-2.0) * (x.astype(np.float32) ** 2).sum(-1, dtype=np.float32)) * np.float32(0.05)).astype(np.float32)
    for step in (0, 16, 256, 272):
        lvl_ids = []
        if step < 256:                                                    # occ_grid.py:372-374 via _get_all_cells
            for lvl in range(levels):
                lvl_ids.append(np.nonzero(occs[lvl * cells:(lvl + 1) * cells] >= 0)[0])
        else:                                                             # :345-364
            n = cells // 4
            for lvl in range(levels):
                uni = torch.randint(cells, (n,)).numpy()
                uni = uni[occs[lvl * cells + uni] >= 0]
                occd = np.nonzero(binaries[lvl * cells:(lvl + 1) * cells])[0]
                if n < len(occd):
                    occd = occd[torch.randint(len(occd), (n,)).numpy()]
                lvl_ids.append(np.concatenate([uni, occd]))
        for lvl, ids in enumerate(lvl_ids):
            jitter = torch.rand((len(ids), 3)).numpy()
            pts = oracle.grid_cell_points(ids, jitter, (res, res, res), aabbs[lvl])
            # the reference evaluates occ_fn in torch; exp may differ from numpy's in the last bit
            occ = (torch.exp(-2.0 * (torch.from_numpy(pts) ** 2).sum(-1)) * 0.05).numpy()
            occs[lvl * cells:(lvl + 1) * cells] = oracle.grid_ema_update(occs[lvl * cells:(lvl + 1) * cells], ids, occ, 0.95)
        binaries, _ = oracle.grid_threshold(occs, 0.01)
    np.testing.assert_array_equal(occs, golden["upd_occs"])
    assert np.array_equal(np.packbits(binaries), golden["upd_bin"])


def test_threaded_oracle_equals_scalar_oracle():
    """orc_set_threads only changes who does the work (bench.py's all-cores CPU baseline)"""
    import k2_cases as K

    c = K.build_case("lego_4k")
    a = oracle.traverse_grids(c["rays_o"], c["rays_d"], c["binaries"], c["aabbs"], **c["kw"])
    rng = np.random.default_rng(0)
    iv, sm, _ = a
    ts, te, ri = iv["vals"][iv["is_left"]], iv["vals"][iv["is_right"]], sm["ray_indices"]
    sig = (rng.random(ts.shape[0]) * 20).astype(np.float32)
    rgb = rng.random((ts.shape[0], 3)).astype(np.float32)
    gw = rng.standard_normal(ts.shape[0]).astype(np.float32)
    one = (oracle.rendering(ts, te, ri, 4096, sig, rgb, np.ones(3, np.float32)),
           oracle.render_weight_from_density_bwd(ts, te, sig, ri, g_w=gw))
    try:
        assert oracle.set_threads(4) == 4
        b = oracle.traverse_grids(c["rays_o"], c["rays_d"], c["binaries"], c["aabbs"], **c["kw"])
        many = (oracle.rendering(ts, te, ri, 4096, sig, rgb, np.ones(3, np.float32)),
                oracle.render_weight_from_density_bwd(ts, te, sig, ri, g_w=gw))
    finally:
        oracle.set_threads(1)
    for x, y in zip(a[:2], b[:2]):
        for k in x:
            assert np.array_equal(x[k], y[k]), k
    for x, y in zip(one[0][:3], many[0][:3]):
        assert np.array_equal(x, y)
    assert np.array_equal(one[0][3]["weights"], many[0][3]["weights"]) and np.array_equal(one[1], many[1])


def test_pure_torch_cpu_composition_matches_oracle():
    """the pure-PyTorch CPU path timed by bench.py (oracle/torch_cpu.py) against the C oracle"""
    import torch

    import k2_cases as K
    from oracle import torch_cpu

    c = K.build_case("m1_sphere")
    iv, sm, _ = oracle.traverse_grids(c["rays_o"], c["rays_d"], c["binaries"], c["aabbs"], **c["kw"])
    ts, te, ri, pk = iv["vals"][iv["is_left"]], iv["vals"][iv["is_right"]], sm["ray_indices"], sm["packed_info"]
    rng = np.random.default_rng(1)
    sig = (rng.random(ts.shape[0]) * 20).astype(np.float32)
    rgb = rng.random((ts.shape[0], 3)).astype(np.float32)
    w, T, a = oracle.render_weight_from_density(ts, te, sig, ri)
    tt = torch.from_numpy
    (p_ts, p_te, p_sig), mask = torch_cpu.pad_rays(tt(pk), tt(ts), tt(te), tt(sig))
    wb, Tb, ab = torch_cpu.weights_batched(p_ts, p_te, p_sig)
    assert np.allclose(wb[mask].numpy(), w, atol=1e-5) and np.allclose(Tb[mask].numpy(), T, atol=1e-5)
    wf, Tf, af = torch_cpu.weights_flat(tt(ts), tt(te), tt(sig), tt(ri), tt(pk))
    assert np.allclose(wf.numpy(), w, atol=1e-5) and np.allclose(af.numpy(), a, atol=1e-6)
    col, opa, dep, _ = oracle.rendering(ts, te, ri, 4096, sig, rgb, np.ones(3, np.float32))
    c2, o2, d2, _ = torch_cpu.rendering_flat(tt(ts), tt(te), tt(sig), tt(rgb), tt(ri), tt(pk), 4096, torch.ones(3))
    assert np.allclose(c2.numpy(), col, atol=1e-5) and np.allclose(o2.numpy(), opa, atol=1e-5)
    assert np.allclose(d2.numpy(), dep, atol=1e-4)


def test_threaded_glue_equals_mask_gathers():
    import k2_cases as K

    c = K.build_case("lego_4k")
    R = c["rays_o"].shape[0]
    near, far = np.zeros(R, np.float32), np.full(R, 1e10, np.float32)
    iv, sm, _ = oracle.traverse_grids(c["rays_o"], c["rays_d"], c["binaries"], c["aabbs"], near, far, 5e-3)
    try:
        oracle.set_threads(3)
        ri, ts, te, pk = oracle.sample_occgrid(c["rays_o"], c["rays_d"], c["binaries"], c["aabbs"], near, far, 5e-3)
        keep = np.random.default_rng(0).random(ri.shape[0]) < 0.6
        ri2, ts2, te2, pk2 = oracle.compact(keep, ri, ts, te, pk)
    finally:
        oracle.set_threads(1)
    assert np.array_equal(ri, sm["ray_indices"]) and np.array_equal(pk, sm["packed_info"])
    assert np.array_equal(ts, iv["vals"][iv["is_left"]]) and np.array_equal(te, iv["vals"][iv["is_right"]])
    assert np.array_equal(ri2, ri[keep]) and np.array_equal(ts2, ts[keep]) and np.array_equal(te2, te[keep])
    assert np.array_equal(pk2, oracle.pack_info(ri2, R))
