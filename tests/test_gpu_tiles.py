"""The tiled streaming kernels (common.hpp: walk_rays_fwd / walk_rays_bwd) under every launch plan: 1 / 2 / 4 elements
per lane, one-chunk and multi-chunk tiles, unaligned views (fall back to one element per lane).  The default plans only
reach E = 4 at N >= 2^22; the `e` / `tile` options (nerfacc_amd.set_option) force the others onto the small oracle-checked cases."""
import numpy as np
import pytest
import torch

import test_gpu_scan
import test_gpu_volrend
from gpu_utils import DEV, n, ragged, t

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("e,tile", [(1, 64), (1, 192), (2, 128), (2, 384), (4, 256), (4, 768)])
def test_every_plan_vs_oracle(force_options, e, tile):
    force_options(e=e, tile=tile)
    for args in ((7, 5, 1), (500, 90, 2), (3000, 700, 3), (3, 5000, 4), (20000, 12, 5)):
        test_gpu_volrend.test_ragged_vs_oracle_fwd_bwd(*args)
    for args in ((9, 3, 1), (700, 150, 2), (5, 3000, 3), (20000, 9, 4)):
        test_gpu_scan.test_ragged_vs_oracle(*args)
    for r in (50, 5000):
        test_gpu_volrend.test_visibility_compact_all_prefix_modes(r)
    test_gpu_volrend.test_accumulate_unsorted_ray_indices_like_index_add()
    test_gpu_volrend.test_accumulate_backward_and_inplace()


@pytest.mark.parametrize("e", [2, 4])
def test_large_n_properties_vectorised(force_options, e):
    force_options(e=e)
    test_gpu_volrend.test_large_n_properties()


def test_unaligned_views_equal_aligned(force_options):
    """views whose storage offset is not a multiple of 16 bytes take the one-element plan: same results"""
    from nerfacc_amd import cuda as C

    force_options(e=4)
    rng = np.random.default_rng(0)
    ri_, _ = ragged(rng, 4000, 60)
    N = ri_.shape[0]
    base = {k: torch.rand(N + 1, device=DEV) for k in ("ts", "dt", "sig")}
    ts, dt, sig = (base[k][1:] for k in ("ts", "dt", "sig"))           # 4-byte offset
    assert ts.data_ptr() % 16 != 0
    ri = t(ri_)
    te = ts + dt * 0.01
    te_al, ts_al, sig_al = te.clone(), ts.clone(), (sig * 30).clone()
    sig_un = torch.empty(N + 1, device=DEV)[1:]
    sig_un.copy_(sig_al)
    te_un = torch.empty(N + 1, device=DEV)[1:]
    te_un.copy_(te_al)
    a = C.render_weight_from_density_fwd(ri, ts_al, te_al, sig_al, None)
    b = C.render_weight_from_density_fwd(ri, ts, te_un, sig_un, None)
    for x, y in zip(a, b):
        assert torch.allclose(x, y, atol=1e-6)
    rgb = torch.rand(N, 3, device=DEV)
    ra = C.rendering_fwd(ri, ts_al, te_al, sig_al, rgb, 4000, None, True)
    rb = C.rendering_fwd(ri, ts, te_un, sig_un, rgb, 4000, None, True)
    for x, y in zip(ra, rb):
        assert torch.allclose(x, y, atol=1e-5)
    assert torch.allclose(C.exclusive_sum_cub(ri, sig_al, False), C.exclusive_sum_cub(ri, sig_un, False), atol=1e-3, rtol=1e-5)
    assert np.isfinite(n(rb[0])).all()


@pytest.mark.parametrize("speculative", [True, False])
def test_sampling_with_and_without_the_speculative_emit(force_options, speculative):
    """sample_occgrid launches its emit pass before the read-back of the totals, into outputs sized from the previous call;
    too small a guess must fall back to exactly sized outputs, and both orders must reproduce the reference's sample lists"""
    import os

    import test_k2_reference as T

    if not speculative:
        force_options(speculative_emit=0)
    k2 = dict(np.load(os.path.join(T.GOLD, "k2_reference.npz")))
    # small -> large -> small: the guess is too small for the second case and generous for the third
    for name in ["degenerate", "lego_4k", "lego_70k", "m1_sphere", "near_far", "lego_4k"]:
        T._sampling_vs_fixture(name, k2)


def test_packed_scans_over_arbitrary_row_tables():
    """scan_packed takes ANY (start, count) table, as the reference does: rows with gaps between them, rows in reverse
    order, a table that covers only part of the input — every row scanned on its own"""
    from nerfacc_amd import cuda as C

    rng = np.random.default_rng(7)
    ri_, pk_ = ragged(rng, 3000, 40)
    N = ri_.shape[0]
    x_ = (rng.random(N) * 0.4 + 0.8).astype(np.float32)
    x = t(x_)

    def rows_reference(starts, cnts, inclusive, reverse):
        out = x_.copy()                                   # elements outside every row are whatever the kernel leaves: compare rows only
        for s0, c in zip(starts, cnts):
            seg = x_[s0:s0 + c]
            if reverse:
                seg = seg[::-1]
            inc = np.cumsum(seg.astype(np.float64)).astype(np.float32)
            r = inc if inclusive else np.concatenate([[0.0], inc[:-1]]).astype(np.float32)
            out[s0:s0 + c] = r[::-1] if reverse else r
        return out

    tables = {
        "gap": (pk_[:, 0] + (np.arange(len(pk_)) >= 1000) * 0, np.where(np.arange(len(pk_)) == 500, np.maximum(pk_[:, 1] - 1, 0), pk_[:, 1])),
        "reordered": (pk_[::-1, 0].copy(), pk_[::-1, 1].copy()),
        "short": (pk_[:2000, 0].copy(), pk_[:2000, 1].copy()),
    }
    for name, (st_, ct_) in tables.items():
        for inclusive in (True, False):
            for reverse in (False, True):
                got = n(C._lazy("_packed")(t(st_.astype(np.int64)), t(ct_.astype(np.int64)), x, 0, inclusive, reverse, False))
                ref = rows_reference(st_, ct_, inclusive, reverse)
                covered = np.zeros(N, bool)
                for s0, c in zip(st_, ct_):
                    covered[s0:s0 + c] = True
                np.testing.assert_allclose(got[covered], ref[covered], rtol=3e-5, atol=1e-4, err_msg=name)


def test_packed_scan_at_scale_equals_keyed():
    """2^21 elements, 60 k rows: the row-per-quarter-wave kernel against the keyed tiled kernel"""
    from nerfacc_amd import cuda as C

    g = torch.Generator(device=DEV).manual_seed(3)
    R = 60000
    cnts = torch.randint(0, 90, (R,), device=DEV, generator=g)
    ri = torch.repeat_interleave(torch.arange(R, device=DEV), cnts)
    N = ri.shape[0]
    assert N > 1 << 21
    starts = torch.cumsum(cnts, 0) - cnts
    x = torch.rand(N, device=DEV, generator=g)
    for inclusive in (True, False):
        for reverse in (False, True):
            a = C._lazy("_packed")(starts, cnts, x, 0, inclusive, reverse, False)
            b = C._lazy("_keyed")(ri, x, 0, inclusive, reverse)
            assert torch.allclose(a, b, rtol=1e-5, atol=1e-4)
    p = C._lazy("_packed")(starts, cnts, x * 0.2 + 0.9, 1, True, False, False)
    q = C._lazy("_keyed")(ri, x * 0.2 + 0.9, 1, True, False)
    assert torch.allclose(p, q, rtol=1e-4, atol=1e-30)


@pytest.mark.parametrize("env", [{"split_blk": 512}, {"split_blk": 512, "split_xt": 0}, {"split_blk": 256}])
def test_count_pass_workgroup_forms(force_options, env):
    """the 16-lanes-per-ray count pass runs in 256-thread workgroups (closed-form seam restart), in 512-thread ones, and in
    512-thread ones with the plane-crossing times written out in LDS (default from 3 k rays): every form on the adversarial
    fuzz cases, the degenerate grids, the overflow paths and the reference fixture"""
    import os

    import test_gpu_fuzz as F
    import test_gpu_grid as G
    import test_k2_reference as T

    force_options(split_p=16, **env)
    for seed in range(6):
        F.test_fuzz_single_level(seed, 700)
    F.test_fuzz_degenerate_grids()
    F.test_many_transitions_overflow_paths()
    G.test_traverse_lego_like_128_bit_exact_and_invariants()
    G.test_traverse_near_far_single_cell()
    k2 = dict(np.load(os.path.join(T.GOLD, "k2_reference.npz")))
    for name in ["m1_sphere", "lego_4k", "near_far", "degenerate"]:
        T._sampling_vs_fixture(name, k2)
    for name in ["ref_test_grid", "m1_noise", "lego_4k", "non_cubic", "steps_limit"]:
        T.test_hip_reproduces_reference_k2(name, k2)
