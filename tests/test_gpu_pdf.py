"""pdf.hip vs the reference's pure-torch twins (tests/test_pdf.py) and the CPU oracle."""
import numpy as np
import pytest
import torch

import oracle
from gpu_utils import DEV, n, t

pytestmark = pytest.mark.gpu


def _intervals(n_rays, n_samples, seed=42):
    from nerfacc_amd.data_specs import RayIntervals

    torch.manual_seed(seed)
    vals = torch.sort(torch.rand((n_rays, n_samples + 1), device=DEV), -1)[0]
    return RayIntervals(vals=vals)


def test_searchsorted():
    # reference: tests/test_pdf.py:45-62
    from nerfacc_amd.pdf import searchsorted

    query, key = _intervals(10, 100, 1), _intervals(10, 100, 2)
    ids_left, ids_right = searchsorted(key, query)
    ref = torch.clamp(torch.searchsorted(key.vals, query.vals, right=True), 0, key.vals.shape[-1] - 1)
    assert torch.equal(ids_right, ref)
    assert torch.equal(ids_left, torch.clamp(ref - 1, min=0))
    l, r = oracle.searchsorted(n(key.vals), n(query.vals))
    assert np.array_equal(n(ids_left), l) and np.array_equal(n(ids_right), r)


def test_importance_sampling_vs_twin_and_oracle(golden):
    # reference: tests/test_pdf.py:65-94 (atol 1e-4)
    from nerfacc_amd.data_specs import RayIntervals
    from nerfacc_amd.pdf import _sample_from_weighted, importance_sampling

    intervals = _intervals(5, 100)
    cdfs = torch.sort(torch.rand_like(intervals.vals), -1)[0]
    new_iv, samples = importance_sampling(intervals, cdfs, 100, False)
    for i in range(5):
        e, m = _sample_from_weighted(intervals.vals[i:i + 1], cdfs[i:i + 1, 1:] - cdfs[i:i + 1, :-1], 100, False,
                                     intervals.vals[i].min(), intervals.vals[i].max())
        assert torch.allclose(new_iv.vals[i:i + 1], e, atol=1e-4)
        assert torch.allclose(samples.vals[i:i + 1], m, atol=1e-4)
    re, rm = oracle.importance_sampling(n(intervals.vals), n(cdfs), 100)
    np.testing.assert_allclose(n(new_iv.vals), re, atol=1e-6)
    np.testing.assert_allclose(n(samples.vals), rm, atol=1e-6)
    # golden from the reference's twin
    g = golden
    iv2, s2 = importance_sampling(RayIntervals(vals=t(g["p_vals"])), t(g["p_cdfs"]), 100)
    np.testing.assert_allclose(n(iv2.vals), g["p_edges"], atol=1e-4)
    np.testing.assert_allclose(n(s2.vals), g["p_mids"], atol=1e-4)
    # stratified: still sorted, inside the ray's range
    iv3, s3 = importance_sampling(intervals, cdfs, 64, True)
    assert (s3.vals[:, 1:] >= s3.vals[:, :-1]).all()
    assert (iv3.vals >= intervals.vals.min(-1, keepdim=True)[0] - 1e-6).all()


@pytest.mark.parametrize("flattened_input", [False, True])
def test_importance_sampling_per_ray_counts(flattened_input):
    """round 5: the Tensor overload (nerfacc.cpp:100-105; the reference's own allocates nothing, pdf.cu:324).  Flattened outputs whose
    rows equal the int call with that ray's count — bit for bit, it is the same arithmetic — and the reference's pure-torch twin
    `_sample_from_weighted` (tests/test_pdf.py:65-94's tolerance); packed_info / ray_indices / flags as pdf.cu:112-116, 207-239 state;
    rays with a count of 0 get nothing."""
    from nerfacc_amd.data_specs import RayIntervals
    from nerfacc_amd.pdf import _sample_from_weighted, importance_sampling

    R, S = 37, 64
    intervals = _intervals(R, S, 7)
    cdfs = torch.sort(torch.rand_like(intervals.vals), -1)[0]
    g = torch.Generator().manual_seed(3)
    cnts = torch.randint(0, 40, (R,), generator=g)
    cnts[[0, 5, 6, R - 1]] = 0                        # leading, adjacent and trailing rays without samples
    cnts[[2, 9]] = 1
    cnts_d = cnts.to(DEV)
    src = intervals
    if flattened_input:
        pk = torch.stack([torch.arange(R, device=DEV) * (S + 1), torch.full((R,), S + 1, device=DEV)], -1)
        src = RayIntervals(vals=intervals.vals.reshape(-1), packed_info=pk)
    iv, sm = importance_sampling(src, cdfs.reshape(-1) if flattened_input else cdfs, cnts_d, False)
    assert sm.vals.shape == (int(cnts.sum()),) and iv.vals.shape == (int((cnts + (cnts > 0)).sum()),)
    assert torch.equal(sm.packed_info[:, 1].cpu(), cnts) and torch.equal(sm.packed_info[:, 0].cpu(), torch.cumsum(cnts, 0) - cnts)
    ecnt = (cnts + 1) * (cnts > 0)
    assert torch.equal(iv.packed_info[:, 1].cpu(), ecnt) and torch.equal(iv.packed_info[:, 0].cpu(), torch.cumsum(ecnt, 0) - ecnt)
    assert torch.equal(sm.ray_indices.cpu(), torch.repeat_interleave(torch.arange(R), cnts))
    assert torch.equal(iv.ray_indices.cpu(), torch.repeat_interleave(torch.arange(R), ecnt))
    for r in range(R):
        k = int(cnts[r])
        s0, e0 = int(sm.packed_info[r, 0]), int(iv.packed_info[r, 0])
        if k == 0:
            continue
        b_iv, b_sm = importance_sampling(RayIntervals(vals=intervals.vals[r:r + 1]), cdfs[r:r + 1], k, False)
        assert torch.equal(sm.vals[s0:s0 + k], b_sm.vals[0]) and torch.equal(iv.vals[e0:e0 + k + 1], b_iv.vals[0]), r
        fl, fr = iv.is_left[e0:e0 + k + 1], iv.is_right[e0:e0 + k + 1]
        assert fl[:-1].all() and not fl[-1] and fr[1:].all() and not fr[0]
        if k > 1:
            e, m = _sample_from_weighted(intervals.vals[r:r + 1], cdfs[r:r + 1, 1:] - cdfs[r:r + 1, :-1], k, False,
                                         intervals.vals[r].min(), intervals.vals[r].max())
            assert torch.allclose(iv.vals[e0:e0 + k + 1], e[0], atol=1e-4) and torch.allclose(sm.vals[s0:s0 + k], m[0], atol=1e-4)
        re, rm = oracle.importance_sampling(n(intervals.vals[r:r + 1]), n(cdfs[r:r + 1]), k)
        np.testing.assert_allclose(n(iv.vals[e0:e0 + k + 1]), re[0], atol=1e-6)
        np.testing.assert_allclose(n(sm.vals[s0:s0 + k]), rm[0], atol=1e-6)
    # stratified, all rays empty, wrong sizes
    iv2, sm2 = importance_sampling(src, cdfs.reshape(-1) if flattened_input else cdfs, cnts_d, True)
    assert sm2.vals.shape == sm.vals.shape and torch.isfinite(sm2.vals).all()
    iv3, sm3 = importance_sampling(src, cdfs.reshape(-1) if flattened_input else cdfs, torch.zeros(R, dtype=torch.long, device=DEV), False)
    assert sm3.vals.numel() == 0 and iv3.vals.numel() == 0
    with pytest.raises(RuntimeError):
        importance_sampling(src, cdfs.reshape(-1) if flattened_input else cdfs, cnts_d[:-1], False)


def test_flattened_docstring_examples():
    # reference: nerfacc/pdf.py:40-56, 108-120
    from nerfacc_amd.data_specs import RayIntervals
    from nerfacc_amd.pdf import importance_sampling, searchsorted

    seq = RayIntervals(vals=torch.tensor([0.0, 1.0, 0.0, 1.0, 2.0], device=DEV),
                       packed_info=torch.tensor([[0, 2], [2, 3]], device=DEV))
    q = RayIntervals(vals=torch.tensor([0.5, 1.5, 2.5], device=DEV), packed_info=torch.tensor([[0, 1], [1, 2]], device=DEV))
    l, r = searchsorted(seq, q)
    assert l.tolist() == [0, 3, 3] and r.tolist() == [1, 4, 4]
    iv, s = importance_sampling(seq, torch.tensor([0.0, 0.5, 0.0, 0.5, 1.0], device=DEV), 2)
    assert torch.allclose(iv.vals, torch.tensor([[0.0, 0.5, 1.0], [0.0, 1.0, 2.0]], device=DEV))
    assert torch.allclose(s.vals, torch.tensor([[0.25, 0.75], [0.5, 1.5]], device=DEV))


def test_pdf_loss_and_propnet_sampling():
    # reference: tests/test_pdf.py:97-127
    from nerfacc_amd.estimators.prop_net import PropNetEstimator, _lossfun_outer, _pdf_loss
    from nerfacc_amd.pdf import importance_sampling

    intervals = _intervals(5, 100)
    cdfs = torch.sort(torch.rand_like(intervals.vals), -1)[0]
    iv2, _ = importance_sampling(intervals, cdfs, 10, False)
    cdfs2 = torch.sort(torch.rand_like(iv2.vals), -1)[0]
    loss = _pdf_loss(intervals, cdfs, iv2, cdfs2)
    loss2 = _lossfun_outer(intervals.vals, cdfs[:, 1:] - cdfs[:, :-1], iv2.vals, cdfs2[:, 1:] - cdfs2[:, :-1])
    assert torch.allclose(loss, loss2, atol=1e-4)
    # config-3 shapes (train_ngp_nerf_prop.py:66,92-93): 4096 rays, 256 -> 96 -> 48
    est = PropNetEstimator().to(DEV)
    fn = lambda a, b: torch.exp(-(0.5 * (a + b) - 2.0) ** 2) * 3.0
    ts, te = est.sampling([fn, fn], [256, 96], 48, n_rays=4096, near_plane=0.2, far_plane=1e3, requires_grad=True)
    assert ts.shape == te.shape == (4096, 48)
    assert (te >= ts).all() and (ts[:, 1:] >= ts[:, :-1] - 1e-6).all() and (ts >= 0.2 - 1e-4).all()
    trans = torch.rand(4096, 48, device=DEV).cumsum(-1).neg().exp()
    assert est.compute_loss(trans).ndim == 0


def test_transform_stot_is_the_torch_expression_bit_for_bit():
    """nfa_transform_stot vs prop_net.py:215-229 written with torch ops: the same float operations in the same order"""
    from nerfacc_amd.estimators.prop_net import _transform_stot

    s = torch.cat([torch.rand(4096, 257, device=DEV), torch.tensor([[0.0, 1.0] + [0.5] * 255], device=DEV)])
    for t_min, t_max in ((0.2, 1e3), (2.0, 6.0), (0.05, 1e10), (1e-3, 0.3), (0.3, 1e-3), (0.7, 3.3)):      # (1e-3, 0.3: float(1/double(t)) != 1/float(t))
        want_u = s * t_max + (1 - s) * t_min
        want_l = 1 / (s * (1 / t_max) + (1 - s) * (1 / t_min))
        assert torch.equal(_transform_stot("uniform", s, t_min, t_max), want_u)
        assert torch.equal(_transform_stot("lindisp", s, t_min, t_max), want_l)
    ts = _transform_stot("lindisp", s[:, :-1].transpose(0, 1), 0.2, 1e3)      # a non-contiguous view
    assert torch.equal(ts, 1 / (s[:, :-1].transpose(0, 1) * (1 / 1e3) + (1 - s[:, :-1].transpose(0, 1)) * (1 / 0.2)))


@pytest.mark.parametrize("R,S", [(4096, 256), (4096, 96), (33, 48), (5, 1), (3, 700)])
def test_level_cdfs_kernel_vs_the_reference_composition(R, S):
    """pdf.hip's edge_cdfs_* (one launch per proposal level) against render_transmittance_from_density + 1 - cat([trans, 0]) written
    with torch ops (prop_net.py:99-112): values, the gradient with respect to the densities, an opaque last sample (sigma = inf)"""
    from nerfacc_amd.estimators.prop_net import _edge_cdfs, _level_cdfs

    torch.manual_seed(R + S)
    t_vals = torch.sort(torch.rand(R, S + 1, device=DEV) * 5 + 0.1, dim=-1)[0]
    for opaque in (False, True):
        sig = (torch.rand(R, S, device=DEV) * 4).requires_grad_(True)
        sig_ref = sig.detach().clone().requires_grad_(True)

        def bk(x):
            if not opaque:
                return x
            x = x.clone()
            x[..., -1] = torch.inf
            return x

        cdfs = _level_cdfs(t_vals, bk(sig))
        x = bk(sig_ref) * (t_vals[:, 1:] - t_vals[:, :-1])
        trans = torch.exp(-torch.cumsum(torch.cat([torch.zeros_like(x[:, :1]), x[:, :-1]], -1), -1))
        want = _edge_cdfs(trans)
        assert cdfs.shape == (R, S + 1) and torch.equal(cdfs[:, -1], torch.ones(R, device=DEV)) and torch.equal(cdfs[:, 0], torch.zeros(R, device=DEV))
        assert torch.allclose(cdfs, want, rtol=1e-5, atol=2e-6)
        coef = torch.randn(R, S + 1, device=DEV)
        (cdfs * coef).sum().backward()
        (want * coef).sum().backward()
        assert torch.isfinite(sig.grad).all()
        assert torch.allclose(sig.grad, sig_ref.grad, rtol=2e-4, atol=2e-5)
    with torch.no_grad():                                  # no graph: no `trans` kept
        assert torch.equal(_level_cdfs(t_vals, sig.detach()), _level_cdfs(t_vals, sig).detach())


@pytest.mark.parametrize("R,nq,nk", [(4096, 48, 256), (4096, 48, 96), (7, 1, 1), (33, 300, 5), (5, 3, 700)])
def test_pdf_loss_kernel_vs_the_reference_composition(R, nq, nk):
    """pdf.hip's pdf_loss_* (one launch forward, one backward, no atomics) against prop_net.py:232-256 written with searchsorted +
    torch ops: the same loss bits, the gradient with respect to the key cdfs within float tolerance, and run-to-run identical"""
    from nerfacc_amd.data_specs import RayIntervals
    from nerfacc_amd.estimators.prop_net import _pdf_loss

    torch.manual_seed(R + nq + nk)
    q = torch.sort(torch.rand(R, nq + 1, device=DEV) * 1.2 - 0.1, -1)[0]      # query edges partly outside the key's range
    k = torch.sort(torch.rand(R, nk + 1, device=DEV), -1)[0]
    if nq > 2:
        q[:, 1] = q[:, 2]                                                       # an empty query interval
    cq = torch.sort(torch.rand(R, nq + 1, device=DEV), -1)[0]
    ck = torch.sort(torch.rand(R, nk + 1, device=DEV), -1)[0].requires_grad_(True)
    ck_ref = ck.detach().clone().requires_grad_(True)
    qi, ki = RayIntervals(vals=q), RayIntervals(vals=k)
    loss = _pdf_loss(qi, cq, ki, ck)
    want = _pdf_loss(qi, cq.clone().requires_grad_(True), ki, ck_ref)            # (a query cdf with a gradient: the composition)
    assert loss.shape == (R, nq) and torch.equal(loss, want.detach())
    coef = torch.rand(R, nq, device=DEV)
    (loss * coef).sum().backward()
    (want * coef).sum().backward()
    # (an edge that is the right index of one stretch of intervals and the left index of another: two sums of up to nq terms each)
    assert torch.allclose(ck.grad, ck_ref.grad, rtol=1e-4, atol=2e-7 * nq * float(ck_ref.grad.abs().max()) + 1e-6)
    g1 = ck.grad.clone()
    ck.grad = None
    (_pdf_loss(qi, cq, ki, ck) * coef).sum().backward()
    assert torch.equal(ck.grad, g1)
    with torch.no_grad():
        assert torch.equal(_pdf_loss(qi, cq, ki, ck), loss.detach())


def test_pdf_loss_backward_with_unsorted_query_edges():
    """the fused path takes any batched float32 input; with query edges that do NOT ascend along a ray (or hold a NaN) the
    forward is still the reference's composition, and the gradient must be too — the backward kernel checks a ray's ids and scans all
    of its intervals per key edge when they do not ascend (the reference's gather backward is order-independent)"""
    from nerfacc_amd.data_specs import RayIntervals
    from nerfacc_amd.estimators.prop_net import _pdf_loss

    torch.manual_seed(5)
    R, nq, nk = 300, 40, 70
    q = torch.rand(R, nq + 1, device=DEV) * 1.2 - 0.1                          # unsorted on purpose
    q[::3] = torch.sort(q[::3], -1)[0]                                         # every third ray ascends: both paths in one launch
    k = torch.sort(torch.rand(R, nk + 1, device=DEV), -1)[0]
    cq = torch.sort(torch.rand(R, nq + 1, device=DEV), -1)[0]
    ck = torch.sort(torch.rand(R, nk + 1, device=DEV), -1)[0].requires_grad_(True)
    ck_ref = ck.detach().clone().requires_grad_(True)
    qi, ki = RayIntervals(vals=q), RayIntervals(vals=k)
    loss = _pdf_loss(qi, cq, ki, ck)
    want = _pdf_loss(qi, cq.clone().requires_grad_(True), ki, ck_ref)
    assert torch.equal(loss, want.detach())
    coef = torch.rand(R, nq, device=DEV)
    (loss * coef).sum().backward()
    (want * coef).sum().backward()
    assert torch.allclose(ck.grad, ck_ref.grad, rtol=1e-4, atol=2e-7 * nq * float(ck_ref.grad.abs().max()) + 1e-6)
    g1 = ck.grad.clone()
    ck.grad = None
    (_pdf_loss(qi, cq, ki, ck) * coef).sum().backward()
    assert torch.equal(ck.grad, g1)                                            # still no atomics: run-to-run identical
