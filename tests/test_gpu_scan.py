"""HIP segmented scans vs torch.cumsum / cumprod (the reference's own test, tests/test_scan.py)
and vs the CPU oracle on ragged layouts."""
import numpy as np
import pytest
import torch

import oracle
from gpu_utils import DEV, n, ragged, t

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", ["inclusive_sum", "exclusive_sum", "inclusive_prod", "exclusive_prod"])
def test_three_modes_match_with_grads(name):
    # reference: tests/test_scan.py:7-172 (seed 42, [5,1000] uniform, tolerances from there)
    import nerfacc_amd.scan as S

    fn = getattr(S, name)
    torch.manual_seed(42)
    data = torch.rand((5, 1000), device=DEV, requires_grad=True)
    if "prod" in name:
        data = (data.detach() * 0.2 + 0.9).requires_grad_(True)   # keep products in range

    def run(**kw):
        out = fn(data.flatten() if kw else data, **kw).flatten()
        out.sum().backward()
        g = data.grad.clone()
        data.grad.zero_()
        return out, g

    o1, g1 = run()
    starts = torch.arange(0, data.numel(), data.shape[1], device=DEV, dtype=torch.long)
    cnts = torch.full((data.shape[0],), data.shape[1], dtype=torch.long, device=DEV)
    o2, g2 = run(packed_info=torch.stack([starts, cnts], -1))
    idx = torch.arange(data.shape[0], device=DEV, dtype=torch.long).repeat_interleave(data.shape[1])
    o3, g3 = run(indices=idx)
    atol = 3e-4 if "sum" in name else 1e-5
    for o, g in ((o2, g2), (o3, g3)):
        assert torch.allclose(o1, o, atol=atol, rtol=1e-5)
        assert torch.allclose(g1, g, rtol=2e-4, atol=1e-3)


def test_docstring_examples():
    import nerfacc_amd.scan as S

    x = torch.arange(1.0, 10.0, device=DEV)
    pk = torch.tensor([[0, 2], [2, 3], [5, 4]], device=DEV)
    idx = torch.tensor([0, 0, 1, 1, 1, 2, 2, 2, 2], device=DEV)
    want = dict(inclusive_sum=[1, 3, 3, 7, 12, 6, 13, 21, 30], exclusive_sum=[0, 1, 0, 3, 7, 0, 6, 13, 21],
                inclusive_prod=[1, 2, 3, 12, 60, 6, 42, 336, 3024], exclusive_prod=[1, 1, 1, 3, 12, 1, 6, 42, 336])
    for k, ref in want.items():
        assert getattr(S, k)(x, packed_info=pk).tolist() == ref
        assert getattr(S, k)(x, indices=idx).tolist() == ref
    with pytest.raises(ValueError):
        S.inclusive_sum(x, packed_info=pk, indices=idx)


@pytest.mark.parametrize("n_rays,max_len,seed", [(1, 1, 0), (9, 3, 1), (700, 150, 2), (5, 3000, 3), (50000, 9, 4)])
def test_ragged_vs_oracle(n_rays, max_len, seed):
    from nerfacc_amd import cuda as C

    rng = np.random.default_rng(seed)
    ri_, pk_ = ragged(rng, n_rays, max_len)
    N = ri_.shape[0]
    if N == 0:
        pytest.skip("empty draw")
    x_ = (rng.random(N) * 0.4 + 0.8).astype(np.float32)
    x, ri = t(x_), t(ri_)
    st, ct = t(pk_[:, 0].copy()), t(pk_[:, 1].copy())
    for op, opn in ((0, "sum"), (1, "prod")):
        for inc in (True, False):
            for rev in (False, True):
                ref = oracle.scan_keyed(x_, ri_, opn, inc, rev)
                tol = dict(rtol=3e-5, atol=1e-4 if opn == "sum" else 1e-6)
                got_k = C._lazy("_keyed")(ri, x, op, inc, rev)
                got_p = C._lazy("_packed")(st, ct, x, op, inc, rev, False)
                np.testing.assert_allclose(n(got_k), ref, **tol)
                np.testing.assert_allclose(n(got_p), ref, **tol)
    # normalise (utils_scan.cuh:101-109)
    ref = oracle.scan_packed(x_, pk_, "sum", True, normalize=True)
    np.testing.assert_allclose(n(C.inclusive_sum(st, ct, x, True, False)), ref, rtol=3e-5, atol=1e-6)
    ref = oracle.scan_packed(x_, pk_, "sum", False, normalize=True)
    np.testing.assert_allclose(n(C.exclusive_sum(st, ct, x, True, False)), ref, rtol=3e-5, atol=1e-6)
    # product backward, both layouts (scan.cu:199-210)
    g_ = rng.standard_normal(N).astype(np.float32)
    for inc in (True, False):
        out_ = oracle.scan_keyed(x_, ri_, "prod", inc)
        ref = oracle.prod_backward(x_, out_, g_, ri_, inc)
        fk = C.inclusive_prod_cub_backward if inc else C.exclusive_prod_cub_backward
        fp = C.inclusive_prod_backward if inc else C.exclusive_prod_backward
        np.testing.assert_allclose(n(fk(ri, x, t(out_), t(g_))), ref, rtol=2e-4, atol=1e-3)
        np.testing.assert_allclose(n(fp(st, ct, x, t(out_), t(g_))), ref, rtol=2e-4, atol=1e-3)


def test_empty_and_cpu_inputs():
    import nerfacc_amd.scan as S

    e = torch.zeros(0, device=DEV)
    assert S.inclusive_sum(e, indices=torch.zeros(0, dtype=torch.long, device=DEV)).shape == (0,)
    assert S.exclusive_prod(e, packed_info=torch.zeros((3, 2), dtype=torch.long, device=DEV)).shape == (0,)
    with pytest.raises(RuntimeError):
        S.inclusive_sum(torch.rand(4), indices=torch.zeros(4, dtype=torch.long))


@pytest.mark.parametrize("rw", [4, 16])
def test_packed_scan_rows_dealt_to_groups(force_options, rw):
    """the packed scan deals a wave's rows to its four 16-lane groups as they finish (scan.hip): both group sizes, rows of very
    different lengths and many empty rows, against the oracle — forward, reverse (the backward pass) and normalised"""
    force_options(scan_rw=rw)
    for args in ((9, 3, 1), (700, 150, 2), (5, 3000, 3), (50000, 9, 4), (333, 700, 5)):
        test_ragged_vs_oracle(*args)
    for name in ("inclusive_sum", "exclusive_prod"):
        test_three_modes_match_with_grads(name)
    test_docstring_examples()
