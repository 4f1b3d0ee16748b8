"""Behaviour the reference has and round 1 had narrowed: pack_info on
unsorted indices, gradients w.r.t. t_starts / t_ends / prefix_trans, broadcast backgrounds, out-of-range ray
indices in accumulate's backward pass, alpha_thre after a grid update, re-entrancy across host threads."""
import threading

import numpy as np
import pytest
import torch

from gpu_utils import DEV, lego_like, n, ragged, t

pytestmark = pytest.mark.gpu


def _ref_pack_info(ri, n_rays):
    """nerfacc/pack.py:38-46 as written there"""
    cnts = torch.zeros((n_rays,), device=ri.device, dtype=torch.long)
    cnts.index_add_(0, ri, torch.ones_like(ri))
    starts = cnts.cumsum(0) - cnts
    return torch.stack([starts, cnts], -1)


@pytest.mark.parametrize("n_rays,N", [(3, 5), (1000, 40000), (70000, 300000), (5, 1), (7, 0)])
def test_pack_info_any_order(n_rays, N):
    from nerfacc_amd.pack import pack_info

    g = torch.Generator(device="cpu").manual_seed(n_rays + N)
    ri = torch.randint(0, n_rays, (N,), generator=g).to(DEV)
    for arr in (ri, torch.sort(ri)[0], torch.sort(ri, descending=True)[0]):
        got = pack_info(arr, n_rays)
        assert torch.equal(got, _ref_pack_info(arr, n_rays))
    if N > 1000:     # long runs of equal keys interleaved: wave-aggregated atomics
        arr = torch.sort(ri)[0].view(-1, 8).t().contiguous().view(-1)
        assert torch.equal(pack_info(arr, n_rays), _ref_pack_info(arr, n_rays))
    # the docstring example of pack.py:29-32
    ex = torch.tensor([0, 0, 1, 1, 1, 2, 2, 2, 2], device=DEV)
    assert pack_info(ex, 3).tolist() == [[0, 2], [2, 3], [5, 4]]


def test_accumulate_backward_ignores_out_of_range_keys():
    from nerfacc_amd.volrend import accumulate_along_rays

    ri = torch.tensor([0, 0, -1, 1, 1, 7, 2], device=DEV)      # -1: the gap marker of unpack_info; 7 >= n_rays
    w = torch.rand(7, device=DEV, requires_grad=True)
    v = torch.rand(7, 3, device=DEV, requires_grad=True)
    out = accumulate_along_rays(w, v, ri, n_rays=3)
    g = torch.rand_like(out)
    (out * g).sum().backward()
    ok = (ri >= 0) & (ri < 3)
    rc = ri.clamp(0, 2)
    want_w = torch.where(ok, (g[rc] * v.detach()).sum(-1), torch.zeros(7, device=DEV))
    want_v = torch.where(ok[:, None], w.detach()[:, None] * g[rc], torch.zeros(7, 3, device=DEV))
    assert torch.allclose(w.grad, want_w, atol=1e-6) and torch.allclose(v.grad, want_v, atol=1e-6)


def _dense_case(R=37, S=21, seed=0):
    g = torch.Generator().manual_seed(seed)
    t0 = torch.sort(torch.rand((R, S + 1), generator=g) * 3.0, -1)[0].to(DEV)
    sig = (torch.rand((R, S), generator=g) * 5.0).to(DEV)
    rgb = torch.rand((R, S, 3), generator=g).to(DEV)
    return t0[:, :-1].contiguous(), t0[:, 1:].contiguous(), sig, rgb


def test_grads_wrt_t_and_prefix_trans_match_batched_torch():
    """volrend.py:266-278 is differentiable w.r.t. t_starts / t_ends / prefix_trans; flattened == batched torch"""
    from nerfacc_amd.volrend import render_weight_from_density

    ts, te, sig, _ = _dense_case()
    R, S = sig.shape
    ri = torch.arange(R, device=DEV).repeat_interleave(S)
    pt = torch.rand(R, S, device=DEV) * 0.5 + 0.5
    leaves_b = [x.clone().requires_grad_(True) for x in (ts, te, sig, pt)]
    sd = leaves_b[2] * (leaves_b[1] - leaves_b[0])
    T = torch.exp(-(torch.cumsum(sd, -1) - sd)) * leaves_b[3]
    wb = T * (1 - torch.exp(-sd))
    gw = torch.rand_like(wb)
    (wb * gw).sum().backward()
    leaves_f = [x.flatten().clone().requires_grad_(True) for x in (ts, te, sig, pt)]
    wf, Tf, af = render_weight_from_density(leaves_f[0], leaves_f[1], leaves_f[2], ray_indices=ri, prefix_trans=leaves_f[3])
    assert torch.allclose(wf, wb.detach().flatten(), atol=1e-5)
    (wf * gw.flatten()).sum().backward()
    for a, b in zip(leaves_f, leaves_b):
        assert a.grad is not None and torch.allclose(a.grad, b.grad.flatten(), atol=2e-4, rtol=1e-4)
    # only t_starts wants a gradient (used to raise in backward)
    ts2 = ts.flatten().clone().requires_grad_(True)
    w2, _, _ = render_weight_from_density(ts2, te.flatten(), sig.flatten(), ray_indices=ri)
    w2.sum().backward()
    assert torch.isfinite(ts2.grad).all() and ts2.grad.abs().sum() > 0


def test_rendering_background_broadcasts_like_the_reference():
    from nerfacc_amd.volrend import rendering

    ts, te, sig, rgb = _dense_case(R=50, S=9, seed=3)
    R, S = sig.shape
    ri = torch.arange(R, device=DEV).repeat_interleave(S)
    fn = lambda *_: (rgb.view(-1, 3), sig.flatten())
    base, opa, _, _ = rendering(ts.flatten(), te.flatten(), ri, R, rgb_sigma_fn=fn)
    per_ray = torch.rand(R, 3, device=DEV)
    for bk in (per_ray, torch.rand(1, device=DEV), torch.rand(1, 3, device=DEV), torch.rand(3, device=DEV)):
        col, _, _, _ = rendering(ts.flatten(), te.flatten(), ri, R, rgb_sigma_fn=fn, render_bkgd=bk)
        assert torch.allclose(col, base + bk * (1 - opa), atol=1e-6)
    # differentiable background (volrend.py:161-162 is a torch expression)
    bk = torch.rand(R, 3, device=DEV, requires_grad=True)
    col, _, _, _ = rendering(ts.flatten(), te.flatten(), ri, R, rgb_sigma_fn=fn, render_bkgd=bk)
    col.sum().backward()
    assert torch.allclose(bk.grad, (1 - opa).expand(R, 3), atol=1e-6)


def test_rendering_depth_gradient_reaches_t():
    from nerfacc_amd.volrend import rendering

    ts, te, sig, rgb = _dense_case(R=20, S=12, seed=5)
    R, S = sig.shape
    ri = torch.arange(R, device=DEV).repeat_interleave(S)
    tsf = ts.flatten().clone().requires_grad_(True)
    fn = lambda *_: (rgb.view(-1, 3), sig.flatten())
    _, _, depth, _ = rendering(tsf, te.flatten(), ri, R, rgb_sigma_fn=fn, expected_depths=False)
    depth.sum().backward()
    tsb = ts.clone().requires_grad_(True)
    sd = sig * (te - tsb)
    w = torch.exp(-(torch.cumsum(sd, -1) - sd)) * (1 - torch.exp(-sd))
    (w * (tsb + te) / 2).sum().backward()
    assert torch.allclose(tsf.grad, tsb.grad.flatten(), atol=2e-4, rtol=1e-4)


def test_alpha_thre_uses_the_mean_of_the_updated_grid():
    """occ_grid.py:183 recomputes occs.mean() on every call; the device update writes occs through a raw pointer"""
    from nerfacc_amd import OccGridEstimator
    from nerfacc_amd.volrend import render_visibility_from_density

    o, d, aabb, occ = lego_like(3, 512, res=32)
    est = OccGridEstimator(roi_aabb=aabb[0].tolist(), resolution=32, levels=1).to(DEV)
    est.binaries = t(occ)
    O, D = t(o), t(d)
    sig_fn = lambda ts, te, ri: torch.full_like(ts, 3.0)          # alpha = 1 - exp(-3 * 0.02) = 0.058
    kw = dict(render_step_size=2e-2, early_stop_eps=0.0, alpha_thre=0.5)
    first = est.sampling(O, D, sigma_fn=sig_fn, **kw)               # occs.mean() == 0 -> alpha_thre = 0: keeps all
    assert first[0].numel() > 0
    est.train()
    est._update(step=0, occ_eval_fn=lambda x: torch.full((x.shape[0], 1), 0.3, device=DEV), occ_thre=0.01)
    mean = est.occs.mean().item()
    assert mean > 0.2
    est.binaries = t(occ)                                           # same geometry, new occs
    got = est.sampling(O, D, sigma_fn=sig_fn, **kw)                 # alpha_thre = min(0.5, 0.3) > 0.058: drops all
    ri, ts, te = est.sampling(O, D, render_step_size=2e-2)
    keep = render_visibility_from_density(ts, te, sig_fn(ts, te, ri), ray_indices=ri, early_stop_eps=0.0,
                                          alpha_thre=min(0.5, mean))
    assert got[0].numel() == int(keep.sum().item()) == 0


def test_sampling_is_reentrant_across_host_threads():
    from nerfacc_amd import OccGridEstimator

    o, d, aabb, occ = lego_like(11, 6000, res=64)
    est = OccGridEstimator(roi_aabb=aabb[0].tolist(), resolution=64, levels=1).to(DEV)
    est.binaries = t(occ)
    O, D = t(o), t(d)
    parts = [(0, 6000), (0, 1500), (1500, 6000), (300, 400)]
    want = [tuple(x.clone() for x in est.sampling(O[a:b], D[a:b], render_step_size=5e-3)) for a, b in parts]
    errors = []

    def worker(k):
        try:
            a, b = parts[k]
            s = torch.cuda.Stream()
            with torch.cuda.stream(s):
                for _ in range(40):
                    got = est.sampling(O[a:b], D[a:b], render_step_size=5e-3)
                    s.synchronize()
                    for x, y in zip(got, want[k]):
                        if x.shape != y.shape or not torch.equal(x, y):
                            errors.append((k, tuple(x.shape), tuple(y.shape)))
                            return
        except Exception as e:      # noqa: BLE001
            errors.append((k, repr(e)))

    threads = [threading.Thread(target=worker, args=(k,)) for k in range(len(parts))]
    [th.start() for th in threads]
    [th.join() for th in threads]
    assert not errors, errors[:3]


def test_exchange_adam_fused_path_matches_torch_adam_on_device():
    """sharding.ExchangeAdam (per-chunk torch._fused_adam_ on views of one flat buffer) vs torch.optim.Adam(fused=True)"""
    from nerfacc_amd.sharding import ExchangeAdam

    torch.manual_seed(0)
    a = [torch.nn.Parameter(torch.randn(4, 33, 33, 33, device=DEV))]
    b = [torch.nn.Parameter(a[0].detach().clone())]
    oa = torch.optim.Adam(a, lr=1e-2, eps=1e-15, weight_decay=1e-6, fused=True)
    ob = ExchangeAdam(b, lr=1e-2, eps=1e-15, weight_decay=1e-6, n_chunks=4)
    assert ob._fused
    for it in range(5):
        for ps, o in ((a, oa), (b, ob)):
            o.zero_grad()
            ((ps[0] * 0.7 - 0.1).square().sum() * (it + 1) * 1024.0).backward()
            o.step()
    assert torch.allclose(a[0], b[0], atol=1e-6, rtol=1e-5)


def test_in_kernel_near_far_planes_equal_the_torch_composition():
    """OccGridEstimator.sampling hands near_plane / far_plane / t_min / t_max / the stratified jitter to the kernel; the planes it
    forms must be the floats occ_grid.py:154-163 forms with torch ops (full_like, clamp, rand * step added in place)"""
    from nerfacc_amd import cuda as C

    o, d, aabb, occ = lego_like(21, 5000, res=64)
    O, D, B, A = t(o), t(d), t(occ), t(aabb)
    R = O.shape[0]
    g = torch.Generator(device=DEV).manual_seed(3)
    t_min = torch.rand(R, device=DEV, generator=g) * 4.0
    t_max = t_min + torch.rand(R, device=DEV, generator=g) * 3.0
    jit = torch.rand(R, device=DEV, generator=g)
    step, near_plane, far_plane = 7e-3, 0.35, 5.5
    near = torch.clamp(torch.full_like(O[..., 0], near_plane), min=t_min)
    far = torch.clamp(torch.full_like(O[..., 0], far_plane), max=t_max)
    near += jit * step
    want = C.sample_occgrid(O, D, B, A, near.contiguous(), far.contiguous(), step, 0.0)
    got = C.sample_occgrid(O, D, B, A, None, None, step, 0.0, near_plane=near_plane, far_plane=far_plane,
                           t_min=t_min, t_max=t_max, jitter=jit, jitter_scale=step)
    assert want[0].numel() > 1000
    for a, b in zip(got, want):
        assert torch.equal(a, b)
    # scalars only (the training call: near 0, far 1e10, stratified)
    want = C.sample_occgrid(O, D, B, A, (jit * step).contiguous(), torch.full((R,), 1e10, device=DEV), step, 0.0)
    got = C.sample_occgrid(O, D, B, A, None, None, step, 0.0, near_plane=0.0, far_plane=1e10, jitter=jit, jitter_scale=step)
    for a, b in zip(got, want):
        assert torch.equal(a, b)
