"""The single-launch form of the visibility filter (round 6, option `fused_vis` = 1 — opt-in, it measures slower than the two kernels:
one pass, survivors staged in LDS, look-back over the workgroups, flush; csrc/render.hip visibility_onepass_kernel<E, true>; C ABI
nfa_visibility_compact_sync) against the boolean-mask gather the reference
does (occ_grid.py:194-220: masks = render_visibility_from_density(...); ray_indices[masks], t_starts[masks], t_ends[masks]) and
against the three-kernel form it replaces — tensor for tensor, at sizes on both sides of the form's window, many times in a row (a
sync block left dirty by one launch would break the next: the same block serves the fused sampling launch)."""
import threading

import pytest
import torch

from gpu_utils import DEV

pytestmark = pytest.mark.gpu


def _ragged(n_rays, max_cnt, seed, opaque_every=0):
    g = torch.Generator().manual_seed(seed)
    cnts = torch.randint(0, max_cnt, (n_rays,), generator=g)
    cnts[torch.randint(0, n_rays, (n_rays // 5 + 1,), generator=g)] = 0
    ri = torch.repeat_interleave(torch.arange(n_rays), cnts).to(DEV)
    N = ri.shape[0]
    ts = (torch.rand(N, generator=g) * 4).to(DEV)
    te = ts + 5e-3
    sig = (torch.rand(N, generator=g) * 40).to(DEV)
    if opaque_every:
        sig[(ri % opaque_every) == 0] = 600.0           # opaque after a few samples: long cut tails, chunks without a survivor
    return ri, ts, te, sig


def _fused(n):
    from nerfacc_amd.cuda import _backend

    return bool(_backend.load_library().nfa_visibility_compact_fused(n, 1))


@pytest.mark.parametrize("n_rays,max_cnt", [(1, 2), (3, 40), (50, 120), (700, 90), (6500, 80), (9000, 70), (20000, 110), (40000, 200), (300, 3000)])
def test_fused_filter_equals_mask_gather_and_three_kernel_form(n_rays, max_cnt, force_options):
    """(300, 3000): rays far longer than a tile — the LDS image overflows and the remainder takes the kernel's counted-only path"""
    from nerfacc_amd import cuda as C

    import nerfacc_amd

    force_options(fused_vis=1)
    for rep, (eps, thre) in enumerate(((1e-4, 0.0), (1e-2, 0.03), (0.0, 0.0))):
        ri, ts, te, sig = _ragged(n_rays, max_cnt, 31 * n_rays + rep, opaque_every=7 if rep else 0)
        o = C.visibility_compact(ri, ts, te, sig, False, eps, thre, True)
        mask = o[3]
        assert torch.equal(o[0], ri[mask]) and torch.equal(o[1], ts[mask]) and torch.equal(o[2], te[mask])
        p = C.visibility_compact(ri, ts, te, sig, False, eps, thre, False)              # without the byte mask (what sampling() asks for)
        assert p[3] is None and all(torch.equal(a, b) for a, b in zip(p[:3], o[:3]))
        with nerfacc_amd.options(fused_vis=0):
            q = C.visibility_compact(ri, ts, te, sig, False, eps, thre, True)
        assert all(torch.equal(a, b) for a, b in zip(q, o))
        # from alphas
        al = 1.0 - torch.exp(-sig * (te - ts))
        a1 = C.visibility_compact(ri, ts, te, al, True, eps, thre, True)
        with nerfacc_amd.options(fused_vis=0):
            a0 = C.visibility_compact(ri, ts, te, al, True, eps, thre, True)
        assert all(torch.equal(a, b) for a, b in zip(a1, a0))
        assert torch.equal(a1[0], ri[a1[3]])


def test_fused_form_window(force_options):
    """what nfa_visibility_compact_fused says: off unless asked for; then the training size and an 800 x 800 frame's chunk are inside,
    2^24 samples are not"""
    assert not _fused(250_000)
    force_options(fused_vis=1)
    assert _fused(1) and _fused(250_000) and _fused(393_216)
    assert not _fused(1 << 24)


def test_fused_filter_back_to_back_with_fused_sampling(force_options):
    """estimator.sampling = fused sampling launch + field + fused filter launch, sharing one sync block per stream: 30 steps, every
    one equal to the step with both single-launch forms switched off"""
    import numpy as np

    from gpu_utils import sampling_is_fused, sparse_like, t

    import nerfacc_amd

    o, d, aabb, occ = sparse_like(11, 6564)
    assert sampling_is_fused(o, d, occ, aabb, 5e-3)
    est = nerfacc_amd.OccGridEstimator(roi_aabb=aabb[0].tolist(), resolution=occ.shape[1], levels=1).to(DEV)
    est.binaries = t(occ)
    O, D = t(o), t(d)

    def sigma_fn(t0, t1, ri):
        return (torch.sin(37.0 * (t0 + ri.float())) + 1.0) * 30.0

    def step():
        return est.sampling(O, D, sigma_fn=sigma_fn, render_step_size=5e-3, early_stop_eps=1e-4, alpha_thre=0.0)

    with nerfacc_amd.options(fused_vis=0, fused_sample=0):
        ref = step()
    assert 0 < ref[0].shape[0]
    force_options(fused_vis=1)
    for _ in range(30):
        out = step()
        assert all(torch.equal(a, b) for a, b in zip(out, ref))


def test_fused_filter_on_two_streams(force_options):
    """two host threads, each with its own stream and sync block, filtering concurrently (the look-back's waits are bounded: a launch
    that cannot finish hands over to the compaction kernel — results are the same either way)"""
    from nerfacc_amd import cuda as C

    ri, ts, te, sig = _ragged(6500, 80, 5, opaque_every=9)
    ref = C.visibility_compact(ri, ts, te, sig, False, 1e-3, 0.0, False)
    torch.cuda.synchronize()
    force_options(fused_vis=1)
    errors = []

    def work():
        try:
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.default_stream())
            with torch.cuda.stream(s):
                for _ in range(40):
                    out = C.visibility_compact(ri, ts, te, sig, False, 1e-3, 0.0, False)
                    if not all(torch.equal(a, b) for a, b in zip(out[:3], ref[:3])):
                        errors.append("mismatch")
            s.synchronize()
        except Exception as e:      # noqa: BLE001
            errors.append(repr(e))

    th = [threading.Thread(target=work) for _ in range(2)]
    for x in th:
        x.start()
    for x in th:
        x.join()
    assert not errors, errors[:3]
