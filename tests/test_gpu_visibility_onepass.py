"""The one-pass form of the visibility filter (render.hip: visibility_onepass_kernel — survivors packed in LDS, tile counts
exchanged by decoupled look-back; an opt-in form, option vis_onepass) against the mask / compaction kernels: the two forms must give
the same survivors in the same order, bit for bit (reference semantics: estimators/occ_grid.py:202-231 — the boolean-mask
gather of ray_indices / t_starts / t_ends by `render_visibility_from_density`, volrend.py:330-366)."""
import pytest
import torch

from gpu_utils import DEV

# (the form spins on other workgroups' state words: provably finite — tickets — but a hung kernel would block the whole suite inside a
# synchronize, so every test of this module is bounded; pytest-timeout's thread method ends the process instead of waiting)
pytestmark = [pytest.mark.gpu, pytest.mark.timeout(180, method="thread")]


def _ragged(n_rays, max_cnt, seed, long_rays=()):
    g = torch.Generator(device="cpu").manual_seed(seed)
    cnts = torch.randint(0, max_cnt + 1, (n_rays,), generator=g)
    for pos, length in long_rays:
        cnts[pos % n_rays] = length
    ri = torch.repeat_interleave(torch.arange(n_rays), cnts).to(DEV)
    N = ri.shape[0]
    ts = (torch.rand(N, generator=g) * 4).to(DEV)
    te = ts + 5e-3
    return ri, ts, te, g


def _both(ri, ts, te, dens, from_alpha, eps, thre, want_mask, force_options, **form):
    from nerfacc_amd import cuda as C

    force_options(vis_onepass=0)
    ref = C.visibility_compact(ri, ts, te, dens, from_alpha, eps, thre, want_mask)
    force_options(vis_onepass=1, **form)
    got = C.visibility_compact(ri, ts, te, dens, from_alpha, eps, thre, want_mask)
    torch.cuda.synchronize()
    for a, b in zip(ref, got):
        if a is None:
            assert b is None
        else:
            assert a.shape == b.shape and torch.equal(a, b)
    return ref


@pytest.mark.parametrize("chunks", [2, 4, 7])
@pytest.mark.parametrize("e", [1, 2])
@pytest.mark.parametrize("n_rays,max_cnt", [(1, 5), (50, 120), (5000, 120), (60000, 40), (3, 9000)])
def test_onepass_equals_the_three_kernel_form(force_options, n_rays, max_cnt, e, chunks):
    ri, ts, te, g = _ragged(n_rays, max_cnt, n_rays + max_cnt)
    N = ri.shape[0]
    if N == 0:
        pytest.skip("empty draw")
    sig = (torch.rand(N, generator=g) * 60).to(DEV)
    force_options(e=e)
    ref = _both(ri, ts, te, sig, False, 1e-2, 0.05, True, force_options, vis_chunks=chunks)
    assert 0 < ref[0].shape[0] < N or N < 8
    _both(ri, ts, te, sig, False, 1e-4, 0.0, False, force_options, vis_chunks=chunks)


def test_onepass_long_rays_overflow_the_lds_image(force_options):
    """rays far longer than a tile, all of whose samples survive: the wave that owns one runs out of LDS image, only counts from
    there on (keep bytes in the mask / workspace) and compacts that remainder once its destination is known; the tiles the ray
    covers own nothing and publish zero"""
    ri, ts, te, g = _ragged(400, 30, 7, long_rays=((0, 3000), (17, 20000), (399, 7000), (200, 513)))
    N = ri.shape[0]
    sig = torch.full((N,), 1e-3, device=DEV)                 # transmittance stays ~1: everything is kept
    for e in (1, 2):
        force_options(e=e)
        for chunks in (2, 4):
            ref = _both(ri, ts, te, sig, False, 1e-4, 0.0, True, force_options, vis_chunks=chunks)
            assert ref[0].shape[0] == N
    # and with half of the long rays' samples cut by the early stop
    sig2 = sig.clone()
    sig2[ri == 17] = 80.0
    force_options(e=2)
    ref = _both(ri, ts, te, sig2, False, 1e-3, 0.0, True, force_options)
    assert ref[0].shape[0] < N


def test_onepass_nothing_and_everything_kept(force_options):
    ri, ts, te, g = _ragged(3000, 60, 11)
    N = ri.shape[0]
    force_options(e=2)
    none = _both(ri, ts, te, torch.zeros(N, device=DEV), False, 1e-4, 0.5, True, force_options)        # alpha = 0 < thre everywhere
    assert none[0].shape[0] == 0 and not bool(none[3].any())
    every = _both(ri, ts, te, torch.zeros(N, device=DEV), False, 1e-4, 0.0, True, force_options)
    assert every[0].shape[0] == N and bool(every[3].all())


def test_onepass_from_alpha(force_options):
    ri, ts, te, g = _ragged(4000, 90, 13)
    N = ri.shape[0]
    alphas = (torch.rand(N, generator=g) * 0.3).to(DEV)
    for e in (1, 2):
        force_options(e=e)
        ref = _both(ri, ts, te, alphas, True, 1e-2, 0.02, True, force_options)
        assert 0 < ref[0].shape[0] < N


def test_onepass_on_a_large_input_and_repeated_calls(force_options):
    """2^21 samples, several persistent workgroups per CU drawing tickets: same survivors as the three-kernel form, repeated calls
    (ticket counter and states are re-zeroed by every call) agree, and the gather equals the boolean-mask gather"""
    from nerfacc_amd import cuda as C

    ri, ts, te, g = _ragged(55000, 76, 5)
    N = ri.shape[0]
    assert N >= 1 << 20
    sig = (torch.rand(N, generator=g) * 30).to(DEV)
    force_options(vis_onepass=0)
    ref = C.visibility_compact(ri, ts, te, sig, False, 1e-3, 0.0, True)
    force_options(vis_onepass=1)
    for _ in range(3):
        got = C.visibility_compact(ri, ts, te, sig, False, 1e-3, 0.0, True)
        assert all(torch.equal(a, b) for a, b in zip(ref, got))
    m = got[3]
    assert torch.equal(got[0], ri[m]) and torch.equal(got[1], ts[m]) and torch.equal(got[2], te[m])
