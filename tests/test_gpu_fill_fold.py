"""nfa_rendering_fwd fills the rays WITHOUT a sample (background colour, opacity 0, depth 0) inside its own kernel (round 6: extra
workgroups that look at the gaps between the ascending ray_indices, csrc/render.hip fill_ray_gaps) instead of a launch in front of
it.  Checked against the separate launch (`fold_fill = 0`) bit for bit and against the reference's composition for those rays
(volrend.py:150-162: accumulate_along_rays leaves zeros, colors + render_bkgd * (1 - opacities))."""
import pytest
import torch

from gpu_utils import DEV

pytestmark = pytest.mark.gpu


def _case(n_rays, keys, seed=0):
    g = torch.Generator().manual_seed(seed)
    ri = torch.as_tensor(keys, dtype=torch.int64).to(DEV)
    N = ri.shape[0]
    ts = (torch.rand(N, generator=g) * 3).to(DEV)
    te = ts + 1e-2
    sig = (torch.rand(N, generator=g) * 30).to(DEV)
    rgb = torch.rand(N, 3, generator=g).to(DEV)
    return ri, ts, te, sig, rgb


def _render(case, n_rays, bkgd):
    from nerfacc_amd import cuda as C

    ri, ts, te, sig, rgb = case
    return C.rendering_fwd(ri, ts, te, sig, rgb, n_rays, bkgd, True)


KEYSETS = {
    "dense": (50, lambda: torch.repeat_interleave(torch.arange(50), 7)),
    "every_third": (300, lambda: torch.repeat_interleave(torch.arange(0, 300, 3), 5)),
    "leading_and_trailing": (5000, lambda: torch.repeat_interleave(torch.arange(2100, 2200), 40)),
    "one_sample_last_ray": (777, lambda: torch.tensor([776])),
    "one_sample_first_ray": (777, lambda: torch.tensor([0])),
    "long_interior_gaps": (200_000, lambda: torch.repeat_interleave(torch.tensor([3, 4, 70_000, 70_001, 199_000]), 33)),
    "random_ragged": (6500, lambda: torch.repeat_interleave(torch.arange(6500), torch.randint(0, 3, (6500,), generator=torch.Generator().manual_seed(3)) * 40)),
    "out_of_range_ends": (100, lambda: torch.repeat_interleave(torch.tensor([-2, -1, 5, 6, 50, 100, 130]), 3)),
}


@pytest.mark.parametrize("name", sorted(KEYSETS))
@pytest.mark.parametrize("with_bkgd", [True, False])
def test_fold_equals_separate_fill(name, with_bkgd, force_options):
    import nerfacc_amd

    n_rays, mk = KEYSETS[name]
    case = _case(n_rays, mk(), seed=len(name))
    bkgd = torch.tensor([0.25, 0.5, 0.75], device=DEV) if with_bkgd else None
    got = _render(case, n_rays, bkgd)
    with nerfacc_amd.options(fold_fill=0):
        want = _render(case, n_rays, bkgd)
    for a, b in zip(got, want):
        assert torch.equal(a, b)
    # the rays without a sample, as the reference composes them
    has = torch.zeros(n_rays, dtype=torch.bool, device=DEV)
    ri = case[0]
    ok = (ri >= 0) & (ri < n_rays)
    has[ri[ok]] = True
    colors, opac, depth = got[:3]
    assert torch.all(opac[~has] == 0) and torch.all(depth[~has] == 0)
    want_c = bkgd if with_bkgd else torch.zeros(3, device=DEV)
    assert torch.all(colors[~has] == want_c)


def test_fold_in_training_shape_many_times():
    """the training step's shape (6.5 k rays, 2.5e5 samples, a third of the rays empty), 20 launches back to back into fresh outputs"""
    import nerfacc_amd

    g = torch.Generator().manual_seed(9)
    cnts = torch.randint(0, 90, (6500,), generator=g)
    cnts[torch.rand(6500, generator=g) < 0.35] = 0
    case = _case(6500, torch.repeat_interleave(torch.arange(6500), cnts), seed=1)
    bkgd = torch.rand(3, device=DEV)
    with nerfacc_amd.options(fold_fill=0):
        want = _render(case, 6500, bkgd)
    for _ in range(20):
        got = _render(case, 6500, bkgd)
        for a, b in zip(got, want):
            assert torch.equal(a, b)
