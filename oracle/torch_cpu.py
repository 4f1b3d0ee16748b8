"""Pure-PyTorch CPU composition of the rendering ops — TEST INFRASTRUCTURE (bench.py's cpu_baseline leg and
tests/ only; nothing under nerfacc_amd/ imports it).

BASELINE.json configs[0] names "pure-PyTorch render_weight_from_density on CPU (tests/test_rendering.py
path)".  In the reference the only branch of that function that can run on CPU tensors is the batched one
(volrend.py:266-278 with `packed_info is None and ray_indices is None` -> scan.py:73-75 `torch.cumsum`); the
flattened branch needs the CUDA extension (pack.py:47-48).  Two restatements:

  * `weights_batched`: exactly that branch on a padded [n_rays, max_samples] layout (padding has sigma = 0, so it
    contributes alpha = 0 and leaves T untouched) — what a CPU user of the reference has to do today;
  * `weights_flat`: the flattened layout without padding — flat cumsum minus the per-ray offset, the SURVEY §8d
    "CPU baseline plan" form — and `rendering_flat` (volrend.py:104-164 with `index_add_`, :547-561).

Both are checked against the C oracle in tests/test_oracle.py before anything is timed.
"""
import torch


def pad_rays(packed_info: torch.Tensor, *flat):
    """flattened samples -> [n_rays, max_cnt] tensors (zero padded) + the boolean mask of real samples"""
    starts, cnts = packed_info[:, 0], packed_info[:, 1]
    R, S = cnts.shape[0], int(cnts.max().item()) if cnts.numel() else 0
    col = torch.arange(S)[None, :]
    mask = col < cnts[:, None]
    src = (starts[:, None] + col)[mask]
    out = []
    for x in flat:
        p = torch.zeros((R, S) + tuple(x.shape[1:]), dtype=x.dtype)
        p[mask] = x[src]
        out.append(p)
    return out, mask


def weights_batched(t_starts, t_ends, sigmas):
    """volrend.py:266-278, 371-376 (batched branch): all inputs [n_rays, n_samples]"""
    sigmas_dt = sigmas * (t_ends - t_starts)
    alphas = 1.0 - torch.exp(-sigmas_dt)
    cs = torch.cumsum(sigmas_dt, dim=-1)                    # scan.py:73-75: exclusive = cumsum shifted
    trans = torch.exp(-(torch.cat([torch.zeros_like(cs[..., :1]), cs[..., :-1]], dim=-1)))
    return trans * alphas, trans, alphas


def weights_flat(t_starts, t_ends, sigmas, ray_indices, packed_info):
    """same quantities on the flattened layout: exclusive per-ray sum = flat cumsum - value - (flat cumsum before
    the ray's first sample)"""
    sigmas_dt = sigmas * (t_ends - t_starts)
    alphas = 1.0 - torch.exp(-sigmas_dt)
    cs = torch.cumsum(sigmas_dt.double(), 0)                # one long fp32 running sum would lose the small terms
    starts = packed_info[:, 0]
    before = torch.where(starts > 0, cs[(starts - 1).clamp_min(0)], torch.zeros((), dtype=cs.dtype))
    excl = (cs - sigmas_dt.double() - before[ray_indices]).float()
    trans = torch.exp(-excl)
    return trans * alphas, trans, alphas


def rendering_flat(t_starts, t_ends, sigmas, rgbs, ray_indices, packed_info, n_rays, bkgd=None):
    """volrend.py:104-164 with accumulate_along_rays = zeros.index_add_ (:547-561)"""
    w, T, a = weights_flat(t_starts, t_ends, sigmas, ray_indices, packed_info)
    colors = torch.zeros((n_rays, 3)).index_add_(0, ray_indices, w[:, None] * rgbs)
    opac = torch.zeros((n_rays, 1)).index_add_(0, ray_indices, w[:, None])
    depth = torch.zeros((n_rays, 1)).index_add_(0, ray_indices, (w * (t_starts + t_ends) / 2.0)[:, None])
    depth = depth / opac.clamp_min(torch.finfo(torch.float32).eps)
    if bkgd is not None:
        colors = colors + bkgd * (1.0 - opac)
    return colors, opac, depth, w
