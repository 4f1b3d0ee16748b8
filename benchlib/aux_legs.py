"""Auxiliary legs of bench.py (never the metric; each reports under `aux` on the line): one PropNet training step (configs[2]) and the
eight-scene sweep at 256^3 (configs[4])."""
import math
import os
import sys
import time

import torch
import torch.nn.functional as F

import nerfacc_amd as nerfacc

from .profiler import profile_steps
from .scene import AABB, GRID_RES, INIT_RAYS, RENDER_STEP, TARGET_SAMPLES, DenseGridField, render_rays, render_rays_reference_style

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# ------------------------------------------------------------------------------------------
# auxiliary leg: one PropNet training step (BASELINE.json configs[2]; examples/train_ngp_nerf_prop.py:150-245 with
# examples/utils.py:170-264).  Not the metric: a few steps so that the pdf path is driver-measured too.
# ------------------------------------------------------------------------------------------
class DensityGrid(torch.nn.Module):
    """proposal network stand-in (NGPDensityField is tiny-cuda-nn): sigma = exp(trilinear lookup in one res^3 grid)"""

    def __init__(self, aabb, res):
        super().__init__()
        a = torch.tensor(aabb, dtype=torch.float32)
        self.register_buffer("u_scale", 2.0 / (a[3:] - a[:3]))
        self.register_buffer("u_shift", -2.0 * a[:3] / (a[3:] - a[:3]) - 1.0)
        self.grid = torch.nn.Parameter(torch.full((1, 1, res, res, res), math.log(0.5)))

    def forward(self, x):
        u = torch.addcmul(self.u_shift, x.reshape(-1, 3), self.u_scale).view(1, 1, 1, -1, 3)
        out = F.grid_sample(self.grid, u, mode="bilinear", padding_mode="border", align_corners=False)
        return torch.exp(out.view(*x.shape[:-1], 1))


def propnet_step_leg(field, pool_o, pool_d, pool_rgb, bkgd, n_steps, n_warmup, n_rays=4096,
                     num_samples=48, num_samples_per_prop=(256, 96), near_plane=2.0, far_plane=6.0):
    """configs[2]'s shapes (4096 rays, proposal levels of 256 and 96 samples, 48 final samples, lindisp, opaque background, two
    proposal networks) on the bench scene; the radiance field is a copy of the bench's field, the proposal networks are 64^3 /
    128^3 density grids.  A step = PropNetEstimator.sampling (2 x importance_sampling + transmittance per level, proposal
    gradients on the reference's schedule: every 5th step after its first 1000) + batched rendering + update_every_n_steps
    (searchsorted-based histogram loss, proposal optimizer) + smooth-L1 loss, backward, Adam."""
    import copy

    device = pool_o.device
    rf = copy.deepcopy(field)
    props = [DensityGrid(AABB, 64).to(device), DensityGrid(AABB, 128).to(device)]
    prop_opt = torch.optim.Adam([q for m in props for q in m.parameters()], lr=1e-2, eps=1e-15, fused=True)
    est = nerfacc.PropNetEstimator(prop_opt, None).to(device)
    opt = torch.optim.Adam(rf.parameters(), lr=1e-2, eps=1e-15, weight_decay=1e-6, fused=True)
    wants_grad = nerfacc.estimators.prop_net.get_proposal_requires_grad_fn()
    counter = {"step": 1000}                      # the schedule's steady state: proposal gradients every 5th step

    def step():
        idx = torch.randint(0, pool_o.shape[0], (n_rays,), device=device)
        o, d, pixels = pool_o[idx], pool_d[idx], pool_rgb[idx]

        def prop_sigma_fn(t_starts, t_ends, net):
            sig = net(o[:, None, :] + d[:, None, :] * (t_starts + t_ends)[..., None] / 2.0).clone()
            sig[..., -1, :] = torch.inf                                  # opaque_bkgd
            return sig.squeeze(-1)

        def rgb_sigma_fn(t_starts, t_ends, ray_indices):
            pos = o[:, None, :] + d[:, None, :] * (t_starts + t_ends)[..., None] / 2.0
            rgb, sig = rf(pos.reshape(-1, 3))
            rgb, sig = rgb.reshape(*pos.shape[:-1], 3), sig.reshape(*pos.shape[:-1], 1).clone()
            sig[..., -1, :] = torch.inf
            return rgb, sig.squeeze(-1)

        req = wants_grad(counter["step"])
        t_starts, t_ends = est.sampling(prop_sigma_fns=[lambda *a, n=n: prop_sigma_fn(*a, n) for n in props],
                                        prop_samples=list(num_samples_per_prop), num_samples=num_samples, n_rays=n_rays,
                                        near_plane=near_plane, far_plane=far_plane, sampling_type="lindisp", stratified=True,
                                        requires_grad=req)
        rgb, _, _, extras = nerfacc.rendering(t_starts, t_ends, ray_indices=None, n_rays=None, rgb_sigma_fn=rgb_sigma_fn,
                                              render_bkgd=bkgd)
        est.update_every_n_steps(extras["trans"], req, loss_scaler=1024)
        loss = F.smooth_l1_loss(rgb, pixels)
        opt.zero_grad()
        (loss * 1024.0).backward()
        opt.step()
        counter["step"] += 1

    for _ in range(n_warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n_steps):
        step()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    prof = profile_steps(step, min(n_steps, 10))
    per_ray = num_samples + sum(num_samples_per_prop)
    out = {"workload": f"configs[2] shapes on the bench scene: {n_rays} rays x proposal levels {list(num_samples_per_prop)} -> {num_samples} "
                       "samples, lindisp, opaque background, 2 proposal density grids (64^3, 128^3) + the bench's field; proposal "
                       "gradients every 5th step (the reference schedule's steady state)",
           "steps": n_steps, "ms_per_step": el / n_steps * 1e3, "rays_per_sec": n_rays * n_steps / el,
           "samples_per_sec": n_rays * num_samples * n_steps / el, "field_queries_per_sec": n_rays * per_ray * n_steps / el}
    if prof is not None and "error" not in prof:
        out["path_us_per_step"] = prof["nfa_us_per_step"]
        out["gpu_busy_us_per_step"] = prof["busy_us_per_step"]
        out["top_kernels_us_per_step"] = prof["top_kernels_us_per_step"]
    return out


# ------------------------------------------------------------------------------------------
# GPU activity of a few profiled steps: union of kernel intervals (idle fraction) and the nfa:: share
# ------------------------------------------------------------------------------------------
def scene_sweep_leg(pool_o, pool_d, bkgd, n_steps, n_warmup, pretrain=300, occ_res=256, n_pool=1 << 18):
    """BASELINE.json configs[4] (the reference's 8-scene nerf_synthetic sweep with a 256^3 occupancy grid, PSNR + rays/s per scene,
    docs/source/examples/static/ngp.rst:36-42) on stand-ins: the eight procedural scenes of tools/scenes.py — thin structures, a
    dense slab, a hollow shell, a near-empty grid, the reference's rand > 0.5 noise, ... — each with its own teacher field, a student
    trained from fog for `pretrain` steps of the configs[1] loop (256^3 occupancy grid, adaptive batch towards 2^18 samples), then
    `n_steps` timed steps: rays/s, samples/s, samples per ray, PSNR on held-out rays.  Not the metric: what it guards is that the
    data-dependent plan switches of the sampling call (tools/scene_sweep.py checks them kernel by kernel) hold up across scenes."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import scenes as S

    device = pool_o.device
    n_pool = min(n_pool, pool_o.shape[0])
    n_held = min(16384, n_pool // 4)
    pool_o, pool_d = pool_o[:n_pool], pool_d[:n_pool]
    held = slice(n_pool - n_held, n_pool)                    # rays the students never train on
    out = {}
    for name, occ_fn in S.SCENES.items():
        torch.manual_seed(7)
        teacher = DenseGridField(AABB, GRID_RES, occ_fn=lambda x, f=occ_fn: f(torch, x)).to(device).eval()
        student = DenseGridField(AABB, GRID_RES).to(device)
        with torch.no_grad():
            student.grid[:, :1].fill_(math.log(0.5))
            student.grid[:, 1:].zero_()
        est_t = nerfacc.OccGridEstimator(roi_aabb=AABB, resolution=occ_res, levels=1).to(device)
        est = nerfacc.OccGridEstimator(roi_aabb=AABB, resolution=occ_res, levels=1).to(device)
        est_t.train()
        for _ in range(4):
            est_t._update(step=0, occ_eval_fn=lambda x: teacher.query_density(x) * RENDER_STEP, occ_thre=1e-2)
        est_t.eval()
        with torch.no_grad():
            rgb_pool = torch.cat([render_rays(teacher, est_t, pool_o[i:i + (1 << 16)], pool_d[i:i + (1 << 16)], bkgd, False)[0]
                                  for i in range(0, n_pool, 1 << 16)])
        opt = torch.optim.Adam(student.parameters(), lr=1e-2, eps=1e-15, weight_decay=1e-6, fused=True)
        est.train()
        st = {"rays": INIT_RAYS, "step": 0, "n": 0, "s": 0}

        def step():
            est.update_every_n_steps(step=st["step"], occ_eval_fn=lambda x: student.query_density(x) * RENDER_STEP, occ_thre=1e-2)
            idx = torch.randint(0, n_pool - n_held, (st["rays"],), device=device)
            rgb, _, _, n_s = render_rays_reference_style(student, est, pool_o[idx], pool_d[idx], bkgd, True)
            opt.zero_grad()
            if n_s > 0:
                (F.smooth_l1_loss(rgb, rgb_pool[idx]) * 1024.0).backward()
                opt.step()
                st["rays"] = min(max(int(st["rays"] * (TARGET_SAMPLES / n_s)), 64), n_pool - n_held)      # train_ngp_nerf_occ.py:187-194
            st["n"] += idx.shape[0]
            st["s"] += n_s
            st["step"] += 1

        for _ in range(pretrain + n_warmup):
            step()
        st.update(n=0, s=0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n_steps):
            step()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        est.eval()
        with torch.no_grad():
            pred = render_rays(student, est, pool_o[held], pool_d[held], bkgd, False)[0]
            mse = F.mse_loss(pred, rgb_pool[held]).item()
        out[name] = {"ms_per_step": dt / n_steps * 1e3, "rays_per_sec": st["n"] / dt, "samples_per_sec": st["s"] / dt,
                     "rays_per_iter": st["n"] / n_steps, "samples_per_ray": st["s"] / max(st["n"], 1),
                     "occupied_fraction": est.binaries.float().mean().item(), "psnr_heldout": -10.0 * math.log10(max(mse, 1e-12))}
    return {"workload": f"configs[4] stand-in: eight procedural scenes (tools/scenes.py), {occ_res}^3 occupancy grid, the configs[1] step; "
                        f"{pretrain} training steps from fog, then {n_steps} timed steps; PSNR against the scene's teacher on {n_held} held-out rays",
            "scenes": out}
