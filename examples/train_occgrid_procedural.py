"""End-to-end training slice on the procedural lego-like scene: the loop of the reference's
examples/train_ngp_nerf_occ.py:150-203 (occupancy-grid update, OccGridEstimator.sampling with a
sigma_fn, nerfacc.rendering, dynamic ray batch, smooth-L1, Adam) with a torch-native dense voxel
field standing in for tiny-cuda-nn and target pixels rendered from the analytic scene.

    python examples/train_occgrid_procedural.py --steps 1500

Prints PSNR on held-out rays; used by tests/test_gpu_training.py as the convergence check of the
whole forward + backward path."""
import argparse
import math
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import nerfacc  # the alias package: resolves to nerfacc_amd  # noqa: E402
from bench import AABB, RENDER_STEP, DenseGridField, make_ray_pool, render_rays  # noqa: E402


def psnr(a, b):
    return -10.0 * math.log10(F.mse_loss(a, b).item() + 1e-12)


def train(steps=1500, device="cuda:0", res=128, seed=0, log=print):
    torch.manual_seed(seed)
    teacher = DenseGridField(AABB, res).to(device).eval()           # analytic scene sampled on a grid
    student = DenseGridField(AABB, res).to(device)
    with torch.no_grad():                                           # start from fog and grey
        student.grid[:, :1].fill_(math.log(0.5))
        student.grid[:, 1:].zero_()
    est_t = nerfacc.OccGridEstimator(roi_aabb=AABB, resolution=res, levels=1).to(device)
    est = nerfacc.OccGridEstimator(roi_aabb=AABB, resolution=res, levels=1).to(device)
    est_t.train()
    for _ in range(3):
        est_t._update(step=0, occ_eval_fn=lambda x: teacher.query_density(x) * RENDER_STEP)
    bkgd = torch.ones(3, device=device)
    pool_o, pool_d = make_ray_pool(1 << 19, seed=7, device=device)
    est_t.eval()
    with torch.no_grad():
        pool_rgb = torch.cat([render_rays(teacher, est_t, pool_o[i:i + 65536], pool_d[i:i + 65536], bkgd, False)[0]
                              for i in range(0, pool_o.shape[0], 65536)])
    test_o, test_d, test_rgb = pool_o[-16384:], pool_d[-16384:], pool_rgb[-16384:]
    n_train = pool_o.shape[0] - 16384

    opt = torch.optim.Adam(student.parameters(), lr=5e-2, eps=1e-15)
    sched = torch.optim.lr_scheduler.MultiStepLR(opt, milestones=[steps // 2, steps * 3 // 4], gamma=0.33)
    num_rays, target = 4096, 1 << 18

    def evaluate():
        student.eval(), est.eval()
        with torch.no_grad():
            rgb = render_rays(student, est, test_o, test_d, bkgd, False)[0]
        student.train(), est.train()
        return psnr(rgb, test_rgb)

    est.train()
    history = [evaluate()]
    for step in range(steps):
        idx = torch.randint(0, n_train, (num_rays,), device=device)
        est.update_every_n_steps(step=step, occ_eval_fn=lambda x: student.query_density(x) * RENDER_STEP, occ_thre=1e-2)
        rgb, acc, depth, n_samples = render_rays(student, est, pool_o[idx], pool_d[idx], bkgd, True)
        if n_samples == 0:
            continue
        num_rays = max(min(int(num_rays * target / n_samples), 1 << 17), 1024)
        loss = F.smooth_l1_loss(rgb, pool_rgb[idx])
        opt.zero_grad()
        (loss * 1024.0).backward()
        opt.step()
        sched.step()
        if (step + 1) % max(steps // 5, 1) == 0:
            history.append(evaluate())
            log(f"step {step + 1:5d}  loss {loss.item():.5f}  rays {num_rays:6d}  samples {n_samples:7d}  "
                f"occupied {est.binaries.float().mean().item():.3f}  test PSNR {history[-1]:.2f} dB")
    return history


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=1500)
    ap.add_argument("--res", type=int, default=128)
    args = ap.parse_args()
    h = train(args.steps, res=args.res)
    print("PSNR history:", ["%.2f" % x for x in h])
