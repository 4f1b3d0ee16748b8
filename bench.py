"""bench.py — training rays/s + samples/s of the OccGrid sampling + rendering hot path.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N > 1 launched through
torch.distributed.run, one rank per GPU (RCCL).  W untimed steps, exactly K timed steps between
barrier + synchronize, max over ranks, ONE JSON line from rank 0.

Workload = BASELINE.json configs[1] (Instant-NGP + OccGridEstimator on nerf_synthetic/lego,
examples/train_ngp_nerf_occ.py) with the two things that do not exist on this machine
replaced and said so in the output (`data`, `config.workload`):
  * dataset  -> procedural "lego-like" scene (union of boxes inside +-1.0 of the +-1.5 aabb),
    100 cameras on a radius-4 sphere, 800x800, focal 1111.1 (nerf_synthetic.py:46-48,68-69),
    white background; target pixels are rendered from the frozen initial field;
  * tiny-cuda-nn hash-grid field -> a torch-native dense voxel field (128^3 density + colour
    grids, trilinear `grid_sample`), 8.4 M parameters.
Everything else follows the script: 128^3 occupancy grid on aabb +-1.5 refreshed every 16
steps, render_step_size 5e-3, stratified sampling, early_stop_eps 1e-4, alpha_thre 0,
rays/iter adapted so that ~2^18 samples are rendered per iteration
(train_ngp_nerf_occ.py:58-78,166-203), Adam(lr 1e-2, eps 1e-15), grad scaler 2^10,
smooth-L1 loss.  A step = update_every_n_steps + sampling (traversal, sigma_fn, visibility
filter) + rendering forward + backward + optimizer step.

Issue order: the traversal of a step's rays depends on the occupancy grid but not on the field's
parameters, so the NEXT step's rays are drawn and traversed on a side HIP stream right after this
step's backward pass has been queued (it overlaps with the backward kernels; `--no-overlap` keeps
everything on one stream in program order).  The work per step is the same either way.

Multi-GPU: each rank draws its own rays (weak scaling: per-GPU work is fixed),
one flat all-reduce of the field gradients + one 16-byte all-reduce of the step's counts.
"""
import argparse
import json
import math
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import nerfacc_amd as nerfacc  # noqa: E402
from nerfacc_amd import sharding  # noqa: E402
from nerfacc_amd.cuda import _backend  # noqa: E402

AABB = [-1.5, -1.5, -1.5, 1.5, 1.5, 1.5]
RENDER_STEP = 5e-3
TARGET_SAMPLES = 1 << 18
INIT_RAYS = 1024
GRID_RES = 128
HBM_PEAK_GBS = 8000.0


# ------------------------------------------------------------------------------------------
# procedural scene + torch-native field (stand-ins for nerf_synthetic/lego and tiny-cuda-nn)
# ------------------------------------------------------------------------------------------
def lego_like_density(x: torch.Tensor) -> torch.Tensor:
    """analytic occupancy of a bulldozer-ish union of boxes; x [..., 3] in world units -> bool"""
    def box(c, h):
        c = torch.tensor(c, device=x.device)
        h = torch.tensor(h, device=x.device)
        return ((x - c).abs() <= h).all(-1)

    body = box([0.0, 0.0, -0.25], [0.75, 0.45, 0.2])
    cabin = box([-0.25, 0.0, 0.2], [0.3, 0.35, 0.25]) & ~box([-0.25, 0.0, 0.25], [0.22, 0.4, 0.12])
    plate = box([0.0, 0.0, -0.55], [0.95, 0.7, 0.06])
    arm = box([0.65, 0.0, 0.1], [0.35, 0.08, 0.08]) | box([0.95, 0.0, -0.1], [0.06, 0.4, 0.25])
    studs = (torch.sin(x[..., 0] * 24.0) * torch.sin(x[..., 1] * 24.0) > 0.5) & box([0.0, 0.0, -0.45], [0.9, 0.65, 0.05])
    return body | cabin | plate | arm | studs


class DenseGridField(torch.nn.Module):
    """sigma = exp(g[0](x)), rgb = sigmoid(g[1:4](x)); one 4-channel voxel grid, trilinear lookups
    (one gather pass forward, one scatter pass backward per query)."""

    def __init__(self, aabb, res=128):
        super().__init__()
        self.register_buffer("aabb", torch.tensor(aabb, dtype=torch.float32))
        g = (torch.arange(res, dtype=torch.float32) + 0.5) / res
        lo, hi = self.aabb[:3], self.aabb[3:]
        X, Y, Z = torch.meshgrid(g, g, g, indexing="ij")
        pts = torch.stack([X, Y, Z], -1) * (hi - lo) + lo
        occ = lego_like_density(pts)
        dens = torch.where(occ, math.log(50.0), math.log(1e-4)).float()
        gen = torch.Generator().manual_seed(42)
        col = torch.randn((3, res, res, res), generator=gen) * 0.5 + (pts.permute(3, 0, 1, 2) * 1.5)
        # stored [1, 4, Z, Y, X] so that grid_sample's (x, y, z) coordinate order needs no shuffle
        vol = torch.cat([dens[None], col], 0).permute(0, 3, 2, 1)
        self.grid = torch.nn.Parameter(vol[None].contiguous())
        self.register_buffer("u_scale", 2.0 / (hi - lo))
        self.register_buffer("u_shift", -2.0 * lo / (hi - lo) - 1.0)

    def _lookup(self, grid, x):
        u = torch.addcmul(self.u_shift, x, self.u_scale).view(1, 1, 1, -1, 3)
        out = F.grid_sample(grid, u, mode="bilinear", padding_mode="border", align_corners=False)
        return out.view(grid.shape[1], -1).t()

    def query_density(self, x):
        return torch.exp(self._lookup(self.grid[:, :1], x))

    def forward(self, x, dirs=None):
        f = self._lookup(self.grid, x)
        return torch.sigmoid(f[:, 1:4]), torch.exp(f[:, :1])


def make_ray_pool(n_pool: int, seed: int, device) -> tuple:
    """random pixels of 100 cameras on a radius-4 sphere looking at the origin (OpenGL camera,
    800x800, focal 1111.1)."""
    gen = torch.Generator(device="cpu").manual_seed(seed)
    n_cams, W, focal = 100, 800, 0.5 * 800 / math.tan(0.5 * 0.6911112070083618)
    cam_pos = torch.randn((n_cams, 3), generator=gen)
    cam_pos[:, 2] = cam_pos[:, 2].abs() * 0.7 + 0.2                 # upper hemisphere like the dataset
    cam_pos = 4.0 * cam_pos / cam_pos.norm(dim=-1, keepdim=True)
    fwd = -cam_pos / cam_pos.norm(dim=-1, keepdim=True)
    up = torch.tensor([0.0, 0.0, 1.0]).expand_as(fwd)
    right = torch.linalg.cross(fwd, up)
    right = right / right.norm(dim=-1, keepdim=True)
    true_up = torch.linalg.cross(right, fwd)
    cam = torch.randint(0, n_cams, (n_pool,), generator=gen)
    px = torch.randint(0, W, (n_pool, 2), generator=gen).float() + 0.5
    dx, dy = (px[:, 0] - W / 2) / focal, -(px[:, 1] - W / 2) / focal
    d = fwd[cam] + dx[:, None] * right[cam] + dy[:, None] * true_up[cam]
    d = d / d.norm(dim=-1, keepdim=True)
    return cam_pos[cam].contiguous().to(device), d.contiguous().to(device)


def render_rays(field, est, rays_o, rays_d, bkgd, training: bool):
    """examples/utils.py:54-167 (render_image_with_occgrid), one chunk."""
    def sigma_fn(t_starts, t_ends, ray_indices):
        if t_starts.shape[0] == 0:
            return torch.empty((0,), device=t_starts.device)
        pos = nerfacc.sample_positions(rays_o, rays_d, ray_indices, t_starts, t_ends)
        return field.query_density(pos).squeeze(-1)

    def rgb_sigma_fn(t_starts, t_ends, ray_indices):
        if t_starts.shape[0] == 0:
            return torch.empty((0, 3), device=t_starts.device), torch.empty((0,), device=t_starts.device)
        pos = nerfacc.sample_positions(rays_o, rays_d, ray_indices, t_starts, t_ends)
        rgb, sigma = field(pos)            # (this stand-in field has no view dependence)
        return rgb, sigma.squeeze(-1)

    ray_indices, t_starts, t_ends = est.sampling(rays_o, rays_d, sigma_fn=sigma_fn, near_plane=0.0, far_plane=1e10,
                                                 render_step_size=RENDER_STEP, stratified=training, cone_angle=0.0,
                                                 alpha_thre=0.0)
    rgb, opacity, depth, _ = nerfacc.rendering(t_starts, t_ends, ray_indices, n_rays=rays_o.shape[0],
                                               rgb_sigma_fn=rgb_sigma_fn, render_bkgd=bkgd)
    return rgb, opacity, depth, t_starts.shape[0]


# ------------------------------------------------------------------------------------------
# CPU baseline: the oracle (single-threaded C port of the reference algorithm) on a bounded
# sample of the same workload.  Only this function touches oracle/.
# ------------------------------------------------------------------------------------------
def cpu_baseline(field, est, pool_o, pool_d, n_rays=16384, budget_s=20.0):
    """oracle (C port of the reference algorithm) on the host: one thread on `n_rays` rays, then all
    host cores on a proportionally larger sample (ctypes releases the GIL: one Python thread per
    core, each running the whole per-ray pipeline on its own contiguous slice of rays)."""
    import concurrent.futures
    import oracle

    binaries = est.binaries.cpu().numpy()
    aabbs = est.aabbs.cpu().numpy()
    field_cpu = DenseGridField(AABB, GRID_RES)
    field_cpu.load_state_dict({k: v.cpu() for k, v in field.state_dict().items()})
    torch.set_num_threads(1)            # the field is excluded from the timing; keep it from oversubscribing

    def once(o, d):
        """returns (seconds spent outside the radiance field, rendered samples)"""
        R = o.shape[0]
        t_field = 0.0
        t0 = time.perf_counter()
        iv, sm, _ = oracle.traverse_grids(o, d, binaries, aabbs, np.zeros(R, np.float32),
                                          np.full(R, 1e10, np.float32), RENDER_STEP, 0.0)
        ts, te, ri = iv["vals"][iv["is_left"]], iv["vals"][iv["is_right"]], sm["ray_indices"]
        tf = time.perf_counter()
        with torch.no_grad():
            pos = torch.from_numpy(o[ri] + d[ri] * ((ts + te)[:, None] / 2.0))
            sig = field_cpu.query_density(pos).squeeze(-1).numpy()
        t_field += time.perf_counter() - tf
        _, T, a = oracle.render_weight_from_density(ts, te, sig, ri)
        keep = oracle.visibility(T, a, 1e-4, 0.0)
        ri, ts, te = ri[keep], ts[keep], te[keep]
        tf = time.perf_counter()
        with torch.no_grad():
            pos = torch.from_numpy(o[ri] + d[ri] * ((ts + te)[:, None] / 2.0))
            rgb, sig = field_cpu(pos)
            rgb, sig = rgb.numpy(), sig.squeeze(-1).numpy()
        t_field += time.perf_counter() - tf
        col, opa, dep, ex = oracle.rendering(ts, te, ri, R, sig, rgb, np.ones(3, np.float32))
        gw = np.ascontiguousarray((rgb * col[ri]).sum(-1).astype(np.float32))      # stand-in for dL/dw
        oracle.render_weight_from_density_bwd(ts, te, sig, ri, g_w=gw)
        return time.perf_counter() - t0 - t_field, ri.shape[0]

    # ---- one thread
    o1 = pool_o[:n_rays].cpu().numpy()
    d1 = pool_d[:n_rays].cpu().numpy()
    once(o1, d1)
    times, n_s = [], 0
    t_begin = time.perf_counter()
    while len(times) < 10 and time.perf_counter() - t_begin < budget_s / 2:
        dt, n_s = once(o1, d1)
        times.append(dt)
    med1 = float(np.median(times))

    # ---- all cores: `per` rays per thread, slices of the same pool
    cores = os.cpu_count() or 1
    per = 2048
    n_all = min(cores * per, pool_o.shape[0])
    oa = pool_o[:n_all].cpu().numpy()
    da = pool_d[:n_all].cpu().numpy()
    slices = [(i, min(i + per, n_all)) for i in range(0, n_all, per)]

    def run_all(pool):
        t0 = time.perf_counter()
        res = list(pool.map(lambda s: once(oa[s[0]:s[1]], da[s[0]:s[1]]), slices))
        wall = time.perf_counter() - t0       # (the slices' field queries run concurrently and stay in)
        return wall, sum(r[1] for r in res)

    with concurrent.futures.ThreadPoolExecutor(max_workers=len(slices)) as pool:
        run_all(pool)
        walls = []
        t_begin = time.perf_counter()
        while len(walls) < 5 and time.perf_counter() - t_begin < budget_s / 2:
            walls.append(run_all(pool))
    wall, n_s_all = sorted(walls)[len(walls) // 2]
    return {
        "value": n_all / wall, "unit": "rays/s", "cores": len(slices), "kind": "port",
        "samples_per_sec": n_s_all / wall,
        "sample": f"{n_all} rays of the same pool/grid in {len(slices)} slices, one thread per host core: oracle traversal + "
                  f"visibility + rendering fwd + weight bwd per slice, wall clock of the whole batch (includes the "
                  f"threads' torch-CPU field queries), median of {len(walls)}",
        "single_thread": {"value": n_rays / med1, "unit": "rays/s", "samples_per_sec": n_s / med1,
                          "sample": f"{n_rays} rays, median of {len(times)}, radiance-field evaluation excluded"},
        "host_cores_available": cores,
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--pool", type=int, default=1 << 21)
    ap.add_argument("--no-overlap", action="store_true",
                    help="issue the next step's ray traversal after the optimizer on the main stream instead of "
                         "on a side stream concurrently with the backward pass")
    ap.add_argument("--occ-res", type=int, default=GRID_RES,
                    help="occupancy-grid resolution: 128 = configs[1] (default), 256 = the configs[4] grid size")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world_size = int(os.environ.get("WORLD_SIZE", "1"))
    assert world_size == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world_size}"
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world_size > 1:
        dist.init_process_group("nccl", device_id=device)

    torch.manual_seed(42)
    field = DenseGridField(AABB, GRID_RES).to(device)
    teacher = DenseGridField(AABB, GRID_RES).to(device).eval()
    est = nerfacc.OccGridEstimator(roi_aabb=AABB, resolution=args.occ_res, levels=1).to(device)
    optimizer = torch.optim.Adam(field.parameters(), lr=1e-2, eps=1e-15, weight_decay=1e-6, fused=True)
    loss_scale = 2.0**10
    bkgd = torch.ones(3, device=device)

    def occ_eval_fn(x):
        return field.query_density(x) * RENDER_STEP

    est.train()
    with sharding.synchronized_rng(1234, device):
        for _ in range(4):                       # bring the grid to its steady state before timing
            est._update(step=0, occ_eval_fn=occ_eval_fn, occ_thre=1e-2)

    pool_o, pool_d = make_ray_pool(args.pool, seed=42 + rank, device=device)
    est.eval()
    with torch.no_grad():                        # target pixels from the frozen initial field
        pool_rgb = torch.empty((args.pool, 3), device=device)
        for i in range(0, args.pool, 1 << 16):
            rgb, _, _, _ = render_rays(teacher, est, pool_o[i:i + (1 << 16)], pool_d[i:i + (1 << 16)], bkgd, False)
            pool_rgb[i:i + (1 << 16)] = rgb
    est.train()
    torch.manual_seed(1000 + rank)

    # steady state of the 20 k-step schedule: past `warmup_steps` (256) the grid update evaluates a
    # quarter of the cells + the occupied ones instead of all 2 M cells (occ_grid.py:372-376)
    state = {"num_rays": INIT_RAYS, "step": 1024}
    stats = {"rays": 0, "samples": 0, "candidates": 0}

    # The traversal of a step (rays -> candidate samples) does not depend on the field's parameters, only
    # on the occupancy grid.  So the NEXT step's rays are drawn and traversed on a side stream right
    # after this step's backward pass has been queued: the count kernel and its host read-back (the
    # first of the two host syncs of a step) overlap with the backward kernels instead of waiting
    # behind them, and the host can queue the sigma_fn / filter launches while the GPU is still busy.
    # Steps that refresh the grid (every 16th) traverse after the refresh, on the main stream's heels.
    side = torch.cuda.Stream(device=device)
    overlap = not args.no_overlap

    def propose(n, wait_for_main):
        main = torch.cuda.current_stream(device)
        if wait_for_main:
            side.wait_stream(main)
        with torch.cuda.stream(side):
            idx = torch.randint(0, args.pool, (n,), device=device)
            rays_o, rays_d, pixels = pool_o[idx], pool_d[idx], pool_rgb[idx]
            near = torch.rand(n, device=device) * RENDER_STEP          # near_plane 0 + stratified jitter (occ_grid.py:162-163)
            far = torch.full((n,), 1e10, device=device)
            ri, ts, te, _ = nerfacc.cuda.sample_occgrid(rays_o, rays_d, est.binaries, est.aabbs, near, far, RENDER_STEP, 0.0)
        return dict(n=n, rays_o=rays_o, rays_d=rays_d, pixels=pixels, ri=ri, ts=ts, te=te)

    def render_proposed(prop):
        """the rest of OccGridEstimator.sampling (visibility filter, occ_grid.py:180-220) + rendering, on the main stream"""
        main = torch.cuda.current_stream(device)
        main.wait_stream(side)
        for t in prop.values():
            if torch.is_tensor(t):
                t.record_stream(main)
        rays_o, rays_d, ri, ts, te = prop["rays_o"], prop["rays_d"], prop["ri"], prop["ts"], prop["te"]
        if ts.shape[0] > 0:
            with torch.no_grad():
                sig = field.query_density(nerfacc.sample_positions(rays_o, rays_d, ri, ts, te)).squeeze(-1)
            ri, ts, te, _ = nerfacc.cuda.visibility_compact(ri, ts, te, sig.contiguous(), False, 1e-4, 0.0)

        def rgb_sigma_fn(t_starts, t_ends, ray_indices):
            if t_starts.shape[0] == 0:
                return torch.empty((0, 3), device=device), torch.empty((0,), device=device)
            rgb, sigma = field(nerfacc.sample_positions(rays_o, rays_d, ray_indices, t_starts, t_ends))
            return rgb, sigma.squeeze(-1)

        rgb, opacity, depth, _ = nerfacc.rendering(ts, te, ri, n_rays=prop["n"], rgb_sigma_fn=rgb_sigma_fn, render_bkgd=bkgd)
        return rgb, ts.shape[0]

    def train_step():
        step = state["step"]
        refresh = step % 16 == 0
        if refresh:
            with sharding.synchronized_rng(5000 + step, device):
                est.update_every_n_steps(step=step, occ_eval_fn=occ_eval_fn, occ_thre=1e-2)
        prop = state.pop("proposal", None)
        if prop is None:
            prop = propose(state["num_rays"], wait_for_main=True)
        n = prop["n"]
        rgb, n_samples = render_proposed(prop)
        # global (samples, rays) of this step, so that all ranks stay in step (train_ngp_nerf_occ.py:187-194)
        pending = sharding.allreduce_counts_begin(n_samples, n, device)
        optimizer.zero_grad()
        if n_samples > 0:
            loss = F.smooth_l1_loss(rgb, prop["pixels"])
            (loss * loss_scale).backward()
        # every rank takes part in the exchange every step, samples or not (a rank that skipped
        # the collective would deadlock the others); missing grads count as zeros
        sharding.allreduce_gradients(field.parameters())
        if n_samples > 0 or world_size > 1:
            optimizer.step()
        with torch.cuda.stream(side):                 # the 16-byte result is read without waiting for the backward pass
            g_samples, g_rays = sharding.allreduce_counts_end(pending)
        if g_samples > 0:
            state["num_rays"] = max(int((g_rays / world_size) * (TARGET_SAMPLES / (g_samples / world_size))), 64)
        if overlap and (step + 1) % 16 != 0:
            state["proposal"] = propose(state["num_rays"], wait_for_main=False)
        stats["rays"] += n
        stats["samples"] += n_samples
        state["step"] += 1

    for _ in range(args.warmup):
        train_step()

    timer = _backend.KernelTimer(names=("traverse_count", "traverse_fill"))
    _backend.set_kernel_timer(timer)
    stats.update(rays=0, samples=0)
    if world_size > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        train_step()
    torch.cuda.synchronize()
    if world_size > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    _backend.set_kernel_timer(None)

    tot = torch.tensor([elapsed, float(stats["rays"]), float(stats["samples"])], dtype=torch.float64, device=device)
    if world_size > 1:
        mx = tot.clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        elapsed = mx[0].item()
    total_rays, total_samples = tot[1].item(), tot[2].item()

    if rank == 0:
        # roofline of the dominant kernel of OUR path (profiles/): the traversal count pass
        # (traverse_count_split_kernel at this ray count), one launch per step, timed with HIP
        # events on the launch stream around its single-kernel C-ABI call; the emit pass
        # (traverse_emit_kernel) is timed the same way and its time is charged too, because the
        # algorithmic bytes below are those of the whole sampling traversal:
        # 16 B per emitted candidate sample (ray_indices i64 + t_starts + t_ends) + 48 B per ray
        # (origin, direction, near, far, start, count) + the grid once — at the size this
        # implementation reads it (V/8 bytes bit-packed instead of V bool bytes).
        summ = timer.summary()
        n_launch, ms_count = summ.get("traverse_count", (0, 0.0))
        _, ms_emit = summ.get("traverse_fill", (0, 0.0))
        ms = ms_count + ms_emit
        rays_per_launch = stats["rays"] / max(args.steps, 1)
        with torch.no_grad():
            idx = torch.randint(0, args.pool, (int(rays_per_launch),), device=device)
            cand = est.sampling(pool_o[idx], pool_d[idx], render_step_size=RENDER_STEP, stratified=True)[0].shape[0]
        alg_bytes = 16.0 * cand + 48.0 * rays_per_launch + args.occ_res**3 / 8
        achieved = alg_bytes / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
        out = {
            "metric": "training rays/sec (+ samples/sec), NGP+OccGrid Lego 800x800",
            "value": total_rays / elapsed,
            "unit": "rays/s",
            "samples_per_sec": total_samples / elapsed,
            "n_gpus": world_size, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {
                "workload": f"configs[1] lego stand-in: procedural lego-like scene, 100 cams 800x800, {args.occ_res}^3 OccGrid on aabb +-1.5, "
                            "render_step 5e-3, ~2^18 rendered samples/iter/GPU, torch dense-grid field (tiny-cuda-nn absent)",
                "rays_per_iter_per_gpu": rays_per_launch,
                "samples_per_iter_per_gpu": stats["samples"] / max(args.steps, 1),
                "candidate_samples_per_iter": cand,
                "parallelism": f"rays sharded over {world_size} GPU(s), 1 flat grad all-reduce/step",
            },
            "roofline": {
                "kernel": "traverse_count_split_kernel (+ traverse_emit_kernel)", "bound": "hbm", "achieved": achieved,
                "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                # HBM bytes per sampling call from PMC counters (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate
                # passes, FETCH doubled per the gfx950 note of MI355X_MICROARCH.md; profiles/r01_pmc_traffic.md:
                # 7.97 MB at 13 120 rays / 328 k candidates = 1.30 x the algorithmic bytes of that call), scaled
                # to this run's algorithmic bytes
                "traffic": 1.30 * alg_bytes,
                "avg_launch_ms": ms_count, "emit_avg_launch_ms": ms_emit, "launches": n_launch,
                "algorithmic_bytes_per_launch": alg_bytes,
                "candidate_samples_per_sec_of_kernel_time": cand / (ms * 1e-3) if ms > 0 else 0.0,
                "note": "issue/latency-bound voxel walk: 16 B per sample from an LDS-resident grid, HBM fraction is small "
                        "by construction (DESIGN.md 3.2); the HBM-streaming kernels of the path reach 3.4-5.5 TB/s at N >= 2^24 "
                        "(profiles/r01_roofline_streaming.md)",
            },
        }
        if not args.no_cpu_baseline and world_size == 1:
            out["cpu_baseline"] = cpu_baseline(field, est, pool_o, pool_d)
        print(json.dumps(out))
    if world_size > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
