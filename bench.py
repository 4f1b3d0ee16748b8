"""bench.py — training rays/s + samples/s of the OccGrid sampling + rendering hot path.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N > 1 launched through
torch.distributed.run, one rank per GPU (RCCL).  W untimed steps, exactly K timed steps between
barrier + synchronize, max over ranks, ONE JSON line from rank 0.

Workload = BASELINE.json configs[1] (Instant-NGP + OccGridEstimator on nerf_synthetic/lego,
examples/train_ngp_nerf_occ.py) with the two things that do not exist on this machine
replaced and said so in the output (`data`, `config.workload`):
  * dataset  -> procedural "lego-like" scene (union of boxes inside +-1.0 of the +-1.5 aabb),
    100 cameras on a radius-4 sphere, 800x800, focal 1111.1 (nerf_synthetic.py:46-48,68-69),
    white background; target pixels are rendered from the analytic scene sampled on a grid;
  * tiny-cuda-nn hash-grid field -> a torch-native dense voxel field (128^3 density + colour
    grids, trilinear `grid_sample`), 8.4 M parameters.
Everything else follows the script: 128^3 occupancy grid on aabb +-1.5 refreshed every 16
steps, render_step_size 5e-3, stratified sampling, early_stop_eps 1e-4, alpha_thre 0,
rays/iter adapted so that ~2^18 samples are rendered per iteration
(train_ngp_nerf_occ.py:58-78,166-203), Adam(lr 1e-2, eps 1e-15), grad scaler 2^10,
smooth-L1 loss.  A step = update_every_n_steps + sampling (traversal, sigma_fn, visibility
filter) + rendering forward + backward + optimizer step.

What is timed.  The student field starts from fog and is TRAINED, untimed, for `--pretrain` steps of the
same loop (grid warm-up included) before the W warm-up and K timed steps, so that the timed region sits
in the steady state of a real run whatever K is (rays/iter and samples/ray are printed).  Two loops exist:
  * `api` (default, the headline): the step exactly as examples/utils.py:137-155 +
    train_ngp_nerf_occ.py:166-203 write it — `estimator.sampling(sigma_fn=...)` -> `nerfacc.rendering`;
  * `overlap`: same work, but the NEXT step's rays are traversed on a side stream while this step's
    backward pass runs (the traversal depends on the occupancy grid, not on the parameters).
Both are timed in one run (K steps each, same barriers); `value` is the `--mode` one, the other is
printed beside it.  A third pass of a few profiled steps (torch.profiler, kernel intervals) gives
`path_us_per_step` (all nfa:: kernels) and `gpu_idle_frac`.

Multi-GPU: each rank draws its own rays (weak scaling: per-GPU work is fixed; with
`--rays-per-iter G` the GLOBAL batch is fixed at G rays, G / N per rank — configs[3]: 65536 / 8),
gradient all-reduce in a few async chunks overlapped with the Adam update (sharding.ExchangeAdam)
+ one 16-byte all-reduce of the step's counts.
"""
import argparse
import json
import math
import os
import sys
import time

# two OpenMP runtimes live in this process during the cpu_baseline leg (torch's bundled one and the oracle's); with
# the default active wait policy the idle team of one spins on the cores the other needs (measured: 3 s instead of
# 1 ms for a torch op on 256 threads).  Must be set before either runtime is loaded.
os.environ.setdefault("OMP_WAIT_POLICY", "passive")
os.environ.setdefault("GOMP_SPINCOUNT", "0")

import numpy as np
import torch
import torch.distributed as dist
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import nerfacc_amd as nerfacc  # noqa: E402
from nerfacc_amd import sharding  # noqa: E402
from nerfacc_amd.cuda import _backend  # noqa: E402
from benchlib.scene import (AABB, GRID_RES, HBM_PEAK_GBS, INIT_RAYS, LAST_CALL, RENDER_STEP, TARGET_SAMPLES, DenseGridField, GridMlpField,  # noqa: E402,F401
                            lego_like_density, make_ray_pool, render_rays, render_rays_reference_style)
from benchlib.aux_legs import DensityGrid, propnet_step_leg, scene_sweep_leg  # noqa: E402,F401
from benchlib.profiler import dda_steps, profile_steps  # noqa: E402,F401


# ------------------------------------------------------------------------------------------
# CPU baseline.  Only this function touches oracle/.
# ------------------------------------------------------------------------------------------
def cpu_baseline(field, est, pool_o, pool_d, budget_s=24.0):
    """The CPU side of the same workload on the host cores of this box (count printed):

    * `value` (rays/s): the C oracle (port of the reference algorithm, OpenMP over rays) on all cores, on a bounded
      sample of the bench's own ray pool and occupancy grid — traversal + visibility + rendering fwd + weight bwd;
      the radiance field's evaluation (torch CPU) runs between the stages and is EXCLUDED from the time in every leg;
    * `single_thread`: the same pipeline with one thread (the scalar checker as the tests use it);
    * `pure_torch`: BASELINE.json configs[0] — pure-PyTorch `render_weight_from_density` forward + backward on CPU
      tensors (oracle/torch_cpu.py: the reference's batched torch.cumsum branch on the padded layout, and the
      flattened cumsum form), torch.set_num_threads(all cores), on C1's shapes: 128^3 sphere grid, 4096 rays into the
      unit cube, step 1/600; 3 warm-ups, median of 10."""
    import oracle
    from oracle import torch_cpu

    cores = os.cpu_count() or 1
    binaries = est.binaries.cpu().numpy()
    aabbs = est.aabbs.cpu().numpy()
    field_cpu = DenseGridField(AABB, GRID_RES)
    field_cpu.load_state_dict({k: v.cpu() for k, v in field.state_dict().items()})
    torch.set_num_threads(cores)
    t_deadline = time.perf_counter() + budget_s

    def pipeline(o, d):
        """returns (seconds outside the radiance field, rendered samples, candidate samples)"""
        R = o.shape[0]
        t_field = 0.0
        t0 = time.perf_counter()
        ri, ts, te, pk = oracle.sample_occgrid(o, d, binaries, aabbs, np.zeros(R, np.float32), np.full(R, 1e10, np.float32),
                                               RENDER_STEP, 0.0)
        n_cand = ri.shape[0]
        tf = time.perf_counter()
        with torch.no_grad():
            pos = torch.from_numpy(o[ri] + d[ri] * ((ts + te)[:, None] / 2.0))
            sig = field_cpu.query_density(pos).squeeze(-1).numpy()
        t_field += time.perf_counter() - tf
        _, T, a = oracle.render_weight_from_density(ts, te, sig, ri)
        keep = oracle.visibility(T, a, 1e-4, 0.0)
        ri, ts, te, pk = oracle.compact(keep, ri, ts, te, pk)
        tf = time.perf_counter()
        with torch.no_grad():
            pos = torch.from_numpy(o[ri] + d[ri] * ((ts + te)[:, None] / 2.0))
            rgb, sig = field_cpu(pos)
            rgb, sig = rgb.numpy(), sig.squeeze(-1).numpy()
            gw_src = rgb
        t_field += time.perf_counter() - tf
        col, opa, dep, ex = oracle.rendering(ts, te, ri, R, sig, rgb, np.ones(3, np.float32))
        tf = time.perf_counter()
        gw = np.ascontiguousarray((gw_src * col[ri]).sum(-1).astype(np.float32))      # stand-in for dL/dw (numpy glue: excluded)
        t_field += time.perf_counter() - tf
        oracle.render_weight_from_density_bwd(ts, te, sig, ri, g_w=gw)
        return time.perf_counter() - t0 - t_field, ri.shape[0], n_cand

    def timed(o, d, reps, share):
        pipeline(o[:2048], d[:2048])
        out = []
        t_begin = time.perf_counter()
        while len(out) < reps and (not out or time.perf_counter() - t_begin < budget_s * share):
            out.append(pipeline(o, d))
        out.sort()
        return out[len(out) // 2], len(out)

    n1 = 8192
    o1, d1 = pool_o[:n1].cpu().numpy(), pool_d[:n1].cpu().numpy()
    oracle.set_threads(1)
    (t1, s1, c1), reps1 = timed(o1, d1, 5, 0.3)
    n_all = int(min(pool_o.shape[0], max(16384, int(os.environ.get('NFA_CPU_RAYS_PER_CORE', '256')) * cores)))
    oa, da = pool_o[:n_all].cpu().numpy(), pool_d[:n_all].cpu().numpy()
    used = oracle.set_threads(cores)
    try:
        (ta, sa, ca), repsa = timed(oa, da, 7, 0.3)
    finally:
        oracle.set_threads(1)

    # ---- the reference's OWN traversal (grid.cu compiled for the host, oracle/_ref; single thread, its serial kernel loop) on the
    # same n1 rays, when that build travelled with the tree: what "the reference on this CPU" means for the part of the path
    # that exists only as CUDA upstream
    ref_leg = None
    try:
        ref_dir = os.path.join(ROOT, "oracle", "_ref")
        if os.path.isdir(ref_dir) and any(f.startswith("nerfacc_ref_fma") for f in os.listdir(ref_dir)):
            sys.path.insert(0, ref_dir)
            import importlib

            ref = importlib.import_module("nerfacc_ref_fma")
            tt_ = lambda a: torch.from_numpy(np.ascontiguousarray(a))
            ro, rd, rb, ra = tt_(o1), tt_(d1), tt_(binaries), tt_(aabbs)
            torch.set_num_threads(1)
            xs = []
            for _ in range(3):
                t0 = time.perf_counter()
                tmin, tmax, hits = ref.ray_aabb_intersect(ro, rd, ra, -float("inf"), float("inf"), float("inf"))
                ts_, ti_ = torch.sort(torch.cat([tmin, tmax], -1), -1)
                iv_, sm_, _ = ref.traverse_grids(ro, rd, torch.ones(n1, dtype=torch.bool), rb, ra, ts_, ti_, hits, torch.zeros(n1),
                                                 torch.full((n1,), 1e10), RENDER_STEP, 0.0, True, True, True, -1, False)
                xs.append(time.perf_counter() - t0)
            torch.set_num_threads(cores)
            ref_leg = {"rays_per_sec": n1 / float(np.median(xs)), "candidate_samples_per_sec": int(sm_.vals.shape[0]) / float(np.median(xs)),
                       "sample": f"{n1} rays: the reference's ray_aabb_intersect + traverse_grids (count pass, cumsum, fill pass) exactly as grid.cu "
                                 "runs them, one host thread; traversal only"}
    except Exception as e:      # noqa: BLE001  (an optional leg)
        ref_leg = {"error": repr(e)[:160]}

    # ---- configs[0]: pure-PyTorch render_weight_from_density on CPU, C1 shapes
    rng = np.random.default_rng(42)
    R = 4096
    g = (np.arange(128) + 0.5) / 128
    X, Y, Z = np.meshgrid(g, g, g, indexing="ij")
    sphere = (((X - 0.5) ** 2 + (Y - 0.5) ** 2 + (Z - 0.5) ** 2) < 0.09)[None]
    o = rng.standard_normal((R, 3))
    o = (0.5 + 1.5 * o / np.linalg.norm(o, axis=-1, keepdims=True)).astype(np.float32)
    tgt = rng.random((R, 3)).astype(np.float32)
    d = tgt - o
    d = (d / np.linalg.norm(d, axis=-1, keepdims=True)).astype(np.float32)
    iv, sm, _ = oracle.traverse_grids(o, d, sphere, np.array([[0, 0, 0, 1, 1, 1]], np.float32), step_size=5e-3 / 3)
    tt = torch.from_numpy
    ts, te, ri, pk = tt(iv["vals"][iv["is_left"]]), tt(iv["vals"][iv["is_right"]]), tt(sm["ray_indices"]), tt(sm["packed_info"])
    N = ts.shape[0]
    sig = tt((rng.random(N) * 20).astype(np.float32))
    (p_ts, p_te, p_sig), mask = torch_cpu.pad_rays(pk, ts, te, sig)

    def median_ms(fn, cap_s=2.5):
        fn()
        xs, t_begin = [], time.perf_counter()
        while len(xs) < 10 and (len(xs) < 2 or time.perf_counter() - t_begin < cap_s):
            t0 = time.perf_counter()
            fn()
            xs.append(time.perf_counter() - t0)
        return float(np.median(xs)) * 1e3

    def batched_fwd_bwd():
        s = p_sig.clone().requires_grad_(True)
        w, _, _ = torch_cpu.weights_batched(p_ts, p_te, s)
        w.sum().backward()

    def flat_fwd_bwd():
        s = sig.clone().requires_grad_(True)
        w, _, _ = torch_cpu.weights_flat(ts, te, s, ri, pk)
        w.sum().backward()

    # torch's intra-op pool does not scale to hundreds of threads on tensors this small: sweep and report all
    sweep = {}
    for nt in sorted({1, 8, 32, cores}):
        if nt > cores:
            continue
        torch.set_num_threads(nt)
        sweep[nt] = (median_ms(batched_fwd_bwd), median_ms(flat_fwd_bwd))
    torch.set_num_threads(cores)
    best_b = min(sweep, key=lambda k: sweep[k][0])
    best_f = min(sweep, key=lambda k: sweep[k][1])
    ms_b, ms_f = sweep[best_b][0], sweep[best_f][1]
    alg = (32.0 + 28.0) * N                       # weights fwd 32 N + bwd 28 N bytes
    return {
        "value": n_all / ta, "unit": "rays/s", "cores": used, "kind": "port",
        "samples_per_sec": sa / ta, "candidate_samples_per_sec": ca / ta,
        "sample": f"{n_all} rays of the bench's pool and occupancy grid, C oracle with OpenMP over rays on {used} threads: traversal + "
                  f"visibility + rendering fwd + weight bwd; radiance-field evaluation excluded; median of {repsa} runs "
                  f"({ta * 1e3:.1f} ms each)",
        "single_thread": {"value": n1 / t1, "unit": "rays/s", "samples_per_sec": s1 / t1,
                          "sample": f"{n1} rays, same pipeline, 1 thread, median of {reps1}, radiance-field evaluation excluded"},
        "all_cores_over_single_thread": (n_all / ta) / (n1 / t1),
        "scaling_note": "the C stages are OpenMP loops over rays; what does not scale is between them: numpy glue (event sort, cumsum, "
                        "dtype casts), waking the thread team for each of ~8 stages, and first-touch page faults of freshly mapped output "
                        "arrays, which serialise in the kernel — a few tens of ms per run against ~2 ms of parallel work at this sample size; "
                        "larger samples shift the cost to page faults (262144 rays: 326 ms per run, 7x over one thread)",
        "pure_torch": {
            "workload": f"configs[0]: 128^3 sphere grid, {R} rays into the unit cube, step 1/600 -> {N} samples "
                        f"(max {int(pk[:, 1].max())} per ray); render_weight_from_density fwd+bwd, torch CPU, warm-up + median of up to 10, "
                        f"best thread count of the sweep",
            "batched_padded_ms": ms_b, "batched_threads": best_b, "flat_ms": ms_f, "flat_threads": best_f,
            "ms_by_threads": {str(k): {"batched_padded": round(v[0], 2), "flat": round(v[1], 2)} for k, v in sweep.items()},
            "batched_samples_per_sec": N / (ms_b * 1e-3), "flat_samples_per_sec": N / (ms_f * 1e-3),
            "flat_GBps_algorithmic": alg / (ms_f * 1e-3) / 1e9,
            "note": "batched = the reference's only CPU-runnable branch (volrend.py:266-278 on a padded [rays, max_samples] layout, "
                    f"{p_sig.numel()} padded elements for {N} samples); flat = cumsum minus per-ray offset on the packed layout",
        },
        "host_cores_available": cores,
        "reference_host_build": ref_leg,
    }


def main():
    """stdout carries the ONE JSON line and nothing else: everything the run prints on file descriptor 1 on the way — RCCL's
    version banner at the first communicator (C stdio, the rank-step leg creates one even at N = 1), library warnings — goes to
    stderr; rank 0's line is written when the run is over."""
    import ctypes
    args = parse_args()          # (`--gpus N` outside a launcher re-launches under torch.distributed.run here, with stdout untouched)
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    line = None
    try:
        line = run(args)
    except BaseException:
        if int(os.environ.get("WORLD_SIZE", "1")) > 1:
            # a rank of a multi-rank job fails FAST: traceback, then the process ends without running the process group's teardown
            # (a collective the other ranks will never join) — the launcher sees the exit code and ends the job.  Seen as a
            # half-hour hang (gloo's timeout) when one rank of eight raised from sample_occgrid (profiles/r06_oversubscription.md)
            import traceback
            traceback.print_exc()
            sys.stderr.flush()
            os._exit(1)
        raise
    finally:
        sys.stdout.flush()
        try:
            ctypes.CDLL(None).fflush(None)
        except Exception:       # noqa: BLE001
            pass
        os.dup2(real_stdout, 1)
        os.close(real_stdout)
    if line is not None:
        print(line, flush=True)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--mode", choices=("api", "overlap"), default="api",
                    help="which loop `value` is quoted on: api = the reference examples' step through the public API "
                         "(estimator.sampling -> nerfacc.rendering); overlap = same work, next step's traversal on a side stream")
    ap.add_argument("--pretrain", type=int, default=1500, help="untimed training steps from the fog initialisation before warm-up")
    ap.add_argument("--rays-per-iter", type=int, default=0,
                    help="fixed GLOBAL rays per iteration (configs[3]: 65536, i.e. 8192 per rank on 8 GPUs); 0 = the script's "
                         "adaptive batch targeting 2^18 rendered samples per iteration per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-mode", action="store_true", help="time only --mode")
    ap.add_argument("--no-profile", action="store_true", help="skip the profiled pass (gpu_idle_frac, path_us_per_step)")
    ap.add_argument("--pool", type=int, default=1 << 21)
    ap.add_argument("--occ-res", type=int, default=GRID_RES,
                    help="occupancy-grid resolution: 128 = configs[1] (default), 256 = the configs[4] grid size")
    ap.add_argument("--grad-chunks", type=int, default=4, help="chunks of the gradient all-reduce (N > 1)")
    ap.add_argument("--exchange-mode", choices=("allreduce", "rs_ag"), default="allreduce",
                    help="gradient exchange of ExchangeAdam (N > 1): chunked all-reduce + Adam on every rank, or reduce-scatter -> Adam on the "
                         "local 1/N -> all-gather of the parameters; the OTHER mode is timed for a few steps as aux.exchange_modes")
    ap.add_argument("--field", choices=("grid", "grid+mlp"), default="grid",
                    help="grid = the headline's dense-grid field (one parameter tensor); grid+mlp = feature grid + two-layer MLP (seven "
                         "tensors: exercises the gradient hooks and the chunk order of the exchange on a multi-tensor graph; dry runs)")
    ap.add_argument("--dist-backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL; gloo for dry runs)")
    ap.add_argument("--all-ranks-on-device0", action="store_true",
                    help="dry run of the N > 1 code path on a single GPU (with --dist-backend gloo): every rank uses cuda:0")
    ap.add_argument("--windows", type=int, default=3,
                    help="timed windows of --steps steps each (every one bracketed by barrier + synchronize); ms_per_step / value are "
                         "the MEDIAN window's, all of them are printed as ms_per_step_windows")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise the process group (and use ExchangeAdam's exchange) even with one rank: the RCCL world-1 smoke test")
    ap.add_argument("--no-aux", action="store_true", help="skip the auxiliary legs (configs[4] 256^3 grid, configs[2] PropNet step)")
    ap.add_argument("--aux-steps", type=int, default=40, help="timed steps of each auxiliary leg")
    ap.add_argument("--no-rank-step", action="store_true", help="skip aux.configs3_rank_step (configs[3]'s per-rank step over RCCL world-of-one)")
    ap.add_argument("--rank-step-pretrain", type=int, default=150, help="training steps from fog of the multi-tensor field of aux.configs3_rank_step")
    ap.add_argument("--no-scene-sweep", action="store_true", help="skip aux.configs4_scene_sweep (eight procedural scenes at 256^3)")
    ap.add_argument("--scene-pretrain", type=int, default=1000, help="training steps from fog of every scene of aux.configs4_scene_sweep")
    ap.add_argument("--dump-sampling-state", default="",
                    help="write the occupancy grid and one ray batch of the timed steady state to this .npz "
                         "(tools/traverse_replay.py replays the sampling call on it under rocprofv3)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` outside a launcher: start the N ranks ourselves, exactly as the driver would
        # (torch.distributed.run, one rank per GPU); rank 0's line is the only thing on stdout
        import socket
        import subprocess
        with socket.socket() as sock:
            sock.bind(("127.0.0.1", 0))
            port = sock.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd))
    return args


def run(args):
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world_size = int(os.environ.get("WORLD_SIZE", "1"))
    assert world_size == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world_size}"
    if args.all_ranks_on_device0:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    exchanging = world_size > 1 or args.force_dist
    if exchanging:
        if "MASTER_ADDR" not in os.environ:                  # --force-dist outside a launcher
            import socket
            with socket.socket() as sock:
                sock.bind(("127.0.0.1", 0))
                os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(sock.getsockname()[1]))
        # (collectives give up after five minutes, not after the backends' 10 / 30: a rank that fell out of step ends the job, it
        #  does not park it)
        import datetime
        pg_timeout = datetime.timedelta(seconds=300)
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=device, rank=rank, world_size=world_size, timeout=pg_timeout)
        else:
            dist.init_process_group(args.dist_backend, rank=rank, world_size=world_size, timeout=pg_timeout)
    fixed_rays = args.rays_per_iter // world_size if args.rays_per_iter > 0 else 0

    torch.manual_seed(42)
    teacher = DenseGridField(AABB, GRID_RES).to(device).eval()          # the analytic scene sampled on a grid
    if args.field == "grid+mlp":
        field = GridMlpField(AABB).to(device)
    else:
        field = DenseGridField(AABB, GRID_RES).to(device)
        with torch.no_grad():                                           # the student starts from fog and grey
            field.grid[:, :1].fill_(math.log(0.5))
            field.grid[:, 1:].zero_()
    est_t = nerfacc.OccGridEstimator(roi_aabb=AABB, resolution=args.occ_res, levels=1).to(device)
    est = nerfacc.OccGridEstimator(roi_aabb=AABB, resolution=args.occ_res, levels=1).to(device)
    if exchanging:
        # overlap_backward: this loop obeys its rule (sharding.ExchangeAdam) — the count exchange of a step is STARTED before
        # backward() and only read after step(); nothing else talks to the process group in between
        optimizer = sharding.ExchangeAdam(field.parameters(), lr=1e-2, eps=1e-15, weight_decay=1e-6, n_chunks=args.grad_chunks,
                                          overlap_backward=True, mode=args.exchange_mode)
    else:
        optimizer = torch.optim.Adam(field.parameters(), lr=1e-2, eps=1e-15, weight_decay=1e-6, fused=True)
    loss_scale = 2.0**10
    bkgd = torch.ones(3, device=device)

    def occ_eval_fn(x):
        return field.query_density(x) * RENDER_STEP

    est_t.train()
    with sharding.synchronized_rng(1234, device):
        for _ in range(4):
            est_t._update(step=0, occ_eval_fn=lambda x: teacher.query_density(x) * RENDER_STEP, occ_thre=1e-2)
    pool_o, pool_d = make_ray_pool(args.pool, seed=42 + rank, device=device)
    est_t.eval()
    with torch.no_grad():                        # target pixels from the analytic scene
        pool_rgb = torch.empty((args.pool, 3), device=device)
        for i in range(0, args.pool, 1 << 16):
            rgb, _, _, _ = render_rays(teacher, est_t, pool_o[i:i + (1 << 16)], pool_d[i:i + (1 << 16)], bkgd, False)
            pool_rgb[i:i + (1 << 16)] = rgb
    del est_t
    est.train()
    torch.manual_seed(1000 + rank)

    state = {"num_rays": fixed_rays or INIT_RAYS, "step": 0, "est": est}     # state["est"]: the estimator the step functions use
    stats = {"rays": 0, "samples": 0, "candidates": 0}
    side = torch.cuda.Stream(device=device)

    def next_num_rays(g_samples, g_rays):
        if fixed_rays or g_samples <= 0:
            return
        # train_ngp_nerf_occ.py:187-194, on the global counts so that all ranks stay in step
        state["num_rays"] = min(max(int((g_rays / world_size) * (TARGET_SAMPLES / (g_samples / world_size))), 64), args.pool)

    def backward_and_update(rgb, pixels, n_samples):
        optimizer.zero_grad()
        if n_samples > 0:
            loss = F.smooth_l1_loss(rgb, pixels)
            (loss * loss_scale).backward()
        # every rank takes part in the exchange every step, samples or not (ExchangeAdam.step all-reduces)
        if n_samples > 0 or exchanging:
            optimizer.step()

    def refresh_grid(step):
        with sharding.synchronized_rng(5000 + step, device):
            state["est"].update_every_n_steps(step=step, occ_eval_fn=occ_eval_fn, occ_thre=1e-2)

    # ---- the reference examples' step, through the public API only ------------------------------------------------
    def step_api():
        step = state["step"]
        if step % 16 == 0:
            refresh_grid(step)
        n = state["num_rays"]
        idx = torch.randint(0, args.pool, (n,), device=device)
        rays_o, rays_d, pixels = pool_o[idx], pool_d[idx], pool_rgb[idx]
        rgb, _, _, n_samples = render_rays_reference_style(field, state["est"], rays_o, rays_d, bkgd, True)
        pending = sharding.allreduce_counts_begin(n_samples, n, device)
        backward_and_update(rgb, pixels, n_samples)
        next_num_rays(*sharding.allreduce_counts_end(pending))
        stats["rays"] += n
        stats["samples"] += n_samples
        stats["candidates"] += LAST_CALL["candidates"]
        state["step"] += 1

    # ---- same work, the next step's traversal overlapped with this step's backward pass ------------------------------
    def propose(n, wait_for_main):
        main = torch.cuda.current_stream(device)
        if wait_for_main:
            side.wait_stream(main)
        with torch.cuda.stream(side):
            idx = torch.randint(0, args.pool, (n,), device=device)
            rays_o, rays_d, pixels = pool_o[idx], pool_d[idx], pool_rgb[idx]
            near = torch.rand(n, device=device) * RENDER_STEP          # near_plane 0 + stratified jitter (occ_grid.py:162-163)
            far = torch.full((n,), 1e10, device=device)
            e = state["est"]
            ri, ts, te, _ = nerfacc.cuda.sample_occgrid(rays_o, rays_d, e.binaries, e.aabbs, near, far, RENDER_STEP, 0.0)
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

        rgb, _, _, _ = nerfacc.rendering(ts, te, ri, n_rays=prop["n"], rgb_sigma_fn=rgb_sigma_fn, render_bkgd=bkgd)
        return rgb, ts.shape[0]

    def step_overlap():
        step = state["step"]
        if step % 16 == 0:
            state.pop("proposal", None)
            refresh_grid(step)
        prop = state.pop("proposal", None)
        if prop is None:
            prop = propose(state["num_rays"], wait_for_main=True)
        n = prop["n"]
        rgb, n_samples = render_proposed(prop)
        pending = sharding.allreduce_counts_begin(n_samples, n, device)
        backward_and_update(rgb, prop["pixels"], n_samples)
        with torch.cuda.stream(side):                 # the 16-byte result is read without waiting for the backward pass
            g = sharding.allreduce_counts_end(pending)
        next_num_rays(*g)
        if (step + 1) % 16 != 0:
            state["proposal"] = propose(state["num_rays"], wait_for_main=False)
        stats["rays"] += n
        stats["samples"] += n_samples
        stats["candidates"] += prop["ts"].shape[0]
        state["step"] += 1

    # ---- the path alone: the same ray draws, sampling and rendering calls (forward + backward), the field replaced by slices of
    # two constant tensors — what the path sustains per step when the user's field costs nothing (an auxiliary figure, not the metric)
    free_sig = torch.rand(1 << 21, device=device) * 20.0
    free_rgb = torch.rand(1 << 21, 3, device=device)

    def step_path_only():
        n = state["num_rays"]
        idx = torch.randint(0, args.pool, (n,), device=device)
        rays_o, rays_d = pool_o[idx], pool_d[idx]
        ri, ts, te = state["est"].sampling(rays_o, rays_d, sigma_fn=lambda a, b, r: free_sig[:a.shape[0]], near_plane=0.0, far_plane=1e10,
                                  render_step_size=RENDER_STEP, stratified=True, cone_angle=0.0, alpha_thre=0.0)
        k = ts.shape[0]
        leaves = (free_rgb[:k].detach().requires_grad_(True), free_sig[:k].detach().requires_grad_(True))
        rgb, _, _, _ = nerfacc.rendering(ts, te, ri, n_rays=n, rgb_sigma_fn=lambda a, b, r: leaves, render_bkgd=bkgd)
        if k > 0:
            rgb.sum().backward()
        stats["rays"] += n
        stats["samples"] += k

    steps = {"api": step_api, "overlap": step_overlap}

    def timed_region(step_fn, n_steps, with_timer):
        """barrier + synchronize, exactly n_steps steps, synchronize + barrier; max over ranks"""
        timer = None
        if with_timer:
            timer = _backend.KernelTimer(names=("traverse_count", "traverse_fill", "traverse_sample"))
            _backend.set_kernel_timer(timer)
        stats.update(rays=0, samples=0, candidates=0)
        if world_size > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n_steps):
            step_fn()
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
        return dict(elapsed=elapsed, rays=tot[1].item(), samples=tot[2].item(), timer=timer,
                    local_rays=stats["rays"], local_samples=stats["samples"], local_candidates=stats["candidates"])

    # ---- untimed: train from fog into the steady state, then warm up the loop that is timed ---------------------------
    t_pre = time.perf_counter()
    for _ in range(args.pretrain):
        step_api()
    torch.cuda.synchronize()
    pretrain_s = time.perf_counter() - t_pre
    state["step"] = max(state["step"], 1024)             # past the grid's warm-up phase (occ_grid.py:372-376) whatever --pretrain was
    state["step"] += (-state["step"]) % 16 + 1           # the timed region starts one step after a grid refresh
    for _ in range(args.warmup):
        steps[args.mode]()
    if hasattr(optimizer, "timing"):
        optimizer.timing = True
    windows = [timed_region(steps[args.mode], args.steps, with_timer=True) for _ in range(max(1, args.windows))]
    comm = optimizer.comm_stats() if hasattr(optimizer, "comm_stats") else None
    if hasattr(optimizer, "timing"):
        optimizer.timing = False
    main_run = sorted(windows, key=lambda w: w["elapsed"])[(len(windows) - 1) // 2]      # the median window
    state.pop("proposal", None)

    # replicas stay bit-identical (north_star: every rank holds the field and the grid, its own rays, ONE gradient exchange per step):
    # a fingerprint of every parameter and of the estimator's state per rank, gathered; rank 0 says whether they agree
    replicas = None
    if exchanging and dist.is_initialized():
        def bits(t):
            t = t.detach().contiguous().view(-1)
            return (t.view(torch.int32) if t.dtype == torch.float32 else t.to(torch.int32)).to(torch.int64)
        fp = torch.zeros(2, dtype=torch.int64, device=device)
        for t in list(field.parameters()) + [est.occs, est.binaries]:
            b = bits(t)
            fp[0] += b.sum()
            fp[1] += (b * (torch.arange(b.shape[0], device=device, dtype=torch.int64) % 8191 + 1)).sum()      # (position-weighted: wrap-around int64 sums)
        gathered = [torch.zeros_like(fp) for _ in range(world_size)]
        dist.all_gather(gathered, fp)
        prints = [[int(x) for x in g.tolist()] for g in gathered]
        replicas = {"identical": all(p_ == prints[0] for p_ in prints), "fingerprints": prints,
                    "of": "every field parameter + the estimator's occs and binaries after the timed steps (int64 sums of the bit patterns, plain and position-weighted)"}

    # the OTHER exchange mode for a few steps (every rank takes part): what the design choice costs on this node
    exchange_modes = None
    if exchanging and isinstance(optimizer, sharding.ExchangeAdam) and not args.no_aux:
        exchange_modes = {args.exchange_mode: {"ms_per_step": main_run["elapsed"] / args.steps * 1e3, "comm_ms_per_step": comm["wait_ms"],
                                               "comm_window_ms_per_step": comm["window_ms"], "steps": args.steps}}
        alt = "rs_ag" if args.exchange_mode == "allreduce" else "allreduce"
        try:                # (an auxiliary leg must never cost the headline its line: every rank takes the same path, errors are reported)
            optimizer.set_mode(alt)
            for _ in range(min(args.warmup, 5)):
                steps[args.mode]()
            optimizer.timing = True
            r = timed_region(steps[args.mode], args.aux_steps, with_timer=False)
            c = optimizer.comm_stats()
            optimizer.timing = False
            exchange_modes[alt] = {"ms_per_step": r["elapsed"] / args.aux_steps * 1e3, "comm_ms_per_step": c["wait_ms"],
                                   "comm_window_ms_per_step": c["window_ms"], "steps": args.aux_steps}
        except Exception as e:      # noqa: BLE001
            if world_size > 1:
                # ONE rank's error (the other ranks keep stepping this leg's collectives): swallowing it here leaves the job
                # waiting for this rank until the backend's timeout — half an hour with gloo.  The rank dies, the launcher ends the job.
                raise
            exchange_modes[alt] = {"error": f"{type(e).__name__}: {e}"[:300]}
        finally:
            optimizer.timing = False
            try:
                optimizer.set_mode(args.exchange_mode)       # (the headline's mode again, also after an error: ADVICE r4)
            except Exception:      # noqa: BLE001
                pass
        state.pop("proposal", None)

    other = None
    if not args.no_other_mode:
        other_mode = "overlap" if args.mode == "api" else "api"
        for _ in range(min(args.warmup, 10)):
            steps[other_mode]()
        other = (other_mode, timed_region(steps[other_mode], args.steps, with_timer=False))
        state.pop("proposal", None)

    path_only = None
    if not args.no_other_mode and world_size == 1 and state["num_rays"] * 200 < free_sig.shape[0] * 4:
        for _ in range(min(args.warmup, 10)):
            step_path_only()
        path_only = timed_region(step_path_only, args.steps, with_timer=False)

    prof = None
    if not args.no_profile and world_size == 1:
        for _ in range(3):
            steps[args.mode]()
        prof = profile_steps(steps[args.mode], min(args.steps, 32))
        state.pop("proposal", None)

    path_prof = None
    if path_only is not None and not args.no_profile:
        path_prof = profile_steps(step_path_only, min(args.steps, 32))

    aux = {}
    if not args.no_aux and world_size == 1 and args.field == "grid":
        if args.occ_res != 256:
            # configs[4]'s grid size: the SAME step with a 256^3 occupancy grid built from the trained field
            est256 = nerfacc.OccGridEstimator(roi_aabb=AABB, resolution=256, levels=1).to(device)
            est256.train()
            with sharding.synchronized_rng(777, device):
                for _ in range(4):
                    est256._update(step=0, occ_eval_fn=occ_eval_fn, occ_thre=1e-2)
            saved = (state["est"], state["num_rays"], state["step"])
            state["est"] = est256
            state["step"] += (-state["step"]) % 16 + 1
            for _ in range(min(args.warmup, 10)):
                step_api()
            r = timed_region(step_api, args.aux_steps, with_timer=False)
            aux["configs4_occgrid_256"] = {
                "workload": "the configs[1] step of this line with a 256^3 occupancy grid (configs[4]'s grid size) on the same scene and field",
                "steps": args.aux_steps, "ms_per_step": r["elapsed"] / args.aux_steps * 1e3, "rays_per_sec": r["rays"] / r["elapsed"],
                "samples_per_sec": r["samples"] / r["elapsed"], "rays_per_iter": r["local_rays"] / args.aux_steps,
                "occupied_fraction": est256.binaries.float().mean().item()}
            state["est"], state["num_rays"], state["step"] = saved
            del est256
        aux["configs2_propnet_step"] = propnet_step_leg(field, pool_o, pool_d, pool_rgb, bkgd, args.aux_steps, min(args.warmup, 10))
        if not args.no_scene_sweep:
            aux["configs4_scene_sweep"] = scene_sweep_leg(pool_o, pool_d, bkgd, args.aux_steps, min(args.warmup, 10), pretrain=args.scene_pretrain)

    # ---- aux.configs3_rank_step: what ONE rank of configs[3] does per step — 8192 fixed rays (65536 / 8) and the
    # gradient exchange of ExchangeAdam over RCCL with a world of one (the collectives are real RCCL calls; with one rank they move
    # nothing, so the times are the exchange's fixed cost and, more to the point, the WINDOW the step leaves for hiding it), in both
    # exchange modes, with the headline's single-tensor field and with a multi-tensor one (feature grid + MLP: chunks leave from
    # inside backward).  The 8-rank figures are a link-model projection, labelled as such — no node has been available.
    rank_step = None
    if not args.no_aux and world_size == 1 and not exchanging and args.field == "grid" and not args.no_rank_step:
        rank_step = {"workload": "configs[3] per-rank share: the configs[1] step at 8192 fixed rays with ExchangeAdam over RCCL (world of one), "
                                 "chunks launched from inside backward (overlap_backward)", "fields": {}}
        saved = (field, optimizer, exchanging, fixed_rays, state["num_rays"], state["est"], state["step"])
        try:
            import copy
            import socket
            with socket.socket() as sock:
                sock.bind(("127.0.0.1", 0))
                os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(sock.getsockname()[1]))
            dist.init_process_group("nccl", device_id=device, rank=0, world_size=1)
            exchanging, fixed_rays = True, 8192
            for kind in ("grid", "grid+mlp"):
                field = copy.deepcopy(saved[0]) if kind == "grid" else GridMlpField(AABB).to(device)
                est_k = saved[5]
                if kind != "grid":                               # its own occupancy grid, trained from fog for a moment
                    est_k = nerfacc.OccGridEstimator(roi_aabb=AABB, resolution=args.occ_res, levels=1).to(device)
                    est_k.train()
                state["est"], state["num_rays"], state["step"] = est_k, 8192, 0 if kind != "grid" else saved[6]
                res_k = {"parameters": int(sum(p.numel() for p in field.parameters())), "tensors": len(list(field.parameters()))}
                optimizer = sharding.ExchangeAdam(field.parameters(), lr=1e-2, eps=1e-15, weight_decay=1e-6, n_chunks=args.grad_chunks,
                                                  overlap_backward=True, mode="allreduce")
                for mode in ("allreduce", "rs_ag"):
                    optimizer.set_mode(mode)
                    for _ in range(args.rank_step_pretrain if (kind != "grid" and mode == "allreduce") else 5):
                        step_api()
                    state["step"] += (-state["step"]) % 16 + 1
                    optimizer.timing = True
                    r = timed_region(step_api, args.aux_steps, with_timer=False)
                    c = optimizer.comm_stats()
                    optimizer.timing = False
                    t_step = r["elapsed"] / args.aux_steps * 1e3
                    # link model (the brief's figures: 7 xGMI links x ~153 GB/s per GPU, ring collectives per-link bound): a ring
                    # all-reduce moves 2 (N - 1) / N of the buffer per rank; one ring = one link, RCCL may run up to seven in parallel
                    B, N = float(c["exchange_bytes"]), 8
                    t_one = 2.0 * (N - 1) / N * B / 153e9 * 1e3
                    t_seven = t_one / (7 * 0.7)
                    budget = max(c["window_ms"] - c["wait_ms"], 0.0)             # compute between the first chunk's launch and the last one's use
                    exposed = [max(t_seven - budget, 0.0), max(t_one - budget, 0.0)]
                    res_k[mode] = {"ms_per_step": t_step, "rays_per_sec": r["rays"] / r["elapsed"], "samples_per_ray": r["samples"] / max(r["rays"], 1),
                                   "comm_ms_per_step": c["wait_ms"], "comm_window_ms_per_step": c["window_ms"], "exchange_bytes": c["exchange_bytes"],
                                   "projected_8_ranks": {"ring_allreduce_ms": {"seven_rings_70pct": t_seven, "one_ring": t_one},
                                                         "overlap_budget_ms": budget, "exposed_ms": exposed,
                                                         "weak_scaling_efficiency": [t_step / (t_step + e) for e in exposed],
                                                         "note": "model, not a measurement: exposed = ring time - the compute window this step leaves between "
                                                                 "launching its first chunk and needing its last"}}
                optimizer.close()
                rank_step["fields"][kind] = res_k
        except Exception as e:      # noqa: BLE001  (an auxiliary leg must never cost the headline its line)
            rank_step["error"] = f"{type(e).__name__}: {e}"[:300]
        finally:
            field, optimizer, exchanging, fixed_rays = saved[:4]
            state["num_rays"], state["est"], state["step"] = saved[4:]
            if dist.is_initialized():
                dist.destroy_process_group()

    line = None
    if rank == 0:
        elapsed = main_run["elapsed"]
        ms_per_step = elapsed / args.steps * 1e3
        # roofline of the dominant kernel of OUR path (profiles/): the sampling traversal — count pass
        # (traverse_count_split_kernel at this ray count) + emit pass (traverse_emit_tiles_kernel), one launch each per
        # step, timed live with HIP events on the launch stream around their single-kernel C-ABI calls.
        # Algorithmic bytes: 16 B per emitted candidate sample (ray_indices i64 + t_starts + t_ends)
        # + 48 B per ray + the grid once, G*V bool bytes as the API hands it over.
        summ = main_run["timer"].summary()
        n_count, ms_count = summ.get("traverse_count", (0, 0.0))
        n_emit, ms_emit = summ.get("traverse_fill", (0, 0.0))
        n_fused, ms_fused = summ.get("traverse_sample", (0, 0.0))
        # round 6: the call is ONE launch at this size (count + look-back + emit in traverse_count_split_kernel<..., fused>,
        # nfa_traverse_sample); steps whose guess of the output size was too small add an emit launch.  Per timed step:
        steps_t = max(args.steps, 1)
        fused = n_fused > 0
        n_launch = n_fused if fused else n_count
        ms_dominant = ms_fused if fused else ms_count                         # average duration of the dominant kernel's launches
        ms = (n_fused * ms_fused + n_count * ms_count + n_emit * ms_emit) / steps_t      # traversal kernel time per step
        rays_per_launch = main_run["local_rays"] / steps_t
        # candidate samples per launch: what the timed launches produced (the totals every sampling call reads back), not a fresh draw
        cand = main_run["local_candidates"] / steps_t
        n_rendered = main_run["local_samples"] / steps_t
        assert cand >= n_rendered, f"candidates per step ({cand}) < rendered samples per step ({n_rendered}): the counters disagree"
        with torch.no_grad():
            idx = torch.randint(0, args.pool, (int(rays_per_launch),), device=device)
            walk = dda_steps(pool_o[idx], pool_d[idx], est.aabbs[0], args.occ_res, 0.0, 1e10)
        if args.dump_sampling_state:
            with torch.no_grad():
                cand_dump = est.sampling(pool_o[idx], pool_d[idx], render_step_size=RENDER_STEP, stratified=True)[0].shape[0]
            np.savez_compressed(args.dump_sampling_state, binaries_bits=np.packbits(est.binaries.cpu().numpy().ravel()),
                                res=np.array(est.binaries.shape), aabbs=est.aabbs.cpu().numpy(), rays_o=pool_o[idx].cpu().numpy(),
                                rays_d=pool_d[idx].cpu().numpy(), jitter=torch.rand(idx.shape[0], device=device).cpu().numpy(),
                                render_step=np.float32(RENDER_STEP), candidates=np.int64(cand_dump))
        alg_bytes = 16.0 * cand + 48.0 * rays_per_launch + float(args.occ_res**3)
        achieved = alg_bytes / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
        roof = {
            "kernel": ("traverse_count_split_kernel<fused> (the sampling traversal: count, offsets by look-back and emit in one launch)" if fused
                       else "traverse_count_split_kernel + traverse_emit_tiles_kernel (the sampling traversal)"),
            "bound": "issue/latency", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
            "traffic": None, "traffic_source": None,
            "avg_launch_ms": ms_dominant, "emit_avg_launch_ms": ms_emit, "emit_launches": n_emit, "launches": n_launch,
            "traversal_kernel_ms_per_step": ms,
            "algorithmic_bytes_per_launch": alg_bytes,
            "candidate_samples_per_sec_of_kernel_time": cand / (ms * 1e-3) if ms > 0 else 0.0,
            "dda_steps_per_launch": walk, "dda_steps_per_sec_of_kernel_time": walk / (ms * 1e-3) if ms > 0 else 0.0,
            "note": "a dependent voxel walk over an LDS-resident bit-packed grid: ~16 B of HBM traffic per sample, so the HBM fraction is small "
                    "by construction; the bound is instruction issue / latency (roofline.issue), the figures of merit are candidate samples/s "
                    "and DDA steps/s of kernel time (round 6: of the whole single launch, emit included).  The HBM-streaming kernels of the "
                    "path are measured at N >= 2^24 in profiles/.",
        }
        # counter-based figures come from a committed rocprofv3 --pmc run of THIS workload (tools/pmc_bench.py); they are
        # attached only when that run's ray count is within 15 % of this run's
        import glob
        pmc_files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_pmc_traverse.json")))      # the latest round's
        pmc_path = pmc_files[-1] if pmc_files else ""
        if pmc_path:
            try:
                pmc = json.load(open(pmc_path))
                if abs(pmc["rays_per_launch"] - rays_per_launch) <= 0.15 * rays_per_launch:
                    roof["traffic"] = pmc["hbm_bytes_per_launch"]
                    roof["traffic_source"] = "profiles/%s (rocprofv3 --pmc, separate passes; %d rays/launch)" % (os.path.basename(pmc_path), pmc["rays_per_launch"])
                    if "issue" in pmc:
                        roof["issue"] = pmc["issue"]
            except Exception:      # noqa: BLE001
                pass
        out = {
            "metric": "training rays/sec (+ samples/sec), NGP+OccGrid Lego 800x800",
            "value": main_run["rays"] / elapsed,
            "unit": "rays/s",
            "samples_per_sec": main_run["samples"] / elapsed,
            "n_gpus": world_size, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "ms_per_step_windows": [w["elapsed"] / args.steps * 1e3 for w in windows],
            "higher_is_better": True, "scaling": "strong" if fixed_rays else "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {
                "workload": f"configs[1] lego stand-in: procedural lego-like scene, 100 cams 800x800, {args.occ_res}^3 OccGrid on aabb +-1.5, "
                            "render_step 5e-3, " + (f"{args.rays_per_iter} rays/iter global (configs[3])" if fixed_rays else "~2^18 rendered samples/iter/GPU")
                            + ", torch dense-grid field (tiny-cuda-nn absent)",
                "loop": args.mode + (" (reference examples' step through the public API)" if args.mode == "api" else " (next step's traversal on a side stream)"),
                "pretrain_steps": args.pretrain, "pretrain_seconds": round(pretrain_s, 2),
                "rays_per_iter_per_gpu": rays_per_launch,
                "samples_per_iter_per_gpu": main_run["local_samples"] / max(args.steps, 1),
                "samples_per_ray": main_run["samples"] / max(main_run["rays"], 1),
                "candidate_samples_per_iter": cand,
                "occupied_fraction": est.binaries.float().mean().item(),
                "parallelism": f"rays sharded over {world_size} GPU(s)" + (f", gradient exchange ({args.exchange_mode}) in {args.grad_chunks} async chunks overlapped with Adam" if world_size > 1 else ""),
                "field": args.field,
            },
            "roofline": roof,
        }
        if other is not None:
            name, r = other
            out["other_loop"] = {"loop": name, "ms_per_step": r["elapsed"] / args.steps * 1e3, "rays_per_sec": r["rays"] / r["elapsed"],
                                 "samples_per_sec": r["samples"] / r["elapsed"]}
        if path_only is not None:
            out["path_only_loop"] = {"ms_per_step": path_only["elapsed"] / args.steps * 1e3, "rays_per_sec": path_only["rays"] / path_only["elapsed"],
                                     "samples_per_sec": path_only["samples"] / path_only["elapsed"],
                                     "note": "estimator.sampling (visibility filter included) + nerfacc.rendering forward and backward on the same ray "
                                             "draws with the field replaced by slices of constant tensors: the path without the stand-in field; not the metric"}
        if path_prof is not None and "error" not in path_prof and "path_only_loop" in out:
            po = out["path_only_loop"]
            po["path_us_per_step"] = path_prof["nfa_us_per_step"]
            po["gpu_idle_frac"] = max(0.0, 1.0 - path_prof["busy_us_per_step"] / (po["ms_per_step"] * 1e3))
        if replicas is not None:
            out["replicas"] = replicas
        if exchange_modes is not None:
            aux["exchange_modes"] = exchange_modes
        if aux:
            out["aux"] = aux
        if comm is not None and comm["steps"] > 0:
            out["comm_ms_per_step"] = comm["wait_ms"]
            out["comm_window_ms_per_step"] = comm["window_ms"]
            out["exchange_bytes"] = comm["exchange_bytes"]
            out["comm_note"] = ("comm_ms_per_step = time the compute stream stood still waiting for gradient chunks (exposed exchange), "
                                "comm_window = first all-reduce launch (inside backward) to last chunk's arrival; rank 0's events")
        if rank_step is not None:
            out.setdefault("aux", {})["configs3_rank_step"] = rank_step
        if prof is not None and "error" not in prof:
            # the PATH's own fraction at this size: algorithmic bytes of every nfa:: kernel of a step
            # (sampling 16 c + 48 R + the bool grid; filter 20 c + 16 N; rendering forward 44 N + 20 R, backward 56 N + 20 R —
            # the backward kernel no longer reads the weights; c = candidate samples, N = rendered samples, R = rays, all of them the
            # timed steps' own averages) over their summed kernel time from the profiled pass
            n_s = main_run["local_samples"] / max(args.steps, 1)
            path_bytes = (16.0 * cand + 48.0 * rays_per_launch + float(args.occ_res**3)) + (20.0 * cand + 16.0 * n_s) \
                + (44.0 * n_s + 20.0 * rays_per_launch) + (56.0 * n_s + 20.0 * rays_per_launch)
            path_gbs = path_bytes / (prof["nfa_us_per_step"] * 1e-6) / 1e9 if prof["nfa_us_per_step"] > 0 else 0.0
            roof["path"] = {"kernels": "every nfa:: kernel of a step (sampling, visibility filter, rendering forward + backward)",
                            "us_per_step": prof["nfa_us_per_step"], "launches_per_step": prof["nfa_kernels_per_step"],
                            "algorithmic_bytes_per_step": path_bytes, "achieved": path_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                            "frac": path_gbs / HBM_PEAK_GBS,
                            "path_only_loop_ms_per_step": (path_only["elapsed"] / args.steps * 1e3) if path_only is not None else None,
                            "note": "latency-bound at this size: 5-6 launches of 9-40 us each; the same kernels reach 0.35-0.7 of 8 TB/s at N = 2^24 (profiles/)"}
        if prof is not None:
            if "error" in prof:
                out["gpu_activity"] = prof
            else:
                out["path_us_per_step"] = prof["nfa_us_per_step"]
                out["gpu_idle_frac"] = max(0.0, 1.0 - prof["busy_us_per_step"] / (ms_per_step * 1e3))
                out["gpu_activity"] = prof
        if not args.no_cpu_baseline and world_size == 1 and args.field == "grid":
            out["cpu_baseline"] = cpu_baseline(field, est, pool_o, pool_d)
        line = json.dumps(out)
    if exchanging:
        dist.destroy_process_group()
    return line


if __name__ == "__main__":
    main()
