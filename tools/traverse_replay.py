"""Replay OccGridEstimator.sampling's traversal (count -> offsets -> emit) on a state dumped by
`bench.py --dump-sampling-state`: the steady-state occupancy grid and one ray batch of the timed region.
    python tools/traverse_replay.py profiles/r02_sampling_state.npz [reps] [--check]
Used under rocprofv3 (--kernel-trace / --pmc) and for kernel tuning; --check compares with the CPU oracle."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nerfacc_amd as nerfacc
from nerfacc_amd.cuda import _backend

st = np.load(sys.argv[1])
reps = int(sys.argv[2]) if len(sys.argv) > 2 and not sys.argv[2].startswith("-") else 20
dev = torch.device("cuda:0")
res = tuple(int(x) for x in st["res"])
binaries = torch.from_numpy(np.unpackbits(st["binaries_bits"])[: int(np.prod(res))].astype(bool).reshape(res)).to(dev)
aabbs = torch.from_numpy(st["aabbs"]).to(dev)
O, D = torch.from_numpy(st["rays_o"]).to(dev), torch.from_numpy(st["rays_d"]).to(dev)
jit = torch.from_numpy(st["jitter"]).to(dev)
for a in sys.argv:
    if a.startswith("--rays="):          # tile / truncate the ray batch to this many rays (lanes-per-ray sweeps)
        n_ = int(a.split("=")[1])
        rep = -(-n_ // O.shape[0])
        O, D, jit = O.repeat(rep, 1)[:n_].contiguous(), D.repeat(rep, 1)[:n_].contiguous(), jit.repeat(rep)[:n_].contiguous()
step = float(st["render_step"])
C = _backend._C
def call():
    return C.sample_occgrid(O, D, binaries, aabbs, None, None, step, 0.0, near_plane=0.0, far_plane=1e10, jitter=jit, jitter_scale=step)
ri, ts, te, pk = call()
torch.cuda.synchronize()
timer = _backend.KernelTimer(names=("traverse_count", "traverse_fill", "traverse_sample"))
_backend.set_kernel_timer(timer)
gap_us = next((float(a.split("=")[1]) for a in sys.argv if a.startswith("--gap-us=")), 0.0)     # idle GPU time between calls (what a host-bound training loop leaves)
t0 = time.perf_counter()
for _ in range(reps):
    call()
    if gap_us > 0:
        torch.cuda.synchronize()
        t_end = time.perf_counter() + gap_us * 1e-6
        while time.perf_counter() < t_end:
            pass
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / reps
summ = timer.summary()
_backend.set_kernel_timer(None)
us = lambda k: summ[k][1] * 1e3 if k in summ and summ[k][0] else float("nan")
print(f"rays {O.shape[0]} candidates {ri.shape[0]} occupied {binaries.float().mean().item():.4f}  "
      f"count {us('traverse_count'):.1f} us  emit {us('traverse_fill'):.1f} us  fused count+offsets+emit {us('traverse_sample'):.1f} us  call wall {wall*1e6:.1f} us  "
      f"sums {int(ri.sum())} {float(ts.double().sum()):.9e} {float(te.double().sum()):.9e}")
if "--check" in sys.argv:
    import oracle
    near = (st["jitter"] * np.float32(step)).astype(np.float32)
    r_ri, r_ts, r_te, _ = oracle.sample_occgrid(st["rays_o"], st["rays_d"], binaries.cpu().numpy(), st["aabbs"], near,
                                                np.full(near.shape, 1e10, np.float32), step)
    ok = np.array_equal(ri.cpu().numpy(), r_ri) and np.array_equal(ts.cpu().numpy(), r_ts) and np.array_equal(te.cpu().numpy(), r_te)
    print("oracle check:", "bit-exact" if ok else "MISMATCH")
    sys.exit(0 if ok else 1)
