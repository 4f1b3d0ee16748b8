"""Where do the cycles of traverse_count_split_kernel go?  Builds an instrumented copy of the
library (-DNFA_PHASE_CYCLES: shader-clock stamps between the phases of the kernel, summed over
waves), runs the count pass on bench.py's ray batch and prints average cycles per wave and phase.

    python tools/phase_cycles.py [n_rays] [iters]        (NFA_SPLIT_P=<P> selects lanes per ray)
    python tools/phase_cycles.py --state=profiles/r02_sampling_state.npz [iters]     (bench.py's steady state)
"""
import ctypes, glob, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
out_dir = os.path.join(ROOT, "tools", "_prof")
os.makedirs(out_dir, exist_ok=True)
extra = os.environ.get("NFA_PHASE_EXTRA", "").split()          # further -D flags (A/B builds), e.g. NFA_PHASE_EXTRA="-DNFA_EVCAP=8"
so = os.path.join(out_dir, "libnerfacc_hip_prof" + "".join(c if c.isalnum() else "_" for c in "".join(extra)) + ".so")
srcs = sorted(glob.glob(os.path.join(ROOT, "nerfacc_amd", "csrc", "*.hip")))
hdrs = glob.glob(os.path.join(ROOT, "nerfacc_amd", "csrc", "*.hpp")) + [os.path.join(ROOT, "include", "nerfacc_hip.h")]
if os.environ.get("NFA_PHASE_LIB"):                 # a prebuilt instrumented library (A/B against another source tree)
    so = os.path.abspath(os.environ["NFA_PHASE_LIB"])
elif not os.path.exists(so) or any(os.path.getmtime(f) > os.path.getmtime(so) for f in srcs + hdrs):
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
                           "-fvisibility=hidden", "-DNFA_PHASE_CYCLES", *extra, "-shared", *srcs, "-o", so])
if "--build-only" in sys.argv:
    sys.exit(0)
os.environ["NERFACC_AMD_BACKEND"] = "ctypes"     # the instrumented copy is loaded through the ctypes face
os.environ["NERFACC_AMD_LIB"] = so
import torch
from nerfacc_amd.cuda import _backend
assert _backend.LIB_PATH == so and _backend.BACKEND == "ctypes"
import bench
import nerfacc_amd as nerfacc
from nerfacc_amd import cuda as C

args = [a for a in sys.argv[1:] if not a.startswith("--")]
state = [a.split("=", 1)[1] for a in sys.argv[1:] if a.startswith("--state=")]
n = int(args[0]) if len(args) > 0 and not state else 13120
iters = int(args[-1]) if args and (state or len(args) > 1) else 20
dev = torch.device("cuda:0")
torch.manual_seed(42)
if state:
    import numpy as np
    st = np.load(state[0])
    res = tuple(int(x) for x in st["res"])
    binaries = torch.from_numpy(np.unpackbits(st["binaries_bits"])[: int(np.prod(res))].astype(bool).reshape(res)).to(dev)
    aabbs = torch.from_numpy(st["aabbs"]).to(dev)
    pool_o, pool_d = torch.from_numpy(st["rays_o"]).to(dev), torch.from_numpy(st["rays_d"]).to(dev)
    step = float(st["render_step"])
    near = torch.from_numpy(st["jitter"]).to(dev) * step
    for a_ in sys.argv[1:]:
        if a_.startswith("--rays="):               # tile / truncate the ray batch (lanes-per-ray regimes beyond the bench's 6.5 k)
            n_ = int(a_.split("=")[1])
            rep = -(-n_ // pool_o.shape[0])
            pool_o, pool_d, near = pool_o.repeat(rep, 1)[:n_].contiguous(), pool_d.repeat(rep, 1)[:n_].contiguous(), near.repeat(rep)[:n_].contiguous()
    n = pool_o.shape[0]
elif os.environ.get("NFA_PHASE_WORKLOAD") == "m1-random":      # rand > 0.5 grid, 4096 rays
    import numpy as np
    g = np.random.default_rng(42)
    v = g.normal(size=(n, 3)); v /= np.linalg.norm(v, axis=1, keepdims=True)
    o = (0.5 + 1.5 * v).astype(np.float32)
    d = g.random((n, 3)).astype(np.float32) - o; d /= np.linalg.norm(d, axis=1, keepdims=True)
    pool_o, pool_d = torch.from_numpy(o).to(dev), torch.from_numpy(d.astype(np.float32)).to(dev)
    binaries = torch.from_numpy(g.random((1, 128, 128, 128)) > 0.5).to(dev)
    aabbs = torch.tensor([[0.0, 0, 0, 1, 1, 1]], device=dev)
    step = 5e-3 / 3
    near = torch.zeros(n, device=dev)
elif os.environ.get("NFA_PHASE_WORKLOAD") == "levels4":        # tools/multilevel_bench.py's scene: 4 levels, segment-per-lane count pass
    import numpy as np
    g = np.random.default_rng(0)
    c = (np.arange(128) + 0.5) / 128 * 2 - 1
    X, Y, Z = np.meshgrid(c, c, c, indexing="ij")
    occ = np.stack([((X * 2.0**l) ** 2 + (Y * 2.0**l) ** 2 + (Z * 2.0**l) ** 2 < 0.25) | (g.random((128, 128, 128)) < (float(os.environ.get("ML_CLUTTER", "0.002")) if l else 0.0))
                    for l in range(4)])
    v = g.normal(size=(n, 3)); v /= np.linalg.norm(v, axis=1, keepdims=True)
    d = g.normal(size=(n, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
    pool_o, pool_d = torch.from_numpy((0.6 * v).astype(np.float32)).to(dev), torch.from_numpy(d.astype(np.float32)).to(dev)
    binaries = torch.from_numpy(occ).to(dev)
    aabbs = torch.from_numpy(np.stack([np.array([-1, -1, -1, 1, 1, 1], np.float32) * 2.0**l for l in range(4)])).to(dev)
    step = 1e-3
    near = torch.full((n,), 0.2, device=dev)
    names = ["stage occupancy into LDS (+barrier)", "ray loads, 4 slab tests + event sort, segment_of", "lattice to the first segment's start",
             "A: voxel walk of the segment", "B: lattice positions (segment start + own boundaries)", "stitch over the segments",
             "run records", "per-ray outputs (+ serial fallback)", "block sums"]
else:
    field = bench.DenseGridField(bench.AABB, 128).to(dev)
    est = nerfacc.OccGridEstimator(roi_aabb=bench.AABB, resolution=128, levels=1).to(dev)
    est.train()
    for _ in range(4):
        est._update(step=0, occ_eval_fn=lambda x: field.query_density(x) * bench.RENDER_STEP, occ_thre=1e-2)
    pool_o, pool_d = bench.make_ray_pool(n, 42, dev)
    binaries, aabbs, step = est.binaries, est.aabbs, bench.RENDER_STEP
    near = torch.rand(n, device=dev) * bench.RENDER_STEP
far = torch.full((n,), 1e10, device=dev)
lib = _backend.load_library()
lib.nfa_debug_phase_cycles.restype = ctypes.c_int
buf = (ctypes.c_ulonglong * 16)()
def run():
    return C.sample_occgrid(pool_o, pool_d, binaries, aabbs, near, far, step, 0.0)
for _ in range(3):
    run()
torch.cuda.synchronize()
lib.nfa_debug_phase_cycles(buf, 1)
for _ in range(iters):
    run()
torch.cuda.synchronize()
lib.nfa_debug_phase_cycles(buf, 0)
waves = buf[15]
names = globals().get("names") or ["stage occupancy into LDS (+barrier)", "ray loads, slab test, lattice to segment start, DDA setup",
         "end-of-walk times (3 closed forms), major axis", "seam restart (closed-form DDA state at the part start)",
         "A: voxel walk of the part", "B: lattice position of own boundaries", "stitch across the P lanes + run records",
         "per-ray outputs (+ serial fallback)", "block sums"]
tot = sum(buf[i] for i in range(12))
print(f"rays {n}  P={os.environ.get('NFA_SPLIT_P', 'auto')}  waves/launch {waves / iters:.0f}  cycles/wave {tot / max(waves, 1):.0f}")
for i, nm in enumerate(names):
    print(f"  {buf[i] / max(waves, 1):9.0f} cyc  {100.0 * buf[i] / max(tot, 1):5.1f} %  {nm}")
for i, nm in ((9, "  (crossing-time form) before the closed-form call"), (10, "  (crossing-time form) the closed-form call"),
              (11, "  (crossing-time form) last lattice steps / chain segments")):
    if buf[i]:
        print(f"  {buf[i] / max(waves, 1):9.0f} cyc  {100.0 * buf[i] / max(tot, 1):5.1f} %  {nm}")
