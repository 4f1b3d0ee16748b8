"""Where do the cycles of traverse_count_cone_kernel go?  Runs the several-level cone-angle call of tools/multilevel_bench.py against the
instrumented library (tools/phase_cycles.py --build-only: -DNFA_PHASE_CYCLES) and prints average cycles per wave and phase.
    python tools/phase_cycles.py --build-only && python tools/cone_phases.py [n_rays] [iters]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
so = os.path.join(ROOT, "tools", "_prof", "libnerfacc_hip_prof.so")
os.environ["NERFACC_AMD_BACKEND"] = "ctypes"
os.environ["NERFACC_AMD_LIB"] = so
import numpy as np, torch
from nerfacc_amd.cuda import _backend
from nerfacc_amd import cuda as C
assert _backend.LIB_PATH == so
R = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dev = torch.device("cuda:0")
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
g = np.random.default_rng(0)
res, levels = 128, 4
c = (np.arange(res) + 0.5) / res * 2 - 1
X, Y, Z = np.meshgrid(c, c, c, indexing="ij")
occ = np.stack([((X * 2.0**l) ** 2 + (Y * 2.0**l) ** 2 + (Z * 2.0**l) ** 2 < 0.25) | (g.random((res, res, res)) < (0.002 if l else 0.0)) for l in range(levels)])
aabbs = np.stack([np.array([-1, -1, -1, 1, 1, 1], np.float32) * 2.0**l for l in range(levels)])
v = g.normal(size=(R, 3)); v /= np.linalg.norm(v, axis=1, keepdims=True)
o = (0.6 * v).astype(np.float32)
d = g.normal(size=(R, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True); d = d.astype(np.float32)
O, D, OCC, AABB = T(o), T(d), T(occ), T(aabbs)
NEAR, FAR = T(np.full(R, 0.2, np.float32)), T(np.full(R, 1e10, np.float32))
cone = float(os.environ.get("CONE", "0.004"))
L = _backend.load_library()
out = (ctypes.c_ulonglong * 16)()
for _ in range(3):
    C.sample_occgrid(O, D, OCC, AABB, NEAR, FAR, 1e-3, cone)
torch.cuda.synchronize()
L2 = ctypes.CDLL(so)
L2.nfa_debug_phase_cycles(out, 1)
for _ in range(iters):
    C.sample_occgrid(O, D, OCC, AABB, NEAR, FAR, 1e-3, cone)
torch.cuda.synchronize()
L2.nfa_debug_phase_cycles(out, 0)
n = out[15]
print(f"{R} rays, cone {cone}: {n} wave records; average per wave:")
names = ["stage", "ray setup + segments", "crossing-time chains + seams", "voxel walk -> records", "barrier", "chain (phase 2)", "publish"]
for i, nm in enumerate(names):
    print(f"  {nm:32s} {out[i] / max(n, 1):10.0f} cycles")
print(f"  first ray of a wave: records {out[10] / max(n, 1):.0f}, fast-loop steps {out[11] / max(n, 1):.0f}, careful-loop steps {out[8] / max(n, 1):.0f}, lists {out[12] / max(n, 1):.1f}, steps to segment starts {out[13] / max(n, 1):.0f}")
print("  (the counters live in the chain's loops: with them the chain runs slower than in the shipped build — read the phases of the walk, not of the chain, from this build)")
