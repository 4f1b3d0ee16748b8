"""Where does the fused sampling launch (csrc/sample_fused.hpp) spend its time?  Builds an instrumented copy of the library
(-DNFA_FUSE_TRACE: 100 MHz wall-clock stamps per workgroup at kernel entry, after its waves have counted, after its look-back and
after its last wave's emit), replays bench.py's steady state through the ctypes face and prints the distribution.
    python tools/fuse_trace.py [state.npz] [reps]      (run on the GPU box; --build-only compiles here)"""
import ctypes, glob, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
out_dir = os.path.join(ROOT, "tools", "_prof")
os.makedirs(out_dir, exist_ok=True)
extra = os.environ.get("NFA_TRACE_EXTRA", "").split()
so = os.path.join(out_dir, "libnerfacc_hip_fusetrace" + "".join(c if c.isalnum() else "_" for c in "".join(extra)) + ".so")
srcs = sorted(glob.glob(os.path.join(ROOT, "nerfacc_amd", "csrc", "*.hip")))
hdrs = glob.glob(os.path.join(ROOT, "nerfacc_amd", "csrc", "*.hpp")) + [os.path.join(ROOT, "include", "nerfacc_hip.h")]
if not os.path.exists(so) or any(os.path.getmtime(f) > os.path.getmtime(so) for f in srcs + hdrs):
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
                           "-fvisibility=hidden", "-DNFA_FUSE_TRACE", *extra, "-shared", *srcs, "-o", so])
if "--build-only" in sys.argv:
    sys.exit(0)
os.environ["NERFACC_AMD_BACKEND"] = "ctypes"
os.environ["NERFACC_AMD_LIB"] = so
import numpy as np, torch
from nerfacc_amd.cuda import _backend
assert _backend.LIB_PATH == so and _backend.BACKEND == "ctypes"
args = [a for a in sys.argv[1:] if not a.startswith("--")]
st = np.load(args[0] if args else os.path.join(ROOT, "profiles", "r02_sampling_state.npz"))
reps = int(args[1]) if len(args) > 1 else 20
dev = torch.device("cuda:0")
res = tuple(int(x) for x in st["res"])
binaries = torch.from_numpy(np.unpackbits(st["binaries_bits"])[: int(np.prod(res))].astype(bool).reshape(res)).to(dev)
aabbs = torch.from_numpy(st["aabbs"]).to(dev)
O, D = torch.from_numpy(st["rays_o"]).to(dev), torch.from_numpy(st["rays_d"]).to(dev)
jit = torch.from_numpy(st["jitter"]).to(dev)
step = float(st["render_step"])
C = _backend._C
L = _backend.load_library()
call = lambda: C.sample_occgrid(O, D, binaries, aabbs, None, None, step, 0.0, near_plane=0.0, far_plane=1e10, jitter=jit, jitter_scale=step)
for _ in range(3):
    call()
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * (512 * 4))()
L.nfa_debug_fuse_trace.argtypes = [ctypes.c_void_p]
L.nfa_debug_fuse_trace(buf)                       # clear
acc = []
for _ in range(reps):
    call()
    torch.cuda.synchronize()
    assert L.nfa_debug_fuse_trace(buf) == 0
    t = np.frombuffer(buf, dtype=np.uint64).reshape(512, 4).astype(np.int64)
    t = t[t[:, 0] > 0]
    acc.append((t - t[:, 0].min()) * 0.01)          # us since the first workgroup's entry
T = np.stack(acc)                                  # [reps, blocks, 4]
nb = T.shape[1]
q = lambda x: "min %6.2f  p50 %6.2f  p90 %6.2f  max %6.2f" % (x.min(), np.median(x), np.percentile(x, 90), x.max())
print(f"{nb} workgroups, {reps} launches; microseconds since the first workgroup entered the kernel")
print("entry                       ", q(T[:, :, 0]))
print("all waves counted           ", q(T[:, :, 1]))
print("look-back through           ", q(T[:, :, 2]))
print("emit ended (last wave)      ", q(T[:, :, 3]))
print("look-back wait  (2 - 1)     ", q(T[:, :, 2] - T[:, :, 1]))
print("  lag behind the slowest predecessor (2 - running max of 1)", q(T[:, :, 2] - np.maximum.accumulate(T[:, :, 1], axis=1)))
print("emit            (3 - 2)     ", q(T[:, :, 3] - T[:, :, 2]))
print("kernel end (max of 3) per launch: ", np.round(T[:, :, 3].max(axis=1), 2)[:10])
print("last count (max of 1) per launch: ", np.round(T[:, :, 1].max(axis=1), 2)[:10])
