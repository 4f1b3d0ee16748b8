"""Round 5 A/B of the count pass (and emit) between two builds of the library on one box:
    NERFACC_AMD_BACKEND=ctypes NERFACC_AMD_LIB=<lib.so> python tools/experiments/r05_count_ab.py <tag> [--quick]
Workloads = VERDICT r4 item 1's "Done" list: the bench's steady state at 6.5 k rays and tiled up to 10^6, 256^3 lego at 8 k rays,
4 x 128^3 at 4 k rays, plus the rand > 0.5 grid.  HIP events around the count / emit C-ABI calls (KernelTimer); one JSON line per row."""
import json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import nerfacc_amd, scenes
from nerfacc_amd import cuda as C
from nerfacc_amd.cuda import _backend

tag = sys.argv[1] if len(sys.argv) > 1 else "lib"
try:
    nerfacc_amd.set_option("skip", None)
    HAS_SKIP = True
except Exception:
    HAS_SKIP = False
quick = "--quick" in sys.argv
dev = torch.device("cuda:0")
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def timed(fn, reps):
    for _ in range(3):
        out = fn()
    torch.cuda.synchronize()
    tm = _backend.KernelTimer(names=("traverse_count", "traverse_fill"))
    _backend.set_kernel_timer(tm)
    for _ in range(reps):
        fn()
    s = tm.summary()
    _backend.set_kernel_timer(None)
    return out, s["traverse_count"][1] * 1e3, s["traverse_fill"][1] * 1e3


def row(name, rays, out, c, e, **kw):
    ri, ts, te = out[0], out[1], out[2]
    d = dict(tag=tag, workload=name, rays=rays, samples=int(ri.shape[0]), count_us=round(c, 1), emit_us=round(e, 1),
             digest=[int(ri.sum()), float(ts.double().sum()), float(te.double().sum())], **kw)
    print(json.dumps(d), flush=True)


st = np.load(os.path.join(ROOT, "profiles", "r02_sampling_state.npz"))
res = tuple(int(x) for x in st["res"])
binaries = T(np.unpackbits(st["binaries_bits"])[: int(np.prod(res))].astype(bool).reshape(res))
aabbs = T(st["aabbs"])
step = float(st["render_step"])
for n in ((6564, 13000, 50000, 200000, 1000000) if not quick else (6564, 1000000)):
    rep = -(-n // st["rays_o"].shape[0])
    O = T(np.tile(st["rays_o"], (rep, 1))[:n]); D = T(np.tile(st["rays_d"], (rep, 1))[:n]); J = T(np.tile(st["jitter"], rep)[:n])
    call = lambda: C.sample_occgrid(O, D, binaries, aabbs, None, None, step, 0.0, near_plane=0.0, far_plane=1e10, jitter=J, jitter_scale=step)
    out, c, e = timed(call, 20 if n < 500000 else 8)
    row("bench state 128^3", n, out, c, e)
    if n in (50000, 200000, 1000000):
        forms = [("P1 lds", dict(split_p=1, count_l2=0)), ("P1 l2", dict(split_p=1, count_l2=1)), ("P8", dict(split_p=8))]
        if HAS_SKIP:
            forms += [(f"P1 {w} skip{k}", dict(split_p=1, count_l2=c, skip=k)) for w, c in (("lds", 0), ("l2", 1)) for k in (0, 1, 2)]
        for form, f in forms:
            if form == "P8" and n > 200000:
                continue
            with nerfacc_amd.options(**f):
                out2, c2, e2 = timed(call, 8)
            assert all(torch.equal(a, b) for a, b in zip(out, out2)), form
            row("bench state 128^3", n, out2, c2, e2, form=form)

# pixel-ordered rays of one 1000 x 1000 frame (what the test-time marcher and a whole-frame sampling call see): neighbouring lanes walk
# neighbouring paths
if "--coherent" in sys.argv or True:
    H = W = 1000
    cam = np.array([0.0, 0.6, 4.0], np.float32)
    jj, ii = np.meshgrid(np.arange(H, dtype=np.float32), np.arange(W, dtype=np.float32), indexing="ij")
    dirs = np.stack([(ii - W / 2 + 0.5) / (1.2 * W), -(jj - H / 2 + 0.5) / (1.2 * H) - 0.15, -np.ones_like(ii)], -1).reshape(-1, 3)
    dirs = (dirs / np.linalg.norm(dirs, axis=1, keepdims=True)).astype(np.float32)
    O, D = T(np.broadcast_to(cam, dirs.shape)), T(dirs)
    n = O.shape[0]
    NEAR, FAR = torch.zeros(n, device=dev), torch.full((n,), 1e10, device=dev)
    call = lambda: C.sample_occgrid(O, D, binaries, aabbs, NEAR, FAR, step, 0.0)
    out, c, e = timed(call, 6)
    row("frame 1000x1000 128^3", n, out, c, e)
    forms = [("P1 l2", dict(split_p=1, count_l2=1)), ("P1 lds", dict(split_p=1, count_l2=0))]
    if HAS_SKIP:
        forms += [(f"P1 {w} skip{k}", dict(split_p=1, count_l2=c_, skip=k)) for w, c_ in (("lds", 0), ("l2", 1)) for k in (0, 1, 2)]
    for form, f in forms:
        with nerfacc_amd.options(**f):
            out2, c2, e2 = timed(call, 6)
        assert all(torch.equal(a, b) for a, b in zip(out, out2)), form
        row("frame 1000x1000 128^3", n, out2, c2, e2, form=form)
    # one round of the test-time marcher's shape: a step limit of 4 samples per ray and a mask (lane-per-ray lattice kernel)
    mask = torch.ones(n, dtype=torch.bool, device=dev)
    for form, f in ([("limit4 skip0", dict(skip=0)), ("limit4 skip1", dict(skip=1))] if HAS_SKIP else [("limit4", {})]):
        with nerfacc_amd.options(**f):
            out3, c3, e3 = timed(lambda: C.sample_occgrid(O, D, binaries, aabbs, NEAR, FAR, step, 0.0, mask, 4), 6)
        row("frame 1000x1000 128^3", n, out3, c3, e3, form=form)

for name, r, n in (("lego", 256, 8192), ("lego", 256, 100000), ("drums", 256, 8192), ("noise", 128, 8192), ("lego", 128, 100000)):
    occ = T(scenes.occupancy_grid(name, r)); ab = T(scenes.AABB[None].copy())
    o, d = (T(x) for x in scenes.rays(n, seed=11))
    near, far = torch.zeros(n, device=dev), torch.full((n,), 1e10, device=dev)
    call = lambda: C.sample_occgrid(o, d, occ, ab, near, far, 5e-3, 0.0)
    out, c, e = timed(call, 10)
    row(f"{name} {r}^3", n, out, c, e)
    forms = [("P1", dict(split_p=1))] + ([("P1 skip0", dict(split_p=1, skip=0))] if HAS_SKIP else [])
    for form, f in forms:
        with nerfacc_amd.options(**f):
            out2, c2, e2 = timed(call, 6)
        assert all(torch.equal(a, b) for a, b in zip(out, out2)), form
        row(f"{name} {r}^3", n, out2, c2, e2, form=form)

# 4 x 128^3 (tools/multilevel_bench.py's scene), step 1e-3, cone 0 and 0.004
g = np.random.default_rng(0)
c_ = (np.arange(128) + 0.5) / 128 * 2 - 1
X, Y, Z = np.meshgrid(c_, c_, c_, indexing="ij")
occ = np.stack([((X * 2.0**l) ** 2 + (Y * 2.0**l) ** 2 + (Z * 2.0**l) ** 2 < 0.25) | (g.random((128, 128, 128)) < (0.002 if l else 0.0)) for l in range(4)])
ab = np.stack([np.array([-1, -1, -1, 1, 1, 1], np.float32) * 2.0**l for l in range(4)])
for R in (4096, 65536):
    v = g.normal(size=(R, 3)); v /= np.linalg.norm(v, axis=1, keepdims=True)
    dd = g.normal(size=(R, 3)); dd /= np.linalg.norm(dd, axis=1, keepdims=True)
    O, D, OCC, AB = T((0.6 * v).astype(np.float32)), T(dd.astype(np.float32)), T(occ), T(ab)
    NEAR, FAR = torch.full((R,), 0.2, device=dev), torch.full((R,), 1e10, device=dev)
    for cone in (0.0, 0.004):
        call = lambda: C.sample_occgrid(O, D, OCC, AB, NEAR, FAR, 1e-3, cone)
        out, c, e = timed(call, 10)
        row(f"4 x 128^3 cone {cone:g}", R, out, c, e)
        if cone == 0.0:
            forms = [("seg0", dict(segments=0))] + ([("seg0 skip0", dict(segments=0, skip=0))] if HAS_SKIP else [])
            for form, f in forms:
                with nerfacc_amd.options(**f):
                    out2, c2, e2 = timed(call, 6)
                assert all(torch.equal(a, b) for a, b in zip(out, out2)), form
                row(f"4 x 128^3 cone {cone:g}", R, out2, c2, e2, form=form)
