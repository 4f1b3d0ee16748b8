"""One-off differential fuzz: random single-level grids (blobs / noise / planes / checkerboards), random
rays (outside, inside, axis-aligned, grazing), random steps and near/far planes; the fused sampling
call under every lanes-per-ray setting vs the CPU oracle, bit for bit.

    python tools/fuzz_campaign.py [n_cases] [seed]
"""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle
from nerfacc_amd import cuda as C

dev = torch.device("cuda:0")
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
bad = 0
total_samples = 0
nonempty = 0
t_begin = time.time()
for case in range(n_cases):
    g = np.random.default_rng(seed0 * 100003 + case)
    res = [int(g.choice([16, 24, 32, 48, 64, 96, 128])) for _ in range(3)]
    if g.random() < 0.5:
        res = [res[0]] * 3
    kind = g.integers(0, 5)
    c = [np.arange(r) for r in res]
    X, Y, Z = np.meshgrid(*c, indexing="ij")
    if kind == 0:
        occ = g.random(res) > g.choice([0.5, 0.9, 0.98])
    elif kind == 1:
        ctr = [r * g.uniform(0.3, 0.7) for r in res]
        rad = min(res) * g.uniform(0.1, 0.45)
        occ = (X - ctr[0]) ** 2 + (Y - ctr[1]) ** 2 + (Z - ctr[2]) ** 2 < rad**2
    elif kind == 2:
        occ = (X + Y + Z) % int(g.choice([2, 3, 5])) == 0
    elif kind == 3:
        occ = (X % int(g.integers(2, 9)) == 0) | (Z == res[2] // 2)
    else:
        occ = np.ones(res, bool) if g.random() < 0.5 else np.zeros(res, bool)
        occ[tuple(g.integers(0, r) for r in res)] ^= True
    lo = g.uniform(-2, 0, 3).astype(np.float32)
    hi = (lo + g.uniform(0.5, 3, 3)).astype(np.float32)
    aabb = np.concatenate([lo, hi])[None].astype(np.float32)
    R = int(g.choice([1, 7, 64, 500, 3000, 9000, 20000, 40000]))
    ctr, ext = (lo + hi) / 2, (hi - lo)
    mode = g.integers(0, 4)
    if mode == 0:                                       # from outside towards the box
        v = g.normal(size=(R, 3)); v /= np.linalg.norm(v, axis=1, keepdims=True)
        o = ctr + v * ext.max() * g.uniform(0.8, 2.5)
        d = (lo + g.random((R, 3)) * ext) - o
    elif mode == 1:                                     # origins inside
        o = lo + g.random((R, 3)) * ext
        d = g.normal(size=(R, 3))
    elif mode == 2:                                     # axis-aligned / planar directions
        o = ctr + g.normal(size=(R, 3)) * ext
        d = g.normal(size=(R, 3)); d[np.arange(R), g.integers(0, 3, R)] = 0.0
    else:                                               # grazing along faces / voxel planes
        o = lo + np.round(g.random((R, 3)) * np.array(res)) / np.array(res) * ext
        d = g.normal(size=(R, 3)) * np.array([1.0, 1e-3, 1.0])
    nrm = np.linalg.norm(d, axis=1, keepdims=True); nrm[nrm == 0] = 1
    o, d = o.astype(np.float32), (d / nrm).astype(np.float32)
    step = float(np.float32(ext.max() / g.choice([40, 150, 600, 2000])))
    near = (g.random(R) * step * g.choice([0.0, 1.0, 50.0])).astype(np.float32)
    far = np.full(R, 1e10, np.float32) if g.random() < 0.7 else (near + g.random(R).astype(np.float32) * 3).astype(np.float32)
    r_iv, r_sm, _ = oracle.traverse_grids(o, d, occ[None], aabb, near, far, step, 0.0)
    r_ri, r_ts, r_te = r_sm["ray_indices"], r_iv["vals"][r_iv["is_left"]], r_iv["vals"][r_iv["is_right"]]
    total_samples += len(r_ri)
    nonempty += len(r_ri) > 0
    O, D, OCC, AABB, NEAR, FAR = T(o), T(d), T(occ[None]), T(aabb), T(near), T(far)
    for p in ("", "1", "2", "4", "8", "16"):
        if p:
            os.environ["NFA_SPLIT_P"] = p
        else:
            os.environ.pop("NFA_SPLIT_P", None)
        ri, ts, te, pk = C.sample_occgrid(O, D, OCC, AABB, NEAR, FAR, step, 0.0)
        ok = (np.array_equal(ri.cpu().numpy(), r_ri) and np.array_equal(ts.cpu().numpy(), r_ts)
              and np.array_equal(te.cpu().numpy(), r_te) and np.array_equal(pk.cpu().numpy(), r_sm["packed_info"]))
        if not ok:
            bad += 1
            print(f"MISMATCH case {case} P={p or 'auto'} res={res} kind={kind} mode={mode} R={R} step={step} samples {len(r_ri)} vs {ri.shape[0]}", flush=True)
    os.environ.pop("NFA_SPLIT_P", None)
print(f"{n_cases} cases x 6 lane settings ({nonempty} with samples, {total_samples} oracle samples in total), {bad} mismatches, {time.time() - t_begin:.0f} s")

# ---- second campaign: the reference-API call (traverse_grids) on multi-level grids, with cone angles,
# per-voxel mode (step <= 0), step limits + over-allocation and ray masks
from nerfacc_amd.grid import traverse_grids
bad2 = 0
tot2 = 0
t_begin = time.time()
for case in range(n_cases // 4):
    g = np.random.default_rng(seed0 * 7919 + 5000 + case)
    levels = int(g.integers(1, 5))
    res = int(g.choice([8, 16, 32, 64]))
    occ = g.random((levels, res, res, res)) > g.choice([0.5, 0.8, 0.97])
    base = np.array([-1, -1, -1, 1, 1, 1], np.float32)
    aabbs = np.stack([base * 2.0**l for l in range(levels)]).astype(np.float32)
    R = int(g.choice([3, 100, 2000, 12000]))
    o = (g.normal(size=(R, 3)) * g.choice([0.3, 1.5, 6.0])).astype(np.float32)
    d = g.normal(size=(R, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True); d = d.astype(np.float32)
    step = float(np.float32(g.choice([-1.0, 2e-2, 5e-3])))
    cone = float(g.choice([0.0, 0.0, 0.004, 0.02]))
    near = (g.random(R) * 0.2).astype(np.float32)
    far = np.full(R, float(g.choice([1e10, 3.0])), np.float32)
    kw = {}
    if g.random() < 0.35:
        kw = dict(traverse_steps_limit=int(g.integers(1, 20)), over_allocate=bool(g.random() < 0.5))
        if kw["over_allocate"]:
            kw["rays_mask"] = g.random(R) < 0.7
    r_iv, r_sm, r_term = oracle.traverse_grids(o, d, occ, aabbs, near, far, step, cone, **kw)
    tkw = {k: (T(v) if isinstance(v, np.ndarray) else v) for k, v in kw.items()}
    iv, sm, term = traverse_grids(T(o), T(d), T(occ), T(aabbs), T(near), T(far), step, cone, **tkw)
    tot2 += int(r_sm["packed_info"][:, 1].sum())
    n_ = lambda x: x.cpu().numpy()
    ok = (np.array_equal(n_(sm.packed_info), r_sm["packed_info"]) and np.array_equal(n_(iv.packed_info), r_iv["packed_info"])
          and np.array_equal(n_(iv.vals), r_iv["vals"]) and np.array_equal(n_(sm.vals), r_sm["vals"])
          and np.array_equal(n_(sm.ray_indices), r_sm["ray_indices"])
          and np.array_equal(n_(iv.is_left), r_iv["is_left"]) and np.array_equal(n_(iv.is_right), r_iv["is_right"])
          and np.array_equal(n_(sm.is_valid), r_sm["is_valid"]))
    live = r_sm["packed_info"][:, 1] > 0
    if "rays_mask" in kw:
        live &= kw["rays_mask"]
    ok = ok and np.array_equal(n_(term)[live], r_term[live])
    if not ok:
        bad2 += 1
        print(f"MISMATCH (API) case {case} levels={levels} res={res} R={R} step={step} cone={cone} kw={ {k: v for k, v in kw.items() if k != 'rays_mask'} }", flush=True)
print(f"traverse_grids API: {n_cases // 4} cases ({tot2} oracle samples), {bad2} mismatches, {time.time() - t_begin:.0f} s")
