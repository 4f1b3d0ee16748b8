"""Differential fuzz of the fused sampling call on MULTI-LEVEL grids (cone_angle = 0): random level counts (2..8), resolutions,
occupancy kinds, rays from inside / outside / axis-aligned, near / far planes; the segment-per-lane count pass
(traverse_count_segments_kernel, NFA_SEGMENTS=1) and the lane-per-ray one (NFA_SEGMENTS=0) vs the CPU oracle, bit for bit
(ray_indices, t_starts, t_ends, packed_info, terminate planes).

    python tools/fuzz_levels.py [n_cases] [seed]
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
bad = total = nonempty = 0
t_begin = time.time()
for case in range(n_cases):
    g = np.random.default_rng(seed0 * 1000003 + case)
    levels = int(g.choice([2, 2, 3, 4, 4, 5, 8]))
    res = [int(g.choice([8, 16, 24, 32, 48, 64]))] * 3 if g.random() < 0.6 else [int(g.choice([8, 16, 32, 48])) for _ in range(3)]
    c = [(np.arange(r) + 0.5) / r * 2 - 1 for r in res]
    X, Y, Z = np.meshgrid(*c, indexing="ij")
    occ = []
    kind = int(g.integers(0, 4))
    for l in range(levels):
        if kind == 0:
            o_l = g.random(res) > g.choice([0.5, 0.9, 0.98])
        elif kind == 1:
            s = 2.0**l
            o_l = ((X * s) ** 2 + (Y * s) ** 2 + (Z * s) ** 2 < g.uniform(0.2, 0.9) ** 2) | (g.random(res) < 0.003 * (l > 0))
        elif kind == 2:
            o_l = np.ones(res, bool) if (l + case) % 2 else np.zeros(res, bool)
        else:
            o_l = (np.indices(res).sum(0) % int(g.choice([2, 3, 5])) == 0)
        occ.append(o_l)
    occ = np.stack(occ)
    base = np.concatenate([g.uniform(-1.5, -0.5, 3), g.uniform(0.5, 1.5, 3)]).astype(np.float32)
    ctr, half = (base[:3] + base[3:]) / 2, (base[3:] - base[:3]) / 2
    aabbs = np.stack([np.concatenate([ctr - half * 2.0**l, ctr + half * 2.0**l]) for l in range(levels)]).astype(np.float32)
    R = int(g.choice([1, 5, 64, 700, 4096, 9000]))
    mode = int(g.integers(0, 4))
    if mode == 0:                                   # inside the first level
        o = ctr + (g.random((R, 3)) * 2 - 1) * half * 0.9
        d = g.normal(size=(R, 3))
    elif mode == 1:                                 # anywhere inside the last level
        o = ctr + (g.random((R, 3)) * 2 - 1) * half * 2.0 ** (levels - 1)
        d = g.normal(size=(R, 3))
    elif mode == 2:                                 # from outside everything, through the centre region
        v = g.normal(size=(R, 3)); v /= np.linalg.norm(v, axis=1, keepdims=True)
        o = ctr + v * half.max() * 2.0**levels
        d = (ctr + (g.random((R, 3)) * 2 - 1) * half * g.choice([1.0, 4.0])) - o
    else:                                           # axis-aligned / planar
        o = ctr + g.normal(size=(R, 3)) * half
        d = g.normal(size=(R, 3)); d[np.arange(R), g.integers(0, 3, R)] = 0.0
    nrm = np.linalg.norm(d, axis=1, keepdims=True); nrm[nrm == 0] = 1
    o, d = o.astype(np.float32), (d / nrm).astype(np.float32)
    step = float(np.float32(half.max() / g.choice([20, 100, 400, 1500])))
    near = (g.random(R) * step * g.choice([0.0, 1.0, 50.0])).astype(np.float32)
    far = np.full(R, 1e10, np.float32) if g.random() < 0.6 else (near + g.random(R).astype(np.float32) * half.max() * 2.0**levels).astype(np.float32)
    r_iv, r_sm, r_term = oracle.traverse_grids(o, d, occ, aabbs, near, far, step, 0.0)
    r_ri, r_ts, r_te = r_sm["ray_indices"], r_iv["vals"][r_iv["is_left"]], r_iv["vals"][r_iv["is_right"]]
    total += len(r_ri)
    nonempty += len(r_ri) > 0
    O, D, OCC, AABB, NEAR, FAR = T(o), T(d), T(occ), T(aabbs), T(near), T(far)
    live_rays = r_sm["packed_info"][:, 1] > 0        # (the reference leaves the terminate plane of a ray without samples unwritten)
    for seg in ("1", "0"):
        os.environ["NFA_SEGMENTS"] = seg
        ri, ts, te, pk, term = C.sample_occgrid(O, D, OCC, AABB, NEAR, FAR, step, 0.0, with_terminate_planes=True)
        ok = (np.array_equal(ri.cpu().numpy(), r_ri) and np.array_equal(ts.cpu().numpy(), r_ts)
              and np.array_equal(te.cpu().numpy(), r_te) and np.array_equal(pk.cpu().numpy(), r_sm["packed_info"])
              and np.array_equal(term.cpu().numpy()[live_rays], r_term[live_rays]))
        if not ok:
            bad += 1
            what = [k for k, v in (("ri", np.array_equal(ri.cpu().numpy(), r_ri)), ("ts", np.array_equal(ts.cpu().numpy(), r_ts)),
                                   ("pk", np.array_equal(pk.cpu().numpy(), r_sm["packed_info"])),
                                   ("term", np.array_equal(term.cpu().numpy()[live_rays], r_term[live_rays]))) if not v]
            print(f"MISMATCH case {case} seg={seg} levels={levels} res={res} kind={kind} mode={mode} R={R} step={step} "
                  f"samples {len(r_ri)} vs {ri.shape[0]} differs: {what}", flush=True)
    os.environ.pop("NFA_SEGMENTS", None)
print(f"{n_cases} cases x 2 count passes ({nonempty} with samples, {total} oracle samples in total), {bad} mismatches, {time.time() - t_begin:.0f} s")
