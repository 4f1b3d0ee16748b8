"""Random traversal scenes + bit-for-bit checkers (HIP vs the CPU oracle), shared by the time-boxed randomised test of
the -m gpu suite (tests/test_gpu_fuzz.py) and by the long differential campaigns (tools/fuzz_campaign.py, tools/fuzz_levels.py).

Three families:
  fused_single  one-level grids (noise / blob / lattice / planes / single voxel), rays from outside / inside /
                axis-aligned / grazing, 4 step sizes, jittered near planes, finite far planes; the fused sampling call
                under every lanes-per-ray setting (NFA_SPLIT_P)
  fused_levels  2..8 levels, cone_angle = 0 or > 0; the segment-per-lane and the lane-per-ray count passes (NFA_SEGMENTS)
  api           the reference-API call `traverse_grids` on 1..4 levels with cone angles, per-voxel mode (step <= 0),
                step limits, over-allocation and ray masks
Each checker returns a list of mismatch descriptions (empty = equal) and the number of oracle samples compared."""
import os

import numpy as np
import torch

import oracle

DEV = "cuda:0"


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def _n(x):
    return x.detach().cpu().numpy()


def _unit(d):
    nrm = np.linalg.norm(d, axis=1, keepdims=True)
    nrm[nrm == 0] = 1
    return (d / nrm).astype(np.float32)


# ------------------------------------------------------------------------------------------ fused, one level
def fused_single_case(g, ray_counts=(1, 7, 64, 500, 3000, 9000, 20000, 40000), cones=(0.0,)):
    res = [int(g.choice([16, 24, 32, 48, 64, 96, 128])) for _ in range(3)]
    if g.random() < 0.5:
        res = [res[0]] * 3
    kind = int(g.integers(0, 5))
    X, Y, Z = np.meshgrid(*[np.arange(r) for r in res], indexing="ij")
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
    R = int(g.choice(ray_counts))
    ctr, ext = (lo + hi) / 2, (hi - lo)
    mode = int(g.integers(0, 4))
    if mode == 0:                                       # from outside towards the box
        v = g.normal(size=(R, 3))
        v /= np.linalg.norm(v, axis=1, keepdims=True)
        o = ctr + v * ext.max() * g.uniform(0.8, 2.5)
        d = (lo + g.random((R, 3)) * ext) - o
    elif mode == 1:                                     # origins inside
        o = lo + g.random((R, 3)) * ext
        d = g.normal(size=(R, 3))
    elif mode == 2:                                     # axis-aligned / planar directions
        o = ctr + g.normal(size=(R, 3)) * ext
        d = g.normal(size=(R, 3))
        d[np.arange(R), g.integers(0, 3, R)] = 0.0
    else:                                               # grazing along faces / voxel planes
        o = lo + np.round(g.random((R, 3)) * np.array(res)) / np.array(res) * ext
        d = g.normal(size=(R, 3)) * np.array([1.0, 1e-3, 1.0])
    step = float(np.float32(ext.max() / g.choice([40, 150, 600, 2000])))
    near = (g.random(R) * step * g.choice([0.0, 1.0, 50.0])).astype(np.float32)
    far = np.full(R, 1e10, np.float32) if g.random() < 0.7 else (near + g.random(R).astype(np.float32) * 3).astype(np.float32)
    cone = float(g.choice(cones))
    return dict(o=o.astype(np.float32), d=_unit(d), occ=occ[None], aabbs=aabb, near=near, far=far, step=step, cone=cone,
                desc=f"fused_single res={res} kind={kind} mode={mode} R={R} step={step} cone={cone}")


# ------------------------------------------------------------------------------------------ fused, several levels
def fused_levels_case(g, ray_counts=(1, 5, 64, 700, 4096, 9000), cones=(0.0,)):
    levels = int(g.choice([2, 2, 3, 4, 4, 5, 8]))
    res = [int(g.choice([8, 16, 24, 32, 48, 64]))] * 3 if g.random() < 0.6 else [int(g.choice([8, 16, 32, 48])) for _ in range(3)]
    X, Y, Z = np.meshgrid(*[(np.arange(r) + 0.5) / r * 2 - 1 for r in res], indexing="ij")
    kind = int(g.integers(0, 4))
    flip = int(g.integers(0, 2))
    occ = []
    for l in range(levels):
        if kind == 0:
            o_l = g.random(res) > g.choice([0.5, 0.9, 0.98])
        elif kind == 1:
            s = 2.0**l
            o_l = ((X * s) ** 2 + (Y * s) ** 2 + (Z * s) ** 2 < g.uniform(0.2, 0.9) ** 2) | (g.random(res) < 0.003 * (l > 0))
        elif kind == 2:
            o_l = np.ones(res, bool) if (l + flip) % 2 else np.zeros(res, bool)
        else:
            o_l = (np.indices(res).sum(0) % int(g.choice([2, 3, 5])) == 0)
        occ.append(o_l)
    occ = np.stack(occ)
    base = np.concatenate([g.uniform(-1.5, -0.5, 3), g.uniform(0.5, 1.5, 3)]).astype(np.float32)
    ctr, half = (base[:3] + base[3:]) / 2, (base[3:] - base[:3]) / 2
    aabbs = np.stack([np.concatenate([ctr - half * 2.0**l, ctr + half * 2.0**l]) for l in range(levels)]).astype(np.float32)
    R = int(g.choice(ray_counts))
    mode = int(g.integers(0, 4))
    if mode == 0:                                   # inside the first level
        o = ctr + (g.random((R, 3)) * 2 - 1) * half * 0.9
        d = g.normal(size=(R, 3))
    elif mode == 1:                                 # anywhere inside the last level
        o = ctr + (g.random((R, 3)) * 2 - 1) * half * 2.0 ** (levels - 1)
        d = g.normal(size=(R, 3))
    elif mode == 2:                                 # from outside everything, through the centre region
        v = g.normal(size=(R, 3))
        v /= np.linalg.norm(v, axis=1, keepdims=True)
        o = ctr + v * half.max() * 2.0**levels
        d = (ctr + (g.random((R, 3)) * 2 - 1) * half * g.choice([1.0, 4.0])) - o
    else:                                           # axis-aligned / planar
        o = ctr + g.normal(size=(R, 3)) * half
        d = g.normal(size=(R, 3))
        d[np.arange(R), g.integers(0, 3, R)] = 0.0
    step = float(np.float32(half.max() / g.choice([20, 100, 400, 1500])))
    near = (g.random(R) * step * g.choice([0.0, 1.0, 50.0])).astype(np.float32)
    far = (np.full(R, 1e10, np.float32) if g.random() < 0.6
           else (near + g.random(R).astype(np.float32) * half.max() * 2.0**levels).astype(np.float32))
    cone = float(g.choice(cones))
    return dict(o=o.astype(np.float32), d=_unit(d), occ=occ, aabbs=aabbs, near=near, far=far, step=step, cone=cone,
                desc=f"fused_levels levels={levels} res={res} kind={kind} mode={mode} R={R} step={step} cone={cone}")


def check_fused(case, env_name, env_values):
    """the fused sampling call (nerfacc_amd.cuda.sample_occgrid: count -> offsets -> emit) under every value of one tuning
    knob vs the oracle: ray_indices, t_starts, t_ends, packed_info, terminate planes of rays with samples"""
    from nerfacc_amd import cuda as C

    c = case
    r_iv, r_sm, r_term = oracle.traverse_grids(c["o"], c["d"], c["occ"], c["aabbs"], c["near"], c["far"], c["step"], c["cone"])
    r_ri, r_ts, r_te = r_sm["ray_indices"], r_iv["vals"][r_iv["is_left"]], r_iv["vals"][r_iv["is_right"]]
    live = r_sm["packed_info"][:, 1] > 0         # (the reference leaves the terminate plane of a ray without samples unwritten)
    args = (T(c["o"]), T(c["d"]), T(c["occ"]), T(c["aabbs"]), T(c["near"]), T(c["far"]), c["step"], c["cone"])
    bad = []
    # env_name None: every value is a list of assignments, "NFA_A=1,NFA_B=2".  (The names are the library's options —
    # nerfacc_amd.set_option accepts the NFA_ spelling; nothing goes through the environment.)
    import nerfacc_amd

    names = [env_name] if env_name else sorted({kv.split("=")[0] for v in env_values for kv in v.split(",") if kv})
    saved = {k: nerfacc_amd.get_option(k) for k in names}
    try:
        for v in env_values:
            for k in names:
                nerfacc_amd.set_option(k, None)
            if env_name and v:
                nerfacc_amd.set_option(env_name, v)
            elif not env_name:
                for kv in filter(None, v.split(",")):
                    nerfacc_amd.set_option(kv.split("=")[0], kv.split("=")[1])
            ri, ts, te, pk, term = C.sample_occgrid(*args, with_terminate_planes=True)
            diff = [k for k, same in (("ray_indices", np.array_equal(_n(ri), r_ri)), ("t_starts", np.array_equal(_n(ts), r_ts)),
                                      ("t_ends", np.array_equal(_n(te), r_te)), ("packed_info", np.array_equal(_n(pk), r_sm["packed_info"])),
                                      ("terminate_planes", np.array_equal(_n(term)[live], r_term[live]))) if not same]
            if diff:
                bad.append(f"{c['desc']} {env_name or ''}{'=' if env_name else ''}{v or 'auto'}: {diff} differ ({len(r_ri)} oracle samples, {ri.shape[0]} here)")
    finally:
        for k, old in saved.items():
            nerfacc_amd.set_option(k, old)
    return bad, len(r_ri)


# ------------------------------------------------------------------------------------------ reference API
def api_case(g, ray_counts=(3, 100, 2000, 12000)):
    levels = int(g.integers(1, 5))
    res = int(g.choice([8, 16, 32, 64]))
    occ = g.random((levels, res, res, res)) > g.choice([0.5, 0.8, 0.97])
    base = np.array([-1, -1, -1, 1, 1, 1], np.float32)
    aabbs = np.stack([base * 2.0**l for l in range(levels)]).astype(np.float32)
    R = int(g.choice(ray_counts))
    o = (g.normal(size=(R, 3)) * g.choice([0.3, 1.5, 6.0])).astype(np.float32)
    d = _unit(g.normal(size=(R, 3)))
    step = float(np.float32(g.choice([-1.0, 2e-2, 5e-3])))
    cone = float(g.choice([0.0, 0.0, 0.004, 0.02]))
    near = (g.random(R) * 0.2).astype(np.float32)
    far = np.full(R, float(g.choice([1e10, 3.0])), np.float32)
    kw = {}
    if g.random() < 0.35:
        kw = dict(traverse_steps_limit=int(g.integers(1, 20)), over_allocate=bool(g.random() < 0.5))
        if kw["over_allocate"]:
            kw["rays_mask"] = g.random(R) < 0.7
    return dict(o=o, d=d, occ=occ, aabbs=aabbs, near=near, far=far, step=step, cone=cone, kw=kw,
                desc=f"api levels={levels} res={res} R={R} step={step} cone={cone} kw={ {k: v for k, v in kw.items() if k != 'rays_mask'} }")


def check_api(case, emit_forms=("rays", "samples", "tiles")):
    """nerfacc_amd.grid.traverse_grids under both emit kernels vs the oracle, every output"""
    bad, n = [], 0
    import nerfacc_amd

    for v in emit_forms:
        with nerfacc_amd.options(emit=v):
            b, n = _check_api(case)
        bad += [f"{line} (emit={v})" for line in b]
    return bad, n


def _check_api(case):
    from nerfacc_amd.grid import traverse_grids

    c, kw = case, case["kw"]
    r_iv, r_sm, r_term = oracle.traverse_grids(c["o"], c["d"], c["occ"], c["aabbs"], c["near"], c["far"], c["step"], c["cone"], **kw)
    tkw = {k: (T(v) if isinstance(v, np.ndarray) else v) for k, v in kw.items()}
    iv, sm, term = traverse_grids(T(c["o"]), T(c["d"]), T(c["occ"]), T(c["aabbs"]), T(c["near"]), T(c["far"]), c["step"], c["cone"], **tkw)
    live = r_sm["packed_info"][:, 1] > 0
    if "rays_mask" in kw:
        live &= kw["rays_mask"]
    diff = [k for k, same in (
        ("samples.packed_info", np.array_equal(_n(sm.packed_info), r_sm["packed_info"])),
        ("intervals.packed_info", np.array_equal(_n(iv.packed_info), r_iv["packed_info"])),
        ("intervals.vals", np.array_equal(_n(iv.vals), r_iv["vals"])), ("samples.vals", np.array_equal(_n(sm.vals), r_sm["vals"])),
        ("ray_indices", np.array_equal(_n(sm.ray_indices), r_sm["ray_indices"])),
        ("is_left", np.array_equal(_n(iv.is_left), r_iv["is_left"])), ("is_right", np.array_equal(_n(iv.is_right), r_iv["is_right"])),
        ("is_valid", np.array_equal(_n(sm.is_valid), r_sm["is_valid"])),
        ("terminate_planes", np.array_equal(_n(term)[live], r_term[live]))) if not same]
    return ([f"{c['desc']}: {diff} differ"] if diff else []), int(r_sm["packed_info"][:, 1].sum())


SPLIT_P_FORMS = ("", "1", "2", "4", "8", "16")
# where the grid image is read from (single level, cone_angle = 0): LDS / L2 with 16 lanes per ray, L2 with 8, and the
# lane-per-ray kernel from LDS / L2
IMAGE_FORMS = ("NFA_SPLIT_L2=0,NFA_SPLIT_P=16", "NFA_SPLIT_L2=1,NFA_SPLIT_P=16", "NFA_SPLIT_P=8", "NFA_COUNT_L2=0,NFA_SPLIT_P=1", "NFA_COUNT_L2=1,NFA_SPLIT_P=1",
               "NFA_SPLIT_CAP=24,NFA_SPLIT_L2=1,NFA_SPLIT_P=16", "NFA_SPLIT_CAP=24,NFA_SPLIT_P=8", "NFA_SPLIT_CAP=16,NFA_SPLIT_L2=1,NFA_SPLIT_P=16",      # (+ list capacities: grids read from L2)
               # round 5: the lane-per-ray walk voxel by voxel / with empty-space macro steps
               "NFA_SKIP=0,NFA_COUNT_L2=1,NFA_SPLIT_P=1", "NFA_SKIP=1,NFA_COUNT_L2=0,NFA_SPLIT_P=1", "NFA_SKIP=1,NFA_COUNT_L2=1,NFA_SPLIT_P=1",
               # round 6: the single-launch sampling call (count + look-back + emit in the count kernel) switched off / on
               "NFA_FUSED_SAMPLE=0", "NFA_FUSED_SAMPLE=2,NFA_FUSED_VIS=1")
FUSED_FORMS = ("2", "2", "1", "0")     # NFA_FUSED_SAMPLE: the sampling call as one launch whatever the rays' length (twice: the second call has a guess of the output size) / automatic / as three
EMIT_FORMS = ("rays", "samples", "tiles")     # NFA_EMIT: 16 lanes per ray walking its run records / a lane per sample with searches
SEGMENT_FORMS = ("1", "0")
SEG_P_FORMS = ("8", "32")        # NFA_SEG_P: one lane per level segment / four (parts), cone_angle = 0, up to 4 levels
CONE_FORMS = ("1", "0")          # NFA_CONE: lane-per-segment walk + serial chain (cone_walk.hpp) / the general lane-per-ray kernel
CONE_P_FORMS = ("8", "16", "32", "64")      # NFA_CONE_P: lanes per ray of the two-phase kernel (8 only applies up to 4 levels)
