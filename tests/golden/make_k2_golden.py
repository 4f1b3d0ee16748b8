"""Generate tests/golden/k2_reference.npz + k2_sensitivity.json: the exact sample lists of the REFERENCE's
own traversal (nerfacc/cuda/csrc/grid.cu, compiled for the host by oracle/ref_shim/Makefile into
oracle/_ref/) driven through the REFERENCE's own Python layer (nerfacc/grid.py:93-192 with
`nerfacc.cuda._backend._C` pointed at oracle/_ref's module).  Nothing of this repo's oracle or kernels is
involved in producing the fixture; they are only compared with it afterwards for the sensitivity report.

Run ONLY in the build container (needs /root/reference):

    make -C oracle/ref_shim && python tests/golden/make_k2_golden.py

Two host builds of the reference exist (no FMA contraction / contraction wherever the compiler can — the
analogue of nvcc's default --fmad=true).  The fixture holds the outputs of the contracting build
always; k2_sensitivity.json records, per case, how many rays change when contraction is switched off and how
many rays of this repo's oracle differ from either build (the FMA-model sensitivity).
"""
import importlib
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"


def main():
    sys.path[:0] = [REF, os.path.join(ROOT, "oracle", "_ref"), os.path.join(ROOT, "tests")]
    import nerfacc  # the reference
    import nerfacc.cuda._backend as backend
    import nerfacc.grid as G

    assert nerfacc.__file__.startswith(REF), nerfacc.__file__
    builds = {k: importlib.import_module("nerfacc_ref_" + k) for k in ("off", "fma")}
    gpu_rule_build = importlib.import_module("nerfacc_ref_gpu")      # fma + the GPU's float -> int conversion rule (prelude.h)
    only_inplane = "--only-inplane" in sys.argv
    import k2_cases as K

    sys.path.append(ROOT)
    import oracle

    def run_reference(build, c):
        backend._C = builds[build] if isinstance(build, str) else build
        tt = lambda a: torch.from_numpy(np.ascontiguousarray(a))
        extra = {k: tt(v) for k, v in c["extra"].items()}
        iv, sm, term = G.traverse_grids(tt(c["rays_o"]), tt(c["rays_d"]), tt(c["binaries"]), tt(c["aabbs"]), **extra, **c["kw"])
        as_map = lambda s: {k: getattr(s, k).numpy() for k in ("vals", "ray_indices", "packed_info") + (("is_left", "is_right") if hasattr(s, "is_left") else ("is_valid",))}
        ivm, smm = as_map(iv), as_map(sm)
        for m in (ivm, smm):
            m["chunk_starts"], m["chunk_cnts"] = m["packed_info"][:, 0], m["packed_info"][:, 1]
        live = c["extra"].get("rays_mask") if c["kw"].get("over_allocate") else None
        return K.pack_outputs(ivm, smm, term.numpy(), live)

    def run_oracle(c):
        iv, sm, term = oracle.traverse_grids(c["rays_o"], c["rays_d"], c["binaries"], c["aabbs"], **c["extra"], **c["kw"])
        live = c["extra"].get("rays_mask") if c["kw"].get("over_allocate") else None
        return K.pack_outputs(iv, sm, term, live)

    def differing_rays(a, b):
        """rays whose sample list differs in any way (count, or any edge / midpoint value)"""
        R = a["sm_chunk_cnts"].shape[0]
        bad = (a["sm_chunk_cnts"] != b["sm_chunk_cnts"]) | (a["iv_chunk_cnts"] != b["iv_chunk_cnts"]) | (a["term_live"] != b["term_live"])
        same = ~bad
        for key_v, key_s, key_c in (("sm_vals", "sm_chunk_starts", "sm_chunk_cnts"), ("iv_vals", "iv_chunk_starts", "iv_chunk_cnts")):
            if a[key_v].shape == b[key_v].shape and np.array_equal(a[key_s], b[key_s]):
                neq = a[key_v] != b[key_v]
                if neq.any():
                    # attribute differing elements to rays through the (equal) chunk layout
                    starts = a[key_s]
                    ray_of = np.searchsorted(starts, np.flatnonzero(neq), side="right") - 1
                    bad[np.unique(ray_of)] = True
        return int(bad.sum()), R

    # ---- rays lying IN bounding planes of a level: pinned by the reference built with the GPU's conversion rule -------------
    fx, rep = {}, {}
    for name in K.GPU_RULE:
        c = K.build_case(name)
        # (the x86-rule builds cannot run these rays: INT_MIN indices make the reference's walk read outside the grid — segfault)
        out, orc = run_reference(gpu_rule_build, c), run_oracle(c)
        d_orc, R = differing_rays(out, orc)
        rep[name] = dict(rays=R, samples=int(out["sm_chunk_cnts"].sum()), rays_differing_oracle_vs_gpu_rule=d_orc, fixture_build="gpu")
        fx[f"{name}/input_sha"] = np.array(K.input_digest(c))
        for k in K.OUTPUT_KEYS:
            fx[f"{name}/sha/{k}"] = np.array(K.sha(out[k]))
        fx[f"{name}/cnts/sm_chunk_cnts"] = out["sm_chunk_cnts"].astype(np.int32)
        fx[f"{name}/cnts/iv_chunk_cnts"] = out["iv_chunk_cnts"].astype(np.int32)
        print(f"{name:20s} rays {R:6d} samples {rep[name]['samples']:8d}  oracle != gpu-rule build {d_orc:4d}")
    np.savez_compressed(os.path.join(HERE, "k2_inplane.npz"), **fx)
    with open(os.path.join(HERE, "k2_inplane.json"), "w") as f:
        json.dump(rep, f, indent=1)
    if only_inplane:
        return

    fixture, report = {}, {}
    # ---- the reference's own test configuration (tests/test_grid.py:38-68), torch CPU generator ---------------
    torch.manual_seed(42)
    rays_o = torch.randn((10, 3))
    rays_d = torch.randn((10, 3))
    rays_d = rays_d / rays_d.norm(dim=-1, keepdim=True)
    base = torch.tensor([-1.0, -1.0, -1.0, 1.0, 1.0, 1.0])
    aabbs = torch.stack([G._enlarge_aabb(base, 2**i) for i in range(4)])
    binaries = torch.rand((4, 32, 32, 32)) > 0.5
    fixture["ref_test_grid/rays_o"] = rays_o.numpy()
    fixture["ref_test_grid/rays_d"] = rays_d.numpy()
    fixture["ref_test_grid/aabbs"] = aabbs.numpy()
    fixture["ref_test_grid/binaries_bits"] = np.packbits(binaries.numpy().ravel())

    for name in K.ALL:
        c = K.build_case(name, fixture)
        outs = {b: run_reference(b, c) for b in builds}
        orc = run_oracle(c)
        d_builds, R = differing_rays(outs["off"], outs["fma"])
        d_orc_off, _ = differing_rays(outs["off"], orc)
        d_orc_fma, _ = differing_rays(outs["fma"], orc)
        keep = "fma"      # the analogue of the CUDA build's default --fmad=true; "off" only feeds the sensitivity report
        out = outs[keep]
        n_samples = int(out["sm_chunk_cnts"].sum())
        report[name] = dict(rays=R, samples=n_samples, intervals=int(out["iv_chunk_cnts"].sum()),
                            rays_differing_off_vs_fma=d_builds, rays_differing_oracle_vs_off=d_orc_off,
                            rays_differing_oracle_vs_fma=d_orc_fma, fixture_build=keep,
                            kw={k: (float(v) if isinstance(v, float) else v) for k, v in c["kw"].items()})
        fixture[f"{name}/input_sha"] = np.array(K.input_digest(c))
        full = out["sm_vals"].shape[0] <= K.FULL_LIMIT
        for k in K.OUTPUT_KEYS:
            fixture[f"{name}/sha/{k}"] = np.array(K.sha(out[k]))
            if full:
                fixture[f"{name}/full/{k}"] = out[k]
        if not full:
            fixture[f"{name}/cnts/sm_chunk_cnts"] = out["sm_chunk_cnts"].astype(np.int32)
            fixture[f"{name}/cnts/iv_chunk_cnts"] = out["iv_chunk_cnts"].astype(np.int32)
        if not c["kw"].get("over_allocate"):
            # what OccGridEstimator.sampling returns from these outputs (the reference's occ_grid.py:174-175):
            # t_starts = vals[is_left], t_ends = vals[is_right] — digests, for the fused sampling path's test (round 4)
            fixture[f"{name}/sha/t_starts"] = np.array(K.sha(out["iv_vals"][out["iv_is_left"]]))
            fixture[f"{name}/sha/t_ends"] = np.array(K.sha(out["iv_vals"][out["iv_is_right"]]))
        print(f"{name:20s} rays {R:6d} samples {n_samples:8d}  off!=fma {d_builds:4d}  oracle!=off {d_orc_off:4d}  oracle!=fma {d_orc_fma:4d}  ({'full' if full else 'digests'})")

    # ---- K1 through the reference's own kernel as well (tests/test_grid.py:7-35 generator: seed 42, 1000 rays x 100 boxes) ----
    torch.manual_seed(42)
    k1_o = torch.rand((1000, 3))
    k1_d = torch.randn((1000, 3))
    k1_d = k1_d / k1_d.norm(dim=-1, keepdim=True)
    k1_min = torch.rand((100, 3))
    k1_boxes = torch.cat([k1_min, k1_min + torch.rand_like(k1_min)], dim=-1)
    k1 = {}
    for b in builds:
        backend._C = builds[b]
        k1[b] = [x.numpy() for x in G.ray_aabb_intersect(k1_o, k1_d, k1_boxes)]            # python defaults: near -inf, far inf, miss inf
        k1[b + "_nf"] = [x.numpy() for x in G.ray_aabb_intersect(k1_o, k1_d, k1_boxes, 0.1, 0.7, -1.0)]
    o_tmin, o_tmax, o_hits = oracle.ray_aabb_intersect(k1_o.numpy(), k1_d.numpy(), k1_boxes.numpy())
    report["k1_ray_aabb"] = dict(pairs=100000, hits=int(k1["fma"][2].sum()),
                                 values_differing_off_vs_fma=int((k1["off"][0] != k1["fma"][0]).sum() + (k1["off"][1] != k1["fma"][1]).sum()),
                                 values_differing_oracle_vs_fma=int((o_tmin != k1["fma"][0]).sum() + (o_tmax != k1["fma"][1]).sum()),
                                 values_differing_oracle_vs_off=int((o_tmin != k1["off"][0]).sum() + (o_tmax != k1["off"][1]).sum()),
                                 hits_differing_oracle=int((o_hits != k1["off"][2]).sum()))
    keep_k1 = "off" if report["k1_ray_aabb"]["values_differing_oracle_vs_off"] == 0 else "fma"
    report["k1_ray_aabb"]["fixture_build"] = keep_k1
    fixture["k1/rays_o"], fixture["k1/rays_d"], fixture["k1/aabbs"] = k1_o.numpy(), k1_d.numpy(), k1_boxes.numpy()
    for tag, key in (("", keep_k1), ("_nf", keep_k1 + "_nf")):
        fixture[f"k1/sha/t_mins{tag}"] = np.array(K.sha(k1[key][0]))
        fixture[f"k1/sha/t_maxs{tag}"] = np.array(K.sha(k1[key][1]))
        fixture[f"k1/hits_bits{tag}"] = np.packbits(k1[key][2].ravel())
    print("k1_ray_aabb", report["k1_ray_aabb"])

    np.savez_compressed(os.path.join(HERE, "k2_reference.npz"), **fixture)
    with open(os.path.join(HERE, "k2_sensitivity.json"), "w") as f:
        json.dump(report, f, indent=1)
    print("wrote", os.path.getsize(os.path.join(HERE, "k2_reference.npz")), "bytes")


if __name__ == "__main__":
    main()
