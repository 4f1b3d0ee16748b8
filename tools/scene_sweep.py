"""configs[4] stand-in at the kernel level: the sampling traversal on eight procedural scenes (tools/scenes.py) at 256^3,
every data-dependent plan switch of the library checked against its forced alternatives.

For every scene and two ray counts — the training size (rays such that ~2^18 samples come out, the reference's adaptive
batch, train_ngp_nerf_occ.py:187-194) and the reference's 8192-ray eval chunk (examples/utils.py:80-88) — the fused sampling
call runs under the automatic plan and under every forced form of the count pass and of the emit pass; HIP events around the
count and emit C-ABI calls (nerfacc_amd.cuda._backend.KernelTimer), outputs compared bit for bit between forms.

    python tools/scene_sweep.py [out.md] [--res=256] [--quick]

Prints one row per (scene, ray count): samples / ray, runs / ray, the automatic plan's count and emit times, the best forced
form of each and the ratio auto / best (what tests/test_gpu_scenes.py asserts on).
"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import nerfacc_amd  # noqa: E402
import scenes  # noqa: E402
from nerfacc_amd import cuda as C  # noqa: E402
from nerfacc_amd.cuda import _backend  # noqa: E402

COUNT_FORMS = {"P16 lds": dict(split_p=16, split_l2=0), "P16 c16": dict(split_p=16, split_l2=1, split_cap=16), "P16 c32": dict(split_p=16, split_l2=1, split_cap=32),
               "P16 c24": dict(split_p=16, split_l2=1, split_cap=24),
               "P8 c16": dict(split_p=8, split_cap=16), "P8 c24": dict(split_p=8, split_cap=24), "P8 c32": dict(split_p=8, split_cap=32), "P4": dict(split_p=4),
               "P1 lds": dict(split_p=1, count_l2=0), "P1 l2": dict(split_p=1, count_l2=1)}      # (c16 / c32: boundary-list capacity, grids read from L2 only)
EMIT_FORMS = {"tiles": dict(emit="tiles"), "rays": dict(emit="rays"), "samples": dict(emit="samples")}
STEP = 5e-3


def time_call(fn, reps):
    for _ in range(3):
        out = fn()
    torch.cuda.synchronize()
    timer = _backend.KernelTimer(names=("traverse_count", "traverse_fill"))
    _backend.set_kernel_timer(timer)
    for _ in range(reps):
        fn()
    s = timer.summary()
    _backend.set_kernel_timer(None)
    return out, s["traverse_count"][1] * 1e3, s["traverse_fill"][1] * 1e3


def sweep_scene(name, res, ray_counts, reps=10, dev="cuda:0", forms=True):
    occ = torch.from_numpy(scenes.occupancy_grid(name, res)).to(dev)
    aabb = torch.from_numpy(scenes.AABB[None].copy()).to(dev)
    rows = []
    for n in ray_counts:
        o, d = (torch.from_numpy(x).to(dev) for x in scenes.rays(n, seed=11))
        near, far = torch.zeros(n, device=dev), torch.full((n,), 1e10, device=dev)
        call = lambda: C.sample_occgrid(o, d, occ, aabb, near, far, STEP, 0.0)
        nerfacc_amd.reset_options()
        ref, c_auto, e_auto = time_call(call, reps)
        pk = ref[3]
        n_samples = int(ref[0].shape[0])
        row = dict(scene=name, res=res, rays=n, occupied=float(occ.float().mean()), samples=n_samples, samples_per_ray=n_samples / n,
                   rays_with_samples=float((pk[:, 1] > 0).float().mean()), count_us={"auto": c_auto}, emit_us={"auto": e_auto})
        if forms:
            for tag, f in COUNT_FORMS.items():
                with nerfacc_amd.options(**f):
                    out, c, _ = time_call(call, reps)
                assert all(torch.equal(a, b) for a, b in zip(ref, out)), (name, n, tag)
                row["count_us"][tag] = c
            for tag, f in EMIT_FORMS.items():
                with nerfacc_amd.options(**f):
                    out, _, e = time_call(call, reps)
                assert all(torch.equal(a, b) for a, b in zip(ref, out)), (name, n, tag)
                row["emit_us"][tag] = e
        rows.append(row)
    return rows


def training_rays(name, res, dev="cuda:0", target=1 << 18):
    """the ray count at which ~2^18 samples come out of this scene (the reference's adaptive batch), from a 4096-ray probe"""
    occ = torch.from_numpy(scenes.occupancy_grid(name, res)).to(dev)
    aabb = torch.from_numpy(scenes.AABB[None].copy()).to(dev)
    o, d = (torch.from_numpy(x).to(dev) for x in scenes.rays(4096, seed=5))
    n = C.sample_occgrid(o, d, occ, aabb, torch.zeros(4096, device=dev), torch.full((4096,), 1e10, device=dev), STEP, 0.0)[0].shape[0]
    return int(min(max(target / max(n / 4096, 1e-3), 1024), 262144))


def main():
    out_md = next((a for a in sys.argv[1:] if not a.startswith("--")), None)
    res = int(next((a.split("=")[1] for a in sys.argv if a.startswith("--res=")), 256))
    quick = "--quick" in sys.argv
    all_rows = []
    for name in scenes.SCENES:
        counts = [training_rays(name, res), 8192]
        all_rows += sweep_scene(name, res, counts, reps=5 if quick else 20)
        for r in all_rows[-2:]:
            bc = min((v, k) for k, v in r["count_us"].items() if k != "auto")
            be = min((v, k) for k, v in r["emit_us"].items() if k != "auto")
            print(f"{r['scene']:10s} {r['rays']:7d} rays  {r['samples_per_ray']:6.1f} samples/ray  count auto {r['count_us']['auto']:7.1f} us, best {bc[1]:8s} {bc[0]:7.1f}"
                  f" ({r['count_us']['auto'] / bc[0]:.2f}x)   emit auto {r['emit_us']['auto']:7.1f} us, best {be[1]:8s} {be[0]:7.1f} ({r['emit_us']['auto'] / be[0]:.2f}x)", flush=True)
    if out_md:
        with open(out_md, "w") as f:
            f.write(f"| scene ({res}^3) | occupied | rays | samples / ray | rays with samples | count: auto | " + " | ".join(COUNT_FORMS) + " | auto / best | emit: auto | "
                    + " | ".join(EMIT_FORMS) + " | auto / best |\n|" + "---|" * (9 + len(COUNT_FORMS) + len(EMIT_FORMS)) + "\n")
            for r in all_rows:
                bc = min(v for k, v in r["count_us"].items() if k != "auto")
                be = min(v for k, v in r["emit_us"].items() if k != "auto")
                f.write(f"| {r['scene']} | {r['occupied']:.4f} | {r['rays']} | {r['samples_per_ray']:.1f} | {r['rays_with_samples']:.2f} | {r['count_us']['auto']:.1f} | "
                        + " | ".join(f"{r['count_us'][k]:.1f}" for k in COUNT_FORMS) + f" | {r['count_us']['auto'] / bc:.2f} | {r['emit_us']['auto']:.1f} | "
                        + " | ".join(f"{r['emit_us'][k]:.1f}" for k in EMIT_FORMS) + f" | {r['emit_us']['auto'] / be:.2f} |\n")
        with open(os.path.splitext(out_md)[0] + ".json", "w") as f:
            json.dump(all_rows, f, indent=1)


if __name__ == "__main__":
    main()
