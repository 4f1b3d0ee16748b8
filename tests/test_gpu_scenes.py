"""The data-dependent plan switches of the sampling call — lanes per ray of the count pass, where the grid image lives, the
emit form — checked on eight procedural scenes at 256^3 (tools/scenes.py: the configs[4] stand-ins, differing the way
nerf_synthetic's eight objects do: thin structures, a dense slab, a hollow shell, a near-empty grid, the reference's
rand > 0.5 noise, ...) and two ray counts each: the training size (~2^18 samples per call) and the reference's 8192-ray
eval chunk.  Every forced form must return the SAME tensors as the automatic plan (always asserted); the automatic plan's
count and emit times must stay near the best forced form's (`perf`-marked: wall-clock ratios, re-measured before failing).
VERDICT r3 item 2; the numbers of one run are profiles/r04_scene_sweep.md (tools/scene_sweep.py)."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

SCENE_NAMES = ["lego", "ficus", "ship", "shell", "speck", "noise", "drums", "materials"]
# auto <= RATIO * best + SLACK_US.  Measured at the end of round 4 (profiles/r04_scene_sweep.md): count pass within 1.09 of the best
# forced form on 15 of 16 rows and 1.26 on one (materials at 9.9 k rays: 8 lanes per ray with 32-entry boundary lists against
# 16 lanes with 16-entry ones; since then 24-entry lists: within 1.09 on all 16 rows
# emit pass within 1.10; boxes differ by +-5 %, and HIP-event times of 10-40 us kernels carry ~1 us of jitter.
RATIO, SLACK_US = 1.4, 2.0


@pytest.mark.parametrize("name", SCENE_NAMES)
def test_forced_forms_return_the_automatic_plan_s_tensors(name):
    import scene_sweep as SW

    # (sweep_scene asserts torch.equal between the automatic plan and every forced count / emit form)
    rows = SW.sweep_scene(name, 128, [3000, 20000], reps=1)
    assert len(rows) == 2 and all(r["samples"] >= 0 for r in rows)


@pytest.mark.perf
@pytest.mark.parametrize("name", SCENE_NAMES)
def test_automatic_plan_is_near_the_best_forced_form(name):
    import scene_sweep as SW

    counts = [SW.training_rays(name, 256), 8192]
    worst = None
    for attempt in range(3):                      # wall-clock ratios: re-measure before failing
        rows = SW.sweep_scene(name, 256, counts, reps=10)
        worst = []
        for r in rows:
            for what in ("count_us", "emit_us"):
                best = min(v for k, v in r[what].items() if k != "auto")
                if r[what]["auto"] > RATIO * best + SLACK_US:
                    worst.append((name, r["rays"], what, {k: round(v, 1) for k, v in r[what].items()}))
        if not worst:
            return
    assert not worst, worst


def _noise_call(n_rays):
    import torch

    import scenes
    from nerfacc_amd import cuda as C

    occ = torch.from_numpy(scenes.occupancy_grid("noise", 256)).to("cuda:0")
    aabb = torch.from_numpy(scenes.AABB[None].copy()).to("cuda:0")
    o, d = (torch.from_numpy(x).to("cuda:0") for x in scenes.rays(n_rays, seed=11))
    near, far = torch.zeros(n_rays, device="cuda:0"), torch.full((n_rays,), 1e10, device="cuda:0")
    return lambda: C.sample_occgrid(o, d, occ, aabb, near, far, 5e-3, 0.0)


@pytest.mark.parametrize("n_rays", [16384, 65536])
def test_noise_grid_emit_rays_per_wave_chosen_on_the_device(force_options, n_rays):
    """the reference's rand > 0.5 grid at 256^3: 30-60 runs per ray.  The tile emit form halves its rays per wave on the device
    from the call's runs (emit_rays_per_wave_log2); whatever it picks, the tensors are those of every forced
    rays-per-wave and of the lane-per-sample form"""
    import torch

    call = _noise_call(n_rays)
    ref = call()
    assert ref[0].shape[0] > 100 * n_rays                      # (the long, many-run rays the rule is about)
    for form in (dict(emit="tiles", emit_rb=0), dict(emit="tiles", emit_rb=2), dict(emit="tiles", emit_rb=4), dict(emit="tiles", emit_rb=6),
                 dict(emit="samples"), dict(emit="rays")):
        force_options(emit=None, emit_rb=None)
        force_options(**form)
        out = call()
        assert all(torch.equal(a, b) for a, b in zip(ref, out)), form


@pytest.mark.perf
@pytest.mark.parametrize("n_rays", [16384, 65536])
def test_noise_grid_emit_auto_within_1_15_of_the_best_rays_per_wave(force_options, n_rays):
    """item 4a's bar: the device-side choice within 1.15 of the best forced `emit_rb` (wall-clock ratio: re-measured before failing)"""
    import scene_sweep as SW

    call = _noise_call(n_rays)
    for attempt in range(3):
        force_options(emit=None, emit_rb=None)
        _, _, auto = SW.time_call(call, 10)
        best = None
        for rb in range(0, 7):
            force_options(emit_rb=rb)
            _, _, e = SW.time_call(call, 10)
            best = e if best is None or e < best else best
        if auto <= 1.15 * best + SLACK_US:
            return
    assert auto <= 1.15 * best + SLACK_US, (auto, best)
