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
