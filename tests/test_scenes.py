"""tools/scenes.py (the eight procedural configs[4] stand-ins) on CPU: the numpy and the torch evaluation of every scene agree point
for point (bench.py builds teacher fields with torch, the sweep and the tests build occupancy grids with numpy), and the scenes keep
the statistics that make them what they claim to be — the library's count-pass plan reads the share of non-empty 4^3 bricks
(grid.hip: grid_is_noisy / grid_is_near_empty), so a scene drifting across 0.02 or 0.5 would silently change what the sweep covers."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import scenes  # noqa: E402

# share of non-empty bricks at 128^3: (low, high)
BRICK_FILL = {"lego": (0.03, 0.12), "ficus": (0.005, 0.02), "ship": (0.2, 0.4), "shell": (0.05, 0.2), "speck": (0.001, 0.01),
              "noise": (0.99, 1.0), "drums": (0.03, 0.15), "materials": (0.03, 0.15)}


@pytest.mark.parametrize("name", list(scenes.SCENES))
def test_scene_backends_agree_and_statistics_hold(name):
    g = torch.Generator().manual_seed(3)
    x = torch.rand(20000, 3, generator=g) * 3 - 1.5
    assert np.array_equal(scenes.SCENES[name](torch, x).numpy(), scenes.SCENES[name](np, x.numpy()))
    occ = scenes.occupancy_grid(name, 128)
    assert occ.shape == (1, 128, 128, 128) and occ.dtype == bool
    fill = occ[0].reshape(32, 4, 32, 4, 32, 4).any(axis=(1, 3, 5)).mean()
    lo, hi = BRICK_FILL[name]
    assert lo <= fill <= hi, (name, fill)
    # the three classes of the count-pass plan are all represented
    assert (fill >= 0.5) == (name == "noise") and (fill <= 0.02) == (name in ("ficus", "speck"))


def test_rays_are_unit_and_seeded():
    o, d = scenes.rays(1000, seed=4)
    o2, d2 = scenes.rays(1000, seed=4)
    assert np.array_equal(o, o2) and np.array_equal(d, d2)
    assert np.allclose(np.linalg.norm(d, axis=1), 1.0, atol=1e-6) and np.allclose(np.linalg.norm(o, axis=1), 4.0, atol=1e-5)
