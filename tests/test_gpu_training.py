"""Convergence check of the whole path (sampling -> rendering -> backward -> Adam) on the
procedural scene: a fog-initialised field must learn the scene through our kernels' gradients."""
import os
import sys

import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples"))
pytestmark = pytest.mark.gpu


def test_training_converges_on_procedural_scene():
    from train_occgrid_procedural import train

    hist = train(steps=400, res=64, log=lambda *_: None)
    assert hist[-1] > hist[0] + 6.0, hist          # at least +6 dB over the fog initialisation
    assert hist[-1] > 20.0, hist
