"""The examples/ scripts run end to end on the GPU (small sizes) and optimise what they claim to."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples"))

pytestmark = pytest.mark.gpu


def test_render_cube_example(device, tmp_path):
    import render_cube
    for specular in (False, True):
        frame = render_cube.render_cube(160, 120, specular=specular, device=str(device)).cpu().numpy()
        assert frame.shape == (120, 160, 4) and frame.dtype == np.uint8
        covered = frame[..., 3] == 255
        assert 0.05 < covered.mean() < 0.6 and frame[covered][:, :3].max() > 100
        assert (frame[~covered] == 0).all()


def test_optimize_rotation_example(device, tmp_path):
    import optimize_rotation
    losses, angles, target = optimize_rotation.optimize(steps=35, width=320, height=240, device=str(device),
                                                        out=str(tmp_path))
    assert losses[-1] < 0.25 * losses[0]
    assert any(name.endswith(".png") for name in os.listdir(tmp_path))


def test_soft_silhouette_example(device, tmp_path):
    import soft_silhouette
    losses, scale = soft_silhouette.optimize(steps=30, size=64, device=str(device), out=str(tmp_path))
    assert losses[-1] < 0.5 * losses[0]
    assert abs(float(scale[1]) - 0.6) < abs(1.0 - 0.6)     # moved towards the target's y scale
