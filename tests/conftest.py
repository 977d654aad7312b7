import hashlib
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden_npz(name):
    return np.load(os.path.join(GOLDEN, name))


def golden_json(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


def golden_sphere_job(clip_file):
    """The benchmark sphere with the STORED clip-space vertices the goldens were made from
    (clip coordinates computed with sin/cos/matmul are not bit-stable across host CPUs)."""
    import torch
    from pytorch_mesh_renderer_amd.common import shapes
    clip = torch.from_numpy(np.load(os.path.join(GOLDEN, clip_file)))
    _, triangles, _ = shapes.sphere(1.0, 50)
    return {"clip": clip, "triangles": triangles}


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def bits_equal(a, b):
    a, b = np.ascontiguousarray(a), np.ascontiguousarray(b)
    return a.shape == b.shape and a.dtype == b.dtype and a.tobytes() == b.tobytes()


def seeded_dbary(shape, seed=0):
    """Upstream gradient of SURVEY.md 8d: randn(seed) / (H*W)."""
    import torch
    g = torch.Generator().manual_seed(seed)
    h, w = shape[-3], shape[-2]
    return torch.randn(shape, generator=g) / (h * w)


def barycentric_png_scenes():
    """Two fixtures of the reference's test_data/ that none of its tests loads (they came along from the TensorFlow
    original): barycentrics as RGB at 640 x 480.  -> {png name: (clip [V,4] float32, triangles [T,3] int32)}.
      Barycentrics_Cube.png   the cube of rasterize_triangles_test.py:79-117 from its first camera, eye (2, 3, 6)
      Simple_Tetrahedron.png  the triangle of rasterize_triangles_test.py:72 plus an apex at the centre of the screen,
                              nearer than the base (found by search over apex depth and face orders: any apex depth
                              below 0.3 and exactly these vertex orders reproduce the file with 0 outliers)"""
    import torch
    from pytorch_mesh_renderer_amd.common import camera_utils
    cube = torch.tensor([[-1, -1, 1], [-1, -1, -1], [-1, 1, -1], [-1, 1, 1], [1, -1, 1], [1, -1, -1], [1, 1, -1], [1, 1, 1]],
                        dtype=torch.float32)
    cube_t = np.array([[0, 1, 2], [2, 3, 0], [3, 2, 6], [6, 7, 3], [7, 6, 5], [5, 4, 7], [4, 5, 1], [1, 0, 4], [5, 6, 2],
                       [2, 1, 5], [7, 4, 0], [0, 3, 7]], np.int32)
    persp = camera_utils.perspective(640 / 480, torch.tensor([40.0]), torch.tensor([0.01]), torch.tensor([10.0]))
    look = camera_utils.look_at(torch.tensor([[2.0, 3.0, 6.0]]), torch.zeros(1, 3), torch.tensor([[0.0, 1.0, 0.0]]))
    proj = torch.matmul(persp, look)[0]
    cube_clip = torch.matmul(torch.cat([cube, torch.ones(8, 1)], 1), proj.T).numpy().astype(np.float32)
    tetra = np.array([[-0.5, -0.5, 0.8, 1.0], [0.0, 0.5, 0.3, 1.0], [0.5, -0.5, 0.3, 1.0], [0.0, 0.0, 0.0, 1.0]], np.float32)
    tetra_t = np.array([[0, 1, 3], [1, 2, 3], [2, 0, 3], [0, 1, 2]], np.int32)   # (the base lies behind the three sides)
    return {"Barycentrics_Cube.png": (cube_clip, cube_t), "Simple_Tetrahedron.png": (tetra, tetra_t)}


def png_outlier_fraction(name, image, pixel_error_threshold=0.01):
    """The reference's soft image comparison (test_utils.py:105-160): fraction of pixels with a channel off by more."""
    from PIL import Image
    baseline = np.asarray(Image.open(os.path.join(GOLDEN, "ref_png", name))).astype(float) / 255.0
    assert baseline.shape == image.shape, (baseline.shape, image.shape)
    return float(np.any(np.abs(baseline - np.clip(image, 0.0, 1.0)) > pixel_error_threshold, axis=2).mean())


TRIANGLE_CASES = ["w_111", "w_perspective", "one_w_negative", "all_w_negative", "collinear",
                  "reversed_winding", "coincident_tie", "beyond_far_plane", "two_overlapping"]


@pytest.fixture(scope="session")
def device():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")
