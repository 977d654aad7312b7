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


TRIANGLE_CASES = ["w_111", "w_perspective", "one_w_negative", "all_w_negative", "collinear",
                  "reversed_winding", "coincident_tie", "beyond_far_plane", "two_overlapping"]


@pytest.fixture(scope="session")
def device():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")
