"""Procedural test meshes in the reference's vertex / triangle layout.

Counterpart of src/common/shapes.py:4-118 (sphere, cube).  Used as the workload
generator for the benchmark configs (sphere K=50 -> V=2502, T=5000; K=158 ->
V=24966, T=49928).  Written vectorised; the index layout
-- including the reference's wrap-around at the longitude seam and at the bottom
pole fan (src/common/shapes.py:57-79), which produces two pole-to-pole sliver
triangles -- is reproduced because the benchmark workload is defined on it.
tests/golden/shapes_hashes.json pins both meshes against the reference's output.
"""
import numpy as np
import torch


def sphere_arrays(radius, resolution=25):
    """numpy version: (vertices [K*K+2,3] f32, triangles [2*K*K,3] i32, normals)."""
    K = int(resolution)
    theta = np.linspace(np.pi / (K + 1), np.pi - np.pi / (K + 1), K, endpoint=True)
    phi = np.linspace(0.0, 2.0 * np.pi, K, endpoint=False)
    st, ct = np.sin(theta)[:, None], np.cos(theta)[:, None]
    ring = np.stack([st * np.sin(phi)[None, :],
                     np.broadcast_to(ct, (K, K)),
                     st * np.cos(phi)[None, :]], axis=-1)          # float64 [K,K,3]
    # the reference scales in float64 and rounds once on assignment
    body = (float(radius) * ring).reshape(K * K, 3).astype(np.float32)
    poles = np.array([[0.0, 1.0, 0.0], [0.0, -1.0, 0.0]], np.float32)  # NOT scaled by radius
    vertices = np.concatenate([body, poles], 0)

    n_vert = K * K + 2
    i = np.arange(K - 1)[:, None]
    j = np.arange(K)[None, :]
    tl = i * K + j
    tr = tl + 1            # wraps into the next latitude ring at j == K-1 (reference quirk)
    bl = tl + K
    br = bl + 1
    upper = np.stack([tl, bl, tr], -1)
    lower = np.stack([tr, bl, br], -1)
    quads = np.stack([upper, lower], 2).reshape(-1, 3)              # interleaved per quad
    k = np.arange(K)
    top_fan = np.stack([np.full(K, n_vert - 2), k, k + 1], -1)
    base = (K - 1) * K
    bottom_fan = np.stack([np.full(K, n_vert - 1), base + k + 1, base + k], -1)
    triangles = np.concatenate([quads, top_fan, bottom_fan], 0).astype(np.int32)

    norm = np.sqrt((vertices.astype(np.float32) ** 2).sum(-1, keepdims=True))
    normals = vertices / np.maximum(norm, 1e-12)
    return vertices, triangles, normals.astype(np.float32)


def sphere(radius, resolution=25):
    """(vertices, triangles, normals) as torch tensors, CCW seen from outside."""
    v, t, _ = sphere_arrays(radius, resolution)
    vertices = torch.from_numpy(v)
    normals = torch.nn.functional.normalize(vertices, p=2.0, dim=-1)
    return vertices, torch.from_numpy(t), normals


_CUBE_CORNERS = [[-1, -1, 1], [-1, -1, -1], [-1, 1, -1], [-1, 1, 1],
                 [1, -1, 1], [1, -1, -1], [1, 1, -1], [1, 1, 1]]
_CUBE_FACES_CCW = [[2, 1, 0], [0, 3, 2], [6, 2, 3], [3, 7, 6], [5, 6, 7], [7, 4, 5],
                   [1, 5, 4], [4, 0, 1], [2, 6, 5], [5, 1, 2], [0, 4, 7], [7, 3, 0]]


def cube(size):
    """Axis-aligned cube of side `size` centred on the origin (CCW from outside)."""
    vertices = 0.5 * size * torch.tensor(_CUBE_CORNERS, dtype=torch.float32)
    normals = torch.nn.functional.normalize(vertices, p=2.0, dim=-1)
    triangles = torch.tensor(_CUBE_FACES_CCW, dtype=torch.int32)
    return vertices, triangles, normals
