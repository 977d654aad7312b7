"""Mesh utilities on the device of their inputs.

Counterpart of src/common/meshes.py:3-35 (compute_vertex_normals): area-weighted face
normals accumulated on the incident vertices with index_add, then normalised (eps 1e-6).
Batched instead of the reference's Python loop over the batch; differentiable.
"""
import torch


def compute_vertex_normals(vertices, triangles):
    """vertices [B,V,3], triangles [T,3] -> unit vertex normals [B,V,3]."""
    tri = triangles.long()
    faces = vertices[:, tri, :]                                    # [B,T,3,3]
    normals = torch.zeros_like(vertices)
    for corner in range(3):
        a, b, c = faces[:, :, corner], faces[:, :, (corner + 1) % 3], faces[:, :, (corner + 2) % 3]
        normals = normals.index_add(1, tri[:, corner], torch.cross(b - a, c - a, dim=-1))
    return torch.nn.functional.normalize(normals, eps=1e-6, p=2, dim=-1)
