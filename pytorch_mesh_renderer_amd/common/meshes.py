"""Mesh utilities on the device of their inputs.

Counterpart of src/common/meshes.py:3-35 (compute_vertex_normals): area-weighted face
normals accumulated on the incident vertices, then normalised (eps 1e-6).  On a HIP device
this is a per-vertex gather kernel over the mesh's cached CSR adjacency (csrc/mesh_ops.hip:
no atomics, forward and hand-derived backward); host tensors -- the CPU test-suite, mesh
preparation before upload -- take the equivalent batched torch expression.  Differentiable
either way.
"""
import torch


class _VertexNormals(torch.autograd.Function):
    @staticmethod
    def forward(ctx, vertices, triangles):
        from .. import _native
        v = vertices.detach().contiguous()
        offsets, entries = _native.vertex_adjacency(triangles, v.shape[1])
        normals, sums = _native.vertex_normals_forward(v, triangles, adjacency=(offsets, entries))
        # the adjacency travels with the node: the backward does not rebuild it (an argsort) when the
        # cache on the caller's tensor object is gone
        ctx.save_for_backward(v, sums, triangles, offsets, entries)
        return normals

    @staticmethod
    def backward(ctx, dnormals):
        from .. import _native
        v, sums, triangles, offsets, entries = ctx.saved_tensors
        return _native.vertex_normals_backward(dnormals.contiguous(), v, sums, triangles,
                                               adjacency=(offsets, entries)), None


def compute_vertex_normals(vertices, triangles):
    """vertices [B,V,3], triangles [T,3] (any integer dtype, as the reference's triangles.long()
    accepts: src/common/meshes.py:18) -> unit vertex normals [B,V,3]."""
    if vertices.is_cuda and vertices.dtype == torch.float32:
        if triangles.dtype in (torch.float16, torch.float32, torch.float64, torch.bfloat16, torch.bool):
            raise RuntimeError("triangles must hold integer vertex indices")
        # the kernels index with int32 (a vertex count beyond 2^31 does not fit a GPU anyway); the
        # conversion keeps the caller's tensor object -- and the adjacency cached on it -- when it
        # already is int32 on the right device
        tri = triangles if triangles.dtype == torch.int32 and triangles.device == vertices.device else \
            triangles.to(device=vertices.device, dtype=torch.int32)
        return _VertexNormals.apply(vertices, tri)
    tri = triangles.long()
    corner_index = tri.t().reshape(-1)                                   # [3T]: all first corners, then ...
    corners = vertices.index_select(1, corner_index).reshape(vertices.shape[0], 3, -1, 3)
    v0, v1, v2 = corners[:, 0], corners[:, 1], corners[:, 2]             # [B,T,3]
    # The reference evaluates (b - a) x (c - a) once per corner (meshes.py:24-33); the three are the
    # same area-weighted face normal up to rounding, so the host expression computes it once.
    face = torch.cross(v1 - v0, v2 - v0, dim=-1)
    normals = torch.zeros_like(vertices).index_add(1, corner_index, face.repeat(1, 3, 1))
    return torch.nn.functional.normalize(normals, eps=1e-6, p=2, dim=-1)
