"""Drop-in for the reference's pybind11 module `rasterize_triangles_cpp`.

The reference's one native module (src/mesh_renderer/kernels/rasterize_triangles.cpp:421-424,
built by kernels/setup.py) exports

    forward(vertices[V,4] f32, triangles[T,3] i32, image_width, image_height)
        -> [px_triangle_ids[H,W] i32, px_barycentric_coords[H,W,3] f32, z_buffer[H,W] f32]
    backward(df_dbarycentric_coords[H,W,3], vertices, triangles, px_triangle_ids,
             px_barycentric_coords) -> [df_dvertices[V,4] f32]

and is imported by name in src/mesh_renderer/rasterize_triangles_ext.py:3.  Put this directory on
sys.path (or alias the module: sys.modules["rasterize_triangles_cpp"] = this module) and that file
runs unchanged against libmesh_raster_hip.so -- with tensors on an MI355X instead of the host: one
image per call, lists returned like the pybind module returns std::vector<Tensor>, inputs borrowed and
never modified, outputs freshly allocated.  Errors: wrong dtype -> RuntimeError (the reference's
accessor<> throws c10::Error), host tensors -> RuntimeError (no CPU fallback).
"""
from pytorch_mesh_renderer_amd import _native


def forward(vertices, triangles, image_width, image_height):
    if vertices.dim() != 2:
        raise RuntimeError("vertices must have shape [vertex_count, 4]")
    ids, bary, z = _native.rasterize_forward(vertices.detach().unsqueeze(0), triangles,
                                             int(image_width), int(image_height))
    return [ids[0], bary[0], z[0]]


def backward(df_dbarycentric_coords, vertices, triangles, px_triangle_ids, px_barycentric_coords):
    if vertices.dim() != 2:
        raise RuntimeError("vertices must have shape [vertex_count, 4]")
    dclip = _native.rasterize_backward(df_dbarycentric_coords.detach().unsqueeze(0).contiguous(),
                                       vertices.detach().unsqueeze(0), triangles,
                                       px_triangle_ids.unsqueeze(0), px_barycentric_coords.detach().unsqueeze(0))
    return [dclip[0]]
