"""MI355X-native differentiable mesh rasterizer.

Drop-in for the `mesh_renderer` package of andrewkchan/pytorch_mesh_renderer:

    from pytorch_mesh_renderer_amd import mesh_renderer          # render, rasterize, tone_mapper
    from pytorch_mesh_renderer_amd.common import camera_utils, shapes

All per-pixel work (coverage, z-buffer, barycentrics, attribute interpolation and
their gradients) runs in hand-written HIP kernels for gfx950 behind the C ABI of
include/mesh_raster.h; there is no CPU or eager fallback.
"""
from . import common, mesh_renderer, soft_mesh_renderer
from .mesh_renderer import render, tone_mapper, rasterize

__all__ = ["common", "mesh_renderer", "soft_mesh_renderer", "render", "tone_mapper", "rasterize"]
