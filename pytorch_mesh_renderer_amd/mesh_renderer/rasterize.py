"""Differentiable triangle rasterizer API (Genova 2018 un-clipped barycentrics).

Counterpart of src/mesh_renderer/rasterize.py:14-152 with the same three entry
points and ValueErrors.  The reference dispatches between its C++ kernel and a
pure-torch twin through the module global USE_CPP_RASTERIZER; here the only
kernel is the HIP one (there is deliberately no CPU fallback), and the flag is
kept so that code which sets it keeps working.
"""
import torch

from ..common import camera_utils

# Kept for drop-in compatibility (src/mesh_renderer/rasterize.py:14).  Both values
# select the MI355X kernel, which follows the C++ kernel's conventions (NDC z,
# perspective-correct barycentrics); the reference's Python twin differs in its z
# convention and is not reproduced.
USE_CPP_RASTERIZER = True

# True: rasterize_clip_space() with up to 16 float32 attributes is one autograd op whose backward
# is a single pass over the G-buffer (attribute and vertex gradients together).  False: the composed
# ops (BarycentricRasterizer + AttributeInterpolator; ceil(A / 4) + 2 passes) -- same results within
# the parity budget, kept as a cross-check and for larger attribute counts.  Read at call time.
USE_FUSED_BACKWARD = True


def rasterize_barycentric(clip_space_vertices, triangles, image_width, image_height):
    """[V,4] (or [B,V,4]) clip-space vertices -> (triangle ids, barycentrics, z)."""
    from . import rasterize_triangles_ext
    return rasterize_triangles_ext.BarycentricRasterizer.apply(
        clip_space_vertices, triangles, image_width, image_height)


def rasterize(world_space_vertices, attributes, triangles, camera_matrices,
              image_width, image_height, background_value):
    """Project with camera_matrices [B,4,4], then rasterize_clip_space()."""
    clip_space_vertices = camera_utils.transform_homogeneous(
        camera_matrices, world_space_vertices)
    return rasterize_clip_space(clip_space_vertices, attributes, triangles,
                                image_width, image_height, background_value)


def rasterize_clip_space(clip_space_vertices, attributes, triangles,
                         image_width, image_height, background_value):
    """[B,V,4] clip-space vertices + [B,V,A] attributes -> [B,H,W,A] images.

    Pixels outside every triangle take background_value [A].  Row 0 is the bottom
    scanline (render() flips; this function does not)."""
    if not image_width > 0:
        raise ValueError("Image width must be > 0.")
    if not image_height > 0:
        raise ValueError("Image height must be > 0.")
    if len(clip_space_vertices.shape) != 3:
        raise ValueError("The vertex buffer must be 3D.")
    from . import rasterize_triangles_ext as ext
    from .. import _native

    background = torch.as_tensor(background_value).to(
        device=clip_space_vertices.device, dtype=torch.float32)
    if (USE_FUSED_BACKWARD and clip_space_vertices.dtype == torch.float32 and
            attributes.dtype == torch.float32 and len(attributes.shape) == 3 and
            1 <= attributes.shape[2] <= _native.interpolate_raster_max_attributes()):
        return ext.FusedAttributeRasterizer.apply(clip_space_vertices, attributes, triangles, background,
                                                  image_width, image_height)
    ids, bary, _ = ext.BarycentricRasterizer.apply(
        clip_space_vertices, triangles, image_width, image_height)
    return ext.AttributeInterpolator.apply(ids, bary, attributes, triangles, background)
