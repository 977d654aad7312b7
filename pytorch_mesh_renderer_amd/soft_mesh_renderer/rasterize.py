"""Soft rasterization (Liu et al. 2019) on MI355X.

Counterpart of src/soft_mesh_renderer/rasterize.py: rasterize() :14-110 and
rasterize_batch() :212-424, same argument order and meaning.  The per-pixel Python
loop and the quadtree of the reference are replaced by the HIP kernels of
csrc/soft.hip (forward and hand-derived backward) behind one autograd.Function.
"""
import torch

from .. import _native
from ..common import camera_utils


class SoftRasterizer(torch.autograd.Function):
    @staticmethod
    def forward(ctx, clip, positions, normals, diffuse, triangles, light_positions, light_intensities,
                image_width, image_height, sigma_val, gamma_val, blur_radius):
        args = [t.detach().contiguous() for t in (clip, positions, normals, diffuse)]
        lp, li = light_positions.detach().contiguous(), light_intensities.detach().contiguous()
        # with a backward pass to come the forward keeps its records and candidate lists for it
        keep = any(ctx.needs_input_grad)
        out = _native.soft_forward(args[0], args[1], args[2], args[3], triangles, lp, li,
                                   int(image_width), int(image_height), float(sigma_val),
                                   float(gamma_val), float(blur_radius), keep_prepared=keep)
        rgba, aux = out[0], out[1]
        ctx.prepared = out[2] if keep else None
        ctx.save_for_backward(rgba, aux, args[0], args[1], args[2], args[3], triangles, lp, li)
        ctx.params = (float(sigma_val), float(gamma_val), float(blur_radius))
        return rgba

    @staticmethod
    def backward(ctx, drgba):
        rgba, aux, clip, positions, normals, diffuse, triangles, lp, li = ctx.saved_tensors
        sigma, gamma, blur = ctx.params
        dclip, dp, dn, dd, dlp, dli = _native.soft_backward(
            drgba.contiguous(), rgba, aux, clip, positions, normals, diffuse, triangles, lp, li,
            sigma, gamma, blur, prepared=ctx.prepared)
        return dclip, dp, dn, dd, None, dlp, dli, None, None, None, None, None


def _check_lights(light_positions):
    n = light_positions.shape[-2]
    if not 1 <= n <= _native.soft_max_lights():
        raise ValueError("the soft rasterizer supports 1..%d lights, got %d"
                         % (_native.soft_max_lights(), n))


def rasterize_batch(clip_space_vertices, triangles, world_space_vertices, normals, diffuse_colors,
                    light_positions, light_intensities, image_width, image_height, sigma_val,
                    gamma_val, blur_radius=0.01):
    """One image: clip [V,4], triangles [T,3], attributes [V,3], lights [L,3] / [L] -> [H,W,4]."""
    _check_lights(light_positions)
    out = SoftRasterizer.apply(
        clip_space_vertices.unsqueeze(0), world_space_vertices.unsqueeze(0), normals.unsqueeze(0),
        diffuse_colors.unsqueeze(0), triangles, light_positions.unsqueeze(0),
        light_intensities.unsqueeze(0), image_width, image_height, float(sigma_val), float(gamma_val),
        float(blur_radius))
    return out[0]


def rasterize(world_space_vertices, triangles, normals, diffuse_colors, light_positions,
              light_intensities, camera_matrices, image_width, image_height, sigma_val, gamma_val,
              blur_radius=0.01):
    """Batched: vertices / normals / diffuse [B,V,3], lights [B,L,3] / [B,L], camera_matrices
    [B,4,4] -> [B,H,W,4] RGBA (row 0 = top; RGB soft-aggregated over depth, A = silhouette).

    As in the reference (rasterize.py:91-107) the blur radius actually used is 0.01: its
    rasterize() accepts the argument but never forwards it to rasterize_batch."""
    _check_lights(light_positions)
    clip = camera_utils.transform_homogeneous(camera_matrices, world_space_vertices)
    return SoftRasterizer.apply(clip, world_space_vertices, normals, diffuse_colors, triangles,
                                light_positions, light_intensities, image_width, image_height,
                                float(sigma_val), float(gamma_val), 0.01)
