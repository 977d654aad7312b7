"""Synthetic render jobs used by bench.py and the parity tests.

Mesh: UV sphere in the reference's layout (common/shapes.py), radius 1, normals =
normalised positions, diffuse = 1.  Cameras: image b of B orbits the origin,
  eye_b = 3 * (sin(phi) cos(theta), sin(theta), cos(phi) cos(theta)),
  phi = 2 pi b / B, theta = 0.3 sin(2 pi b / B),
look-at origin, up (0,1,0), fov_y 40 deg, near 0.01, far 10 (the render()
defaults, src/mesh_renderer/render.py:31-33); one white light at the eye.
Everything is computed on the CPU in fp32 so that the CPU baseline and the GPU
kernels are fed identical bits.
"""
import math

import torch

from . import camera_utils, shapes


def orbit_eyes(batch, radius=3.0):
    b = torch.arange(batch, dtype=torch.float64)
    phi = 2.0 * math.pi * b / batch
    theta = 0.3 * torch.sin(2.0 * math.pi * b / batch)
    eye = radius * torch.stack([torch.sin(phi) * torch.cos(theta), torch.sin(theta),
                                torch.cos(phi) * torch.cos(theta)], dim=1)
    return eye.to(torch.float32)


def clip_transforms(eyes, width, height, fov_y=40.0, near=0.01, far=10.0):
    batch = eyes.shape[0]
    center = torch.zeros(batch, 3)
    up = torch.tensor([[0.0, 1.0, 0.0]]).repeat(batch, 1)
    view = camera_utils.look_at(eyes, center, up)
    proj = camera_utils.perspective(width / height, torch.full((batch,), fov_y),
                                    torch.full((batch,), near), torch.full((batch,), far))
    return torch.matmul(proj, view)


def sphere_job(batch, width, height, resolution=50):
    """Returns a dict of CPU tensors describing `batch` independent render jobs."""
    vertices, triangles, normals = shapes.sphere(1.0, resolution)
    eyes = orbit_eyes(batch)
    world = vertices.unsqueeze(0).repeat(batch, 1, 1).contiguous()
    clip = camera_utils.transform_homogeneous(clip_transforms(eyes, width, height), world)
    return {
        "vertices": world,                                   # [B,V,3]
        "normals": normals.unsqueeze(0).repeat(batch, 1, 1).contiguous(),
        "diffuse": torch.ones_like(world),
        "triangles": triangles,                              # [T,3] int32
        "clip": clip.contiguous(),                           # [B,V,4]
        "eyes": eyes,                                        # [B,3]
        "light_positions": eyes.unsqueeze(1).contiguous(),   # [B,1,3]
        "light_intensities": torch.ones(batch, 1, 3),
        "width": width, "height": height,
    }
