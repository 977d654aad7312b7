"""Soft-rendering scene API; counterpart of src/soft_mesh_renderer/render.py:15-165."""
import torch

from ..common import camera_utils, meshes
from .rasterize import rasterize


def _per_batch(value, batch_size, device, name):
    if isinstance(value, float):
        return torch.full((batch_size,), value, dtype=torch.float32, device=device)
    if len(value.shape) == 0:
        return value.to(device).unsqueeze(0).repeat(batch_size)
    if list(value.shape) != [batch_size]:
        raise ValueError("%s must be a float, a 0D tensor, or a 1D tensor with "
                         "shape [batch_size]." % name)
    return value.to(device)


def _per_batch_vec3(value, batch_size, name):
    if list(value.shape) == [3]:
        return value.unsqueeze(0).repeat(batch_size, 1)
    if list(value.shape) != [batch_size, 3]:
        raise ValueError("%s must have shape [batch_size, 3] or [3]." % name)
    return value


def render(vertices, triangles, diffuse_colors, camera_position, camera_lookat, camera_up,
           light_positions, light_intensities, image_width, image_height, sigma_val=1e-5,
           gamma_val=1e-4, blur_radius=0.01, fov_y=40.0, near_clip=0.01, far_clip=10.0):
    """Soft-render a batch of scenes -> [B, H, W, 4] RGBA.

    vertices / diffuse_colors [B,V,3]; triangles [T,3] int32, COUNTER-clockwise = front
    (back faces are culled); camera_* [B,3] or [3]; light_positions [B,L,3];
    light_intensities [B,L] (scalar per light).  Vertex normals are computed from the mesh.
    """
    if len(vertices.shape) != 3 or vertices.shape[-1] != 3:
        raise ValueError("Vertices must have shape [batch_size, vertex_count, 3].")
    batch_size = vertices.shape[0]
    device = vertices.device
    if len(light_positions.shape) != 3 or light_positions.shape[-1] != 3:
        raise ValueError("light_positions must have shape [batch_size, light_count, 3].")
    if len(light_intensities.shape) != 2:
        raise ValueError("light_intensities must have shape [batch_size, light_count].")
    if len(diffuse_colors.shape) != 3 or diffuse_colors.shape[-1] != 3:
        raise ValueError("diffuse_colors must have shape [batch_size, vertex_count, 3].")
    camera_position = _per_batch_vec3(camera_position, batch_size, "camera_position")
    camera_lookat = _per_batch_vec3(camera_lookat, batch_size, "camera_lookat")
    if list(camera_up.shape) == [3]:
        camera_up = camera_up.unsqueeze(0).repeat(batch_size, 1)
    elif list(camera_up.shape) != [batch_size, 3]:
        raise ValueError("camera_up must have shape [batch_size, 3] or [3].")
    fov_y = _per_batch(fov_y, batch_size, camera_position.device, "fov_y")
    near_clip = _per_batch(near_clip, batch_size, camera_position.device, "near_clip")
    far_clip = _per_batch(far_clip, batch_size, camera_position.device, "far_clip")

    clip_space_transforms = camera_utils.clip_space_transforms(
        camera_position, camera_lookat, camera_up, fov_y, near_clip, far_clip,
        image_width / image_height, device)
    normals = meshes.compute_vertex_normals(vertices, triangles)
    # NB: like the reference (render.py:150-165), blur_radius is accepted but not forwarded.
    return rasterize(vertices, triangles, normals, diffuse_colors, light_positions.to(device),
                     light_intensities.to(device), clip_space_transforms, image_width, image_height,
                     sigma_val, gamma_val)
