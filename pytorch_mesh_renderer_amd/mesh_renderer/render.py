"""Differentiable Phong rendering of a triangle mesh on MI355X.

Counterpart of src/mesh_renderer/render.py: render() :16-228, phong_shader()
:231-386, tone_mapper() :389-419 -- same signatures, defaults, argument checks
(ValueError messages included) and output conventions: RGBA [B,H,W,4], rows
flipped so that row 0 is the TOP of the image, alpha in {0,1}.

Every tensor lives on the device of `vertices`; rasterization, attribute
interpolation and their gradients run in the HIP kernels behind
mesh_renderer.rasterize; shading is evaluated per pixel from the interpolated
attribute buffer.
"""
import torch

from ..common import camera_utils
from .rasterize import rasterize


# True: render() uses the fused HIP shading kernels whenever they cover the request.  False:
# always the composed path (HIP rasterization + interpolation, torch Phong) -- the same results
# within the parity budget, kept as a cross-check (tests) and for requests the fused kernels do
# not cover.  Read at call time, like rasterize.USE_CPP_RASTERIZER.
USE_FUSED_SHADING = True


def _per_batch(value, batch_size, device, name):
    """float | 0-D tensor | [B] tensor -> [B] float32 tensor on `device`."""
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


def render(vertices, triangles, normals, diffuse_colors, camera_position, camera_lookat,
           camera_up, light_positions, light_intensities, image_width, image_height,
           specular_colors=None, shininess_coefficients=None, ambient_color=None,
           fov_y=40.0, near_clip=0.01, far_clip=10.0):
    """Render a batch of scenes with Phong shading; returns [B, H, W, 4] RGBA.

    Arguments are those of the reference's render() (src/mesh_renderer/render.py
    :16-96): vertices / normals / diffuse_colors [B,V,3], triangles [T,3] int32
    (clockwise winding faces the viewer), camera_* [B,3] or [3], lights [B,L,3],
    optional specular_colors [B,V,3] with shininess_coefficients (float, 0-D,
    [B] or [B,V]), optional ambient_color [B,3], fov_y in degrees.
    """
    if len(vertices.shape) != 3 or vertices.shape[-1] != 3:
        raise ValueError("Vertices must have shape [batch_size, vertex_count, 3].")
    batch_size = vertices.shape[0]
    device = vertices.device
    if len(normals.shape) != 3 or normals.shape[-1] != 3:
        raise ValueError("Normals must have shape [batch_size, vertex_count, 3].")
    if len(light_positions.shape) != 3 or light_positions.shape[-1] != 3:
        raise ValueError("light_positions must have shape [batch_size, light_count, 3].")
    if len(light_intensities.shape) != 3 or light_intensities.shape[-1] != 3:
        raise ValueError("light_intensities must have shape [batch_size, light_count, 3].")
    if len(diffuse_colors.shape) != 3 or diffuse_colors.shape[-1] != 3:
        raise ValueError("diffuse_colors must have shape [batch_size, vertex_count, 3].")
    if ambient_color is not None and list(ambient_color.shape) != [batch_size, 3]:
        raise ValueError("ambient_color must have shape [batch_size, 3].")
    camera_position = _per_batch_vec3(camera_position, batch_size, "camera_position")
    camera_lookat = _per_batch_vec3(camera_lookat, batch_size, "camera_lookat")
    if list(camera_up.shape) == [3]:
        camera_up = camera_up.unsqueeze(0).repeat(batch_size, 1)
    elif list(camera_up.shape) != [batch_size, 3]:
        raise ValueError("camera_up must have shape [batch_size, 3] or [3].")
    fov_y = _per_batch(fov_y, batch_size, camera_position.device, "fov_y")
    near_clip = _per_batch(near_clip, batch_size, camera_position.device, "near_clip")
    far_clip = _per_batch(far_clip, batch_size, camera_position.device, "far_clip")
    if specular_colors is not None and shininess_coefficients is None:
        raise ValueError("Specular colors were supplied without shininess coefficients.")
    if shininess_coefficients is not None and specular_colors is None:
        raise ValueError("Shininess coefficients were supplied without specular colors.")

    if _fused_path_applies(vertices, normals, diffuse_colors, light_positions, specular_colors is not None):
        if specular_colors is None:
            return _render_fused(vertices, triangles, normals, diffuse_colors, camera_position,
                                 camera_lookat, camera_up, light_positions, light_intensities,
                                 image_width, image_height, ambient_color, fov_y, near_clip, far_clip)
        shininess = _fused_shininess(shininess_coefficients, batch_size, vertices.shape[1], device)
        if shininess is not None and specular_colors.shape == vertices.shape:
            return _render_fused(vertices, triangles, normals, diffuse_colors, camera_position,
                                 camera_lookat, camera_up, light_positions, light_intensities,
                                 image_width, image_height, ambient_color, fov_y, near_clip, far_clip,
                                 specular_colors=specular_colors, shininess=shininess)

    pieces = [normals, vertices, diffuse_colors]  # attribute layout, render.py:171-181
    per_vertex_shininess = False
    if specular_colors is not None:
        if isinstance(shininess_coefficients, float):
            shininess_coefficients = torch.tensor(shininess_coefficients, dtype=torch.float32,
                                                  device=device)
        if len(specular_colors.shape) != 3:
            raise ValueError("The specular colors must have shape [batch_size, "
                             "vertex_count, 3].")
        if len(shininess_coefficients.shape) > 2:
            raise ValueError("The shininess coefficients must have shape at "
                             "most [batch_size, vertex_count].")
        pieces.append(specular_colors)
        per_vertex_shininess = len(shininess_coefficients.shape) == 2
        if per_vertex_shininess:
            pieces.append(shininess_coefficients.unsqueeze(2))
    vertex_attributes = torch.cat(pieces, 2)

    clip_space_transforms = camera_utils.clip_space_transforms(
        camera_position, camera_lookat, camera_up, fov_y, near_clip, far_clip,
        image_width / image_height, device)

    # background -1 marks uncovered pixels: a real diffuse colour is never negative
    background = torch.full((vertex_attributes.shape[2],), -1.0, device=device)
    pixel_attributes = rasterize(vertices, vertex_attributes, triangles, clip_space_transforms,
                                 image_width, image_height, background)

    pixel_normals = torch.nn.functional.normalize(pixel_attributes[..., 0:3], p=2, dim=3)
    pixel_positions = pixel_attributes[..., 3:6]
    pixel_diffuse = pixel_attributes[..., 6:9]
    pixel_specular = None
    if specular_colors is not None:
        pixel_specular = pixel_attributes[..., 9:12]
        if per_vertex_shininess:
            shininess_coefficients = pixel_attributes[..., 12]
        else:
            shininess_coefficients = shininess_coefficients.to(device).reshape(-1, 1, 1)
    pixel_mask = (pixel_diffuse >= 0.0).any(dim=3).to(torch.float32)

    return phong_shader(
        normals=pixel_normals, alphas=pixel_mask, pixel_positions=pixel_positions,
        light_positions=light_positions.to(device),
        light_intensities=light_intensities.to(device),
        diffuse_colors=pixel_diffuse,
        camera_position=camera_position.to(device) if pixel_specular is not None else None,
        specular_colors=pixel_specular, shininess_coefficients=shininess_coefficients,
        ambient_color=ambient_color.to(device) if ambient_color is not None else None)


def _fused_path_applies(vertices, normals, diffuse_colors, light_positions, specular):
    """The fused HIP shading kernels cover Phong shading on float32 inputs of matching [B,V,3] shape
    with 1..32 lights: ambient + diffuse, and the specular term (per-image or per-vertex shininess;
    four lights per pass, rasterize_triangles_ext.FusedSpecularPhongRenderer); everything else takes
    the composed path.

    COST beyond four lights: the diffuse forward and the vertex-side backward loop over any light count in one
    pass, but light GRADIENTS are formed four lights at a time -- 1 + ceil(L / 4) passes over the G-buffer --
    and the specular kernels render / differentiate the lights in groups of four (ceil(L / 4) forward passes,
    as many backward ones, their images / gradients added with torch ops): 32 lights with every gradient cost
    about 8 times the pixel work of 4."""
    from .. import _native
    limit = _native.shade_max_lights()
    return (USE_FUSED_SHADING and vertices.dtype == torch.float32 and normals.shape == vertices.shape and
            diffuse_colors.shape == vertices.shape and 1 <= light_positions.shape[1] <= limit)


def _fused_shininess(shininess_coefficients, batch_size, vertex_count, device):
    """float | 0-D | [B] shininess -> [B] float32 tensor on `device` (one exponent per image);
    [B,V] -> itself (per vertex).  Differentiable views of the argument.  None for anything the
    fused specular kernels do not take (other shapes go to the composed path, which raises the
    reference's errors)."""
    if isinstance(shininess_coefficients, float):
        return torch.full((batch_size,), shininess_coefficients, dtype=torch.float32, device=device)
    if not torch.is_tensor(shininess_coefficients):
        return None
    shin = shininess_coefficients.to(device=device, dtype=torch.float32)
    if shin.dim() == 0:
        return shin.reshape(1).expand(batch_size)
    if list(shin.shape) in ([batch_size], [batch_size, vertex_count]):
        return shin
    return None


def _render_fused(vertices, triangles, normals, diffuse_colors, camera_position, camera_lookat,
                  camera_up, light_positions, light_intensities, image_width, image_height,
                  ambient_color, fov_y, near_clip, far_clip, specular_colors=None, shininess=None):
    from .rasterize_triangles_ext import FusedPhongRenderer, FusedSpecularPhongRenderer
    device = vertices.device
    clip_space_transforms = camera_utils.clip_space_transforms(
        camera_position, camera_lookat, camera_up, fov_y, near_clip, far_clip,
        image_width / image_height, device)
    if specular_colors is not None:
        clip = camera_utils.transform_homogeneous(clip_space_transforms, vertices)
        inputs = (clip, vertices, normals, diffuse_colors, specular_colors.to(torch.float32),
                  light_positions.to(device), light_intensities.to(device).to(torch.float32),
                  ambient_color.to(device) if ambient_color is not None else None,
                  camera_position.to(device=device, dtype=torch.float32), shininess)
        image = FusedSpecularPhongRenderer.apply(
            *inputs[:5], triangles, *inputs[5:], image_width, image_height,
            clip_space_transforms if not clip_space_transforms.requires_grad else None)
        if image.grad_fn is None and type(image) is not torch.Tensor:
            image = image.as_subclass(torch.Tensor)
        if image.grad_fn is not None:   # (see the diffuse branch below: losses.l1_loss -> FusedSpecularL1Loss)
            from .rasterize_triangles_ext import remember_fused_render
            remember_fused_render(image.grad_fn, inputs, image, kind="specular")
        return image
    lp, li = light_positions.to(device), light_intensities.to(device).to(torch.float32)
    amb = ambient_color.to(device) if ambient_color is not None else None
    transforms = clip_space_transforms.to(torch.float32)
    image, frames = FusedPhongRenderer.apply(vertices, transforms, normals, diffuse_colors, triangles, lp, li,
                                             amb, image_width, image_height)
    if image.grad_fn is None and type(image) is not torch.Tensor:
        image = image.as_subclass(torch.Tensor)   # nothing to differentiate: no loss spelling to recognise either
    if frames is not None:  # rasterize_triangles_ext.emit_uint8_frames: see to_uint8
        image._mr_frames_u8 = (frames, image._version)
    if image.grad_fn is not None:
        # lets losses.l1_loss differentiate straight to these inputs (FusedPhongL1Loss).  The record
        # is keyed by the autograd node in a weak map -- nothing is stored on the tensor (it stays
        # picklable / deep-copyable), a tensor derived from `image` has another node and never takes
        # that path, and the record dies with the node.
        from .rasterize_triangles_ext import remember_fused_render
        remember_fused_render(image.grad_fn, (vertices, transforms, normals, diffuse_colors, lp, li, amb), image)
    return image


def phong_shader(normals, alphas, pixel_positions, light_positions, light_intensities,
                 diffuse_colors=None, camera_position=None, specular_colors=None,
                 shininess_coefficients=None, ambient_color=None):
    """Per-pixel Phong lighting from rasterized buffers -> [B, H, W, 4] RGBA (flipped).

    normals / pixel_positions / diffuse_colors / specular_colors [B,H,W,3],
    alphas [B,H,W], lights [B,L,3], camera_position [B,3] (enables the specular
    term), shininess broadcastable to [B,H,W], ambient_color [B,3].
    """
    batch_size, image_height, image_width = normals.shape[:-1]
    pixel_count = image_height * image_width
    normals = normals.reshape(batch_size, pixel_count, 3)
    diffuse = diffuse_colors.reshape(batch_size, pixel_count, 3)
    positions = pixel_positions.reshape(batch_size, pixel_count, 3)

    rgb = torch.zeros(batch_size, pixel_count, 3, dtype=normals.dtype, device=normals.device)
    if ambient_color is not None:
        rgb = rgb + ambient_color.unsqueeze(1) * diffuse

    # [B, L, P, 3] unit vectors from each pixel's surface point to each light
    to_light = torch.nn.functional.normalize(
        light_positions.unsqueeze(2) - positions.unsqueeze(1), p=2, dim=3)
    n_dot_l = torch.clamp((normals.unsqueeze(1) * to_light).sum(dim=3), 0.0, 1.0)  # [B,L,P]
    rgb = rgb + (diffuse.unsqueeze(1) * n_dot_l.unsqueeze(3) *
                 light_intensities.unsqueeze(2)).sum(dim=1)

    if camera_position is not None:
        specular = specular_colors.reshape(batch_size, pixel_count, 3)
        mirror = torch.nn.functional.normalize(
            2.0 * n_dot_l.unsqueeze(3) * normals.unsqueeze(1) - to_light, p=2, dim=3)
        to_camera = torch.nn.functional.normalize(
            camera_position.reshape(batch_size, 1, 3) - positions, p=2, dim=2)
        r_dot_v = (mirror * to_camera.unsqueeze(1)).sum(dim=3)                      # [B,L,P]
        # The reference L2-normalises this dot product ACROSS ALL PIXELS of an image
        # (render.py:342-348, dim=2 is the pixel axis) before clamping; kept as is.
        r_dot_v = torch.clamp(torch.nn.functional.normalize(r_dot_v, p=2, dim=2), 0.0, 1.0)
        r_dot_v = torch.where(n_dot_l != 0.0, r_dot_v, torch.zeros_like(r_dot_v))
        r_dot_v = r_dot_v.reshape(batch_size, -1, image_height, image_width)
        specularity = torch.pow(r_dot_v, shininess_coefficients.unsqueeze(1)).reshape(
            batch_size, -1, pixel_count, 1)
        rgb = rgb + (specular.unsqueeze(1) * specularity *
                     light_intensities.unsqueeze(2)).sum(dim=1)

    rgb = rgb.reshape(batch_size, image_height, image_width, 3)
    alpha = alphas.reshape(batch_size, image_height, image_width, 1)
    rgb = torch.where(alpha > 0.5, rgb, torch.zeros_like(rgb))
    return torch.flip(torch.cat([rgb, alpha], dim=3), dims=[1])


def tone_mapper(image, gamma):
    """image_out = A * image_in ** gamma with A chosen per image so that max == 1;
    clipped to [0, 1].  image [B,H,W,C].  Counterpart of src/mesh_renderer/render.py:389-419.

    On a HIP device this is two streaming kernels (per-image max of the powers, then scale /
    clamp).  When a gradient is wanted -- the reference's version is differentiable through
    autograd -- the same expression is evaluated with torch ops on the device instead."""
    from .. import _native
    needs_grad = torch.is_grad_enabled() and (image.requires_grad or
                                              (torch.is_tensor(gamma) and gamma.requires_grad))
    if image.is_cuda and image.dtype == torch.float32 and not needs_grad:
        return _native.tone_map(image, float(gamma))
    batch_size = image.shape[0]
    corrected = torch.pow(image, gamma)
    image_max = corrected.reshape(batch_size, -1).max(dim=1).values
    return torch.clamp(corrected / image_max.reshape(batch_size, 1, 1, 1), 0.0, 1.0)


def tone_mapper_uint8(image, gamma):
    """tone_mapper followed by the examples' 8-bit frame conversion, `(x * 255.0).astype(np.uint8)`,
    in the same two kernels: 4 B/element read twice, 1 B/element written (display / file / hand-over
    frames; not differentiable)."""
    from .. import _native
    if image.dtype != torch.float32:
        raise ValueError("tone_mapper_uint8 expects a float32 image")
    return _native.tone_map(image.detach(), float(gamma), as_uint8=True)


def to_uint8(image):
    """8-bit frames for display, files and the multi-GPU hand-over: the conversion the
    reference's examples apply on the host, `(image * 255.0).astype(np.uint8)`
    (src/examples/example1.py:52, example5.py:81), as one HIP pass on the device (the value is
    clamped to [0, 1] first; NaN exports as 0).  Not differentiable."""
    from .. import _native
    if image.dtype != torch.float32:
        raise ValueError("to_uint8 expects a float32 image")
    ready = getattr(image, "_mr_frames_u8", None)   # written by render()'s forward kernel, if asked to
    if ready is not None and ready[1] == image._version and ready[0].shape == image.shape:
        return ready[0]
    return _native.export_u8(image.detach())
