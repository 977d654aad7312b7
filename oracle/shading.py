"""CPU restatement of the reference's eager render path (TEST INFRASTRUCTURE ONLY).

Restates, with torch CPU ops and autograd (which is exactly what the reference
runs):
  rasterize_clip_space   src/mesh_renderer/rasterize.py:66-152
  rasterize              src/mesh_renderer/rasterize.py:27-63
  render / phong_shader  src/mesh_renderer/render.py:157-228, 287-386
  look_at / perspective / transform_homogeneous
                         src/common/camera_utils.py:45-170
The barycentric kernel underneath is oracle/mr_oracle.c (or oracle/_ref when
`use_reference_kernel=True`), wrapped in an autograd.Function like
src/mesh_renderer/rasterize_triangles_ext.py:6-63.

Pinned by tests/golden/render_*.npz, generated from the reference's own Python by
tools/make_goldens.py (tests/test_oracle.py compares).
"""
import math

import numpy as np
import torch

from . import kernel


class _Rasterizer(torch.autograd.Function):
    @staticmethod
    def forward(ctx, clip, triangles, width, height, use_ref):
        fwd = kernel.reference_forward if use_ref else kernel.forward
        ids, bary, z = fwd(clip.detach().numpy(), triangles.numpy(), width, height)
        ids, bary, z = torch.from_numpy(ids), torch.from_numpy(bary), torch.from_numpy(z)
        ctx.save_for_backward(clip.detach(), triangles, ids, bary)
        ctx.use_ref = use_ref
        ctx.mark_non_differentiable(ids)
        return ids, bary, z

    @staticmethod
    def backward(ctx, _, dbary, __):
        clip, triangles, ids, bary = ctx.saved_tensors
        bwd = kernel.reference_backward if ctx.use_ref else kernel.backward
        d = bwd(dbary.contiguous().numpy(), clip.numpy(), triangles.numpy(), ids.numpy(),
                bary.numpy())
        return torch.from_numpy(np.ascontiguousarray(d)), None, None, None, None


def rasterize_barycentric(clip, triangles, width, height, use_reference_kernel=False):
    return _Rasterizer.apply(clip, triangles, width, height, use_reference_kernel)


def look_at(eye, center, up):
    f = center - eye
    f = f / torch.linalg.norm(f, dim=1, keepdim=True)
    s = torch.cross(f, up, dim=-1)
    s = s / torch.linalg.norm(s, dim=1, keepdim=True)
    u = torch.cross(s, f, dim=-1)
    n = eye.shape[0]
    w = torch.tensor([[0.0, 0.0, 0.0, 1.0]]).repeat(n, 1).reshape(n, 4, 1)
    rot = torch.cat([torch.stack([s, u, -f, torch.zeros_like(s)], dim=1), w], dim=2)
    trans = torch.cat([torch.eye(3).unsqueeze(0).repeat(n, 1, 1), (-eye).unsqueeze(2)], 2)
    trans = torch.cat([trans, w.reshape(n, 1, 4)], 1)
    return torch.matmul(rot, trans)


def perspective(aspect, fov_y, near, far):
    fy = 1.0 / torch.tan(fov_y * (math.pi / 360.0))
    rng = far - near
    p22 = -(far + near) / rng
    p23 = -2.0 * (far * near / rng)
    z = torch.zeros_like(p23)
    rows = [fy / aspect, z, z, z, z, fy, z, z, z, z, p22, p23, z, z, -torch.ones_like(p23), z]
    return torch.stack(rows, dim=1).reshape(-1, 4, 4)


def transform_homogeneous(matrices, vertices):
    ones = torch.ones(vertices.shape[0], vertices.shape[1], 1)
    return torch.matmul(torch.cat([vertices, ones], 2), matrices.transpose(1, 2))


def rasterize_clip_space(clip, attributes, triangles, width, height, background,
                         use_reference_kernel=False):
    batch, vertex_count = clip.shape[0], clip.shape[1]
    bary_list, vid_list = [], []
    for b in range(batch):  # the reference's serial batch loop
        ids, bary, _ = rasterize_barycentric(clip[b], triangles, width, height,
                                             use_reference_kernel)
        bary_list.append(bary.reshape(-1, 3))
        vid_list.append(triangles[ids.reshape(-1).long()].long() + b * vertex_count)
    bary = torch.stack(bary_list, 0).reshape(-1, 3)
    vids = torch.stack(vid_list, 0).reshape(-1, 3)
    corners = attributes.reshape(batch * vertex_count, -1)[vids]       # [P,3,A]
    images = (corners * bary.unsqueeze(2)).sum(dim=1).reshape(batch, height, width, -1)
    alpha = torch.clamp((2.0 * bary).sum(dim=1), 0.0, 1.0).reshape(batch, height, width, 1)
    return alpha * images + (1.0 - alpha) * background


class _ForwardBits(torch.autograd.Function):
    """value := bits in the forward pass, identity in the backward pass."""

    @staticmethod
    def forward(ctx, value, bits):
        return bits.clone()

    @staticmethod
    def backward(ctx, grad):
        return grad, None


def rasterize(world_vertices, attributes, triangles, camera_matrices, width, height,
              background, use_reference_kernel=False, clip_bits=None):
    """clip_bits: clip-space vertices to rasterize INSTEAD of this function's own matmul result
    (gradients still flow through the matmul).  The device computes the same product with a
    different summation order; silhouette triangles are seen edge-on and their barycentrics amplify
    that last-bit difference, so a test that means to check the rasterizer and the shading -- not
    the conditioning of the camera transform -- feeds both sides the same bits."""
    clip = transform_homogeneous(camera_matrices, world_vertices)
    if clip_bits is not None:
        clip = _ForwardBits.apply(clip, clip_bits)
    return rasterize_clip_space(clip, attributes, triangles, width, height, background,
                                use_reference_kernel)


def render(vertices, triangles, normals, diffuse_colors, camera_position, camera_lookat,
           camera_up, light_positions, light_intensities, width, height,
           specular_colors=None, shininess_coefficients=None, ambient_color=None,
           fov_y=40.0, near_clip=0.01, far_clip=10.0, use_reference_kernel=False, clip_bits=None,
           power_inside_mask_only=False):
    """Inputs already batched ([B,3] cameras); shininess: None, 0-D tensor or [B,V]; clip_bits: see
    rasterize().

    power_inside_mask_only: NOT the reference's arithmetic -- the specular power is evaluated on
    (base, exponent) = (0, 1) wherever render.py:215's mask is off.  The values are the same (those
    pixels are zeroed by the mask anyway), but autograd no longer multiplies their zero upstream
    gradient by pow(0, -1) = inf of the background's exponent -1, which is how the reference comes
    to return NaN gradients for per-vertex shininess.  This is the semantics the HIP kernels define
    at those entries (shade_spec.hip); the tests compare them with it there."""
    batch = vertices.shape[0]
    pieces = [normals, vertices, diffuse_colors]
    per_vertex_shine = False
    if specular_colors is not None:
        pieces.append(specular_colors)
        per_vertex_shine = shininess_coefficients.dim() == 2
        if per_vertex_shine:
            pieces.append(shininess_coefficients.unsqueeze(2))
    attrs = torch.cat(pieces, 2)
    full = lambda v: torch.full((batch,), float(v))
    proj = perspective(width / height, full(fov_y), full(near_clip), full(far_clip))
    transforms = torch.matmul(proj, look_at(camera_position, camera_lookat, camera_up))
    px = rasterize(vertices, attrs, triangles, transforms, width, height,
                   torch.full((attrs.shape[2],), -1.0), use_reference_kernel, clip_bits)

    P = height * width
    n = torch.nn.functional.normalize(px[..., 0:3], p=2, dim=3).reshape(batch, P, 3)
    pos = px[..., 3:6].reshape(batch, P, 3)
    kd = px[..., 6:9].reshape(batch, P, 3)
    mask = (px[..., 6:9] >= 0.0).any(dim=3).to(torch.float32)

    rgb = torch.zeros(batch, P, 3)
    if ambient_color is not None:
        rgb = rgb + ambient_color.unsqueeze(1) * kd
    to_light = torch.nn.functional.normalize(light_positions.unsqueeze(2) - pos.unsqueeze(1),
                                             p=2, dim=3)
    ndl = torch.clamp((n.unsqueeze(1) * to_light).sum(3), 0.0, 1.0)
    rgb = rgb + (kd.unsqueeze(1) * ndl.unsqueeze(3) * light_intensities.unsqueeze(2)).sum(1)
    if specular_colors is not None:
        ks = px[..., 9:12].reshape(batch, P, 3)
        shine = px[..., 12] if per_vertex_shine else shininess_coefficients.reshape(-1, 1, 1)
        mirror = torch.nn.functional.normalize(
            2.0 * ndl.unsqueeze(3) * n.unsqueeze(1) - to_light, p=2, dim=3)
        to_cam = torch.nn.functional.normalize(camera_position.reshape(batch, 1, 3) - pos,
                                               p=2, dim=2)
        rv = (mirror * to_cam.unsqueeze(1)).sum(3)
        rv = torch.clamp(torch.nn.functional.normalize(rv, p=2, dim=2), 0.0, 1.0)  # over pixels
        rv = torch.where(ndl != 0.0, rv, torch.zeros_like(rv))
        rv = rv.reshape(batch, -1, height, width)
        exponent = shine.unsqueeze(1)
        if power_inside_mask_only:
            inside = mask.reshape(batch, 1, height, width) > 0.5
            rv = torch.where(inside, rv, torch.zeros_like(rv))
            exponent = torch.where(inside, exponent.expand_as(rv), torch.ones_like(rv))
        spec = torch.pow(rv, exponent).reshape(batch, -1, P, 1)
        rgb = rgb + (ks.unsqueeze(1) * spec * light_intensities.unsqueeze(2)).sum(1)
    rgb = rgb.reshape(batch, height, width, 3)
    alpha = mask.reshape(batch, height, width, 1)
    rgb = torch.where(alpha > 0.5, rgb, torch.zeros_like(rgb))
    return torch.flip(torch.cat([rgb, alpha], 3), dims=[1])
