"""Image losses used with the renderer.

The reference has no loss module; its tests and examples spell the loss out as
`torch.mean(torch.abs(render - target))` (src/mesh_renderer/mesh_renderer_test.py:250,
src/examples/example5.py:70-92).  In eager torch that is five passes over the image; this is
the same quantity as one HIP pass forward (which also packs sign(image - target), 2 bits per
element) and one backward that reads only those signs.
"""
import torch

from .. import _native


class _MeanAbsError(torch.autograd.Function):
    @staticmethod
    def forward(ctx, image, target):
        need_grad = any(ctx.needs_input_grad)
        loss, signs = _native.l1_loss_forward(image.detach(), target.detach(), want_signs=need_grad)
        if need_grad:
            ctx.save_for_backward(signs)  # 1 byte per pixel instead of both images
        ctx.shape = image.shape
        return loss

    @staticmethod
    def backward(ctx, grad):
        (signs,) = ctx.saved_tensors
        da = _native.l1_loss_backward(signs, ctx.shape, grad.to(torch.float32).reshape(1))
        return (da if ctx.needs_input_grad[0] else None), (-da if ctx.needs_input_grad[1] else None)


import os

USE_FUSED_RENDER_LOSS = os.environ.get("MR_FUSED_RENDER_LOSS", "1") != "0"   # False: always the generic op (dense gradient image)


def remember_target(target):
    """Opt-in for a FIXED loss target (an optimisation loop compares every step's render with the same image): finds
    which 64 x 64 blocks of `target` are all zeros, once, and keeps that map on the tensor; see l1_loss, TARGET."""
    from .rasterize_triangles_ext import remember_target_map
    remember_target_map(target)
    return target


def forget_target(target):
    """Drops remember_target's map."""
    from .rasterize_triangles_ext import forget_target_map
    forget_target_map(target)


def l1_loss(image, target):
    """mean(|image - target|) over every element; same value and gradients as
    torch.mean(torch.abs(image - target)) (sign(0) = 0) -- which, written on render()'s own output, arrives here by
    itself (mesh_renderer/rendered_image.py).

    When `image` is the direct output of render()'s fused diffuse path, the backward skips the
    dense gradient image: the loss's sign codes go straight into the shading backward
    (rasterize_triangles_ext.FusedPhongL1Loss; FusedSpecularL1Loss for the specular path, round 5).  Whether something observes d loss / d image -- image.retain_grad(),
    a hook on the image, torch.autograd.grad(loss, image) -- is decided when the BACKWARD runs (round 5; until then it
    was decided here, and a hook registered after this call never fired): the node then behaves like the generic op.
    (loss.backward(inputs=[image]) is seen too: the engine retain_grad()s what it is given.)
    USE_FUSED_RENDER_LOSS = False takes the generic op always.

    TARGET: on that route the 64 x 64 blocks that are all zeros in BOTH images are not read, if the caller has
    named the target with remember_target(target): the renderer knows its own empty blocks, the target's map is made by
    that call and used while the tensor's data pointer, shape and autograd version counter stay what they were (any
    in-place torch operation on the target moves the counter: the loss then reads every block until the next
    remember_target).  MR_EMPTY_REGIONS=0 switches the maps off."""
    if image.shape != target.shape:
        raise ValueError("image and target must have the same shape")
    if image.dtype != torch.float32 or target.dtype != torch.float32:
        raise ValueError("l1_loss expects float32 tensors")
    if USE_FUSED_RENDER_LOSS and torch.is_grad_enabled() and image.requires_grad:
        from .rasterize_triangles_ext import FusedPhongL1Loss, FusedSpecularL1Loss, take_fused_render
        record = take_fused_render(image)
        if record is not None and record["kind"] == "specular":
            return FusedSpecularL1Loss.apply(image, target, *record["inputs"], record["saved"],
                                             record["has_ambient"], record["has_transforms"])
        if record is not None:
            return FusedPhongL1Loss.apply(image, target, *record["inputs"], record["saved"],
                                          record.get("prepared_state"), record.get("empty_regions"))
    return _MeanAbsError.apply(image, target)
