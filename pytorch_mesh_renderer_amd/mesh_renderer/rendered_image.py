"""render()'s output type, and the reference's own loss spelling on the fast path.

The reference has no loss module: its tests and examples write the L1 image loss as
`torch.mean(torch.abs(render - target))` (src/mesh_renderer/mesh_renderer_test.py:250,
src/examples/example5.py:70-92).  In eager torch on a 32 x 1024^2 x 4 image that is three forward passes and
three backward ones over 537 MB tensors, more than the whole rendering step.  render() therefore returns a
`RenderedImage`: a torch.Tensor subclass (same storage, same autograd node, same everything) whose
`__torch_function__` recognises exactly that chain

    image - target   (either order, same shape / dtype / device)     -> a pending difference, nothing launched
    torch.abs(.)                                                       -> still pending
    torch.mean(.)    (no dim, no dtype)                                -> losses.l1_loss(image, target)

(and torch.nn.functional.l1_loss(image, target) with the default reduction) and evaluates it as ONE HIP pass forward
and the fused backward of losses.l1_loss.  Anything else that touches the image or a pending result -- another op, a
method, an attribute, a hook, printing it -- gets an ordinary tensor: the pending difference is then computed with
torch.sub / torch.abs as if this module did not exist, and every result of an op on a RenderedImage is a plain
torch.Tensor.  RECOGNISE_L1_SPELLING = False (or MR_RECOGNISE_L1=0) switches the recognition off, RETURN_SUBCLASS =
False (MR_RENDERED_IMAGE=0) makes render() return plain tensors.

Observing d loss / d image stays possible: FusedPhongL1Loss decides in its BACKWARD whether the image's gradient is
looked at -- retain_grad(), a tensor hook (registered before or after the loss was built, on an image the caller still
holds or has dropped: the loss node saves the image, so its hooks and its retains_grad flag stay visible), torch.autograd.grad naming the image (the image is among that function's arguments, so this class
sees the call), loss.backward(inputs=[image]) / torch.autograd.backward(loss, inputs=[image]) (the engine
retain_grad()s the tensors it is given) -- and then behaves exactly like the generic op
(tests/test_reference_spelling_gpu.py pins every one of them).  losses.USE_FUSED_RENDER_LOSS = False restores stock
autograd.

A pending difference is evaluated LATE, so it remembers its operands' version counters: if the image or the target was
written in place between `image - target` and the use of the result, eager torch would have computed the difference
from the old values; here the use raises the error autograd raises for a saved tensor modified in place.
"""
import os
import threading

import torch

RECOGNISE_L1_SPELLING = os.environ.get("MR_RECOGNISE_L1", "1") != "0"
RETURN_SUBCLASS = os.environ.get("MR_RENDERED_IMAGE", "1") != "0"

_T = torch.Tensor
_SUB = {torch.sub, torch.subtract, _T.sub, _T.subtract, _T.__sub__, _T.__rsub__}
_ABS = {torch.abs, torch.absolute, _T.abs, _T.absolute, _T.__abs__}
_MEAN = {torch.mean, _T.mean}

# torch.autograd.grad(..., inputs containing a RenderedImage) in progress, on any thread (the engine runs backward
# nodes on its own threads): the fused loss node then hands the image its dense gradient like the generic op
_grad_of_image_wanted = 0
_lock = threading.Lock()


def image_gradient_requested():
    return _grad_of_image_wanted > 0


class _ImageGradientRequest:
    def __enter__(self):
        global _grad_of_image_wanted
        with _lock:
            _grad_of_image_wanted += 1

    def __exit__(self, *exc):
        global _grad_of_image_wanted
        with _lock:
            _grad_of_image_wanted -= 1
        return False


def _plain(func, args, kwargs):
    with torch._C.DisableTorchFunctionSubclass():
        return func(*args, **kwargs)


def _names_an_image(inputs):
    if inputs is None:
        return False
    if isinstance(inputs, torch.Tensor):
        inputs = (inputs,)
    try:
        return any(isinstance(t, RenderedImage) for t in inputs)
    except TypeError:
        return False


class RenderedImage(torch.Tensor):
    """A rendered [B, H, W, 4] image: an ordinary tensor that also recognises the reference's L1 loss spelling (module
    docstring).  Results of operations on it are plain torch.Tensors."""

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        if RECOGNISE_L1_SPELLING and torch.is_grad_enabled():
            if func in _SUB:
                pending = _PendingL1.difference(func, args, kwargs)
                if pending is not None:
                    return pending
            elif func is torch.nn.functional.l1_loss:
                loss = _functional_l1(args, kwargs)
                if loss is not None:
                    return loss
        if func is torch.autograd.grad:
            inputs = args[1] if len(args) > 1 else kwargs.get("inputs")
            if _names_an_image(inputs):
                with _ImageGradientRequest():
                    return _plain(func, args, kwargs)
        return _plain(func, args, kwargs)

    def __reduce_ex__(self, proto):   # pickles / deep-copies as the plain tensor it is
        return self.as_subclass(torch.Tensor).__reduce_ex__(proto)

    def __deepcopy__(self, memo):
        return self.as_subclass(torch.Tensor).__deepcopy__(memo)


def _eligible_pair(a, b):
    """(image, target, negated) if exactly this pair can go to losses.l1_loss, else None."""
    if not (isinstance(a, torch.Tensor) and isinstance(b, torch.Tensor)):
        return None
    if isinstance(a, _PendingL1) or isinstance(b, _PendingL1):
        return None
    if isinstance(a, RenderedImage):
        image, target, negated = a, b, False
    elif isinstance(b, RenderedImage):
        image, target, negated = b, a, True
    else:
        return None
    with torch._C.DisableTorchFunctionSubclass():
        ok = (image.grad_fn is not None and image.dim() == 4 and image.shape == target.shape and
              image.dtype == torch.float32 and target.dtype == torch.float32 and image.device == target.device and
              image.is_cuda)
    return (image, target, negated) if ok else None


def _functional_l1(args, kwargs):
    extra = {k: v for k, v in kwargs.items() if k not in ("input", "target")}
    # (size_average, reduce, and in newer torch versions weight: only their defaults)
    if extra.get("reduction", "mean") != "mean" or any(v is not None for k, v in extra.items() if k != "reduction"):
        return None
    values = list(args) + [kwargs[k] for k in ("input", "target") if k in kwargs]
    if len(values) != 2 or len(args) > 2:
        return None
    pair = _eligible_pair(values[0], values[1])
    if pair is None:
        return None
    from . import losses
    return losses.l1_loss(pair[0], pair[1])


class _PendingL1(torch.Tensor):
    """`image - target` (stage "sub") or `|image - target|` (stage "abs") that has not been computed: torch.mean of the
    latter is the fused loss; any other use computes the real tensor first (once) and proceeds with it."""

    @staticmethod
    def _make(image, target, negated, stage):
        with torch._C.DisableTorchFunctionSubclass():
            t = torch.Tensor._make_subclass(_PendingL1, torch.empty(0, device=image.device))
        t._mr_pending = (image, target, negated, stage)
        t._mr_versions = (image._version, target._version)
        t._mr_value = None
        return t

    def _operands(self):
        image, target, negated, stage = self._mr_pending
        if (image._version, target._version) != self._mr_versions:
            raise RuntimeError("one of the operands of this `image - target` has been modified by an inplace operation "
                               "before the difference was used (it is evaluated late: mesh_renderer/rendered_image.py)")
        return image, target, negated, stage

    @staticmethod
    def difference(func, args, kwargs):
        if len(args) != 2 or (kwargs and not (set(kwargs) == {"alpha"} and kwargs["alpha"] == 1)):
            return None
        a, b = (args[1], args[0]) if func is _T.__rsub__ else (args[0], args[1])
        pair = _eligible_pair(a, b)
        return None if pair is None else _PendingL1._make(*pair, "sub")

    def _materialise(self):
        if self._mr_value is None:
            image, target, negated, stage = self._operands()
            with torch._C.DisableTorchFunctionSubclass():   # (the subclass then behaves as the plain tensor it is)
                d = torch.sub(target, image) if negated else torch.sub(image, target)
                self._mr_value = torch.abs(d) if stage == "abs" else d
        return self._mr_value

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        if RECOGNISE_L1_SPELLING and len(args) == 1 and isinstance(args[0], cls) and args[0]._mr_value is None:
            image, target, negated, stage = args[0]._mr_pending
            if func in _ABS and not kwargs and stage == "sub":
                pending = _PendingL1._make(image, target, negated, "abs")
                pending._mr_versions = args[0]._mr_versions
                return pending
            if func in _MEAN and stage == "abs" and all(v is None for v in kwargs.values()) and \
                    set(kwargs) <= {"dtype"} and torch.is_grad_enabled():
                from . import losses
                args[0]._operands()
                return losses.l1_loss(image, target)
            answer = _CHEAP.get(func)
            if answer is not None:
                return answer(image, target)

        def real(x):
            if isinstance(x, cls):
                return x._materialise()
            if isinstance(x, (list, tuple)):
                return type(x)(real(v) for v in x)
            return x
        return func(*[real(a) for a in args], **{k: real(v) for k, v in kwargs.items()})


def _requires_grad(image, target):
    with torch._C.DisableTorchFunctionSubclass():
        return bool(image.requires_grad or target.requires_grad)


# metadata a pending result can answer without being computed
_CHEAP = {
    _T.shape.__get__: lambda i, t: i.shape, _T.dtype.__get__: lambda i, t: i.dtype,
    _T.device.__get__: lambda i, t: i.device, _T.ndim.__get__: lambda i, t: 4,
    _T.is_cuda.__get__: lambda i, t: True, _T.requires_grad.__get__: _requires_grad,
    _T.dim: lambda i, t: 4, _T.size: lambda i, t: i.shape, _T.numel: lambda i, t: _plain(_T.numel, (i,), {}),
}


def wrap(image):
    """`image` as a RenderedImage sharing its storage (inside an autograd.Function's forward: the function's output
    then carries the node)."""
    if not RETURN_SUBCLASS:
        return image
    return torch.Tensor._make_subclass(RenderedImage, image)
