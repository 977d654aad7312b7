"""ctypes binding of the C-ABI HIP library (include/mesh_raster.h).

This is the ONLY compute back end of the package: there is no CPU or eager
fallback.  If libmesh_raster_hip.so is missing or a tensor is not on a HIP
device, the call raises.  PyTorch supplies device memory and the stream; the
kernels are ours.
"""
import ctypes
import os
import subprocess

import torch

_CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
# MR_NATIVE_LIB_PATH: development only (tools/raster_bench.py --variant loads the stage-timing
# build libmesh_raster_hip_probes.so this way); the product always loads the in-tree library.
LIB_PATH = os.environ.get("MR_NATIVE_LIB_PATH") or os.path.join(_CSRC, "libmesh_raster_hip.so")

ABI_VERSION = 354
GBUFFER_NORMALISED = 1   # mesh_raster.h, MR_GBUFFER_NORMALISED
TIMER_RASTER_FORWARD, TIMER_SHADE_BACKWARD, TIMER_SHADE_FORWARD, TIMER_RASTER_BACKWARD, TIMER_L1_FORWARD = 0, 1, 2, 3, 4
MR_OK, MR_EINVAL, MR_EWORKSPACE, MR_ELAUNCH = 0, -1, -2, -3
_ERR = {MR_EINVAL: "invalid argument", MR_EWORKSPACE: "workspace too small or misaligned",
        MR_ELAUNCH: "HIP launch failed"}

_lib = None
_workspaces = {}
_pending_timers = {}


_deterministic = False


def set_deterministic(on):
    """Bit-reproducible gradients for the fused render / rasterize backward passes (see
    mr_set_deterministic in include/mesh_raster.h): fixed-point integer accumulation instead of float
    atomics, ~10 % slower.  Process-wide on the Python side: the flag is handed to the library by
    whichever thread launches a backward kernel (autograd runs them on its own thread).  Returns the
    previous setting."""
    global _deterministic
    before = _deterministic
    _deterministic = bool(on)
    return before


def deterministic():
    """Whether set_deterministic(True) is in force."""
    return _deterministic


def _sync_deterministic():
    lib().mr_set_deterministic(1 if _deterministic else 0)


# Test / measurement hook (include/mesh_raster_debug.h): pixel kernel of the shading backward --
# 0 automatic, 1 the rows kernel, 2 the lane-accumulating kernel wherever it exists.  Like the
# deterministic flag it is handed over by the launching thread.
_shade_backward_kernel = int(os.environ.get("MR_SHADE_BACKWARD_KERNEL", "0"))


def debug_set_shade_backward_kernel(which):
    global _shade_backward_kernel
    before = _shade_backward_kernel
    _shade_backward_kernel = int(which)
    return before


def debug_set_raster_repeat(n):
    """Measurement only (include/mesh_raster_debug.h): this thread's next forward calls launch k_raster n times."""
    _check(lib().mr_debug_set_raster_repeat(int(n)), "mr_debug_set_raster_repeat")


def debug_last_accumulate_kernel():
    """Tests only (include/mesh_raster_debug.h): the functor of the most recent backward pixel pass any thread
    launched, e.g. 'ShadeFoldLaneFn<1, true>' ('' before the first)."""
    text = lib().mr_debug_last_accumulate_kernel().decode()
    return text.split("Fn = mr::")[-1].rstrip("]").replace("(anonymous namespace)::", "") if "Fn = " in text else text


def debug_soft_nearest(points, seg_a, seg_b):
    """Tests only (include/mesh_raster_debug.h): the SoftRas kernels' nearest-point-on-a-segment
    evaluation for [n,2] device points / segment ends -> [n,4] = (nearest x, y, t, squared distance)."""
    for name, t in (("points", points), ("seg_a", seg_a), ("seg_b", seg_b)):
        _chk(name, t, _F32, None, 2)
    if not (points.shape == seg_a.shape == seg_b.shape):
        raise ValueError("points, seg_a and seg_b must have the same [n,2] shape")
    dev = _require_device(points, seg_a, seg_b)
    out = torch.empty(points.shape[0], 4, dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        rc = lib().mr_debug_soft_nearest(_ptr(points.contiguous()), _ptr(seg_a.contiguous()), _ptr(seg_b.contiguous()),
                                         points.shape[0], _ptr(out), _stream(dev))
    _check(rc, "mr_debug_soft_nearest")
    return out


def time_next_kernel(which, start_event, stop_event):
    """Measurement (bench.py): HIP events (ctypes.c_void_p) to record around the NEXT launch of
    kernel `which` (TIMER_*) made through this module, from whatever thread makes it -- autograd runs
    backward passes on its own thread, and the library's one-shot timers are per thread, so the pair
    is handed to mr_time_next_kernel by the launching wrapper itself."""
    _pending_timers[which] = (start_event, stop_event)


def _arm_timer(which):
    pair = _pending_timers.pop(which, None)
    if pair is not None:
        lib().mr_time_next_kernel(which, pair[0], pair[1])



class NativeLibraryError(RuntimeError):
    pass


def build(verbose=False):
    """Compile libmesh_raster_hip.so in-tree for gfx950 (hipcc cross-compiles on CPU)."""
    out = subprocess.run(["make", "-C", _CSRC, "all"], capture_output=True, text=True)
    if verbose:
        print(out.stdout)
    if out.returncode != 0:
        raise NativeLibraryError("hipcc build failed:\n" + out.stdout + out.stderr)
    return LIB_PATH


def lib():
    """Load the shared library (never builds implicitly, never falls back)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise NativeLibraryError(
                "%s not found: run `python -c 'import __graft_entry__ as g; g.build()'` "
                "(or make -C pytorch_mesh_renderer_amd/csrc). There is no fallback path." % LIB_PATH)
        L = ctypes.CDLL(LIB_PATH)
        vp, ci, sz = ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t
        L.mr_version.restype = ci
        if L.mr_version() != ABI_VERSION:   # a stale .so would be called with the wrong signatures
            raise NativeLibraryError("%s has ABI version %d, this package needs %d: rebuild it (make -C "
                                     "pytorch_mesh_renderer_amd/csrc)" % (LIB_PATH, L.mr_version(), ABI_VERSION))
        L.mr_last_hip_error.restype = ci
        L.mr_set_deterministic.argtypes = [ci]
        L.mr_set_deterministic.restype = ci
        L.mr_time_next_kernel.argtypes = [ci, vp, vp]
        L.mr_time_next_kernel.restype = ci
        # include/mesh_raster_debug.h (tests and tools only)
        L.mr_debug_set_raster_probe.argtypes = [ci]
        L.mr_debug_set_raster_probe.restype = ci
        L.mr_debug_set_raster_region_edge.argtypes = [ci]
        L.mr_debug_set_raster_region_edge.restype = ci
        L.mr_debug_set_shade_backward_kernel.argtypes = [ci]
        L.mr_debug_set_shade_backward_kernel.restype = ci
        L.mr_debug_set_raster_repeat.argtypes = [ci]
        L.mr_debug_set_raster_repeat.restype = ci
        L.mr_debug_last_accumulate_kernel.argtypes = []
        L.mr_debug_last_accumulate_kernel.restype = ctypes.c_char_p
        L.mr_debug_soft_nearest.argtypes = [vp, vp, vp, ci, vp, vp]
        L.mr_debug_soft_nearest.restype = ci
        L.mr_rasterize_forward_workspace_bytes.argtypes = [ci] * 5
        L.mr_rasterize_forward_workspace_bytes.restype = sz
        L.mr_rasterize_forward.argtypes = [vp, vp, ci, ci, ci, ci, ci, vp, vp, vp, vp, sz, vp]
        L.mr_rasterize_forward.restype = ci
        L.mr_rasterize_backward_workspace_bytes.argtypes = [ci] * 5
        L.mr_rasterize_backward_workspace_bytes.restype = sz
        L.mr_rasterize_backward.argtypes = [vp, vp, vp, vp, vp, ci, ci, ci, ci, ci, vp, vp, sz, vp]
        L.mr_rasterize_backward.restype = ci
        L.mr_interpolate_forward.argtypes = [vp, vp, vp, vp, vp, ci, ci, ci, ci, ci, ci, vp, vp]
        L.mr_interpolate_forward.restype = ci
        L.mr_interpolate_backward_workspace_bytes.argtypes = [ci] * 6
        L.mr_interpolate_backward_workspace_bytes.restype = sz
        L.mr_interpolate_backward.argtypes = [vp, vp, vp, vp, vp, vp, ci, ci, ci, ci, ci, ci,
                                              vp, vp, vp, sz, vp]
        L.mr_interpolate_backward.restype = ci
        L.mr_shade_max_lights.restype = ci
        L.mr_shade_fast_lights.argtypes = []
        L.mr_shade_fast_lights.restype = ci
        L.mr_shade_forward_workspace_bytes.argtypes = [ci] * 5
        L.mr_shade_forward_workspace_bytes.restype = sz
        L.mr_shade_forward.argtypes = [vp] * 9 + [ci] * 6 + [vp, vp, sz, vp]
        L.mr_shade_forward.restype = ci
        L.mr_shade_backward_workspace_bytes.argtypes = [ci] * 5
        L.mr_shade_backward_workspace_bytes.restype = sz
        L.mr_shade_backward_prepared_bytes.argtypes = [ci] * 2
        L.mr_shade_backward_prepared_bytes.restype = sz
        L.mr_shade_backward.argtypes = [vp] * 11 + [ci] * 6 + [vp] * 9 + [ci, vp, vp, vp, sz, vp]
        L.mr_shade_backward.restype = ci
        L.mr_shade_backward_l1_workspace_bytes.argtypes = [ci] * 5
        L.mr_shade_backward_l1_workspace_bytes.restype = sz
        L.mr_shade_backward_l1.argtypes = [vp] * 12 + [ci] * 6 + [vp] * 9 + [ci, vp, vp, vp, sz, vp]
        L.mr_shade_backward_l1.restype = ci
        L.mr_soft_max_lights.restype = ci
        L.mr_soft_workspace_bytes.argtypes = [ci] * 5
        L.mr_soft_workspace_bytes.restype = sz
        cf = ctypes.c_float
        L.mr_soft_forward.argtypes = [vp] * 7 + [ci] * 6 + [cf] * 3 + [vp, vp, vp, sz, vp]
        L.mr_soft_forward.restype = ci
        L.mr_soft_prepared_bytes.argtypes = [ci] * 5
        L.mr_soft_prepared_bytes.restype = sz
        L.mr_soft_backward.argtypes = [vp] * 10 + [ci] * 6 + [cf] * 3 + [vp] * 6 + [vp, vp, sz, vp]
        L.mr_soft_backward.restype = ci
        fp = vp
        L.mr_camera_transforms.argtypes = [fp, fp, fp, fp, fp, fp, ctypes.c_float, ci, fp, vp, vp]
        L.mr_camera_transforms.restype = ci
        L.mr_camera_transforms_backward.argtypes = [fp, fp, fp, fp, fp, fp, fp, ctypes.c_float, ci, fp, fp, fp, vp]
        L.mr_camera_transforms_backward.restype = ci
        L.mr_l1_loss_partials.argtypes = []
        L.mr_l1_loss_partials.restype = ci
        L.mr_l1_loss_forward.argtypes = [vp, vp, sz, vp, vp, vp, vp]
        L.mr_l1_loss_forward.restype = ci
        L.mr_l1_loss_backward.argtypes = [vp, sz, vp, vp, vp]
        L.mr_l1_loss_backward.restype = ci
        L.mr_interpolate_raster_max_attributes.restype = ci
        L.mr_interpolate_raster_backward_workspace_bytes.argtypes = [ci] * 6
        L.mr_interpolate_raster_backward_workspace_bytes.restype = sz
        L.mr_interpolate_records_bytes.argtypes = [ci] * 3
        L.mr_interpolate_records_bytes.restype = sz
        L.mr_interpolate_forward_records.argtypes = [vp] * 5 + [ci] * 6 + [vp, vp, sz, vp]
        L.mr_interpolate_forward_records.restype = ci
        L.mr_rasterize_interpolate_forward.argtypes = [vp] * 4 + [ci] * 6 + [vp] * 5 + [sz, vp, sz, vp]
        L.mr_rasterize_interpolate_forward.restype = ci
        L.mr_interpolate_raster_backward.argtypes = [vp] * 10 + [ci] * 6 + [vp, vp, ci, vp, sz, vp]
        L.mr_interpolate_raster_backward.restype = ci
        L.mr_vertex_transform.argtypes = [vp, vp, ci, ci, vp, vp]
        L.mr_vertex_transform.restype = ci
        L.mr_render_forward.argtypes = [vp] * 8 + [ci] * 6 + [vp, vp, vp, vp, ci] + [vp] * 6 + [sz, vp]
        L.mr_empty_regions_bytes.argtypes = [ci] * 3
        L.mr_empty_regions_bytes.restype = sz
        L.mr_image_empty_regions.argtypes = [vp, ci, ci, ci, vp, vp]
        L.mr_image_empty_regions.restype = ci
        L.mr_l1_loss_forward_regions.argtypes = [vp, vp, ci, ci, ci, vp, vp, vp, vp, vp, vp]
        L.mr_l1_loss_forward_regions.restype = ci
        L.mr_render_forward.restype = ci
        L.mr_shade_specular_forward_workspace_bytes.argtypes = [ci] * 5
        L.mr_shade_specular_forward_workspace_bytes.restype = sz
        L.mr_shade_specular_forward.argtypes = [vp] * 12 + [ci] * 7 + [vp, vp, ci, vp, sz, vp]
        L.mr_shade_specular_forward.restype = ci
        L.mr_rasterize_specular_norms_workspace_bytes.argtypes = [ci] * 5
        L.mr_rasterize_specular_norms_workspace_bytes.restype = sz
        L.mr_rasterize_specular_norms_forward.argtypes = [vp] * 6 + [ci] * 6 + [vp, vp, vp, ci, vp, vp, sz, vp]
        L.mr_rasterize_specular_norms_forward.restype = ci
        L.mr_shade_specular_backward_workspace_bytes.argtypes = [ci] * 5
        L.mr_shade_specular_backward_workspace_bytes.restype = sz
        L.mr_shade_specular_backward.argtypes = [vp] * 14 + [ci, vp] + [ci] * 6 + [vp] * 7 + [vp, vp] + [vp, ci, ci] + [vp, sz, vp]
        L.mr_shade_specular_backward.restype = ci
        L.mr_shade_specular_backward_l1_workspace_bytes.argtypes = [ci] * 5
        L.mr_shade_specular_backward_l1_workspace_bytes.restype = sz
        L.mr_shade_specular_backward_l1.argtypes = [vp] * 15 + [ci, vp] + [ci] * 6 + [vp] * 7 + [vp, vp] + [vp, ci, ci] + [vp, sz, vp]
        L.mr_shade_specular_backward_l1.restype = ci
        L.mr_export_u8.argtypes = [vp, sz, vp, vp]
        L.mr_export_u8.restype = ci
        L.mr_tone_map.argtypes = [vp, ci, sz, cf, vp, vp, vp, vp]
        L.mr_tone_map.restype = ci
        L.mr_vertex_normals_forward.argtypes = [vp] * 4 + [ci] * 3 + [vp, vp, vp]
        L.mr_vertex_normals_forward.restype = ci
        L.mr_vertex_normals_backward.argtypes = [vp] * 6 + [ci] * 3 + [vp, vp]
        L.mr_vertex_normals_backward.restype = ci
        _lib = L
    return _lib


def _check(rc, what):
    if rc != MR_OK:
        extra = ""
        if rc == MR_ELAUNCH:
            extra = " (hipError %d)" % lib().mr_last_hip_error()
        raise RuntimeError("%s failed: %s%s" % (what, _ERR.get(rc, "error %d" % rc), extra))


def _require_device(*tensors):
    dev = tensors[0].device
    if dev.type != "cuda":
        raise RuntimeError(
            "pytorch_mesh_renderer_amd runs on MI355X only: got a %s tensor. Move inputs to "
            "a HIP device ('cuda'); there is no CPU fallback." % dev.type)
    for t in tensors:
        if t.device != dev:
            raise RuntimeError("all tensors must be on the same device")
    return dev


def _chk(name, t, dtype, *shape):
    """One argument of a C-ABI call: dtype as the reference's accessor<> demands (RuntimeError, like
    c10::Error there), rank and extents as the call's other arguments imply (ValueError, like the
    reference's Python layer).  The library takes raw device pointers: a tensor of another dtype or
    shape would be read as garbage or out of bounds, so NOTHING reaches it unchecked.  None in
    `shape` = any extent."""
    if not torch.is_tensor(t):
        raise TypeError("%s must be a tensor" % name)
    if t.dtype != dtype:
        raise RuntimeError("%s must be %s, got %s" % (name, str(dtype).replace("torch.", ""),
                                                      str(t.dtype).replace("torch.", "")))
    if t.dim() != len(shape) or any(want is not None and want != have for want, have in zip(shape, t.shape)):
        raise ValueError("%s must have shape [%s], got %s" % (
            name, ", ".join("*" if d is None else str(d) for d in shape), list(t.shape)))


_F32, _I32, _U8 = torch.float32, torch.int32, torch.uint8


def _chk_mesh(clip, triangles):
    _chk("clip-space vertices", clip, _F32, None, None, 4)
    _chk("triangles", triangles, _I32, None, 3)
    return clip.shape[0], clip.shape[1], triangles.shape[0]


def _chk_gbuffer(ids, bary, B):
    _chk("triangle ids", ids, _I32, B, None, None)
    _chk("barycentrics", bary, _F32, B, ids.shape[1], ids.shape[2], 3)
    return ids.shape[1], ids.shape[2]


def _chk_lights(light_positions, light_intensities, ambient, B, max_lights):
    _chk("light_positions", light_positions, _F32, B, None, 3)
    L = light_positions.shape[1]
    if not 1 <= L <= max_lights:
        raise ValueError("1..%d lights are supported, got %d" % (max_lights, L))
    _chk("light_intensities", light_intensities, _F32, B, L, 3)
    if ambient is not None:
        _chk("ambient_color", ambient, _F32, B, 3)
    return L


def _chk_shininess(shininess, B, V):
    """[B] -> False (one exponent per image), [B,V] -> True (per vertex)."""
    if torch.is_tensor(shininess) and shininess.dim() == 2:
        _chk("shininess", shininess, _F32, B, V)
        return True
    _chk("shininess", shininess, _F32, B)
    return False


_WORKSPACE_LIMIT_BYTES = 64 << 30   # refuse absurd scratch requests instead of trying to allocate them


def _workspace(dev, nbytes):
    """Per-(device, stream) scratch tensor, grown on demand.  Reuse is safe because every consumer is
    enqueued on the same stream.  While a stream is being captured into a HIP graph the scratch comes
    from the capture's own memory pool and is NOT cached: a tensor of that pool must not outlive the
    graph or be handed to eager launches.  At most one buffer per (device, stream) is kept; a
    request beyond _WORKSPACE_LIMIT_BYTES (sizes: INTEGRATION.md, "Scratch memory") is an error."""
    if nbytes == 0:
        return None, 0
    if nbytes > _WORKSPACE_LIMIT_BYTES:
        raise RuntimeError("this call needs %.1f GiB of scratch memory (limit %.0f GiB): split the batch"
                           % (nbytes / 2.0 ** 30, _WORKSPACE_LIMIT_BYTES / 2.0 ** 30))
    if torch.cuda.is_current_stream_capturing():
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        return ws, ws.numel()
    key = (dev.index, torch.cuda.current_stream(dev).cuda_stream)
    ws = _workspaces.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, device=dev)
        _workspaces[key] = ws
    return ws, ws.numel()


def _stream(dev):
    return ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def rasterize_forward(clip, triangles, width, height):
    """clip [B,V,4] f32, triangles [T,3] i32 (device) -> ids [B,H,W] i32, bary [B,H,W,3], z [B,H,W]."""
    _chk_mesh(clip, triangles)
    dev = _require_device(clip, triangles)
    L = lib()
    clip = clip.contiguous()
    triangles = triangles.contiguous()
    B, V, _ = clip.shape
    T = triangles.shape[0]
    ids = torch.empty(B, height, width, dtype=torch.int32, device=dev)
    bary = torch.empty(B, height, width, 3, dtype=torch.float32, device=dev)
    z = torch.empty(B, height, width, dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        need = L.mr_rasterize_forward_workspace_bytes(B, V, T, width, height)
        ws, have = _workspace(dev, need)
        _arm_timer(TIMER_RASTER_FORWARD)
        rc = L.mr_rasterize_forward(_ptr(clip), _ptr(triangles), B, V, T, width, height,
                                    _ptr(ids), _ptr(bary), _ptr(z), _ptr(ws), have, _stream(dev))
    _check(rc, "mr_rasterize_forward")
    return ids, bary, z


def rasterize_backward(dbary, clip, triangles, ids, bary):
    """-> dclip [B,V,4] f32."""
    B, _, _ = _chk_mesh(clip, triangles)
    h, w = _chk_gbuffer(ids, bary, B)
    _chk("df_dbarycentric_coords", dbary, _F32, B, h, w, 3)
    dev = _require_device(dbary, clip, triangles, ids, bary)
    L = lib()
    dbary, clip, triangles = dbary.contiguous(), clip.contiguous(), triangles.contiguous()
    ids, bary = ids.contiguous(), bary.contiguous()
    B, V, _ = clip.shape
    T = triangles.shape[0]
    _, H, W = ids.shape
    dclip = torch.empty(B, V, 4, dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        need = L.mr_rasterize_backward_workspace_bytes(B, V, T, W, H)
        ws, have = _workspace(dev, need)
        _arm_timer(TIMER_RASTER_BACKWARD)
        _sync_deterministic()
        rc = L.mr_rasterize_backward(_ptr(dbary), _ptr(clip), _ptr(triangles), _ptr(ids),
                                     _ptr(bary), B, V, T, W, H, _ptr(dclip), _ptr(ws), have,
                                     _stream(dev))
    _check(rc, "mr_rasterize_backward")
    return dclip


def interpolate_forward(ids, bary, attrs, triangles, background):
    """attrs [B,V,A], background [A] -> [B,H,W,A]."""
    _chk("triangles", triangles, _I32, None, 3)
    _chk("attributes", attrs, _F32, None, None, None)
    _chk_gbuffer(ids, bary, attrs.shape[0])
    _chk("background", background, _F32, attrs.shape[2])
    dev = _require_device(ids, bary, attrs, triangles, background)
    L = lib()
    ids, bary, attrs = ids.contiguous(), bary.contiguous(), attrs.contiguous()
    triangles, background = triangles.contiguous(), background.contiguous()
    B, H, W = ids.shape
    V, A = attrs.shape[1], attrs.shape[2]
    T = triangles.shape[0]
    out = torch.empty(B, H, W, A, dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        rc = L.mr_interpolate_forward(_ptr(ids), _ptr(bary), _ptr(attrs), _ptr(triangles),
                                      _ptr(background), B, V, T, W, H, A, _ptr(out), _stream(dev))
    _check(rc, "mr_interpolate_forward")
    return out


def interpolate_backward(dout, ids, bary, attrs, triangles, background):
    """-> (dattrs [B,V,A], dbary [B,H,W,3])."""
    _chk("triangles", triangles, _I32, None, 3)
    _chk("attributes", attrs, _F32, None, None, None)
    h, w = _chk_gbuffer(ids, bary, attrs.shape[0])
    _chk("background", background, _F32, attrs.shape[2])
    _chk("upstream gradient", dout, _F32, attrs.shape[0], h, w, attrs.shape[2])
    dev = _require_device(dout, ids, bary, attrs, triangles, background)
    L = lib()
    dout, ids, bary = dout.contiguous(), ids.contiguous(), bary.contiguous()
    attrs, triangles, background = attrs.contiguous(), triangles.contiguous(), background.contiguous()
    B, H, W = ids.shape
    V, A = attrs.shape[1], attrs.shape[2]
    T = triangles.shape[0]
    dattrs = torch.empty(B, V, A, dtype=torch.float32, device=dev)
    dbary = torch.empty(B, H, W, 3, dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        need = L.mr_interpolate_backward_workspace_bytes(B, V, T, W, H, A)
        ws, have = _workspace(dev, need)
        rc = L.mr_interpolate_backward(_ptr(dout), _ptr(ids), _ptr(bary), _ptr(attrs),
                                       _ptr(triangles), _ptr(background), B, V, T, W, H, A,
                                       _ptr(dattrs), _ptr(dbary), _ptr(ws), have, _stream(dev))
    _check(rc, "mr_interpolate_backward")
    return dattrs, dbary


def shade_max_lights():
    return int(lib().mr_shade_max_lights())


def shade_fast_lights():
    """Lights per call of the specular kernels and of the light gradients (kept in registers)."""
    return int(lib().mr_shade_fast_lights())


def _aligned_bytes(nbytes, dev):
    """A fresh uint8 tensor of `nbytes` whose data pointer is 256-byte aligned."""
    raw = torch.empty(max(int(nbytes), 1) + 256, dtype=torch.uint8, device=dev)
    off = (-raw.data_ptr()) % 256
    return raw[off:off + max(int(nbytes), 1)]


def shade_forward(ids, bary, normals, positions, diffuse, triangles, light_positions,
                  light_intensities, ambient, keep_corner_records=False):
    """Fused interpolation + diffuse/ambient Phong: -> rgba [B,H,W,4] (row 0 = top); with
    keep_corner_records also the gathered per-triangle attribute records, for shade_backward."""
    tensors = [ids, bary, normals, positions, diffuse, triangles, light_positions, light_intensities]
    _chk("triangles", triangles, _I32, None, 3)
    _chk("positions", positions, _F32, None, None, 3)
    B, V = positions.shape[0], positions.shape[1]
    _chk("normals", normals, _F32, B, V, 3)
    _chk("diffuse colors", diffuse, _F32, B, V, 3)
    _chk_gbuffer(ids, bary, B)
    _chk_lights(light_positions, light_intensities, ambient, B, shade_max_lights())
    dev = _require_device(*(tensors + ([ambient] if ambient is not None else [])))
    L = lib()
    ids, bary, normals, positions, diffuse, triangles, light_positions, light_intensities = [
        t.contiguous() for t in tensors]
    ambient = ambient.contiguous() if ambient is not None else None
    B, H, W = ids.shape
    V, T, nl = normals.shape[1], triangles.shape[0], light_positions.shape[1]
    rgba = torch.empty(B, H, W, 4, dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        need = L.mr_shade_forward_workspace_bytes(B, V, T, W, H)
        if keep_corner_records:   # own buffer (not the shared workspace): handed to the backward
            ws, have = _aligned_bytes(need, dev), need
        else:
            ws, have = _workspace(dev, need)
        _arm_timer(TIMER_SHADE_FORWARD)
        rc = L.mr_shade_forward(_ptr(ids), _ptr(bary), _ptr(normals), _ptr(positions), _ptr(diffuse),
                                _ptr(triangles), _ptr(light_positions), _ptr(light_intensities),
                                _ptr(ambient), B, V, T, W, H, nl, _ptr(rgba), _ptr(ws), have,
                                _stream(dev))
    _check(rc, "mr_shade_forward")
    return (rgba, ws) if keep_corner_records else rgba


def vertex_transform(vertices, transforms):
    """clip [B,V,4] = transforms [B,4,4] . (vertices [B,V,3], 1), as render_forward forms it."""
    _chk("vertices", vertices, _F32, None, None, 3)
    B, V = vertices.shape[0], vertices.shape[1]
    _chk("clip-space transforms", transforms, _F32, B, 4, 4)
    dev = _require_device(vertices, transforms)
    vertices, transforms = vertices.contiguous(), transforms.contiguous()
    clip = torch.empty(B, V, 4, dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        rc = lib().mr_vertex_transform(_ptr(vertices), _ptr(transforms), B, V, _ptr(clip), _stream(dev))
    _check(rc, "mr_vertex_transform")
    return clip


def render_forward(vertices, transforms, normals, diffuse, triangles, light_positions, light_intensities,
                   ambient, width, height, want_z=True, want_u8=False, prepare_backward=False, want_empty_regions=False):
    """render()'s forward from world-space vertices: clip-space transform, rasterizer and shading
    (the shading is the epilogue of the rasterizer's tile walk: one pass over the pixels)
    -> (clip, ids, bary, z, rgba, corner_records); with want_z=False the depth plane is not written
    (z is returned as None); with want_u8=True a seventh value follows, the image as [B,H,W,4] uint8
    frames (what export_u8(rgba) would return).

    prepare_backward=True (the caller will differentiate to the world-space vertices only): the setup kernel
    also writes the folded shading backward's records and clears its accumulator rows; the block is returned as
    the LAST value and goes to shade_backward(..., prepared=), which then launches no setup kernel.

    want_empty_regions=True: a [B, ceil(H/64), ceil(W/64)] uint8 map follows the optional frames (before the
    prepared block): 1 = that 64 x 64 block of the G-buffer holds no candidate triangle (background / transparent
    black); l1_loss_forward(..., empty_a=, empty_b=) and shade_backward(..., empty_regions=) skip such blocks."""
    tensors = [vertices, transforms, normals, diffuse, triangles, light_positions, light_intensities]
    _chk("vertices", vertices, _F32, None, None, 3)
    _chk("triangles", triangles, _I32, None, 3)
    B, V, T = vertices.shape[0], vertices.shape[1], triangles.shape[0]
    _chk("clip-space transforms", transforms, _F32, B, 4, 4)
    for name, t in (("normals", normals), ("diffuse colors", diffuse)):
        _chk(name, t, _F32, B, V, 3)
    _chk_lights(light_positions, light_intensities, ambient, B, shade_max_lights())
    dev = _require_device(*(tensors + ([ambient] if ambient is not None else [])))
    L = lib()
    vertices, transforms, normals, diffuse, triangles, light_positions, light_intensities = [
        t.contiguous() for t in tensors]
    ambient = ambient.contiguous() if ambient is not None else None
    nl = light_positions.shape[1]
    clip = torch.empty(B, V, 4, dtype=torch.float32, device=dev)
    ids = torch.empty(B, height, width, dtype=torch.int32, device=dev)
    bary = torch.empty(B, height, width, 3, dtype=torch.float32, device=dev)
    z = torch.empty(B, height, width, dtype=torch.float32, device=dev)
    rgba = torch.empty(B, height, width, 4, dtype=torch.float32, device=dev)
    frames = torch.empty(B, height, width, 4, dtype=torch.uint8, device=dev) if want_u8 else None
    with torch.cuda.device(dev):
        records = _aligned_bytes(L.mr_shade_forward_workspace_bytes(B, V, T, width, height), dev)
        prepared = _aligned_bytes(L.mr_shade_backward_prepared_bytes(B, T), dev) if prepare_backward else None
        empty = (torch.empty(B, (height + 63) // 64, (width + 63) // 64, dtype=torch.uint8, device=dev)
                 if want_empty_regions else None)
        need = L.mr_rasterize_forward_workspace_bytes(B, V, T, width, height)
        ws, have = _workspace(dev, need)
        _arm_timer(TIMER_RASTER_FORWARD)
        rc = L.mr_render_forward(_ptr(vertices), _ptr(transforms), _ptr(normals), _ptr(diffuse), _ptr(triangles),
                                 _ptr(light_positions), _ptr(light_intensities), _ptr(ambient), B, V, T,
                                 width, height, nl, _ptr(clip), _ptr(ids), _ptr(bary), _ptr(z), int(bool(want_z)),
                                 _ptr(rgba), _ptr(frames), _ptr(records), _ptr(prepared), _ptr(empty), _ptr(ws), have,
                                 _stream(dev))
    _check(rc, "mr_render_forward")
    out = (clip, ids, bary, (z if want_z else None), rgba, records) + ((frames,) if want_u8 else ())
    return out + ((empty,) if want_empty_regions else ()) + ((prepared,) if prepare_backward else ())


def interpolate_raster_max_attributes():
    return int(lib().mr_interpolate_raster_max_attributes())


def interpolate_forward_records(ids, bary, attrs, triangles, background):
    """interpolate_forward through per-(image, triangle) corner records, for at most
    interpolate_raster_max_attributes() attributes -> (out [B,H,W,A], records for the backward)."""
    _chk("triangles", triangles, _I32, None, 3)
    _chk("attributes", attrs, _F32, None, None, None)
    _chk_gbuffer(ids, bary, attrs.shape[0])
    _chk("background", background, _F32, attrs.shape[2])
    if not 1 <= attrs.shape[2] <= interpolate_raster_max_attributes():
        raise ValueError("1..%d attributes are supported here" % interpolate_raster_max_attributes())
    dev = _require_device(ids, bary, attrs, triangles, background)
    L = lib()
    ids, bary, attrs = ids.contiguous(), bary.contiguous(), attrs.contiguous()
    triangles, background = triangles.contiguous(), background.contiguous()
    B, H, W = ids.shape
    V, A, T = attrs.shape[1], attrs.shape[2], triangles.shape[0]
    out = torch.empty(B, H, W, A, dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        need = L.mr_interpolate_records_bytes(B, T, A)
        records = _aligned_bytes(need, dev)
        rc = L.mr_interpolate_forward_records(_ptr(ids), _ptr(bary), _ptr(attrs), _ptr(triangles),
                                              _ptr(background), B, V, T, W, H, A, _ptr(out), _ptr(records),
                                              need, _stream(dev))
    _check(rc, "mr_interpolate_forward_records")
    return out, records


def rasterize_interpolate_forward(clip, attrs, triangles, background, width, height):
    """rasterize_clip_space()'s forward in one pass over the pixels (the interpolation is the epilogue of the
    rasterizer's tile walk) for at most interpolate_raster_max_attributes() attributes
    -> (ids [B,H,W], bary [B,H,W,3], out [B,H,W,A], records for interpolate_raster_backward)."""
    B, V, T = _chk_mesh(clip, triangles)
    _chk("attributes", attrs, _F32, B, V, None)
    A = attrs.shape[2]
    _chk("background", background, _F32, A)
    if not 1 <= A <= interpolate_raster_max_attributes():
        raise ValueError("1..%d attributes are supported here" % interpolate_raster_max_attributes())
    dev = _require_device(clip, attrs, triangles, background)
    L = lib()
    clip, attrs, triangles, background = [t.contiguous() for t in (clip, attrs, triangles, background)]
    ids = torch.empty(B, height, width, dtype=torch.int32, device=dev)
    bary = torch.empty(B, height, width, 3, dtype=torch.float32, device=dev)
    z = torch.empty(B, height, width, dtype=torch.float32, device=dev)
    out = torch.empty(B, height, width, A, dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        need_rec = L.mr_interpolate_records_bytes(B, T, A)
        records = _aligned_bytes(need_rec, dev)
        need = L.mr_rasterize_forward_workspace_bytes(B, V, T, width, height)
        ws, have = _workspace(dev, need)
        rc = L.mr_rasterize_interpolate_forward(_ptr(clip), _ptr(attrs), _ptr(triangles), _ptr(background), B, V, T,
                                                width, height, A, _ptr(ids), _ptr(bary), _ptr(z), _ptr(out),
                                                _ptr(records), need_rec, _ptr(ws), have, _stream(dev))
    _check(rc, "mr_rasterize_interpolate_forward")
    return ids, bary, out, records


def interpolate_raster_backward(dout, ids, bary, clip, attrs, triangles, background, adjacency,
                                corner_records=None, normalised_gbuffer=False):
    """One-pass backward of interpolation + rasterization -> (dattributes [B,V,A], dclip [B,V,4])."""
    tensors = [dout, ids, bary, clip, attrs, triangles, background, adjacency[0], adjacency[1]]
    B, V, _ = _chk_mesh(clip, triangles)
    _chk("attributes", attrs, _F32, B, V, None)
    h, w = _chk_gbuffer(ids, bary, B)
    _chk("background", background, _F32, attrs.shape[2])
    _chk("upstream gradient", dout, _F32, B, h, w, attrs.shape[2])
    _chk("adjacency offsets", adjacency[0], _I32, V + 1)
    _chk("adjacency entries", adjacency[1], _I32, None)
    dev = _require_device(*tensors)
    L = lib()
    dout, ids, bary, clip, attrs, triangles, background, offsets, entries = [t.contiguous() for t in tensors]
    B, H, W = ids.shape
    V, A, T = attrs.shape[1], attrs.shape[2], triangles.shape[0]
    dattrs = torch.empty(B, V, A, dtype=torch.float32, device=dev)
    dclip = torch.empty(B, V, 4, dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        _sync_deterministic()
        L.mr_debug_set_shade_backward_kernel(_shade_backward_kernel)
        need = L.mr_interpolate_raster_backward_workspace_bytes(B, V, T, W, H, A)
        ws, have = _workspace(dev, need)
        rc = L.mr_interpolate_raster_backward(
            _ptr(dout), _ptr(ids), _ptr(bary), _ptr(clip), _ptr(attrs), _ptr(triangles), _ptr(background),
            _ptr(offsets), _ptr(entries), _ptr(corner_records), B, V, T, W, H, A, _ptr(dattrs), _ptr(dclip),
            GBUFFER_NORMALISED if normalised_gbuffer else 0, _ptr(ws), have, _stream(dev))
    _check(rc, "mr_interpolate_raster_backward")
    return dattrs, dclip


def vertex_adjacency(triangles, vertex_count):
    """CSR vertex -> (triangle, corner) adjacency of an int32 [T,3] triangle array on its device:
    (offsets [V+1] i32, entries [n] i32), entry = 3 * triangle + corner grouped by vertex; corners
    whose vertex index is out of range are left out.  Cached on the tensor object (keyed by its
    version counter), so a mesh's topology is analysed once."""
    cached = getattr(triangles, "_mr_adjacency", None)
    if cached is not None and cached[0] == (triangles._version, int(vertex_count), triangles.data_ptr()):
        return cached[1], cached[2]
    flat = triangles.reshape(-1).to(torch.int64)
    key = torch.where((flat >= 0) & (flat < vertex_count), flat, torch.full_like(flat, vertex_count))
    order = torch.argsort(key, stable=True)
    # scatter_add, not bincount: bincount reads the maximum back to the host (a device sync on every
    # cache miss, e.g. when the caller passes a fresh `triangles.to(device)` each step)
    counts = torch.zeros(vertex_count + 1, dtype=torch.int64, device=triangles.device)
    counts.scatter_add_(0, key, torch.ones_like(key))
    offsets = torch.zeros(vertex_count + 1, dtype=torch.int64, device=triangles.device)
    offsets[1:] = torch.cumsum(counts[:vertex_count], 0)
    offsets, entries = offsets.to(torch.int32), order.to(torch.int32).contiguous()
    try:
        triangles._mr_adjacency = ((triangles._version, int(vertex_count), triangles.data_ptr()), offsets, entries)
    except AttributeError:
        pass
    return offsets, entries


def shade_backward(drgba, ids, bary, clip, normals, positions, diffuse, triangles, light_positions,
                   light_intensities, ambient, corner_records=None, adjacency=None, l1_signs=None,
                   transforms=None, want_light_grads=True, want_normal_grads=True, want_diffuse_grads=True,
                   normalised_gbuffer=False, want_clip_grads=True, prepared=None, empty_regions=None):
    """_shade_backward_call for any light count up to shade_max_lights().  The kernels keep the light
    gradients' 6 L sums in registers, four lights per call; with more lights the vertex-side gradients
    come from one call over all lights (a run-time loop, no light gradients) and each group of four
    lights gets a call of its own for d light_positions / d light_intensities -- a light's gradient
    depends on that light, the upstream gradient and the pixel's attributes, not on the other lights.
    (1 + ceil(L / 4) passes over the G-buffer: more than four lights WITH light gradients is the rare
    case -- the reference's tests and examples use one to three.)"""
    nl = light_positions.shape[1]
    kw = dict(corner_records=corner_records, adjacency=adjacency, l1_signs=l1_signs, transforms=transforms,
              want_normal_grads=want_normal_grads, want_diffuse_grads=want_diffuse_grads,
              normalised_gbuffer=normalised_gbuffer, want_clip_grads=want_clip_grads, prepared=prepared,
              empty_regions=empty_regions)
    fast = shade_fast_lights() if nl > 4 else nl
    if nl <= fast or not want_light_grads:
        return _shade_backward_call(drgba, ids, bary, clip, normals, positions, diffuse, triangles,
                                    light_positions, light_intensities, ambient, want_light_grads=want_light_grads,
                                    **kw)
    out = _shade_backward_call(drgba, ids, bary, clip, normals, positions, diffuse, triangles, light_positions,
                               light_intensities, ambient, want_light_grads=False, **kw)
    chunk_kw = dict(kw, want_normal_grads=False, want_diffuse_grads=False) if adjacency is not None else kw
    dlpos, dlint, damb = [], [], None
    for first in range(0, nl, fast):
        part = _shade_backward_call(drgba, ids, bary, clip, normals, positions, diffuse, triangles,
                                    light_positions[:, first:first + fast].contiguous(),
                                    light_intensities[:, first:first + fast].contiguous(),
                                    ambient if first == 0 else None, want_light_grads=True, **chunk_kw)
        dlpos.append(part[4])
        dlint.append(part[5])
        if first == 0:
            damb = part[6]
    return out[:4] + (torch.cat(dlpos, 1), torch.cat(dlint, 1), damb)


def _shade_backward_call(drgba, ids, bary, clip, normals, positions, diffuse, triangles, light_positions,
                         light_intensities, ambient, corner_records=None, adjacency=None, l1_signs=None,
                         transforms=None, want_light_grads=True, want_normal_grads=True, want_diffuse_grads=True,
                         normalised_gbuffer=False, want_clip_grads=True, prepared=None, empty_regions=None):
    """-> (dclip [B,V,4], dnormals, dpositions, ddiffuse [B,V,3], dlight_positions,
    dlight_intensities [B,L,3], dambient [B,3] or None); with want_light_grads=False the last three
    are None and the kernel leaves their accumulation out; want_normal_grads / want_diffuse_grads=False
    (needs `adjacency`) return None for that gradient and its sums are not formed either -- the
    backward then runs the lane-accumulating kernel (18 or 27 sums per triangle instead of 36).

    transforms ([B,4,4], needs `adjacency`): clip = transforms . (positions, 1); dpositions then also
    holds the clip-space gradient pulled back through that product (the whole d / d world vertices).

    l1_signs: the packed sign codes of l1_loss_forward(rgba, target); `drgba` is then the 1-element
    upstream gradient of that loss and the [B,H,W,4] gradient image is never materialised.

    normalised_gbuffer: ids / bary are what rasterize_forward / render_forward wrote for these vertices
    (MR_GBUFFER_NORMALISED): the pixel pass leaves the alpha terms out, same bits.

    want_clip_grads=False (needs `transforms`): dclip is returned as None -- the caller differentiates to
    the world-space vertices only; the pull-back through the transforms is then folded into the pixel
    pass where the library has that variant (9 sums per triangle instead of 18).

    prepared: the block render_forward(..., prepare_backward=True) returned for these inputs (or None).
    empty_regions: render_forward(..., want_empty_regions=True)'s map for this G-buffer (or None)."""
    if empty_regions is not None:
        _chk("empty_regions", empty_regions, _U8, clip.shape[0], (ids.shape[1] + 63) // 64, (ids.shape[2] + 63) // 64)
        empty_regions = empty_regions.contiguous()
    if prepared is not None and (prepared.dtype != torch.uint8 or
                                 prepared.numel() < lib().mr_shade_backward_prepared_bytes(clip.shape[0], triangles.shape[0])):
        raise ValueError("prepared must be the block render_forward(prepare_backward=True) returned for these inputs")
    if not want_clip_grads and transforms is None:
        raise ValueError("without transforms the clip-space gradient is the vertex gradient: it cannot be left out")
    tensors = [drgba, ids, bary, clip, normals, positions, diffuse, triangles, light_positions,
               light_intensities]
    B, V, _ = _chk_mesh(clip, triangles)
    for name, t in (("normals", normals), ("positions", positions), ("diffuse colors", diffuse)):
        _chk(name, t, _F32, B, V, 3)
    h, w = _chk_gbuffer(ids, bary, B)
    _chk_lights(light_positions, light_intensities, ambient, B, shade_max_lights())
    if l1_signs is None:
        _chk("upstream gradient", drgba, _F32, B, h, w, 4)
    else:
        _chk("upstream gradient of the loss", drgba, _F32, 1)
    if adjacency is not None:
        _chk("adjacency offsets", adjacency[0], _I32, V + 1)
        _chk("adjacency entries", adjacency[1], _I32, None)
    if transforms is not None:
        if adjacency is None:
            raise ValueError("transforms are applied by the per-vertex gather: pass the adjacency too")
        _chk("clip-space transforms", transforms, _F32, B, 4, 4)
        tensors = tensors + [transforms]
        transforms = transforms.contiguous()
    dev = _require_device(*(tensors + ([ambient] if ambient is not None else [])))
    L = lib()
    (drgba, ids, bary, clip, normals, positions, diffuse, triangles, light_positions,
     light_intensities) = [t.contiguous() for t in tensors[:10]]
    ambient = ambient.contiguous() if ambient is not None else None
    B, H, W = ids.shape
    V, T, nl = normals.shape[1], triangles.shape[0], light_positions.shape[1]
    # one allocation, laid out back to back: the library then zeroes all outputs with one memset
    n4, n3, nlg = B * V * 4, B * V * 3, (B * (6 * nl + 3) if want_light_grads else 0)
    flat = torch.empty(n4 + 3 * n3 + nlg, dtype=torch.float32, device=dev)
    dclip = flat[:n4].view(B, V, 4)
    dn = flat[n4:n4 + n3].view(B, V, 3)
    dp = flat[n4 + n3:n4 + 2 * n3].view(B, V, 3)
    dd = flat[n4 + 2 * n3:n4 + 3 * n3].view(B, V, 3)
    lg = flat[n4 + 3 * n3:].view(B, 6 * nl + 3) if want_light_grads else None
    if adjacency is None and not (want_normal_grads and want_diffuse_grads):
        raise ValueError("leaving a gradient out needs the per-vertex gather: pass the adjacency")
    if not want_normal_grads:
        dn = None
    if not want_diffuse_grads:
        dd = None
    if not want_clip_grads:
        dclip = None
    tail = (B, V, T, W, H, nl, _ptr(dclip), _ptr(dn), _ptr(dp), _ptr(dd), _ptr(lg), _ptr(corner_records),
            _ptr(adjacency[0]) if adjacency is not None else None,
            _ptr(adjacency[1]) if adjacency is not None else None, _ptr(transforms),
            GBUFFER_NORMALISED if normalised_gbuffer else 0, _ptr(prepared), _ptr(empty_regions))
    with torch.cuda.device(dev):
        _arm_timer(TIMER_SHADE_BACKWARD)
        _sync_deterministic()
        L.mr_debug_set_shade_backward_kernel(_shade_backward_kernel)
        if l1_signs is not None:
            if l1_signs.dtype != torch.uint8 or l1_signs.numel() != B * H * W or drgba.numel() != 1:
                raise ValueError("l1_signs must hold one byte per pixel and drgba the scalar upstream gradient")
            need = L.mr_shade_backward_l1_workspace_bytes(B, V, T, W, H)
            ws, have = _workspace(dev, need)
            rc = L.mr_shade_backward_l1(_ptr(l1_signs.contiguous()), _ptr(drgba), _ptr(ids), _ptr(bary), _ptr(clip),
                                        _ptr(normals), _ptr(positions), _ptr(diffuse), _ptr(triangles),
                                        _ptr(light_positions), _ptr(light_intensities), _ptr(ambient),
                                        *tail, _ptr(ws), have, _stream(dev))
        else:
            need = L.mr_shade_backward_workspace_bytes(B, V, T, W, H)
            ws, have = _workspace(dev, need)
            rc = L.mr_shade_backward(_ptr(drgba), _ptr(ids), _ptr(bary), _ptr(clip), _ptr(normals),
                                     _ptr(positions), _ptr(diffuse), _ptr(triangles),
                                     _ptr(light_positions), _ptr(light_intensities), _ptr(ambient),
                                     *tail, _ptr(ws), have, _stream(dev))
    _check(rc, "mr_shade_backward")
    if lg is None:
        return dclip, dn, dp, dd, None, None, None
    dlpos = lg[:, :3 * nl].reshape(B, nl, 3)
    dlint = lg[:, 3 * nl:6 * nl].reshape(B, nl, 3)
    damb = lg[:, 6 * nl:] if ambient is not None else None
    return dclip, dn, dp, dd, dlpos, dlint, damb


def rasterize_specular_norms_forward(clip, triangles, normals, positions, light_positions, camera_position,
                                     width, height, want_z=False):
    """rasterize_forward(clip, triangles, width, height) AND the specular term's across-pixels norms in one pass over
    the pixels -> (ids, bary, z or None, norms2 [B,L]); 1 <= L <= shade_fast_lights().  norms2 goes to
    shade_specular_forward(..., norms2=) -- its norm pass over the G-buffer then does not run -- and to
    shade_specular_backward as before."""
    B, V, T = _chk_mesh(clip, triangles)
    for name, t in (("normals", normals), ("positions", positions)):
        _chk(name, t, _F32, B, V, 3)
    _chk("light_positions", light_positions, _F32, B, None, 3)
    nl = light_positions.shape[1]
    if not 1 <= nl <= shade_fast_lights():
        raise ValueError("1..%d lights per call" % shade_fast_lights())
    _chk("camera_position", camera_position, _F32, B, 3)
    dev = _require_device(clip, triangles, normals, positions, light_positions, camera_position)
    clip, triangles, normals, positions, light_positions, camera_position = [
        t.contiguous() for t in (clip, triangles, normals, positions, light_positions, camera_position)]
    width, height = int(width), int(height)
    ids = torch.empty(B, height, width, dtype=torch.int32, device=dev)
    bary = torch.empty(B, height, width, 3, dtype=torch.float32, device=dev)
    z = torch.empty(B, height, width, dtype=torch.float32, device=dev)
    norms2 = torch.empty(B, nl, dtype=torch.float32, device=dev)
    L = lib()
    with torch.cuda.device(dev):
        need = L.mr_rasterize_specular_norms_workspace_bytes(B, V, T, width, height)
        ws, have = _workspace(dev, need)
        rc = L.mr_rasterize_specular_norms_forward(
            _ptr(clip), _ptr(triangles), _ptr(normals), _ptr(positions), _ptr(light_positions), _ptr(camera_position),
            B, V, T, width, height, nl, _ptr(ids), _ptr(bary), _ptr(z), int(bool(want_z)), _ptr(norms2), _ptr(ws), have,
            _stream(dev))
    _check(rc, "mr_rasterize_specular_norms_forward")
    return ids, bary, (z if want_z else None), norms2


def shade_specular_forward(ids, bary, normals, positions, diffuse, specular, triangles, light_positions,
                           light_intensities, ambient, camera_position, shininess, norms2=None):
    """Fused interpolation + Phong with the specular term -> (rgba [B,H,W,4], norms2 [B,L]).
    shininess: [B] (one exponent per image) or [B,V] (per vertex).
    norms2 ([B,L], rasterize_specular_norms_forward's for the same G-buffer, normals, positions, lights and camera):
    given, the norm pass does not run and the same tensor is returned."""
    tensors = [ids, bary, normals, positions, diffuse, specular, triangles, light_positions,
               light_intensities, camera_position, shininess]
    _chk("triangles", triangles, _I32, None, 3)
    _chk("positions", positions, _F32, None, None, 3)
    B, V = positions.shape[0], positions.shape[1]
    for name, t in (("normals", normals), ("diffuse colors", diffuse), ("specular colors", specular)):
        _chk(name, t, _F32, B, V, 3)
    _chk_gbuffer(ids, bary, B)
    _chk_lights(light_positions, light_intensities, ambient, B, shade_fast_lights())
    _chk("camera_position", camera_position, _F32, B, 3)
    per_vertex = _chk_shininess(shininess, B, V)
    dev = _require_device(*(tensors + ([ambient] if ambient is not None else [])))
    L = lib()
    (ids, bary, normals, positions, diffuse, specular, triangles, light_positions, light_intensities,
     camera_position, shininess) = [t.contiguous() for t in tensors]
    ambient = ambient.contiguous() if ambient is not None else None
    B, H, W = ids.shape
    V, T, nl = normals.shape[1], triangles.shape[0], light_positions.shape[1]
    rgba = torch.empty(B, H, W, 4, dtype=torch.float32, device=dev)
    given = norms2 is not None
    if given:
        _chk("norms2", norms2, _F32, B, nl)
        _require_device(norms2)
        norms2 = norms2.contiguous()
    else:
        norms2 = torch.empty(B, nl, dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        need = L.mr_shade_specular_forward_workspace_bytes(B, V, T, W, H)
        ws, have = _workspace(dev, need)
        rc = L.mr_shade_specular_forward(
            _ptr(ids), _ptr(bary), _ptr(normals), _ptr(positions), _ptr(diffuse), _ptr(specular),
            _ptr(triangles), _ptr(light_positions), _ptr(light_intensities), _ptr(ambient),
            _ptr(camera_position), _ptr(shininess), int(per_vertex), B, V, T, W, H, nl, _ptr(rgba),
            _ptr(norms2), int(given), _ptr(ws), have, _stream(dev))
    _check(rc, "mr_shade_specular_forward")
    return rgba, norms2


# grads_wanted bits of shade_specular_backward (include/mesh_raster.h: MR_GRAD_*)
GRAD_NORMALS, GRAD_POSITIONS, GRAD_DIFFUSE, GRAD_SPECULAR, GRAD_SHININESS, GRAD_LIGHTS, GRAD_CLIP = 1, 2, 4, 8, 16, 32, 64
GRAD_ALL = 127


def shade_specular_backward(drgba, ids, bary, clip, normals, positions, diffuse, specular, triangles,
                            light_positions, light_intensities, ambient, camera_position, shininess,
                            norms2, adjacency=None, transforms=None, normalised_gbuffer=False,
                            grads_wanted=GRAD_ALL, l1_signs=None):
    """-> (dclip [B,V,4], dnormals, dpositions, ddiffuse, dspecular [B,V,3], dlight_positions,
    dlight_intensities [B,L,3], dambient [B,3] or None, dcamera_position [B,3], dshininess shaped
    like shininess ([B] or [B,V])).  adjacency: vertex_adjacency(triangles, V) -- the per-triangle
    sums are then gathered per vertex (no atomics; required by the deterministic mode).

    grads_wanted: GRAD_* bits of the results the caller will read (the others come back unspecified).  With only
    GRAD_POSITIONS / GRAD_CLIP wanted and normalised_gbuffer=True (ids / bary are rasterize_forward's own output
    for `clip`) the pixel pass is the lane-accumulating kernel; with transforms ([B,4,4], clip = M (position, 1))
    and GRAD_CLIP not wanted, dpositions is the whole gradient w.r.t. the world-space vertices.

    l1_signs ([B,H,W] uint8, l1_loss_forward's sign codes for the image these inputs shaded): drgba is then the
    device scalar d L / d loss and the upstream image is upstream * sign / (B*H*W*4) without being written out where
    the pixel kernel can read the codes (mr_shade_specular_backward_l1)."""
    tensors = [drgba, ids, bary, clip, normals, positions, diffuse, specular, triangles,
               light_positions, light_intensities, camera_position, shininess, norms2]
    B, V, _ = _chk_mesh(clip, triangles)
    for name, t in (("normals", normals), ("positions", positions), ("diffuse colors", diffuse),
                    ("specular colors", specular)):
        _chk(name, t, _F32, B, V, 3)
    h, w = _chk_gbuffer(ids, bary, B)
    nl_ = _chk_lights(light_positions, light_intensities, ambient, B, shade_fast_lights())
    if l1_signs is None:
        _chk("upstream gradient", drgba, _F32, B, h, w, 4)
    else:
        if drgba.dtype != torch.float32 or drgba.numel() != 1:
            raise ValueError("with l1_signs the upstream gradient is one float32 (d L / d loss)")
        if l1_signs.dtype != torch.uint8 or l1_signs.numel() != B * h * w:
            raise ValueError("l1_signs must hold one byte per pixel")
        _require_device(l1_signs)
        l1_signs = l1_signs.contiguous()
    _chk("camera_position", camera_position, _F32, B, 3)
    per_vertex = _chk_shininess(shininess, B, V)
    _chk("norms2", norms2, _F32, B, nl_)
    dev = _require_device(*(tensors + ([ambient] if ambient is not None else [])))
    L = lib()
    (drgba, ids, bary, clip, normals, positions, diffuse, specular, triangles, light_positions,
     light_intensities, camera_position, shininess, norms2) = [t.contiguous() for t in tensors]
    ambient = ambient.contiguous() if ambient is not None else None
    B, H, W = ids.shape
    V, T, nl = normals.shape[1], triangles.shape[0], light_positions.shape[1]
    dclip = torch.empty(B, V, 4, dtype=torch.float32, device=dev)
    dn, dp, dd, dsp = [torch.empty(B, V, 3, dtype=torch.float32, device=dev) for _ in range(4)]
    lg = torch.empty(B, 6 * nl + 7, dtype=torch.float32, device=dev)
    dshin_v = torch.empty(B, V, dtype=torch.float32, device=dev) if per_vertex else None
    if adjacency is not None:
        _chk("adjacency offsets", adjacency[0], _I32, V + 1)
        _chk("adjacency entries", adjacency[1], _I32, None)
    if transforms is not None:
        _chk("transforms", transforms, _F32, B, 4, 4)
        _require_device(transforms)
        transforms = transforms.contiguous()
    with torch.cuda.device(dev):
        _sync_deterministic()
        if l1_signs is not None:
            need = L.mr_shade_specular_backward_l1_workspace_bytes(B, V, T, W, H)
            entry, head = L.mr_shade_specular_backward_l1, (_ptr(l1_signs), _ptr(drgba))
        else:
            need = L.mr_shade_specular_backward_workspace_bytes(B, V, T, W, H)
            entry, head = L.mr_shade_specular_backward, (_ptr(drgba),)
        ws, have = _workspace(dev, need)
        rc = entry(
            *head, _ptr(ids), _ptr(bary), _ptr(clip), _ptr(normals), _ptr(positions),
            _ptr(diffuse), _ptr(specular), _ptr(triangles), _ptr(light_positions),
            _ptr(light_intensities), _ptr(ambient), _ptr(camera_position), _ptr(shininess),
            int(per_vertex), _ptr(norms2), B, V, T, W, H, nl, _ptr(dclip), _ptr(dn), _ptr(dp), _ptr(dd),
            _ptr(dsp), _ptr(dshin_v), _ptr(lg), _ptr(adjacency[0]) if adjacency is not None else None,
            _ptr(adjacency[1]) if adjacency is not None else None, _ptr(transforms),
            GBUFFER_NORMALISED if normalised_gbuffer else 0, int(grads_wanted), _ptr(ws), have, _stream(dev))
    _check(rc, "mr_shade_specular_backward_l1" if l1_signs is not None else "mr_shade_specular_backward")
    dlpos = lg[:, :3 * nl].reshape(B, nl, 3)
    dlint = lg[:, 3 * nl:6 * nl].reshape(B, nl, 3)
    damb = lg[:, 6 * nl:6 * nl + 3] if ambient is not None else None
    dcam = lg[:, 6 * nl + 3:6 * nl + 6]
    dshin = dshin_v if per_vertex else lg[:, 6 * nl + 6]
    return dclip, dn, dp, dd, dsp, dlpos, dlint, damb, dcam, dshin


def soft_max_lights():
    return int(lib().mr_soft_max_lights())


def soft_forward(clip, positions, normals, diffuse, triangles, light_positions, light_intensities,
                 width, height, sigma, gamma, blur, keep_prepared=False):
    """SoftRas forward: -> (rgba [B,H,W,4] with row 0 = top, aux [B,H,W,4] for the backward).

    keep_prepared=True runs in a workspace of its own and returns it as a third value: handed to
    soft_backward(prepared=...) for the same inputs, the per-triangle records and candidate lists the
    forward built there are not built again."""
    tensors = [clip, positions, normals, diffuse, triangles, light_positions, light_intensities]
    B, V, _ = _chk_mesh(clip, triangles)
    for name, t in (("positions", positions), ("normals", normals), ("diffuse colors", diffuse)):
        _chk(name, t, _F32, B, V, 3)
    _chk("light_positions", light_positions, _F32, B, None, 3)
    _chk("light_intensities", light_intensities, _F32, B, light_positions.shape[1])
    if not 1 <= light_positions.shape[1] <= soft_max_lights():
        raise ValueError("the soft rasterizer supports 1..%d lights" % soft_max_lights())
    dev = _require_device(*tensors)
    L = lib()
    clip, positions, normals, diffuse, triangles, light_positions, light_intensities = [
        t.contiguous() for t in tensors]
    B, V, _ = clip.shape
    T, nl = triangles.shape[0], light_positions.shape[1]
    rgba = torch.empty(B, height, width, 4, dtype=torch.float32, device=dev)
    aux = torch.empty(B, height, width, 4, dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        need = L.mr_soft_workspace_bytes(B, V, T, width, height)
        if keep_prepared:
            ws = torch.empty(max(int(L.mr_soft_prepared_bytes(B, V, T, width, height)), 256),
                             dtype=torch.uint8, device=dev)
            have = ws.numel()
        else:
            ws, have = _workspace(dev, need)
        rc = L.mr_soft_forward(_ptr(clip), _ptr(positions), _ptr(normals), _ptr(diffuse), _ptr(triangles),
                               _ptr(light_positions), _ptr(light_intensities), B, V, T, width, height, nl,
                               float(sigma), float(gamma), float(blur), _ptr(rgba), _ptr(aux), _ptr(ws),
                               have, _stream(dev))
    _check(rc, "mr_soft_forward")
    if keep_prepared:
        return rgba, aux, ws
    return rgba, aux


def soft_backward(drgba, rgba, aux, clip, positions, normals, diffuse, triangles, light_positions,
                  light_intensities, sigma, gamma, blur, prepared=None):
    """-> (dclip [B,V,4], dpositions, dnormals, ddiffuse [B,V,3], dlight_positions [B,L,3],
    dlight_intensities [B,L]).  prepared: the third value of soft_forward(keep_prepared=True) for the
    same inputs, or None."""
    tensors = [drgba, rgba, aux, clip, positions, normals, diffuse, triangles, light_positions,
               light_intensities]
    B, V, _ = _chk_mesh(clip, triangles)
    for name, t in (("positions", positions), ("normals", normals), ("diffuse colors", diffuse)):
        _chk(name, t, _F32, B, V, 3)
    _chk("light_positions", light_positions, _F32, B, None, 3)
    _chk("light_intensities", light_intensities, _F32, B, light_positions.shape[1])
    _chk("rgba", rgba, _F32, B, None, None, 4)
    _chk("aux", aux, _F32, B, rgba.shape[1], rgba.shape[2], 4)
    _chk("upstream gradient", drgba, _F32, B, rgba.shape[1], rgba.shape[2], 4)
    dev = _require_device(*tensors)
    L = lib()
    (drgba, rgba, aux, clip, positions, normals, diffuse, triangles, light_positions,
     light_intensities) = [t.contiguous() for t in tensors]
    B, V, _ = clip.shape
    T, nl = triangles.shape[0], light_positions.shape[1]
    _, H, W, _ = rgba.shape
    # one allocation, laid out back to back (dclip, dpositions, dnormals, ddiffuse, dlight_positions,
    # dlight_intensities): the library then zeroes all six with one memset instead of six launches
    n4, n3, nl3 = B * V * 4, B * V * 3, B * nl * 3
    flat = torch.empty(n4 + 3 * n3 + nl3 + B * nl, dtype=torch.float32, device=dev)
    dclip = flat[:n4].view(B, V, 4)
    dp = flat[n4:n4 + n3].view(B, V, 3)
    dn = flat[n4 + n3:n4 + 2 * n3].view(B, V, 3)
    dd = flat[n4 + 2 * n3:n4 + 3 * n3].view(B, V, 3)
    dlp = flat[n4 + 3 * n3:n4 + 3 * n3 + nl3].view(B, nl, 3)
    dli = flat[n4 + 3 * n3 + nl3:].view(B, nl)
    with torch.cuda.device(dev):
        _sync_deterministic()
        if prepared is not None and (prepared.device != dev or prepared.dtype != torch.uint8 or
                                     prepared.numel() < L.mr_soft_prepared_bytes(B, V, T, W, H)):
            raise ValueError("prepared must be the workspace soft_forward(keep_prepared=True) returned "
                             "for these sizes")
        need = L.mr_soft_workspace_bytes(B, V, T, W, H)
        ws, have = _workspace(dev, need)
        rc = L.mr_soft_backward(_ptr(drgba), _ptr(rgba), _ptr(aux), _ptr(clip), _ptr(positions),
                                _ptr(normals), _ptr(diffuse), _ptr(triangles), _ptr(light_positions),
                                _ptr(light_intensities), B, V, T, W, H, nl, float(sigma), float(gamma),
                                float(blur), _ptr(dclip), _ptr(dp), _ptr(dn), _ptr(dd), _ptr(dlp),
                                _ptr(dli), _ptr(prepared) if prepared is not None else None, _ptr(ws), have,
                                _stream(dev))
    _check(rc, "mr_soft_backward")
    return dclip, dp, dn, dd, dlp, dli


def image_empty_regions(image):
    """[B, ceil(H/64), ceil(W/64)] uint8 map of a [B,H,W,4] float32 device image: 1 = the 64 x 64 block (counted in
    G-buffer rows, i.e. from the image's LAST row up) is whole and all zeros.  What render_forward writes for its own
    image; l1_loss_forward skips blocks that are empty on both sides."""
    _chk("image", image, _F32, None, None, None, 4)
    dev = _require_device(image)
    image = image.contiguous()
    B, H, W = image.shape[:3]
    out = torch.empty(B, (H + 63) // 64, (W + 63) // 64, dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        rc = lib().mr_image_empty_regions(_ptr(image), B, H, W, _ptr(out), _stream(dev))
    _check(rc, "mr_image_empty_regions")
    return out


def l1_loss_forward(a, b, want_signs=True, empty_a=None, empty_b=None):
    """mean |a - b| over all elements -> (0-D tensor, packed signs or None), on the device.

    The signs ((n + 3) // 4 bytes, 2 bits per element) are all the backward pass needs.

    empty_a, empty_b (both or neither; [B,H,W,4] images only): the images' empty-block maps (render_forward's
    want_empty_regions / image_empty_regions): blocks empty on both sides are not read."""
    if a.dtype != _F32 or b.dtype != _F32:
        raise RuntimeError("l1_loss expects float32 tensors")
    if a.shape != b.shape:
        raise ValueError("image and target must have the same shape")
    dev = _require_device(a, b)
    a, b = a.contiguous(), b.contiguous()
    out = torch.empty((), dtype=torch.float32, device=dev)
    signs = torch.empty((a.numel() + 3) // 4, dtype=torch.uint8, device=dev) if want_signs else None
    partials = torch.empty(lib().mr_l1_loss_partials(), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        _arm_timer(TIMER_L1_FORWARD)
        if empty_a is not None and empty_b is not None:
            _chk("image", a, _F32, None, None, None, 4)
            B, H, W = a.shape[:3]
            for name, m in (("empty_a", empty_a), ("empty_b", empty_b)):
                _chk(name, m, _U8, B, (H + 63) // 64, (W + 63) // 64)
            rc = lib().mr_l1_loss_forward_regions(_ptr(a), _ptr(b), B, H, W, _ptr(empty_a.contiguous()),
                                                  _ptr(empty_b.contiguous()), _ptr(out),
                                                  _ptr(signs) if want_signs else None, _ptr(partials), _stream(dev))
        else:
            rc = lib().mr_l1_loss_forward(_ptr(a), _ptr(b), a.numel(), _ptr(out),
                                          _ptr(signs) if want_signs else None, _ptr(partials), _stream(dev))
    _check(rc, "mr_l1_loss_forward")
    return out, signs


def l1_loss_backward(signs, shape, upstream):
    """upstream * sign(a - b) / n as a float32 tensor of `shape`, from the packed signs."""
    _chk("signs", signs, _U8, None)
    _chk("upstream gradient of the loss", upstream, _F32, 1)
    dev = _require_device(signs, upstream)
    da = torch.empty(shape, dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        rc = lib().mr_l1_loss_backward(_ptr(signs), da.numel(), _ptr(upstream.contiguous()), _ptr(da),
                                       _stream(dev))
    _check(rc, "mr_l1_loss_backward")
    return da


def export_u8(image):
    """trunc(clamp(image, 0, 1) * 255) as a uint8 tensor of the same shape (device)."""
    dev = _require_device(image)
    image = image.contiguous()
    out = torch.empty(image.shape, dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        rc = lib().mr_export_u8(_ptr(image), image.numel(), _ptr(out), _stream(dev))
    _check(rc, "mr_export_u8")
    return out


def tone_map(image, gamma, as_uint8=False):
    """tone_mapper of the reference on the device: clamp(image ** gamma / per-image max, 0, 1) for a
    [B, ...] float32 image -> float32 tensor of the same shape, or uint8 frames (as_uint8)."""
    if not torch.is_tensor(image) or image.dim() < 1:
        raise ValueError("image must be a [batch, ...] tensor")
    _chk("image", image, _F32, *([None] * image.dim()))
    dev = _require_device(image)
    image = image.contiguous()
    B = image.shape[0]
    per_image = image.numel() // B if B else 0
    scratch = torch.empty(max(B, 1), dtype=torch.int32, device=dev)
    out = torch.empty(image.shape, dtype=torch.uint8 if as_uint8 else torch.float32, device=dev)
    with torch.cuda.device(dev):
        rc = lib().mr_tone_map(_ptr(image), B, per_image, float(gamma), _ptr(scratch),
                               None if as_uint8 else _ptr(out), _ptr(out) if as_uint8 else None, _stream(dev))
    _check(rc, "mr_tone_map")
    return out


def vertex_normals_forward(vertices, triangles, adjacency=None):
    """-> (normals [B,V,3], sums [B,V,3]); compute_vertex_normals of the reference as a per-vertex gather.
    adjacency: vertex_adjacency(triangles, V) if the caller already holds it."""
    _chk("vertices", vertices, _F32, None, None, 3)
    _chk("triangles", triangles, _I32, None, 3)
    dev = _require_device(vertices, triangles)
    vertices, triangles = vertices.contiguous(), triangles.contiguous()
    B, V, _ = vertices.shape
    offsets, entries = adjacency if adjacency is not None else vertex_adjacency(triangles, V)
    sums, normals = torch.empty_like(vertices), torch.empty_like(vertices)
    with torch.cuda.device(dev):
        rc = lib().mr_vertex_normals_forward(_ptr(vertices), _ptr(triangles), _ptr(offsets), _ptr(entries), B, V,
                                             triangles.shape[0], _ptr(sums), _ptr(normals), _stream(dev))
    _check(rc, "mr_vertex_normals_forward")
    return normals, sums


def vertex_normals_backward(dnormals, vertices, sums, triangles, adjacency=None):
    """-> dvertices [B,V,3]."""
    _chk("vertices", vertices, _F32, None, None, 3)
    B, V, _ = vertices.shape
    _chk("dnormals", dnormals, _F32, B, V, 3)
    _chk("sums", sums, _F32, B, V, 3)
    _chk("triangles", triangles, _I32, None, 3)
    dev = _require_device(dnormals, vertices, sums, triangles)
    dnormals, vertices, sums, triangles = [t.contiguous() for t in (dnormals, vertices, sums, triangles)]
    offsets, entries = adjacency if adjacency is not None else vertex_adjacency(triangles, V)
    dvertices = torch.empty_like(vertices)
    with torch.cuda.device(dev):
        rc = lib().mr_vertex_normals_backward(_ptr(dnormals), _ptr(vertices), _ptr(sums), _ptr(triangles),
                                              _ptr(offsets), _ptr(entries), B, V, triangles.shape[0],
                                              _ptr(dvertices), _stream(dev))
    _check(rc, "mr_vertex_normals_backward")
    return dvertices


def _chk_cameras(eye, center, up, fov_y, near_clip, far_clip):
    _chk("camera position", eye, _F32, None, 3)
    B = eye.shape[0]
    _chk("camera lookat", center, _F32, B, 3)
    _chk("camera up", up, _F32, B, 3)
    for name, t in (("fov_y", fov_y), ("near_clip", near_clip), ("far_clip", far_clip)):
        _chk(name, t, _F32, B)
    return B


def camera_transforms(eye, center, up, fov_y, near_clip, far_clip, aspect_ratio):
    """perspective . look_at per image as ONE launch -> (transforms [B,4,4], degenerate flags: a 1-element
    int32 device tensor, bit 0 = eye ~ center, bit 1 = up ~ gaze for some image)."""
    B = _chk_cameras(eye, center, up, fov_y, near_clip, far_clip)
    tensors = [t.contiguous() for t in (eye, center, up, fov_y, near_clip, far_clip)]
    dev = _require_device(*tensors)
    out = torch.empty(B, 4, 4, dtype=torch.float32, device=dev)
    flags = torch.empty(1, dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        rc = lib().mr_camera_transforms(*[_ptr(t) for t in tensors], float(aspect_ratio), B, _ptr(out), _ptr(flags),
                                        _stream(dev))
    _check(rc, "mr_camera_transforms")
    return out, flags


def camera_transforms_backward(dtransforms, eye, center, up, fov_y, near_clip, far_clip, aspect_ratio):
    """-> (deye, dcenter, dup) [B,3]."""
    B = _chk_cameras(eye, center, up, fov_y, near_clip, far_clip)
    _chk("gradient of the transforms", dtransforms, _F32, B, 4, 4)
    tensors = [t.contiguous() for t in (dtransforms, eye, center, up, fov_y, near_clip, far_clip)]
    dev = _require_device(*tensors)
    outs = [torch.empty(B, 3, dtype=torch.float32, device=dev) for _ in range(3)]
    with torch.cuda.device(dev):
        rc = lib().mr_camera_transforms_backward(*[_ptr(t) for t in tensors], float(aspect_ratio), B,
                                                 *[_ptr(t) for t in outs], _stream(dev))
    _check(rc, "mr_camera_transforms_backward")
    return tuple(outs)
