"""ctypes loader for oracle/libmr_oracle.so and the optional oracle/_ref build.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).
"""
import ctypes
import os
import subprocess
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libmr_oracle.so")
_REF_DIR = os.path.join(_HERE, "_ref")
_lib = None
_ref = None


def build(quiet=True):
    """Compile the C restatement (and oracle/_ref when /root/reference exists)."""
    out = subprocess.run(["make", "-C", _HERE, "all"], capture_output=True, text=True)
    if out.returncode != 0:
        raise RuntimeError("oracle build failed:\n" + out.stdout + out.stderr)
    if not quiet:
        print(out.stdout)


def _load():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        lib = ctypes.CDLL(_LIB_PATH)
        f32p = ctypes.POINTER(ctypes.c_float)
        i32p = ctypes.POINTER(ctypes.c_int32)
        ci = ctypes.c_int
        lib.mro_forward.argtypes = [f32p, i32p, ci, ci, ci, ci, ci, i32p, f32p, f32p, ci]
        lib.mro_forward.restype = ci
        lib.mro_backward.argtypes = [f32p, f32p, i32p, i32p, f32p, ci, ci, ci, ci, ci, f32p, ci]
        lib.mro_backward.restype = ci
        lib.mro_max_threads.restype = ci
        _lib = lib
    return _lib


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _p(a, ty):
    return a.ctypes.data_as(ctypes.POINTER(ty))


def max_threads():
    return int(_load().mro_max_threads())


def forward(clip, tris, width, height, threads=1):
    """clip [B,V,4] or [V,4] f32, tris [T,3] i32 -> (ids, bary, z) numpy arrays.

    A 2-D clip gives un-batched outputs, like the reference's forward()."""
    lib = _load()
    clip = _f32(clip)
    tris = _i32(tris)
    squeeze = clip.ndim == 2
    if squeeze:
        clip = clip[None]
    B, V, _ = clip.shape
    T = tris.shape[0]
    ids = np.empty((B, height, width), np.int32)
    bary = np.empty((B, height, width, 3), np.float32)
    z = np.empty((B, height, width), np.float32)
    rc = lib.mro_forward(_p(clip, ctypes.c_float), _p(tris, ctypes.c_int32), B, V, T,
                         width, height, _p(ids, ctypes.c_int32),
                         _p(bary, ctypes.c_float), _p(z, ctypes.c_float), threads)
    if rc != 0:
        raise ValueError("mro_forward rc=%d" % rc)
    if squeeze:
        return ids[0], bary[0], z[0]
    return ids, bary, z


def backward(dbary, clip, tris, ids, bary, threads=1):
    """Returns dclip with the shape of clip ([B,V,4] or [V,4])."""
    lib = _load()
    clip = _f32(clip)
    tris = _i32(tris)
    squeeze = clip.ndim == 2
    dbary, ids, bary = _f32(dbary), _i32(ids), _f32(bary)
    if squeeze:
        clip, dbary, ids, bary = clip[None], dbary[None], ids[None], bary[None]
    B, V, _ = clip.shape
    _, H, W = ids.shape
    assert dbary.shape == (B, H, W, 3) and bary.shape == (B, H, W, 3)
    dclip = np.empty((B, V, 4), np.float32)
    rc = lib.mro_backward(_p(dbary, ctypes.c_float), _p(clip, ctypes.c_float),
                          _p(tris, ctypes.c_int32), _p(ids, ctypes.c_int32),
                          _p(bary, ctypes.c_float), B, V, tris.shape[0], W, H,
                          _p(dclip, ctypes.c_float), threads)
    if rc != 0:
        raise ValueError("mro_backward rc=%d" % rc)
    return dclip[0] if squeeze else dclip


# --- the compiled reference kernel (oracle/_ref), when it was built ----------

def _load_ref():
    global _ref
    if _ref is None:
        so = os.path.join(_REF_DIR, "rasterize_triangles_cpp.so")
        if not os.path.exists(so):
            return None
        import torch  # noqa: F401  (the module links libtorch)
        if _REF_DIR not in sys.path:
            sys.path.insert(0, _REF_DIR)
        import rasterize_triangles_cpp
        _ref = rasterize_triangles_cpp
    return _ref


def have_reference_kernel():
    return _load_ref() is not None


def reference_forward(clip, tris, width, height):
    """Single image through the reference's own compiled kernel."""
    import torch
    ref = _load_ref()
    ids, bary, z = ref.forward(torch.from_numpy(_f32(clip)), torch.from_numpy(_i32(tris)),
                               width, height)
    return ids.numpy(), bary.detach().numpy(), z.numpy()


def reference_backward(dbary, clip, tris, ids, bary):
    import torch
    ref = _load_ref()
    (dv,) = ref.backward(torch.from_numpy(_f32(dbary)), torch.from_numpy(_f32(clip)),
                         torch.from_numpy(_i32(tris)), torch.from_numpy(_i32(ids)),
                         torch.from_numpy(_f32(bary)))
    return dv.numpy()
