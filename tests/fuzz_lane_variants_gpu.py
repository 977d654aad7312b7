"""Fuzz of the other two lane-accumulating backwards on random soups (GPU), each against its rows-kernel form:
  * rasterize()'s one-pass backward, mr_interpolate_raster_backward: normalised G-buffer (AttrFoldLaneFn) vs not (rows);
  * the specular shading backward: vertex gradients only (SpecFoldLaneFn, 18 sums; folded, 9) vs everything (rows).
Looks for races and bookkeeping errors (order-one deviations); sliver triangles make the two forms differ by up to
a few 1e-3 of an output's largest element (see tests/fuzz_shade_backward_gpu.py).

    python tests/fuzz_lane_variants_gpu.py [--trials N] [--seed S]
"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from pytorch_mesh_renderer_amd import _native

ap = argparse.ArgumentParser()
ap.add_argument("--trials", type=int, default=200)
ap.add_argument("--seed", type=int, default=1)
args = ap.parse_args()
rng = np.random.default_rng(args.seed)
dev = torch.device("cuda:0")
bad, worst = 0, 0.0
t0 = time.time()


def close(name, got, want, what):
    global bad, worst
    scale = max(float(want.abs().max()), 1e-30)
    dev_ = float((got - want).abs().max()) / scale
    worst = max(worst, dev_)
    # a sliver triangle (tiny |det|) makes the two forms of the rasterizer's backward differ by a few percent of the
    # largest element at its three vertices, in either direction against a float64 evaluation (see
    # fuzz_shade_backward_gpu.py); a race or a bookkeeping error touches many elements or is of order one
    off = ~torch.isclose(got, want, rtol=1e-3, atol=5e-3 * scale)
    if int(off.sum()) > 12 or dev_ > 0.2:
        bad += 1
        print(f"MISMATCH {what}: {name} deviates by {dev_:.3e} of {scale:.3e}", flush=True)


for trial in range(args.trials):
    B = int(rng.integers(1, 4))
    V = int(rng.integers(4, 250))
    T = int(rng.integers(1, 2000 if trial % 4 == 3 else 350))
    W, H = int(rng.integers(8, 400)), int(rng.integers(8, 280))
    pos = (rng.normal(size=(B, V, 3)) * [1.0, 1.0, 0.3] * (0.2 if trial % 4 == 1 else 1.0)).astype(np.float32)
    xf = np.tile(np.eye(4, dtype=np.float32), (B, 1, 1))
    xf[:, 3, 2] = 0.5
    xf[:, 3, 3] = 1.2
    tris = rng.integers(0, V, size=(T, 3)).astype(np.int32)
    t = lambda a: torch.from_numpy(a).to(dev)
    pos_d, xf_d, tris_d = t(pos), t(xf), t(tris)
    clip = _native.vertex_transform(pos_d, xf_d)
    ids, bary, _ = _native.rasterize_forward(clip, tris_d, W, H)
    adjacency = _native.vertex_adjacency(tris_d, V)
    what = f"trial {trial} B={B} V={V} T={T} {W}x{H}"
    # --- rasterize(): attributes
    A = int(rng.integers(1, 13))
    attrs = t(rng.normal(size=(B, V, A)).astype(np.float32))
    bg = t(rng.normal(size=(A,)).astype(np.float32))
    dout = t(rng.normal(size=(B, H, W, A)).astype(np.float32)) / (H * W)
    _, records = _native.interpolate_forward_records(ids, bary, attrs, tris_d, bg)
    rows = _native.interpolate_raster_backward(dout, ids, bary, clip, attrs, tris_d, bg, adjacency, corner_records=records)
    lanes = _native.interpolate_raster_backward(dout, ids, bary, clip, attrs, tris_d, bg, adjacency, corner_records=records,
                                                normalised_gbuffer=True)
    close("d attributes", lanes[0], rows[0], what + f" A={A} rasterize")
    close("d clip", lanes[1], rows[1], what + f" A={A} rasterize")
    # --- specular shading
    L = int(rng.integers(1, 5))
    nrm, kd, ks = (t(rng.normal(size=(B, V, 3)).astype(np.float32)), t(rng.random(size=(B, V, 3)).astype(np.float32)),
                   t(rng.random(size=(B, V, 3)).astype(np.float32)))
    lp = t((rng.normal(size=(B, L, 3)) * 3.0 + [0.0, 0.0, 4.0]).astype(np.float32))
    li = t((rng.random(size=(B, L, 3)) + 0.1).astype(np.float32))
    amb = t((rng.random(size=(B, 3)) * 0.3).astype(np.float32)) if trial % 2 else None
    cam = t((rng.normal(size=(B, 3)) + [0.0, 0.0, 5.0]).astype(np.float32))
    # (exponents above 1: below, d pow / d base is unbounded at base -> 0+ and a pixel whose reflection . camera product
    #  rounds to either side of zero moves the gradient by its whole, arbitrarily large, contribution -- in any evaluation)
    shin = t((1.2 + 2.0 * rng.random(size=(B, V))).astype(np.float32)) if trial % 3 == 0 else t((1.2 + 3.0 * rng.random(size=(B,))).astype(np.float32))
    rgba, norms2 = _native.shade_specular_forward(ids, bary, nrm, pos_d, kd, ks, tris_d, lp, li, amb, cam, shin)
    g = t(rng.normal(size=(B, H, W, 4)).astype(np.float32)) / (H * W)
    sargs = (g, ids, bary, clip, nrm, pos_d, kd, ks, tris_d, lp, li, amb, cam, shin, norms2)
    rows = _native.shade_specular_backward(*sargs, adjacency=adjacency)
    lanes = _native.shade_specular_backward(*sargs, adjacency=adjacency, normalised_gbuffer=True,
                                            grads_wanted=_native.GRAD_POSITIONS | _native.GRAD_CLIP)
    folded = _native.shade_specular_backward(*sargs, adjacency=adjacency, normalised_gbuffer=True, transforms=xf_d,
                                             grads_wanted=_native.GRAD_POSITIONS)
    if float(rows[0].abs().max()) > 0:
        close("d clip", lanes[0], rows[0], what + f" L={L} specular")
        close("d positions", lanes[2], rows[2], what + f" L={L} specular")
        whole = rows[2] + torch.einsum("bij,bvi->bvj", xf_d[:, :, :3], rows[0])
        close("whole vertex gradient", folded[2], whole, what + f" L={L} specular, folded")
    if trial % 40 == 39:
        print(f"{trial + 1} trials, {bad} mismatches, {time.time() - t0:.0f} s", flush=True)
print("FUZZ", "FAILED" if bad else "OK", f"{args.trials} trials, {bad} mismatches; largest deviation {worst:.2e} of an output's largest element")
sys.exit(1 if bad else 0)
