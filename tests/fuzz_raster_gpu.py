"""One-off heavy fuzz of the forward rasterizer against the oracle (not collected by pytest).

    python tests/fuzz_raster_gpu.py [--trials 300] [--seed 1]

Random soups with every mix the reference's arithmetic branches on: vertices behind the eye,
zero / negative / huge w, coincident and collinear vertices, duplicated triangles (exact z ties),
NaN / Inf coordinates, tiny and huge triangles, images from 1x1 to 700x500, up to 6000 triangles
(more than one LDS bin per region).  Every output must be bit-identical."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import oracle
from pytorch_mesh_renderer_amd import _native

ap = argparse.ArgumentParser()
ap.add_argument("--trials", type=int, default=300)
ap.add_argument("--seed", type=int, default=1)
ap.add_argument("--edge", type=int, default=0, help="force 32 / 64 pixel regions (0 = automatic)")
args = ap.parse_args()
rng = np.random.default_rng(args.seed)
dev = torch.device("cuda:0")
assert _native.lib().mr_debug_set_raster_region_edge(args.edge) == 0
bad = 0
nan_only = 0   # trials whose only difference is the bit pattern of a NaN present on both sides
t0 = time.time()
for trial in range(args.trials):
    kind = trial % 6
    V = int(rng.integers(3, 400))
    T = int(rng.integers(1, 6000 if kind == 5 else 600))
    W, H = int(rng.integers(1, 700)), int(rng.integers(1, 500))
    B = int(rng.integers(1, 4))
    scale = float(rng.choice([0.05, 0.3, 1.0, 3.0]))
    clip = (rng.normal(size=(B, V, 4)) * [scale, scale, 1.0, 1.0]).astype(np.float32)
    if kind in (0, 5):
        clip[..., 3] = np.abs(clip[..., 3]) + 0.05              # everything in front of the eye
    elif kind == 1:
        clip[..., 3] = rng.choice([1.0, -1.0, 0.0, 1e-30, 1e30], size=(B, V)).astype(np.float32)
    elif kind == 2:
        src = clip[:, 1::3]
        clip[:, 0:3 * src.shape[1]:3] = src                        # coincident vertices
    elif kind == 3:
        flat = clip.reshape(-1)
        idx = rng.integers(0, flat.size, size=max(1, flat.size // 50))
        flat[idx] = rng.choice([np.nan, np.inf, -np.inf, 1e38, -1e38], size=idx.size).astype(np.float32)
    tris = rng.integers(0, V, size=(T, 3)).astype(np.int32)
    if T > 4:
        tris[T // 2] = tris[0]
        tris[T - 1] = tris[1][::-1]
    want = oracle.forward(clip, tris, W, H, threads=8)
    ids, bary, z = _native.rasterize_forward(torch.from_numpy(clip).to(dev), torch.from_numpy(tris).to(dev), W, H)
    got = (ids.cpu().numpy(), bary.cpu().numpy(), z.cpu().numpy())
    ok = (np.array_equal(got[0], want[0]) and np.array_equal(got[1].view(np.uint32), want[1].view(np.uint32))
          and np.array_equal(got[2].view(np.uint32), want[2].view(np.uint32)))
    if not ok:
        bad += 1
        dz = got[2].view(np.uint32) != want[2].view(np.uint32)
        db = (got[1].view(np.uint32) != want[1].view(np.uint32))
        both_nan_z = np.isnan(got[2]) & np.isnan(want[2])
        both_nan_b = np.isnan(got[1]) & np.isnan(want[1])
        real = int((dz & ~both_nan_z).sum() + (db & ~both_nan_b).sum() + (got[0] != want[0]).sum())
        if real == 0:
            nan_only += 1
            bad -= 1
        sample = np.argwhere(dz)[:1]
        ex = "" if not len(sample) else " e.g. got %08x want %08x" % (
            got[2].view(np.uint32)[tuple(sample[0])], want[2].view(np.uint32)[tuple(sample[0])])
        print(f"{'nan-bits' if real == 0 else 'MISMATCH'} trial {trial} kind {kind} B={B} V={V} T={T} {W}x{H}: "
              f"ids {(got[0] != want[0]).sum()} z {dz.sum()} bary {db.sum()} non-NaN differences {real}{ex}", flush=True)
    if trial % 50 == 49:
        print(f"{trial + 1} trials, {bad} mismatches, {time.time() - t0:.0f} s", flush=True)
print("FUZZ", "FAILED" if bad else "OK", f"{args.trials} trials, {bad} mismatches, {nan_only} with NaN-encoding differences only")
sys.exit(1 if bad else 0)
