"""Forward-rasterizer micro-benchmark for profiling (rocprofv3 --pmc / --kernel-trace).

    python tools/raster_bench.py [--config c3|c2|c4] [--iters N] [--variant V] [--backward]
"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if "--variant" in sys.argv:   # stage probes live in the -DMR_PROBES build only (make -C csrc probes)
    os.environ.setdefault("MR_NATIVE_LIB_PATH", os.path.join(
        ROOT, "pytorch_mesh_renderer_amd", "csrc", "libmesh_raster_hip_probes.so"))
import torch
from pytorch_mesh_renderer_amd import _native
from pytorch_mesh_renderer_amd.common import synthetic

ap = argparse.ArgumentParser()
ap.add_argument("--config", default="c3")
ap.add_argument("--iters", type=int, default=40)
ap.add_argument("--variant", type=int, default=0)
ap.add_argument("--backward", action="store_true")
ap.add_argument("--edge", type=int, default=0, help="force 32 / 64 pixel regions")
args = ap.parse_args()
B, W, H, K = {"c2": (8, 256, 256, 50), "c3": (32, 1024, 1024, 50), "c4": (8, 2048, 2048, 158),
              "c3s": (4, 1024, 1024, 50),
              # configs[3]'s pixel count on row strides that are not powers of two (HBM channel spread of the stores)
              "c4a": (8, 2176, 1928, 158), "c4b": (8, 2080, 2016, 158), "c3a": (32, 1088, 964, 50)}[args.config]
dev = torch.device("cuda:0")
job = synthetic.sphere_job(B, W, H, K)
clip, tris = job["clip"].to(dev), job["triangles"].to(dev)
assert _native.lib().mr_debug_set_raster_probe(args.variant) == 0, "unknown probe (see include/mesh_raster_debug.h)"
assert _native.lib().mr_debug_set_raster_region_edge(args.edge) == 0
for _ in range(3):
    ids, bary, z = _native.rasterize_forward(clip, tris, W, H)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(args.iters):
    ids, bary, z = _native.rasterize_forward(clip, tris, W, H)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / args.iters
print(f"{args.config} variant={args.variant}: fwd {dt*1e3:.3f} ms  {B*H*W*20/dt/1e9:.0f} GB/s  covered={float((bary.sum(-1)>0.5).float().mean()):.3f}")
if args.backward:
    g = torch.randn(bary.shape, device=dev) / (H * W)
    for _ in range(3):
        _native.rasterize_backward(g, clip, tris, ids, bary)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.iters):
        _native.rasterize_backward(g, clip, tris, ids, bary)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.iters
    print(f"   bwd {dt*1e3:.3f} ms  {B*H*W*28/dt/1e9:.0f} GB/s")
