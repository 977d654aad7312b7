"""Shading-kernel micro-benchmark (fused forward / backward), for profiling and A/B runs."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pytorch_mesh_renderer_amd import _native
from pytorch_mesh_renderer_amd.common import synthetic

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=10)
args = ap.parse_args()
B, W, H, K = 32, 1024, 1024, 50
dev = torch.device("cuda:0")
job = synthetic.sphere_job(B, W, H, K)
d = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in job.items()}
ids, bary, z = _native.rasterize_forward(d["clip"], d["triangles"], W, H)
def fwd():
    return _native.shade_forward(ids, bary, d["normals"], d["vertices"], d["diffuse"], d["triangles"],
                                 d["light_positions"], d["light_intensities"], None)
rgba = fwd()
g = torch.randn_like(rgba) / (H * W)
def bwd():
    return _native.shade_backward(g, ids, bary, d["clip"], d["normals"], d["vertices"], d["diffuse"],
                                  d["triangles"], d["light_positions"], d["light_intensities"], None)
for name, fn, nbytes in (("shade fwd", fwd, 32), ("shade bwd", bwd, 32)):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(args.iters): fn()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / args.iters
    print(f"{name}: {dt*1e3:.3f} ms  {B*H*W*nbytes/dt/1e9:.0f} GB/s", flush=True)
