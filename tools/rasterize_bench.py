"""mesh_renderer.rasterize() (generic attribute interpolation, rasterize.py:27-152) forward + backward."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pytorch_mesh_renderer_amd import mesh_renderer
from pytorch_mesh_renderer_amd.common import synthetic

ap = argparse.ArgumentParser()
ap.add_argument("--attrs", type=int, default=9)
ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--size", type=int, default=1024)
args = ap.parse_args()
B, W, H, A = args.batch, args.size, args.size, args.attrs
dev = torch.device("cuda:0")
job = synthetic.sphere_job(B, W, H, 50)
v = job["vertices"].to(dev).requires_grad_(True)
attrs = torch.rand(B, v.shape[1], A, device=dev, requires_grad=True)
tri = job["triangles"].to(dev)
proj = synthetic.clip_transforms(job["eyes"], W, H).to(dev)
bg = torch.full((A,), -1.0, device=dev)
def step():
    v.grad = None; attrs.grad = None
    out = mesh_renderer.rasterize(v, attrs, tri, proj, W, H, bg)
    out.mean().backward()
for _ in range(3): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
n = 5
for _ in range(n): step()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
print(f"rasterize A={A} B={B} {W}x{H}: fwd+bwd {dt*1e3:.2f} ms -> {B*W*H/dt/1e6:.0f} Mpix/s", flush=True)
