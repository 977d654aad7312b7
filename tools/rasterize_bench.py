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
upstream = torch.randn(B, H, W, A, device=dev) / (B * H * W * A)
def step(loss):
    v.grad = None; attrs.grad = None
    out = mesh_renderer.rasterize(v, attrs, tri, proj, W, H, bg)
    if loss == "mean":      # the figure of rounds 3-4: ~0.49 ms of it at A = 9 are torch's mean() and its backward
        out.mean().backward()
    else:                   # a given upstream gradient: the rasterizer's kernels and nothing else
        out.backward(gradient=upstream)
def forward_only():
    with torch.no_grad():
        mesh_renderer.rasterize(v, attrs, tri, proj, W, H, bg)
for name, fn in (("fwd+bwd, mean() loss", lambda: step("mean")), ("fwd+bwd, given upstream", lambda: step("given")),
                 ("forward only", forward_only)):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 10
    for _ in range(n): fn()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    print(f"rasterize A={A} B={B} {W}x{H} {name}: {dt*1e3:.3f} ms -> {B*W*H/dt/1e6:.0f} Mpix/s", flush=True)
