"""BASELINE config 5 timing: soft_mesh_renderer.render, 5k-tri sphere, 512x512, batch 16, fwd+bwd."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pytorch_mesh_renderer_amd import soft_mesh_renderer
from pytorch_mesh_renderer_amd.common import synthetic

B, W, H = 16, 512, 512
dev = torch.device("cuda:0")
job = synthetic.sphere_job(B, W, H, 50)
v = job["vertices"].to(dev).requires_grad_(True)
tri, kd = job["triangles"].to(dev), job["diffuse"].to(dev)
eyes, lp = job["eyes"], job["light_positions"].to(dev)   # cameras stay host tensors, as in the reference's usage (and bench.py)
li = torch.ones(B, 1, device=dev)
zero, up = torch.zeros(B, 3), torch.tensor([0.0, 1.0, 0.0])
def step():
    v.grad = None
    img = soft_mesh_renderer.render(v, tri, kd, eyes, zero, up, lp, li, W, H)
    img.mean().backward()
for _ in range(10 if len(sys.argv) < 2 else 1): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
for _ in range(n): step()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
print(f"config5 soft render fwd+bwd: {dt*1e3:.2f} ms/step -> {B*W*H/dt/1e6:.1f} Mpix/s (reference CPU: ~165 px/s)")
