"""render() + L1 loss + backward at one shape: python tools/shape_one.py B W H K [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pytorch_mesh_renderer_amd import mesh_renderer
from pytorch_mesh_renderer_amd.common import synthetic
B, W, H, K = [int(a) for a in sys.argv[1:5]]
n = int(sys.argv[5]) if len(sys.argv) > 5 else 20
dev = torch.device("cuda:0")
job = synthetic.sphere_job(B, W, H, K)
v = job["vertices"].to(dev).requires_grad_(True)
d = {k: (t.to(dev) if torch.is_tensor(t) else t) for k, t in job.items()}
target = torch.rand(B, H, W, 4, device=dev)
def step():
    img = mesh_renderer.render(v, d["triangles"], d["normals"], d["diffuse"], job["eyes"], torch.zeros(B, 3),
                               torch.tensor([0.0, 1.0, 0.0]), d["light_positions"], d["light_intensities"], W, H)
    loss = mesh_renderer.losses.l1_loss(img, target)
    v.grad = None
    loss.backward()
for _ in range(3): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(n): step()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
print("B=%d %dx%d T=%d: %.3f ms/step  %.0f Mpix/s" % (B, W, H, job["triangles"].shape[0], dt * 1e3, B * W * H / dt / 1e6))
