"""render() + L1 loss + backward at an arbitrary configuration (bench.py is fixed to configs[2]).

    python tools/step_bench.py --batch 8 --size 2048 --k 158      # configs[3] per-GPU shape (C4)
    python tools/step_bench.py --batch 8 --size 256 --k 50        # configs[1] (C2)
"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pytorch_mesh_renderer_amd import mesh_renderer
from pytorch_mesh_renderer_amd.common import synthetic

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--size", type=int, default=1024)
ap.add_argument("--k", type=int, default=50)
ap.add_argument("--iters", type=int, default=10)
args = ap.parse_args()
B, W, H = args.batch, args.size, args.size
dev = torch.device("cuda:0")
job = synthetic.sphere_job(B, W, H, args.k)
v = job["vertices"].to(dev).requires_grad_(True)
tri, n, kd = job["triangles"].to(dev), job["normals"].to(dev), job["diffuse"].to(dev)
eyes = job["eyes"]
lp, li = job["light_positions"].to(dev), job["light_intensities"].to(dev)
render = lambda: mesh_renderer.render(v, tri, n, kd, eyes, torch.zeros_like(eyes), torch.tensor([0.0, 1.0, 0.0]),
                                      lp, li, W, H)
with torch.no_grad():
    target = render().roll(3, 2)
def step():
    v.grad = None
    mesh_renderer.losses.l1_loss(render(), target).backward()
def fwd():
    with torch.no_grad():
        render()
for name, fn in (("fwd+loss+bwd", step), ("fwd only", fwd)):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(args.iters): fn()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / args.iters
    print(f"B={B} {W}x{H} T={tri.shape[0]} {name}: {dt*1e3:.3f} ms -> {B*W*H/dt/1e6:.0f} Mpix/s", flush=True)
