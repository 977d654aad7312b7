"""Where does the HOST time of a small render() go?  (cProfile, tiny image so the GPU is idle.)"""
import cProfile, os, pstats, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pytorch_mesh_renderer_amd import mesh_renderer
from pytorch_mesh_renderer_amd.common import synthetic

B, W, H = 1, 64, 64
dev = torch.device("cuda:0")
job = synthetic.sphere_job(B, W, H, 6)
v = job["vertices"].to(dev).requires_grad_(True)
tri, n, kd = job["triangles"].to(dev), job["normals"].to(dev), job["diffuse"].to(dev)
eyes = job["eyes"]
lp, li = job["light_positions"].to(dev), job["light_intensities"].to(dev)
center, up = torch.zeros_like(eyes), torch.tensor([0.0, 1.0, 0.0])
with torch.no_grad():
    target = mesh_renderer.render(v, tri, n, kd, eyes, center, up, lp, li, W, H).roll(3, 2)
def step():
    v.grad = None
    mesh_renderer.losses.l1_loss(mesh_renderer.render(v, tri, n, kd, eyes, center, up, lp, li, W, H), target).backward()
for _ in range(20): step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(300): step()
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(28)
