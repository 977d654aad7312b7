"""Where the host's time goes in one eager optimisation step on a tiny scene (launch-bound: the GPU work is
microseconds).   python tools/host_profile.py [--cameras host|device]"""
import argparse, cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pytorch_mesh_renderer_amd import mesh_renderer
from pytorch_mesh_renderer_amd.common import synthetic

ap = argparse.ArgumentParser()
ap.add_argument("--cameras", default="host")
ap.add_argument("--size", type=int, default=64)
ap.add_argument("--batch", type=int, default=1)
ap.add_argument("--no-camera-cache", action="store_true")
args = ap.parse_args()
dev = torch.device("cuda:0")
if args.no_camera_cache:
    from pytorch_mesh_renderer_amd.common import camera_utils
    camera_utils.CACHE_HOST_CAMERAS = False
job = synthetic.sphere_job(args.batch, args.size, args.size, 6)
d = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in job.items()}
vertices = d["vertices"].clone().requires_grad_(True)
eyes = job["eyes"] if args.cameras == "host" else d["eyes"]
center, up = torch.zeros_like(eyes), torch.tensor([0.0, 1.0, 0.0], device=eyes.device)
with torch.no_grad():
    target = mesh_renderer.render(vertices, d["triangles"], d["normals"], d["diffuse"], eyes, center, up,
                                  d["light_positions"], d["light_intensities"], args.size, args.size).roll(2, 2).contiguous()
def step():
    vertices.grad = None
    img = mesh_renderer.render(vertices, d["triangles"], d["normals"], d["diffuse"], eyes, center, up,
                               d["light_positions"], d["light_intensities"], args.size, args.size)
    loss = mesh_renderer.losses.l1_loss(img, target)
    loss.backward()
for _ in range(50): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(500): step()
torch.cuda.synchronize(); print("eager step: %.3f ms" % ((time.perf_counter() - t0) / 500 * 1e3))
pr = cProfile.Profile(); pr.enable()
for _ in range(500): step()
torch.cuda.synchronize(); pr.disable()
st = pstats.Stats(pr); st.sort_stats("cumulative").print_stats(28)
