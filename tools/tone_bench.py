"""mesh_renderer.tone_mapper / tone_mapper_uint8 on a 32 x 1024^2 x 4 image (DESIGN section 1 quotes the time)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pytorch_mesh_renderer_amd import mesh_renderer

dev = torch.device("cuda:0")
img = torch.rand(32, 1024, 1024, 4, device=dev) * 3.0
for name, fn in (("tone_mapper", lambda: mesh_renderer.tone_mapper(img, 2.2)),
                 ("tone_mapper_uint8", lambda: mesh_renderer.tone_mapper_uint8(img, 2.2))):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 50
    for _ in range(n): fn()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    print(f"{name} 32x1024x1024x4 gamma=2.2: {dt*1e3:.3f} ms  ({img.numel()*4/dt/1e9:.0f} GB/s of input)")
