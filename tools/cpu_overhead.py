"""How long the host needs to ENQUEUE one bench step (no synchronisation): if this approaches the GPU's
ms_per_step the loop stops being GPU-bound.   python tools/cpu_overhead.py [c3|c4]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from pytorch_mesh_renderer_amd.common import synthetic

cfg = sys.argv[1] if len(sys.argv) > 1 else "c3"
_, batch, width, height, k = bench.CONFIGS[cfg]
dev = torch.device("cuda:0")
job = synthetic.sphere_job(batch, width, height, k)
step, vertices, _ = bench.make_step(job, dev, None)
for _ in range(20): step()
torch.cuda.synchronize()
n = 200
t0 = time.perf_counter()
for _ in range(n): step()
t_enqueue = (time.perf_counter() - t0) / n
torch.cuda.synchronize()
t_total = (time.perf_counter() - t0) / n
print(f"{cfg}: host enqueues a step in {t_enqueue*1e3:.3f} ms; GPU-paced step {t_total*1e3:.3f} ms "
      f"({'GPU' if t_enqueue < 0.9 * t_total else 'HOST'}-bound)")
