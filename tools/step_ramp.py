# per-step GPU time over the first steps after process start (clock ramp?)
import sys, time, torch
sys.path.insert(0, ".")
import bench
from pytorch_mesh_renderer_amd.common import synthetic
_, B, W, H, K = bench.CONFIGS["c3"]
dev = torch.device("cuda:0")
step, vertices, state = bench.make_step(synthetic.sphere_job(B, W, H, K), dev, None)
torch.cuda.synchronize()
time.sleep(1.0)
n = 400
ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
ev[0].record()
for i in range(n):
    step()
    ev[i + 1].record()
torch.cuda.synchronize()
ms = [ev[i].elapsed_time(ev[i + 1]) for i in range(n)]
for a in range(0, n, 20):
    print("steps %3d-%3d: mean %.4f ms  min %.4f" % (a, a + 19, sum(ms[a:a + 20]) / 20, min(ms[a:a + 20])))
