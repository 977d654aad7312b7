"""Per-step GPU time of the benchmarked step from an idle chip (events around every step): the clock ramp a short
timed region sits on.   python tools/step_series.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from pytorch_mesh_renderer_amd.common import synthetic
dev = torch.device("cuda:0")
job = synthetic.sphere_job(32, 1024, 1024, 50)
step, v, st = bench.make_step(job, dev, None)
torch.cuda.synchronize()
time.sleep(0.5)      # idle chip, like a fresh process after its setup
n = 600
ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
ev[0].record()
for i in range(n):
    step()
    ev[i + 1].record()
torch.cuda.synchronize()
ms = [ev[i].elapsed_time(ev[i + 1]) for i in range(n)]
print("per-step ms, every 10th window mean:", " ".join("%.3f" % (sum(ms[i:i+10])/10) for i in range(0, n, 10)))
print("mean of steps 100-600 %.4f" % (sum(ms[100:]) / 500))
