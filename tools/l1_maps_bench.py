# the loss pass: flat kernel vs the row-walk kernel with nothing to skip vs with the benchmark's maps
import sys, torch
sys.path.insert(0, ".")
import bench
from pytorch_mesh_renderer_amd import _native
from pytorch_mesh_renderer_amd.common import synthetic
_, B, W, H, K = bench.CONFIGS["c3"]
dev = torch.device("cuda:0")
step, vertices, state = bench.make_step(synthetic.sphere_job(B, W, H, K), dev, None)
step()
a, b = state["image"].detach(), state["target"]
ma, mb = _native.image_empty_regions(a), _native.image_empty_regions(b)
zero = torch.zeros_like(ma)
def t(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
print("flat                    %.4f ms" % t(lambda: _native.l1_loss_forward(a, b)))
print("rows, nothing skipped   %.4f ms" % t(lambda: _native.l1_loss_forward(a, b, empty_a=zero, empty_b=zero)))
print("rows, maps (%.3f empty)  %.4f ms" % (float((ma & mb).float().mean()), t(lambda: _native.l1_loss_forward(a, b, empty_a=ma, empty_b=mb))))
