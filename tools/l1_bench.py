"""L1 loss kernels alone (forward with sign codes, backward), 32 x 1024^2 x 4 floats."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pytorch_mesh_renderer_amd import _native
dev = torch.device("cuda:0")
a, b = torch.rand(32, 1024, 1024, 4, device=dev), torch.rand(32, 1024, 1024, 4, device=dev)
up = torch.ones(1, device=dev)
loss, signs = _native.l1_loss_forward(a, b)
def fwd(): return _native.l1_loss_forward(a, b)
def bwd(): return _native.l1_loss_backward(signs, a.shape, up)
def u8(): return _native.export_u8(a)
for name, fn, nbytes in (("l1 fwd", fwd, 33), ("l1 bwd", bwd, 17), ("export u8", u8, 20)):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(30): fn()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 30
    print(f"{name}: {dt*1e3:.3f} ms  {a.numel()//4*nbytes/dt/1e9:.0f} GB/s", flush=True)
