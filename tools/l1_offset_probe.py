"""Does the loss kernel's time depend on WHERE its two operands lie relative to each other?

Round 6 saw `k_l1_forward_regions` at 0.160 ms in two of five process runs on one box and at 0.184-0.185 in the other three
(gpurun_out/ab_sched.log: same kernel, same data, different processes).  Its two 537 MB streams are read at the same offset at
the same time; if the spread is a matter of DRAM banks, it follows the distance between the operands.  This probe carves both
operands out of ONE allocation at a swept distance and times mr_l1_loss_forward on them.

    python tools/l1_offset_probe.py
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from pytorch_mesh_renderer_amd import _native  # noqa: E402

dev = torch.device("cuda:0")
B, H, W = 32, 1024, 1024
n = B * H * W * 4                      # floats per operand
pads = [0, 64, 256, 1024, 4096, 16 << 10, 64 << 10, 256 << 10, 1 << 20, 2 << 20, 3 << 20, 4 << 20, 6 << 20, 8 << 20, 16 << 20,
        32 << 20, 48 << 20, 64 << 20, 96 << 20, 128 << 20]
pool = torch.empty(2 * n + (max(pads) // 4) + 1024, dtype=torch.float32, device=dev)
pool.uniform_(0.0, 1.0)
print("pool at 0x%x (%.2f GB)" % (pool.data_ptr(), pool.numel() * 4 / 2 ** 30), flush=True)


def timed(a, b, reps=30):
    for _ in range(5):
        _native.l1_loss_forward(a, b)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        _native.l1_loss_forward(a, b)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


a = pool[:n].view(B, H, W, 4)
for pad in pads:
    off = n + pad // 4
    b = pool[off:off + n].view(B, H, W, 4)
    print("operands %11d bytes apart (size + %9d): %.4f ms per call (loss + finish kernels)" % (4 * off, pad, timed(a, b)), flush=True)
# and as torch hands them out: two separate allocations
x, y = torch.rand(B, H, W, 4, device=dev), torch.rand(B, H, W, 4, device=dev)
print("two torch allocations, 0x%x and 0x%x (%d bytes apart): %.4f ms" % (x.data_ptr(), y.data_ptr(), y.data_ptr() - x.data_ptr(), timed(x, y)))
