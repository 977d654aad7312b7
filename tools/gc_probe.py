"""Is the rare 3-4 x reading of the SoftRas leg (BASELINE configs[4]) the host's garbage collector?

profiles/r06_a_bench.json: five chunks of 20 steps read 0.788 / 0.787 / 2.959 / 0.781 / 0.779 ms per step, and the slow one
spent 2.67 ms per step ENQUEUEING on the host: ~44 ms of host stall inside one 16-ms chunk.  This probe runs the same step
in chunks of 20 with a gc callback that times every collection, first after bench.py's earlier legs have filled the heap the
way they do there (--heap 1: a configs[3]-shaped job and its step), and prints every chunk above 1.2 ms next to the
collections that ran inside it; then the same with the collector frozen / disabled around the chunks.

    python tools/gc_probe.py [--chunks 60]
"""
import argparse
import gc
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from pytorch_mesh_renderer_amd import mesh_renderer, soft_mesh_renderer  # noqa: E402
from pytorch_mesh_renderer_amd.common import synthetic  # noqa: E402
from pytorch_mesh_renderer_amd.mesh_renderer import losses  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--chunks", type=int, default=60)
ap.add_argument("--heap", type=int, default=1)
args = ap.parse_args()
dev = torch.device("cuda:0")
events = []
_t0 = {}


def on_gc(phase, info):
    if phase == "start":
        _t0["t"] = time.perf_counter()
    else:
        events.append((time.perf_counter(), info["generation"], (time.perf_counter() - _t0["t"]) * 1e3, info["collected"]))


gc.callbacks.append(on_gc)
if args.heap:
    j4 = synthetic.sphere_job(8, 2048, 2048, 158)
    v4 = j4["vertices"].to(dev).requires_grad_(True)
    t4 = torch.rand(8, 2048, 2048, 4, device=dev)
    for _ in range(20):
        v4.grad = None
        img = mesh_renderer.render(v4, j4["triangles"].to(dev), j4["normals"].to(dev), j4["diffuse"].to(dev), j4["eyes"],
                                   torch.zeros_like(j4["eyes"]), torch.tensor([0.0, 1.0, 0.0]), j4["light_positions"].to(dev),
                                   j4["light_intensities"].to(dev), 2048, 2048)
        losses.l1_loss(img, t4).backward()
    del t4, img
j5 = synthetic.sphere_job(16, 512, 512, 50)
v5 = j5["vertices"].to(dev).requires_grad_(True)
tri5, kd5, lp5 = j5["triangles"].to(dev), j5["diffuse"].to(dev), j5["light_positions"].to(dev)
eyes5, zero5, up5 = j5["eyes"], torch.zeros(16, 3), torch.tensor([0.0, 1.0, 0.0])
li5 = torch.ones(16, 1, device=dev)


def step5():
    v5.grad = None
    soft_mesh_renderer.render(v5, tri5, kd5, eyes5, zero5, up5, lp5, li5, 512, 512).mean().backward()


def series(tag, chunks):
    for _ in range(24):
        step5()
    torch.cuda.synchronize()
    slow = 0
    walls = []
    for c in range(chunks):
        n_before = len(events)
        t0 = time.perf_counter()
        for _ in range(20):
            step5()
        t_host = time.perf_counter() - t0
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / 20 * 1e3
        walls.append(wall)
        inside = [(g, round(ms, 2), n) for (_, g, ms, n) in events[n_before:]]
        if wall > 1.2:
            slow += 1
            print("  %s chunk %3d: %.3f ms/step (host enqueue %.3f); collections inside (generation, ms, collected): %s" % (
                tag, c, wall, t_host / 20 * 1e3, inside), flush=True)
    walls.sort()
    print("%s: %d chunks of 20 steps, median %.3f ms/step, max %.3f, %d above 1.2 ms; objects tracked by the collector: %d" % (
        tag, chunks, walls[len(walls) // 2], walls[-1], slow, len(gc.get_objects())), flush=True)


series("collector on ", args.chunks)
gen2 = [(round(ms, 1), n) for (_, g, ms, n) in events if g == 2]
print("generation-2 collections so far (ms, collected): %s" % gen2, flush=True)
gc.collect()
gc.freeze()
series("after freeze ", args.chunks)
gc.disable()
series("collector off", args.chunks)
gc.enable()
