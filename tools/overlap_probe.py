"""Does the step gain from running image CHUNKS on separate HIP streams, staggered by one stage?
The three big kernels of configs[2]'s step stress different units (fused forward: tile walk + 32 B/px of stores;
L1 loss: 33 B/px of streaming loads; shading backward: vector ALU), and they depend on each other only per image.
Chunk i's whole step runs on stream i; stream i + 1 starts its forward when stream i's forward is done, so that
chunk i's loss pass shares the machine with chunk i + 1's forward, and its backward with chunk i + 1's loss pass.

    python tools/overlap_probe.py            # ms per step for 1, 2, 4 chunks, eager and as one replayed HIP graph

Measured (round 5, MI355X, configs[2]): graph replay 1 chunk 0.640 ms, 2 chunks lockstep 0.646, 2 staggered 0.724,
4 staggered 0.90; eager 0.70 / 1.05-1.44 / 2.0 (host-bound).  No gain: not adopted.
(Capturing 4 lockstep chunks crashed the process inside hipGraph capture on ROCm 7.2; the list below stops before it.)
"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pytorch_mesh_renderer_amd import mesh_renderer
from pytorch_mesh_renderer_amd.common import synthetic

B, W, H = int(os.environ.get("OP_B", 32)), 1024, 1024
dev = torch.device("cuda:0")
job = synthetic.sphere_job(B, W, H, 50)
tri = job["triangles"].to(dev)
eyes = job["eyes"].to(dev)     # (device cameras: what a captured HIP graph needs)
center, up = torch.zeros_like(eyes), torch.tensor([0.0, 1.0, 0.0], device=dev)


def build(chunks):
    n = B // chunks
    parts = []
    for i in range(chunks):
        sl = slice(i * n, (i + 1) * n)
        v = job["vertices"][sl].to(dev).requires_grad_(True)
        nrm, kd = job["normals"][sl].to(dev), job["diffuse"][sl].to(dev)
        lp, li = job["light_positions"][sl].to(dev), job["light_intensities"][sl].to(dev)
        args = (v, tri, nrm, kd, eyes[sl], center[sl], up, lp, li, W, H)
        with torch.no_grad():
            c, s = torch.cos(torch.tensor(0.2)), torch.sin(torch.tensor(0.2))
            rot = torch.tensor([[c, 0.0, s], [0.0, 1.0, 0.0], [-s, 0.0, c]], device=dev)
            target = mesh_renderer.render(v @ rot.T, tri, nrm @ rot.T, kd, eyes[sl], center[sl], up, lp, li, W, H)
        parts.append((args, target, torch.cuda.Stream(device=dev) if chunks > 1 else None))
    return parts


def step(parts, stagger=True):
    main = torch.cuda.current_stream(dev)
    images, prev_done = [], None
    for args, target, stream in parts:
        args[0].grad = None
        if stream is None:     # one chunk: today's step on the current stream
            mesh_renderer.losses.l1_loss(mesh_renderer.render(*args), target).backward()
            return
        stream.wait_stream(main)
        if stagger and prev_done is not None:
            stream.wait_event(prev_done)
        with torch.cuda.stream(stream):
            images.append(mesh_renderer.render(*args))
            prev_done = torch.cuda.Event()
            prev_done.record(stream)
    for (args, target, stream), image in zip(parts, images):
        with torch.cuda.stream(stream):
            mesh_renderer.losses.l1_loss(image, target).backward()   # (autograd runs the backward on the forward's stream)
    for _, _, stream in parts:
        main.wait_stream(stream)


def timed(fn, n_it=100, lead=40):
    for _ in range(lead):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n_it):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n_it


for chunks in (1, 2, 1, 2):
    parts = build(chunks)
    for stagger in ((True,) if chunks == 1 else (True, False)):
        dt = timed(lambda: step(parts, stagger))
        line = "chunks %d %s: eager %.4f ms/step" % (chunks, "staggered" if stagger else "lockstep ", dt * 1e3)
        # the same step as ONE replayed HIP graph: the host's launch rate (0.25-0.4 ms of Python per chunk) drops out
        try:
            side = torch.cuda.Stream(device=dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):
                for _ in range(3):
                    step(parts, stagger)
            torch.cuda.current_stream(dev).wait_stream(side)
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                step(parts, stagger)
            dt = timed(graph.replay)
            line += ", graph %.4f ms/step -> %.0f Mpix/s" % (dt * 1e3, B * W * H / dt / 1e6)
        except Exception as exc:   # noqa
            line += ", graph capture failed: %r" % (exc,)
        print(line, flush=True)
