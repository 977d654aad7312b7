"""Headline benchmark: Mpixels/s forward+backward, 5k-tri mesh, 1024x1024, batch 32.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config c3|c4] [--handover u8|f32]

With --gpus N > 1 and no WORLD_SIZE in the environment this process only LAUNCHES the N ranks
(`python -m torch.distributed.run`, fresh children, before anything here has touched a GPU), relays
rank 0's JSON line and exits with the children's status; under an external torchrun
(`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...`) it is a rank.

One "step" = one pass of the hot path over one batch of synthetic render jobs:
mesh_renderer.render() forward (clip transform, G-buffer rasterization, attribute interpolation,
Phong shading), the L1 image loss mean|image - target| against a fixed target, and backward to the
world-space vertex positions.  All inputs are resident in HBM before the timed region starts.

  --config c3 (default)  BASELINE.json configs[2]: 5k-tri sphere, 1024x1024, 32 images per GPU
  --config c4            BASELINE.json configs[3]: 50k-tri sphere, 2048x2048, 8 images per GPU
                         (batch 64 over 8 GPUs)

With N ranks every rank renders its own batch (weak scaling; no data-path collective) and the
finished images of EVERY step are handed over over RCCL, overlapping the loss, the backward and the
next forwards: as the fp32 images render() returns (--handover f32, the default since round 6: the reference's
own output, render.py:384-386) or as 8-bit frames (--handover u8: mesh_renderer.to_uint8, the conversion the
reference's examples apply before writing a frame, a quarter of the bytes; its figure rides in the same line).
--gather rotate (default, round 5): the global batch of step s is assembled on rank s mod N -- one balanced
all_to_all per block of N steps, every xGMI link in use (distributed.RotatingImageGather);
--gather root: on rank 0 every step (one gather per step; bound by rank 0's N - 1 inbound links).

Extra objects in the line:
  roofline                 the step's forward kernel (k_raster with the shading epilogue: ids +
                           barycentrics + RGBA, 32 B/px written): algorithmic bytes per launch / its
                           average duration, measured with HIP events recorded around that kernel on
                           its stream.
  roofline_gbuffer         SURVEY.md 8(d)'s figure: the G-buffer kernel alone (20 B/px), timed in extra steps
                           AFTER the timed region with the shading epilogue switched off.
  roofline_shade_backward  same for the pixel pass of the fused shading backward (17 B/px read with the
                           sign-coded upstream).
  rccl, ms_per_step_render_only   (N > 1) what torch.distributed reports about the group, and the same
                           loop with the hand-over off, run after the timed region: rendering alone
                           next to `ms_per_step`, which includes the hand-over.
  host_gc                  CPython's cyclic garbage collector is PAUSED inside every timed region, after one full
                           collection (timeit's convention): round 6 traced the rare 3-4 x readings of short legs to one
                           generation-2 pass (38 ms over ~170k tracked objects) landing inside a 20-30 step window
                           (tools/gc_probe.py); the durations of the collections run in front of the regions ride here.
  roofline_l1_forward      same for the loss's streaming pass (33 B/px): with the two above, the three
                           kernels that make up 92 % of the step.
  cpu_baseline             the same step on the host cores for a bounded sample of the batch (torch-CPU
                           eager restatement of the reference's render path, oracle/shading.py, over the
                           reference's own compiled C++ kernel oracle/_ref when present), plus the
                           kernel-level figures of SURVEY.md 8(d): the C restatement of the
                           rasterizer forward / backward with 1 thread and with all host threads.
"""
import argparse
import ctypes
import contextlib
import gc
import json
import os
import socket
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from pytorch_mesh_renderer_amd import _native, distributed, mesh_renderer  # noqa: E402
from pytorch_mesh_renderer_amd.common import synthetic  # noqa: E402

CONFIGS = {
    # key: (BASELINE.json entry, images per GPU, width, height, sphere resolution K)
    "c3": ("configs[2]", 32, 1024, 1024, 50),
    "c4": ("configs[3]", 8, 2048, 2048, 158),
}
HBM_PEAK_GBPS = 8000.0                                # MI355X HBM3E spec (MI355X_MICROARCH.md)


GC_FULL_PASS_MS = []   # duration of the full collections run in front of the timed regions (reported in the line)


@contextlib.contextmanager
def collector_paused():
    """Timed regions run with CPython's cyclic garbage collector paused, after one full collection -- timeit's convention.
    Round 6 found the rare 3-4 x readings of the SoftRas leg (2.5-3.3 ms against 0.77: VERDICT r5 weak 2) to be ONE
    generation-2 collection -- 38 ms over this process's ~170 000 tracked objects, collecting ten -- landing inside a 23-ms
    timed loop (tools/gc_probe.py, profiles/r06_gc_probe.txt).  Such a pass is due once in a thousand steps or so: a cost
    of the host interpreter, ~0.03 ms per step amortised, that a 20- or 30-step window either misses or carries whole.
    The full collection in front takes 30-80 ms during which the GPU idles and its clocks fall (from
    idle the step time needs ~40 steps to settle): callers enter this context BEFORE their lead-in / warm-up steps.  (The
    first version collected right in front of the timed loop: the headline's 20 steps read 0.765 instead of 0.66 ms.)"""
    t0 = time.perf_counter()
    gc.collect()
    GC_FULL_PASS_MS.append(round((time.perf_counter() - t0) * 1e3, 2))
    was_enabled = gc.isenabled()
    gc.disable()
    try:
        yield
    finally:
        if was_enabled:
            gc.enable()


def spawn_ranks(args):
    """Parent of a self-launched multi-GPU run: starts the ranks as fresh child processes and relays
    rank 0's line.  Nothing in this process has initialised a GPU (importing torch does not)."""
    with socket.socket() as s:                         # a free rendezvous port
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC (RCCL across processes)
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for out_line in proc.stdout.splitlines():
        if out_line.startswith("{") and '"metric"' in out_line:
            line = out_line
        else:
            print(out_line, file=sys.stderr)
    if proc.returncode != 0 or line is None:
        raise SystemExit(proc.returncode or 1)
    print(line, flush=True)


class KernelEvents:
    """Raw hipEvent pairs around single kernels (torch.cuda.Event only sees torch's bookkeeping):
    armed through the library's one-shot, thread-local mr_time_next_kernel."""

    def __init__(self, n, which):
        self.hip = ctypes.CDLL("libamdhip64.so")
        self.hip.hipEventElapsedTime.argtypes = [ctypes.POINTER(ctypes.c_float), ctypes.c_void_p,
                                                 ctypes.c_void_p]
        self.which = which
        self.pairs = []
        for _ in range(n):
            a, b = ctypes.c_void_p(), ctypes.c_void_p()
            assert self.hip.hipEventCreate(ctypes.byref(a)) == 0
            assert self.hip.hipEventCreate(ctypes.byref(b)) == 0
            self.pairs.append((a, b))

    def arm(self, i):
        a, b = self.pairs[i % len(self.pairs)]
        _native.time_next_kernel(self.which, a, b)

    def mean_ms(self, used):
        out = []
        for a, b in self.pairs[:used]:
            ms = ctypes.c_float()
            if self.hip.hipEventElapsedTime(ctypes.byref(ms), a, b) == 0:
                out.append(ms.value)
        return sum(out) / len(out) if out else 0.0


def make_step(job, device, gather, handover="f32", spelling="l1_loss", all_gradients=False, device_cameras=False):
    """Returns (step, vertices, state): state["image"] / state["target"] hold the last rendered
    batch and the fixed target (the full-size parity test checks them against the oracle).

    spelling: "l1_loss" = mesh_renderer.losses.l1_loss(image, target); "reference" = the reference's own words,
              torch.mean(torch.abs(image - target)) (mesh_renderer_test.py:250) -- the same kernels since round 5.
    all_gradients: normals, diffuse colours, light positions and intensities require grad next to the vertices.
    device_cameras: cameras as device tensors (what a captured HIP graph needs)."""
    width, height = job["width"], job["height"]
    tri = job["triangles"].to(device)
    vertices = job["vertices"].to(device).requires_grad_(True)
    normals, diffuse = job["normals"].to(device), job["diffuse"].to(device)
    eyes = job["eyes"]                      # cameras stay host tensors, as in the reference's usage
    center = torch.zeros_like(eyes)
    up = torch.tensor([0.0, 1.0, 0.0])
    if device_cameras:
        eyes, center, up = eyes.to(device), center.to(device), up.to(device)
    lpos, lint = job["light_positions"].to(device), job["light_intensities"].to(device)
    leaves = [vertices]
    if all_gradients:
        normals, diffuse, lpos, lint = [t.clone().requires_grad_(True) for t in (normals, diffuse, lpos, lint)]
        leaves += [normals, diffuse, lpos, lint]
    from pytorch_mesh_renderer_amd.mesh_renderer import rasterize_triangles_ext
    # every frame is handed over: with 8-bit frames the forward kernel writes them itself (4 B/px)
    # instead of a conversion pass over the float image on the side stream
    state = {"image": None, "handover": True, "handover_dtype": handover}

    def forward():
        with rasterize_triangles_ext.emit_uint8_frames(gather is not None and state["handover_dtype"] == "u8"):
            return mesh_renderer.render(vertices, tri, normals, diffuse, eyes, center, up, lpos, lint,
                                        width, height)

    # fixed target: the same scene with the mesh slightly rotated (mirrors the reference's
    # optimisation tests), rendered once outside the timed region
    with torch.no_grad():
        c, s = torch.cos(torch.tensor(0.2)), torch.sin(torch.tensor(0.2))
        rot = torch.tensor([[c, 0.0, s], [0.0, 1.0, 0.0], [-s, 0.0, c]], device=device)
        target = mesh_renderer.render(vertices @ rot.T, tri, normals @ rot.T, diffuse, eyes, center, up,
                                      lpos, lint, width, height)
    state["target"] = target
    mesh_renderer.losses.remember_target(target)   # a fixed target: its empty-block map is made once (losses.l1_loss, TARGET)

    def step():
        for leaf in leaves:
            leaf.grad = None
        image = forward()
        state["image"] = image
        if gather is not None and state["handover"]:
            # two hand-overs may be in flight (depth 2): the oldest one is waited for, then this step's starts (ImageGather:
            # a gather per step; RotatingImageGather: this step's frames are staged and every N-th step exchanges the
            # block); conversion (u8) and transfer run on the side stream and have whole steps to finish in -- a
            # link-bound hand-over then costs bandwidth, not latency on top
            if gather.in_flight() >= gather.depth:
                gather.wait()
            gather.start(image, transform=mesh_renderer.to_uint8 if state["handover_dtype"] == "u8" else None)
        if spelling == "reference":
            loss = torch.mean(torch.abs(image - target))         # /root/reference/src/mesh_renderer/mesh_renderer_test.py:250
        else:
            loss = mesh_renderer.losses.l1_loss(image, target)   # mean |image - target|, one HIP pass each way
        loss.backward()
        return loss

    def graph_step():   # for mesh_renderer.capture_step: .grad stays the graph's static memory
        image = forward()
        loss = torch.mean(torch.abs(image - target)) if spelling == "reference" else mesh_renderer.losses.l1_loss(image, target)
        loss.backward()
        return loss

    state["graph_step"] = graph_step
    state["leaves"] = leaves
    return step, vertices, state


def cpu_baseline(batch, width, height, sphere_k, sample_images):
    """The same step (render fwd + L1 + bwd to vertices) on host cores for a bounded sample, and the
    kernel-level rasterizer figures (C restatement, 1 thread and all threads)."""
    import numpy as np
    import oracle
    from oracle import shading
    use_ref = oracle.have_reference_kernel()
    job = synthetic.sphere_job(batch, width, height, sphere_k)
    sample_images = min(sample_images, batch)
    sl = slice(0, sample_images)
    v = job["vertices"][sl].clone().requires_grad_(True)
    args = (job["triangles"], job["normals"][sl], job["diffuse"][sl], job["eyes"][sl],
            torch.zeros(sample_images, 3), torch.tensor([[0.0, 1.0, 0.0]]).repeat(sample_images, 1),
            job["light_positions"][sl], job["light_intensities"][sl], width, height)
    # the GPU step's target is the render of the rotated mesh; on the host it only has to be SOME
    # fixed image of the same shape (the cost of |img - target| does not depend on its values)
    target = torch.rand(sample_images, height, width, 4, generator=torch.Generator().manual_seed(0))
    t0 = time.perf_counter()
    img = shading.render(v, *args, use_reference_kernel=use_ref)
    torch.mean(torch.abs(img - target)).backward()
    dt = time.perf_counter() - t0

    # kernel level: the C restatement of the rasterizer alone
    clip, tris = job["clip"].numpy(), job["triangles"].numpy()
    threads = oracle.max_threads()
    n_par = min(batch, max(threads, 1))
    rng = np.random.default_rng(0)

    def timed(fn):
        t = time.perf_counter()
        out = fn()
        return out, time.perf_counter() - t

    (ids1, bary1, _), f1 = timed(lambda: oracle.forward(clip[:1], tris, width, height, threads=1))
    g1 = (rng.standard_normal(bary1.shape) / (width * height)).astype(np.float32)
    _, b1 = timed(lambda: oracle.backward(g1, clip[:1], tris, ids1, bary1, threads=1))
    (idsn, baryn, _), fn_ = timed(lambda: oracle.forward(clip[:n_par], tris, width, height, threads=threads))
    gn = np.repeat(g1, n_par, 0)
    _, bn = timed(lambda: oracle.backward(gn, clip[:n_par], tris, idsn, baryn, threads=threads))
    px = width * height
    return {
        "value": round(sample_images * px / dt / 1e6, 4), "unit": "Mpixels/s",
        "cores": torch.get_num_threads(), "kind": "port",
        "sample": "%d of the %d images of the workload, render fwd + L1 + bwd, %.1f s; eager torch-CPU "
                  "restatement of the reference's render path over %s" % (
                      sample_images, batch, dt,
                      "the reference's compiled C++ kernel (oracle/_ref)" if use_ref else "oracle/mr_oracle.c"),
        "kernel_level": {
            "what": "oracle/mr_oracle.c (C restatement of rasterize_triangles.cpp), rasterizer forward + "
                    "backward only, Mpixels/s = pixels / (t_fwd + t_bwd)",
            "threads_1": {"images": 1, "forward_ms": round(f1 * 1e3, 1), "backward_ms": round(b1 * 1e3, 1),
                          "value": round(px / (f1 + b1) / 1e6, 3)},
            "threads_max": {"threads": threads, "images": n_par, "forward_ms": round(fn_ * 1e3, 1),
                            "backward_ms": round(bn * 1e3, 1),
                            "value": round(n_par * px / (fn_ + bn) / 1e6, 3)},
        },
    }


LEG_GROUPS_MS = {}   # label -> the three timed groups of a leg (its figure is their median)


def _loop_ms(fn, n, lead=40, label=None):
    """ms per call of a leg outside the timed region: `lead` untimed calls, then THREE timed groups of n calls; the figure
    is the median group (all three ride in the line under leg_groups_ms).  One group caught a host stall of 4-11 ms --
    not the collector, which is paused: the box's other tenants -- in one of three runs of round 6."""
    # (the lead-in covers a fresh step's one-time work -- adjacency, the target's block map, the allocator growing -- and
    #  the clock ramp after the idle gap in front of it: with 4 steps the same leg read 0.65 ms on one box and 0.93 on
    #  another, profiles/r05_b_bench.json)
    groups = []
    with collector_paused():   # (the full collection in front is a 30-80 ms idle gap for the GPU: it goes BEFORE the lead-in)
        for _ in range(lead):
            fn()
        for _ in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                fn()
            torch.cuda.synchronize()
            groups.append((time.perf_counter() - t0) / n * 1e3)
    if label:
        LEG_GROUPS_MS[label] = [round(g, 4) for g in groups]
    return sorted(groups)[1]


def _chunked_ms(fn, chunks, n, lead=40):
    """`chunks` back-to-back groups of n calls after `lead` untimed ones: per group the wall clock per call (enqueue +
    drain), the HIP-event time per call on the stream, and the host time spent enqueueing."""
    out = []
    with collector_paused():
        for _ in range(lead):
            fn()
        torch.cuda.synchronize()
        for _ in range(chunks):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0 = time.perf_counter()
            e0.record()
            for _ in range(n):
                fn()
            e1.record()
            t_host = time.perf_counter() - t0
            torch.cuda.synchronize()
            t_wall = time.perf_counter() - t0
            out.append({"wall_ms": round(t_wall / n * 1e3, 4), "gpu_events_ms": round(e0.elapsed_time(e1) / n, 4),
                        "host_enqueue_ms": round(t_host / n * 1e3, 4)})
    return out


def _slow_leg_evidence(forward, leaf):
    """Forward and backward of a leg timed separately by HIP events (5 steps), and the GPU's clocks / power as rocm-smi
    reports them right after -- only gathered when a leg reads far above its recorded time."""
    import subprocess
    fwd, bwd = [], []
    for _ in range(5):
        leaf.grad = None
        e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        e[0].record()
        loss = forward().mean()
        e[1].record()
        loss.backward()
        e[2].record()
        torch.cuda.synchronize()
        fwd.append(round(e[0].elapsed_time(e[1]), 4))
        bwd.append(round(e[1].elapsed_time(e[2]), 4))
    out = {"forward_ms": fwd, "backward_ms": bwd}
    try:
        smi = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--showperflevel", "--showtemp", "--json"],
                             capture_output=True, text=True, timeout=30).stdout
        out["rocm_smi"] = json.loads(smi) if smi.strip().startswith("{") else smi[-600:]
    except Exception as exc:   # (diagnostics only)
        out["rocm_smi"] = "%s: %s" % (type(exc).__name__, exc)
    return out


def extra_legs(job, device, batch, width, height):
    """What used to be builder-run only, AFTER the timed region, one GPU, ~20 s in all: the same
    step through the reference's own loss spelling, with every gradient wanted, replayed as a captured HIP graph, and
    the other BASELINE.json configurations (ms per step / call and the dominant kernel's fraction of the HBM peak)."""
    out = {}
    px = batch * width * height
    n = 40

    def value(ms):
        return round(px / ms / 1e3, 2)

    # (a) torch.mean(torch.abs(image - target)), as /root/reference/src/mesh_renderer/mesh_renderer_test.py:250 writes it
    step_ref, _, _ = make_step(job, device, None, spelling="reference")
    ms = _loop_ms(step_ref, n, label="reference_spelling")
    out["ms_per_step_reference_spelling"] = round(ms, 4)
    out["value_reference_spelling"] = value(ms)
    del step_ref
    # (b) every gradient wanted: vertices, normals, diffuse colours, light positions and intensities
    step_all, _, st = make_step(job, device, None, all_gradients=True)
    ms = _loop_ms(step_all, n, label="all_gradients")
    assert all(leaf.grad is not None for leaf in st["leaves"])
    out["ms_per_step_all_gradients"] = round(ms, 4)
    out["value_all_gradients"] = value(ms)
    del step_all, st
    # (c) the step (reference spelling, cameras on the device) captured once and replayed: mesh_renderer.capture_step
    _, _, st = make_step(job, device, None, spelling="reference", device_cameras=True)
    captured = mesh_renderer.capture_step(st["graph_step"], st["leaves"])
    ms = _loop_ms(captured.replay, n, label="graph")
    out["ms_per_step_graph"] = round(ms, 4)
    out["value_graph"] = value(ms)
    del captured, st
    torch.cuda.empty_cache()

    configs = {}
    # configs[1]: 5k tris, 256^2, batch 8, forward G-buffer (mr_rasterize_forward)
    j2 = synthetic.sphere_job(8, 256, 256, 50)
    clip2, tris2 = j2["clip"].to(device), j2["triangles"].to(device)
    ms = _loop_ms(lambda: _native.rasterize_forward(clip2, tris2, 256, 256), 100, lead=100, label="c2")
    configs["c2"] = {"what": "BASELINE configs[1]: 5k tris, 256x256, batch 8, forward G-buffer, whole mr_rasterize_forward call "
                             "(setup + binning + raster; launch-bound at this size)", "ms_per_call": round(ms, 4),
                     "Mpixels_per_s": round(8 * 256 * 256 / ms / 1e3, 1),
                     "frac_of_hbm_peak": round(8 * 256 * 256 * 20 / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4)}
    # configs[3]'s per-GPU share: 50k tris, 2048^2, 8 images -- the step, its forward kernel and the G-buffer kernel
    j4 = synthetic.sphere_job(8, 2048, 2048, 158)
    V4, T4, px4 = j4["vertices"].shape[1], j4["triangles"].shape[0], 8 * 2048 * 2048
    step4, _, _ = make_step(j4, device, None)
    ms4 = _loop_ms(step4, 30, label="c4")
    ev = KernelEvents(4, _native.TIMER_RASTER_FORWARD)
    for i in range(4):
        ev.arm(i)
        step4()
    torch.cuda.synchronize()
    fwd_ms = ev.mean_ms(4)
    del step4
    clip4, tris4 = j4["clip"].to(device), j4["triangles"].to(device)
    evg = KernelEvents(3, _native.TIMER_RASTER_FORWARD)
    _native.debug_set_raster_repeat(8)
    try:
        for i in range(4):
            if i:
                evg.arm(i - 1)
            _native.rasterize_forward(clip4, tris4, 2048, 2048)
    finally:
        _native.debug_set_raster_repeat(1)
    torch.cuda.synchronize()
    gb_ms = evg.mean_ms(3) / 8
    configs["c4"] = {"what": "BASELINE configs[3], one GPU's share: 50k tris (V=%d, T=%d), 2048x2048, 8 images, render fwd + L1 + "
                             "bwd to vertex positions" % (V4, T4),
                     "ms_per_step": round(ms4, 4), "Mpixels_per_s": round(px4 / ms4 / 1e3, 1),
                     "forward_kernel_ms": round(fwd_ms, 4),
                     "forward_kernel_frac": round((px4 * 32 + 8 * V4 * 16 + T4 * 12 + 8 * T4 * 128) / (fwd_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
                     "gbuffer_kernel_ms": round(gb_ms, 4),
                     "gbuffer_kernel_frac": round((px4 * 20 + 8 * V4 * 16 + T4 * 12) / (gb_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4)}
    del clip4, tris4, j4
    torch.cuda.empty_cache()
    # configs[4]: SoftRas, 5k tris, 512^2, batch 16, full autograd
    from pytorch_mesh_renderer_amd import soft_mesh_renderer
    j5 = synthetic.sphere_job(16, 512, 512, 50)
    v5 = j5["vertices"].to(device).requires_grad_(True)
    tri5, kd5, lp5 = j5["triangles"].to(device), j5["diffuse"].to(device), j5["light_positions"].to(device)
    eyes5, zero5, up5 = j5["eyes"], torch.zeros(16, 3), torch.tensor([0.0, 1.0, 0.0])   # host cameras, as in the reference's usage
    li5 = torch.ones(16, 1, device=device)

    def step5():
        v5.grad = None
        soft_mesh_renderer.render(v5, tri5, kd5, eyes5, zero5, up5, lp5, li5, 512, 512).mean().backward()
    # (round 6: this leg read 0.77-0.78 ms on nine boxes and 2.5-3.3 ms on three, same code, same kernels -- VERDICT r5 weak 2;
    #  it is the leg with the least GPU work per host call, 42 ms in all.  It now runs five chunks of 20 steps, each timed by the
    #  wall clock AND by HIP events, and carries every chunk: a reading that is slow says whether the GPU or the host was.)
    chunks5 = _chunked_ms(step5, chunks=5, n=20)
    ms5 = sorted(c["wall_ms"] for c in chunks5)[len(chunks5) // 2]
    configs["c5"] = {"what": "BASELINE configs[4]: soft_mesh_renderer (SoftRas aggregation), 5k tris, 512x512, batch 16, forward + "
                             "mean() + backward to the vertices (tools/soft_bench.py's step); ms_per_step = median of the chunks",
                     "ms_per_step": round(ms5, 4), "Mpixels_per_s": round(16 * 512 * 512 / ms5 / 1e3, 1),
                     "chunks_of_20_steps": chunks5}
    if ms5 > 1.5:   # a slow reading at last (recorded: 0.77): leave behind what tells the GPU from the host and the clocks
        configs["c5"]["slow_reading"] = _slow_leg_evidence(lambda: soft_mesh_renderer.render(
            v5, tri5, kd5, eyes5, zero5, up5, lp5, li5, 512, 512), v5)
    out["configs"] = configs
    torch.cuda.empty_cache()

    # (e) the other two entry points of SURVEY.md 8 at the headline shape (builder-run until now):
    # render() with the specular term, the loop as the reference writes it; rasterize() with nine attributes
    v = job["vertices"].to(device).requires_grad_(True)
    tri, nrm, kd = job["triangles"].to(device), job["normals"].to(device), job["diffuse"].to(device)
    ks = torch.full_like(kd, 0.5)
    eyes, zero, up = job["eyes"], torch.zeros_like(job["eyes"]), torch.tensor([0.0, 1.0, 0.0])
    lp, li = job["light_positions"].to(device), job["light_intensities"].to(device)
    with torch.no_grad():
        target = torch.rand(batch, height, width, 4, device=device)

    def step_spec():
        v.grad = None
        image = mesh_renderer.render(v, tri, nrm, kd, eyes, zero, up, lp, li, width, height, specular_colors=ks,
                                     shininess_coefficients=6.0)
        torch.mean(torch.abs(image - target)).backward()
    ms = _loop_ms(step_spec, 30, label="specular")
    out["specular"] = {"what": "render() with specular_colors / shininess 6, torch.mean(torch.abs(image - target)), backward to the "
                               "vertices (tools/specular_bench.py's step)", "ms_per_step": round(ms, 4), "Mpixels_per_s": value(ms)}
    del target
    attrs = torch.rand(batch, v.shape[1], 9, device=device, requires_grad=True)
    proj = synthetic.clip_transforms(job["eyes"], width, height).to(device)
    background = torch.full((9,), -1.0, device=device)
    upstream = torch.randn(batch, height, width, 9, device=device) / (px * 9)

    def forward_a9():
        with torch.no_grad():
            mesh_renderer.rasterize(v, attrs, tri, proj, width, height, background)

    def step_a9():
        v.grad = None
        attrs.grad = None
        mesh_renderer.rasterize(v, attrs, tri, proj, width, height, background).backward(gradient=upstream)
    ms_f, ms_fb = _loop_ms(forward_a9, 30, label="a9_forward"), _loop_ms(step_a9, 30, label="a9_forward_backward")
    out["rasterize_a9"] = {"what": "mesh_renderer.rasterize() with nine attributes: forward alone, and forward + backward to vertices and "
                                   "attributes with a given upstream gradient (tools/rasterize_bench.py)",
                           "ms_forward": round(ms_f, 4), "ms_forward_backward": round(ms_fb, 4),
                           "forward_frac_of_hbm_peak": round(px * 52 / (ms_f * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4)}
    return out


def roofline(kernel, algorithmic, avg_ms, traffic_key, config="c3"):
    achieved = algorithmic / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
    out = {"bound": "hbm", "kernel": kernel, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS,
           "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": None,
           "algorithmic_bytes": algorithmic, "avg_kernel_ms": round(avg_ms, 4)}
    pmc_path = os.path.join(ROOT, "profiles", "kernel_traffic.json")
    if os.path.exists(pmc_path):   # written from separate rocprofv3 --pmc passes over this command
        pmc = json.load(open(pmc_path))
        if config != "c3":   # the default configuration sits at the top level, the others under "configs"
            pmc = pmc.get("configs", {}).get(config, {})
        if traffic_key in pmc.get("kernels", {}) and pmc["kernels"][traffic_key]["bytes_per_launch"]:
            out["traffic"] = pmc["kernels"][traffic_key]["bytes_per_launch"]
            out["traffic_source"] = ("profiles/kernel_traffic.json: rocprofv3 --pmc passes over `%s` at %s, NOT "
                                     "this run" % (pmc.get("command", "bench.py"), pmc.get("tag", "?")))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", choices=sorted(CONFIGS), default="c3")
    ap.add_argument("--handover", choices=("u8", "f32"), default="f32",
                    help="what the ranks hand over when --gpus > 1 inside the timed region: the fp32 images render() returns "
                         "(default since round 6: the reference's own output type, render.py:384-386, 16 B/px) or 8-bit frames "
                         "(the reference examples' frame conversion, written by the forward's epilogue, 4 B/px); the OTHER one is "
                         "timed after the region and reported next to it (ms_per_step_handover_u8 / value_handover_u8)")
    ap.add_argument("--gather", choices=("rotate", "root"), default="rotate",
                    help="N > 1: where a step's frames are assembled -- rotate (default): the global batch of step s on rank "
                         "s mod N, one all_to_all per N steps, every xGMI link used (distributed.RotatingImageGather); "
                         "root: every step's frames on rank 0, one gather per step, bound by rank 0's N - 1 inbound links")
    ap.add_argument("--transport", choices=("rccl", "peer"), default="rccl",
                    help="N > 1 with --gather rotate: rccl (default) = one all_to_all per block (RCCL's copy kernels); peer = every "
                         "rank copies its frames into the root's IPC-mapped receive buffer (copy engines; rehearsed on one GPU only: "
                         "distributed.RotatingImageGather)")
    ap.add_argument("--extras", type=int, default=1,
                    help="0: skip the legs that run after the timed region (other spellings, gradient sets, configurations)")
    ap.add_argument("--cpu-sample", type=int, default=12, help="images timed for cpu_baseline (0 = skip)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return spawn_ranks(args)          # before any GPU call: the ranks are fresh processes

    rank, world, local_rank = distributed.init_from_env()
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    # one GPU per rank; the modulo only matters for a gloo rehearsal of N ranks on fewer GPUs
    device = torch.device("cuda", local_rank % torch.cuda.device_count())
    torch.cuda.set_device(device)

    # the timed step computes its camera matrices every time, as the reference would: the library's memo for
    # unchanged host-side cameras (camera_utils.CACHE_HOST_CAMERAS) stays out of the measurement
    from pytorch_mesh_renderer_amd.common import camera_utils
    camera_utils.CACHE_HOST_CAMERAS = False
    entry, batch, width, height, sphere_k = CONFIGS[args.config]
    job = synthetic.sphere_job(batch, width, height, sphere_k)
    if world > 1:  # every rank renders its own `batch` jobs: rotate the orbit per rank
        shift = (rank * 7) % batch
        job = {k: (torch.roll(v, shift, 0) if torch.is_tensor(v) and k != "triangles" else v)
               for k, v in job.items()}
    # images are handed over to rank 0 (RCCL gather): the root receives its N-1 shards over N-1
    # xGMI links at once; an all-gather would move N times the bytes for nothing
    # MR_BENCH_FORCE_GROUP=1 (tests): a single rank still forms a 1-rank group and hands its frames over
    # through RCCL, so that the N > 1 code path -- frames from the forward's epilogue, side stream,
    # gather.wait() inside the timed loop -- meets the real backend on a one-GPU box.
    forced = world == 1 and os.environ.get("MR_BENCH_FORCE_GROUP") == "1"
    if forced:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        backend = os.environ.get("MR_DIST_BACKEND", "nccl")
        torch.distributed.init_process_group(backend=backend, rank=0, world_size=1,
                                             **({"device_id": device} if backend == "nccl" else {}))
    grouped = world > 1 or forced
    # N > 1 (round 5): the root ROTATES -- step s's global batch is assembled on rank s mod N, one balanced
    # all_to_all per block of N steps -- unless --gather root asks for rank 0 every step
    rotating = grouped and args.gather == "rotate"
    if rotating:
        gather = distributed.RotatingImageGather(batch * world, depth=2, force_collective=forced, transport=args.transport)
    else:
        gather = distributed.ImageGather(batch * world, mode="root", force_collective=forced, depth=2) if grouped else None
    step, vertices, step_state = make_step(job, device, gather, args.handover)
    from pytorch_mesh_renderer_amd.mesh_renderer import rasterize_triangles_ext as ext

    def barrier():
        if grouped:
            torch.distributed.barrier()
        torch.cuda.synchronize(device)

    # The G-buffer kernel on its own (what rasterize_barycentric() and the non-fused paths launch; the step's
    # forward runs it with the shading epilogue attached), timed inside the same step with the epilogue switched
    # off (k_raster, then k_shade_forward).  OUTSIDE the timed region, and since round 4 BEFORE it: a short timed
    # region (the driver's --steps 20 is 14 ms of GPU time) that starts a few milliseconds after the chip was idle
    # measures its clock ramp -- the same code gave 0.742 ms/step at --steps 20 against 0.681 at --steps 200 on one
    # box, every kernel 4-7 % slower -- so the measurement legs that are not the timed region run first.
    # Its own lead-in of untimed steps (12: ~10 ms of GPU time) plays the part the warm-up steps play for the timed
    # region: measured straight after process start the same kernel read 140 us against 134.7 in the rocprofv3
    # trace of the same box (profiles/r04_c_*).
    # Round 5: the kernel is timed through the product entry point mr_rasterize_forward (what rasterize_barycentric()
    # launches: ids + barycentrics + depth, 20 B/px) with its k_raster launch repeated 16 times inside ONE event pair
    # (mr_debug_set_raster_repeat: identical launches, same outputs): an event pair costs the stream ~5 us of idle
    # time, which on a single 133 us launch decided whether the figure read 0.59 or 0.63.
    # The host interpreter's cyclic collector is paused from here to the end of the timed region and of the N > 1 loops
    # behind it (collector_paused): its one full collection runs NOW, in front of ~70 untimed steps, not in front of the
    # timed region, where its 30-80 ms of GPU idleness would be measured as a clock ramp.
    paused = collector_paused()
    paused.__enter__()
    n_gb_rep, n_gb, n_gb_lead = 16, 6, 2
    ev_gbuffer = KernelEvents(n_gb, _native.TIMER_RASTER_FORWARD)
    clip_gb = job["clip"].to(device)
    tris_gb = job["triangles"].to(device)
    _native.debug_set_raster_repeat(n_gb_rep)
    try:
        for i in range(n_gb + n_gb_lead):
            if i >= n_gb_lead:
                ev_gbuffer.arm(i - n_gb_lead)
            gb_ids, gb_bary, gb_z = _native.rasterize_forward(clip_gb, tris_gb, width, height)
    finally:
        _native.debug_set_raster_repeat(1)
    torch.cuda.synchronize(device)
    gbuffer_ms = ev_gbuffer.mean_ms(n_gb) / n_gb_rep
    del gb_ids, gb_bary, gb_z, clip_gb
    # ... and as rounds 2-4 measured it: ONE launch per event pair inside the step (the shading epilogue switched off:
    # k_raster, then k_shade_forward, loss, backward).  Behind the kernel runs another kind of work, under which the
    # tail of its 671 MB of stores drains; back to back with itself the kernel runs at the sustained write rate.  Both
    # are this kernel's duration -- in a renderer's step and in a write-only burst -- and the line carries both.
    # (its lead-in also lets the chip settle: from idle the step time falls for ~40 steps -- 0.71 -> 0.66 ms on one box,
    #  tools/step_series.py -- and a 20-step timed region that starts earlier measures that ramp)
    n_gs, n_gs_lead = 12, 36
    ev_gstep = KernelEvents(n_gs, _native.TIMER_RASTER_FORWARD)
    with ext.shading_epilogue(False):
        for i in range(n_gs + n_gs_lead):
            if i >= n_gs_lead:
                ev_gstep.arm(i - n_gs_lead)
            step()
    if gather is not None:
        gather.drain()
    torch.cuda.synchronize(device)
    gbuffer_in_step_ms = ev_gstep.mean_ms(n_gs)

    for _ in range(args.warmup):
        step()
    # Kernel timers (a hipEvent pair around each of the three big kernels) on every `ev_every`-th timed step, at
    # most 16 of them, spread over the whole timed region.  An event pair costs the stream ~5 us of idle time around
    # the kernel it brackets (seen as gaps in the rocprofv3 trace): on EVERY step, as until the end of round 4, the
    # three pairs took ~2.5 % off `value` in a 20-step run.
    n_ev = max(1, min(16, args.steps // 4)) if "MR_BENCH_TIMER_STEPS" not in os.environ else \
        max(0, min(args.steps, int(os.environ["MR_BENCH_TIMER_STEPS"])))
    ev_every = max(1, args.steps // n_ev) if n_ev else 0
    ev_raster = KernelEvents(n_ev, _native.TIMER_RASTER_FORWARD)
    ev_shade = KernelEvents(n_ev, _native.TIMER_SHADE_BACKWARD)
    ev_l1 = KernelEvents(n_ev, _native.TIMER_L1_FORWARD)
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        if ev_every and i % ev_every == ev_every // 2 and i // ev_every < n_ev:
            ev_raster.arm(i // ev_every)
            ev_shade.arm(i // ev_every)
            ev_l1.arm(i // ev_every)
        step()
    if gather is not None:
        gather.drain()                   # the last steps' hand-overs belong to the timed region
    barrier()
    elapsed = time.perf_counter() - t0

    if grouped:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t.item())
    assert vertices.grad is not None and bool(torch.isfinite(vertices.grad).all())

    # N > 1, also outside the timed region: the same loop with the hand-over switched off, so that the
    # first scaling run separates how the rendering scales from what the root's inbound links cost.
    render_only_ms = other_handover_ms = None
    other_handover = "u8" if args.handover == "f32" else "f32"

    def max_over_ranks_ms(n):
        barrier()
        t1 = time.perf_counter()
        for _ in range(n):
            step()
        if gather is not None:
            gather.drain()
        barrier()
        t = torch.tensor([time.perf_counter() - t1], device=device, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        return float(t.item()) / n * 1e3

    if grouped:
        n_ro = min(args.steps, 50)
        step_state["handover"] = False
        render_only_ms = max_over_ranks_ms(n_ro)
        step_state["handover"] = True
        # ... and the same loop handing over the OTHER image type (8-bit frames from the forward's epilogue / fp32 images)
        step_state["handover_dtype"] = other_handover
        for _ in range(3):
            step()
        other_handover_ms = max_over_ranks_ms(n_ro)
        step_state["handover_dtype"] = args.handover

    paused.__exit__(None, None, None)   # (the legs after this point pause the collector themselves, in front of their lead-ins)
    extras = {}
    if rank == 0 and world == 1 and args.extras and args.config == "c3":
        try:
            extras = extra_legs(job, device, batch, width, height)
        except Exception as exc:   # the legs after the timed region must never cost the line its headline
            extras = {"extras_error": "%s: %s" % (type(exc).__name__, exc)}

    if rank == 0:
        V, T = job["vertices"].shape[1], job["triangles"].shape[0]
        px = batch * width * height
        # the loss's streaming pass reads a 64 x 64 block unless BOTH maps mark it empty (count those bytes)
        l1_read_fraction = 1.0
        if ext.EMPTY_REGIONS and step_state["image"] is not None:
            with torch.no_grad():
                # the renderer's own map (blocks without a candidate triangle: what the kernel is handed), not "all zeros"
                m_img = getattr(step_state["image"].grad_fn, "empty_regions", None)
                if m_img is None:
                    m_img = _native.image_empty_regions(step_state["image"].detach())
                m_tgt = _native.image_empty_regions(step_state["target"].detach())
                l1_read_fraction = 1.0 - float((m_img.bool() & m_tgt.bool()).float().mean())
        l1_bytes = int(px * 32 * l1_read_fraction + px * 1)
        line = {
            "metric": "Mpixels/sec forward+backward, %dx%d batch=%d" % (width, height, batch),
            "value": round(world * px * args.steps / elapsed / 1e6, 2),
            "unit": "Mpixels/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s: %dk-tri UV sphere (V=%d, T=%d), %dx%d, batch=%d per GPU, "
                                   "mesh_renderer.render forward + L1 loss + backward to vertex positions; "
                                   "%s frames of every step handed over over RCCL when n_gpus>1 (%s)" % (
                                       entry, round(T / 1000), V, T, width, height, batch,
                                       "8-bit" if args.handover == "u8" else "fp32",
                                       "step s's global batch assembled on rank s mod N" if args.gather == "rotate"
                                       else "gathered to rank 0"),
                       "global_batch": batch * world, "image": [height, width], "triangles": T,
                       "handover": args.handover, "gather": args.gather},
            # the step's forward kernel: ids + barycentrics (16 B/px; render() does not ask for the
            # depth plane) + RGBA (16 B/px) written, clip-space vertices, triangle list and the
            # per-triangle attribute records (128 B) read
            "roofline": roofline("k_raster<shade> (forward: ids + barycentrics + shaded RGBA write)",
                                 px * 32 + batch * V * 16 + T * 12 + batch * T * 128,
                                 ev_raster.mean_ms(n_ev), "k_raster_shade", args.config),
            "roofline_gbuffer": dict(
                roofline("k_raster (G-buffer write: ids + barycentrics + depth, 20 B/px, through mr_rasterize_forward)",
                         px * 20 + batch * V * 16 + T * 12, gbuffer_ms, "k_raster", args.config),
                timed="OUTSIDE the timed region, BEFORE the warm-up steps: %d calls of mr_rasterize_forward (after %d untimed "
                      "ones), each launching its k_raster kernel %d times back to back inside ONE hipEvent pair "
                      "(mr_debug_set_raster_repeat); avg_kernel_ms = pair time / %d" % (n_gb, n_gb_lead, n_gb_rep, n_gb_rep)),
            "roofline_gbuffer_in_step": dict(
                roofline("k_raster (the same kernel, one launch per hipEvent pair inside the step with the shading epilogue off)",
                         px * 20 + batch * V * 16 + T * 12, gbuffer_in_step_ms, "k_raster", args.config),
                timed="OUTSIDE the timed region, BEFORE the warm-up steps: %d steps (after %d untimed ones) with "
                      "rasterize_triangles_ext.shading_epilogue(False); an event pair costs the stream ~5 us around a single "
                      "launch (rounds 2-4 reported this figure)" % (n_gs, n_gs_lead)),
            # ids + barycentrics (16 B/px) and the loss's sign codes (1 B/px) read, the triangles'
            # difference-basis records (FoldRec, 160 B) read
            "roofline_shade_backward": roofline(
                "k_accumulate_lanes<ShadeFoldLaneFn> (fused shading backward, pixel pass: vertex gradients only, "
                "clip-space pull-back folded in: 9 sums per triangle kept in registers down each lane's vertical run)",
                px * 17 + batch * T * 160, ev_shade.mean_ms(n_ev), "shade_backward", args.config),
            # the loss: image and target read (2 x 16 B/px), the sign codes written (1 B/px)
            "roofline_l1_forward": dict(
                roofline("k_l1_forward_regions (mean |image - target| and its sign codes, one streaming pass; 64 x 64 blocks "
                         "that the renderer's and the target's empty-block maps both mark are not read)",
                         l1_bytes, ev_l1.mean_ms(n_ev), "l1_forward", args.config),
                nominal_bytes=px * 33, blocks_read_fraction=round(l1_read_fraction, 4),
                note="`achieved` counts the bytes of the blocks that ARE read (32 B/px) plus the sign codes written for every "
                     "pixel (1 B/px); nominal_bytes is the figure earlier rounds divided by the same time"),
        }
        line.update(extras)
        if LEG_GROUPS_MS:
            line["leg_groups_ms"] = LEG_GROUPS_MS   # the three timed groups behind each leg's (median) figure
        line["host_gc"] = {"policy": "CPython's cyclic collector is paused inside every timed region after one full collection "
                                     "(timeit's convention; bench.py: collector_paused)",
                           "full_collection_ms": GC_FULL_PASS_MS[:12], "tracked_objects": len(gc.get_objects())}
        if grouped:
            line["rccl"] = {"backend": torch.distributed.get_backend(), "ranks": torch.distributed.get_world_size(),
                            "device_per_rank": "cuda:%d of %d visible" % (device.index, torch.cuda.device_count()),
                            "handover_bytes_per_rank_per_step": px * (4 if args.handover == "u8" else 16)}
            # the same step with the hand-over off (after the timed region): rendering alone, max over ranks
            line["ms_per_step_render_only"] = round(render_only_ms, 4)
            line["ms_per_step_with_handover"] = line["ms_per_step"]
            # the same two figures as the line's own units, so that a scaling curve can be read off either one:
            # `value` contains the hand-over of every step's frames, value_render_only does not
            line["value_render_only"] = round(world * px / render_only_ms / 1e3, 2)
            if rotating:
                line["rccl"]["gather"] = ("rotating root: the frames of step s (global batch) land on rank s mod N; one "
                                          "all_to_all per block of N steps on a side stream, depth 2")
                line["rccl"]["transport"] = args.transport
                # what every rank sends (and receives) per second while the timed loop runs: (N - 1) / N of a shard per step,
                # spread over its N - 1 links
                line["handover_GBps_out_of_each_rank"] = round(
                    (world - 1) / world * line["rccl"]["handover_bytes_per_rank_per_step"] / (elapsed / args.steps) / 1e9, 2)
            else:
                line["rccl"]["gather"] = "every step's frames to rank 0 (one gather per step, side stream, depth 2)"
                # what arrives at the root per second while the timed loop runs (N - 1 shards per step)
                line["handover_GBps_into_root"] = round(
                    (world - 1) * line["rccl"]["handover_bytes_per_rank_per_step"] / (elapsed / args.steps) / 1e9, 2)
            line["handover_depth"] = gather.depth
            # the other image type through the same loop, after the timed region (f32 is what render() returns and what
            # SURVEY.md row E sizes; u8 is the reference examples' frame conversion, written by the forward's epilogue)
            line["ms_per_step_handover_" + other_handover] = round(other_handover_ms, 4)
            line["value_handover_" + other_handover] = round(world * px / other_handover_ms / 1e3, 2)
            line["value_handover_" + args.handover] = line["value"]
        if world == 1 and args.cpu_sample > 0:
            line["cpu_baseline"] = cpu_baseline(batch, width, height, sphere_k, args.cpu_sample)
        print(json.dumps(line), flush=True)

    if gather is not None and hasattr(gather, "close"):
        gather.close()   # (peer transport: the mapped views of the other ranks' buffers go before anybody frees a buffer)
    if grouped:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
