"""Headline benchmark: Mpixels/s forward+backward, 5k-tri mesh, 1024x1024, batch 32.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one pass of the hot path over one batch of synthetic render jobs
(BASELINE.json configs[2]): mesh_renderer.render() forward (clip
transform, G-buffer rasterization, attribute interpolation, Phong shading), the L1
image loss mean|image - target| against a fixed target, and backward to the world-space vertex
positions.  All inputs are resident in HBM before the timed region starts.  With N
ranks every rank renders its own 32 jobs (weak scaling; no data-path collective) and
the finished images are handed over to rank 0 as 8-bit frames (mesh_renderer.to_uint8: the
conversion the reference's examples apply before writing a frame) with one RCCL gather per
step that overlaps the loss, the backward and the next forward.  Rank 0 prints ONE JSON line.

Extra objects in the line:
  roofline      the forward G-buffer kernel (k_raster): algorithmic bytes per launch
                (20 B/px written + clip and triangle reads) / its average duration,
                measured with HIP events recorded around that kernel on its stream.
  cpu_baseline  the same step on the host cores for a bounded sample of the batch:
                torch-CPU eager restatement of the reference's render path
                (oracle/shading.py) over the reference's own compiled C++ kernel
                (oracle/_ref) when present, else over oracle/mr_oracle.c.
"""
import argparse
import ctypes
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from pytorch_mesh_renderer_amd import _native, distributed, mesh_renderer  # noqa: E402
from pytorch_mesh_renderer_amd.common import synthetic  # noqa: E402

BATCH, WIDTH, HEIGHT, SPHERE_K = 32, 1024, 1024, 50   # BASELINE.json configs[2]
HBM_PEAK_GBPS = 8000.0                                # MI355X HBM3E spec (MI355X_MICROARCH.md)


class HipEvents:
    """Raw hipEvent pairs (torch.cuda.Event only sees torch's bookkeeping)."""

    def __init__(self, n):
        self.hip = ctypes.CDLL("libamdhip64.so")
        self.hip.hipEventElapsedTime.argtypes = [ctypes.POINTER(ctypes.c_float), ctypes.c_void_p,
                                                 ctypes.c_void_p]
        self.pairs = []
        for _ in range(n):
            a, b = ctypes.c_void_p(), ctypes.c_void_p()
            assert self.hip.hipEventCreate(ctypes.byref(a)) == 0
            assert self.hip.hipEventCreate(ctypes.byref(b)) == 0
            self.pairs.append((a, b))

    def arm(self, i):
        a, b = self.pairs[i]
        _native.lib().mr_time_next_kernel(_native.TIMER_RASTER_FORWARD, a, b)

    @staticmethod
    def disarm():
        _native.lib().mr_time_next_kernel(_native.TIMER_RASTER_FORWARD, None, None)

    def elapsed_ms(self):
        out = []
        for a, b in self.pairs:
            ms = ctypes.c_float()
            if self.hip.hipEventElapsedTime(ctypes.byref(ms), a, b) == 0:
                out.append(ms.value)
        return out


def make_step(job, device, gather):
    tri = job["triangles"].to(device)
    vertices = job["vertices"].to(device).requires_grad_(True)
    normals, diffuse = job["normals"].to(device), job["diffuse"].to(device)
    eyes = job["eyes"]                      # cameras stay host tensors, as in the reference's usage
    center = torch.zeros_like(eyes)
    up = torch.tensor([0.0, 1.0, 0.0])
    lpos, lint = job["light_positions"].to(device), job["light_intensities"].to(device)

    def forward():
        return mesh_renderer.render(vertices, tri, normals, diffuse, eyes, center, up, lpos, lint,
                                    WIDTH, HEIGHT)

    # fixed target: the same scene with the mesh slightly rotated (mirrors the reference's
    # optimisation tests), rendered once outside the timed region
    with torch.no_grad():
        c, s = torch.cos(torch.tensor(0.2)), torch.sin(torch.tensor(0.2))
        rot = torch.tensor([[c, 0.0, s], [0.0, 1.0, 0.0], [-s, 0.0, c]], device=device)
        target = mesh_renderer.render(vertices @ rot.T, tri, normals @ rot.T, diffuse, eyes, center, up,
                                      lpos, lint, WIDTH, HEIGHT)

    def step():
        vertices.grad = None
        image = forward()
        if gather is not None:
            gather.wait()                # the previous step's hand-over (no-op the first time) ...
            # ... then this one: 8-bit frames (what the reference's examples write out); conversion
            # and transfer both run on the side stream, overlapping the loss, the backward and the
            # next step's forward
            gather.start(image, transform=mesh_renderer.to_uint8)
        loss = mesh_renderer.losses.l1_loss(image, target)   # mean |image - target|, one HIP pass each way
        loss.backward()
        return loss

    return step, vertices


def cpu_baseline(sample_images):
    """The same step (render fwd + L1 + bwd to vertices) on host cores, bounded sample."""
    import oracle
    from oracle import shading
    use_ref = oracle.have_reference_kernel()
    job = synthetic.sphere_job(BATCH, WIDTH, HEIGHT, SPHERE_K)
    sl = slice(0, sample_images)
    v = job["vertices"][sl].clone().requires_grad_(True)
    args = (job["triangles"], job["normals"][sl], job["diffuse"][sl], job["eyes"][sl],
            torch.zeros(sample_images, 3), torch.tensor([[0.0, 1.0, 0.0]]).repeat(sample_images, 1),
            job["light_positions"][sl], job["light_intensities"][sl], WIDTH, HEIGHT)
    target = torch.zeros(sample_images, HEIGHT, WIDTH, 4)
    t0 = time.perf_counter()
    img = shading.render(v, *args, use_reference_kernel=use_ref)
    torch.mean(torch.abs(img - target)).backward()
    dt = time.perf_counter() - t0
    return {
        "value": round(sample_images * WIDTH * HEIGHT / dt / 1e6, 4), "unit": "Mpixels/s",
        "cores": torch.get_num_threads(), "kind": "port",
        "sample": "%d of the %d images of the workload, render fwd+bwd, %.1f s; eager torch-CPU "
                  "restatement of the reference's render path over %s" % (
                      sample_images, BATCH, dt,
                      "the reference's compiled C++ kernel (oracle/_ref)" if use_ref
                      else "oracle/mr_oracle.c"),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--cpu-sample", type=int, default=12, help="images timed for cpu_baseline (0 = skip)")
    args = ap.parse_args()

    rank, world, local_rank = distributed.init_from_env()
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    # one GPU per rank; the modulo only matters for a gloo rehearsal of N ranks on fewer GPUs
    device = torch.device("cuda", local_rank % torch.cuda.device_count())
    torch.cuda.set_device(device)

    job = synthetic.sphere_job(BATCH, WIDTH, HEIGHT, SPHERE_K)
    if world > 1:  # every rank renders its own BATCH jobs: rotate the orbit per rank
        shift = (rank * 7) % BATCH
        job = {k: (torch.roll(v, shift, 0) if torch.is_tensor(v) and k != "triangles" else v)
               for k, v in job.items()}
    # images are handed over to rank 0 (RCCL gather): the root receives its N-1 shards over N-1
    # xGMI links at once; an all-gather would move N times the bytes for nothing
    gather = distributed.ImageGather(BATCH * world, mode="root") if world > 1 else None
    # ImageGather shards n_total evenly: each rank contributes exactly BATCH images
    step, vertices = make_step(job, device, gather)

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize(device)

    for _ in range(args.warmup):
        step()
    events = HipEvents(args.steps)
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        events.arm(i)
        step()
    if gather is not None:
        gather.wait()                    # the last step's hand-over belongs to the timed region
    barrier()
    elapsed = time.perf_counter() - t0
    events.disarm()

    if world > 1:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t.item())
    assert vertices.grad is not None and bool(torch.isfinite(vertices.grad).all())

    if rank == 0:
        raster_ms = events.elapsed_ms()
        V, T = job["vertices"].shape[1], job["triangles"].shape[0]
        algorithmic = BATCH * WIDTH * HEIGHT * 20 + BATCH * V * 16 + T * 12   # bytes per launch
        avg_ms = sum(raster_ms) / max(len(raster_ms), 1)
        achieved = algorithmic / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        traffic = None
        pmc_path = os.path.join(ROOT, "profiles", "raster_traffic.json")
        if os.path.exists(pmc_path):   # written from a separate rocprofv3 --pmc run of this command
            traffic = json.load(open(pmc_path)).get("bytes_per_launch")
        line = {
            "metric": "Mpixels/sec forward+backward, 1024x1024 batch=32",
            "value": round(world * BATCH * WIDTH * HEIGHT * args.steps / elapsed / 1e6, 2),
            "unit": "Mpixels/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "configs[2]: 5k-tri UV sphere (V=%d, T=%d), %dx%d, batch=%d per GPU, "
                                   "mesh_renderer.render forward + L1 loss + backward to vertex positions; "
                                   "8-bit frames gathered to rank 0 over RCCL when n_gpus>1" % (V, T, WIDTH, HEIGHT, BATCH),
                       "global_batch": BATCH * world, "image": [HEIGHT, WIDTH], "triangles": T},
            "roofline": {"bound": "hbm", "kernel": "k_raster (forward G-buffer write)",
                         "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": traffic,
                         "algorithmic_bytes": algorithmic, "avg_kernel_ms": round(avg_ms, 4)},
        }
        if world == 1 and args.cpu_sample > 0:
            line["cpu_baseline"] = cpu_baseline(args.cpu_sample)
        print(json.dumps(line), flush=True)

    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
