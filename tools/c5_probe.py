"""Where does the SoftRas step (BASELINE configs[4]) spend its time, alone and in bench.py's position?

VERDICT r5 weak 2: tools/soft_bench.py reads 0.78 ms/step, bench.py's c5 leg (which runs AFTER the configs[3]-shaped
leg) 2.5-3.3 ms.  This probe times the same step (a) in a fresh process state, (b) after a configs[3]-shaped step has
grown the per-stream scratch tensor and the allocator's pool, (c) after dropping that scratch again -- host wall clock
per step next to HIP-event time per step, with the scratch size the soft kernels were handed.

  python tools/c5_probe.py [--steps 30]            # prints one line per context
  rocprofv3 --kernel-trace --stats -d DIR -- python3 tools/c5_probe.py   # kernel-level view of the same contexts
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from pytorch_mesh_renderer_amd import _native, mesh_renderer, soft_mesh_renderer  # noqa: E402
from pytorch_mesh_renderer_amd.common import synthetic  # noqa: E402
from pytorch_mesh_renderer_amd.mesh_renderer import losses  # noqa: E402


def timed(fn, n, lead):
    for _ in range(lead):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    t_host = (time.perf_counter() - t0) / n * 1e3     # host time to ENQUEUE n steps
    torch.cuda.synchronize()
    t_wall = (time.perf_counter() - t0) / n * 1e3
    return t_wall, e0.elapsed_time(e1) / n, t_host


def scratch_bytes():
    return {k: v.numel() for k, v in _native._workspaces.items()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--lead", type=int, default=24)
    ap.add_argument("--smi", type=int, default=0,
                    help="1: after the three contexts, time the step again WHILE a child process polls rocm-smi in a loop -- "
                         "the driver samples rocm-smi every 5 s during its bench run (BENCH_r05.json: gpu_busy.samples); does "
                         "a poll that lands on this 40-ms leg explain a 3-4 x reading?")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    j5 = synthetic.sphere_job(16, 512, 512, 50)
    v5 = j5["vertices"].to(dev).requires_grad_(True)
    tri5, kd5, lp5 = j5["triangles"].to(dev), j5["diffuse"].to(dev), j5["light_positions"].to(dev)
    eyes5, zero5, up5 = j5["eyes"], torch.zeros(16, 3), torch.tensor([0.0, 1.0, 0.0])
    li5 = torch.ones(16, 1, device=dev)

    def step5():
        v5.grad = None
        soft_mesh_renderer.render(v5, tri5, kd5, eyes5, zero5, up5, lp5, li5, 512, 512).mean().backward()

    slow = []

    def report(tag):
        wall, gpu, host = timed(step5, args.steps, args.lead)
        if wall > 1.5:
            slow.append(tag)
        need = _native.lib().mr_soft_workspace_bytes(16, v5.shape[1], tri5.shape[0], 512, 512)
        print("%-46s wall %.3f ms/step  events %.3f  host enqueue %.3f | soft needs %.1f MB, scratch held %s MB, "
              "allocator reserved %.0f MB" % (tag, wall, gpu, host, need / 1e6,
                                               [round(b / 1e6, 1) for b in scratch_bytes().values()],
                                               torch.cuda.memory_reserved() / 1e6), flush=True)

    report("(a) fresh process")
    # (b) a configs[3]-shaped step first, as bench.py's extra legs run it
    j4 = synthetic.sphere_job(8, 2048, 2048, 158)
    v4 = j4["vertices"].to(dev).requires_grad_(True)
    tri4, nrm4, kd4 = j4["triangles"].to(dev), j4["normals"].to(dev), j4["diffuse"].to(dev)
    lp4, li4 = j4["light_positions"].to(dev), j4["light_intensities"].to(dev)
    eyes4, zero4 = j4["eyes"], torch.zeros_like(j4["eyes"])
    with torch.no_grad():
        target4 = torch.rand(8, 2048, 2048, 4, device=dev)

    def step4():
        v4.grad = None
        img = mesh_renderer.render(v4, tri4, nrm4, kd4, eyes4, zero4, up5, lp4, li4, 2048, 2048)
        losses.l1_loss(img, target4).backward()
    wall4, gpu4, host4 = timed(step4, args.steps, args.lead)
    print("    configs[3]-shaped step: wall %.3f events %.3f host %.3f" % (wall4, gpu4, host4), flush=True)
    del target4
    torch.cuda.empty_cache()
    report("(b) after the configs[3]-shaped leg")
    _native._workspaces.clear()
    torch.cuda.empty_cache()
    report("(c) scratch dropped, allocator emptied")
    if args.smi:
        import subprocess
        import threading
        stop = threading.Event()
        polls = []

        def poll():
            while not stop.is_set():
                t0 = time.perf_counter()
                try:
                    subprocess.run(["rocm-smi", "--showuse", "--showmemuse", "--showpower", "--showclocks", "--json"],
                                   capture_output=True, timeout=60)
                except Exception as exc:
                    polls.append(str(exc))
                    return
                polls.append(round(time.perf_counter() - t0, 3))
        th = threading.Thread(target=poll, daemon=True)
        th.start()
        time.sleep(0.3)
        for i in range(6):
            wall, gpu, host = timed(step5, 20, 4)
            print("while rocm-smi polls, chunk %d of 20 steps: wall %.3f events %.3f host %.3f" % (i, wall, gpu, host), flush=True)
        stop.set()
        th.join(70)
        print("rocm-smi poll durations (s): %s" % polls, flush=True)
        report("(d) after the polling stopped")
    if slow:   # a slow box at last: what state is the GPU in, and does the step recover under sustained load?
        import subprocess
        print("SLOW in: %s" % slow, flush=True)
        for flags in (["--showclocks", "--showpower", "--showperflevel"], ["--showuse", "--showmemuse", "--showtemp"]):
            try:
                print(subprocess.run(["rocm-smi"] + flags, capture_output=True, text=True, timeout=60).stdout, flush=True)
            except Exception as exc:
                print("rocm-smi %s: %s" % (flags, exc))
        for i in range(12):
            wall, gpu, host = timed(step5, 100, 0)
            print("sustained chunk %2d: wall %.3f events %.3f host %.3f" % (i, wall, gpu, host), flush=True)


if __name__ == "__main__":
    main()
