"""The benchmark step as a two-stream software pipeline over sub-batches (experiment).

    python tools/pipelined_step.py [--batch 32] [--size 1024] [--k 50] [--eager]

The step's three big kernels are bound by different things: the forward and the loss by HBM, the
shading backward by instruction issue (0.20 of the HBM peak).  Here the batch of independent render
jobs is cut into `parts` sub-batches; sub-batch i+1's forward + loss are enqueued on the other stream
and start when sub-batch i's backward starts, so that the two kinds of work share the chip.  The
whole step (all sub-batches, both streams, fork/join) is captured into ONE HIP graph and replayed.
Prints ms per full step for the plain step (also as a graph) and the pipelined one, and checks that
the gradients agree.  RESULT (one MI355X, ROCm 7.2): no gain -- 0.966 ms plain vs 1.105 ms with two
sub-batches as one graph; eager at batch 64 (GPU-bound from Python): 2.15 vs 2.32 ms.  Kernels of two
streams do not share the CUs finely enough for the instruction-bound backward to hide HBM-bound work
(DESIGN.md section 7).  (More than two sub-batches in one capture crashed the process at capture
time and is not offered.)
"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pytorch_mesh_renderer_amd import mesh_renderer
from pytorch_mesh_renderer_amd.common import synthetic


def build_parts(batch, size, k, device, parts):
    job = synthetic.sphere_job(batch, size, size, k)
    d = {key: (v.to(device) if torch.is_tensor(v) else v) for key, v in job.items()}
    up = torch.tensor([0.0, 1.0, 0.0], device=device)
    per = batch // parts
    out = []
    for p in range(parts):
        sl = slice(p * per, (p + 1) * per)
        vertices = d["vertices"][sl].clone().requires_grad_(True)
        args = (d["triangles"], d["normals"][sl].contiguous(), d["diffuse"][sl].contiguous(), d["eyes"][sl].contiguous(),
                torch.zeros_like(d["eyes"][sl]), up, d["light_positions"][sl].contiguous(),
                d["light_intensities"][sl].contiguous(), size, size)
        with torch.no_grad():
            c, s = torch.cos(torch.tensor(0.2)), torch.sin(torch.tensor(0.2))
            rot = torch.tensor([[c, 0.0, s], [0.0, 1.0, 0.0], [-s, 0.0, c]], device=device)
            target = mesh_renderer.render(vertices @ rot.T, args[0], args[1] @ rot.T, *args[2:])
        out.append({"vertices": vertices, "args": args, "target": target, "weight": 1.0 / parts})
    return out


def run_parts(parts, streams):
    """One full step: sub-batch i on streams[i % 2]; its forward waits for sub-batch i-1's loss."""
    main = torch.cuda.current_stream()
    for s in streams:
        s.wait_stream(main)
    loss_done = None
    losses = []
    for i, part in enumerate(parts):
        s = streams[i % len(streams)]
        with torch.cuda.stream(s):
            if loss_done is not None:
                s.wait_event(loss_done)
            part["vertices"].grad = None
            image = mesh_renderer.render(part["vertices"], *part["args"])
            loss = mesh_renderer.losses.l1_loss(image, part["target"]) * part["weight"]
            loss_done = torch.cuda.Event()
            loss_done.record(s)
            loss.backward()
            losses.append(loss)
    for s in streams:
        main.wait_stream(s)
    return losses


def capture(parts, streams):
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            run_parts(parts, streams)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        losses = run_parts(parts, streams)
    return graph, losses


def timed(graph, iters):
    for _ in range(5):
        graph.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        graph.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--size", type=int, default=1024)
    ap.add_argument("--k", type=int, default=50)
    ap.add_argument("--iters", type=int, default=100)
    ap.add_argument("--eager", action="store_true", help="no graphs: enqueue from Python (needs a batch large enough to stay GPU-bound)")
    args = ap.parse_args()
    device = torch.device("cuda:0")
    px = args.batch * args.size * args.size
    if args.eager:
        class Eager:
            def __init__(self, parts, streams): self.parts, self.streams = parts, streams
            def replay(self): self.losses = run_parts(self.parts, self.streams)
        global capture
        def capture(parts, streams):
            e = Eager(parts, streams)
            e.replay()
            return e, e.losses
    whole = build_parts(args.batch, args.size, args.k, device, 1)
    g1, l1 = capture(whole, [torch.cuda.Stream()])
    t1 = timed(g1, args.iters)
    ref_grad = whole[0]["vertices"].grad.clone()
    print("plain step (one graph, one stream): %.4f ms -> %.0f Mpix/s, loss %.9g" % (t1 * 1e3, px / t1 / 1e6, float(l1[0])))
    for n in (2,):
        parts = build_parts(args.batch, args.size, args.k, device, n)
        g, losses = capture(parts, [torch.cuda.Stream(), torch.cuda.Stream()])
        t = timed(g, args.iters)
        grad = torch.cat([p["vertices"].grad for p in parts], 0) 
        err = float((grad - ref_grad).abs().max()) / float(ref_grad.abs().max())
        print("%d sub-batches on two streams: %.4f ms -> %.0f Mpix/s, loss %.9g, max grad diff %.2e (relative)"
              % (n, t * 1e3, px / t / 1e6, float(sum(float(x) for x in losses)), err))


if __name__ == "__main__":
    main()
