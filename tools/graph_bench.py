"""Launch-bound configurations (small images) with and without a captured HIP graph.

    python tools/graph_bench.py [--batch 8] [--size 256] [--k 50]

The whole optimisation step (render forward, L1 loss, backward) is captured once with
torch.cuda.CUDAGraph (hipGraph on ROCm) and replayed: ~25 kernel launches and the Python above them
collapse into one graph launch.  Cameras must live on the device (host-side camera math cannot be
captured); vertices are updated in place between replays.
"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pytorch_mesh_renderer_amd import mesh_renderer
from pytorch_mesh_renderer_amd.common import synthetic


def build(batch, size, k, device):
    job = synthetic.sphere_job(batch, size, size, k)
    d = {key: (v.to(device) if torch.is_tensor(v) else v) for key, v in job.items()}
    vertices = d["vertices"].clone().requires_grad_(True)
    center = torch.zeros_like(d["eyes"])
    up = torch.tensor([0.0, 1.0, 0.0], device=device)

    def render():
        return mesh_renderer.render(vertices, d["triangles"], d["normals"], d["diffuse"], d["eyes"], center, up,
                                    d["light_positions"], d["light_intensities"], size, size)
    with torch.no_grad():
        target = render().roll(3, 2).contiguous()

    def step():
        loss = mesh_renderer.losses.l1_loss(render(), target)
        loss.backward()
        return loss
    return vertices, step


def capture(vertices, step):
    """Standard whole-step capture: warm up on a side stream, then capture fwd + bwd with static grads."""
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            vertices.grad = None
            step()
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    vertices.grad = None
    with torch.cuda.graph(graph):
        loss = step()
    return graph, loss


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--k", type=int, default=50)
    ap.add_argument("--iters", type=int, default=50)
    args = ap.parse_args()
    device = torch.device("cuda:0")
    vertices, step = build(args.batch, args.size, args.k, device)

    def eager():
        vertices.grad = None
        return step()
    for _ in range(5): eager()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(args.iters): eager()
    torch.cuda.synchronize(); t_eager = (time.perf_counter() - t0) / args.iters
    eager_loss, eager_grad = float(eager()), vertices.grad.clone()

    graph, loss = capture(vertices, step)
    for _ in range(5): graph.replay()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(args.iters): graph.replay()
    torch.cuda.synchronize(); t_graph = (time.perf_counter() - t0) / args.iters
    err = float((vertices.grad - eager_grad).abs().max())
    px = args.batch * args.size * args.size
    print(f"B={args.batch} {args.size}x{args.size} k={args.k}: eager {t_eager*1e3:.3f} ms ({px/t_eager/1e6:.0f} Mpix/s), "
          f"graph replay {t_graph*1e3:.3f} ms ({px/t_graph/1e6:.0f} Mpix/s); loss {eager_loss:.6f} vs {float(loss):.6f}, "
          f"max |grad diff| {err:.2e}")


if __name__ == "__main__":
    main()
