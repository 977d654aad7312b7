"""Fit a sphere to a target silhouette with the soft renderer -- the soft-renderer half of
src/examples/example7b.py.

    python examples/soft_silhouette.py --out /tmp/frames [--steps 40]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from PIL import Image

from pytorch_mesh_renderer_amd import mesh_renderer, soft_mesh_renderer
from pytorch_mesh_renderer_amd.common import shapes


def optimize(steps=40, size=128, device="cuda:0", out=None):
    device = torch.device(device)
    vertices, triangles, _ = shapes.sphere(1.0, 8)
    vertices, triangles = vertices.to(device), triangles.to(device)
    diffuse = torch.ones(1, vertices.shape[0], 3, device=device)
    eye = torch.tensor([[0.0, 0.0, 3.0]], device=device)
    center, up = torch.zeros(1, 3, device=device), torch.tensor([0.0, 1.0, 0.0], device=device)
    light_positions = torch.tensor([[[0.0, 2.0, 3.0]]], device=device)
    light_intensities = torch.ones(1, 1, device=device)

    def render(v):
        return soft_mesh_renderer.render(v.unsqueeze(0), triangles, diffuse, eye, center, up,
                                         light_positions, light_intensities, size, size)

    with torch.no_grad():   # target: the same sphere squashed along y
        target_alpha = render(vertices * torch.tensor([1.0, 0.6, 1.0], device=device))[..., 3]
    scale = torch.ones(3, device=device, requires_grad=True)
    optimizer = torch.optim.Adam([scale], lr=0.05)
    losses = []
    for step in range(steps):
        optimizer.zero_grad()
        image = render(vertices * scale)
        loss = torch.mean((image[..., 3] - target_alpha) ** 2)
        loss.backward()
        optimizer.step()
        losses.append(float(loss.detach()))
        if out is not None and step % 10 == 0:
            Image.fromarray(mesh_renderer.to_uint8(image)[0].cpu().numpy(), "RGBA").save(
                os.path.join(out, "soft_%03d.png" % step))
    return losses, scale.detach().cpu()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="frames")
    ap.add_argument("--steps", type=int, default=40)
    args = ap.parse_args()
    os.makedirs(args.out, exist_ok=True)
    losses, scale = optimize(args.steps, out=args.out)
    print("loss %.5f -> %.5f; scale %s (target [1, 0.6, 1])" % (losses[0], losses[-1], scale.tolist()))


if __name__ == "__main__":
    main()
