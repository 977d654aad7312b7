"""Render a lit cube to a PNG -- counterpart of the reference's src/examples/example1.py.

    python examples/render_cube.py --out /tmp/frames [--specular]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from PIL import Image

from pytorch_mesh_renderer_amd import mesh_renderer
from pytorch_mesh_renderer_amd.common import camera_utils, shapes


def render_cube(width=640, height=480, specular=False, device="cuda:0"):
    device = torch.device(device)
    vertices, triangles, normals = [t.to(device) for t in shapes.cube(2.0)]
    # rotate the cube a little so that three faces are visible (example1.py:28-33)
    rotation = camera_utils.euler_matrices(torch.tensor([[-20.0, 0.0, 60.0]], device=device) * 3.14159265 / 180.0)
    rotation = rotation[0, :3, :3]
    vertices = (vertices @ rotation.T).unsqueeze(0)
    normals = (normals @ rotation.T).unsqueeze(0)
    diffuse = torch.ones_like(vertices)
    eye = torch.tensor([[0.0, 0.0, 6.0]])
    center, up = torch.zeros(1, 3), torch.tensor([[0.0, 1.0, 0.0]])
    light_positions = torch.tensor([[[0.0, 0.0, 6.0]]], device=device)
    light_intensities = torch.ones(1, 1, 3, device=device)
    kwargs = {}
    if specular:
        kwargs = dict(specular_colors=torch.full_like(vertices, 0.5), shininess_coefficients=6.0)
    image = mesh_renderer.render(vertices, triangles, normals, diffuse, eye, center, up,
                                 light_positions, light_intensities, width, height, **kwargs)
    return mesh_renderer.to_uint8(image)[0]      # [H, W, 4] uint8, row 0 = top


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="frames")
    ap.add_argument("--specular", action="store_true")
    args = ap.parse_args()
    os.makedirs(args.out, exist_ok=True)
    frame = render_cube(specular=args.specular).cpu().numpy()
    path = os.path.join(args.out, "cube.png")
    Image.fromarray(frame, "RGBA").save(path)
    print("wrote", path, frame.shape)


if __name__ == "__main__":
    main()
