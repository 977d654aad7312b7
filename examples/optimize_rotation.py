"""Recover a cube's rotation from a target image -- counterpart of src/examples/example5.py.

    python examples/optimize_rotation.py --out /tmp/frames [--steps 35]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from PIL import Image

from pytorch_mesh_renderer_amd import mesh_renderer
from pytorch_mesh_renderer_amd.common import camera_utils, shapes


def optimize(steps=35, width=320, height=240, device="cuda:0", out=None):
    device = torch.device(device)
    cube_v, triangles, cube_n = [t.to(device) for t in shapes.cube(2.0)]
    eye = torch.tensor([[0.0, 0.0, 6.0]])
    center, up = torch.zeros(1, 3), torch.tensor([[0.0, 1.0, 0.0]])
    light_positions = torch.tensor([[[0.0, 0.0, 6.0]]], device=device)
    light_intensities = torch.ones(1, 1, 3, device=device)
    diffuse = torch.ones(1, 8, 3, device=device)

    def render(angles):
        rot = camera_utils.euler_matrices(angles)[0, :3, :3]
        return mesh_renderer.render((cube_v @ rot.T).unsqueeze(0), triangles, (cube_n @ rot.T).unsqueeze(0),
                                    diffuse, eye, center, up, light_positions, light_intensities, width, height)

    # the reference's own optimisation test and example feed these numbers to euler_matrices as they
    # are (mesh_renderer_test.py:231-236, example5.py:37-40) and start from no rotation
    target_angles = torch.tensor([[-20.0, 0.0, 60.0]], device=device)
    with torch.no_grad():
        target = render(target_angles)
    angles = torch.zeros(1, 3, device=device, requires_grad=True)
    optimizer = torch.optim.SGD([angles], lr=0.7, momentum=0.1)            # example5.py:59
    losses = []
    for step in range(steps):
        optimizer.zero_grad()
        image = render(angles)
        loss = mesh_renderer.losses.l1_loss(image, target)
        loss.backward()
        torch.nn.utils.clip_grad_norm_([angles], 1.0)                      # mesh_renderer_test.py:258
        optimizer.step()
        losses.append(float(loss.detach()))
        if out is not None and step % 10 == 0:
            Image.fromarray(mesh_renderer.to_uint8(image)[0].cpu().numpy(), "RGBA").save(
                os.path.join(out, "rotation_%03d.png" % step))
    return losses, angles.detach().cpu(), target_angles.cpu()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="frames")
    ap.add_argument("--steps", type=int, default=35)
    args = ap.parse_args()
    os.makedirs(args.out, exist_ok=True)
    losses, angles, target = optimize(args.steps, out=args.out)
    print("loss %.5f -> %.5f; angles %s (target %s)" % (losses[0], losses[-1], angles.tolist(), target.tolist()))


if __name__ == "__main__":
    main()
