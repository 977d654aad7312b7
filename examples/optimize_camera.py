"""Recover a camera (eye position and orientation) from a target image -- counterpart of
src/examples/example4.py:58-89 (and, with the mesh rotated instead of the camera, example6.py:52-94).

    python examples/optimize_camera.py --out /tmp/frames [--steps 50]

The reference keeps the camera on the host and differentiates through look_at / perspective with
torch autograd; here the camera tensors live on the MI355X next to the mesh, common.camera_utils is
device-aware, and the whole step -- camera matrices, render, L1 loss, backward -- stays on the device.
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from PIL import Image

from pytorch_mesh_renderer_amd import mesh_renderer
from pytorch_mesh_renderer_amd.common import camera_utils, shapes


def scene(device):
    """A sphere squashed into an ellipsoid with a bump: no symmetry the optimiser could hide in."""
    vertices, triangles, _ = shapes.sphere(1.0, 16)
    vertices = vertices * torch.tensor([1.0, 0.7, 0.5]) + 0.25 * torch.exp(
        -4.0 * ((vertices - torch.tensor([0.6, 0.5, 0.3])) ** 2).sum(1, keepdim=True)) * vertices
    from pytorch_mesh_renderer_amd.common import meshes
    vertices, triangles = vertices.to(device), triangles.to(device)
    normals = meshes.compute_vertex_normals(vertices.unsqueeze(0), triangles)
    return vertices.unsqueeze(0), triangles, normals


def optimize(steps=50, width=160, height=120, device="cuda:0", out=None):
    device = torch.device(device)
    vertices, triangles, normals = scene(device)
    diffuse = torch.ones_like(vertices)
    light_positions = torch.tensor([[[0.0, 3.0, 0.0]]], device=device)        # example4.py:44
    light_intensities = torch.ones(1, 1, 3, device=device)
    initial_eye = torch.tensor([0.0, 3.0, 3.0], device=device)                # example4.py:50-51
    initial_world_up = torch.tensor([0.0, 3.0, -3.0], device=device)

    def render(eye, euler_angles):
        rot = camera_utils.euler_matrices(euler_angles)[0, :3, :3]            # example4.py:60-64
        forward = torch.reshape(torch.matmul(-initial_eye, rot.T), [1, 3])
        world_up = torch.reshape(torch.matmul(initial_world_up, rot.T), [1, 3])
        return mesh_renderer.render(vertices, triangles, normals, diffuse, eye, eye + forward, world_up,
                                    light_positions, light_intensities, width, height)

    target_eye = torch.tensor([[0.5, 2.6, 3.3]], device=device)
    target_angles = torch.tensor([[0.05, -0.1, 0.0]], device=device)
    with torch.no_grad():
        target = render(target_eye, target_angles)
    eye = initial_eye.clone().unsqueeze(0).requires_grad_(True)
    angles = torch.zeros(1, 3, device=device, requires_grad=True)
    # example4.py:57 uses SGD(0.7, 0.1) with the gradient norm clipped to 1: steps of up to 0.7 scene
    # units, fine for its teapot; this smaller object leaves the frame (and every gradient with it)
    # after two such steps, hence the smaller rate
    optimizer = torch.optim.SGD([eye, angles], 0.05, 0.1)
    losses = []
    for step in range(steps):
        optimizer.zero_grad()
        image = render(eye, angles)
        loss = mesh_renderer.losses.l1_loss(image, target)                    # example4.py:79
        loss.backward()
        torch.nn.utils.clip_grad_norm_([eye, angles], 1.0)                    # example4.py:81
        optimizer.step()
        losses.append(float(loss.detach()))
        if out is not None and step % 10 == 0:
            Image.fromarray(mesh_renderer.to_uint8(image)[0].cpu().numpy(), "RGBA").save(
                os.path.join(out, "camera_%03d.png" % step))
    return losses, eye.detach().cpu(), target_eye.cpu(), angles.detach().cpu(), target_angles.cpu()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="frames")
    ap.add_argument("--steps", type=int, default=50)
    args = ap.parse_args()
    os.makedirs(args.out, exist_ok=True)
    losses, eye, target_eye, angles, target_angles = optimize(args.steps, out=args.out)
    print("loss %.5f -> %.5f; eye %s (target %s); angles %s (target %s)" % (
        losses[0], losses[-1], eye.tolist(), target_eye.tolist(), angles.tolist(), target_angles.tolist()))


if __name__ == "__main__":
    main()
