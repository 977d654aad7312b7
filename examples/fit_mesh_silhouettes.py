"""Deform a sphere until its silhouettes match four target views -- counterpart of the mesh-fitting
loop of src/examples/example7b.py:219-285 (soft renderer, silhouette MSE + edge-length + uniform
Laplacian regularisers, SGD lr 4 / momentum 0.1, gradient-norm clipping).

    python examples/fit_mesh_silhouettes.py --out /tmp/frames [--steps 200]

The four views share ONE vertex set: they form a batch of four jobs for the soft renderer, and the
vertex gradient is the sum over the views (on several GPUs: distributed.allreduce_shared_mesh_grad).
The regularisers run on the device as well (a sparse Laplacian built once from the mesh's edges).
The reference reads its targets from PNG files; here they are rendered from a known ellipsoid so
that the script needs no data files and the result can be checked.
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from PIL import Image

from pytorch_mesh_renderer_amd import mesh_renderer, soft_mesh_renderer
from pytorch_mesh_renderer_amd.common import shapes


def compute_edges_list(faces):
    """Unique undirected edges [E,2] of a triangle list (example7b.py:80-98)."""
    edges = torch.cat([faces[:, :2], faces[:, 1:], faces[:, ::2]]).long()
    edges = torch.sort(edges, dim=1).values
    return torch.unique(edges, dim=0)


def compute_laplacian(vertex_count, edges):
    """Uniform graph Laplacian as a sparse [V,V] matrix: L[i,j] = 1/deg(i) on edges, -1 on the
    diagonal (example7b.py:17-78)."""
    e0, e1 = edges.unbind(1)
    idx = torch.cat([torch.stack([e0, e1]), torch.stack([e1, e0])], dim=1)
    deg = torch.zeros(vertex_count, device=edges.device).index_add_(0, idx[0], torch.ones(idx.shape[1], device=edges.device))
    inv = torch.where(deg > 0, 1.0 / deg, deg)
    diag = torch.arange(vertex_count, device=edges.device)
    indices = torch.cat([idx, torch.stack([diag, diag])], dim=1)
    values = torch.cat([inv[idx[0]], -torch.ones(vertex_count, device=edges.device)])
    return torch.sparse_coo_tensor(indices, values, (vertex_count, vertex_count)).coalesce()


def mesh_laplacian_smoothing_loss(vertices, laplacian):       # example7b.py:100-114
    return (torch.sparse.mm(laplacian, vertices).norm(dim=1) / vertices.shape[0]).sum()


def mesh_edge_loss(vertices, edges):                          # example7b.py:116-129
    return (vertices[edges[:, 0]] - vertices[edges[:, 1]]).norm(dim=1, p=2).mean()


def optimize(steps=200, size=96, resolution=12, device="cuda:0", out=None):
    device = torch.device(device)
    vertices, triangles, _ = shapes.sphere(1.0, resolution)
    vertices, triangles = vertices.to(device), triangles.to(device)
    edges = compute_edges_list(triangles)
    laplacian = compute_laplacian(vertices.shape[0], edges)
    eye = torch.tensor([[0.0, 0.0, -3.0], [3.0, 0.0, 0.0], [-3.0, 0.0, 0.0], [0.0, 0.0, 3.0]], device=device)
    center = torch.zeros_like(eye)                                                  # example7b.py:150-163
    world_up = torch.tensor([[0.0, 1.0, 0.0]] * 4, device=device)
    light_positions = torch.tensor([[[0.0, 0.0, -3.0], [0.0, 3.0, 0.0], [0.0, 0.0, 3.0]]] * 4, device=device)
    light_intensities = torch.ones(4, 3, device=device)
    diffuse = torch.ones(4, vertices.shape[0], 3, device=device)

    def render(v):
        return soft_mesh_renderer.render(torch.stack([v] * 4, dim=0), triangles, diffuse, eye, center, world_up,
                                         light_positions, light_intensities, size, size, sigma_val=1e-4,
                                         fov_y=60.0, blur_radius=0.1)               # example7b.py:232-246

    target_shape = torch.tensor([0.65, 1.0, 0.8], device=device)
    with torch.no_grad():
        target_alpha = render(vertices * target_shape)[..., 3]
    v = vertices.clone().requires_grad_(True)
    optimizer = torch.optim.SGD([v], 4.0, 0.1)                                      # example7b.py:223-227
    losses = []
    for step in range(steps):
        optimizer.zero_grad()
        renders = render(v)
        silhouette = torch.mean((renders[..., 3] - target_alpha) ** 2)              # example7b.py:248
        loss = silhouette + 0.1 * mesh_edge_loss(v, edges) + 0.1 * mesh_laplacian_smoothing_loss(v, laplacian)
        loss.backward()
        torch.nn.utils.clip_grad_norm_([v], 1.0)                                    # example7b.py:253
        optimizer.step()
        losses.append(float(silhouette.detach()))
        if out is not None and step % 50 == 0:
            Image.fromarray(mesh_renderer.to_uint8(renders)[0].cpu().numpy(), "RGBA").save(
                os.path.join(out, "fit_%04d.png" % step))
    extent = (v.detach().max(0).values - v.detach().min(0).values).cpu() / 2.0
    return losses, extent, target_shape.cpu()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="frames")
    ap.add_argument("--steps", type=int, default=200)
    args = ap.parse_args()
    os.makedirs(args.out, exist_ok=True)
    losses, extent, target = optimize(args.steps, out=args.out)
    print("silhouette loss %.5f -> %.5f; half extents %s (target %s)" % (losses[0], losses[-1], extent.tolist(),
                                                                       target.tolist()))


if __name__ == "__main__":
    main()
