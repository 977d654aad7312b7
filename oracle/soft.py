"""CPU restatement of the reference's SoftRas renderer (TEST INFRASTRUCTURE ONLY).

Restates src/soft_mesh_renderer/rasterize.py:212-424 (rasterize_batch) and :14-110
(rasterize), src/soft_mesh_renderer/render.py:15-165 (render) and
src/common/meshes.py:3-35 (compute_vertex_normals) with torch CPU ops, vectorised over
(pixel, triangle) pairs instead of the reference's Python loops and quadtree.  The quadtree
is only an index: the candidate set of a pixel is exactly the triangles whose
blur-inflated NDC bbox contains the pixel centre, inclusive
(src/soft_mesh_renderer/quadtree.py:18-31), which is what is evaluated here.
Gradients come from torch autograd, as in the reference.

Pinned by tests/golden/soft_*.npz (generated from the reference by tools/make_goldens.py)
and by the reference's own known-answer matrices (test_rasterize.py:128-215).
"""
import torch

from . import shading

EPS = 1e-10  # src/soft_mesh_renderer/rasterize.py:211


def _nearest_on_segment(p, a, b):
    """p [P,1,2], a/b [1,T,2] -> (squared distance [P,T], t [P,T]); rasterize.py:169-176."""
    ab = b - a
    length = torch.linalg.vector_norm(ab, ord=2, dim=-1)                   # [1,T]
    n = ab / torch.clamp(length, min=1e-12).unsqueeze(-1)
    proj = ((p - a) * n).sum(-1, keepdim=True) * n                          # [P,T,2]
    t = torch.clamp((proj * n).sum(-1) / length, 0.0, 1.0)                  # [P,T]
    x = a + t.unsqueeze(-1) * ab
    d = x - p
    return (d * d).sum(-1), t


def rasterize_batch(clip, triangles, world, normals, diffuse, light_positions, light_intensities,
                    width, height, sigma_val, gamma_val, blur_radius=0.01, window=None):
    """One image: clip [V,4], triangles [T,3], world/normals/diffuse [V,3], lights [L,3]/[L]
    -> [H,W,4].  window = (x0, y0, w, h): only those pixels of the width x height image are
    evaluated (-> [h,w,4]) -- the dense [pixels x triangles] evaluation then fits a crop of a
    full-size image (tests/test_full_size_gpu.py); every pixel is independent of the others."""
    T = triangles.shape[0]
    tri = triangles.long()
    cv = clip[tri]                                   # [T,3,4]
    w = cv[:, :, 3]                                  # [T,3]
    ndc = cv[:, :, :3] / w.unsqueeze(-1)             # [T,3(vertex),3(xyz)]
    M = ndc.transpose(1, 2)                          # [T,3(xyz),3(vertex)]: columns are vertices
    M2d = M.clone()
    M2d[:, 2, :] = 1.0
    inv_list, ok = [], []
    for i in range(T):                               # per-triangle inverse, like :291-298
        try:
            inv_list.append(M2d[i].inverse())
            ok.append(True)
        except Exception:
            inv_list.append(torch.zeros(3, 3))
            ok.append(False)
    Minv = torch.stack(inv_list) if T else torch.zeros(0, 3, 3)
    ok = torch.tensor(ok, dtype=torch.bool) if T else torch.zeros(0, dtype=torch.bool)
    v0, v1, v2 = ndc[:, 0, :2], ndc[:, 1, :2], ndc[:, 2, :2]
    area = (v0 - v1)[:, 0] * (v2 - v1)[:, 1] - (v0 - v1)[:, 1] * (v2 - v1)[:, 0]   # :120-123,301
    area = torch.where(ok, area, torch.zeros_like(area))
    xs, ys = ndc[:, :, 0], ndc[:, :, 1]
    lo = torch.stack([xs.min(1).values - blur_radius, ys.min(1).values - blur_radius], -1)
    hi = torch.stack([xs.max(1).values + blur_radius, ys.max(1).values + blur_radius], -1)

    x0, y0, out_w, out_h = window if window is not None else (0, 0, width, height)
    yy, xx = torch.meshgrid(torch.arange(out_h), torch.arange(out_w), indexing="ij")
    px = torch.tensor([2.0 * ((x + 0.5) / width) - 1.0 for x in range(x0, x0 + out_w)], dtype=torch.float32)
    py = torch.tensor([-2.0 * ((y + 0.5) / height) + 1.0 for y in range(y0, y0 + out_h)], dtype=torch.float32)
    p2 = torch.stack([px[xx.reshape(-1)], py[yy.reshape(-1)]], -1)         # [P,2], row 0 = top
    P = p2.shape[0]
    p3 = torch.cat([p2, torch.ones(P, 1)], -1)

    cand = ((p2[:, None, 0] <= hi[None, :, 0].detach()) & (p2[:, None, 0] >= lo[None, :, 0].detach()) &
            (p2[:, None, 1] <= hi[None, :, 1].detach()) & (p2[:, None, 1] >= lo[None, :, 1].detach()))
    cand = cand & ok[None, :] & ~(area[None, :].detach() >= 0)             # back faces / zero area culled
    bc = torch.einsum("tij,pj->pti", Minv, p3)                              # [P,T,3]
    pp = p2[:, None, :]
    d01, t01 = _nearest_on_segment(pp, v0[None], v1[None])
    d12, t12 = _nearest_on_segment(pp, v1[None], v2[None])
    d20, t20 = _nearest_on_segment(pp, v2[None], v0[None])
    dist = torch.stack([d01, d12, d20], -1)
    sq_dist, which = dist.min(-1)
    zero, one = torch.zeros_like(t01), torch.ones_like(t01)
    bc_edge = torch.where((which == 0).unsqueeze(-1), torch.stack([one - t01, t01, zero], -1),
                          torch.where((which == 1).unsqueeze(-1), torch.stack([zero, one - t12, t12], -1),
                                      torch.stack([t20, zero, one - t20], -1)))
    inside = ~(bc.detach() < 0).any(-1)
    keep = cand & (inside | ~(sq_dist.detach() > blur_radius ** 2))
    u = torch.where(inside.unsqueeze(-1), bc, bc_edge) / w[None]
    sb = u / torch.clamp(u.abs().sum(-1, keepdim=True), min=1e-12)          # F.normalize(p=1)
    z = 0.5 - (sb * ndc[None, :, :, 2]).sum(-1) / 2.0
    keep = keep & ~((z.detach() < 0.0) | (z.detach() > 1.0))

    kd = torch.einsum("ptk,tkc->ptc", sb, diffuse[tri])
    pos = torch.einsum("ptk,tkc->ptc", sb, world[tri])
    nrm = torch.nn.functional.normalize(torch.einsum("ptk,tkc->ptc", sb, normals[tri]), p=2, dim=-1)
    to_l = torch.nn.functional.normalize(light_positions[None, None] - pos.unsqueeze(2), p=2, dim=-1)
    ndl = torch.clamp((to_l * nrm.unsqueeze(2)).sum(-1), 0.0, 1.0)          # [P,T,L]
    color = kd * (ndl * light_intensities[None, None]).sum(-1, keepdim=True)

    sgn = torch.where(inside, one, -one)
    frag = torch.where(keep, torch.special.expit(sgn * sq_dist / sigma_val), zero)
    logit = torch.where(keep, z / gamma_val, zero)
    color = torch.where(keep.unsqueeze(-1), color, torch.zeros_like(color))
    m = torch.maximum(logit.max(-1).values if T else torch.zeros(P), torch.tensor(EPS / gamma_val))
    wts = frag * torch.exp(logit - m.unsqueeze(-1))
    bg = torch.clamp(torch.exp(EPS / gamma_val - m), min=EPS)
    wts = wts / (wts.sum(-1) + bg).unsqueeze(-1)
    rgb = torch.einsum("pt,ptc->pc", wts, color)
    alpha = 1.0 - torch.prod(1.0 - frag, dim=-1)
    return torch.cat([rgb, alpha.unsqueeze(-1)], -1).reshape(out_h, out_w, 4)


def rasterize(world, triangles, normals, diffuse, light_positions, light_intensities,
              camera_matrices, width, height, sigma_val, gamma_val, window=None):
    clip = shading.transform_homogeneous(camera_matrices, world)
    return torch.stack([
        rasterize_batch(clip[b], triangles, world[b], normals[b], diffuse[b], light_positions[b],
                        light_intensities[b], width, height, sigma_val, gamma_val, window=window)
        for b in range(world.shape[0])], 0)


def compute_vertex_normals(vertices, triangles):
    """src/common/meshes.py:3-35."""
    tri = triangles.long()
    out = []
    for b in range(vertices.shape[0]):
        f = vertices[b][tri]                                               # [T,3,3]
        n = torch.zeros_like(vertices[b])
        n = n.index_add(0, tri[:, 0], torch.cross(f[:, 1] - f[:, 0], f[:, 2] - f[:, 0], dim=-1))
        n = n.index_add(0, tri[:, 1], torch.cross(f[:, 2] - f[:, 1], f[:, 0] - f[:, 1], dim=-1))
        n = n.index_add(0, tri[:, 2], torch.cross(f[:, 0] - f[:, 2], f[:, 1] - f[:, 2], dim=-1))
        out.append(n)
    return torch.nn.functional.normalize(torch.stack(out), eps=1e-6, p=2, dim=-1)


def render(vertices, triangles, diffuse, camera_position, camera_lookat, camera_up, light_positions,
           light_intensities, width, height, sigma_val=1e-5, gamma_val=1e-4, fov_y=40.0,
           near_clip=0.01, far_clip=10.0, window=None):
    batch = vertices.shape[0]
    full = lambda v: torch.full((batch,), float(v))
    proj = shading.perspective(width / height, full(fov_y), full(near_clip), full(far_clip))
    transforms = torch.matmul(proj, shading.look_at(camera_position, camera_lookat, camera_up))
    normals = compute_vertex_normals(vertices, triangles)
    return rasterize(vertices, triangles, normals, diffuse, light_positions, light_intensities,
                     transforms, width, height, sigma_val, gamma_val, window=window)
