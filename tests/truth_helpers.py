"""Shared by the CPU and GPU tests: the float64 truth (oracle/truth64.py) of a reference golden scene or of a
random soup, from the G-buffer a forward pass produced."""
import numpy as np
import torch

from oracle import shading, truth64


def golden_transforms(g):
    """The [B,4,4] clip-space transforms render() forms for a golden scene (render.py:183-192), float32 on the host."""
    B, H, W = g["image"].shape[:3]
    fov = float(g["fov_y"]) if "fov_y" in g.files else 40.0
    full = lambda v: torch.full((B,), float(v))
    t = lambda k: torch.tensor(g[k])
    return torch.matmul(shading.perspective(W / H, full(fov), full(0.01), full(10.0)),
                        shading.look_at(t("eye"), t("center"), t("up")))


def golden_upstream(g):
    """d loss / d image of the goldens' loss_weight * mean|image - target| at the reference's own image."""
    lw = float(g["loss_weight"]) if "loss_weight" in g.files else 1.0
    return np.sign(g["image"] - g["target"]).astype(np.float64) * lw / g["image"].size


def scene_truth(ids, bary, clip, transforms, triangles, normals, positions, diffuse, light_positions,
                light_intensities, ambient, drgba, specular=None, shininess=None, camera_position=None):
    """truth64.phong + raster_pullback + whole_vertex_gradient -> the phong dict plus d_clip, noise_clip,
    d_vertices (whole gradient w.r.t. the world-space vertices) and noise_vertices."""
    out = truth64.phong(ids, bary, triangles, normals, positions, diffuse, light_positions, light_intensities,
                        ambient, drgba, specular=specular, shininess=shininess, camera_position=camera_position)
    out["d_clip"], out["noise_clip"] = truth64.raster_pullback(clip, triangles, ids, bary, out["dbary"], out["gabs"])
    out["d_vertices"], out["noise_vertices"] = truth64.whole_vertex_gradient(
        transforms, out["d_positions"], out["d_clip"], out["noise_positions"], out["noise_clip"])
    return out


def golden_scene_truth(g, ids, bary, clip, transforms):
    B = g["image"].shape[0]
    spec = g["specular"] if "specular" in g.files else None
    shin = g["shininess"] if spec is not None else None
    if shin is not None and shin.ndim == 0:
        shin = np.full((B,), float(shin), np.float32)
    return scene_truth(ids, bary, clip, transforms, g["triangles"], g["normals"], g["vertices"], g["diffuse"],
                       g["light_positions"], g["light_intensities"], g["ambient"] if "ambient" in g.files else None,
                       golden_upstream(g), specular=spec, shininess=shin,
                       camera_position=g["eye"] if spec is not None else None)
