"""Generate tests/golden/* by running the REFERENCE itself (this container only).

Imports the reference's Python from /root/reference and its C++ kernel from
oracle/_ref (built by oracle/Makefile from the reference's own source), runs the
hot path on seeded inputs and stores inputs + expected outputs as small fixtures.
Nothing of the reference's source is copied; the fixtures are data.  The GPU box
and the test-suite never need /root/reference -- only these files.

    python tools/make_goldens.py            # rewrites tests/golden/
"""
import hashlib
import json
import os
import shutil
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFERENCE = "/root/reference"
GOLDEN = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, ROOT)

from pytorch_mesh_renderer_amd.common import synthetic  # noqa: E402  (deterministic inputs)


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def load_reference():
    if not os.path.isdir(REFERENCE):
        raise SystemExit("reference tree not present: goldens can only be made where it is")
    sys.path.insert(0, os.path.join(ROOT, "oracle", "_ref"))  # rasterize_triangles_cpp
    sys.path.insert(0, REFERENCE)
    import rasterize_triangles_cpp  # noqa: F401
    import src.mesh_renderer as mr
    sys.modules["src.mesh_renderer.rasterize"].USE_CPP_RASTERIZER = True
    from src.common import camera_utils, shapes
    from src.mesh_renderer import rasterize_triangles_ext as ext
    raster_mod = sys.modules["src.mesh_renderer.rasterize"]
    return mr, raster_mod, ext, camera_utils, shapes, rasterize_triangles_cpp


CUBE_V = torch.tensor([[-1, -1, 1], [-1, -1, -1], [-1, 1, -1], [-1, 1, 1], [1, -1, 1],
                       [1, -1, -1], [1, 1, -1], [1, 1, 1]], dtype=torch.float32)
CUBE_T = torch.tensor([[0, 1, 2], [2, 3, 0], [3, 2, 6], [6, 7, 3], [7, 6, 5], [5, 4, 7],
                       [4, 5, 1], [1, 0, 4], [5, 6, 2], [2, 1, 5], [7, 4, 0], [0, 3, 7]],
                      dtype=torch.int32)


def seeded_dbary(shape, seed=0):
    g = torch.Generator().manual_seed(seed)
    h, w = shape[-3], shape[-2]
    return torch.randn(shape, generator=g) / (h * w)


def kernel_case(cpp, clip, tris, w, h, seed=0):
    ids, bary, z = cpp.forward(clip, tris, w, h)
    dbary = seeded_dbary((h, w, 3), seed)
    (dclip,) = cpp.backward(dbary, clip, tris, ids, bary.detach())
    return ids.numpy(), bary.detach().numpy(), z.numpy(), dbary.numpy(), dclip.numpy()


def main():
    mr, raster_mod, ext, cam, shapes, cpp = load_reference()
    os.makedirs(GOLDEN, exist_ok=True)
    manifest = {}

    # ---- G9: the two benchmark meshes ------------------------------------------------
    mesh_hashes = {}
    for k in (50, 158):
        v, t, n = shapes.sphere(1.0, k)
        mesh_hashes["sphere_%d" % k] = {"V": int(v.shape[0]), "T": int(t.shape[0]),
                                        "vertices": sha(v.numpy()), "triangles": sha(t.numpy()),
                                        "normals": sha(n.numpy())}
    v, t, n = shapes.cube(2.0)
    mesh_hashes["cube_2"] = {"vertices": sha(v.numpy()), "triangles": sha(t.numpy()),
                             "normals": sha(n.numpy())}
    json.dump(mesh_hashes, open(os.path.join(GOLDEN, "shapes_hashes.json"), "w"), indent=1)

    # ---- G1: cube, 64x64 (BASELINE config 1) ---------------------------------------------
    persp = cam.perspective(1.0, torch.tensor([40.0]), torch.tensor([0.01]), torch.tensor([10.0]))
    look = cam.look_at(torch.tensor([[2.0, 3.0, 6.0]]), torch.zeros(1, 3),
                       torch.tensor([[0.0, 1.0, 0.0]]))
    clip = cam.transform_homogeneous(torch.matmul(persp, look), CUBE_V.unsqueeze(0))[0].contiguous()
    ids, bary, z, dbary, dclip = kernel_case(cpp, clip, CUBE_T, 64, 64)
    np.savez_compressed(os.path.join(GOLDEN, "raster_cube64.npz"), clip=clip.numpy(),
                        triangles=CUBE_T.numpy(), ids=ids, bary=bary, z=z, dbary=dbary, dclip=dclip)
    manifest["raster_cube64"] = {"covered": int((bary.sum(-1) > 0.5).sum())}

    # ---- G2: single / paired triangles: w-scaling, behind-eye, degenerate, ties ------------
    base = np.array([[-0.5, -0.5, 0.8, 1.0], [0.0, 0.5, 0.3, 1.0], [0.5, -0.5, 0.3, 1.0]], np.float32)
    one = torch.tensor([[0, 1, 2]], dtype=torch.int32)
    cases = {
        "w_111": (base, one),
        "w_perspective": (base * np.array([[0.2], [0.5], [2.0]], np.float32), one),
        "one_w_negative": (base * np.array([[1.0], [-0.5], [2.0]], np.float32), one),
        "all_w_negative": (base * np.array([[-1.0], [-1.0], [-1.0]], np.float32), one),
        "collinear": (np.array([[-0.5, -0.5, 0.5, 1], [0, 0, 0.5, 1], [0.5, 0.5, 0.5, 1]], np.float32), one),
        "reversed_winding": (base, torch.tensor([[2, 1, 0]], dtype=torch.int32)),
        "coincident_tie": (base, torch.tensor([[0, 1, 2], [0, 1, 2]], dtype=torch.int32)),
        "beyond_far_plane": (base * np.array([[1, 1, 4, 1]], np.float32), one),
        "two_overlapping": (np.concatenate([base, base * np.array([[0.6, 0.6, 0.5, 1]], np.float32)]),
                            torch.tensor([[0, 1, 2], [3, 4, 5]], dtype=torch.int32)),
    }
    tri_out = {}
    for name, (c, t) in cases.items():
        ct = torch.tensor(c)
        ids, bary, z, dbary, dclip = kernel_case(cpp, ct, t, 160, 120, seed=1)
        for k, a in (("clip", c), ("triangles", t.numpy()), ("ids", ids), ("bary", bary), ("z", z),
                     ("dclip", dclip)):
            tri_out["%s.%s" % (name, k)] = a
        manifest["tri_" + name] = {"covered": int((bary.sum(-1) > 0.5).sum())}
    np.savez_compressed(os.path.join(GOLDEN, "raster_triangles_160x120.npz"), **tri_out)
    # the reference's own two test triangles at their native 640x480: hashes only
    big = {}
    for name in ("w_111", "w_perspective"):
        c, t = cases[name]
        ids, bary, z, _, dclip = kernel_case(cpp, torch.tensor(c), t, 640, 480, seed=1)
        big[name] = {"ids": sha(ids), "bary": sha(bary), "z": sha(z), "dclip": dclip.tolist()}
    json.dump(big, open(os.path.join(GOLDEN, "raster_triangles_640x480.json"), "w"), indent=1)

    # ---- G3: the reference test's 28x21 cube, full analytic Jacobian ------------------------
    clip_g3 = torch.tensor(
        [[-0.43889722, -0.53184521, 0.85293502, 1.0], [-0.37635487, 0.22206162, 0.90555805, 1.0],
         [-0.22849123, 0.76811147, 0.80993629, 1.0], [-0.2805393, -0.14092168, 0.71602166, 1.0],
         [0.18631913, -0.62634289, 0.88603103, 1.0], [0.16183566, 0.08129397, 0.93020856, 1.0],
         [0.44147962, 0.53497446, 0.85076219, 1.0], [0.53008741, -0.31276882, 0.77620775, 1.0]],
        dtype=torch.float32)
    ids, bary, z = cpp.forward(clip_g3, CUBE_T, 28, 21)
    jac = np.zeros((32, 21 * 28 * 3), np.float32)
    for i in range(21 * 28 * 3):
        e = torch.zeros(21 * 28 * 3)
        e[i] = 1.0
        (d,) = cpp.backward(e.reshape(21, 28, 3), clip_g3, CUBE_T, ids, bary.detach())
        jac[:, i] = d.reshape(-1).numpy()
    np.savez_compressed(os.path.join(GOLDEN, "raster_jacobian_28x21.npz"), clip=clip_g3.numpy(),
                        triangles=CUBE_T.numpy(), ids=ids.numpy(), bary=bary.detach().numpy(),
                        z=z.numpy(), jacobian=jac)

    # ---- G4 / G5: the benchmark sphere workloads (inputs come from common.synthetic) -----
    sphere = {}
    job = synthetic.sphere_job(8, 256, 256, 50)
    per_cam = []
    for b in range(8):
        ids, bary, z, dbary, dclip = kernel_case(cpp, job["clip"][b].contiguous(), job["triangles"], 256, 256, seed=b)
        per_cam.append({"ids": sha(ids), "bary": sha(bary), "z": sha(z), "dclip": sha(dclip)})
        if b == 0:
            np.savez_compressed(os.path.join(GOLDEN, "raster_sphere256_cam0.npz"), ids=ids, bary=bary,
                                z=z, dclip=dclip)
    sphere["c2_256x256_b8"] = {"clip": sha(job["clip"].numpy()), "cameras": per_cam}
    # clip-space inputs are stored, not regenerated: sin/cos/matmul differ across host CPUs
    np.save(os.path.join(GOLDEN, "sphere_clip_256_b8.npy"), job["clip"].numpy())
    job = synthetic.sphere_job(32, 1024, 1024, 50)
    picked = {}
    dclips = {}
    for b in (0, 7, 16, 29):
        ids, bary, z, dbary, dclip = kernel_case(cpp, job["clip"][b].contiguous(), job["triangles"], 1024, 1024, seed=b)
        picked[str(b)] = {"ids": sha(ids), "bary": sha(bary), "z": sha(z),
                          "covered": int((bary.sum(-1) > 0.5).sum()),
                          "ids_sample": ids[::97, ::89].tolist()}
        dclips["dclip_%d" % b] = dclip
    sphere["c3_1024x1024_b32"] = {"clip": sha(job["clip"].numpy()), "cameras": picked}
    np.save(os.path.join(GOLDEN, "sphere_clip_1024_b32.npy"), job["clip"].numpy())
    np.savez_compressed(os.path.join(GOLDEN, "raster_sphere1024_dclip.npz"), **dclips)
    json.dump(sphere, open(os.path.join(GOLDEN, "raster_sphere_hashes.json"), "w"), indent=1)

    # ---- G6: rasterize() and render() on the cube, 64x48, with gradients -------------------
    W, H = 64, 48
    persp = cam.perspective(W / H, torch.tensor([40.0]), torch.tensor([0.01]), torch.tensor([10.0]))
    center, up = torch.zeros(1, 3), torch.tensor([[0.0, 1.0, 0.0]])
    proj = torch.cat([torch.matmul(persp, cam.look_at(torch.tensor([[2.0, 3.0, 6.0]]), center, up)),
                      torch.matmul(persp, cam.look_at(torch.tensor([[-3.0, 1.0, 6.0]]), center, up))], 0)
    verts = torch.stack([CUBE_V, CUBE_V]).clone().requires_grad_(True)
    rgba = torch.cat([CUBE_V * 0.5 + 0.5, torch.ones(8, 1)], 1)
    attrs = torch.stack([rgba, rgba]).clone().requires_grad_(True)
    bg = torch.tensor([0.1, 0.2, 0.3, 0.0])
    out = mr.rasterize(verts, attrs, CUBE_T, proj, W, H, bg)
    target = torch.rand(out.shape, generator=torch.Generator().manual_seed(3))
    loss = torch.mean(torch.abs(out - target))
    loss.backward()
    np.savez_compressed(os.path.join(GOLDEN, "rasterize_unlit_cube_64x48.npz"),
                        vertices=verts.detach().numpy(), attributes=attrs.detach().numpy(),
                        triangles=CUBE_T.numpy(), projection=proj.numpy(), background=bg.numpy(),
                        out=out.detach().numpy(), target=target.numpy(),
                        dvertices=verts.grad.numpy(), dattributes=attrs.grad.numpy())

    def render_case(path, vertices, normals, diffuse, triangles, eye, center, up, lpos, lint, w, h,
                    specular=None, shininess=None, ambient=None, target_seed=5, **kw):
        leaves = {"vertices": vertices, "normals": normals, "diffuse": diffuse, "light_positions": lpos,
                  "light_intensities": lint}
        if specular is not None:
            leaves["specular"] = specular
        if ambient is not None:
            leaves["ambient"] = ambient
        # (the reference's look_at calls numpy on the eye, so the eye cannot require grad there)
        leaves = {k: v.clone().requires_grad_(True) for k, v in leaves.items()}
        img = mr.render(leaves["vertices"], triangles, leaves["normals"], leaves["diffuse"],
                        eye, center, up, leaves["light_positions"],
                        leaves["light_intensities"], w, h,
                        specular_colors=leaves.get("specular"), shininess_coefficients=shininess,
                        ambient_color=leaves.get("ambient"), **kw)
        target = torch.rand(img.shape, generator=torch.Generator().manual_seed(target_seed))
        loss = torch.mean(torch.abs(img - target))
        loss.backward()
        data = {"triangles": triangles.numpy(), "eye": eye.numpy(), "center": center.numpy(), "up": up.numpy(),
                "image": img.detach().numpy(), "target": target.numpy(), "loss": np.float32(loss.item())}
        if shininess is not None:
            data["shininess"] = shininess.numpy() if torch.is_tensor(shininess) else np.float32(shininess)
        for k, v in leaves.items():
            data[k] = v.detach().numpy()
            data["d_" + k] = v.grad.numpy() if v.grad is not None else np.zeros_like(v.detach().numpy())
        np.savez_compressed(path, **data)

    # Gray cube of mesh_renderer_test.testRendersSimpleCube, shrunk to 64x48
    rot = cam.euler_matrices(torch.tensor([[-20.0, 0.0, 60.0], [45.0, 60.0, 0.0]]))[:, :3, :3]
    cube_n = torch.nn.functional.normalize(CUBE_V, dim=1, p=2)
    vw = torch.matmul(torch.stack([CUBE_V, CUBE_V]), rot.transpose(1, 2)).contiguous()
    nw = torch.matmul(torch.stack([cube_n, cube_n]), rot.transpose(1, 2)).contiguous()
    eye = torch.tensor(2 * [[0.0, 0.0, 6.0]])
    center2 = torch.zeros(2, 3)
    up2 = torch.tensor(2 * [[0.0, 1.0, 0.0]])
    lpos = torch.tensor([[[0.0, 0.0, 6.0]], [[0.0, 0.0, 6.0]]])
    lint = torch.ones(2, 1, 3)
    render_case(os.path.join(GOLDEN, "render_gray_cube_64x48.npz"), vw, nw, torch.ones_like(vw),
                CUBE_T, eye, center2, up2, lpos, lint, 64, 48)
    # two lights + ambient + random diffuse (exercises every diffuse-path gradient)
    g = torch.Generator().manual_seed(11)
    diffuse = torch.rand(2, 8, 3, generator=g)
    lpos2 = torch.tensor([[[0.0, 0.0, 6.0], [1.0, 2.0, 6.0]], [[0.0, -2.0, 4.0], [1.0, 3.0, 4.0]]])
    lint2 = torch.tensor([[[1.0, 1.0, 1.0], [0.5, 0.7, 0.9]], [[2.0, 0.0, 1.0], [0.0, 2.0, 1.0]]])
    ambient = torch.tensor([[0.0, 0.0, 0.0], [0.1, 0.1, 0.2]])
    render_case(os.path.join(GOLDEN, "render_lit_cube_64x48.npz"), vw, nw, diffuse, CUBE_T, eye,
                center2, up2, lpos2, lint2, 64, 48, ambient=ambient)
    # specular (row F1): per-vertex shininess and scalar shininess
    specular = torch.rand(2, 8, 3, generator=g)
    render_case(os.path.join(GOLDEN, "render_specular_cube_64x48.npz"), vw, nw, diffuse, CUBE_T, eye,
                center2, up2, lpos2, lint2, 64, 48, specular=specular,
                shininess=6.0 * torch.ones(2, 8), ambient=ambient)
    render_case(os.path.join(GOLDEN, "render_specular_scalar_cube_64x48.npz"), vw, nw, diffuse, CUBE_T,
                eye, center2, up2, lpos2, lint2, 64, 48, specular=specular,
                shininess=torch.tensor(4.0), ambient=ambient)

    # ---- G7: render() on the 5k sphere, 128x128, B=2 ------------------------------------------
    job = synthetic.sphere_job(2, 128, 128, 50)
    render_case(os.path.join(GOLDEN, "render_sphere5k_128.npz"), job["vertices"], job["normals"],
                job["diffuse"], job["triangles"], job["eyes"], torch.zeros(2, 3),
                torch.tensor(2 * [[0.0, 1.0, 0.0]]), job["light_positions"], job["light_intensities"],
                128, 128)

    # ---- G8: SoftRas renderer -------------------------------------------------
    from src import soft_mesh_renderer as soft
    soft_rb = sys.modules["src.soft_mesh_renderer.rasterize"].rasterize_batch
    # the single-triangle scene of the reference's test_single_triangle_forward, both blur settings
    st = dict(
        clip=torch.tensor([[1.0, -1.0, 0.25, 1.0], [1.0, 1.0, 0.25, 1.0], [-1.0, -1.0, 0.25, 1.0]]),
        triangles=torch.tensor([[0, 1, 2]], dtype=torch.int32),
        world=torch.tensor([[1.0, -1.0, 0.0], [1.0, 1.0, 0.0], [-1.0, -1.0, 0.0]]),
        normals=torch.tensor([[0.0, 0.0, 1.0]] * 3), diffuse=torch.tensor([[1.0, 0.0, 0.0]] * 3),
        light_positions=torch.tensor([[0.0, 0.0, 100000.0]]), light_intensities=torch.tensor([1.0]))
    blur2 = 0.1 * np.sqrt(2.0) + 1e-6
    sigma2 = float(-blur2 ** 2 / torch.special.logit(torch.tensor(1e-3)))
    soft_out = {k: v.numpy() for k, v in st.items()}
    for tag, (sig, gam, blur) in {"a": (1e-5, 1e-4, 0.01), "b": (sigma2, 1e-4, blur2)}.items():
        img = soft_rb(st["clip"], st["triangles"], st["world"], st["normals"], st["diffuse"],
                      st["light_positions"], st["light_intensities"], 10, 10, sig, gam, blur)
        soft_out["image_" + tag] = img.numpy()
        soft_out["params_" + tag] = np.array([sig, gam, blur], np.float64)
    np.savez_compressed(os.path.join(GOLDEN, "soft_single_triangle_10x10.npz"), **soft_out)

    def soft_case(path, k, size, sigma, gamma, batch=1):
        job = synthetic.sphere_job(batch, size, size, k)
        tris_ccw = job["triangles"]                       # shapes.sphere winds CCW seen from outside
        leaves = {"vertices": job["vertices"], "diffuse": torch.rand(batch, job["vertices"].shape[1], 3,
                                                                      generator=torch.Generator().manual_seed(k)),
                  "light_positions": job["light_positions"]}
        leaves = {n: v.clone().requires_grad_(True) for n, v in leaves.items()}
        lint = torch.full((batch, 1), 1.3)
        img = soft.render(leaves["vertices"], tris_ccw, leaves["diffuse"], job["eyes"], torch.zeros(batch, 3),
                          torch.tensor(batch * [[0.0, 1.0, 0.0]]), leaves["light_positions"], lint, size, size,
                          sigma_val=sigma, gamma_val=gamma)
        target = torch.rand(img.shape, generator=torch.Generator().manual_seed(9))
        torch.mean(torch.abs(img - target)).backward()
        data = {"triangles": tris_ccw.numpy(), "eye": job["eyes"].numpy(), "light_intensities": lint.numpy(),
                "image": img.detach().numpy(), "target": target.numpy(),
                "params": np.array([sigma, gamma], np.float64)}
        for n, v in leaves.items():
            data[n] = v.detach().numpy()
            data["d_" + n] = v.grad.numpy()
        np.savez_compressed(path, **data)

    soft_case(os.path.join(GOLDEN, "soft_sphere_k6_32.npz"), 6, 32, 1e-5, 1e-4)       # reference defaults
    soft_case(os.path.join(GOLDEN, "soft_sphere_k10_32.npz"), 10, 32, 1e-4, 1e-2)     # softer blend

    # ---- camera utilities ---------------------------------------------------------------------
    eyes = synthetic.orbit_eyes(5)
    np.savez_compressed(
        os.path.join(GOLDEN, "camera_utils.npz"), eyes=eyes.numpy(),
        look_at=cam.look_at(eyes, torch.zeros(5, 3), torch.tensor(5 * [[0.0, 1.0, 0.0]])).numpy(),
        perspective=cam.perspective(1.25, torch.tensor([40.0, 13.3]), torch.tensor([0.01, 0.1]),
                                    torch.tensor([10.0, 25.0])).numpy(),
        euler_in=np.array([[-20.0, 0.0, 60.0], [45.0, 60.0, 0.0], [0.1, 0.2, 0.3]], np.float32),
        euler=cam.euler_matrices(torch.tensor([[-20.0, 0.0, 60.0], [45.0, 60.0, 0.0], [0.1, 0.2, 0.3]])).numpy(),
        tone_in=np.linspace(0, 3, 2 * 3 * 4 * 3, dtype=np.float32).reshape(2, 3, 4, 3),
        tone_out=mr.tone_mapper(torch.linspace(0, 3, 72).reshape(2, 3, 4, 3), 0.7).numpy())

    # ---- the reference tests' own data files (8-bit PNGs used by its test-suite) ---------------
    png_dir = os.path.join(GOLDEN, "ref_png")
    os.makedirs(png_dir, exist_ok=True)
    for name in ("Simple_Triangle.png", "Perspective_Corrected_Triangle.png", "Unlit_Cube_0.png",
                 "Unlit_Cube_1.png", "Gray_Cube_0.png", "Gray_Cube_1.png"):
        shutil.copyfile(os.path.join(REFERENCE, "src/mesh_renderer/test_data", name),
                        os.path.join(png_dir, name))

    json.dump(manifest, open(os.path.join(GOLDEN, "manifest.json"), "w"), indent=1)
    total = sum(os.path.getsize(os.path.join(dp, f)) for dp, _, fs in os.walk(GOLDEN) for f in fs)
    print("goldens written to %s (%.1f MB)" % (GOLDEN, total / 1e6))


if __name__ == "__main__":
    main()
