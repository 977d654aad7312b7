"""GPU parity, API level: rasterize() / render() vs the reference's own outputs.

Counterparts of the reference's rasterize_triangles_test.py and mesh_renderer_test.py
(golden-image, Jacobian and optimisation tests) plus float goldens captured from the
reference (tools/make_goldens.py).  Bar: RGBA and gradients within 1e-4 abs.
"""
import math
import os
import sys

import numpy as np
import pytest
import torch
from PIL import Image

from conftest import GOLDEN, golden_npz
from oracle import shading
from pytorch_mesh_renderer_amd import mesh_renderer
from pytorch_mesh_renderer_amd.common import camera_utils, synthetic
from pytorch_mesh_renderer_amd.mesh_renderer.rasterize import rasterize_barycentric

pytestmark = pytest.mark.gpu
ATOL = 1e-4

CUBE_V = torch.tensor([[-1, -1, 1], [-1, -1, -1], [-1, 1, -1], [-1, 1, 1], [1, -1, 1],
                       [1, -1, -1], [1, 1, -1], [1, 1, 1]], dtype=torch.float32)
CUBE_T = torch.tensor([[0, 1, 2], [2, 3, 0], [3, 2, 6], [6, 7, 3], [7, 6, 5], [5, 4, 7],
                       [4, 5, 1], [1, 0, 4], [5, 6, 2], [2, 1, 5], [7, 4, 0], [0, 3, 7]],
                      dtype=torch.int32)


def expect_image_file_and_render_are_near(name, image, max_outlier_fraction=0.001,
                                          pixel_error_threshold=0.01):
    """Same soft comparison as the reference's test_utils.py:105-160."""
    baseline = np.asarray(Image.open(os.path.join(GOLDEN, "ref_png", name))).astype(float) / 255.0
    result = np.clip(image.detach().cpu().numpy(), 0.0, 1.0)
    assert baseline.shape == result.shape
    outliers = np.any(np.abs(baseline - result) > pixel_error_threshold, axis=2)
    fraction = np.count_nonzero(outliers) / np.prod(baseline.shape[:2])
    assert fraction <= max_outlier_fraction, "%s: %.5f of pixels are outliers" % (name, fraction)


def _leaf(g, key, device):
    return torch.tensor(g[key], device=device, requires_grad=True)


def test_rasterize_unlit_cube_golden(device):
    g = golden_npz("rasterize_unlit_cube_64x48.npz")
    v, a = _leaf(g, "vertices", device), _leaf(g, "attributes", device)
    out = mesh_renderer.rasterize(v, a, torch.tensor(g["triangles"], device=device),
                                  torch.tensor(g["projection"], device=device), 64, 48,
                                  torch.tensor(g["background"], device=device))
    np.testing.assert_allclose(out.detach().cpu().numpy(), g["out"], atol=ATOL, rtol=0)
    torch.mean(torch.abs(out - torch.tensor(g["target"], device=device))).backward()
    np.testing.assert_allclose(v.grad.cpu().numpy(), g["dvertices"], atol=ATOL, rtol=0)
    np.testing.assert_allclose(a.grad.cpu().numpy(), g["dattributes"], atol=ATOL, rtol=0)


def _assert_close_where_reference_is_finite(got, want, what):
    """With per-vertex shininess and a background pixel some of the reference's own gradients are
    NaN (0 * pow(0, -1) in torch autograd, see shade_spec.hip); those entries pin nothing beyond
    ours being finite."""
    assert np.isfinite(got).all(), what
    ok = np.isfinite(want)
    np.testing.assert_allclose(got[ok], want[ok], atol=ATOL, rtol=0, err_msg=what)


def _render_golden(name, device):
    g = golden_npz(name)
    h, w = g["image"].shape[1:3]
    leaves = {k: _leaf(g, k, device) for k in ("vertices", "normals", "diffuse", "light_positions",
                                               "light_intensities")}
    spec = _leaf(g, "specular", device) if "specular" in g.files else None
    amb = _leaf(g, "ambient", device) if "ambient" in g.files else None
    shine = torch.tensor(g["shininess"], device=device) if "shininess" in g.files else None
    dev = lambda k: torch.tensor(g[k], device=device)
    img = mesh_renderer.render(leaves["vertices"], dev("triangles"), leaves["normals"],
                               leaves["diffuse"], dev("eye"), dev("center"), dev("up"),
                               leaves["light_positions"], leaves["light_intensities"], w, h,
                               specular_colors=spec, shininess_coefficients=shine, ambient_color=amb)
    assert img.shape == g["image"].shape and img.dtype == torch.float32
    np.testing.assert_allclose(img.detach().cpu().numpy(), g["image"], atol=ATOL, rtol=0)
    alpha = img[..., 3].detach().cpu().numpy()
    assert set(np.unique(alpha)) <= {0.0, 1.0} and np.array_equal(alpha, g["image"][..., 3])
    torch.mean(torch.abs(img - dev("target"))).backward()
    for k, t in leaves.items():
        _assert_close_where_reference_is_finite(t.grad.cpu().numpy(), g["d_" + k], k)
    if spec is not None:
        _assert_close_where_reference_is_finite(spec.grad.cpu().numpy(), g["d_specular"], "specular")
    if amb is not None:
        _assert_close_where_reference_is_finite(amb.grad.cpu().numpy(), g["d_ambient"], "ambient")


@pytest.mark.parametrize("name", ["render_gray_cube_64x48.npz", "render_lit_cube_64x48.npz",
                                  "render_sphere5k_128.npz"])
def test_render_diffuse_goldens(device, name):
    _render_golden(name, device)


def test_render_six_lights_golden_takes_the_fused_path(device):
    """Round 3: more than four lights no longer fall back to HIP interpolation + ~40 eager torch ops (a
    10x cliff): the forward loops over them from LDS, the backward over a run-time count, the light
    gradients come four lights at a time.  The reference's capture with SIX lights + ambient, image and
    every gradient, and the fused entry points must have been the ones that ran."""
    from pytorch_mesh_renderer_amd import _native
    assert _native.shade_max_lights() >= 6 and _native.shade_fast_lights() == 4
    with _CountCalls("render_forward") as fwd, _CountCalls("_shade_backward_call") as bwd, \
            _CountCalls("interpolate_forward") as composed:
        _render_golden("render_six_lights_64x48.npz", device)
    assert fwd.calls == 1 and composed.calls == 0
    assert bwd.calls == 3, "one pass for the vertex-side gradients, two for the light gradients (4 + 2 lights)"


@pytest.mark.parametrize("n_lights,light_grads", [(5, True), (9, True), (7, False), (32, False)])
def test_fused_render_with_many_lights_matches_composed_path(device, n_lights, light_grads):
    """5 / 7 / 9 / 32 lights through the fused kernels vs the composed path (HIP interpolation + torch
    Phong + autograd), image and gradients -- with and without the lights requiring grad, sign-coded
    loss route included (losses.l1_loss on render()'s own output)."""
    import importlib
    render_mod = importlib.import_module("pytorch_mesh_renderer_amd.mesh_renderer.render")
    job = synthetic.sphere_job(2, 96, 72, 10)
    gen = torch.Generator().manual_seed(n_lights)
    target = torch.rand(2, 72, 96, 4, generator=gen).to(device)
    base = {"vertices": job["vertices"], "normals": job["normals"],
            "diffuse": torch.rand(job["vertices"].shape, generator=gen),
            "light_positions": torch.rand(2, n_lights, 3, generator=gen) * 8.0 - 4.0,
            "light_intensities": torch.rand(2, n_lights, 3, generator=gen) * (1.5 / n_lights),
            "ambient": torch.rand(2, 3, generator=gen) * 0.2}
    results = {}
    for fused in (True, False):
        scene = {k: v.clone().to(device) for k, v in base.items()}
        for k in scene:
            if light_grads or k in ("vertices", "normals", "diffuse"):
                scene[k].requires_grad_(True)
        render_mod.USE_FUSED_SHADING = fused
        try:
            with _CountCalls("render_forward") as counter:
                img = mesh_renderer.render(scene["vertices"], job["triangles"].to(device), scene["normals"],
                                           scene["diffuse"], job["eyes"], torch.zeros(2, 3), torch.tensor([0.0, 1.0, 0.0]),
                                           scene["light_positions"], scene["light_intensities"], 96, 72,
                                           ambient_color=scene["ambient"])
            (mesh_renderer.losses.l1_loss(img, target) * 10.0).backward()
        finally:
            render_mod.USE_FUSED_SHADING = True
        assert counter.calls == (1 if fused else 0)
        results[fused] = (img.detach().cpu().numpy(),
                          {k: v.grad.cpu().numpy() for k, v in scene.items() if v.requires_grad})
    np.testing.assert_allclose(results[True][0], results[False][0], atol=ATOL, rtol=0)
    assert set(results[True][1]) == set(results[False][1])
    for k, want in results[False][1].items():
        assert np.abs(want).max() > 1e-5, k
        np.testing.assert_allclose(results[True][1][k], want, atol=ATOL, rtol=2e-3, err_msg=k)


def test_render_nine_lights_specular_golden(device):
    """Round 4: the reference's capture with NINE lights, diffuse + specular (scalar shininess per image),
    ambient, every input differentiated -- shininess included -- through the fused specular kernels (groups
    of four lights: 4 + 4 + 1)."""
    g = golden_npz("render_nine_lights_64x48.npz")
    keys = ("vertices", "normals", "diffuse", "specular", "shininess", "light_positions", "light_intensities",
            "ambient")
    leaves = {k: _leaf(g, k, device) for k in keys}
    dev = lambda k: torch.tensor(g[k], device=device)
    with _CountCalls("shade_specular_forward") as fwd, _CountCalls("interpolate_forward") as composed:
        img = mesh_renderer.render(leaves["vertices"], dev("triangles"), leaves["normals"], leaves["diffuse"],
                                   dev("eye"), dev("center"), dev("up"), leaves["light_positions"],
                                   leaves["light_intensities"], 64, 48, specular_colors=leaves["specular"],
                                   shininess_coefficients=leaves["shininess"], ambient_color=leaves["ambient"])
    assert fwd.calls == 3 and composed.calls == 0
    np.testing.assert_allclose(img.detach().cpu().numpy(), g["image"], atol=ATOL, rtol=0)
    (float(g["loss_weight"]) * torch.mean(torch.abs(img - dev("target")))).backward()
    for k, t in leaves.items():
        assert np.abs(g["d_" + k]).max() > 1e-5, k
        np.testing.assert_allclose(t.grad.cpu().numpy(), g["d_" + k], atol=ATOL, rtol=0, err_msg=k)


@pytest.mark.parametrize("n_lights,specular", [(9, False), (32, False), (9, True), (32, True), (7, "vertex")])
def test_wide_light_paths_match_the_oracle(device, n_lights, specular):
    """the many-light paths (run-time light loop of the diffuse kernels, groups of four in
    the specular kernels, light gradients 1 + ceil(L / 4) passes) against the CPU restatement of the
    reference (oracle/shading.render; render.py:304-372) instead of against the composed HIP path: image
    and every gradient -- lights, ambient, specular colours, shininess (per image, or per vertex for
    `specular == "vertex"`) -- within 1e-4 on a 64 x 48 sphere."""
    job = synthetic.sphere_job(2, 64, 48, 8)
    gen = torch.Generator().manual_seed(100 + n_lights)
    V = job["vertices"].shape[1]
    base = {"vertices": job["vertices"], "normals": job["normals"],
            "diffuse": torch.rand(2, V, 3, generator=gen),
            "light_positions": torch.rand(2, n_lights, 3, generator=gen) * 8.0 - 4.0,
            "light_intensities": torch.rand(2, n_lights, 3, generator=gen) * (2.0 / n_lights),
            "ambient": torch.rand(2, 3, generator=gen) * 0.2}
    if specular:
        base["specular"] = torch.rand(2, V, 3, generator=gen)
        base["shininess"] = (0.3 + torch.rand(2, V, generator=gen)) if specular == "vertex" else torch.tensor([0.5, 0.9])
    target = torch.rand(2, 48, 64, 4, generator=gen)
    center, up = torch.zeros(2, 3), torch.tensor([[0.0, 1.0, 0.0]]).repeat(2, 1)
    results = {}
    for where in ("hip", "oracle"):
        dev_ = device if where == "hip" else torch.device("cpu")
        scene = {k: v.clone().to(dev_).requires_grad_(True) for k, v in base.items()}
        kwargs = {"ambient_color": scene["ambient"]}
        if specular:
            kwargs.update(specular_colors=scene["specular"], shininess_coefficients=scene["shininess"])
        if where == "hip":
            img = mesh_renderer.render(scene["vertices"], job["triangles"].to(dev_), scene["normals"], scene["diffuse"],
                                       job["eyes"], center, up, scene["light_positions"], scene["light_intensities"],
                                       64, 48, **kwargs)
        else:
            # (the power inside the mask only: per-vertex shininess with background pixels, see shade_spec.hip)
            img = shading.render(scene["vertices"], job["triangles"], scene["normals"], scene["diffuse"], job["eyes"],
                                 center, up, scene["light_positions"], scene["light_intensities"], 64, 48,
                                 power_inside_mask_only=specular == "vertex", **kwargs)
        (20.0 * torch.mean(torch.abs(img - target.to(dev_)))).backward()
        results[where] = (img.detach().cpu().numpy(), {k: v.grad.cpu().numpy() for k, v in scene.items()})
    np.testing.assert_allclose(results["hip"][0], results["oracle"][0], atol=ATOL, rtol=0)
    assert 0.2 < (results["oracle"][0][..., 3] > 0.5).mean() < 0.95
    for k, want in results["oracle"][1].items():
        assert np.isfinite(want).all() and np.abs(want).max() > 1e-6, k
        np.testing.assert_allclose(results["hip"][1][k], want, atol=ATOL, rtol=0, err_msg=k)


def _random_soup(seed, n_attrs, device=None):
    gen = torch.Generator().manual_seed(seed)
    V, T = 80, 60
    soup = {"vertices": torch.rand(2, V, 3, generator=gen) * 2.0 - 1.0,
            "attributes": torch.rand(2, V, n_attrs, generator=gen),
            "background": torch.linspace(-1.0, 0.5, n_attrs)}
    tris = torch.randint(0, V, (T, 3), generator=gen, dtype=torch.int32)
    eye = torch.tensor([[0.0, 0.0, 4.0], [1.0, 1.0, 4.0]])
    proj = torch.matmul(camera_utils.perspective(57 / 41, torch.tensor([40.0, 40.0]), torch.tensor([0.01, 0.01]),
                                                 torch.tensor([10.0, 10.0])),
                        camera_utils.look_at(eye, torch.zeros(2, 3), torch.tensor(2 * [[0.0, 1.0, 0.0]])))
    return soup, tris, proj


@pytest.mark.parametrize("n_attrs", [1, 4, 9, 16, 17])
def test_rasterize_matches_oracle_on_random_soups(device, n_attrs):
    """rasterize() (rasterize.py:27-152) against oracle/shading.rasterize on random
    triangle soups -- values, dL/dattributes, dL/dvertices and dL/dbackground within 1e-4 -- for attribute
    counts on both sides of the fused path's limit of 16 and both of its backward kernels (<= 8: lanes)."""
    soup, tris, proj = _random_soup(300 + n_attrs, n_attrs)
    target = torch.rand(2, 41, 57, n_attrs, generator=torch.Generator().manual_seed(n_attrs))
    results = {}
    for where in ("hip", "oracle"):
        dev_ = device if where == "hip" else torch.device("cpu")
        leaves = {k: v.clone().to(dev_).requires_grad_(True) for k, v in soup.items()}
        if where == "hip":
            out = mesh_renderer.rasterize(leaves["vertices"], leaves["attributes"], tris.to(dev_), proj.to(dev_), 57, 41,
                                          leaves["background"])
        else:
            # a soup's triangles intersect each other: along those lines the winner of the depth test hangs on
            # the last bit of the clip coordinates, so the oracle rasterizes the device's clip bits (its own
            # matmul still carries the gradient; see oracle/shading.rasterize)
            # (rasterize() forms them with transform_homogeneous -- a batched GEMM on the device)
            clip_bits = camera_utils.transform_homogeneous(proj.to(device), soup["vertices"].to(device)).cpu()
            out = shading.rasterize(leaves["vertices"], leaves["attributes"], tris, proj, 57, 41, leaves["background"],
                                    clip_bits=clip_bits)
        torch.mean(torch.abs(out - target.to(dev_))).backward()
        results[where] = (out.detach().cpu().numpy(), {k: v.grad.cpu().numpy() for k, v in leaves.items()})
    np.testing.assert_allclose(results["hip"][0], results["oracle"][0], atol=ATOL, rtol=0)
    for k, want in results["oracle"][1].items():
        assert np.abs(want).max() > 1e-6, k
        np.testing.assert_allclose(results["hip"][1][k], want, atol=ATOL, rtol=0, err_msg=k)


@pytest.mark.parametrize("n_attrs", [1, 3, 4, 7, 9, 12, 13, 16])
@pytest.mark.parametrize("scene", ["sphere_ragged", "crowded", "soup"])
def test_rasterize_interpolation_epilogue_matches_two_passes(device, scene, n_attrs):
    """Round 4: rasterize()'s forward as ONE pass -- the interpolation is the epilogue of k_raster's tile walk, the
    winners' attribute records come from LDS -- against k_raster followed by k_interp_forward_rec over the
    G-buffer: ids and barycentrics bit-identical, the interpolated image to 1e-6.  "crowded": a 28k-triangle
    sphere at 64 x 64, whose regions need several bin rounds (the records are then read per lane from memory);
    "sphere_ragged": a size that is no multiple of the region / tile edges; "soup": random triangles with an
    infinite attribute on triangle 0 -- uncovered pixels multiply it by zero, as the reference does."""
    from pytorch_mesh_renderer_amd import _native
    gen = torch.Generator().manual_seed(n_attrs)
    if scene == "soup":
        soup, tris, proj = _random_soup(40 + n_attrs, n_attrs)
        clip = camera_utils.transform_homogeneous(proj, soup["vertices"]).contiguous()
        attrs, w, h = soup["attributes"].clone(), 57, 41
        attrs[0, tris[0, 1].long(), 0] = float("inf")
    else:
        w, h, res = (201, 123, 10) if scene == "sphere_ragged" else (64, 64, 120)
        job = synthetic.sphere_job(2, w, h, res)
        clip, tris = job["clip"], job["triangles"]
        attrs = torch.rand(2, clip.shape[1], n_attrs, generator=gen)
    bg = torch.linspace(-1.0, 0.5, n_attrs)
    clip, tris, attrs, bg = clip.to(device), tris.to(device), attrs.to(device), bg.to(device)
    ids, bary, out, records = _native.rasterize_interpolate_forward(clip, attrs, tris, bg, w, h)
    ids2, bary2, _ = _native.rasterize_forward(clip, tris, w, h)
    out2, records2 = _native.interpolate_forward_records(ids2, bary2, attrs, tris, bg)
    assert torch.equal(ids, ids2) and torch.equal(bary.view(torch.int32), bary2.view(torch.int32))
    used = clip.shape[0] * tris.shape[0] * 3 * (4 * ((n_attrs + 3) // 4)) * 4     # bytes: [B*T][3][A padded to 4 / 8 / 12 / 16]
    assert torch.equal(records.view(torch.uint8)[:used], records2.view(torch.uint8)[:used])
    got, want = out.cpu().numpy(), out2.cpu().numpy()
    assert np.array_equal(np.isnan(got), np.isnan(want))
    ok = np.isfinite(want)
    np.testing.assert_allclose(got[ok], want[ok], atol=1e-6, rtol=1e-6)
    assert 0.05 < float((bary.sum(-1) > 0.5).float().mean()) < 0.98


def test_wide_store_hazard_soup(device):
    """Round 4: the soup on which the fuzzer caught a hardware hazard (k_raster's 12- / 16-byte buffer stores with a
    register soffset, followed at once by a vector write of their data registers: attribute 0 of lanes 12-15 of every
    16 came out as the pixel's triangle id, on a few tiles, differently from run to run; LLVM inserts the wait
    state only for stores without a register soffset).  The one-pass forward must equal the two-pass one, every time."""
    from pytorch_mesh_renderer_amd import _native
    d = np.load(os.path.join(GOLDEN, "soup_a3_wide_store_hazard.npz"))
    clip, tris, attrs, bg = [torch.from_numpy(d[k]).to(device) for k in ("clip", "tris", "attrs", "bg")]
    w, h = int(d["W"]), int(d["H"])
    ids, bary, _ = _native.rasterize_forward(clip, tris, w, h)
    want, _ = _native.interpolate_forward_records(ids, bary, attrs, tris, bg)
    for _ in range(6):
        ids2, bary2, out, _ = _native.rasterize_interpolate_forward(clip, attrs, tris, bg, w, h)
        assert torch.equal(ids2, ids) and torch.equal(bary2.view(torch.int32), bary.view(torch.int32))
        assert bool(torch.isclose(out, want, atol=1e-5, rtol=1e-5, equal_nan=True).all())


def test_rasterize_seventeen_attributes_golden(device):
    """The reference's own rasterize() on a random soup with 17 attributes (tools/make_goldens_r4.py attrs)."""
    g = golden_npz("rasterize_soup_a17_40x30.npz")
    leaves = {k: _leaf(g, k, device) for k in ("vertices", "attributes", "background")}
    out = mesh_renderer.rasterize(leaves["vertices"], leaves["attributes"], torch.tensor(g["triangles"], device=device),
                                  torch.tensor(g["transforms"], device=device), 40, 30, leaves["background"])
    np.testing.assert_allclose(out.detach().cpu().numpy(), g["image"], atol=ATOL, rtol=0)
    torch.mean(torch.abs(out - torch.tensor(g["target"], device=device))).backward()
    for k, t in leaves.items():
        np.testing.assert_allclose(t.grad.cpu().numpy(), g["d_" + k], atol=ATOL, rtol=0, err_msg=k)


class _CountCalls:
    """Counts the calls of a _native entry point (which path did render() take?)."""

    def __init__(self, name):
        from pytorch_mesh_renderer_amd import _native
        self.mod, self.name, self.calls = _native, name, 0

    def __enter__(self):
        self.orig = getattr(self.mod, self.name)

        def wrapper(*a, **k):
            self.calls += 1
            return self.orig(*a, **k)
        setattr(self.mod, self.name, wrapper)
        return self

    def __exit__(self, *exc):
        setattr(self.mod, self.name, self.orig)
        return False


@pytest.mark.parametrize("name", ["render_specular_cube_64x48.npz", "render_specular_scalar_cube_64x48.npz"])
def test_render_specular_goldens(device, name):
    """Reference-captured image and gradients, per-vertex and scalar shininess: both take the fused
    specular kernels.  (Where the reference's gradient is NaN -- per-vertex shininess with a
    background pixel -- only finiteness is asked of ours.)"""
    with _CountCalls("shade_specular_forward") as counter:
        _render_golden(name, device)
    assert counter.calls == 1


@pytest.mark.parametrize("name", ["render_shininess_vertex_filled_64x48.npz", "render_shininess_vertex_64x48.npz",
                                  "render_shininess_image_64x48.npz"])
def test_render_shininess_gradient_goldens(device, name):
    """Row F1: shininess_coefficients that require grad, per vertex ([B,V]; mesh filling the frame, and
    with background where the reference is partly NaN) and per image ([B]), against the reference's
    image and gradients -- d shininess included."""
    g = golden_npz(name)
    keys = ("vertices", "normals", "diffuse", "specular", "light_positions", "light_intensities", "ambient",
            "shininess")
    leaves = {k: _leaf(g, k, device) for k in keys}
    dev = lambda k: torch.tensor(g[k], device=device)
    with _CountCalls("shade_specular_backward") as counter:
        img = mesh_renderer.render(leaves["vertices"], dev("triangles"), leaves["normals"], leaves["diffuse"],
                                   dev("eye"), dev("center"), dev("up"), leaves["light_positions"],
                                   leaves["light_intensities"], 64, 48, specular_colors=leaves["specular"],
                                   shininess_coefficients=leaves["shininess"], ambient_color=leaves["ambient"],
                                   fov_y=float(g["fov_y"]))
        np.testing.assert_allclose(img.detach().cpu().numpy(), g["image"], atol=ATOL, rtol=0)
        (float(g["loss_weight"]) * torch.mean(torch.abs(img - dev("target")))).backward()
    assert counter.calls == 1
    assert np.abs(np.nan_to_num(g["d_shininess"])).max() > 1e-4
    for k, t in leaves.items():
        _assert_close_where_reference_is_finite(t.grad.cpu().numpy(), g["d_" + k], k)
    if "filled" in name or "image" in name:
        assert all(np.isfinite(g["d_" + k]).all() for k in keys)
        return
    # Where the reference returns NaN (per-vertex exponents with a background pixel) the kernels are
    # pinned against the oracle with the power evaluated inside render()'s mask only: same values,
    # finite gradients (oracle/shading.py, power_inside_mask_only).
    from oracle import shading
    cpu = {k: torch.tensor(g[k]).requires_grad_(True) for k in keys}
    ref = shading.render(cpu["vertices"], torch.tensor(g["triangles"]), cpu["normals"], cpu["diffuse"],
                         torch.tensor(g["eye"]), torch.tensor(g["center"]), torch.tensor(g["up"]),
                         cpu["light_positions"], cpu["light_intensities"], 64, 48,
                         specular_colors=cpu["specular"], shininess_coefficients=cpu["shininess"],
                         ambient_color=cpu["ambient"], fov_y=float(g["fov_y"]), power_inside_mask_only=True)
    np.testing.assert_allclose(ref.detach().numpy(), g["image"], atol=2e-6, rtol=0)
    (float(g["loss_weight"]) * torch.mean(torch.abs(ref - torch.tensor(g["target"])))).backward()
    n_nan = 0
    for k, t in leaves.items():
        want = cpu[k].grad.numpy()
        assert np.isfinite(want).all(), k
        n_nan += int((~np.isfinite(g["d_" + k])).sum())
        finite = np.isfinite(g["d_" + k])
        np.testing.assert_allclose(want[finite], g["d_" + k][finite], atol=1e-5, rtol=0, err_msg=k)  # same function
        np.testing.assert_allclose(t.grad.cpu().numpy(), want, atol=ATOL, rtol=0, err_msg=k)
    assert n_nan > 0


def test_renders_simple_and_perspective_triangle_png(device):
    """rasterize_triangles_test.py:72-77."""
    tris = torch.tensor([[0, 1, 2]], dtype=torch.int32, device=device)
    base = np.array([[-0.5, -0.5, 0.8, 1.0], [0.0, 0.5, 0.3, 1.0], [0.5, -0.5, 0.3, 1.0]], np.float32)
    for w_vec, png in (((1.0, 1.0, 1.0), "Simple_Triangle.png"),
                       ((0.2, 0.5, 2.0), "Perspective_Corrected_Triangle.png")):
        clip = torch.tensor(base * np.reshape(np.array(w_vec, np.float32), [3, 1]), device=device)
        _, bary, _ = rasterize_barycentric(clip, tris, 640, 480)
        image = torch.cat([bary, torch.ones(480, 640, 1, device=device)], dim=2)
        expect_image_file_and_render_are_near(png, image)


def test_renders_two_cubes_in_batch_png(device):
    """rasterize_triangles_test.py:79-117."""
    cube = CUBE_V.to(device)
    rgba = torch.cat([cube * 0.5 + 0.5, torch.ones(8, 1, device=device)], dim=1)
    persp = camera_utils.perspective(640 / 480, torch.tensor([40.0], device=device),
                                     torch.tensor([0.01], device=device),
                                     torch.tensor([10.0], device=device))
    center = torch.zeros(1, 3, device=device)
    up = torch.tensor([[0.0, 1.0, 0.0]], device=device)
    looks = [camera_utils.look_at(torch.tensor([eye], device=device), center, up)
             for eye in ([2.0, 3.0, 6.0], [-3.0, 1.0, 6.0])]
    projection = torch.cat([torch.matmul(persp, l) for l in looks], dim=0)
    rendered = mesh_renderer.rasterize(torch.stack([cube, cube]), torch.stack([rgba, rgba]),
                                       CUBE_T.to(device), projection, 640, 480,
                                       torch.tensor([0.0, 0.0, 0.0, 0.0]))
    for i in (0, 1):
        expect_image_file_and_render_are_near("Unlit_Cube_%d.png" % i, rendered[i])


def _gray_cube_scene(device, vertices=None):
    cube = CUBE_V.to(device) if vertices is None else vertices
    normals = torch.nn.functional.normalize(CUBE_V, dim=1, p=2).to(device)
    rot = camera_utils.euler_matrices(
        torch.tensor([[-20.0, 0.0, 60.0], [45.0, 60.0, 0.0]], device=device))[:, :3, :3]
    vw = torch.matmul(torch.stack([cube, cube]), rot.transpose(1, 2))
    nw = torch.matmul(torch.stack([normals, normals]), rot.transpose(1, 2))
    return vw, nw


def test_renders_simple_cube_png(device):
    """mesh_renderer_test.py:30-70."""
    vw, nw = _gray_cube_scene(device)
    eye = torch.tensor(2 * [[0.0, 0.0, 6.0]], device=device)
    images = mesh_renderer.render(vw, CUBE_T.to(device), nw, torch.ones_like(vw), eye,
                                  torch.zeros(2, 3, device=device),
                                  torch.tensor(2 * [[0.0, 1.0, 0.0]], device=device),
                                  eye.unsqueeze(1), torch.ones(2, 1, 3, device=device), 640, 480)
    for i in range(2):
        expect_image_file_and_render_are_near("Gray_Cube_%d.png" % i, images[i])


def test_complex_shading_runs_and_broadcasts(device):
    """mesh_renderer_test.py:72-149 (the reference asserts nothing; we also check that
    per-vertex and scalar shininess agree, and compare with the CPU oracle)."""
    vw, nw = _gray_cube_scene(device)
    g = torch.Generator().manual_seed(2)
    kd, ks = torch.rand(2, 8, 3, generator=g), torch.rand(2, 8, 3, generator=g)
    eye = torch.tensor([[0.0, 0.0, 6.0], [0.0, 0.2, 18.0]])
    center = torch.tensor([[0.0, 0.0, 0.0], [0.1, -0.1, 0.1]])
    up = torch.tensor([[0.0, 1.0, 0.0], [0.1, 1.0, 0.15]])
    lpos = torch.tensor([[[0.0, 0.0, 6.0], [1.0, 2.0, 6.0]], [[0.0, -2.0, 4.0], [1.0, 3.0, 4.0]]])
    lint = torch.tensor([[[1.0, 1.0, 1.0], [1.0, 1.0, 1.0]], [[2.0, 0.0, 1.0], [0.0, 2.0, 1.0]]])
    amb = torch.tensor([[0.0, 0.0, 0.0], [0.1, 0.1, 0.2]])
    d = lambda t: t.to(device)
    kw = dict(ambient_color=d(amb), fov_y=d(torch.tensor([40.0, 13.3])),
              near_clip=d(torch.tensor(0.1)), far_clip=d(torch.tensor(25.0)))
    a = mesh_renderer.render(vw, CUBE_T.to(device), nw, d(kd), d(eye), d(center), d(up), d(lpos),
                             d(lint), 160, 120, d(ks), 6.0 * torch.ones(2, 8, device=device), **kw)
    b = mesh_renderer.render(vw, CUBE_T.to(device), nw, d(kd), d(eye), d(center), d(up), d(lpos),
                             d(lint), 160, 120, d(ks), 6.0, **kw)
    np.testing.assert_allclose(a.cpu().numpy(), b.cpu().numpy(), atol=1e-5, rtol=0)
    tm = mesh_renderer.tone_mapper(a[..., 0:3], 0.7)
    assert float(tm.max()) <= 1.0 and float(tm.min()) >= 0.0


def test_full_render_jacobian_vs_oracle(device):
    """Counterpart of testFullRenderGradientComputation (28x21): instead of finite
    differences, rows of the Jacobian are compared with the CPU oracle's autograd."""
    def scene(vertices, dev):
        normals = torch.nn.functional.normalize(CUBE_V, dim=1, p=2).to(dev)
        rot = camera_utils.euler_matrices(
            torch.tensor([[-20.0, 0.0, 60.0], [45.0, 60.0, 0.0]]))[:, :3, :3].to(dev)
        vw = torch.matmul(torch.stack([vertices, vertices]), rot.transpose(1, 2))
        nw = torch.matmul(torch.stack([normals, normals]), rot.transpose(1, 2))
        eye = torch.tensor(2 * [[0.0, 0.0, 6.0]], device=dev)
        return vw, nw, eye

    vg = CUBE_V.clone().to(device).requires_grad_(True)
    vw, nw, eye = scene(vg, device)
    img = mesh_renderer.render(vw, CUBE_T.to(device), nw, torch.ones_like(vw), eye,
                               torch.zeros(2, 3, device=device),
                               torch.tensor(2 * [[0.0, 1.0, 0.0]], device=device), eye.unsqueeze(1),
                               torch.ones(2, 1, 3, device=device), 28, 21)
    vc = CUBE_V.clone().requires_grad_(True)
    vw_c, nw_c, eye_c = scene(vc, torch.device("cpu"))
    ref = shading.render(vw_c, CUBE_T, nw_c, torch.ones_like(vw_c), eye_c, torch.zeros(2, 3),
                         torch.tensor(2 * [[0.0, 1.0, 0.0]]), eye_c.unsqueeze(1), torch.ones(2, 1, 3),
                         28, 21)
    np.testing.assert_allclose(img.detach().cpu().numpy(), ref.detach().numpy(), atol=ATOL, rtol=0)
    rng = np.random.default_rng(0)
    for _ in range(6):  # random projections of the Jacobian
        w = torch.tensor(rng.normal(size=tuple(ref.shape)).astype(np.float32)) / ref.numel()
        (gd,) = torch.autograd.grad(img, vg, w.to(device), retain_graph=True)
        (gc,) = torch.autograd.grad(ref, vc, w, retain_graph=True)
        np.testing.assert_allclose(gd.cpu().numpy(), gc.numpy(), atol=ATOL, rtol=0)


def test_that_cube_rotates(device):
    """mesh_renderer_test.py:204-271: the gradients are useful for optimisation."""
    cube = CUBE_V.to(device)
    normals = torch.nn.functional.normalize(cube, dim=1, p=2)
    eye = torch.tensor([[0.0, 0.0, 6.0]], device=device)

    def render_with_rotation(angles):
        rot = camera_utils.euler_matrices(angles)[0, :3, :3]
        vw = torch.matmul(cube, rot.T).reshape(1, 8, 3)
        nw = torch.matmul(normals, rot.T).reshape(1, 8, 3)
        out = mesh_renderer.render(vw, CUBE_T.to(device), nw, torch.ones_like(vw), eye,
                                   torch.zeros(1, 3, device=device),
                                   torch.tensor([[0.0, 1.0, 0.0]], device=device),
                                   eye.reshape(1, 1, 3), torch.ones(1, 1, 3, device=device), 640, 480)
        return out.reshape(480, 640, 4)

    angles = torch.zeros(1, 3, device=device, requires_grad=True)
    desired = render_with_rotation(torch.tensor([[-20.0, 0.0, 60.0]], device=device))
    optimizer = torch.optim.SGD([angles], 0.7, 0.1)

    def step():
        optimizer.zero_grad()
        loss = torch.mean(torch.abs(render_with_rotation(angles) - desired))
        loss.backward()
        torch.nn.utils.clip_grad_norm_([angles], 1.0)
        return loss

    for _ in range(35):
        optimizer.step(step)
    expect_image_file_and_render_are_near("Gray_Cube_0.png", desired)
    expect_image_file_and_render_are_near("Gray_Cube_0.png", render_with_rotation(angles).detach(),
                                          max_outlier_fraction=0.01, pixel_error_threshold=0.04)


def test_render_sphere_matches_oracle_at_256(device):
    """The bench workload (smaller): full render fwd + grads vs the CPU oracle."""
    job = synthetic.sphere_job(2, 256, 256, 50)
    cpu = {k: job[k].clone().requires_grad_(True) for k in ("vertices", "normals", "diffuse")}
    gpu = {k: job[k].clone().to(device).requires_grad_(True) for k in ("vertices", "normals", "diffuse")}
    up = torch.tensor(2 * [[0.0, 1.0, 0.0]])
    ref = shading.render(cpu["vertices"], job["triangles"], cpu["normals"], cpu["diffuse"], job["eyes"],
                         torch.zeros(2, 3), up, job["light_positions"], job["light_intensities"], 256, 256)
    img = mesh_renderer.render(gpu["vertices"], job["triangles"].to(device), gpu["normals"],
                               gpu["diffuse"], job["eyes"].to(device), torch.zeros(2, 3, device=device),
                               up.to(device), job["light_positions"].to(device),
                               job["light_intensities"].to(device), 256, 256)
    np.testing.assert_allclose(img.detach().cpu().numpy(), ref.detach().numpy(), atol=ATOL, rtol=0)
    target = torch.rand(ref.shape, generator=torch.Generator().manual_seed(1))
    torch.mean(torch.abs(ref - target)).backward()
    torch.mean(torch.abs(img - target.to(device))).backward()
    for k in cpu:
        np.testing.assert_allclose(gpu[k].grad.cpu().numpy(), cpu[k].grad.numpy(), atol=ATOL, rtol=0,
                                   err_msg=k)


def test_l1_loss_matches_torch(device):
    """losses.l1_loss == torch.mean(torch.abs(a - b)), value and both gradients."""
    gen = torch.Generator().manual_seed(4)
    for shape in ((2, 48, 64, 4), (1, 7, 3, 4), (3, 5), (1,), (2, 3, 3), (1030,)):
        a0, b0 = torch.rand(shape, generator=gen), torch.rand(shape, generator=gen)
        b0.view(-1)[0] = a0.view(-1)[0]  # an exact zero difference: sign(0) = 0
        a1, b1 = a0.clone().to(device).requires_grad_(True), b0.clone().to(device).requires_grad_(True)
        a2, b2 = a0.clone().to(device).requires_grad_(True), b0.clone().to(device).requires_grad_(True)
        l1 = mesh_renderer.losses.l1_loss(a1, b1)
        l2 = torch.mean(torch.abs(a2 - b2))
        assert abs(float(l1) - float(l2)) < 1e-6
        (3.0 * l1).backward()
        (3.0 * l2).backward()
        np.testing.assert_allclose(a1.grad.cpu().numpy(), a2.grad.cpu().numpy(), atol=1e-9, rtol=1e-6)
        np.testing.assert_allclose(b1.grad.cpu().numpy(), b2.grad.cpu().numpy(), atol=1e-9, rtol=1e-6)
        # no gradient wanted: the sign buffer is skipped, the value is the same
        with torch.no_grad():
            l3 = mesh_renderer.losses.l1_loss(a1.detach(), b1.detach())
        assert abs(float(l3) - float(l2)) < 1e-6


def test_to_uint8_matches_numpy_cast(device):
    """to_uint8 == (clip(x, 0, 1) * 255.0).astype(np.uint8), the examples' frame conversion."""
    gen = torch.Generator().manual_seed(9)
    for shape in ((2, 33, 17, 4), (5,), (1, 3, 3, 3)):
        x = torch.rand(shape, generator=gen) * 1.4 - 0.2          # some values outside [0, 1]
        flat = x.view(-1)
        flat[0] = 1.0
        if flat.numel() > 4:
            flat[1], flat[2], flat[3], flat[4] = 0.0, float("nan"), float("inf"), -float("inf")
        got = mesh_renderer.to_uint8(x.to(device)).cpu().numpy()
        xn = np.nan_to_num(x.numpy(), nan=0.0, posinf=1.0, neginf=0.0)
        want = (np.clip(xn, 0.0, 1.0) * np.float32(255.0)).astype(np.uint8)
        assert got.dtype == np.uint8 and got.shape == want.shape
        np.testing.assert_array_equal(got, want)
    with pytest.raises(ValueError):
        mesh_renderer.to_uint8(torch.zeros(4, dtype=torch.float64, device=device))


def _specular_scene(device, n_lights=2, ambient=True):
    job = synthetic.sphere_job(2, 96, 80, 12)
    gen = torch.Generator().manual_seed(3)
    leaf = lambda t: t.clone().to(device).requires_grad_(True)
    # (no light with x + y + z = -3: for the background's attributes of -1 the normal would be exactly
    # perpendicular to the light direction, N.D = 0 sits on the clamp's edge, and float rounding -- in the
    # reference too -- decides whether thousands of background pixels pass a gradient or not)
    all_lights = torch.tensor([[[2.0, 3.0, 4.0], [-3.0, 1.0, 2.5], [0.3, -4.0, 1.0], [1.0, 0.5, 5.0],
                                [4.0, -1.0, 2.0], [-1.5, 3.5, 3.0], [2.5, 2.5, -2.0], [-4.0, -0.5, 2.5], [0.5, 4.5, 0.5]],
                               [[0.5, -2.0, 3.0], [3.0, 3.0, -1.0], [-2.0, 2.0, 2.0], [0.0, 0.0, 4.0],
                                [1.5, -3.5, 2.0], [-3.0, -1.0, 3.5], [2.0, 0.5, 4.5], [-0.5, 2.5, -3.0], [3.5, -2.5, 1.0]]])
    scene = {
        "vertices": leaf(job["vertices"]), "normals": leaf(job["normals"]),
        "diffuse": leaf(torch.rand(job["vertices"].shape, generator=gen)),
        "specular": leaf(torch.rand(job["vertices"].shape, generator=gen)),
        "light_positions": leaf(all_lights[:, :n_lights]),
        "light_intensities": leaf(torch.rand(2, n_lights, 3, generator=gen) + 0.2),
        "eye": leaf(job["eyes"]),
    }
    scene["ambient"] = leaf(torch.rand(2, 3, generator=gen) * 0.3) if ambient else None
    return job, scene


def _render_specular(job, scene, device, shininess, fov_y=40.0):
    return mesh_renderer.render(
        scene["vertices"], job["triangles"].to(device), scene["normals"], scene["diffuse"], scene["eye"],
        torch.zeros(2, 3, device=device), torch.tensor([0.0, 1.0, 0.0], device=device),
        scene["light_positions"], scene["light_intensities"], 96, 80,
        specular_colors=scene["specular"], shininess_coefficients=shininess, ambient_color=scene["ambient"],
        fov_y=fov_y)


@pytest.mark.parametrize("n_lights,ambient", [(1, False), (2, True), (4, True), (6, True), (9, False)])
def test_fused_specular_matches_composed_path(device, n_lights, ambient):
    """render() with a per-image shininess: fused HIP kernels vs the composed path (HIP raster +
    interpolation, torch Phong and autograd), image and every gradient."""
    render_mod = sys.modules["pytorch_mesh_renderer_amd.mesh_renderer.render"]
    shininess = torch.tensor([1.0, 2.5], device=device)
    gen = torch.Generator().manual_seed(5)
    target = torch.rand(2, 80, 96, 4, generator=gen).to(device)
    results = {}
    for fused in (True, False):
        job, scene = _specular_scene(device, n_lights, ambient)
        render_mod.USE_FUSED_SHADING = fused
        try:
            with _CountCalls("shade_specular_forward") as counter:
                img = _render_specular(job, scene, device, shininess)
        finally:
            render_mod.USE_FUSED_SHADING = True
        assert counter.calls == ((n_lights + 3) // 4 if fused else 0)   # four lights per pass
        # the specular term is tiny after the across-pixels normalisation: weight it up
        (torch.mean(torch.abs(img - target)) * 50.0).backward()
        results[fused] = (img.detach().cpu().numpy(), {k: v.grad.cpu().numpy() for k, v in scene.items()
                                                       if v is not None and v.grad is not None})
    img_f, grads_f = results[True]
    img_c, grads_c = results[False]
    np.testing.assert_allclose(img_f, img_c, atol=ATOL, rtol=0)
    expected = {"vertices", "normals", "diffuse", "specular", "light_positions", "light_intensities", "eye"}
    assert set(grads_c) == set(grads_f) == (expected | {"ambient"} if ambient else expected)
    for k in grads_c:
        assert np.isfinite(grads_f[k]).all() and np.abs(grads_c[k]).max() > 0, k
        np.testing.assert_allclose(grads_f[k], grads_c[k], atol=ATOL, rtol=0, err_msg=k)


def test_image_wide_sums_are_bit_reproducible(device):
    """The sums that run over a whole image -- the specular term's across-pixels norm, and the light,
    ambient, camera and per-image shininess gradients -- are fixed-order sums of per-workgroup partials
    (no float atomics): two runs give the same bits, in the default mode, with 4 lights and ambient."""
    # light, ambient and shininess gradients; the image (through the norm)
    shininess = torch.tensor([1.0, 2.5], device=device)
    gen = torch.Generator().manual_seed(5)
    target = torch.rand(2, 80, 96, 4, generator=gen).to(device)
    runs = []
    for _ in range(2):
        job, scene = _specular_scene(device, 4, True)
        shin = shininess.clone().requires_grad_(True)
        img = _render_specular(job, scene, device, shin)
        (torch.mean(torch.abs(img - target)) * 50.0).backward()
        # (not the eye: its gradient also runs through the clip-space transform, i.e. through the
        # per-triangle sums, which are float atomics outside the deterministic mode)
        runs.append((img.detach(), [scene[k].grad for k in ("light_positions", "light_intensities", "ambient")]
                     + [shin.grad]))
        # ... and the diffuse path's light gradients (dense upstream)
        job, scene = _specular_scene(device, 4, True)
        img = mesh_renderer.render(
            scene["vertices"], job["triangles"].to(device), scene["normals"], scene["diffuse"], scene["eye"],
            torch.zeros(2, 3, device=device), torch.tensor([0.0, 1.0, 0.0], device=device),
            scene["light_positions"], scene["light_intensities"], 96, 80, ambient_color=scene["ambient"])
        torch.mean(torch.abs(img - target)).backward()
        runs[-1][1].extend(scene[k].grad for k in ("light_positions", "light_intensities", "ambient"))
    assert torch.equal(runs[0][0], runs[1][0])          # the image depends on the norm
    for a, b in zip(runs[0][1], runs[1][1]):
        assert a is not None and float(a.abs().max()) > 0 and torch.equal(a, b)


def test_deterministic_mode_covers_specular_and_rasterize_backward(device):
    """Round 3: mr_set_deterministic also covers the specular backward (fixed-point rows + per-vertex
    gather over the adjacency instead of k_spec_scatter's float atomics) and the fused interpolation
    backward of rasterize(): every gradient is bit-identical between two runs, and equals the default
    float-atomic kernels within the parity tolerance.  Sizes with thousands of merge-table flushes per
    image (a 50-subdivision sphere at 320x240)."""
    from pytorch_mesh_renderer_amd import _native
    job = synthetic.sphere_job(2, 320, 240, 50)
    gen = torch.Generator().manual_seed(17)
    target = torch.rand(2, 240, 320, 4, generator=gen).to(device)
    base = {"vertices": job["vertices"], "normals": job["normals"],
            "diffuse": torch.rand(job["vertices"].shape, generator=gen),
            "specular": torch.rand(job["vertices"].shape, generator=gen),
            "shininess": 0.3 + torch.rand(2, job["vertices"].shape[1], generator=gen)}
    attrs = torch.rand(2, job["vertices"].shape[1], 7, generator=gen)
    weights = torch.randn(2, 240, 320, 7, generator=gen).to(device) / (240 * 320)

    def run():
        leaves = {k: v.clone().to(device).requires_grad_(True) for k, v in base.items()}
        img = mesh_renderer.render(leaves["vertices"], job["triangles"].to(device), leaves["normals"], leaves["diffuse"],
                                   job["eyes"], torch.zeros(2, 3), torch.tensor([0.0, 1.0, 0.0]),
                                   job["light_positions"].to(device), job["light_intensities"].to(device), 320, 240,
                                   specular_colors=leaves["specular"], shininess_coefficients=leaves["shininess"])
        (torch.mean(torch.abs(img - target)) * 20.0).backward()
        grads = [leaves[k].grad.clone() for k in sorted(leaves)]
        clip = job["clip"].clone().to(device).requires_grad_(True)
        a = attrs.clone().to(device).requires_grad_(True)
        from pytorch_mesh_renderer_amd.mesh_renderer.rasterize import rasterize_clip_space
        out = rasterize_clip_space(clip, a, job["triangles"].to(device), 320, 240,
                                   torch.zeros(7, device=device))
        (out * weights).sum().backward()
        return grads + [clip.grad.clone(), a.grad.clone()]

    default = run()
    before = _native.set_deterministic(True)
    try:
        first, second = run(), run()
    finally:
        _native.set_deterministic(before)
    for i, (a, b, d) in enumerate(zip(first, second, default)):
        assert float(a.abs().max()) > 0, i
        assert torch.equal(a, b), "output %d differs between two deterministic runs" % i
        scale = float(d.abs().max())
        np.testing.assert_allclose(a.cpu().numpy(), d.cpu().numpy(), atol=max(1e-4 * scale, 1e-9), rtol=1e-3,
                                   err_msg="deterministic vs default, output %d" % i)


@pytest.mark.parametrize("batch,size,k,n_lights", [(2, (96, 72), 10, 1), (16, (512, 512), 50, 2), (3, (333, 257), 20, 4),
                                                   (17, (512, 448), 100, 3)])
def test_rasterizer_forms_the_specular_norms_in_its_own_pass(device, batch, size, k, n_lights):
    """mr_rasterize_specular_norms_forward (round 5): the G-buffer bit for bit mr_rasterize_forward's, and the
    across-pixels norms of render.py:342-348 equal to those of shade_spec.hip's norm pass over that G-buffer -- 32- and
    64-pixel regions, ragged sizes, 1..4 lights, a crowded mesh (20k triangles: regions whose records do not fit the
    bin's top, several bin rounds); and render() with the specular term gives the same image either way."""
    from pytorch_mesh_renderer_amd import _native
    from pytorch_mesh_renderer_amd.mesh_renderer import rasterize_triangles_ext as ext
    w, h = size
    job = synthetic.sphere_job(batch, w, h, k)
    d = {key: (v.to(device) if torch.is_tensor(v) else v) for key, v in job.items()}
    gen = torch.Generator().manual_seed(batch)
    lp = (torch.randn(batch, n_lights, 3, generator=gen) * 3.0 + torch.tensor([0.0, 0.0, 4.0])).to(device)
    li = torch.rand(batch, n_lights, 3, generator=gen).to(device)
    cam = d["eyes"].to(device)
    ks = torch.rand(d["vertices"].shape, generator=gen).to(device)
    shin = torch.full((batch,), 5.0, device=device)
    ids, bary, z = _native.rasterize_forward(d["clip"], d["triangles"], w, h)
    rgba, norms = _native.shade_specular_forward(ids, bary, d["normals"], d["vertices"], d["diffuse"], ks, d["triangles"], lp, li,
                                                 None, cam, shin)
    ids2, bary2, z2, norms2 = _native.rasterize_specular_norms_forward(d["clip"], d["triangles"], d["normals"], d["vertices"],
                                                                       lp, cam, w, h, want_z=True)
    assert torch.equal(ids2, ids) and torch.equal(bary2.view(torch.int32), bary.view(torch.int32))
    assert torch.equal(z2.view(torch.int32), z.view(torch.int32))
    np.testing.assert_allclose(norms2.cpu().numpy(), norms.cpu().numpy(), rtol=2e-5, atol=0)
    assert float(norms.min()) > 0
    images = []
    for fused in (True, False):
        ext.FUSE_SPECULAR_NORMS = fused
        try:
            images.append(mesh_renderer.render(d["vertices"], d["triangles"], d["normals"], d["diffuse"], job["eyes"],
                                               torch.zeros(batch, 3), torch.tensor([0.0, 1.0, 0.0]), lp, li, w, h,
                                               specular_colors=ks, shininess_coefficients=5.0))
        finally:
            ext.FUSE_SPECULAR_NORMS = True
    np.testing.assert_allclose(images[0].cpu().numpy(), images[1].cpu().numpy(), atol=2e-5, rtol=0)


def test_specular_backward_gather_matches_scatter(device):
    """mr_shade_specular_backward with the CSR vertex adjacency (per-vertex gather, what render() uses
    since round 3) vs without it (float-atomic scatter), per-vertex shininess included."""
    from pytorch_mesh_renderer_amd import _native
    job = synthetic.sphere_job(2, 120, 90, 10)
    d = {k: (v.to(device) if torch.is_tensor(v) else v) for k, v in job.items()}
    gen = torch.Generator().manual_seed(3)
    V = d["vertices"].shape[1]
    spec = torch.rand(2, V, 3, generator=gen).to(device)
    ids, bary, _ = _native.rasterize_forward(d["clip"], d["triangles"], 120, 90)
    cam = job["eyes"].to(device)
    g = torch.randn(2, 90, 120, 4, generator=gen).to(device) / (90 * 120)
    for shin in (torch.tensor([1.5, 0.7], device=device), (0.3 + torch.rand(2, V, generator=gen)).to(device)):
        rgba, norms2 = _native.shade_specular_forward(ids, bary, d["normals"], d["vertices"], d["diffuse"], spec,
                                                      d["triangles"], d["light_positions"], d["light_intensities"],
                                                      None, cam, shin)
        args = (g, ids, bary, d["clip"], d["normals"], d["vertices"], d["diffuse"], spec, d["triangles"],
                d["light_positions"], d["light_intensities"], None, cam, shin, norms2)
        scatter = _native.shade_specular_backward(*args)
        gather = _native.shade_specular_backward(*args, adjacency=_native.vertex_adjacency(d["triangles"], V))
        for i, (a, b) in enumerate(zip(scatter, gather)):
            if a is None:
                assert b is None
                continue
            assert float(a.abs().max()) > 0, i
            np.testing.assert_allclose(b.cpu().numpy(), a.cpu().numpy(), atol=1e-7 * max(1.0, float(a.abs().max()) * 1e3),
                                       rtol=1e-4, err_msg="output %d" % i)


@pytest.mark.parametrize("n_lights,per_vertex", [(1, False), (2, True), (3, False), (4, True)])
def test_specular_backward_lane_kernel_matches_rows_kernel(device, n_lights, per_vertex):
    """Round 4: mr_shade_specular_backward's grads_wanted.  With the vertex gradients alone wanted
    (MR_GRAD_POSITIONS | MR_GRAD_CLIP) on the rasterizer's own G-buffer, the pixel pass is the lane-accumulating
    kernel (SpecFoldLaneFn, 18 sums in a difference basis); with the transforms and no clip gradient wanted, the
    pull-back is folded in (9 sums).  Same d clip / d positions as the rows kernel's 45-sum pass."""
    from pytorch_mesh_renderer_amd import _native
    job = synthetic.sphere_job(2, 200, 150, 12)
    d = {k: (v.to(device) if torch.is_tensor(v) else v) for k, v in job.items()}
    gen = torch.Generator().manual_seed(n_lights)
    V = d["vertices"].shape[1]
    spec = torch.rand(2, V, 3, generator=gen).to(device)
    diffuse = torch.rand(2, V, 3, generator=gen).to(device)
    lp = (torch.randn(2, n_lights, 3, generator=gen) * 3.0 + torch.tensor([0.0, 0.0, 4.0])).to(device)
    li = (torch.rand(2, n_lights, 3, generator=gen) + 0.2).to(device)
    amb = (torch.rand(2, 3, generator=gen) * 0.3).to(device)
    xf = synthetic.clip_transforms(job["eyes"], 200, 150).to(device)
    clip = _native.vertex_transform(d["vertices"], xf)
    ids, bary, _ = _native.rasterize_forward(clip, d["triangles"], 200, 150)
    cam = job["eyes"].to(device)
    g = torch.randn(2, 150, 200, 4, generator=gen).to(device) / (150 * 200)
    shin = (0.3 + 2.0 * torch.rand(2, V, generator=gen)).to(device) if per_vertex else torch.tensor([1.5, 3.0], device=device)
    rgba, norms2 = _native.shade_specular_forward(ids, bary, d["normals"], d["vertices"], diffuse, spec, d["triangles"],
                                                  lp, li, amb, cam, shin)
    args = (g, ids, bary, clip, d["normals"], d["vertices"], diffuse, spec, d["triangles"], lp, li, amb, cam, shin, norms2)
    adjacency = _native.vertex_adjacency(d["triangles"], V)
    rows = _native.shade_specular_backward(*args, adjacency=adjacency)
    if True:
        lanes = _native.shade_specular_backward(*args, adjacency=adjacency, normalised_gbuffer=True,
                                                grads_wanted=_native.GRAD_POSITIONS | _native.GRAD_CLIP)
        folded = _native.shade_specular_backward(*args, adjacency=adjacency, normalised_gbuffer=True, transforms=xf,
                                                 grads_wanted=_native.GRAD_POSITIONS)
        scattered = _native.shade_specular_backward(*args, normalised_gbuffer=True,
                                                    grads_wanted=_native.GRAD_POSITIONS | _native.GRAD_CLIP)
    dclip, dpos = rows[0].cpu().numpy(), rows[2].cpu().numpy()
    assert np.abs(dclip).max() > 0 and np.abs(dpos).max() > 0
    for name, got in (("lanes", lanes), ("lanes + scatter", scattered)):
        np.testing.assert_allclose(got[0].cpu().numpy(), dclip, rtol=2e-4, atol=2e-6 * np.abs(dclip).max(), err_msg=name)
        np.testing.assert_allclose(got[2].cpu().numpy(), dpos, rtol=2e-4, atol=2e-6 * np.abs(dpos).max(), err_msg=name)
    whole = dpos + np.einsum("bij,bvi->bvj", xf.cpu().numpy()[:, :, :3], dclip)
    np.testing.assert_allclose(folded[2].cpu().numpy(), whole, rtol=2e-4, atol=2e-6 * np.abs(whole).max())
    # through render(): differentiated to the vertices alone (the folded pass) against all leaves requiring grad
    shininess = shin if per_vertex else torch.tensor([1.5, 3.0], device=device)
    grads = {}
    for only_vertices in (True, False):
        v = d["vertices"].clone().requires_grad_(True)
        n = d["normals"].clone().requires_grad_(not only_vertices)
        img = mesh_renderer.render(v, d["triangles"], n, diffuse, job["eyes"], torch.zeros(2, 3),
                                   torch.tensor([0.0, 1.0, 0.0]), lp, li, 200, 150, specular_colors=spec,
                                   shininess_coefficients=shininess, ambient_color=amb)
        (img * g).sum().backward()
        grads[only_vertices] = v.grad.cpu().numpy()
    np.testing.assert_allclose(grads[True], grads[False], rtol=2e-4, atol=2e-6 * np.abs(grads[False]).max())
    np.testing.assert_allclose(grads[True], whole, rtol=2e-4, atol=2e-6 * np.abs(whole).max())


@pytest.mark.parametrize("kind", ["vertex", "image", "scalar"])
def test_fused_specular_shininess_gradient_matches_composed_path(device, kind):
    """A shininess that requires grad -- [B,V], [B] or 0-D -- through the fused kernels vs torch autograd
    over the composed path, on the 5k-style sphere zoomed in until it fills the frame (no background pixel,
    where the composed per-vertex path, like the reference, returns NaN)."""
    render_mod = sys.modules["pytorch_mesh_renderer_amd.mesh_renderer.render"]
    gen = torch.Generator().manual_seed(5)
    target = torch.rand(2, 80, 96, 4, generator=gen).to(device)
    results = {}
    for fused in (True, False):
        job, scene = _specular_scene(device, 3, True)
        g2 = torch.Generator().manual_seed(17)
        base = {"vertex": 0.3 + 1.2 * torch.rand(2, scene["vertices"].shape[1], generator=g2),
                "image": torch.tensor([0.5, 1.5]), "scalar": torch.tensor(0.8)}[kind]
        scene["shininess"] = base.to(device).requires_grad_(True)
        render_mod.USE_FUSED_SHADING = fused
        try:
            with _CountCalls("shade_specular_backward") as counter:
                img = _render_specular(job, scene, device, scene["shininess"], fov_y=12.0)
                assert float(img.detach()[..., 3].min()) == 1.0     # the sphere fills the frame
                (torch.mean(torch.abs(img - target)) * 50.0).backward()
        finally:
            render_mod.USE_FUSED_SHADING = True
        assert counter.calls == (1 if fused else 0)
        results[fused] = (img.detach().cpu().numpy(), {k: v.grad.cpu().numpy() for k, v in scene.items()
                                                       if v is not None and v.grad is not None})
    img_f, grads_f = results[True]
    img_c, grads_c = results[False]
    np.testing.assert_allclose(img_f, img_c, atol=ATOL, rtol=0)
    assert set(grads_c) == set(grads_f) and "shininess" in grads_f
    assert grads_f["shininess"].shape == grads_c["shininess"].shape
    for k in grads_c:
        assert np.isfinite(grads_c[k]).all() and np.isfinite(grads_f[k]).all(), k
        assert np.abs(grads_c[k]).max() > 0, k
        np.testing.assert_allclose(grads_f[k], grads_c[k], atol=ATOL, rtol=0, err_msg=k)


@pytest.mark.parametrize("seed", [0, 1, 2, 3])
def test_render_forward_on_triangle_soup(device, seed):
    """mr_render_forward vs mr_rasterize_forward + mr_shade_forward on random triangle soups under random
    projective transforms: vertices behind the eye (w < 0, the full-screen bbox path), slivers,
    interpenetrating and repeated triangles, out-of-range vertex indices, negative diffuse colours
    (render()'s mask), several lights.  G-buffer bit for bit, RGBA within the shading budget."""
    from pytorch_mesh_renderer_amd import _native
    gen = torch.Generator().manual_seed(100 + seed)
    B, V, T, W, H = 3, 70, 160, 97, 61
    vertices = (torch.rand(B, V, 3, generator=gen) * 2 - 1).to(device)
    tris = torch.randint(0, V, (T, 3), generator=gen, dtype=torch.int32)
    tris[5] = tris[4]                    # a repeated triangle: the depth tie goes to the larger id
    tris[9, 2] = V + 3                   # out of range: never drawn
    tris[11, 1] = tris[11, 0]            # degenerate
    tris = tris.to(device)
    xf = torch.eye(4).repeat(B, 1, 1)
    xf[:, 3, 2] = torch.rand(B, generator=gen) * 2 - 1           # w = c z + d: some vertices get w < 0
    xf[:, 3, 3] = torch.rand(B, generator=gen) * 0.8 + 0.2
    xf[:, :3, 3] = (torch.rand(B, 3, generator=gen) - 0.5) * 0.3
    xf = xf.to(device)
    normals = torch.randn(B, V, 3, generator=gen).to(device)
    diffuse = (torch.rand(B, V, 3, generator=gen) * 1.2 - 0.2).to(device)    # some negative: masked pixels
    n_lights = 1 + seed % 4
    lp = (torch.rand(B, n_lights, 3, generator=gen) * 6 - 3).to(device)
    li = (torch.rand(B, n_lights, 3, generator=gen) + 0.1).to(device)
    amb = (torch.rand(B, 3, generator=gen) * 0.3).to(device) if seed % 2 else None
    clip = _native.vertex_transform(vertices, xf)
    assert int((clip[..., 3] < 0).sum()) > 0
    ids, bary, z = _native.rasterize_forward(clip, tris, W, H)
    rgba = _native.shade_forward(ids, bary, normals, vertices, diffuse, tris, lp, li, amb)
    covered = float((rgba[..., 3] > 0).float().mean())
    assert covered > 0.05
    clip2, ids2, bary2, z2, rgba2, _ = _native.render_forward(vertices, xf, normals, diffuse, tris, lp, li, amb, W, H)
    assert torch.equal(clip2, clip) and torch.equal(ids2, ids) and torch.equal(bary2, bary) and torch.equal(z2, z)
    assert torch.equal(rgba2[..., 3], rgba[..., 3])
    np.testing.assert_allclose(rgba2.cpu().numpy(), rgba.cpu().numpy(), atol=2e-6, rtol=1e-6)


def test_fused_diffuse_render_matches_composed_path_including_the_camera(device):
    """render() without a specular term: FusedPhongRenderer (clip transform, rasterizer, shading and their
    backward inside the library; camera gradient through one batched product) vs the composed path (torch
    clip transform and Phong over the HIP rasterizer / interpolation, torch autograd): image and the
    gradients w.r.t. vertices, normals, colours, lights, ambient AND the eye."""
    render_mod = sys.modules["pytorch_mesh_renderer_amd.mesh_renderer.render"]
    job = synthetic.sphere_job(2, 112, 84, 12)
    gen = torch.Generator().manual_seed(9)
    target = torch.rand(2, 84, 112, 4, generator=gen).to(device)
    results = {}
    for fused in (True, False):
        leaf = lambda t: t.clone().to(device).requires_grad_(True)
        scene = {"vertices": leaf(job["vertices"]), "normals": leaf(job["normals"]),
                 "diffuse": leaf(torch.rand(job["vertices"].shape, generator=torch.Generator().manual_seed(4))),
                 "light_positions": leaf(torch.tensor([[[2.0, 3.0, 4.0], [-3.0, 1.0, 2.5]],
                                                       [[0.5, -2.0, 3.0], [3.0, 3.0, -1.0]]])),
                 "light_intensities": leaf(torch.rand(2, 2, 3, generator=torch.Generator().manual_seed(5)) + 0.2),
                 "ambient": leaf(torch.rand(2, 3, generator=torch.Generator().manual_seed(6)) * 0.3),
                 "eye": leaf(job["eyes"])}
        render_mod.USE_FUSED_SHADING = fused
        try:
            with _CountCalls("render_forward") as counter:
                img = mesh_renderer.render(scene["vertices"], job["triangles"].to(device), scene["normals"],
                                           scene["diffuse"], scene["eye"], torch.zeros(2, 3, device=device),
                                           torch.tensor([0.0, 1.0, 0.0], device=device), scene["light_positions"],
                                           scene["light_intensities"], 112, 84, ambient_color=scene["ambient"])
            (torch.mean(torch.abs(img - target)) * 10.0).backward()
        finally:
            render_mod.USE_FUSED_SHADING = True
        assert counter.calls == (1 if fused else 0)
        results[fused] = (img.detach().cpu().numpy(), {k: v.grad.cpu().numpy() for k, v in scene.items()})
    np.testing.assert_allclose(results[True][0], results[False][0], atol=ATOL, rtol=0)
    for k, want in results[False][1].items():
        assert np.abs(want).max() > 1e-4, k
        np.testing.assert_allclose(results[True][1][k], want, atol=ATOL, rtol=0, err_msg=k)


def test_render_emits_uint8_frames_on_request(device):
    """rasterize_triangles_ext.emit_uint8_frames(True): the forward kernel also writes the 8-bit frames;
    to_uint8(image) hands them out (no conversion pass) and they equal mr_export_u8 of the float image;
    an image modified in place falls back to the conversion."""
    import importlib
    ext = importlib.import_module("pytorch_mesh_renderer_amd.mesh_renderer.rasterize_triangles_ext")
    job = synthetic.sphere_job(2, 130, 67, 10)
    d = {k: (v.to(device) if torch.is_tensor(v) else v) for k, v in job.items()}
    def render():
        return mesh_renderer.render(d["vertices"].clone().requires_grad_(True), d["triangles"], d["normals"],
                                    torch.rand(d["vertices"].shape, generator=torch.Generator().manual_seed(1)).to(device) * 1.5,
                                    job["eyes"], torch.zeros(2, 3), torch.tensor([0.0, 1.0, 0.0]),
                                    d["light_positions"], d["light_intensities"] * 1.3, 130, 67)
    plain = render()
    with ext.emit_uint8_frames(True):
        image = render()
    assert getattr(render(), "_mr_frames_u8", None) is None     # the switch ended with its block
    with _CountCalls("export_u8") as counter:
        frames = mesh_renderer.to_uint8(image)
    assert counter.calls == 0 and frames.dtype == torch.uint8 and frames.shape == image.shape
    assert torch.equal(image.detach(), plain.detach())
    from pytorch_mesh_renderer_amd import _native
    assert torch.equal(frames, _native.export_u8(image.detach()))
    assert int(frames[..., :3].max()) == 255                  # over-exposed pixels saturate
    torch.mean(image).backward()                             # the extra output does not disturb autograd
    with torch.no_grad():
        image.mul_(0.5)
    with _CountCalls("export_u8") as counter:
        again = mesh_renderer.to_uint8(image)
    assert counter.calls == 1 and torch.equal(again, _native.export_u8(image.detach()))


def test_render_with_and_without_shading_epilogue(device):
    """render() + L1 loss + backward with FusedPhongRenderer's one-pass forward (mr_render_forward) and with
    its two-kernel forward (torch clip transform, k_raster, k_shade_forward): same image, same gradients --
    camera gradient included (the transforms are differentiated through one batched product)."""
    import importlib
    ext = importlib.import_module("pytorch_mesh_renderer_amd.mesh_renderer.rasterize_triangles_ext")
    job = synthetic.sphere_job(2, 120, 90, 14)
    target = torch.rand(2, 90, 120, 4, generator=torch.Generator().manual_seed(2)).to(device)
    results = {}
    for epilogue in (True, False):
        leaves = {k: job[k].clone().to(device).requires_grad_(True)
                  for k in ("vertices", "normals", "diffuse", "light_positions", "eyes")}
        with ext.shading_epilogue(epilogue):
            with _CountCalls("render_forward") as counter:
                img = mesh_renderer.render(leaves["vertices"], job["triangles"].to(device), leaves["normals"],
                                           leaves["diffuse"], leaves["eyes"], torch.zeros(2, 3, device=device),
                                           torch.tensor([0.0, 1.0, 0.0], device=device), leaves["light_positions"],
                                           job["light_intensities"].to(device), 120, 90)
                mesh_renderer.losses.l1_loss(img, target).backward()
        assert counter.calls == (1 if epilogue else 0)
        results[epilogue] = (img.detach().cpu().numpy(), {k: v.grad.cpu().numpy() for k, v in leaves.items()})
    np.testing.assert_allclose(results[True][0], results[False][0], atol=ATOL, rtol=0)
    assert np.array_equal(results[True][0][..., 3], results[False][0][..., 3])
    for k, want in results[False][1].items():
        assert np.abs(want).max() > 0, k
        np.testing.assert_allclose(results[True][1][k], want, atol=ATOL, rtol=0, err_msg=k)


@pytest.mark.parametrize("res", [52, 70, 110])
def test_render_forward_with_extra_record_slots_matches_raster_then_shade(device, res):
    """Round 5: launches with 32 or more triangles per 64 x 64 pixels of image take k_raster<..., XREC = 64>, whose bin is
    followed by 64 more LDS slots for the shading epilogue's corner records (144 in all instead of 106).  Spheres of
    5.4k / 9.8k / 24k triangles at 192 x 128 with 64-pixel regions forced: regions below 106 entries, between 106 and
    144 (records in the extra slots), above (the scalar-cache loop), and several bin rounds -- all against
    mr_rasterize_forward + mr_shade_forward: G-buffer bit for bit, RGBA to 1e-6."""
    from pytorch_mesh_renderer_amd import _native
    w, h = 192, 128
    job = synthetic.sphere_job(2, w, h, res)
    d = {k: (v.to(device) if torch.is_tensor(v) else v) for k, v in job.items()}
    assert d["triangles"].shape[0] * 4096 >= 32 * w * h, "dense enough for the XREC instantiation"
    gen = torch.Generator().manual_seed(res)
    diffuse = torch.rand(d["vertices"].shape, generator=gen).to(device)
    lp = (torch.rand(2, 2, 3, generator=gen) * 6 - 3).to(device)
    li = (torch.rand(2, 2, 3, generator=gen) + 0.2).to(device)
    xf = synthetic.clip_transforms(job["eyes"], w, h).to(device)
    assert _native.lib().mr_debug_set_raster_region_edge(64) == 0
    try:
        clip = _native.vertex_transform(d["vertices"], xf)
        ids, bary, z = _native.rasterize_forward(clip, d["triangles"], w, h)
        rgba = _native.shade_forward(ids, bary, d["normals"], d["vertices"], diffuse, d["triangles"], lp, li, None)
        _, ids2, bary2, _, rgba2, _ = _native.render_forward(d["vertices"], xf, d["normals"], diffuse, d["triangles"], lp, li,
                                                            None, w, h, want_z=False)
    finally:
        _native.lib().mr_debug_set_raster_region_edge(0)
    assert torch.equal(ids2, ids) and torch.equal(bary2.view(torch.int32), bary.view(torch.int32))
    np.testing.assert_allclose(rgba2.cpu().numpy(), rgba.cpu().numpy(), atol=1e-6, rtol=0)
    assert float(rgba[..., 3].mean()) > 0.3


@pytest.mark.parametrize("w,h,res,n_lights,ambient", [(96, 80, 12, 1, False), (130, 67, 10, 3, True),
                                                        (64, 64, 120, 2, True), (33, 31, 6, 4, False)])
def test_render_forward_matches_raster_then_shade(device, w, h, res, n_lights, ambient):
    """mr_render_forward (shading as the epilogue of the rasterizer's tile walk) vs mr_rasterize_forward +
    mr_shade_forward: the G-buffer bit for bit, RGBA within the shading budget.  Ragged image sizes, 1-4
    lights, ambient, and a 64x64 image with 28k triangles: its regions need several bin rounds, so the
    pixel state makes the round trip through the G-buffer before it is shaded."""
    from pytorch_mesh_renderer_amd import _native
    job = synthetic.sphere_job(2, w, h, res)
    d = {k: (v.to(device) if torch.is_tensor(v) else v) for k, v in job.items()}
    gen = torch.Generator().manual_seed(3)
    diffuse = torch.rand(d["vertices"].shape, generator=gen).to(device)
    lp = (torch.rand(2, n_lights, 3, generator=gen) * 6 - 3).to(device)
    li = (torch.rand(2, n_lights, 3, generator=gen) + 0.2).to(device)
    amb = (torch.rand(2, 3, generator=gen) * 0.3).to(device) if ambient else None
    xf = synthetic.clip_transforms(job["eyes"], w, h).to(device)
    clip = _native.vertex_transform(d["vertices"], xf)
    np.testing.assert_allclose(clip.cpu().numpy(), job["clip"].numpy(), atol=2e-6, rtol=1e-6)
    ids, bary, z = _native.rasterize_forward(clip, d["triangles"], w, h)
    rgba = _native.shade_forward(ids, bary, d["normals"], d["vertices"], diffuse, d["triangles"], lp, li, amb)
    assert float(rgba[..., 3].mean()) > 0.2
    for want_z in (True, False):
        clip2, ids2, bary2, z2, rgba2, records = _native.render_forward(
            d["vertices"], xf, d["normals"], diffuse, d["triangles"], lp, li, amb, w, h, want_z=want_z)
        assert torch.equal(clip2, clip)
        assert torch.equal(ids2, ids) and torch.equal(bary2, bary)
        assert (z2 is None) if not want_z else torch.equal(z2, z)
        np.testing.assert_allclose(rgba2.cpu().numpy(), rgba.cpu().numpy(), atol=1e-6, rtol=0)
    # the records it leaves behind are the ones the shading backward would build itself
    g = torch.randn(2, h, w, 4, generator=torch.Generator().manual_seed(1)).to(device) / (h * w)
    args = (g, ids, bary, clip, d["normals"], d["vertices"], diffuse, d["triangles"], lp, li, amb)
    plain = _native.shade_backward(*args)
    for a, b in zip(plain, _native.shade_backward(*args, corner_records=records)):
        if a is not None:
            np.testing.assert_allclose(b.cpu().numpy(), a.cpu().numpy(), atol=1e-7, rtol=1e-5)
    # with the transforms, dpositions also carries the clip-space gradient pulled back through them
    adjacency = _native.vertex_adjacency(d["triangles"], d["vertices"].shape[1])
    gathered = _native.shade_backward(*args, adjacency=adjacency)
    pulled = _native.shade_backward(*args, adjacency=adjacency, transforms=xf)
    want = gathered[2] + torch.matmul(gathered[0], xf)[..., :3]
    np.testing.assert_allclose(pulled[2].cpu().numpy(), want.cpu().numpy(), atol=1e-7, rtol=1e-5)
    for k in (0, 1, 3):
        np.testing.assert_allclose(pulled[k].cpu().numpy(), gathered[k].cpu().numpy(), atol=1e-8, rtol=1e-5)
    # without the light / ambient gradients (none of them requires grad): the same vertex-side outputs
    lean = _native.shade_backward(*args, adjacency=adjacency, want_light_grads=False)
    assert lean[4] is None and lean[5] is None and lean[6] is None
    for k in range(4):
        np.testing.assert_allclose(lean[k].cpu().numpy(), gathered[k].cpu().numpy(), atol=1e-8, rtol=1e-5)
    signs_loss, signs = _native.l1_loss_forward(rgba, torch.zeros_like(rgba))
    up = torch.ones(1, device=rgba.device)
    l1_args = (up,) + args[1:]
    with_lights = _native.shade_backward(*l1_args, adjacency=adjacency, l1_signs=signs)
    lean_l1 = _native.shade_backward(*l1_args, adjacency=adjacency, l1_signs=signs, want_light_grads=False)
    assert float(with_lights[4].abs().max()) > 0 and lean_l1[4] is None
    for k in range(4):
        np.testing.assert_allclose(lean_l1[k].cpu().numpy(), with_lights[k].cpu().numpy(), atol=1e-8, rtol=1e-5)


@pytest.mark.parametrize("w,h,res,n_lights,ambient", [(96, 80, 12, 1, False), (130, 67, 10, 3, True),
                                                        (64, 64, 120, 2, True), (33, 31, 6, 4, False),
                                                        (200, 150, 50, 1, False), (80, 56, 10, 6, True)])
def test_shade_backward_lane_kernel_matches_rows_kernel(device, w, h, res, n_lights, ambient):
    """Round 3: without light gradients mr_shade_backward runs k_accumulate_lanes -- the 18 / 27 / 36
    products the caller wants (autograd's needs_input_grad: d normals and / or d diffuse may be left
    out) stay in registers down each lane's vertical run -- instead of the rows kernel's 36 sums per
    row.  Same outputs as the rows kernel (forced through the debug hook) for both upstream forms; the
    outputs left out are None; and the same again with the G-buffer declared normalised.
    The 120-subdivision sphere at 64x64 has thousands of one-pixel runs per strip (merge-table
    overflow), (200, 150) spans several strips and a ragged last column block."""
    from pytorch_mesh_renderer_amd import _native
    job = synthetic.sphere_job(2, w, h, res)
    d = {k: (v.to(device) if torch.is_tensor(v) else v) for k, v in job.items()}
    gen = torch.Generator().manual_seed(5)
    diffuse = torch.rand(d["vertices"].shape, generator=gen).to(device)
    lp = (torch.rand(2, n_lights, 3, generator=gen) * 6 - 3).to(device)
    li = (torch.rand(2, n_lights, 3, generator=gen) + 0.2).to(device)
    amb = (torch.rand(2, 3, generator=gen) * 0.3).to(device) if ambient else None
    xf = synthetic.clip_transforms(job["eyes"], w, h).to(device)
    clip, ids, bary, _, rgba, records = _native.render_forward(
        d["vertices"], xf, d["normals"], diffuse, d["triangles"], lp, li, amb, w, h, want_z=False)
    adjacency = _native.vertex_adjacency(d["triangles"], d["vertices"].shape[1])
    g = torch.randn(2, h, w, 4, generator=torch.Generator().manual_seed(1)).to(device) / (h * w)
    _, signs = _native.l1_loss_forward(rgba, torch.zeros_like(rgba))
    up = torch.full((1,), 0.7, device=rgba.device)
    tail = (ids, bary, clip, d["normals"], d["vertices"], diffuse, d["triangles"], lp, li, amb)
    kw = dict(corner_records=records, adjacency=adjacency, transforms=xf, want_light_grads=False)
    try:
        for upstream, extra in ((g, {}), (up, {"l1_signs": signs})):
            _native.debug_set_shade_backward_kernel(1)
            full = _native.shade_backward(upstream, *tail, **kw, **extra)
            assert all(float(full[k].abs().max()) > 0 for k in range(4))
            for want_n, want_d in ((False, False), (True, False), (True, True)):
                # which = 3: the lane kernel told that the G-buffer is the rasterizer's own
                # (MR_GBUFFER_NORMALISED: alpha = 1 exactly, its terms are left out) -- same outputs
                for which in (1, 2, 3):
                    _native.debug_set_shade_backward_kernel(min(which, 2))
                    lean = _native.shade_backward(upstream, *tail, **kw, **extra, want_normal_grads=want_n,
                                                  want_diffuse_grads=want_d, normalised_gbuffer=which == 3)
                    assert (lean[3] is None) == (not want_d) and (lean[1] is None) == (not want_n)
                    for k in (0, 1, 2, 3):
                        if lean[k] is None:
                            continue
                        scale = float(full[k].abs().max())
                        np.testing.assert_allclose(lean[k].cpu().numpy(), full[k].cpu().numpy(), rtol=2e-4,
                                                   atol=2e-6 * scale, err_msg="output %d kernel %d" % (k, which))
            # round 4: the clip-space gradient not wanted on its own (render() differentiated to the vertices
            # only): the pull-back through the transforms is folded into the pixel pass -- 9 sums per triangle
            # (ShadeLaneFn<..., FOLD>) -- and where that variant does not exist (rows kernel forced, G-buffer
            # not declared normalised, normals wanted) the clip gradient goes to scratch: dclip is None and
            # d positions is the same whole-vertex gradient every time
            scale = float(full[2].abs().max())
            for which, normalised, want_n in ((2, True, False), (1, True, False), (2, False, False), (2, True, True)):
                _native.debug_set_shade_backward_kernel(which)
                folded = _native.shade_backward(upstream, *tail, **kw, **extra, want_normal_grads=want_n,
                                                want_diffuse_grads=False, normalised_gbuffer=normalised,
                                                want_clip_grads=False)
                assert folded[0] is None and folded[3] is None
                np.testing.assert_allclose(folded[2].cpu().numpy(), full[2].cpu().numpy(), rtol=2e-4, atol=2e-6 * scale,
                                           err_msg="d positions without d clip, kernel %d normalised %s normals %s" % (
                                               which, normalised, want_n))
            if n_lights <= 2:
                # second half of round 3: with one or two lights the lane kernel also carries the light
                # gradients (6 L + 3 per-lane sums, one row per strip) -- against the rows kernel
                kw_l = dict(kw, want_light_grads=True)
                _native.debug_set_shade_backward_kernel(1)
                rows = _native.shade_backward(upstream, *tail, **kw_l, **extra)
                _native.debug_set_shade_backward_kernel(2)
                lanes = _native.shade_backward(upstream, *tail, **kw_l, **extra)
                for k, (a, b) in enumerate(zip(lanes, rows)):
                    if b is None:
                        assert a is None
                        continue
                    scale = float(b.abs().max())
                    assert scale > 0 or k >= 4, k
                    np.testing.assert_allclose(a.cpu().numpy(), b.cpu().numpy(), rtol=2e-4, atol=2e-6 * scale + 1e-12,
                                               err_msg="light-gradient variant, output %d" % k)
    finally:
        _native.debug_set_shade_backward_kernel(0)


def test_prepared_backward_block_serves_repeated_backward_calls(device):
    """Round 4: render() differentiated to the vertices only has its forward's setup kernel write the folded
    backward's records and clear its accumulator rows (mr_render_forward's backward_prepared); the backward then
    launches no setup kernel and its per-vertex gather leaves the rows clear again.  Same gradients as with the
    backward's own setup (PREPARE_BACKWARD = False), through both routes (losses.l1_loss's sign-coded one and a
    dense upstream gradient), and the SAME gradient again from a second backward over a retained graph."""
    from pytorch_mesh_renderer_amd import _native
    from pytorch_mesh_renderer_amd.mesh_renderer import rasterize_triangles_ext as ext
    job = synthetic.sphere_job(2, 200, 150, 12)
    target = torch.rand(2, 150, 200, 4, generator=torch.Generator().manual_seed(3)).to(device)
    tris = job["triangles"].to(device)

    def grads(prepare, loss_fn, repeats):
        before = ext.PREPARE_BACKWARD
        ext.PREPARE_BACKWARD = prepare
        try:
            v = job["vertices"].clone().to(device).requires_grad_(True)
            with _CountCalls("render_forward") as fwd:
                img = mesh_renderer.render(v, tris, job["normals"].to(device), job["diffuse"].to(device), job["eyes"],
                                           torch.zeros(2, 3), torch.tensor([0.0, 1.0, 0.0]),
                                           job["light_positions"].to(device), job["light_intensities"].to(device),
                                           200, 150)
            assert fwd.calls == 1
            loss = loss_fn(img)
            out = []
            for k in range(repeats):
                v.grad = None
                loss.backward(retain_graph=k + 1 < repeats)
                out.append(v.grad.clone())
            return out
        finally:
            ext.PREPARE_BACKWARD = before

    for loss_fn in (lambda img: mesh_renderer.losses.l1_loss(img, target) * 50.0,
                    lambda img: ((img - target) ** 2).mean() * 50.0):
        own = grads(False, loss_fn, 1)[0]
        first, second, third = grads(True, loss_fn, 3)
        scale = float(own.abs().max())
        assert scale > 0
        for name, got in (("first", first), ("second", second), ("third", third)):
            np.testing.assert_allclose(got.cpu().numpy(), own.cpu().numpy(), rtol=2e-4, atol=2e-6 * scale,
                                       err_msg="%s backward over the prepared block" % name)
    # the block is only asked for when the vertices alone require grad
    v = job["vertices"].clone().to(device).requires_grad_(True)
    n = job["normals"].clone().to(device).requires_grad_(True)
    seen = {}
    orig = _native.render_forward
    def spy(*a, **k):
        seen["prepare"] = k.get("prepare_backward")
        return orig(*a, **k)
    _native.render_forward = spy
    try:
        mesh_renderer.render(v, tris, n, job["diffuse"].to(device), job["eyes"], torch.zeros(2, 3),
                             torch.tensor([0.0, 1.0, 0.0]), job["light_positions"].to(device),
                             job["light_intensities"].to(device), 200, 150)
        assert seen["prepare"] is False
        mesh_renderer.render(v, tris, n.detach(), job["diffuse"].to(device), job["eyes"], torch.zeros(2, 3),
                             torch.tensor([0.0, 1.0, 0.0]), job["light_positions"].to(device),
                             job["light_intensities"].to(device), 200, 150)
        assert seen["prepare"] is True
    finally:
        _native.render_forward = orig


def test_shade_backward_gather_matches_scatter(device):
    """mr_shade_backward with the CSR vertex adjacency (per-vertex gather, what render() uses) vs
    without it (float-atomic scatter), incl. a triangle with a repeated and an out-of-range vertex."""
    from pytorch_mesh_renderer_amd import _native
    job = synthetic.sphere_job(2, 120, 90, 10)
    d = {k: (v.to(device) if torch.is_tensor(v) else v) for k, v in job.items()}
    tris = d["triangles"].clone()
    tris[3, 1] = tris[3, 0]                       # repeated vertex
    tris[5, 2] = d["vertices"].shape[1] + 7       # out of range: ignored by both
    ids, bary, _ = _native.rasterize_forward(d["clip"], tris, 120, 90)
    args = (ids, bary, d["clip"], d["normals"], d["vertices"], d["diffuse"], tris, d["light_positions"],
            d["light_intensities"], None)
    g = torch.randn(2, 90, 120, 4, generator=torch.Generator().manual_seed(1)).to(device) / (90 * 120)
    scatter = _native.shade_backward(g, *args)
    gather = _native.shade_backward(g, *args, adjacency=_native.vertex_adjacency(tris, d["vertices"].shape[1]))
    for name, a, b in zip(("dclip", "dnormals", "dpositions", "ddiffuse", "dlpos", "dlint"), scatter, gather):
        assert float(a.abs().max()) > 0, name
        np.testing.assert_allclose(b.cpu().numpy(), a.cpu().numpy(), atol=1e-7, rtol=1e-5, err_msg=name)


def test_step_is_capturable_into_a_hip_graph(device):
    """render() + l1_loss + backward captured with torch.cuda.CUDAGraph (hipGraph) and replayed after an
    in-place vertex update gives the eager result: the C ABI neither allocates nor synchronises."""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import graph_bench
    vertices, step = graph_bench.build(2, 96, 8, device)
    graph, loss = graph_bench.capture(vertices, step)
    with torch.no_grad():
        vertices.mul_(1.03)                       # new input values, same storage
    graph.replay()
    torch.cuda.synchronize()
    got_loss, got_grad = float(loss), vertices.grad.clone()
    vertices.grad = None
    want_loss = float(step().detach())
    assert abs(got_loss - want_loss) < 1e-7
    np.testing.assert_allclose(got_grad.cpu().numpy(), vertices.grad.cpu().numpy(), atol=1e-9, rtol=1e-5)
    assert float(got_grad.abs().max()) > 0


def test_capture_step_replays_the_references_loop(device):
    """mesh_renderer.capture_step (round 5): the step as the reference's optimisation tests write it
    (mesh_renderer_test.py:238-262: render, mean(abs(image - target)), backward, an optimizer update in place) captured
    once and replayed: every replay equals the eager step on the same parameter values."""
    job = synthetic.sphere_job(2, 96, 96, 8)
    d = {k: (v.to(device) if torch.is_tensor(v) else v) for k, v in job.items()}
    vertices = d["vertices"].clone().requires_grad_(True)
    center, up = torch.zeros_like(d["eyes"]), torch.tensor([0.0, 1.0, 0.0], device=device)

    def render():
        return mesh_renderer.render(vertices, d["triangles"], d["normals"], d["diffuse"], d["eyes"], center, up,
                                    d["light_positions"], d["light_intensities"], 96, 96)
    with torch.no_grad():
        target = render().roll(4, 2).contiguous()

    def step():
        loss = torch.mean(torch.abs(render() - target))     # the reference's spelling
        loss.backward()
        return loss
    captured = mesh_renderer.capture_step(step, [vertices])
    assert isinstance(captured, mesh_renderer.CapturedStep)
    optimizer = torch.optim.SGD([vertices], lr=0.5)
    losses = []
    for it in range(4):
        loss = captured.replay()
        torch.cuda.synchronize()
        got_loss, got_grad = float(loss), vertices.grad.clone()
        # the same step eagerly, on the same values
        keep = vertices.grad
        vertices.grad = None
        want_loss = float(step().detach())
        want_grad = vertices.grad
        vertices.grad = keep
        assert abs(got_loss - want_loss) < 1e-7
        np.testing.assert_allclose(got_grad.cpu().numpy(), want_grad.cpu().numpy(), atol=1e-9, rtol=1e-5, err_msg="replay %d" % it)
        losses.append(got_loss)
        optimizer.step()                                     # in place: the next replay reads the new vertices
    assert losses[-1] < losses[0], losses


def test_host_camera_memo_stays_out_of_graph_captures_and_orders_streams(device):
    """(1) the host-camera memo is neither read nor written while a stream is capturing -- a
    warm-up render() must not make a capture with host cameras succeed by baking the kept tensor's address
    into the graph (the next eager call with other cameras would free it under the replays); (2) a hit on
    another stream is ordered behind the upload of the call that stored the entry; (3) the key carries
    the resolved device index."""
    from pytorch_mesh_renderer_amd.common import camera_utils as cu
    eye = torch.tensor([[0.0, 0.0, 3.0], [1.0, 0.5, 3.0]])
    center, up = torch.zeros(2, 3), torch.tensor([[0.0, 1.0, 0.0]]).repeat(2, 1)
    fov, near, far = torch.tensor([40.0, 40.0]), torch.tensor([0.01, 0.01]), torch.tensor([10.0, 10.0])
    want = torch.matmul(cu.perspective(1.0, fov, near, far), cu.look_at(eye, center, up))
    before = cu.CACHE_HOST_CAMERAS
    cu.CACHE_HOST_CAMERAS = True
    try:
        kept = cu.clip_space_transforms(eye, center, up, fov, near, far, 1.0, "cuda")     # index-less device
        assert cu.clip_space_transforms(eye, center, up, fov, near, far, 1.0, device) is kept
        assert cu._host_cache.entry[2] == torch.device("cuda", torch.cuda.current_device())
        side = torch.cuda.Stream()
        with torch.cuda.stream(side):
            hit = cu.clip_space_transforms(eye, center, up, fov, near, far, 1.0, device)
            got = hit.clone()
        side.synchronize()
        assert hit is kept and torch.equal(got.cpu(), want)
        # a capture neither looks the memo up nor stores into it
        graph = torch.cuda.CUDAGraph()
        captured = None
        with torch.cuda.graph(graph):
            assert cu._host_camera_key(eye, center, up, fov, near, far, 1.0, device) is None
            captured = cu._host_cache.entry
        assert captured is cu._host_cache.entry and captured[3] is kept
        # other cameras eagerly: the entry is replaced, the earlier tensor object is untouched
        eye2 = eye + 0.25
        other = cu.clip_space_transforms(eye2, center, up, fov, near, far, 1.0, device)
        assert other is not kept and torch.equal(kept.cpu(), want)
    finally:
        cu.CACHE_HOST_CAMERAS = before


@pytest.mark.parametrize("n_attrs", [1, 4, 7, 9, 13, 16, 17])
def test_fused_rasterize_backward_matches_composed_ops(device, n_attrs):
    """rasterize(): the one-pass fused backward (<= 16 attributes) vs the composed
    BarycentricRasterizer + AttributeInterpolator ops, outputs and all gradients."""
    rast_mod = sys.modules["pytorch_mesh_renderer_amd.mesh_renderer.rasterize"]
    job = synthetic.sphere_job(2, 121, 87, 12)   # 121 * 87 is not a multiple of 64: ragged last wavefront
    gen = torch.Generator().manual_seed(n_attrs)
    proj = synthetic.clip_transforms(job["eyes"], 121, 87).to(device)
    target = torch.rand(2, 87, 121, n_attrs, generator=gen).to(device)
    results = {}
    for fused in (True, False):
        v = job["vertices"].clone().to(device).requires_grad_(True)
        a = torch.rand(2, v.shape[1], n_attrs, generator=torch.Generator().manual_seed(7)).to(device).requires_grad_(True)
        bg = torch.linspace(-1.0, 0.5, n_attrs).to(device).requires_grad_(True)
        rast_mod.USE_FUSED_BACKWARD = fused
        try:
            with _CountCalls("interpolate_raster_backward") as counter:
                out = mesh_renderer.rasterize(v, a, job["triangles"].to(device), proj, 121, 87, bg)
                torch.mean(torch.abs(out - target)).backward()
        finally:
            rast_mod.USE_FUSED_BACKWARD = True
        assert counter.calls == (1 if fused and n_attrs <= 16 else 0)
        results[fused] = [t.detach().cpu().numpy() for t in (out, v.grad, a.grad, bg.grad)]
    for name, got, want in zip(("out", "dvertices", "dattributes", "dbackground"), results[True], results[False]):
        assert np.abs(want).max() > 0, name
        np.testing.assert_allclose(got, want, atol=ATOL * 1e-2, rtol=1e-4, err_msg=name)
    if n_attrs <= 8:
        # round 3: up to 8 attributes the fused backward is the lane-accumulating kernel (AttrLaneFn);
        # the rows kernel, forced through the debug switch, gives the same gradients
        from pytorch_mesh_renderer_amd import _native
        before = _native.debug_set_shade_backward_kernel(1)
        try:
            v = job["vertices"].clone().to(device).requires_grad_(True)
            a = torch.rand(2, v.shape[1], n_attrs, generator=torch.Generator().manual_seed(7)).to(device).requires_grad_(True)
            bg = torch.linspace(-1.0, 0.5, n_attrs).to(device).requires_grad_(True)
            out = mesh_renderer.rasterize(v, a, job["triangles"].to(device), proj, 121, 87, bg)
            torch.mean(torch.abs(out - target)).backward()
        finally:
            _native.debug_set_shade_backward_kernel(before)
        for name, got, want in zip(("dvertices", "dattributes"), results[True][1:3], (v.grad, a.grad)):
            np.testing.assert_allclose(got, want.cpu().numpy(), atol=ATOL * 1e-2, rtol=1e-4, err_msg="rows kernel " + name)


@pytest.mark.parametrize("w,h", [(1, 1), (7, 3), (65, 2), (63, 65)])
def test_fused_rasterize_tiny_and_ragged_images(device, w, h):
    """Fused rasterize() forward / backward on images smaller than a wavefront or a region."""
    rast_mod = sys.modules["pytorch_mesh_renderer_amd.mesh_renderer.rasterize"]
    job = synthetic.sphere_job(3, w, h, 6)
    proj = synthetic.clip_transforms(job["eyes"], max(w, 2), max(h, 2)).to(device)
    results = {}
    for fused in (True, False):
        v = job["vertices"].clone().to(device).requires_grad_(True)
        a = torch.rand(3, v.shape[1], 5, generator=torch.Generator().manual_seed(1)).to(device).requires_grad_(True)
        rast_mod.USE_FUSED_BACKWARD = fused
        try:
            out = mesh_renderer.rasterize(v, a, job["triangles"].to(device), proj, w, h, torch.full((5,), 0.25))
            (out * out).mean().backward()
        finally:
            rast_mod.USE_FUSED_BACKWARD = True
        assert out.shape == (3, h, w, 5)
        results[fused] = [t.detach().cpu().numpy() for t in (out, v.grad, a.grad)]
    for name, got, want in zip(("out", "dvertices", "dattributes"), results[True], results[False]):
        np.testing.assert_allclose(got, want, atol=1e-6, rtol=1e-4, err_msg=name)


@pytest.mark.parametrize("ambient,target_grad", [(False, False), (True, True)])
def test_fused_render_l1_loss_matches_generic_loss(device, ambient, target_grad):
    """l1_loss on render()'s direct output takes FusedPhongL1Loss (sign codes -> shading backward, no
    dense gradient image); value and every gradient must equal the generic op's, also when the image
    feeds a second consumer and when the target needs a gradient."""
    losses = sys.modules["pytorch_mesh_renderer_amd.mesh_renderer.losses"]
    job = synthetic.sphere_job(3, 130, 70, 14)
    gen = torch.Generator().manual_seed(11)
    target0 = torch.rand(3, 70, 130, 4, generator=gen)
    results = {}
    for fused in (True, False):
        leaves = {k: job[k].clone().to(device).requires_grad_(True) for k in ("vertices", "normals", "diffuse")}
        leaves["lpos"] = job["light_positions"].clone().to(device).requires_grad_(True)
        leaves["lint"] = (job["light_intensities"] * 0.8).to(device).requires_grad_(True)
        if ambient:
            leaves["amb"] = torch.tensor([[0.1, 0.2, 0.05]] * 3, device=device, requires_grad=True)
        target = target0.clone().to(device).requires_grad_(target_grad)
        losses.USE_FUSED_RENDER_LOSS = fused
        try:
            with _CountCalls("l1_loss_backward") as dense:
                img = mesh_renderer.render(leaves["vertices"], job["triangles"].to(device), leaves["normals"],
                                           leaves["diffuse"], job["eyes"], torch.zeros(3, 3),
                                           torch.tensor([0.0, 1.0, 0.0]), leaves["lpos"], leaves["lint"], 130, 70,
                                           ambient_color=leaves.get("amb"))
                loss = mesh_renderer.losses.l1_loss(img, target)
                total = 3.0 * loss + (0.25 * img[..., 1].mean() if ambient else 0.0)   # a second consumer
                total.backward()
        finally:
            losses.USE_FUSED_RENDER_LOSS = True
        # the dense gradient image is only formed by the generic op (or for the target's gradient)
        assert dense.calls == (1 if (not fused or target_grad) else 0)
        grads = {k: v.grad.cpu().numpy() for k, v in leaves.items()}
        if target_grad:
            grads["target"] = target.grad.cpu().numpy()
        results[fused] = (float(loss), grads)
    assert abs(results[True][0] - results[False][0]) < 1e-7
    for k, want in results[False][1].items():
        got = results[True][1][k]
        assert np.abs(want).max() > 0, k
        np.testing.assert_allclose(got, want, atol=1e-9, rtol=2e-4, err_msg=k)


@pytest.mark.gpu
@pytest.mark.parametrize("h,w", [(256, 320), (130, 200), (64, 64), (70, 50), (1, 3)])
def test_l1_loss_over_empty_block_maps_matches_the_flat_loss(device, h, w):
    """Round 4: mr_image_empty_regions / mr_l1_loss_forward_regions.  A 64 x 64 block (counted from the image's last
    row up, the G-buffer's row order) that is whole and all zeros (-0.0 included) on both sides is not read: same
    loss (another fixed summation order: 1e-6 relative), byte-identical sign codes, ragged edges never skipped, and
    a NaN in the target keeps its block in play."""
    from pytorch_mesh_renderer_amd import _native
    gen = torch.Generator().manual_seed(h * 1000 + w)
    a = torch.rand(2, h, w, 4, generator=gen) - 0.5
    b = torch.rand(2, h, w, 4, generator=gen) - 0.5
    by, bx = (h + 63) // 64, (w + 63) // 64
    want_a = np.zeros((2, by, bx), np.uint8)
    want_b = np.zeros((2, by, bx), np.uint8)
    pick = np.random.RandomState(5)
    for img in range(2):
        for j in range(by):
            for i in range(bx):
                if (i + 1) * 64 > w or (j + 1) * 64 > h:
                    continue
                rows = slice(h - 64 * (j + 1), h - 64 * j)
                kind = pick.randint(4)
                if kind in (0, 1):
                    a[img, rows, 64 * i:64 * i + 64] = 0.0
                    want_a[img, j, i] = 1
                if kind in (0, 2):
                    b[img, rows, 64 * i:64 * i + 64] = -0.0 if (i + j) % 2 else 0.0
                    want_b[img, j, i] = 1
    a, b = a.to(device), b.to(device)
    map_a, map_b = _native.image_empty_regions(a), _native.image_empty_regions(b)
    assert np.array_equal(map_a.cpu().numpy(), want_a) and np.array_equal(map_b.cpu().numpy(), want_b)
    flat, flat_signs = _native.l1_loss_forward(a, b)
    got, got_signs = _native.l1_loss_forward(a, b, empty_a=map_a, empty_b=map_b)
    assert abs(float(got) - float(flat)) <= 1e-6 * abs(float(flat))
    assert abs(float(got) - float((a - b).abs().mean())) <= 1e-6 * abs(float(flat))
    assert torch.equal(got_signs, flat_signs)
    no_signs, none = _native.l1_loss_forward(a, b, want_signs=False, empty_a=map_a, empty_b=map_b)
    assert none is None and float(no_signs) == float(got)
    if h >= 64 and w >= 64:
        # a NaN target pixel inside a block that is zero in the image: the target's map keeps the block
        a[0, h - 64:h, 0:64] = 0.0
        b[0, h - 64:h, 0:64] = 0.0
        b[0, h - 10, 7, 2] = float("nan")
        map_a, map_b = _native.image_empty_regions(a), _native.image_empty_regions(b)
        assert int(map_a[0, 0, 0]) == 1 and int(map_b[0, 0, 0]) == 0
        assert math.isnan(float(_native.l1_loss_forward(a, b, empty_a=map_a, empty_b=map_b)[0]))


@pytest.mark.gpu
@pytest.mark.parametrize("w,h,batch", [(768, 768, 8), (448, 320, 2), (200, 150, 2)])
def test_render_empty_block_map_serves_the_loss_and_the_backward(device, w, h, batch):
    """Round 4: mr_render_forward's empty_regions map (1 = the 64 x 64 block had no candidate triangle).  Every
    marked block is transparent black in the image (the map is conservative, so it is a subset of the image's own
    map); with it the fused loss and the shading backward skip those blocks: the same loss to 1e-6 and the same
    gradients as with ext.EMPTY_REGIONS = False."""
    from pytorch_mesh_renderer_amd import _native
    from pytorch_mesh_renderer_amd.mesh_renderer import rasterize_triangles_ext as ext
    job = synthetic.sphere_job(batch, w, h, 12)
    tris = job["triangles"].to(device)
    target = torch.zeros(batch, h, w, 4)
    target[:, h // 3:h // 2, w // 4:w // 2] = torch.rand(batch, h // 2 - h // 3, w // 2 - w // 4, 4,
                                                        generator=torch.Generator().manual_seed(2))
    target = mesh_renderer.losses.remember_target(target.to(device))   # (round 6: the target's map is opt-in)
    out = _native.render_forward(job["vertices"].to(device), synthetic.clip_transforms(job["eyes"], w, h).to(device),
                                 job["normals"].to(device), job["diffuse"].to(device), tris,
                                 job["light_positions"].to(device), job["light_intensities"].to(device), None, w, h,
                                 want_empty_regions=True)
    rgba, marked = out[4], out[-1]
    assert marked.shape == (batch, (h + 63) // 64, (w + 63) // 64)
    own = _native.image_empty_regions(rgba)
    assert bool(((marked == 0) | (own == 1)).all()), "a block marked empty holds pixels"
    if batch == 8:   # (64-pixel regions: smaller jobs run 32-pixel regions and mark nothing)
        assert int(marked.sum()) > 0, "the scene leaves whole blocks empty"

    def run(use_map, loss_kind):
        before = ext.EMPTY_REGIONS
        ext.EMPTY_REGIONS = use_map
        try:
            v = job["vertices"].clone().to(device).requires_grad_(True)
            d = job["diffuse"].clone().to(device).requires_grad_(loss_kind == "l1-all")
            img = mesh_renderer.render(v, tris, job["normals"].to(device), d, job["eyes"], torch.zeros(batch, 3),
                                       torch.tensor([0.0, 1.0, 0.0]), job["light_positions"].to(device),
                                       job["light_intensities"].to(device), w, h)
            loss = (mesh_renderer.losses.l1_loss(img, target) if loss_kind.startswith("l1")
                    else ((img - target) ** 2).mean())
            loss.backward()
            return float(loss), v.grad.clone(), (d.grad.clone() if d.grad is not None else None)
        finally:
            ext.EMPTY_REGIONS = before

    for kind in ("l1", "l1-all", "mse"):
        l0, gv0, gd0 = run(False, kind)
        l1, gv1, gd1 = run(True, kind)
        assert abs(l1 - l0) <= 1e-6 * abs(l0), kind
        scale = float(gv0.abs().max())
        assert scale > 0
        # (the skipped strips add nothing; the per-triangle sums are float atomics, so not bit for bit)
        np.testing.assert_allclose(gv1.cpu().numpy(), gv0.cpu().numpy(), rtol=1e-4, atol=1e-6 * scale, err_msg=kind)
        assert (gd0 is None) == (gd1 is None)
        if gd0 is not None:
            np.testing.assert_allclose(gd1.cpu().numpy(), gd0.cpu().numpy(), rtol=1e-4,
                                       atol=1e-6 * float(gd0.abs().max()), err_msg=kind)


def test_target_empty_block_map_is_opt_in_and_dies_with_an_in_place_edit(device):
    """Round 6: the target's empty-block map exists only for a target named with losses.remember_target, rides on the
    tensor object and is used while data pointer, shape and version counter are unchanged: an in-place torch edit
    voids it (the loss reads every block again and is right), remember_target after the edit makes a new one,
    forget_target drops it; an un-named target never gets one."""
    from pytorch_mesh_renderer_amd.mesh_renderer import rasterize_triangles_ext as ext
    job = synthetic.sphere_job(8, 768, 768, 12)
    tris = job["triangles"].to(device)
    target = torch.zeros(8, 768, 768, 4, device=device)

    def loss_of():
        v = job["vertices"].clone().to(device).requires_grad_(True)
        img = mesh_renderer.render(v, tris, job["normals"].to(device), job["diffuse"].to(device), job["eyes"],
                                   torch.zeros(8, 3), torch.tensor([0.0, 1.0, 0.0]), job["light_positions"].to(device),
                                   job["light_intensities"].to(device), 768, 768)
        loss = mesh_renderer.losses.l1_loss(img, target)
        return float(loss), float((img.detach() - target).abs().mean()), loss.grad_fn

    got, want, _ = loss_of()
    assert abs(got - want) <= 1e-6 * want
    assert not hasattr(target, "_mr_empty_regions") and ext._target_empty_regions(target) is None   # not named: no map
    assert mesh_renderer.losses.remember_target(target) is target
    first = ext._target_empty_regions(target)
    assert first is not None and int(first.sum()) == first.numel()   # an all-zero target: every block empty
    got, want, _ = loss_of()
    assert abs(got - want) <= 1e-6 * want
    assert ext._target_empty_regions(target) is first                # the same map, not a new one per use
    target[3, 710:760, 5:60] = 0.75            # inside ONE corner block the sphere does not reach (G-buffer rows 7..57)
    assert ext._target_empty_regions(target) is None, "an in-place edit voids the map"
    got, want, _ = loss_of()
    assert abs(got - want) <= 1e-6 * want, "the edit was not seen"
    mesh_renderer.losses.remember_target(target)
    again = ext._target_empty_regions(target)
    assert int(again.sum()) == first.numel() - 1
    got, want, _ = loss_of()
    assert abs(got - want) <= 1e-6 * want
    mesh_renderer.losses.forget_target(target)
    assert ext._target_empty_regions(target) is None and not hasattr(target, "_mr_empty_regions")
    with pytest.raises(ValueError):
        mesh_renderer.losses.remember_target(torch.zeros(4, 4, device=device))


def test_l1_loss_drops_the_renderers_map_when_the_image_was_edited(device):
    """an in-place edit of render()'s output under no_grad keeps its grad_fn, so the fused loss route still
    applies -- but the renderer's empty-block map no longer describes the image.  The record carries the image's version
    counter; after an edit the loss reads every pixel and equals torch.mean(torch.abs())."""
    job = synthetic.sphere_job(4, 512, 512, 12)
    tris = job["triangles"].to(device)
    target = torch.zeros(4, 512, 512, 4, device=device)
    v = job["vertices"].clone().to(device).requires_grad_(True)
    img = mesh_renderer.render(v, tris, job["normals"].to(device), job["diffuse"].to(device), job["eyes"],
                               torch.zeros(4, 3), torch.tensor([0.0, 1.0, 0.0]), job["light_positions"].to(device),
                               job["light_intensities"].to(device), 512, 512)
    node = img.grad_fn
    with torch.no_grad():
        img[..., 3] = 1.0          # fills every block that was empty (corner blocks of the sphere's frame)
    assert img.grad_fn is node
    with _CountCalls("_shade_backward_call") as bwd:
        loss = mesh_renderer.losses.l1_loss(img, target)
        want = torch.mean(torch.abs(img.detach() - target))
        assert abs(float(loss) - float(want)) <= 1e-6 * float(want)
        loss.backward()
    assert bwd.calls == 1 and v.grad is not None and float(v.grad.abs().max()) > 0


def test_l1_loss_on_a_derived_image_takes_the_generic_path(device):
    """Only render()'s own output carries the fused-loss record; a slice or a scaled copy does not."""
    job = synthetic.sphere_job(1, 64, 48, 8)
    v = job["vertices"].clone().to(device).requires_grad_(True)
    img = mesh_renderer.render(v, job["triangles"].to(device), job["normals"].to(device), job["diffuse"].to(device),
                               job["eyes"], torch.zeros(1, 3), torch.tensor([0.0, 1.0, 0.0]),
                               job["light_positions"].to(device), job["light_intensities"].to(device), 64, 48)
    from pytorch_mesh_renderer_amd.mesh_renderer import rasterize_triangles_ext as ext
    assert img.grad_fn in ext._fused_renders and "_mr_fused_render" not in img.__dict__
    derived = img * 1.0
    assert ext.take_fused_render(derived) is None
    with _CountCalls("l1_loss_backward") as dense:
        mesh_renderer.losses.l1_loss(derived, torch.zeros_like(derived)).backward()
    assert dense.calls == 1 and float(v.grad.abs().max()) > 0


def test_fused_render_loss_keeps_autograd_semantics_of_the_image(device):
    """the fused render-loss route differentiates from image.detach() straight to the
    renderer's inputs, so it must step aside whenever the image's own gradient is observed
    (retain_grad, a tensor hook), must not leave Python state on the tensor (torch.save / deepcopy
    work), is taken once per rendered image, and its record dies with the autograd node."""
    import copy
    import gc
    import io
    from pytorch_mesh_renderer_amd.mesh_renderer import rasterize_triangles_ext as ext
    job = synthetic.sphere_job(1, 64, 48, 8)
    tri, normals, diffuse = job["triangles"].to(device), job["normals"].to(device), job["diffuse"].to(device)
    target = torch.rand(1, 48, 64, 4, generator=torch.Generator().manual_seed(4)).to(device)

    def render(v):
        return mesh_renderer.render(v, tri, normals, diffuse, job["eyes"], torch.zeros(1, 3),
                                    torch.tensor([0.0, 1.0, 0.0]), job["light_positions"].to(device),
                                    job["light_intensities"].to(device), 64, 48)

    def leaf():
        return job["vertices"].clone().to(device).requires_grad_(True)

    # reference gradient: the fused route
    v0 = leaf()
    with _CountCalls("l1_loss_backward") as dense:
        mesh_renderer.losses.l1_loss(render(v0), target).backward()
    assert dense.calls == 0
    # retain_grad: the image receives its gradient, the vertices the same one as before
    v1 = leaf()
    img = render(v1)
    img.retain_grad()
    with _CountCalls("l1_loss_backward") as dense:
        mesh_renderer.losses.l1_loss(img, target).backward()
    assert dense.calls == 1 and img.grad is not None and float(img.grad.abs().max()) > 0
    np.testing.assert_allclose(v1.grad.cpu().numpy(), v0.grad.cpu().numpy(), rtol=2e-4, atol=1e-9)
    # a hook on the image fires
    v2 = leaf()
    img = render(v2)
    seen = []
    img.register_hook(lambda g: seen.append(float(g.abs().sum())))
    mesh_renderer.losses.l1_loss(img, target).backward()
    assert len(seen) == 1 and seen[0] > 0
    np.testing.assert_allclose(v2.grad.cpu().numpy(), v0.grad.cpu().numpy(), rtol=2e-4, atol=1e-9)
    # Round 5: the decision is taken when the BACKWARD runs (until round 4: when l1_loss() was called, and a hook put on
    # the image afterwards never fired -- ADVICE r3's documented hole).  A hook / retain_grad registered AFTER the loss
    # was built is honoured: the node then behaves like the generic op.
    v2b = leaf()
    img = render(v2b)
    loss = mesh_renderer.losses.l1_loss(img, target)
    late = []
    img.register_hook(lambda g: late.append(1))
    img.retain_grad()
    with _CountCalls("l1_loss_backward") as dense:
        loss.backward()
    assert late == [1] and img.grad is not None and dense.calls == 1
    np.testing.assert_allclose(v2b.grad.cpu().numpy(), v0.grad.cpu().numpy(), rtol=2e-4, atol=1e-9)
    before = mesh_renderer.losses.USE_FUSED_RENDER_LOSS
    mesh_renderer.losses.USE_FUSED_RENDER_LOSS = False
    try:
        v2c = leaf()
        img = render(v2c)
        loss = mesh_renderer.losses.l1_loss(img, target)
        img.register_hook(lambda g: late.append(2))
        loss.backward()
        assert late == [1, 2]
    finally:
        mesh_renderer.losses.USE_FUSED_RENDER_LOSS = before
    # torch.autograd.grad w.r.t. the image (no retain_grad needed since round 5: the call names a RenderedImage)
    v3 = leaf()
    img = render(v3)
    (dimg,) = torch.autograd.grad(mesh_renderer.losses.l1_loss(img, target), img)
    assert dimg.shape == img.shape and type(dimg) is torch.Tensor
    want = torch.sign(img.detach() - target) / img.numel()
    assert torch.equal(dimg, want)
    # ... together with the vertices: both gradients, neither counted twice
    v3b = leaf()
    img = render(v3b)
    dimg, dv = torch.autograd.grad(mesh_renderer.losses.l1_loss(img, target), [img, v3b])
    assert torch.equal(dimg, want)
    np.testing.assert_allclose(dv.cpu().numpy(), v0.grad.cpu().numpy(), rtol=2e-4, atol=1e-9)
    # nothing on the tensor: it pickles and deep-copies; two losses on one image add up
    v4 = leaf()
    img = render(v4)
    assert not [k for k in img.__dict__ if k.startswith("_mr_fused")]
    buf = io.BytesIO()
    torch.save(img.detach(), buf)
    torch.save(img, buf)
    assert torch.equal(copy.deepcopy(img.detach()), img.detach())
    (mesh_renderer.losses.l1_loss(img, target) + mesh_renderer.losses.l1_loss(img, target)).backward()
    np.testing.assert_allclose(v4.grad.cpu().numpy(), 2.0 * v0.grad.cpu().numpy(), rtol=2e-4, atol=1e-9)
    # the record lives exactly as long as the renderer's node
    del img
    gc.collect()
    n_before = len(ext._fused_renders)
    img = render(leaf())
    assert len(ext._fused_renders) == n_before + 1
    del img
    gc.collect()
    assert len(ext._fused_renders) == n_before


def test_camera_transform_kernel_matches_reference_golden_and_torch_autograd(device):
    """Round 3: perspective . look_at on device-resident cameras as ONE launch each way
    (mr_camera_transforms[_backward]) instead of ~30 tiny torch kernels: the reference's captured
    look_at / perspective matrices, torch autograd of the same expressions for the gradients w.r.t. eye,
    center and up, the reference's two degeneracy assertions, and render() end to end."""
    from pytorch_mesh_renderer_amd import _native
    g = golden_npz("camera_utils.npz")
    eyes = torch.tensor(g["eyes"], device=device)
    n = eyes.shape[0]
    zeros, up = torch.zeros(n, 3, device=device), torch.tensor([[0.0, 1.0, 0.0]], device=device).repeat(n, 1)
    full = lambda v: torch.full((n,), float(v), device=device)
    got, flags = _native.camera_transforms(eyes, zeros, up, full(40.0), full(0.01), full(10.0), 1.25)
    want = np.matmul(g["perspective"][0], g["look_at"])           # perspective(1.25, 40, 0.01, 10) . look_at
    np.testing.assert_allclose(got.cpu().numpy(), want, atol=2e-6, rtol=1e-6)
    assert int(flags.item()) == 0
    # gradients: the kernel vs torch autograd through camera_utils' own expressions
    gen = torch.Generator().manual_seed(12)
    base = {"eye": torch.randn(7, 3, generator=gen) * 3.0, "center": torch.randn(7, 3, generator=gen) * 0.5,
            "up": torch.nn.functional.normalize(torch.randn(7, 3, generator=gen) + torch.tensor([0.0, 2.0, 0.0]), dim=1)}
    fov = torch.tensor([20.0, 40.0, 60.0, 35.0, 50.0, 75.0, 10.0], device=device)
    near, far = torch.full((7,), 0.05, device=device), torch.tensor([10.0, 20.0, 5.0, 8.0, 100.0, 9.0, 30.0], device=device)
    weights = torch.randn(7, 4, 4, generator=gen).to(device)
    results = {}
    for use_kernel in (True, False):
        leaves = {k: v.clone().to(device).requires_grad_(True) for k, v in base.items()}
        camera_utils.USE_CAMERA_KERNEL = use_kernel
        try:
            with _CountCalls("camera_transforms") as counter:
                m = camera_utils.clip_space_transforms(leaves["eye"], leaves["center"], leaves["up"], fov, near, far,
                                                       1.5, device)
        finally:
            camera_utils.USE_CAMERA_KERNEL = True
        assert counter.calls == (1 if use_kernel else 0)
        (m * weights).sum().backward()
        results[use_kernel] = (m.detach().cpu().numpy(), {k: v.grad.cpu().numpy() for k, v in leaves.items()})
    np.testing.assert_allclose(results[True][0], results[False][0], atol=1e-5, rtol=1e-5)
    for k, want_grad in results[False][1].items():
        assert np.abs(want_grad).max() > 1e-3, k
        np.testing.assert_allclose(results[True][1][k], want_grad, atol=2e-5, rtol=2e-4, err_msg=k)
    # the reference's assertions (camera_utils.py:68-69, 74-76)
    one = lambda *v: torch.tensor([list(v)], dtype=torch.float32, device=device)
    f1 = lambda v: torch.full((1,), float(v), device=device)
    with pytest.raises(AssertionError, match="eye and center are close"):
        camera_utils.clip_space_transforms(one(0, 0, 0), one(0, 0, 0), one(0, 1, 0), f1(40), f1(0.01), f1(10), 1.0, device)
    with pytest.raises(AssertionError, match="up and gaze are too close"):
        camera_utils.clip_space_transforms(one(0, 0, 0), one(0, 1, 0), one(0, 1, 0), f1(40), f1(0.01), f1(10), 1.0, device)
    # render() with the camera on the device: one launch, and the eye receives a gradient
    job = synthetic.sphere_job(2, 64, 48, 8)
    eye = job["eyes"].clone().to(device).requires_grad_(True)
    with _CountCalls("camera_transforms") as counter:
        img = mesh_renderer.render(job["vertices"].to(device), job["triangles"].to(device), job["normals"].to(device),
                                   job["diffuse"].to(device), eye, torch.zeros(2, 3, device=device),
                                   torch.tensor([0.0, 1.0, 0.0], device=device), job["light_positions"].to(device),
                                   job["light_intensities"].to(device), 64, 48)
    assert counter.calls == 1
    img[..., :3].mean().backward()
    assert eye.grad is not None and float(eye.grad.abs().max()) > 0


def test_tone_mapper_hip_matches_reference_golden_and_torch(device):
    """tone_mapper (render.py:389-419) as HIP passes: the reference's captured output, then torch's
    own pow / max / clamp semantics on images with zeros, values > 1, a negative entry (NaN power ->
    NaN maximum -> NaN image, as torch.max propagates it) and an all-zero image (0 / 0)."""
    g = golden_npz("camera_utils.npz")
    got = mesh_renderer.tone_mapper(torch.tensor(g["tone_in"], device=device), 0.7)
    assert got.is_cuda and got.dtype == torch.float32
    np.testing.assert_allclose(got.cpu().numpy(), g["tone_out"], atol=1e-6, rtol=0)

    gen = torch.Generator().manual_seed(3)
    x = torch.rand(4, 37, 29, 4, generator=gen) * 2.5
    x[0, 0, 0, 0] = 0.0
    x[1] = 0.0                      # all-zero image: 0 / 0
    x[2, 5, 5, 1] = -0.25           # negative base: NaN for a fractional gamma
    for gamma in (0.7, 1.0, 2.2):
        want = torch.clamp(torch.pow(x, gamma) / torch.pow(x, gamma).reshape(4, -1).max(dim=1).values
                           .reshape(4, 1, 1, 1), 0.0, 1.0).numpy()
        got = mesh_renderer.tone_mapper(x.to(device), gamma).cpu().numpy()
        assert np.array_equal(np.isnan(got), np.isnan(want)), gamma
        np.testing.assert_allclose(np.nan_to_num(got), np.nan_to_num(want), atol=2e-6, rtol=0)
        # fused 8-bit frames == the examples' host-side cast of the tone-mapped image
        frames = mesh_renderer.tone_mapper_uint8(x.to(device), gamma).cpu().numpy()
        ref = (np.clip(np.nan_to_num(got, nan=0.0), 0.0, 1.0) * np.float32(255.0)).astype(np.uint8)
        assert frames.dtype == np.uint8 and np.abs(frames.astype(int) - ref.astype(int)).max() <= 0
    # negative bases with an odd integer exponent give NEGATIVE powers ((-2)^3 = -8): they must lose the
    # maximum to every non-negative power, and win it -- as the largest value -- in an all-negative image
    y = torch.rand(3, 16, 16, 4, generator=gen) + 0.5
    y[0, 2, 3, 1] = -2.0                       # |power| = 8 > every positive power (< 3.4)
    y[1] = -(torch.rand(16, 16, 4, generator=gen) + 0.5)   # all negative: max = the one closest to zero
    y[2, 0, 0, 0] = -0.0
    p3 = torch.pow(y, 3.0)
    want = torch.clamp(p3 / p3.reshape(3, -1).max(dim=1).values.reshape(3, 1, 1, 1), 0.0, 1.0).numpy()
    got = mesh_renderer.tone_mapper(y.to(device), 3.0).cpu().numpy()
    np.testing.assert_allclose(got, want, atol=2e-6, rtol=0)
    # a gradient request takes the torch expression (the reference's op is differentiable)
    xg = (torch.rand(2, 8, 8, 3, generator=gen) + 0.1).to(device).requires_grad_(True)
    mesh_renderer.tone_mapper(xg, 0.7).sum().backward()
    assert xg.grad is not None and bool(torch.isfinite(xg.grad).all())
