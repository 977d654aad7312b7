"""GPU parity at the BASELINE configurations' REAL sizes.

The smaller parity tests pin the arithmetic; these pin the code paths that only go live at
size -- saturated LDS tables, 32-image grids, XCD remapping with padding blocks, 64 coarse cells,
~1000-entry cell lists -- by comparing single images of the full-size launches with the CPU oracle
(oracle/mr_oracle.c bit for bit, oracle/shading.py and oracle/soft.py within the north_star
tolerance of 1e-4 abs, plus a relative bound where 1e-4 would be vacuous)."""
import os
import sys

import numpy as np
import pytest
import torch

import oracle
from conftest import ROOT, seeded_dbary
from oracle import shading
from oracle import soft as oracle_soft
from pytorch_mesh_renderer_amd import _native, mesh_renderer, soft_mesh_renderer
from pytorch_mesh_renderer_amd.common import camera_utils, synthetic

pytestmark = pytest.mark.gpu
ATOL = 1e-4  # north_star: shaded RGBA and gradients within 1e-4 abs


def assert_close_abs_and_rel(got, want, what, rel=2e-3):
    """1e-4 abs (the bar) AND |delta| <= rel * max|want|: gradients of a mean over 10^8 elements are
    ~1e-8, so the absolute bar alone would pass anything."""
    got, want = np.asarray(got), np.asarray(want)
    assert np.isfinite(got).all(), what
    scale = float(np.abs(want).max())
    assert scale > 0, what + ": oracle gradient is identically zero"
    err = float(np.abs(got - want).max())
    assert err <= ATOL, "%s: max|d| = %.3e" % (what, err)
    assert err <= rel * scale, "%s: max|d| = %.3e vs max|want| = %.3e" % (what, err, scale)


def device_clip_bits(job, device):
    """The clip-space vertices exactly as render() forms them on the device (render.py's
    _render_fused: host-side camera matrices, then the library's own per-vertex transform -- the
    first stage of mr_render_forward), back on the host."""
    b = job["vertices"].shape[0]
    full = lambda v: torch.full((b,), float(v))
    transforms = camera_utils.clip_space_transforms(
        job["eyes"], torch.zeros(b, 3), torch.tensor([[0.0, 1.0, 0.0]]).repeat(b, 1), full(40.0), full(0.01),
        full(10.0), job["width"] / job["height"], device)
    return _native.vertex_transform(job["vertices"].to(device), transforms.to(torch.float32)).cpu()


def oracle_step_for_image(job, b, upstream_b, clip_bits):
    """oracle/shading.py render of image b, then backward of the loss's gradient image to that
    image's vertices; the rasterizer sees the device's clip-space bits.  upstream_b is
    d mean|img - target| / d img = sign(img - target) / N formed from the DEVICE's image: the sphere
    is rotation-symmetric, so image and (rotated-mesh) target differ by tessellation noise only and a
    pixel whose difference is below fp32 resolution would otherwise flip its sign -- and its whole
    gradient contribution -- between the two sides."""
    one = lambda t: t[b:b + 1].clone()
    v = one(job["vertices"]).requires_grad_(True)
    w, h = job["width"], job["height"]
    img = shading.render(v, job["triangles"], one(job["normals"]), one(job["diffuse"]), one(job["eyes"]),
                         torch.zeros(1, 3), torch.tensor([[0.0, 1.0, 0.0]]), one(job["light_positions"]),
                         one(job["light_intensities"]), w, h,
                         use_reference_kernel=False, clip_bits=clip_bits[b:b + 1])
    img.backward(gradient=upstream_b)
    return img.detach().numpy()[0], v.grad.numpy()[0]


def test_bench_step_configs2_full_size(device):
    """The EXACT step bench.py times (configs[2]: 5k tris, 1024x1024, batch 32; fused render forward +
    L1 loss + backward): images 5 and 21 and their vertex gradients against the oracle."""
    sys.path.insert(0, ROOT)
    import bench
    _, batch, width, height, k = bench.CONFIGS["c3"]
    assert (batch, width, height) == (32, 1024, 1024)
    job = synthetic.sphere_job(batch, width, height, k)
    step, vertices, state = bench.make_step(job, device, None)
    loss = step()
    image = state["image"].detach()
    assert image.shape == (32, 1024, 1024, 4) and bool(torch.isfinite(loss))
    grad = vertices.grad.detach().cpu().numpy()
    n_loss = image.numel()
    clip_bits = device_clip_bits(job, device)
    for b in (5, 21):
        upstream_b = (torch.sign(image[b:b + 1] - state["target"][b:b + 1]) / n_loss).cpu()
        want_img, want_grad = oracle_step_for_image(job, b, upstream_b, clip_bits)
        got_img = image[b].cpu().numpy()
        np.testing.assert_array_equal(got_img[..., 3], want_img[..., 3])      # coverage mask: exact
        np.testing.assert_allclose(got_img, want_img, atol=ATOL, rtol=0)
        assert_close_abs_and_rel(grad[b], want_grad, "d loss / d vertices[%d]" % b)
    # every image took part: no all-zero gradient rows, loss equals the mean of the per-image means
    assert (np.abs(grad).reshape(32, -1).max(1) > 0).all()
    per_image = torch.abs(image - state["target"]).mean(dim=(1, 2, 3))
    assert abs(float(per_image.mean()) - float(loss)) < 1e-6


def test_bench_step_from_world_space_inputs_records_the_conditioning_bound(device):
    """What test_bench_step_configs2_full_size holds fixed, measured instead of assumed:
    the oracle here forms its OWN clip-space vertices from the world-space inputs (host matmul, another
    summation order than the device's per-vertex transform).  Silhouette triangles -- seen edge-on --
    amplify that last-bit difference: a handful of the 4.2 M values of an image then differ by more than
    1e-4.  The bound below is the recorded state (round 3, images 5 and 21: 12 values per image above 1e-4,
    max 3.7e-4, coverage mask identical; round 2 had seen 12 values, max 1.73e-4, on one image);
    a regression of the transform, the rasterizer or the shading shows up as more / larger outliers.
    The coverage mask may differ in a few pixels for the same reason; everything else agrees to 1e-4."""
    sys.path.insert(0, ROOT)
    import bench
    _, batch, width, height, k = bench.CONFIGS["c3"]
    job = synthetic.sphere_job(batch, width, height, k)
    step, vertices, state = bench.make_step(job, device, None)
    step()
    image = state["image"].detach()
    worst = {"outliers": 0, "max": 0.0, "alpha": 0}
    for b in (5, 21):
        one = lambda t: t[b:b + 1].clone()
        want = shading.render(one(job["vertices"]), job["triangles"], one(job["normals"]), one(job["diffuse"]),
                              one(job["eyes"]), torch.zeros(1, 3), torch.tensor([[0.0, 1.0, 0.0]]),
                              one(job["light_positions"]), one(job["light_intensities"]), width, height,
                              use_reference_kernel=False).numpy()[0]
        got = image[b].cpu().numpy()
        d = np.abs(got - want)
        worst["outliers"] = max(worst["outliers"], int((d > ATOL).sum()))
        worst["max"] = max(worst["max"], float(d.max()))
        worst["alpha"] = max(worst["alpha"], int((got[..., 3] != want[..., 3]).sum()))
    print("host-clip-bits bound (per image):", worst)
    assert worst["outliers"] <= 24, worst        # of 4 194 304 values per image (measured: 12)
    assert worst["alpha"] <= 4, worst
    # an outlier is a pixel on a silhouette edge that one side covers and the other does not, or whose
    # winning triangle differs: its colour moves by at most the shading's range, not by garbage
    assert worst["max"] <= 1.0 + 1e-6, worst
    if worst["alpha"] == 0:
        assert worst["max"] <= 5e-4, worst


def test_device_clip_transform_against_float64(device):
    """"conditioning" as a checked fact.  The two sides of the test above differ in
    how they form clip = M (v, 1): the device evaluates each row as ((m0 x + m1 y) + m2 z) + m3 per vertex,
    un-fused (k_vertex_transform), the oracle -- like the reference -- through a batched GEMM on the host
    (fused multiply-adds, another order).  Both are compared here with the same product evaluated in
    float64 from the SAME float32 matrices and vertices, for all 32 x 2502 x 4 components of configs[2],
    in units of one rounding step -- an ulp of the largest of the component's four products (x and y pass
    through 0 where the products cancel; an ulp of the result itself would be meaningless there):
      * the device stays within 3 such steps everywhere (four products and three sums, half a step each:
        3.5 at most; recorded 2.3, mean 0.32) and equals the correctly rounded value in > 70 % of the
        components; the host GEMM: recorded max 2.0, mean 0.29, 81 % in the build container (fused
        multiply-adds: slightly closer) and exactly the device's figures on the GPU box's host,
      * and the two float32 sides are never more than 5 steps apart.
    So neither transform is defective and neither is off by more than its own rounding: the handful of
    silhouette pixels that move between the two sides in the test above follow from WHICH last bit each
    side rounded to, amplified by triangles seen edge-on."""
    sys.path.insert(0, ROOT)
    import bench
    _, batch, width, height, k = bench.CONFIGS["c3"]
    job = synthetic.sphere_job(batch, width, height, k)
    b = batch
    full = lambda v: torch.full((b,), float(v))
    transforms = camera_utils.clip_space_transforms(
        job["eyes"], torch.zeros(b, 3), torch.tensor([[0.0, 1.0, 0.0]]).repeat(b, 1), full(40.0), full(0.01),
        full(10.0), width / height, torch.device("cpu")).to(torch.float32)
    dev_clip = _native.vertex_transform(job["vertices"].to(device), transforms.to(device)).cpu().numpy()
    host_clip = camera_utils.transform_homogeneous(transforms, job["vertices"]).numpy()
    m64 = transforms.numpy().astype(np.float64)                       # [B,4,4]
    v64 = np.concatenate([job["vertices"].numpy().astype(np.float64),
                          np.ones((b, job["vertices"].shape[1], 1))], axis=2)   # [B,V,4]
    exact = np.einsum("brk,bvk->bvr", m64, v64)
    rounded = exact.astype(np.float32)
    terms = np.abs(m64[:, None, :, :] * v64[:, :, None, :]).max(-1)   # [B,V,4]: largest product per component
    step = np.spacing(np.maximum(terms, np.abs(exact)).astype(np.float32)).astype(np.float64)
    dev_err = np.abs(dev_clip.astype(np.float64) - exact) / step
    host_err = np.abs(host_clip.astype(np.float64) - exact) / step
    print("clip transform vs float64, in rounding steps of the largest product: device max %.2f mean %.3f | host GEMM "
          "max %.2f mean %.3f | correctly rounded: device %.1f %%, host %.1f %% of the components" % (
              dev_err.max(), dev_err.mean(), host_err.max(), host_err.mean(),
              100.0 * (dev_clip == rounded).mean(), 100.0 * (host_clip == rounded).mean()))
    assert dev_err.max() <= 3.0, dev_err.max()
    assert host_err.max() <= 3.0, host_err.max()
    assert (dev_clip == rounded).mean() >= 0.7
    assert abs(dev_err.mean() - host_err.mean()) <= 0.1, (dev_err.mean(), host_err.mean())
    assert (np.abs(dev_clip.astype(np.float64) - host_clip.astype(np.float64)) / step).max() <= 5.0
    # the emulation of the device's own expression in numpy float32 gives the device's bits exactly
    m, v = transforms.numpy(), job["vertices"].numpy()
    for r in range(4):
        want = ((m[:, None, r, 0] * v[..., 0] + m[:, None, r, 1] * v[..., 1]) + m[:, None, r, 2] * v[..., 2]) + m[:, None, r, 3]
        np.testing.assert_array_equal(dev_clip[..., r], want)


def test_config5_soft_renderer_default_parameters_crop_against_oracle(device):
    """configs[4] at the DEFAULT sigma / gamma (1e-5 / 1e-4), 5k triangles, 512x512, B = 16 (VERDICT r2,
    weak 2): a 64x64 crop of image 9 of the full-size launch -- across the sphere's silhouette, where
    the soft aggregation is live -- against oracle/soft.py evaluated on exactly those pixels
    (5000 triangles x 4096 pixels, dense), and d/dvertices of a loss restricted to the crop."""
    B, W, H, K = 16, 512, 512, 50
    job = synthetic.sphere_job(B, W, H, K)
    tris_d = job["triangles"].to(device)
    v = job["vertices"].to(device).requires_grad_(True)
    up = torch.tensor([0.0, 1.0, 0.0], device=device)
    img = soft_mesh_renderer.render(v, tris_d, job["diffuse"].to(device), job["eyes"].to(device),
                                    torch.zeros(B, 3, device=device), up, job["light_positions"].to(device),
                                    torch.ones(B, 1, device=device), W, H)
    b = 9
    alpha = img[b, ..., 3].detach()
    # a window that straddles the silhouette: the leftmost covered column of the middle row
    row = alpha[H // 2] > 0.5
    x_edge = int(torch.nonzero(row)[0])
    x0, y0, cw, ch = max(0, x_edge - 32), H // 2 - 32, 64, 64
    window = (x0, y0, cw, ch)
    one = lambda t: t[b:b + 1].clone()
    vc = one(job["vertices"]).requires_grad_(True)
    want = oracle_soft.render(vc, job["triangles"], one(job["diffuse"]), one(job["eyes"]), torch.zeros(1, 3),
                              torch.tensor([[0.0, 1.0, 0.0]]), one(job["light_positions"]), torch.ones(1, 1),
                              W, H, window=window)
    got = img[b:b + 1, y0:y0 + ch, x0:x0 + cw]
    frac = float((want[..., 3] > 0.5).float().mean())
    assert 0.2 < frac < 0.9, "the crop must straddle the silhouette (covered fraction %.2f)" % frac
    np.testing.assert_array_equal((got[..., 3] > 0.5).cpu().numpy(), (want[..., 3] > 0.5).numpy())
    np.testing.assert_allclose(got.detach().cpu().numpy(), want.detach().numpy(), atol=ATOL, rtol=0)
    wts = torch.rand(want.shape, generator=torch.Generator().manual_seed(5)) / want.numel()
    (want * wts).sum().backward()
    (got * wts.to(device)).sum().backward()
    assert_close_abs_and_rel(v.grad[b].cpu().numpy(), vc.grad[0].numpy(), "config5 d/dvertices, default sigma/gamma, crop",
                             rel=5e-3)


def test_config4_shape_backward_and_render(device):
    """configs[3] per-GPU share: 50k-tri sphere (K=158), 2048x2048, 8 images.  Rasterizer forward for
    the batch (image 6 bit-exact vs the oracle), rasterizer backward for all 8 images vs the oracle,
    render() forward + backward with image 3 vs oracle/shading.py."""
    B, W, H, K = 8, 2048, 2048, 158
    job = synthetic.sphere_job(B, W, H, K)
    assert job["triangles"].shape[0] == 49928
    clip, tris = job["clip"], job["triangles"]
    clip_d, tris_d = clip.to(device), tris.to(device)
    ids, bary, z = _native.rasterize_forward(clip_d, tris_d, W, H)
    o_ids, o_bary, o_z = oracle.forward(clip[6].numpy(), tris.numpy(), W, H)
    assert ids[6].cpu().numpy().tobytes() == o_ids.tobytes()
    assert bary[6].cpu().numpy().tobytes() == o_bary.tobytes()
    assert z[6].cpu().numpy().tobytes() == o_z.tobytes()
    covered = bary.sum(-1) > 0.5
    assert 0.70 < float(covered.float().mean()) < 0.78
    # rasterizer backward, all 8 images, upstream gradient randn / (H W)
    dbary = torch.stack([seeded_dbary((H, W, 3), seed=b) for b in range(B)])
    d = _native.rasterize_backward(dbary.to(device), clip_d, tris_d, ids, bary).cpu().numpy()
    want = oracle.backward(dbary.numpy(), clip.numpy(), tris.numpy(), ids.cpu().numpy(), bary.cpu().numpy(),
                           threads=min(8, oracle.max_threads()))
    assert_close_abs_and_rel(d, want, "config4 rasterizer backward")
    assert np.all(d[:, :, 2] == 0)
    del dbary, d, want
    # render() forward + L1 + backward on the batch; image 3 against the oracle
    leaves = {k2: job[k2].clone().to(device).requires_grad_(True) for k2 in ("vertices", "normals", "diffuse")}
    up = torch.tensor([0.0, 1.0, 0.0])
    img = mesh_renderer.render(leaves["vertices"], tris_d, leaves["normals"], leaves["diffuse"], job["eyes"],
                               torch.zeros(B, 3), up, job["light_positions"].to(device),
                               job["light_intensities"].to(device), W, H)
    target = torch.rand(1, H, W, 4, generator=torch.Generator().manual_seed(3))
    target_d = target.to(device).expand(B, H, W, 4)
    mesh_renderer.losses.l1_loss(img, target_d.contiguous()).backward()
    b = 3
    one = lambda t: t[b:b + 1].clone()
    cpu = {k2: one(job[k2]).requires_grad_(True) for k2 in leaves}
    ref = shading.render(cpu["vertices"], tris, cpu["normals"], cpu["diffuse"], one(job["eyes"]), torch.zeros(1, 3),
                         up.unsqueeze(0), one(job["light_positions"]), one(job["light_intensities"]), W, H,
                         clip_bits=device_clip_bits(job, device)[b:b + 1])
    ref.backward(gradient=(torch.sign(img[b:b + 1].detach().cpu() - target) / img.numel()))
    np.testing.assert_allclose(img[b].detach().cpu().numpy(), ref[0].detach().numpy(), atol=ATOL, rtol=0)
    for k2 in leaves:
        assert_close_abs_and_rel(leaves[k2].grad[b].cpu().numpy(), cpu[k2].grad[0].numpy(), "config4 d/d" + k2)


def test_config5_soft_renderer_full_batch(device):
    """configs[4]: SoftRas, 5k-tri sphere, 512x512, batch 16, forward + full backward AT SIZE; the
    oracle (a dense [triangles x pixels] torch evaluation) checks image 9 on a 96x96 crop-free
    rerender of the same job at a size it can afford, and batch-consistency pins the rest."""
    B, W, H, K = 16, 512, 512, 50
    job = synthetic.sphere_job(B, W, H, K)
    tris_d = job["triangles"].to(device)
    v = job["vertices"].to(device).requires_grad_(True)
    up = torch.tensor([0.0, 1.0, 0.0], device=device)
    args = (job["diffuse"].to(device), job["eyes"].to(device), torch.zeros(B, 3, device=device), up,
            job["light_positions"].to(device), torch.ones(B, 1, device=device))
    img = soft_mesh_renderer.render(v, tris_d, *args, W, H)
    assert img.shape == (B, H, W, 4) and bool(torch.isfinite(img).all())
    alpha = img[..., 3]
    frac = (alpha > 0.5).float().mean(dim=(1, 2))
    assert bool(((frac > 0.70) & (frac < 0.78)).all())          # every image shows the sphere's silhouette
    w8 = torch.rand(img.shape, generator=torch.Generator().manual_seed(8)).to(device) / img.numel()
    (img * w8).sum().backward()
    g_full = v.grad.detach().clone()
    assert bool(torch.isfinite(g_full).all()) and bool((g_full.abs().reshape(B, -1).max(1).values > 0).all())
    # batch consistency: image 9 rendered alone (B = 1) gives the same pixels and the same gradient
    b = 9
    v1 = job["vertices"][b:b + 1].to(device).requires_grad_(True)
    img1 = soft_mesh_renderer.render(v1, tris_d, *[a[b:b + 1] if a.dim() > 1 else a for a in args], W, H)
    np.testing.assert_allclose(img1[0].detach().cpu().numpy(), img[b].detach().cpu().numpy(), atol=1e-6, rtol=0)
    (img1 * w8[b:b + 1]).sum().backward()
    scale = float(g_full[b].abs().max())
    assert float((v1.grad[0] - g_full[b]).abs().max()) <= 1e-3 * scale
    # the same camera / mesh at 96x96 against the oracle (5000 triangles x 9216 pixels, dense)
    Ws = Hs = 96
    small = synthetic.sphere_job(B, Ws, Hs, K)
    one = lambda t: t[b:b + 1].clone()
    vg = one(small["vertices"]).to(device).requires_grad_(True)
    soft = dict(sigma_val=1e-4, gamma_val=1e-2)   # sub-pixel triangles at 96x96: the softer blend of SURVEY P15
    got = soft_mesh_renderer.render(vg, tris_d, one(small["diffuse"]).to(device), one(small["eyes"]).to(device),
                                    torch.zeros(1, 3, device=device), up, one(small["light_positions"]).to(device),
                                    torch.ones(1, 1, device=device), Ws, Hs, **soft)
    vc = one(small["vertices"]).requires_grad_(True)
    want = oracle_soft.render(vc, small["triangles"], one(small["diffuse"]), one(small["eyes"]), torch.zeros(1, 3),
                              torch.tensor([[0.0, 1.0, 0.0]]), one(small["light_positions"]), torch.ones(1, 1),
                              Ws, Hs, **soft)
    np.testing.assert_allclose(got.detach().cpu().numpy(), want.detach().numpy(), atol=ATOL, rtol=0)
    wts = torch.rand(want.shape, generator=torch.Generator().manual_seed(2)) / want.numel()
    (want * wts).sum().backward()
    (got * wts.to(device)).sum().backward()
    assert_close_abs_and_rel(vg.grad.cpu().numpy(), vc.grad.numpy(), "config5 d/dvertices (96x96)", rel=5e-3)


def test_deterministic_mode_is_bit_reproducible_at_configs2(device):
    """mr_set_deterministic: the reference accumulates sequentially and is reproducible
    (rasterize_triangles.cpp:156-157, 232-269); with the flag set, two runs of the benchmarked step
    (configs[2], render + L1 + backward) and of the rasterizer backward give IDENTICAL bits, and the
    values agree with the default float-atomic kernels far inside the parity tolerance."""
    sys.path.insert(0, ROOT)
    import bench
    _, batch, width, height, k = bench.CONFIGS["c3"]
    job = synthetic.sphere_job(batch, width, height, k)
    step, vertices, state = bench.make_step(job, device, None)
    step()
    default_grad = vertices.grad.detach().clone()
    clip_d, tris_d = job["clip"].to(device), job["triangles"].to(device)
    ids, bary, _ = _native.rasterize_forward(clip_d, tris_d, width, height)
    dbary = torch.stack([seeded_dbary((height, width, 3), seed=b) for b in range(batch)]).to(device)
    default_dclip = _native.rasterize_backward(dbary, clip_d, tris_d, ids, bary)
    before = _native.set_deterministic(True)
    try:
        runs = []
        for _ in range(2):
            loss = step()
            runs.append((float(loss), vertices.grad.detach().clone(),
                         _native.rasterize_backward(dbary, clip_d, tris_d, ids, bary)))
    finally:
        _native.set_deterministic(before)
    assert runs[0][0] == runs[1][0]
    assert torch.equal(runs[0][1], runs[1][1]), "vertex gradients differ between two deterministic runs"
    assert torch.equal(runs[0][2], runs[1][2]), "rasterizer backward differs between two deterministic runs"
    assert_close_abs_and_rel(runs[0][1].cpu().numpy(), default_grad.cpu().numpy(), "deterministic vs default step", rel=1e-4)
    assert_close_abs_and_rel(runs[0][2].cpu().numpy(), default_dclip.cpu().numpy(), "deterministic vs default raster bwd",
                             rel=1e-4)


def test_deterministic_mode_at_configs4_and_at_the_specular_c3_shape(device):
    """VERDICT r2 item 7, at the sizes it names: with mr_set_deterministic the SoftRas backward at configs[4]
    (5k triangles, 512^2, batch 16, default sigma / gamma) and the specular backward at configs[2]'s shape
    (5k triangles, 1024^2, batch 32) give bit-identical gradients on two runs, and agree with the default
    float-atomic kernels within the parity tolerance."""
    from pytorch_mesh_renderer_amd import _native, soft_mesh_renderer

    def soft_run():
        job = synthetic.sphere_job(16, 512, 512, 50)
        v = job["vertices"].clone().to(device).requires_grad_(True)
        kd = job["diffuse"].clone().to(device).requires_grad_(True)
        img = soft_mesh_renderer.render(v, job["triangles"].to(device), kd, job["eyes"], torch.zeros(16, 3),
                                        torch.tensor([0.0, 1.0, 0.0]), job["light_positions"].to(device),
                                        torch.ones(16, 1, device=device), 512, 512)
        img.mean().backward()
        return [v.grad.clone(), kd.grad.clone()]

    def spec_run():
        job = synthetic.sphere_job(32, 1024, 1024, 50)
        v = job["vertices"].clone().to(device).requires_grad_(True)
        kd = job["diffuse"].clone().to(device).requires_grad_(True)
        ks = torch.full_like(job["diffuse"], 0.5).to(device)
        # (tools/specular_bench.py's scene; d / d specular colours is ~1e-26 here -- the image-wide norm makes
        #  rn ~ 1e-3 and the exponent is 6 -- so the gradients compared are the vertices' and the diffuse colours')
        img = mesh_renderer.render(v, job["triangles"].to(device), job["normals"].to(device), kd,
                                   job["eyes"], torch.zeros(32, 3), torch.tensor([0.0, 1.0, 0.0]),
                                   job["light_positions"].to(device), job["light_intensities"].to(device), 1024, 1024,
                                   specular_colors=ks, shininess_coefficients=6.0)
        img.mean().backward()
        return [v.grad.clone(), kd.grad.clone()]

    for name, run in (("SoftRas configs[4]", soft_run), ("specular C3 shape", spec_run)):
        default = run()
        before = _native.set_deterministic(True)
        try:
            first, second = run(), run()
        finally:
            _native.set_deterministic(before)
        for i, (a, b, d) in enumerate(zip(first, second, default)):
            assert bool(torch.isfinite(a).all()) and float(a.abs().max()) > 0, (name, i)
            assert torch.equal(a, b), "%s: gradient %d differs between two deterministic runs" % (name, i)
            scale = float(d.abs().max())
            np.testing.assert_allclose(a.cpu().numpy(), d.cpu().numpy(), atol=1e-4 * scale, rtol=1e-3,
                                       err_msg="%s: deterministic vs default, gradient %d" % (name, i))


def test_config5_soft_step_time_guard(device):
    """the SoftRas step of BASELINE configs[4] (5k triangles, 512 x 512, batch 16, forward + mean() +
    backward to the vertices) is recorded at 0.76-0.79 ms (README); round 5's 2.5-3.3 ms readings
    were one generation-2 pass of CPython's garbage collector inside the 30-step loop (tools/gc_probe.py).  This guard
    runs bench.py's own c5 leg -- five chunks of 20 steps with the collector paused, each timed by the wall clock and by
    HIP events -- and fails if the MEDIAN chunk exceeds twice the recorded time, printing every chunk so that a failure
    says whether the GPU (events) or the host (enqueue) was slow."""
    sys.path.insert(0, ROOT)
    import bench
    j5 = synthetic.sphere_job(16, 512, 512, 50)
    v5 = j5["vertices"].to(device).requires_grad_(True)
    tri5, kd5, lp5 = j5["triangles"].to(device), j5["diffuse"].to(device), j5["light_positions"].to(device)
    eyes5, zero5, up5 = j5["eyes"], torch.zeros(16, 3), torch.tensor([0.0, 1.0, 0.0])
    li5 = torch.ones(16, 1, device=device)

    def step5():
        v5.grad = None
        soft_mesh_renderer.render(v5, tri5, kd5, eyes5, zero5, up5, lp5, li5, 512, 512).mean().backward()
    chunks = bench._chunked_ms(step5, chunks=5, n=20)
    print("configs[4] step, chunks of 20:", chunks)
    median = sorted(c["wall_ms"] for c in chunks)[2]
    recorded = 0.78
    assert median <= 2.0 * recorded, "SoftRas step %.3f ms (recorded %.2f): %s" % (median, recorded, chunks)
