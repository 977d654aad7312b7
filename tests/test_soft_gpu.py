"""GPU parity for the SoftRas renderer: HIP kernels vs the reference's
known-answer matrices, vs goldens captured from the reference, and vs the CPU oracle."""
import numpy as np
import pytest
import torch

from conftest import golden_npz
from oracle import soft as oracle_soft
from pytorch_mesh_renderer_amd import soft_mesh_renderer
from pytorch_mesh_renderer_amd.common import synthetic
from pytorch_mesh_renderer_amd.soft_mesh_renderer.rasterize import rasterize_batch

pytestmark = pytest.mark.gpu
ATOL = 1e-4


def test_single_triangle_known_answers(device):
    """Counterpart of test_single_triangle_forward (test_rasterize.py:46-215)."""
    g = golden_npz("soft_single_triangle_10x10.npz")
    d = lambda k: torch.tensor(g[k], device=device)
    for tag in ("a", "b"):
        sig, gam, blur = [float(v) for v in g["params_" + tag]]
        img = rasterize_batch(d("clip"), d("triangles"), d("world"), d("normals"), d("diffuse"),
                              d("light_positions"), d("light_intensities"), 10, 10, sig, gam, blur)
        np.testing.assert_allclose(img.cpu().numpy(), g["image_" + tag], atol=ATOL, rtol=0)


def test_point_to_segment_nearest_vectors_on_the_device(device):
    """Counterpart of the reference's test_point_to_segment_nearest (test_rasterize.py:9-44): its three
    cases with the answers written there, and 256 more through the reference function, against the two
    device functions every SoftRas kernel evaluates (edge_setup + edge_nearest in soft.hip, reached through
    mr_debug_soft_nearest): nearest point, t and squared distance."""
    from pytorch_mesh_renderer_amd import _native
    g = golden_npz("soft_point_to_segment_nearest.npz")
    out = _native.debug_soft_nearest(torch.tensor(g["p"], device=device), torch.tensor(g["a"], device=device),
                                     torch.tensor(g["b"], device=device)).cpu().numpy()
    np.testing.assert_allclose(out[:3, :2], g["held_nearest"], atol=1e-6, rtol=0)     # the reference test's own
    np.testing.assert_allclose(out[:3, 2], g["held_t"], atol=1e-6, rtol=0)            # expected values
    np.testing.assert_allclose(out[:, 2], g["t"], atol=2e-6, rtol=0)
    np.testing.assert_allclose(out[:, :2], g["nearest"], atol=2e-6, rtol=0)
    np.testing.assert_allclose(out[:, 3], ((g["nearest"] - g["p"]) ** 2).sum(-1), atol=1e-6, rtol=1e-4)
    # t is clamped: never outside [0, 1], and the ends are reached exactly by most of the points the
    # reference clamps (soft.hip uses the 1-ulp hardware reciprocal / square root: a t within an ulp of an
    # end may land on either side of it)
    assert out[:, 2].min() >= 0.0 and out[:, 2].max() <= 1.0
    assert ((out[:, 2] == 0.0) & (g["t"] == 0.0)).sum() >= 0.9 * (g["t"] == 0.0).sum()
    assert ((out[:, 2] == 1.0) & (g["t"] == 1.0)).sum() >= 0.9 * (g["t"] == 1.0).sum()


@pytest.mark.parametrize("name", ["soft_sphere_k6_32.npz", "soft_sphere_k10_32.npz"])
def test_render_sphere_goldens(device, name):
    g = golden_npz(name)
    sig, gam = [float(v) for v in g["params"]]
    leaves = {k: torch.tensor(g[k], device=device, requires_grad=True)
              for k in ("vertices", "diffuse", "light_positions")}
    b = g["vertices"].shape[0]
    img = soft_mesh_renderer.render(
        leaves["vertices"], torch.tensor(g["triangles"], device=device), leaves["diffuse"],
        torch.tensor(g["eye"], device=device), torch.zeros(b, 3, device=device),
        torch.tensor(b * [[0.0, 1.0, 0.0]], device=device), leaves["light_positions"],
        torch.tensor(g["light_intensities"], device=device), 32, 32, sigma_val=sig, gamma_val=gam)
    got = img.detach().cpu().numpy()
    np.testing.assert_allclose(got[..., 3], g["image"][..., 3], atol=ATOL, rtol=0)
    np.testing.assert_allclose(got, g["image"], atol=ATOL, rtol=0)
    torch.mean(torch.abs(img - torch.tensor(g["target"], device=device))).backward()
    for k, t in leaves.items():
        np.testing.assert_allclose(t.grad.cpu().numpy(), g["d_" + k], atol=ATOL, rtol=0, err_msg=k)


def test_rasterize_batch_all_gradients_vs_oracle(device):
    """Every differentiable input of rasterize_batch, two lights, softer blend."""
    job = synthetic.sphere_job(1, 48, 40, 8)
    gen = torch.Generator().manual_seed(3)
    leaves_cpu = {
        "clip": job["clip"][0].clone(), "world": job["vertices"][0].clone(),
        "normals": job["normals"][0].clone(), "diffuse": torch.rand(job["vertices"].shape[1], 3, generator=gen),
        "lpos": torch.tensor([[0.5, 1.0, 3.0], [-2.0, 0.3, 2.0]]), "lint": torch.tensor([0.9, 0.6])}
    params = (3e-4, 5e-3, 0.03)
    cpu = {k: v.clone().requires_grad_(True) for k, v in leaves_cpu.items()}
    gpu = {k: v.clone().to(device).requires_grad_(True) for k, v in leaves_cpu.items()}
    ref = oracle_soft.rasterize_batch(cpu["clip"], job["triangles"], cpu["world"], cpu["normals"],
                                      cpu["diffuse"], cpu["lpos"], cpu["lint"], 48, 40, *params)
    img = rasterize_batch(gpu["clip"], job["triangles"].to(device), gpu["world"], gpu["normals"],
                          gpu["diffuse"], gpu["lpos"], gpu["lint"], 48, 40, *params)
    np.testing.assert_allclose(img.detach().cpu().numpy(), ref.detach().numpy(), atol=ATOL, rtol=0)
    wts = torch.rand(ref.shape, generator=gen) / ref.numel()
    (ref * wts).sum().backward()
    (img * wts.to(device)).sum().backward()
    for k in cpu:
        np.testing.assert_allclose(gpu[k].grad.cpu().numpy(), cpu[k].grad.numpy(), atol=ATOL, rtol=1e-3,
                                   err_msg=k)


def test_optimize_single_triangle_translation(device):
    """Counterpart of test_optimize_single_triangle_translation (test_rasterize.py:217-326)."""
    clip = torch.tensor([[-0.5, 0.0, 0.25, 1.0], [0.5, 1.0, 0.25, 1.0], [-0.5, 1.0, 0.25, 1.0]], device=device)
    world = torch.tensor([[-0.5, 0.0, 0.0], [0.5, 1.0, 0.0], [-0.5, 1.0, 0.0]], device=device)
    tris = torch.tensor([[0, 1, 2]], dtype=torch.int32, device=device)
    normals = torch.tensor([[0.0, 0.0, 1.0]] * 3, device=device)
    diffuse = torch.tensor([[1.0, 0.0, 0.0]] * 3, device=device)
    lpos = torch.tensor([[0.0, 0.0, 100000.0]], device=device)
    lint = torch.tensor([1.0], device=device)
    shift4 = lambda t: torch.stack([t, torch.zeros_like(t), torch.zeros_like(t), torch.zeros_like(t)])
    shift3 = lambda t: torch.stack([t, torch.zeros_like(t), torch.zeros_like(t)])
    target_x = torch.tensor(0.25, device=device)
    target = rasterize_batch(clip + shift4(target_x), tris, world + shift3(target_x), normals, diffuse,
                             lpos, lint, 10, 10, 1e-5, 1e-1, 0.01)
    sigma = float(-0.5 ** 2 / torch.special.logit(torch.tensor(1e-5)))
    tx = torch.zeros((), device=device, requires_grad=True)
    opt = torch.optim.SGD([tx], lr=0.3)
    for _ in range(60):
        opt.zero_grad()
        out = rasterize_batch(clip + shift4(tx), tris, world + shift3(tx), normals, diffuse, lpos, lint,
                              10, 10, sigma, 1e-1, 0.5)
        torch.mean(torch.abs(out - target)).backward()
        opt.step()
    assert abs(float(tx) - 0.25) < 0.1  # within half a pixel (a pixel is 0.2 NDC units)


def test_config5_shape_runs(device):
    """BASELINE config 5 at reduced batch: 5k-tri sphere, 512x512, forward + backward."""
    job = synthetic.sphere_job(2, 512, 512, 50)
    v = job["vertices"].to(device).requires_grad_(True)
    img = soft_mesh_renderer.render(v, job["triangles"].to(device), job["diffuse"].to(device),
                                    job["eyes"].to(device), torch.zeros(2, 3, device=device),
                                    torch.tensor([0.0, 1.0, 0.0], device=device),
                                    job["light_positions"].to(device), torch.ones(2, 1, device=device),
                                    512, 512)
    assert img.shape == (2, 512, 512, 4)
    alpha = img[..., 3]
    assert 0.70 < float((alpha > 0.5).float().mean()) < 0.78      # the sphere's silhouette
    img.mean().backward()
    assert torch.isfinite(v.grad).all() and float(v.grad.abs().max()) > 0


def test_soft_backward_deterministic_mode_is_bit_reproducible(device):
    """Round 3: mr_set_deterministic covers the SoftRas backward too -- the (wavefront, triangle) sums
    leave as 64-bit fixed-point integer atomics, scaled for the 1 / sigma and 1 / gamma the
    contributions carry at the default parameters: two runs give identical bits and agree with the
    float-atomic kernel.  5k-triangle sphere at 128x128, B = 2, default sigma / gamma."""
    from pytorch_mesh_renderer_amd import _native, soft_mesh_renderer
    from pytorch_mesh_renderer_amd.common import synthetic
    job = synthetic.sphere_job(2, 128, 128, 50)
    tris = job["triangles"].to(device)
    w = torch.rand(2, 128, 128, 4, generator=torch.Generator().manual_seed(3)).to(device) / (128 * 128)

    def run():
        leaves = {k: job[k].clone().to(device).requires_grad_(True) for k in ("vertices", "diffuse", "light_positions")}
        img = soft_mesh_renderer.render(leaves["vertices"], tris, leaves["diffuse"], job["eyes"].to(device),
                                        torch.zeros(2, 3, device=device), torch.tensor([0.0, 1.0, 0.0], device=device),
                                        leaves["light_positions"], torch.ones(2, 1, device=device), 128, 128)
        (img * w).sum().backward()
        return [leaves[k].grad.clone() for k in sorted(leaves)]

    default = run()
    before = _native.set_deterministic(True)
    try:
        first, second = run(), run()
    finally:
        _native.set_deterministic(before)
    for i, (a, b, d) in enumerate(zip(first, second, default)):
        assert bool(torch.isfinite(a).all()) and float(a.abs().max()) > 0, i
        assert torch.equal(a, b), "output %d differs between two deterministic runs" % i
        scale = float(d.abs().max())
        np.testing.assert_allclose(a.cpu().numpy(), d.cpu().numpy(), atol=1e-4 * scale, rtol=1e-3, err_msg=str(i))


def test_soft_backward_with_the_forward_pass_s_prepared_records(device):
    """Round 3: the autograd path hands the forward's records / candidate lists to the backward
    (mr_soft_prepared_bytes, `prepared`).  In deterministic mode the gradients are bit-identical to the
    ones of a backward that rebuilds them (prepared = NULL), also after other SoftRas calls have run in
    between (the kept workspace is private to the forward call)."""
    from pytorch_mesh_renderer_amd import _native
    from pytorch_mesh_renderer_amd.common import synthetic
    g = torch.Generator().manual_seed(11)
    job = synthetic.sphere_job(2, 96, 80, 20)
    V = job["vertices"].shape[1]
    pos = job["vertices"].to(device)
    clip = torch.cat([pos[..., :2] * 0.8, pos[..., 2:] * 0.1 + 0.5, torch.ones(2, V, 1, device=device) * 1.1], -1).contiguous()
    nrm = torch.nn.functional.normalize(pos, dim=-1).contiguous()
    dif = torch.rand(2, V, 3, generator=g).to(device)
    tris = job["triangles"].to(device)
    lp = torch.tensor([[[0.0, 3.0, 3.0]], [[2.0, 1.0, 3.0]]], device=device)
    li = torch.ones(2, 1, device=device)
    args = (clip, pos, nrm, dif, tris, lp, li)
    before = _native.set_deterministic(True)
    try:
        rgba, aux, kept = _native.soft_forward(*args, 96, 80, 1e-3, 1e-2, 0.02, keep_prepared=True)
        rgba2, aux2 = _native.soft_forward(*args, 96, 80, 1e-3, 1e-2, 0.02)
        assert torch.equal(rgba, rgba2) and torch.equal(aux, aux2)
        drgba = torch.rand(2, 80, 96, 4, generator=g).to(device)
        # another size in between: the shared workspace is rewritten, the kept one is not
        _native.soft_forward(clip[:1], pos[:1], nrm[:1], dif[:1], tris, lp[:1], li[:1], 64, 64, 1e-3, 1e-2, 0.05)
        with_kept = _native.soft_backward(drgba, rgba, aux, *args, 1e-3, 1e-2, 0.02, prepared=kept)
        rebuilt = _native.soft_backward(drgba, rgba, aux, *args, 1e-3, 1e-2, 0.02)
    finally:
        _native.set_deterministic(before)
    for i, (a, b) in enumerate(zip(with_kept, rebuilt)):
        assert float(a.abs().max()) > 0, i
        assert torch.equal(a, b), i
    with pytest.raises(ValueError):
        _native.soft_backward(drgba, rgba, aux, *args, 1e-3, 1e-2, 0.02, prepared=kept[:1000])
