"""GPU parity, kernel level: HIP rasterizer (through the C ABI) vs oracle and goldens.

Bar (BASELINE.json north_star): triangle ids, z-buffer and barycentrics BIT-EXACT;
gradients within 1e-4 abs under the normalised upstream gradient randn/(H*W).
"""
import numpy as np
import pytest
import torch

import oracle
from conftest import (golden_sphere_job, TRIANGLE_CASES, bits_equal, golden_json, golden_npz, seeded_dbary, sha)
from pytorch_mesh_renderer_amd import _native
from pytorch_mesh_renderer_amd.common import synthetic
from pytorch_mesh_renderer_amd.mesh_renderer.rasterize import rasterize_barycentric

pytestmark = pytest.mark.gpu
GRAD_ATOL = 1e-4  # north_star tolerance for gradients


def hip_forward(clip, tris, w, h, device):
    c = torch.as_tensor(clip).to(device)
    t = torch.as_tensor(tris).to(device)
    if c.dim() == 2:
        ids, bary, z = _native.rasterize_forward(c.unsqueeze(0), t, w, h)
        return ids[0].cpu().numpy(), bary[0].cpu().numpy(), z[0].cpu().numpy()
    ids, bary, z = _native.rasterize_forward(c, t, w, h)
    return ids.cpu().numpy(), bary.cpu().numpy(), z.cpu().numpy()


def hip_backward(dbary, clip, tris, ids, bary, device):
    args = [torch.as_tensor(np.ascontiguousarray(a)).to(device) for a in (dbary, clip, tris, ids, bary)]
    if args[1].dim() == 2:
        args = [a.unsqueeze(0) if i != 2 else a for i, a in enumerate(args)]
        return _native.rasterize_backward(*args)[0].cpu().numpy()
    return _native.rasterize_backward(*args).cpu().numpy()


def assert_forward_bitwise(got, want):
    for name, a, b in zip(("ids", "bary", "z"), got, want):
        assert bits_equal(a, b), "%s differs in %d elements" % (name, int((a != b).sum()))


def test_cube64_golden(device):
    g = golden_npz("raster_cube64.npz")
    got = hip_forward(g["clip"], g["triangles"], 64, 64, device)
    assert_forward_bitwise(got, (g["ids"], g["bary"], g["z"]))
    d = hip_backward(g["dbary"], g["clip"], g["triangles"], g["ids"], g["bary"], device)
    np.testing.assert_allclose(d, g["dclip"], atol=GRAD_ATOL, rtol=0)
    assert np.all(d[:, 2] == 0.0)


@pytest.mark.parametrize("case", TRIANGLE_CASES)
def test_triangle_cases_golden(device, case):
    g = golden_npz("raster_triangles_160x120.npz")
    clip, tris = g[case + ".clip"], g[case + ".triangles"]
    got = hip_forward(clip, tris, 160, 120, device)
    assert_forward_bitwise(got, (g[case + ".ids"], g[case + ".bary"], g[case + ".z"]))
    d = hip_backward(seeded_dbary((120, 160, 3), seed=1).numpy(), clip, tris, got[0], got[1], device)
    np.testing.assert_allclose(d, g[case + ".dclip"], atol=GRAD_ATOL, rtol=0)


def test_reference_test_triangles_640x480(device):
    g = golden_npz("raster_triangles_160x120.npz")
    h = golden_json("raster_triangles_640x480.json")
    for case in ("w_111", "w_perspective"):
        ids, bary, z = hip_forward(g[case + ".clip"], g[case + ".triangles"], 640, 480, device)
        assert sha(ids) == h[case]["ids"] and sha(bary) == h[case]["bary"] and sha(z) == h[case]["z"]


def test_jacobian_28x21(device):
    """Counterpart of testInternalRenderGradientComputation: the analytic Jacobian itself."""
    g = golden_npz("raster_jacobian_28x21.npz")
    got = hip_forward(g["clip"], g["triangles"], 28, 21, device)
    assert_forward_bitwise(got, (g["ids"], g["bary"], g["z"]))
    n = 21 * 28 * 3
    cols = list(range(0, n, 5))
    # one batched launch: image k carries the k-th unit upstream gradient
    e = np.zeros((len(cols), n), np.float32)
    e[np.arange(len(cols)), cols] = 1.0
    rep = lambda a: np.repeat(a[None], len(cols), 0)
    d = hip_backward(e.reshape(len(cols), 21, 28, 3), rep(g["clip"]), g["triangles"],
                     rep(g["ids"]), rep(g["bary"]), device)
    want = g["jacobian"][:, cols].T.reshape(len(cols), 8, 4)
    np.testing.assert_allclose(d, want, atol=1e-4, rtol=1e-5)


def test_sphere_256_b8_config2(device):
    """BASELINE config 2: 5k-tri sphere, 256x256, batch 8, forward G-buffer."""
    h = golden_json("raster_sphere_hashes.json")["c2_256x256_b8"]
    job = golden_sphere_job("sphere_clip_256_b8.npy")
    assert sha(job["clip"].numpy()) == h["clip"]
    ids, bary, z = hip_forward(job["clip"].numpy(), job["triangles"].numpy(), 256, 256, device)
    for b in range(8):
        assert sha(ids[b]) == h["cameras"][b]["ids"], b
        assert sha(bary[b]) == h["cameras"][b]["bary"], b
        assert sha(z[b]) == h["cameras"][b]["z"], b
    dbary = np.stack([seeded_dbary((256, 256, 3), seed=b).numpy() for b in range(8)])
    d = hip_backward(dbary, job["clip"].numpy(), job["triangles"].numpy(), ids, bary, device)
    cam0 = golden_npz("raster_sphere256_cam0.npz")
    np.testing.assert_allclose(d[0], cam0["dclip"], atol=GRAD_ATOL, rtol=0)
    want = oracle.backward(dbary, job["clip"].numpy(), job["triangles"].numpy(), ids, bary, threads=8)
    np.testing.assert_allclose(d, want, atol=GRAD_ATOL, rtol=0)


def test_sphere_1024_b32_config3_full_size(device):
    """BASELINE config 3 at full size: hashes of the reference's own output for 4 of
    the 32 cameras, size-independent properties for all of them."""
    h = golden_json("raster_sphere_hashes.json")["c3_1024x1024_b32"]
    dgold = golden_npz("raster_sphere1024_dclip.npz")
    job = golden_sphere_job("sphere_clip_1024_b32.npy")
    assert sha(job["clip"].numpy()) == h["clip"]
    clip_d, tris_d = job["clip"].to(device), job["triangles"].to(device)
    ids, bary, z = _native.rasterize_forward(clip_d, tris_d, 1024, 1024)
    for b in (0, 7, 16, 29):
        hb = h["cameras"][str(b)]
        assert sha(ids[b].cpu().numpy()) == hb["ids"], b
        assert sha(bary[b].cpu().numpy()) == hb["bary"], b
        assert sha(z[b].cpu().numpy()) == hb["z"], b
    # properties for every image: barycentrics sum to 1 on covered pixels, are exactly 0 with
    # z == 1 and id == 0 elsewhere; z in [-1, 1]; ids in range
    s = bary.sum(-1)
    covered = s > 0.5
    assert torch.all((s[covered] - 1.0).abs() < 1e-5)
    assert torch.all(bary[~covered] == 0) and torch.all(z[~covered] == 1.0) and torch.all(ids[~covered] == 0)
    assert torch.all(z[covered] >= -1.0) and torch.all(z[covered] <= 1.0)
    assert int(ids.min()) >= 0 and int(ids.max()) < 5000
    frac = covered.float().mean().item()
    assert 0.70 < frac < 0.78  # the sphere fills ~74 % of the frame
    # idempotence: a second launch gives the same bits
    ids2, bary2, z2 = _native.rasterize_forward(clip_d, tris_d, 1024, 1024)
    assert torch.equal(ids, ids2) and torch.equal(bary, bary2) and torch.equal(z, z2)
    # backward vs the reference's own df_dvertices
    dbary = torch.stack([seeded_dbary((1024, 1024, 3), seed=b) for b in range(32)]).to(device)
    d = _native.rasterize_backward(dbary, clip_d, tris_d, ids, bary).cpu().numpy()
    for b in (0, 7, 16, 29):
        np.testing.assert_allclose(d[b], dgold["dclip_%d" % b], atol=GRAD_ATOL, rtol=0)
    assert np.all(d[:, :, 2] == 0)
    # linearity of the backward in the upstream gradient
    d2 = _native.rasterize_backward(2.0 * dbary, clip_d, tris_d, ids, bary).cpu().numpy()
    np.testing.assert_allclose(d2, 2.0 * d, atol=1e-6, rtol=1e-4)
    # g == const  =>  gradient ~ 0 (barycentrics sum to one)
    ones = torch.full_like(dbary, 1.0 / (1024 * 1024))
    d1 = _native.rasterize_backward(ones, clip_d, tris_d, ids, bary).cpu().numpy()
    assert np.abs(d1).max() < 1e-4


@pytest.mark.parametrize("edge", [32, 64])
def test_both_region_sizes_are_bit_exact(device, edge):
    """The forward kernel picks 32x32 or 64x64 regions from the launch size; force each on the same
    inputs (soups with w <= 0, a 5k sphere, more triangles than one LDS bin holds)."""
    L = _native.lib()
    assert L.mr_debug_set_raster_region_edge(48) == _native.MR_EINVAL
    assert L.mr_debug_set_raster_region_edge(edge) == _native.MR_OK
    try:
        rng = np.random.default_rng(123)
        for trial in range(6):
            V, T = int(rng.integers(3, 200)), int(rng.integers(1, 1500))
            W, H = int(rng.integers(1, 300)), int(rng.integers(1, 260))
            clip = rng.normal(size=(2, V, 4)).astype(np.float32)
            if trial % 2:
                clip[..., 3] = np.abs(clip[..., 3]) + 0.1
            tris = rng.integers(0, V, size=(T, 3)).astype(np.int32)
            assert_forward_bitwise(hip_forward(clip, tris, W, H, device), oracle.forward(clip, tris, W, H))
        job = synthetic.sphere_job(3, 200, 136, 50)
        assert_forward_bitwise(hip_forward(job["clip"].numpy(), job["triangles"].numpy(), 200, 136, device),
                               oracle.forward(job["clip"].numpy(), job["triangles"].numpy(), 200, 136))
    finally:
        L.mr_debug_set_raster_region_edge(0)


def test_production_library_has_no_stage_probes(device):
    """The stage-timing probes of k_raster exist only in the -DMR_PROBES build (make probes)."""
    L = _native.lib()
    for probe in (1, 2, 3, 8, 16, 32, 40, 99):
        assert L.mr_debug_set_raster_probe(probe) == _native.MR_EINVAL
    assert L.mr_debug_set_raster_probe(0) == _native.MR_OK


@pytest.mark.parametrize("w,h", [(1, 1), (7, 3), (63, 65), (64, 64), (65, 129), (300, 200), (1000, 37)])
def test_ragged_image_sizes(device, w, h):
    job = synthetic.sphere_job(3, w, h, 12)
    want = oracle.forward(job["clip"].numpy(), job["triangles"].numpy(), w, h)
    got = hip_forward(job["clip"].numpy(), job["triangles"].numpy(), w, h, device)
    assert_forward_bitwise(got, want)
    dbary = np.stack([seeded_dbary((h, w, 3), seed=b).numpy() for b in range(3)])
    d = hip_backward(dbary, job["clip"].numpy(), job["triangles"].numpy(), got[0], got[1], device)
    np.testing.assert_allclose(
        d, oracle.backward(dbary, job["clip"].numpy(), job["triangles"].numpy(), want[0], want[1]),
        atol=GRAD_ATOL, rtol=0)


def test_random_soups_vs_oracle(device):
    """Random triangle soups: vertices behind the eye, degenerate and repeated triangles."""
    rng = np.random.default_rng(99)
    for trial in range(10):
        V, T = int(rng.integers(3, 60)), int(rng.integers(1, 300))
        W, H = int(rng.integers(1, 200)), int(rng.integers(1, 150))
        B = int(rng.integers(1, 4))
        clip = rng.normal(size=(B, V, 4)).astype(np.float32)
        if trial % 2:
            clip[..., 3] = np.abs(clip[..., 3]) + 0.1
        tris = rng.integers(0, V, size=(T, 3)).astype(np.int32)
        tris[T // 2] = tris[0]            # duplicate triangle -> exact z tie
        want = oracle.forward(clip, tris, W, H)
        got = hip_forward(clip, tris, W, H, device)
        assert_forward_bitwise(got, want)


def assert_forward_equal_up_to_nan_encoding(got, want):
    """ids exact; z and barycentrics bit-identical wherever either side is a number.  Where BOTH are
    NaN only NaN-ness is compared: the sign / payload of a NaN is an artifact of the instruction
    set (x86 SSE makes 0xffc00000 for an invalid operation, gfx950 0x7fc00000; the fuzz in
    tests/fuzz_raster_gpu.py sees both directions), not a property of the algorithm."""
    assert np.array_equal(got[0], want[0])
    for name, a, b in (("bary", got[1], want[1]), ("z", got[2], want[2])):
        both_nan = np.isnan(a) & np.isnan(b)
        differ = (a.view(np.uint32) != b.view(np.uint32)) & ~both_nan
        assert not differ.any(), "%s differs in %d non-NaN elements" % (name, int(differ.sum()))


@pytest.mark.parametrize("exponent", [-62, -40, 0, 30, 48, 50, 52, 61])
def test_coverage_tolerance_across_coefficient_magnitudes(device, exponent):
    """Round 4: the tile walk selects a lane's candidates with a CONSERVATIVE coverage test (fused multiply-adds
    against a per-entry tolerance of 2^-20 (|a| + |b| + |c|), -inf once a coefficient could overflow: max > 2^100) and
    applies rasterize_triangles.cpp:96-97 exactly in the depth trip.  Scaling all clip coordinates by 2^k leaves
    the geometry alone and moves the edge coefficients (products of two coordinates) by 2^2k -- from denormal
    to either side of 2^100: bit-exact against the oracle at every magnitude, on a soup and on the 5k sphere."""
    rng = np.random.default_rng(7)
    scale = np.float32(2.0) ** np.float32(exponent)
    for trial in range(3):
        V, T = int(rng.integers(20, 120)), int(rng.integers(50, 900))
        W, H = int(rng.integers(60, 300)), int(rng.integers(40, 200))
        clip = (rng.normal(size=(2, V, 4)) * [0.6, 0.6, 1.0, 1.0]).astype(np.float32)
        clip[..., 3] = np.abs(clip[..., 3]) + 0.05
        tris = rng.integers(0, V, size=(T, 3)).astype(np.int32)
        assert_forward_bitwise(hip_forward(clip * scale, tris, W, H, device), oracle.forward(clip * scale, tris, W, H))
    job = synthetic.sphere_job(2, 192, 128, 50)
    clip = job["clip"].numpy() * scale
    got = hip_forward(clip, job["triangles"].numpy(), 192, 128, device)
    assert_forward_bitwise(got, oracle.forward(clip, job["triangles"].numpy(), 192, 128))
    assert (got[1].sum(-1) > 0.5).mean() > 0.3   # the sphere is drawn at every magnitude


def test_random_soups_with_nonfinite_vertices(device):
    """NaN / Inf / 1e38 coordinates sprinkled over random soups: a NaN depth PASSES the
    reference's z-test (cpp:401) and is stored, so NaNs reach the G-buffer."""
    rng = np.random.default_rng(7)
    nan_pixels = 0
    for trial in range(16):
        V, T = int(rng.integers(3, 300)), int(rng.integers(1, 500))
        W, H = int(rng.integers(1, 400)), int(rng.integers(1, 300))
        B = int(rng.integers(1, 4))
        clip = rng.normal(size=(B, V, 4)).astype(np.float32)
        flat = clip.reshape(-1)
        idx = rng.integers(0, flat.size, size=max(1, flat.size // 50))
        flat[idx] = rng.choice([np.nan, np.inf, -np.inf, 1e38, -1e38], size=idx.size).astype(np.float32)
        tris = rng.integers(0, V, size=(T, 3)).astype(np.int32)
        want = oracle.forward(clip, tris, W, H)
        got = hip_forward(clip, tris, W, H, device)
        assert_forward_equal_up_to_nan_encoding(got, want)
        nan_pixels += int(np.isnan(want[2]).sum())
    assert nan_pixels > 0   # the case is actually exercised


@pytest.mark.parametrize("edge", [0, 32, 64])
def test_many_triangles_take_the_super_cell_level_bit_exactly(device, edge):
    """Round 5: from 16384 triangles on (and more than one super-cell of 4 x 4 cells in the image) the coarse binning gets
    a level above the cells (k_coarse_top: per-super-cell id-ordered lists, the cells scan those instead of all T).  20 000
    random small triangles -- duplicates for exact depth ties, a few huge ones that touch every super-cell, some behind
    the eye -- on an image that is not a multiple of anything: ids, depths and barycentrics bit for bit the oracle's."""
    rng = np.random.default_rng(20)
    T, w, h = 20000, 1100, 1300
    centre = rng.uniform(-1.1, 1.1, size=(T, 1, 2))
    offs = rng.normal(size=(T, 3, 2)) * 0.015
    offs[::997] *= 60.0                                        # a few triangles as large as the image
    xy = (centre + offs).astype(np.float32)
    clip = np.zeros((1, 3 * T, 4), np.float32)
    clip[0, :, :2] = xy.reshape(-1, 2)
    clip[0, :, 2] = rng.uniform(-0.9, 0.9, size=3 * T)
    clip[0, :, 3] = rng.uniform(0.7, 1.3, size=3 * T)
    clip[0, rng.integers(0, 3 * T, size=40), 3] *= -1.0        # some vertices behind the eye (full-screen bbox path)
    tris = np.arange(3 * T, dtype=np.int32).reshape(T, 3)
    tris[T // 2] = tris[7]
    tris[T - 1] = tris[11][::-1]                               # duplicates: exact ties, the later id wins
    assert _native.lib().mr_debug_set_raster_region_edge(edge) == 0
    try:
        ids, bary, z = _native.rasterize_forward(torch.from_numpy(clip).to(device), torch.from_numpy(tris).to(device), w, h)
    finally:
        _native.lib().mr_debug_set_raster_region_edge(0)
    o_ids, o_bary, o_z = oracle.forward(clip, tris, w, h, threads=8)
    assert bits_equal(ids.cpu().numpy(), o_ids) and bits_equal(z.cpu().numpy(), o_z) and bits_equal(bary.cpu().numpy(), o_bary)
    assert (o_ids > 0).mean() > 0.3


def test_bin_overflow_many_triangles_one_region(device):
    """More triangles over one 64x64 region than the LDS bin holds (multi-pass path)."""
    rng = np.random.default_rng(5)
    T = 3000
    centers = rng.uniform(-0.9, 0.9, size=(T, 1, 2)).astype(np.float32)
    offs = rng.uniform(-0.3, 0.3, size=(T, 3, 2)).astype(np.float32)
    xy = (centers + offs).reshape(-1, 2)
    zz = rng.uniform(-0.9, 0.9, size=(T * 3, 1)).astype(np.float32)
    clip = np.concatenate([xy, zz, np.ones((T * 3, 1), np.float32)], 1)
    tris = np.arange(T * 3, dtype=np.int32).reshape(T, 3)
    want = oracle.forward(clip, tris, 64, 48)
    got = hip_forward(clip, tris, 64, 48, device)
    assert_forward_bitwise(got, want)
    assert (want[1].sum(-1) > 0.5).mean() > 0.9


def test_region_lists_past_the_cell_stash(device):
    """6000 small triangles over one 256x256 cell: the cell's list is far longer than k_coarse's LDS stash
    (1024 hits) while most 64x64 regions keep a list of their own (<= 512 ids) and the crowded ones
    fall back to the cell's -- all three paths of the second binning level, bit-exact."""
    rng = np.random.default_rng(11)
    T = 6000
    centers = rng.uniform(-1.0, 1.0, size=(T, 1, 2)).astype(np.float32)
    centers[: T // 4] *= 0.25          # a crowd in the middle four regions
    offs = rng.uniform(-0.04, 0.04, size=(T, 3, 2)).astype(np.float32)
    xy = (centers + offs).reshape(-1, 2)
    zz = rng.uniform(-0.9, 0.9, size=(T * 3, 1)).astype(np.float32)
    clip = np.concatenate([xy, zz, np.ones((T * 3, 1), np.float32)], 1)
    tris = np.arange(T * 3, dtype=np.int32).reshape(T, 3)
    want = oracle.forward(clip, tris, 256, 256)
    got = hip_forward(clip, tris, 256, 256, device)
    assert_forward_bitwise(got, want)


def test_empty_inputs(device):
    clip = torch.zeros(2, 5, 4, device=device)
    tris = torch.zeros(0, 3, dtype=torch.int32, device=device)
    ids, bary, z = _native.rasterize_forward(clip, tris, 33, 17)
    assert torch.all(ids == 0) and torch.all(bary == 0) and torch.all(z == 1.0)
    d = _native.rasterize_backward(torch.ones(2, 17, 33, 3, device=device), clip, tris, ids, bary)
    assert d.shape == (2, 5, 4) and torch.all(d == 0)


def test_out_of_range_vertex_ids_are_skipped(device):
    g = golden_npz("raster_triangles_160x120.npz")
    clip = g["w_111.clip"]
    tris = np.array([[0, 1, 2], [0, 1, 99], [-1, 1, 2]], np.int32)
    got = hip_forward(clip, tris, 160, 120, device)
    assert_forward_bitwise(got, (g["w_111.ids"], g["w_111.bary"], g["w_111.z"]))


def test_non_finite_vertices_do_not_fault(device):
    clip = np.array([[np.nan, 0, 0, 1], [np.inf, 1, 0, 1], [1, 1, 0.5, 1], [-1, -1, 0.5, 1],
                     [1e30, -1e30, 0, 1e-30]], np.float32)
    tris = np.array([[0, 1, 2], [2, 3, 4], [1, 2, 3]], np.int32)
    want = oracle.forward(clip, tris, 40, 30)
    got = hip_forward(clip, tris, 40, 30, device)
    assert bits_equal(got[0], want[0])
    np.testing.assert_array_equal(got[1], want[1])  # NaN-aware equality
    np.testing.assert_array_equal(got[2], want[2])


def test_autograd_function_matches_reference_contract(device):
    """BarycentricRasterizer: return tuple, dtypes, init values, backward tuple (ext.py:6-63)."""
    g = golden_npz("raster_cube64.npz")
    clip = torch.tensor(g["clip"], device=device, requires_grad=True)
    tris = torch.tensor(g["triangles"], device=device)
    ids, bary, z = rasterize_barycentric(clip, tris, 64, 64)
    assert ids.dtype == torch.int32 and ids.shape == (64, 64)
    assert bary.dtype == torch.float32 and bary.shape == (64, 64, 3)
    assert z.dtype == torch.float32 and z.shape == (64, 64)
    assert bits_equal(bary.detach().cpu().numpy(), g["bary"])
    (bary * torch.tensor(g["dbary"], device=device)).sum().backward()
    np.testing.assert_allclose(clip.grad.cpu().numpy(), g["dclip"], atol=GRAD_ATOL, rtol=0)
    # batched form used internally
    idsb, baryb, zb = rasterize_barycentric(clip.detach().unsqueeze(0).repeat(3, 1, 1), tris, 64, 64)
    assert idsb.shape == (3, 64, 64) and torch.equal(idsb[2], ids)
    with pytest.raises(RuntimeError):
        rasterize_barycentric(clip, tris.long(), 64, 64)


def test_gradcheck_single_pixel(device):
    """Counterpart of testSimpleTriangleGradientComputation (eps 4e-2, atol 0.1, rtol 0.01)."""
    tris = torch.tensor([[0, 1, 2]], dtype=torch.int32, device=device)

    def pixel(clip):
        _, bary, _ = rasterize_barycentric(clip, tris, 640, 480)
        return bary[245:246, 325:326, :]

    clip = torch.tensor([[-0.5, -0.5, 0.8, 1.0], [0.0, 0.5, 0.3, 1.0], [0.5, -0.5, 0.3, 1.0]],
                        dtype=torch.float32, device=device, requires_grad=True)
    assert torch.autograd.gradcheck(pixel, clip, eps=4e-2, atol=0.1, rtol=0.01,
                                    nondet_tol=1e-5)


def test_config4_scale_single_image(device):
    """BASELINE config 4's per-image shape: 50k-tri sphere (K=158), 2048x2048 -- exercises 64
    coarse cells and ~900-entry cell lists; bit-exact vs the CPU oracle."""
    job = synthetic.sphere_job(1, 2048, 2048, 158)
    assert job["triangles"].shape[0] == 49928
    want = oracle.forward(job["clip"].numpy(), job["triangles"].numpy(), 2048, 2048)
    got = hip_forward(job["clip"].numpy(), job["triangles"].numpy(), 2048, 2048, device)
    assert_forward_bitwise(got, want)


def test_rccl_image_gather_single_rank(device):
    """The RCCL + side-stream hand-over path on one GPU (1-rank nccl group)."""
    import os
    import torch.distributed as dist
    from pytorch_mesh_renderer_amd import distributed
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29517")
    if not dist.is_initialized():
        dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=device)
    try:
        x = torch.rand(3, 17, 9, 4, device=device)
        g = distributed.ImageGather(3, force_collective=True)
        g.start(x)
        y = (x * 2).sum()            # work on the main stream while the gather runs
        out = g.wait()
        assert torch.equal(out, x) and float(y) > 0
        r = distributed.ImageGather(3, force_collective=True, mode="root")   # what bench.py uses
        r.wait()                      # nothing pending yet: must be a no-op
        r.start(x)
        r.wait()
        r.start(x + 1.0)              # back-to-back hand-overs, waited one step late
        assert torch.equal(r.wait(), x + 1.0)
        frames = (x * 255).to(torch.uint8)   # 8-bit frames, as bench.py hands them over
        r.start(frames)
        assert torch.equal(r.wait(), frames)
        # round 4: two hand-overs in flight (bench.py's depth), each waited for by its own event
        d2 = distributed.ImageGather(3, force_collective=True, mode="root", depth=2)
        d2.start(x)
        d2.start(x * 3.0)
        assert d2.in_flight() == 2
        first = d2.wait()
        d2.start(x * 5.0)                    # does not overwrite `first` (depth + 1 receive buffers)
        assert torch.equal(first, x) and torch.equal(d2.wait(), x * 3.0) and torch.equal(d2.wait(), x * 5.0)
        assert d2.wait() is None
        grad = torch.ones(5, 3, device=device)
        assert torch.equal(distributed.allreduce_shared_mesh_grad(grad.clone()), grad)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("handover,gather,transport", [("u8", "rotate", "rccl"), ("f32", "rotate", "rccl"), ("u8", "root", "rccl"),
                                                       ("f32", "rotate", "peer")])
def test_bench_rank_path_meets_rccl_on_one_gpu(device, handover, gather, transport):
    """bench.py's N > 1 path -- 8-bit frames written by the forward's epilogue (or the fp32 image), the
    side-stream hand-over, gather.wait() inside the timed loop, the render-only loop after it -- under
    a 1-rank RCCL group (MR_BENCH_FORCE_GROUP=1): the line names the backend and the rank count that
    torch.distributed reports, and carries both per-step figures.  gather = rotate: bench.py's default at N > 1
    (distributed.RotatingImageGather: the side-stream bookkeeping + all_to_all on RCCL); root: one gather per step;
    transport = peer (round 6): at one rank the frames are handed back without a copy."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MR_BENCH_FORCE_GROUP="1", MASTER_ADDR="127.0.0.1",
               MASTER_PORT={"u8rotaterccl": "29541", "f32rotaterccl": "29542", "u8rootrccl": "29543",
                            "f32rotatepeer": "29544"}[handover + gather + transport])
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    proc = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "4", "--warmup", "1",
                           "--cpu-sample", "0", "--handover", handover, "--gather", gather, "--transport", transport],
                          env=env, capture_output=True,
                          text=True, timeout=600)
    assert proc.returncode == 0, proc.stderr[-2000:]
    line = json.loads([ln for ln in proc.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["rccl"]["backend"] == "nccl" and line["rccl"]["ranks"] == 1
    assert line["rccl"]["handover_bytes_per_rank_per_step"] == 32 * 1024 * 1024 * (4 if handover == "u8" else 16)
    assert 0 < line["ms_per_step_render_only"] and line["ms_per_step_with_handover"] == line["ms_per_step"]
    assert line["handover_depth"] == 2 and line["config"]["gather"] == gather
    # (one rank: nothing arrives / leaves)
    assert line["handover_GBps_into_root" if gather == "root" else "handover_GBps_out_of_each_rank"] == 0.0
    assert line["rccl"]["gather"].startswith("rotating root" if gather == "rotate" else "every step's frames to rank 0")
    px = 32 * 1024 * 1024
    assert abs(line["value_render_only"] / (px / line["ms_per_step_render_only"] / 1e3) - 1.0) < 1e-3   # (rounded figures)
    assert line["config"]["handover"] == handover and line["n_gpus"] == 1
    if gather == "rotate":
        assert line["rccl"]["transport"] == transport
    if transport == "peer":   # (a 1-rank group hands the frames tensor back: the hand-over costs the step next to nothing)
        assert line["ms_per_step_with_handover"] <= line["ms_per_step_render_only"] * 1.10 + 0.03, line


def test_external_triangle_matches_reference_png_and_golden(device):
    """A triangle with one vertex behind the eye: the reference's own (unused) fixture
    test_data/External_Triangle.png under its own comparison, plus the float golden."""
    import os
    from PIL import Image
    from conftest import GOLDEN
    g = golden_npz("clip_external_triangle.npz")
    got = hip_forward(g["clip"], g["triangles"], 160, 120, device)
    assert_forward_bitwise(got, (g["ids"], g["bary"], g["z"]))
    d = hip_backward(seeded_dbary((120, 160, 3), seed=4).numpy(), g["clip"], g["triangles"], got[0], got[1], device)
    np.testing.assert_allclose(d, g["dclip"], atol=GRAD_ATOL, rtol=0)
    ids, bary, z = hip_forward(g["clip"], g["triangles"], 640, 480, device)
    assert [sha(ids), sha(bary), sha(z)] == list(g["sha_640x480"])
    baseline = np.asarray(Image.open(os.path.join(GOLDEN, "ref_png", "External_Triangle.png"))).astype(float) / 255.0
    outliers = np.any(np.abs(baseline - bary[::-1]) > 0.01, axis=2)     # test_utils.py:105-160
    assert outliers.mean() <= 0.001


@pytest.mark.parametrize("name", ["Barycentrics_Cube.png", "Simple_Tetrahedron.png"])
def test_unused_barycentric_fixtures_of_the_reference(device, name):
    """Two more of the reference's own (unused) fixtures: barycentrics as RGB, 640 x 480, under its own comparison;
    bit for bit the oracle's G-buffer."""
    from conftest import barycentric_png_scenes, png_outlier_fraction
    clip, tris = barycentric_png_scenes()[name]
    got = hip_forward(clip, tris, 640, 480, device)
    assert png_outlier_fraction(name, got[1]) <= 0.001
    assert_forward_bitwise(got, oracle.forward(clip, tris, 640, 480))


def test_camera_inside_cube_golden(device):
    """The eye inside a cube: side faces with vertices behind the eye cover the whole frame."""
    g = golden_npz("clip_camera_inside_cube.npz")
    got = hip_forward(g["clip"], g["triangles"], 160, 120, device)
    assert_forward_bitwise(got, (g["ids"], g["bary"], g["z"]))
    d = hip_backward(seeded_dbary((120, 160, 3), seed=6).numpy(), g["clip"], g["triangles"], got[0], got[1], device)
    np.testing.assert_allclose(d, g["dclip"], atol=GRAD_ATOL, rtol=0)


def test_vertex_normals_kernel_matches_reference(device):
    """compute_vertex_normals (src/common/meshes.py:3-35) as a HIP gather: reference capture, both ways."""
    from pytorch_mesh_renderer_amd.common import meshes
    g = golden_npz("vertex_normals_sphere_k8.npz")
    v = torch.tensor(g["vertices"], device=device, requires_grad=True)
    tris = torch.tensor(g["triangles"], device=device)
    n = meshes.compute_vertex_normals(v, tris)
    np.testing.assert_allclose(n.detach().cpu().numpy(), g["normals"], atol=1e-6, rtol=0)
    (n * torch.tensor(g["weights"], device=device)).sum().backward()
    np.testing.assert_allclose(v.grad.cpu().numpy(), g["d_vertices"], atol=2e-5, rtol=1e-4)
    # deterministic: no atomics anywhere in the kernel
    v2 = torch.tensor(g["vertices"], device=device, requires_grad=True)
    n2 = meshes.compute_vertex_normals(v2, tris)
    (n2 * torch.tensor(g["weights"], device=device)).sum().backward()
    assert torch.equal(n, n2) and torch.equal(v.grad, v2.grad)
    # a triangle with a vertex id out of range is skipped, a degenerate vertex gets the eps path
    bad = tris.clone()
    bad[0, 1] = v.shape[1] + 3
    assert bool(torch.isfinite(meshes.compute_vertex_normals(v.detach(), bad)).all())
    # any integer dtype, as the reference's triangles.long() takes (meshes.py:18); floats are an error
    assert torch.equal(meshes.compute_vertex_normals(v.detach(), tris.long()), n.detach())
    assert torch.equal(meshes.compute_vertex_normals(v.detach(), tris.cpu().to(torch.int16)), n.detach())
    with pytest.raises(RuntimeError):
        meshes.compute_vertex_normals(v.detach(), tris.float())
    # the backward reuses the adjacency its forward saved (no argsort when the cache on the tensor is gone)
    from pytorch_mesh_renderer_amd import _native
    v3 = torch.tensor(g["vertices"], device=device, requires_grad=True)
    n3 = meshes.compute_vertex_normals(v3, tris.long())
    calls = []
    real = _native.vertex_adjacency
    _native.vertex_adjacency = lambda *a, **k: calls.append(1) or real(*a, **k)
    try:
        (n3 * torch.tensor(g["weights"], device=device)).sum().backward()
    finally:
        _native.vertex_adjacency = real
    assert not calls and torch.equal(v3.grad, v.grad)


def test_vertex_normals_high_valence_and_odd_vertex_counts(device):
    """Round 3: eight lanes per vertex, a second trip for valence > 8.  A 21-triangle fan (hub valence 21,
    V = 23: the last group of eight lanes is partly past the end) in two images against the reference's
    scatter formulation (meshes.py:18-35) written with torch ops, both ways."""
    from pytorch_mesh_renderer_amd.common import meshes
    g = torch.Generator().manual_seed(5)
    n_rim = 22
    ang = torch.linspace(0, 5.5, n_rim)
    rim = torch.stack([torch.cos(ang), torch.sin(ang), 0.2 * torch.rand(n_rim, generator=g)], 1)
    verts = torch.cat([torch.tensor([[0.0, 0.0, 0.5]]), rim]).unsqueeze(0).repeat(2, 1, 1)
    verts[1] += 0.1 * torch.rand(23, 3, generator=g)
    tris = torch.tensor([[0, i, i + 1] for i in range(1, n_rim)], dtype=torch.int32)
    w = torch.rand(2, 23, 3, generator=g)

    def reference(v):
        t = tris.long()
        a, b, c = v[:, t[:, 0]], v[:, t[:, 1]], v[:, t[:, 2]]
        sums = torch.zeros_like(v)
        sums = sums.index_add(1, t[:, 0], torch.cross(b - a, c - a, dim=-1))
        sums = sums.index_add(1, t[:, 1], torch.cross(c - b, a - b, dim=-1))
        sums = sums.index_add(1, t[:, 2], torch.cross(a - c, b - c, dim=-1))
        return torch.nn.functional.normalize(sums, dim=2, eps=1e-6)

    vr = verts.clone().double().requires_grad_(True)
    (reference(vr) * w.double()).sum().backward()
    vd = verts.clone().to(device).requires_grad_(True)
    n = meshes.compute_vertex_normals(vd, tris.to(device))
    (n * w.to(device)).sum().backward()
    np.testing.assert_allclose(n.detach().cpu().numpy(), reference(verts.double()).numpy(), atol=2e-6, rtol=0)
    np.testing.assert_allclose(vd.grad.cpu().numpy(), vr.grad.numpy(), atol=2e-5, rtol=1e-4)


def test_deterministic_mode_flags_contributions_outside_its_fixed_point_range(device):
    """the deterministic mode converts every contribution to 64-bit fixed point; one that
    does not fit -- here an infinite upstream gradient on one pixel, and a NaN -- used to come back
    as a finite number of arbitrary sign.  Now the launch is flagged and the outputs are NaN; inputs
    in range still give the float path's values."""
    from pytorch_mesh_renderer_amd import _native
    g = golden_npz("raster_cube64.npz")
    clip, tris = torch.tensor(g["clip"], device=device), torch.tensor(g["triangles"], device=device)
    if clip.dim() == 2:
        clip = clip.unsqueeze(0)
    ids, bary, _ = _native.rasterize_forward(clip, tris, 64, 64)
    dbary = seeded_dbary((64, 64, 3), seed=2).unsqueeze(0).to(device)
    covered = (bary.sum(-1) > 0.5).nonzero()
    _, y, x = [int(t) for t in covered[len(covered) // 2]]
    want = _native.rasterize_backward(dbary, clip, tris, ids, bary)
    before = _native.set_deterministic(True)
    try:
        fine = _native.rasterize_backward(dbary, clip, tris, ids, bary)
        np.testing.assert_allclose(fine.cpu().numpy(), want.cpu().numpy(), atol=1e-7, rtol=1e-4)
        for poison in (float("inf"), float("nan")):
            bad = dbary.clone()
            bad[0, y, x, 1] = poison
            got = _native.rasterize_backward(bad, clip, tris, ids, bary)
            assert not bool(torch.isfinite(got).any()), "a contribution outside the range must poison the result"
    finally:
        _native.set_deterministic(before)
    floaty = dbary.clone()
    floaty[0, y, x, 1] = float("inf")
    assert not bool(torch.isfinite(_native.rasterize_backward(floaty, clip, tris, ids, bary)).all())


def test_rasterize_triangles_cpp_shim_is_a_drop_in(device):
    """`import rasterize_triangles_cpp` + the exact call sequence of the reference's
    rasterize_triangles_ext.py:37-59 against the shim module, on device tensors."""
    import importlib
    import os
    import sys
    shim_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                            "pytorch_mesh_renderer_amd", "shims")
    sys.path.insert(0, shim_dir)
    try:
        sys.modules.pop("rasterize_triangles_cpp", None)
        rasterize_triangles_cpp = importlib.import_module("rasterize_triangles_cpp")
    finally:
        sys.path.remove(shim_dir)
    g = golden_npz("raster_cube64.npz")
    clip_space_vertices = torch.tensor(g["clip"], device=device)
    triangles = torch.tensor(g["triangles"], device=device)
    px_triangle_ids, px_barycentric_coords, z_buffer = rasterize_triangles_cpp.forward(
        clip_space_vertices, triangles, 64, 64)
    assert bits_equal(px_triangle_ids.cpu().numpy(), g["ids"])
    assert bits_equal(px_barycentric_coords.cpu().numpy(), g["bary"])
    assert bits_equal(z_buffer.cpu().numpy(), g["z"])
    output = rasterize_triangles_cpp.backward(torch.tensor(g["dbary"], device=device), clip_space_vertices,
                                              triangles, px_triangle_ids, px_barycentric_coords)
    df_dvertices = output[0]
    assert isinstance(output, list) and df_dvertices.shape == (8, 4)
    np.testing.assert_allclose(df_dvertices.cpu().numpy(), g["dclip"], atol=GRAD_ATOL, rtol=0)
    with pytest.raises(RuntimeError):
        rasterize_triangles_cpp.forward(clip_space_vertices, triangles.long(), 64, 64)
    sys.modules.pop("rasterize_triangles_cpp", None)
