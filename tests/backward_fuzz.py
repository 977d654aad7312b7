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

A failure carries its evidence (round 6; there is NO retry: round 5's one unexplained excess left nothing behind but a
log line).  Every case keeps host copies of all device inputs of its backward calls, taken BEFORE the first of them,
and after the last one (a) downloads the same tensors again and compares them bit for bit -- an input that changed
under the kernels (a stray write of any launch in between, a copy that raced) is reported as such, with the tensor's
name and the first differing element --, (b) holds the forward image the kernels were differentiated from against the
float64 truth's own image, which tells a bad forward from a bad backward.  On any failure of the case the inputs
(before and after), the truth and every kernel's outputs go into one .npz under Report.dump_dir
(MR_FUZZ_DUMP_DIR, default gpurun_out/fuzz_failures/) and the failure message names the file.
"""
import os

import numpy as np
import torch

from oracle import truth64
from pytorch_mesh_renderer_amd import _native

K_ROUNDING = 64.0   # see truth64.assert_within_rounding
FLOOR = 1e-7


class Report:
    """Largest excess (error / bound) seen per kernel, and the failures."""

    def __init__(self, dump_dir=None):
        self.worst, self.failures, self.trials, self.with_gradients = {}, [], 0, 0
        self.dump_dir = dump_dir or os.environ.get("MR_FUZZ_DUMP_DIR") or os.path.join(
            os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "fuzz_failures")
        self.dumps = []

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


class Evidence:
    """One case's evidence (module docstring): host copies of the device inputs of its backward calls taken before the
    first of them, every kernel's outputs, the truth; close() re-downloads the inputs, compares, and dumps on failure."""

    def __init__(self, report, what, truth_saw=None, **device_inputs):
        """truth_saw: {name: host array the float64 truth was computed from} for the inputs of the same name -- the
        device tensor must hold exactly those bits when the first backward call is made."""
        self.report, self.what, self.first_failure = report, what, len(report.failures)
        self.device = {k: v for k, v in device_inputs.items() if v is not None}
        torch.cuda.synchronize()
        self.before = {k: v.detach().cpu().numpy().copy() for k, v in self.device.items()}
        self.outputs, self.truth, self.notes = {}, {}, []
        for k, h in (truth_saw or {}).items():
            if h is None:
                continue
            h, d = np.ascontiguousarray(h), self.before[k]
            if h.dtype != d.dtype or h.shape != d.shape or h.tobytes() != d.tobytes():
                bad = np.flatnonzero(h.reshape(-1) != d.reshape(-1)) if h.shape == d.shape else np.zeros(1, np.int64)
                first = int(bad[0]) if bad.size else 0
                self.report.failures.append(
                    "%s: DEVICE INPUT `%s` is not what the truth was computed from: %d elements differ, first at %d (host %r, device %r)"
                    % (self.what, k, bad.size, first, h.reshape(-1)[first] if h.size else None, d.reshape(-1)[first] if d.size else None))
                self.truth["truth_saw/" + k] = h

    def keep(self, kernel, **tensors):
        for name, t in tensors.items():
            if t is not None:
                self.outputs["%s | %s" % (kernel, name)] = t

    def keep_truth(self, t, prefix=""):
        self.truth.update({prefix + k: np.asarray(v) for k, v in t.items() if isinstance(v, np.ndarray)})

    def forward_image(self, rgba, truth_image):
        """Diagnostic, not a criterion (a sliver's interpolated normal may flip a clamp in binary32): how far the image
        the kernels were differentiated from is from the float64 truth's own."""
        d = np.abs(rgba.detach().cpu().numpy().astype(np.float64) - truth_image)
        at = np.unravel_index(int(np.argmax(d)), d.shape)
        self.notes.append("forward image: max |device - float64| %.3e at %s, %d of %d values beyond 1e-3" % (
            float(d.max()), tuple(int(i) for i in at), int((d > 1e-3).sum()), d.size))

    def close(self, raised=False):
        torch.cuda.synchronize()
        after = {k: v.detach().cpu().numpy() for k, v in self.device.items()}
        changed = {}
        for k, a in self.before.items():
            b = after[k]
            if a.shape != b.shape or a.tobytes() != b.tobytes():
                changed[k] = b
                where = np.flatnonzero(np.frombuffer(a.tobytes(), np.uint8) != np.frombuffer(b.tobytes(), np.uint8)) \
                    if a.shape == b.shape else np.zeros(1, np.int64)
                self.report.failures.append(
                    "%s: DEVICE INPUT `%s` CHANGED under the backward calls: %d bytes differ, first at element %d (%r -> %r)" % (
                        self.what, k, where.size, int(where[0]) // a.itemsize,
                        a.reshape(-1)[int(where[0]) // a.itemsize], b.reshape(-1)[int(where[0]) // a.itemsize]))
        if len(self.report.failures) == self.first_failure and not raised:
            return
        os.makedirs(self.report.dump_dir, exist_ok=True)
        safe = "".join(c if c.isalnum() else "_" for c in self.what)[:80]
        path = os.path.join(self.report.dump_dir, "%s_%d.npz" % (safe, len(self.report.dumps)))
        arrays = {"in/" + k: v for k, v in self.before.items()}
        arrays.update({"in_after/" + k: v for k, v in changed.items()})
        arrays.update({"truth/" + k: v for k, v in self.truth.items()})
        arrays.update({"out/" + k: v.detach().cpu().numpy() for k, v in self.outputs.items()})
        np.savez_compressed(path, **arrays)
        self.report.dumps.append(path)
        note = "%s: evidence in %s%s" % (self.what, path, "".join("; " + n for n in self.notes))
        self.report.failures.append(note) if not raised else print(note, flush=True)


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
    g_h = g_d.cpu().numpy()
    ev = Evidence(report, what, dict(upstream=g_h, ids=ids_h, bary=bary_h, clip=clip_h, normals=nrm, positions=pos, diffuse=kd, triangles=tris,
                                     light_positions=lp, light_intensities=li, ambient=amb, transforms=xf),
                  upstream=g_d, sign_codes=signs, sign_upstream=up, ids=ids, bary=bary, clip=clip, normals=nrm_d,
                  positions=pos_d, diffuse=kd_d, triangles=tris_d, light_positions=lp_d, light_intensities=li_d, ambient=amb_d,
                  transforms=xf_d, corner_records=records, adjacency_offsets=adjacency[0], adjacency_entries=adjacency[1],
                  rgba=rgba, target=target_d)
    raised = True
    try:
        for upstream, upstream_h, extra, name in ((g_d, g_h, {}, "dense"), (up, sign_g, {"l1_signs": signs}, "signs")):
            t = truth64.phong(ids_h, bary_h, tris, nrm, pos, kd, lp, li, amb, upstream_h)
            t["d_clip"], t["noise_clip"] = truth64.raster_pullback(clip_h, tris, ids_h, bary_h, t["dbary"], t["gabs"])
            t["d_vertices"], t["noise_vertices"] = truth64.whole_vertex_gradient(xf, t["d_positions"], t["d_clip"],
                                                                                 t["noise_positions"], t["noise_clip"])
            had_gradient = had_gradient or float(np.abs(t["d_vertices"]).max()) > 0
            ev.keep_truth(t, name + "/")
            if name == "dense":
                ev.forward_image(rgba, t["image"])
            variants = [(1, True, True, True, False)] + [(2, n, d, c, True) for n, d, c in
                                                        ((False, False, False), (False, False, True), (True, False, False), (True, True, True))]
            for which, want_n, want_d, want_clip, normalised in variants:
                _native.debug_set_shade_backward_kernel(which)
                out = _native.shade_backward(upstream, *tail, **kw, **extra, want_normal_grads=want_n, want_diffuse_grads=want_d,
                                             normalised_gbuffer=normalised, want_clip_grads=want_clip)
                kernel = _native.debug_last_accumulate_kernel().split("<")[0] + ("/" + name)
                tag = "%s %s normals=%s diffuse=%s clip=%s" % (what, name, want_n, want_d, want_clip)
                ev.keep("%s normals=%d diffuse=%d clip=%d" % (kernel, want_n, want_d, want_clip),
                        d_clip=out[0], d_normals=out[1], d_vertices=out[2], d_diffuse=out[3])
                if out[0] is not None:
                    report.check(kernel, "d clip", out[0], t["d_clip"], t["noise_clip"], tag)
                if out[1] is not None:
                    report.check(kernel, "d normals", out[1], t["d_normals"], t["noise_normals"], tag)
                report.check(kernel, "d vertices", out[2], t["d_vertices"], t["noise_vertices"], tag)
                if out[3] is not None:
                    report.check(kernel, "d diffuse", out[3], t["d_diffuse"], t["noise_diffuse"], tag)
        raised = False
    finally:
        _native.debug_set_shade_backward_kernel(0)
        ev.close(raised)
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
    g_h = g_d.cpu().numpy()
    t = truth64.phong(ids_h, bary_h, tris, nrm, pos, kd, lp, li, amb, g_h, specular=ks, shininess=shin, camera_position=cam)
    t["d_clip"], t["noise_clip"] = truth64.raster_pullback(clip_h, tris, ids_h, bary_h, t["dbary"], t["gabs"])
    t["d_vertices"], t["noise_vertices"] = truth64.whole_vertex_gradient(xf, t["d_positions"], t["d_clip"],
                                                                         t["noise_positions"], t["noise_clip"])
    report.trials += 1
    report.with_gradients += int(float(np.abs(t["d_vertices"]).max()) > 0)
    sargs = (g_d, ids, bary, clip, nrm_d, pos_d, kd_d, ks_d, tris_d, lp_d, li_d, amb_d, cam_d, shin_d, norms2)
    ev = Evidence(report, what, dict(upstream=g_h, ids=ids_h, bary=bary_h, clip=clip_h, normals=nrm, positions=pos, diffuse=kd,
                                     specular=ks, triangles=tris, light_positions=lp, light_intensities=li, ambient=amb,
                                     camera=cam, shininess=shin, transforms=xf),
                  upstream=g_d, ids=ids, bary=bary, clip=clip, normals=nrm_d, positions=pos_d, diffuse=kd_d, specular=ks_d, triangles=tris_d, light_positions=lp_d, light_intensities=li_d, ambient=amb_d, camera=cam_d,
                  shininess=shin_d, norms2=norms2, norms2_fused_pass=norms_f, transforms=xf_d, adjacency_offsets=adjacency[0],
                  adjacency_entries=adjacency[1], rgba=rgba)
    ev.keep_truth(t)
    ev.forward_image(rgba, t["image"])
    raised = True
    try:
        _specular_backwards(report, what, ev, t, sargs, adjacency, xf_d, rgba, lp.shape[1], W, H)
        raised = False
    finally:
        ev.close(raised)


def _specular_backwards(report, what, ev, t, sargs, adjacency, xf_d, rgba, n_lights, W, H):
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
        k_folded.startswith("SpecCoupledLaneFn" if n_lights <= 2 else "SpecFoldLaneFn"), (k_rows, k_lanes, k_folded)
    ev.keep(k_rows, d_clip=rows[0], d_normals=rows[1], d_positions=rows[2], d_diffuse=rows[3], d_specular=rows[4])
    ev.keep(k_lanes, d_clip=lanes[0], d_positions=lanes[2])
    ev.keep(k_folded, d_vertices=folded[2])
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
    ev.keep("l1 route", target=target, sign_codes=signs, dense_upstream=dense, from_dense=from_dense, from_signs=from_signs)
    if n_lights <= 2:
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
    ev = Evidence(report, what, dict(upstream=dout, ids=ids_h, bary=bary_h, attributes=attrs, triangles=tris),
                  upstream=dout_d, ids=ids, bary=bary, clip=clip, attributes=attrs_d, triangles=tris_d,
                  background=bg_d, corner_records=records, adjacency_offsets=adjacency[0], adjacency_entries=adjacency[1])
    ev.keep_truth(dict(t, d_clip=d_clip, noise_clip=noise_clip))
    raised = True
    try:
        rows = _native.interpolate_raster_backward(dout_d, ids, bary, clip, attrs_d, tris_d, bg_d, adjacency, corner_records=records)
        k_rows = _native.debug_last_accumulate_kernel().split("<")[0]
        lanes = _native.interpolate_raster_backward(dout_d, ids, bary, clip, attrs_d, tris_d, bg_d, adjacency, corner_records=records,
                                                    normalised_gbuffer=True)
        k_lanes = _native.debug_last_accumulate_kernel().split("<")[0]
        assert k_rows != k_lanes and k_lanes.startswith("AttrFoldLaneFn"), (k_rows, k_lanes)
        for kernel, out in ((k_rows, rows), (k_lanes, lanes)):
            ev.keep(kernel, d_attributes=out[0], d_clip=out[1])
            report.check(kernel, "d attributes", out[0], t["d_attributes"], t["noise_attributes"], what)
            report.check(kernel, "d clip", out[1], d_clip, noise_clip, what)
        raised = False
    finally:
        ev.close(raised)


def run(trial_fn, trials, seed, small=False, progress=None):
    """`trials` trials from one seeded generator.  No trial is ever repeated: the kernels are deterministic up to the
    order of their float atomics, so a value beyond the bound is a failure, and it leaves its evidence behind
    (Evidence: module docstring).  The generator's state at the start of each trial is in Report.rng_states, so a
    single trial can be regenerated without running the ones before it."""
    rng = np.random.default_rng(seed)
    report = Report()
    report.rng_states = []
    for trial in range(trials):
        report.rng_states.append(rng.bit_generator.state)
        before = len(report.failures)
        trial_fn(rng, trial, report, small=small)
        if len(report.failures) > before:
            report.failures.append("trial %d of seed %d starts from generator state %r" % (trial, seed, report.rng_states[-1]))
        if progress and trial % progress == progress - 1:
            print("%d trials, %d failures; worst excess per kernel: %s" % (trial + 1, len(report.failures), report.summary()), flush=True)
    return report
