"""Fuzz of the fused shading backward on random triangle soups (GPU): every lane-accumulating variant
(k_accumulate_lanes: folded / difference-basis / general, dense and sign-coded upstream) against the rows kernel
(k_accumulate_rows, forced through the debug hook) on the same inputs.  The rows kernel itself is pinned to the
oracle by tests/test_render_gpu.py; this looks for races and corner cases of the lane kernels' run bookkeeping
(one-pixel runs, merge-table overflow, strips that end mid-run, ragged image edges).

    python tests/fuzz_shade_backward_gpu.py [--trials N] [--seed S]
"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from pytorch_mesh_renderer_amd import _native

ap = argparse.ArgumentParser()
ap.add_argument("--trials", type=int, default=120)
ap.add_argument("--seed", type=int, default=1)
args = ap.parse_args()
rng = np.random.default_rng(args.seed)
dev = torch.device("cuda:0")
def truth_f64(pos, xf, tris, nrm, kd, lp, li, amb, ids, bary, g, W, H):
    """The whole vertex gradient in float64 by torch autograd on the CPU, the G-buffer's ids held fixed: barycentrics
    as the perspective-correct function of the clip-space corners at the pixel centre (what rasterize_triangles.cpp
    :202-269 differentiates by hand), attributes, Phong, sum(g * rgb)."""
    P = torch.tensor(pos, dtype=torch.float64, requires_grad=True)
    M = torch.tensor(xf, dtype=torch.float64)
    Bn = P.shape[0]
    clip = torch.einsum("bij,bvj->bvi", M[:, :, :3], P) + M[:, None, :, 3]
    tri = torch.tensor(tris, dtype=torch.long)
    covered = torch.tensor(bary.sum(-1) > 0.5)
    total = torch.zeros((), dtype=torch.float64)
    ys, xs = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
    px = ((xs.double() + 0.5) / (0.5 * W) - 1.0)
    py = ((ys.double() + 0.5) / (0.5 * H) - 1.0)
    for b in range(Bn):
        m = covered[b]
        if not bool(m.any()):
            continue
        t = torch.tensor(ids[b])[m].long()
        c = clip[b][tri[t]]                                  # [n, 3 corners, 4]
        x, y, w = c[..., 0], c[..., 1], c[..., 3]
        # adjugate rows (edge functions) of [[x0 x1 x2], [y0 y1 y2], [w0 w1 w2]]
        e0 = (y[:, 1] * w[:, 2] - w[:, 1] * y[:, 2]) * px[m] + (x[:, 2] * w[:, 1] - w[:, 2] * x[:, 1]) * py[m] + (x[:, 1] * y[:, 2] - y[:, 1] * x[:, 2])
        e1 = (y[:, 2] * w[:, 0] - w[:, 2] * y[:, 0]) * px[m] + (x[:, 0] * w[:, 2] - w[:, 0] * x[:, 2]) * py[m] + (x[:, 2] * y[:, 0] - y[:, 2] * x[:, 0])
        e2 = (y[:, 0] * w[:, 1] - w[:, 0] * y[:, 1]) * px[m] + (x[:, 1] * w[:, 0] - w[:, 1] * x[:, 0]) * py[m] + (x[:, 0] * y[:, 1] - y[:, 0] * x[:, 1])
        ssum = e0 + e1 + e2
        bb = torch.stack([e0 / ssum, e1 / ssum, e2 / ssum], 1)            # [n, 3]
        def interp(a):
            return (torch.tensor(a[b], dtype=torch.float64)[tri[t]] * bb[..., None]).sum(1)
        N = interp(nrm)
        Pw = (P[b][tri[t]] * bb[..., None]).sum(1)
        Kd = interp(kd)
        N = N / N.norm(dim=1, keepdim=True).clamp_min(1e-12)
        rgb = torch.zeros_like(Kd)
        if amb is not None:
            rgb = rgb + torch.tensor(amb[b], dtype=torch.float64) * Kd
        for l in range(lp.shape[1]):
            D = torch.tensor(lp[b, l], dtype=torch.float64) - Pw
            D = D / D.norm(dim=1, keepdim=True).clamp_min(1e-12)
            ndl = (N * D).sum(1).clamp(0.0, 1.0)
            rgb = rgb + Kd * ndl[:, None] * torch.tensor(li[b, l], dtype=torch.float64)
        mask = (Kd >= 0).any(1)
        gg = torch.tensor(g[b], dtype=torch.float64).flip(0)[m][:, :3]     # image rows are flipped w.r.t. the G-buffer
        total = total + (gg * rgb * mask[:, None]).sum()
    total.backward()
    return P.grad.numpy()


bad = 0
worst_rel = 0.0
nontrivial = 0   # trials with a non-zero vertex gradient (a soup may miss the image)
t0 = time.time()
for trial in range(args.trials):
    kind = trial % 4
    B = int(rng.integers(1, 4))
    V = int(rng.integers(4, 300))
    T = int(rng.integers(1, 2500 if kind == 3 else 400))
    W, H = int(rng.integers(8, 420)), int(rng.integers(8, 300))
    L = int(rng.integers(1, 5))
    pos = (rng.normal(size=(B, V, 3)) * [1.0, 1.0, 0.3]).astype(np.float32)
    if kind == 1:
        pos *= 0.2                                            # tiny triangles: one-pixel runs
    xf = np.tile(np.eye(4, dtype=np.float32), (B, 1, 1))
    xf[:, 3, 2] = 0.5                                         # w = 1 + 0.5 z: mild perspective, everything in front
    xf[:, 3, 3] = 1.2
    tris = rng.integers(0, V, size=(T, 3)).astype(np.int32)
    nrm = rng.normal(size=(B, V, 3)).astype(np.float32)
    kd = rng.random(size=(B, V, 3)).astype(np.float32)
    lp = (rng.normal(size=(B, L, 3)) * 3.0).astype(np.float32)
    li = (rng.random(size=(B, L, 3)) + 0.1).astype(np.float32)
    amb = (rng.random(size=(B, 3)) * 0.3).astype(np.float32) if trial % 3 == 0 else None
    t = lambda a: torch.from_numpy(a).to(dev) if a is not None else None
    pos_d, xf_d, tris_d, nrm_d, kd_d, lp_d, li_d, amb_d = map(t, (pos, xf, tris, nrm, kd, lp, li, amb))
    clip, ids, bary, _, rgba, records = _native.render_forward(pos_d, xf_d, nrm_d, kd_d, tris_d, lp_d, li_d, amb_d, W, H, want_z=False)
    adjacency = _native.vertex_adjacency(tris_d, V)
    g = torch.from_numpy(rng.normal(size=(B, H, W, 4)).astype(np.float32)).to(dev) / (H * W)
    _, signs = _native.l1_loss_forward(rgba, torch.from_numpy(rng.random(size=(B, H, W, 4)).astype(np.float32)).to(dev))
    up = torch.full((1,), 0.9, device=dev)
    tail = (ids, bary, clip, nrm_d, pos_d, kd_d, tris_d, lp_d, li_d, amb_d)
    kw = dict(corner_records=records, adjacency=adjacency, transforms=xf_d, want_light_grads=False)
    try:
        for upstream, extra, name in ((g, {}, "dense"), (up, {"l1_signs": signs}, "signs")):
            _native.debug_set_shade_backward_kernel(1)
            full = _native.shade_backward(upstream, *tail, **kw, **extra)
            nontrivial += int(name == "dense" and float(full[2].abs().max()) > 0)
            for want_n, want_d, want_clip in ((False, False, False), (False, False, True), (True, False, False), (True, True, True)):
                _native.debug_set_shade_backward_kernel(2)
                lean = _native.shade_backward(upstream, *tail, **kw, **extra, want_normal_grads=want_n, want_diffuse_grads=want_d,
                                              normalised_gbuffer=True, want_clip_grads=want_clip)
                for k in (0, 1, 2, 3):
                    if lean[k] is None:
                        continue
                    want = full[k]
                    scale = max(float(want.abs().max()), 1e-30)
                    # Sliver triangles make both kernels noisy in different ways (difference basis vs three-term brackets:
                    # against a float64 evaluation either can be the worse one, by up to ~2e-3 of the largest gradient --
                    # MR_FUZZ_TRUTH=1 prints both); a race or a bookkeeping bug shows as errors of order one.
                    worst_rel = max(worst_rel, float((lean[k] - want).abs().max()) / scale)
                    if not bool(torch.isclose(lean[k], want, rtol=1e-3, atol=(3e-6 if os.environ.get("MR_FUZZ_TRUTH") else 5e-3) * scale).all()):
                        bad += 1
                        worst = float((lean[k] - want).abs().max())
                        note = ""
                        if k == 2 and name == "dense" and not want_clip and os.environ.get("MR_FUZZ_TRUTH"):
                            truth = truth_f64(pos, xf, tris, nrm, kd, lp, li, amb, ids.cpu().numpy(), bary.cpu().numpy(),
                                              g.cpu().numpy(), W, H)
                            note = " | vs float64: lanes %.3e rows %.3e" % (np.abs(lean[k].cpu().numpy() - truth).max(),
                                                                            np.abs(want.cpu().numpy() - truth).max())
                        print(f"MISMATCH trial {trial} kind {kind} B={B} V={V} T={T} {W}x{H} L={L} {name} normals={want_n} "
                              f"diffuse={want_d} clip={want_clip} output {k}: max |diff| {worst:.3e} of {scale:.3e}{note}", flush=True)
    finally:
        _native.debug_set_shade_backward_kernel(0)
    if trial % 20 == 19:
        print(f"{trial + 1} trials, {bad} mismatches, {time.time() - t0:.0f} s", flush=True)
print("FUZZ", "FAILED" if bad else "OK", f"{args.trials} trials ({nontrivial} with gradients), {bad} mismatches; largest "
      f"lanes-vs-rows deviation {worst_rel:.2e} of the output's largest element")
sys.exit(1 if bad or nontrivial < args.trials // 2 else 0)
