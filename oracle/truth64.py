"""Float64 evaluation of the reference's gradient formulas on a FIXED G-buffer (TEST INFRASTRUCTURE ONLY).

What the specialised backward kernels are held against when "the other HIP kernel" is not good enough
(sliver triangles: every binary32 evaluation is noisy there, and two of them differ by more than either
differs from the truth).  Inputs are exactly what a backward kernel gets -- the clip-space vertices as
float32 bits, the STORED float32 barycentrics and ids of the forward pass (the reference differentiates
through the stored values too, rasterize_triangles.cpp:159-161), the upstream gradient -- and everything
downstream of them is evaluated in binary64:

  raster_pullback   rasterize_triangles.cpp:131-273:
                    d b_i / d M_kj = -Minv_ik b_j + (sum_l Minv_lk) b_i b_j, skip rule of :162
  interpolate       src/mesh_renderer/rasterize.py:130-150 on covered pixels (alpha = 1 there)
  phong             src/mesh_renderer/render.py:201-215, 287-386 (autograd in float64, as the reference
                    runs autograd in float32), the background's attributes = -1 (render.py:197)

Next to every gradient comes a NOISE SCALE: the sum of the absolute values of the terms the gradient is
the sum of IN THE REFERENCE'S OWN EVALUATION ORDER (cofactors and determinant as differences of products,
normalisations as projections, the brackets of rasterize_triangles.cpp:202-230, the sum over pixels).  A
binary32 evaluation of a sum of n products is within ~n * 2^-24 of it times that scale whatever its
association order -- the reference's own included; `assert_within_rounding` states the bound the tests use.

Pinned by tests/test_oracle.py: raster_pullback against oracle/mr_oracle.c (itself bit-for-bit the
reference), phong against oracle/shading.py on the reference's golden scenes.
"""
import numpy as np
import torch

U32 = 2.0 ** -24


def covered_mask(ids, bary):
    """Pixels the reference's backward processes (rasterize_triangles.cpp:162)."""
    return (np.asarray(ids) != 0) | (np.asarray(bary, dtype=np.float64).sum(-1) >= 0.9)


def raster_pullback(clip, tris, ids, bary, dbary, gabs=None):
    """-> (dclip [B,V,4] float64, noise [B,V,4] float64).

    clip [B,V,4], ids [B,H,W], bary [B,H,W,3]: the forward's inputs / outputs (float32 bits);
    dbary [B,H,W,3] float64: dL / d barycentrics; gabs [B,H,W] (optional): a magnitude for dL/db's
    entries in the noise scale (default: max_i |dbary_i|) -- pass the larger sum of |products| when
    dbary itself comes out of a cancellation."""
    clip = np.asarray(clip, dtype=np.float64)
    tris = np.asarray(tris).astype(np.int64)
    B, V = clip.shape[:2]
    dclip = np.zeros((B, V, 4))
    noise = np.zeros((B, V, 4))
    cov = covered_mask(ids, bary)
    comp = (0, 1, 3)
    for b in range(B):
        m = cov[b]
        if not m.any():
            continue
        tri = tris[np.asarray(ids[b])[m].astype(np.int64)]          # [n,3]
        c = clip[b][tri]                                            # [n, corner j, 4]
        M = np.stack([c[..., 0], c[..., 1], c[..., 3]], 1)          # [n, k, j]
        # sign(det) adj(M) / |det| = inverse (rasterize_triangles.cpp:61-87,180-185)
        adj = np.empty_like(M)                                      # adj[n, i, k]
        adj_abs = np.empty_like(M)                                  # |p q| + |r s| of each cofactor p q - r s
        for i in range(3):
            for k in range(3):
                r = [x for x in range(3) if x != k]
                q = [x for x in range(3) if x != i]
                pq, rs = M[:, r[0], q[0]] * M[:, r[1], q[1]], M[:, r[0], q[1]] * M[:, r[1], q[0]]
                adj[:, i, k] = (-1.0) ** (i + k) * (pq - rs)
                adj_abs[:, i, k] = np.abs(pq) + np.abs(rs)
        det = (M[:, 0, :] * adj[:, :, 0]).sum(1)
        det_abs = (np.abs(M[:, 0, :]) * adj_abs[:, :, 0]).sum(1)
        with np.errstate(divide="ignore", invalid="ignore"):
            minv = adj / det[:, None, None]
            # The reference forms the cofactors and the determinant in binary32 (rasterize_triangles.cpp:61-87): each is
            # a difference of products, so an entry of the inverse carries the rounding of its own cofactor plus that
            # of the determinant -- on a sliver (|det| << the products it is made of) far more than 2^-24 of itself.
            minv_abs = adj_abs / np.abs(det)[:, None, None] + np.abs(minv) * (det_abs / np.abs(det))[:, None, None]
        bb = np.asarray(bary[b], dtype=np.float64)[m]               # [n,3]
        g = np.asarray(dbary[b], dtype=np.float64)[m]
        ga = np.abs(g).max(1) if gabs is None else np.asarray(gabs[b], dtype=np.float64)[m]
        S = minv.sum(1)                                             # [n,k]
        bracket = -(g[:, :, None] * minv).sum(1) + S * (g * bb).sum(1)[:, None]   # [n,k]
        nb = 2.0 * ga[:, None] * minv_abs.sum(1)                    # [n,k]
        for j in range(3):
            for k in range(3):
                np.add.at(dclip[b, :, comp[k]], tri[:, j], bb[:, j] * bracket[:, k])
                np.add.at(noise[b, :, comp[k]], tri[:, j], np.abs(bb[:, j]) * nb[:, k])
    return dclip, noise


def interpolate(ids, bary, tris, attrs, dout):
    """rasterize()'s attribute image on covered pixels, out = sum_i b_i attr_i (alpha = 1 exactly there and
    nothing flows through it; uncovered pixels show the background) against upstream dout [B,H,W,A]
    (G-buffer orientation) -> dict(d_attributes [B,V,A], noise_attributes, dbary [B,H,W,3], gabs [B,H,W])."""
    attrs = np.asarray(attrs, dtype=np.float64)
    dout = np.asarray(dout, dtype=np.float64)
    tris = np.asarray(tris).astype(np.int64)
    B, V, A = attrs.shape
    H, W = ids.shape[1:]
    d_attr, n_attr = np.zeros((B, V, A)), np.zeros((B, V, A))
    dbary, gabs = np.zeros((B, H, W, 3)), np.zeros((B, H, W))
    cov = covered_mask(ids, bary)
    for b in range(B):
        m = cov[b]
        if not m.any():
            continue
        tri = tris[np.asarray(ids[b])[m].astype(np.int64)]
        bb = np.asarray(bary[b], dtype=np.float64)[m]
        g = dout[b][m]                                               # [n,A]
        corner = attrs[b][tri]                                       # [n,3,A]
        dbary[b][m] = (corner * g[:, None, :]).sum(2)
        gabs[b][m] = (np.abs(corner) * np.abs(g)[:, None, :]).sum(2).max(1)
        for j in range(3):
            np.add.at(d_attr[b], tri[:, j], bb[:, j, None] * g)
            np.add.at(n_attr[b], tri[:, j], np.abs(bb[:, j, None] * g))
    return {"d_attributes": d_attr, "noise_attributes": n_attr, "dbary": dbary, "gabs": gabs}


def _gather_pixels(ids, bary, tris, leaves, groups, per_vertex, bb):
    """Interpolated attribute image px [B,H,W,A] (background -1, render.py:197) + its corner tensor and index."""
    tris_t = torch.tensor(np.asarray(tris).astype(np.int64))
    B = ids.shape[0]
    cov = torch.tensor(covered_mask(ids, bary))
    idx = tris_t[torch.tensor(np.asarray(ids).astype(np.int64))]     # [B,H,W,3]
    pieces = [leaves[k] for k in groups]
    if per_vertex:
        pieces.append(leaves["shininess"].unsqueeze(2))
    attrs = torch.cat(pieces, 2)                                     # [B,V,A]
    bi = torch.arange(B).reshape(B, 1, 1, 1)
    corner = attrs[bi, idx]                                          # [B,H,W,3,A]
    px = (corner * bb.unsqueeze(-1)).sum(3)
    px = torch.where(cov.unsqueeze(-1), px, torch.full_like(px, -1.0))
    return px, corner, idx, bi, cov, attrs


def _shade(px, leaves, has_ambient, has_specular, per_vertex):
    """render.py:201-215,287-386 on the attribute image -> (rgba [B,H,W,4] in G-buffer orientation, aux) where aux
    holds the quantities whose clamps / comparisons make the image non-smooth: ndl_raw [B,L,P], rv_unit [B,L,P]
    (None without specular), kd_max [B,P]."""
    B, H, W = px.shape[:3]
    P = H * W
    n = torch.nn.functional.normalize(px[..., 0:3], p=2, dim=3, eps=1e-12).reshape(B, P, 3)
    kept = {"n": n, "n_len": px[..., 0:3].detach().norm(dim=3).reshape(B, P)}
    pos = px[..., 3:6].reshape(B, P, 3)
    kd = px[..., 6:9].reshape(B, P, 3)
    mask = (px[..., 6:9] >= 0.0).any(dim=3)
    rgb = torch.zeros(B, P, 3, dtype=torch.float64)
    if has_ambient:
        rgb = rgb + leaves["ambient"].unsqueeze(1) * kd
    lp, li = leaves["light_positions"], leaves["light_intensities"]
    to_light = torch.nn.functional.normalize(lp.unsqueeze(2) - pos.unsqueeze(1), p=2, dim=3, eps=1e-12)
    kept["to_light"], kept["to_light_len"] = to_light, (lp.unsqueeze(2) - pos.unsqueeze(1)).detach().norm(dim=3)
    ndl_raw = (n.unsqueeze(1) * to_light).sum(3)
    ndl = torch.clamp(ndl_raw, 0.0, 1.0)
    rgb = rgb + (kd.unsqueeze(1) * ndl.unsqueeze(3) * li.unsqueeze(2)).sum(1)
    rv_unit = None
    if has_specular:
        ks = px[..., 9:12].reshape(B, P, 3)
        shine = px[..., 12].reshape(B, 1, P) if per_vertex else leaves["shininess"].reshape(B, 1, 1)
        mirror = torch.nn.functional.normalize(2.0 * ndl.unsqueeze(3) * n.unsqueeze(1) - to_light, p=2, dim=3, eps=1e-12)
        to_cam = torch.nn.functional.normalize(leaves["camera"].reshape(B, 1, 3) - pos, p=2, dim=2, eps=1e-12)
        kept["to_cam"], kept["to_cam_len"] = to_cam, (leaves["camera"].reshape(B, 1, 3) - pos).detach().norm(dim=2)
        rv = (mirror * to_cam.unsqueeze(1)).sum(3)                   # [B,L,P]
        rv_unit = torch.nn.functional.normalize(rv, p=2, dim=2, eps=1e-12)   # over ALL pixels (render.py:356)
        rv = torch.clamp(rv_unit, 0.0, 1.0)
        rv = torch.where(ndl != 0.0, rv, torch.zeros_like(rv))
        # the power only where the mask is on (oracle/shading.py: power_inside_mask_only -- the kernels' semantics)
        inside = mask.reshape(B, 1, P)
        rv = torch.where(inside, rv, torch.zeros_like(rv))
        expo = torch.where(inside, shine.expand_as(rv), torch.ones_like(rv))
        spec = torch.pow(rv, expo).unsqueeze(3)
        rgb = rgb + (ks.unsqueeze(1) * spec * li.unsqueeze(2)).sum(1)
        # conditioning of the power: x^y evaluated as exp2(y log2 x) carries the rounding of log2 x and of the product
        # into the EXPONENT -- a relative error of ~|y log2 x| roundings in the result (a base of 1e-4 to the fourth
        # power: ~50 of them, on a value of 1e-16) -- next to y roundings inherited from the base
        with torch.no_grad():
            base = torch.clamp(rv, min=1e-300)
            cond = 1.0 + expo.abs() * (1.0 + torch.log2(base).abs())
            kept["pow_cond"] = torch.where(rv > 0.0, cond, torch.ones_like(cond)).max(1).values      # [B,P]
    rgb = rgb.reshape(B, H, W, 3)
    alpha = mask.reshape(B, H, W, 1).to(torch.float64)
    rgb = torch.where(alpha > 0.5, rgb, torch.zeros_like(rgb))
    return torch.cat([rgb, alpha], 3), {"ndl_raw": ndl_raw, "rv_unit": rv_unit, "kd_max": kd.max(2).values, "kept": kept}


def _leaves(normals, positions, diffuse, light_positions, light_intensities, ambient, specular, shininess, camera_position):
    f64 = lambda a: None if a is None else torch.tensor(np.asarray(a), dtype=torch.float64)
    leaves = {"normals": f64(normals), "positions": f64(positions), "diffuse": f64(diffuse),
              "light_positions": f64(light_positions), "light_intensities": f64(light_intensities)}
    if ambient is not None:
        leaves["ambient"] = f64(ambient)
    per_vertex = False
    if specular is not None:
        leaves["specular"] = f64(specular)
        leaves["shininess"] = f64(shininess)
        leaves["camera"] = f64(camera_position)
        per_vertex = leaves["shininess"].dim() == 2
    groups = ["normals", "positions", "diffuse"] + (["specular"] if specular is not None else [])
    return leaves, groups, per_vertex


def borderline_pixels(ids, bary, tris, normals, positions, diffuse, light_positions, light_intensities, ambient,
                      specular=None, shininess=None, camera_position=None, tol=1e-4, flipped=True):
    """[B,H,W] bool (image orientation when `flipped`): covered pixels within `tol` of one of the shading's kinks --
    a light's n.l at the ends of its clamp (render.py:310), the specular base at 0 (render.py:357-360), the largest
    diffuse component at the mask's threshold (render.py:215).  There a binary32 and a binary64 evaluation may take
    different branches and differ by the pixel's whole contribution: a fuzz test switches such pixels off (zero
    upstream gradient) on both sides instead of comparing the two branches."""
    leaves, groups, per_vertex = _leaves(normals, positions, diffuse, light_positions, light_intensities, ambient,
                                         specular, shininess, camera_position)
    with torch.no_grad():
        bb = torch.tensor(np.asarray(bary), dtype=torch.float64)
        px, _, _, _, cov, _ = _gather_pixels(ids, bary, tris, leaves, groups, per_vertex, bb)
        _, aux = _shade(px, leaves, ambient is not None, specular is not None, per_vertex)
    B, H, W = ids.shape
    near = ((aux["ndl_raw"].abs() < tol) | ((aux["ndl_raw"] - 1.0).abs() < tol)).any(1)
    if aux["rv_unit"] is not None:
        near = near | (aux["rv_unit"].abs() < tol).any(1)
    near = near | (aux["kd_max"].abs() < tol)
    near = near.reshape(B, H, W) & cov
    return (near.flip(1) if flipped else near).numpy()


def phong(ids, bary, tris, normals, positions, diffuse, light_positions, light_intensities, ambient, drgba,
          specular=None, shininess=None, camera_position=None, flipped=True):
    """render()'s shading (render.py:201-215,287-386) on a fixed G-buffer, differentiated by float64 autograd
    against the upstream drgba [B,H,W,4] (image orientation: rows flipped w.r.t. the G-buffer when `flipped`,
    render.py:384-386).  Arrays in, dict of float64 arrays out:
      d_normals, d_positions (the attribute path only), d_diffuse [B,V,3], d_specular, d_shininess,
      d_light_positions, d_light_intensities [B,L,3], d_ambient [B,3], d_camera [B,3], image [B,H,W,4],
      dbary [B,H,W,3], gabs [B,H,W] (for raster_pullback), noise_normals / _positions / _diffuse [B,V,3]."""
    B, H, W = ids.shape
    leaves, groups, per_vertex = _leaves(normals, positions, diffuse, light_positions, light_intensities, ambient,
                                         specular, shininess, camera_position)
    for t in leaves.values():
        t.requires_grad_(True)
    bb = torch.tensor(np.asarray(bary), dtype=torch.float64).requires_grad_(True)
    g = torch.tensor(np.asarray(drgba), dtype=torch.float64)
    if flipped:
        g = g.flip(1)
    px, corner, idx, bi, cov, attrs = _gather_pixels(ids, bary, tris, leaves, groups, per_vertex, bb)
    px.retain_grad()
    rgba, aux = _shade(px, leaves, ambient is not None, specular is not None, per_vertex)
    kept = aux["kept"]
    for k in ("n", "to_light", "to_cam"):
        if k in kept and kept[k].requires_grad:
            kept[k].retain_grad()
    (rgba * g).sum().backward()
    out = {"image": (rgba.flip(1) if flipped else rgba).detach().numpy(), "dbary": bb.grad.numpy()}
    # |d loss / d attribute image| for the noise scales -- where an attribute goes through a normalisation
    # (x / |x|: the gradient is a projection, i.e. a cancellation) the magnitude BEFORE the cancellation:
    # |d loss / d unit vector|_1 / |x|
    dat = px.grad.abs().clone()                                      # [B,H,W,A]
    tiny = 1e-300
    # ... times the conditioning of x itself where x is a cancelling sum (corner normals that nearly cancel, a light
    # next to the surface): the rounding of x, ~2^-24 of the sum of |b_k c_k|, moves the projection by that over |x|
    mag = (corner.detach().abs() * bb.detach().abs().unsqueeze(-1)).sum(3)            # [B,H,W,A]: sum_k |b_k c_k|
    if kept["n"].grad is not None:
        cond = torch.clamp(mag[..., 0:3].sum(3).reshape(B, -1) / (kept["n_len"] + tiny), min=1.0)
        dn = (kept["n"].grad.abs().sum(2) / (kept["n_len"] + tiny) * cond).reshape(B, H, W, 1)
        dat[..., 0:3] = torch.maximum(dat[..., 0:3], dn)
    dp = torch.zeros(B, H * W, dtype=torch.float64)
    pmag = mag[..., 3:6].sum(3).reshape(B, 1, -1)
    if kept["to_light"].grad is not None:
        cond = torch.clamp((pmag + leaves["light_positions"].detach().abs().sum(2).unsqueeze(2)) / (kept["to_light_len"] + tiny), min=1.0)
        dp = dp + (kept["to_light"].grad.abs().sum(3) / (kept["to_light_len"] + tiny) * cond).sum(1)
    if "to_cam" in kept and kept["to_cam"].grad is not None:
        cond = torch.clamp((pmag[:, 0] + leaves["camera"].detach().abs().sum(1).unsqueeze(1)) / (kept["to_cam_len"] + tiny), min=1.0)
        dp = dp + kept["to_cam"].grad.abs().sum(2) / (kept["to_cam_len"] + tiny) * cond
    dat[..., 3:6] = torch.maximum(dat[..., 3:6], dp.reshape(B, H, W, 1))
    if "pow_cond" in kept:   # the gradients that exist only through the power: specular colours, per-vertex exponents
        dat[..., 9:] = dat[..., 9:] * kept["pow_cond"].reshape(B, H, W, 1)
    out["gabs"] = (corner.detach().abs() * dat.unsqueeze(3)).sum(4).max(3).values.numpy()
    out["gabs"] = np.where(cov.numpy(), out["gabs"], 0.0)
    names = {"normals": "d_normals", "positions": "d_positions", "diffuse": "d_diffuse", "specular": "d_specular",
             "shininess": "d_shininess", "light_positions": "d_light_positions",
             "light_intensities": "d_light_intensities", "ambient": "d_ambient", "camera": "d_camera"}
    for k, t in leaves.items():
        out[names[k]] = t.grad.numpy() if t.grad is not None else np.zeros(t.shape)
    # noise of the attribute gradients: sum over pixels of |d attribute| b_j
    V = attrs.shape[1]
    w = (dat.unsqueeze(3) * bb.detach().abs().unsqueeze(-1)) * cov.reshape(B, H, W, 1, 1)   # [B,H,W,3,A]
    noise = torch.zeros(B, V, attrs.shape[2], dtype=torch.float64)
    noise.index_put_((bi.expand(B, H, W, 3).reshape(-1), idx.reshape(-1)), w.reshape(-1, attrs.shape[2]), accumulate=True)
    for gi, k in enumerate(groups):
        out["noise_" + k] = noise[..., 3 * gi:3 * gi + 3].numpy()
    return out


def whole_vertex_gradient(transforms, d_positions, dclip, noise_positions=None, noise_clip=None):
    """d / d world vertices of clip = M (v, 1): the attribute path plus the clip-space gradient pulled back
    (camera_utils.py:142-170's matmul, transposed) -> (gradient [B,V,3], noise [B,V,3] or None)."""
    M = np.asarray(transforms, dtype=np.float64)
    grad = d_positions + np.einsum("bkc,bvk->bvc", M[:, :, :3], dclip)
    if noise_clip is None:
        return grad, None
    return grad, noise_positions + np.einsum("bkc,bvk->bvc", np.abs(M[:, :, :3]), noise_clip)


def excess(got, truth, noise, k_rounding=64.0, floor=1e-7):
    """max over elements of |got - truth| / (k_rounding * 2^-24 * noise + floor * max|truth|): <= 1 means within
    the rounding model.  NaN / inf in `got` where the truth is finite count as infinite excess."""
    got = np.asarray(got, dtype=np.float64)
    allowed = k_rounding * U32 * noise + floor * max(float(np.abs(truth).max()), 1e-300)
    err = np.abs(got - truth)
    err = np.where(np.isfinite(got) | ~np.isfinite(truth), err, np.inf)
    ok = np.isfinite(truth) & np.isfinite(allowed)
    return float((err[ok] / allowed[ok]).max()) if ok.any() else 0.0


def assert_within_rounding(got, truth, noise, what, k_rounding=64.0, floor=1e-7):
    """|got - truth| <= k_rounding * 2^-24 * noise + floor * max|truth|, elementwise.  k_rounding = 64: the
    longest chain of these kernels (stored barycentrics -> attributes -> Phong -> three brackets -> outer
    product -> sum over pixels) is about forty roundings deep."""
    e = excess(got, truth, noise, k_rounding, floor)
    assert e <= 1.0, "%s: %.2f times the rounding bound (max |diff| %.3e, max |truth| %.3e)" % (
        what, e, float(np.abs(np.asarray(got, dtype=np.float64) - truth).max()), float(np.abs(truth).max()))
