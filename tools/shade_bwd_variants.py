"""mr_shade_backward at configs[2] (5k tris, 1024^2, B=32) by what the caller wants: which attribute gradients,
light gradients or not, dense or sign-coded upstream, rows kernel (1) vs lane-accumulating kernel (2).
    python tools/shade_bwd_variants.py [--iters N]
(the 36-sum lane kernel is part of every build)"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pytorch_mesh_renderer_amd import _native
from pytorch_mesh_renderer_amd.common import synthetic

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=20)
args = ap.parse_args()
B, W, H, K = 32, 1024, 1024, 50
dev = torch.device("cuda:0")
job = synthetic.sphere_job(B, W, H, K)
d = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in job.items()}
xf = synthetic.clip_transforms(job["eyes"], W, H).to(dev)
clip, ids, bary, _, rgba, records = _native.render_forward(
    d["vertices"], xf, d["normals"], d["diffuse"], d["triangles"], d["light_positions"], d["light_intensities"],
    None, W, H, want_z=False)
adjacency = _native.vertex_adjacency(d["triangles"], d["vertices"].shape[1])
g = torch.randn_like(rgba) / (H * W)
_, signs = _native.l1_loss_forward(rgba, torch.zeros_like(rgba))
up = torch.ones(1, device=dev)
tail = (ids, bary, clip, d["normals"], d["vertices"], d["diffuse"], d["triangles"], d["light_positions"],
        d["light_intensities"], None)
for upstream_name, upstream, extra in (("signs", up, {"l1_signs": signs}), ("dense", g, {})):
    for want_n, want_d in ((False, False), (True, False), (True, True)):
        for lights in (False, True):
            # rows kernel | lane kernel, general G-buffer | lane kernel, normalised G-buffer (difference-basis records,
            # round 4) | the same without the clip-space gradient (pull-back folded into the pixel pass)
            for kernel, normalised, want_clip in ((1, False, True), (2, False, True), (2, True, True), (2, True, False)):
                _native.debug_set_shade_backward_kernel(kernel)
                def fn():
                    return _native.shade_backward(upstream, *tail, corner_records=records, adjacency=adjacency,
                                                  transforms=xf, want_light_grads=lights, want_normal_grads=want_n,
                                                  want_diffuse_grads=want_d, normalised_gbuffer=normalised,
                                                  want_clip_grads=want_clip, **extra)
                for _ in range(3): fn()
                torch.cuda.synchronize(); t0 = time.perf_counter()
                for _ in range(args.iters): fn()
                torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / args.iters
                print("upstream=%s grads=%s lights=%d kernel=%s: %.3f ms" % (
                    upstream_name, "P" + ("N" if want_n else "") + ("K" if want_d else ""), lights,
                    "rows" if kernel == 1 else "lanes" + (" normalised" if normalised else "") + ("" if want_clip else " no-dclip"),
                    dt * 1e3), flush=True)
_native.debug_set_shade_backward_kernel(0)
