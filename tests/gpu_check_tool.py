"""Quick on-GPU parity + timing probe against the oracle (development aid; lives under tests/ because
only tests, smoke() and bench.py's cpu_baseline may use the oracle).  python tests/gpu_check_tool.py"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import oracle
from pytorch_mesh_renderer_amd import _native
from pytorch_mesh_renderer_amd.common import synthetic

dev = torch.device("cuda:0")

def check(name, clip, tris, W, H, time_it=False):
    clip_d, tris_d = clip.to(dev), tris.to(dev)
    ids, bary, z = _native.rasterize_forward(clip_d, tris_d, W, H)
    torch.cuda.synchronize()
    o_ids, o_bary, o_z = oracle.forward(clip.numpy(), tris.numpy(), W, H, threads=8)
    ok_ids = np.array_equal(ids.cpu().numpy(), o_ids)
    ok_z = np.array_equal(z.cpu().numpy().view(np.uint32), o_z.view(np.uint32))
    ok_b = np.array_equal(bary.cpu().numpy().view(np.uint32), o_bary.view(np.uint32))
    g = torch.Generator().manual_seed(0)
    dbary = torch.randn(bary.shape, generator=g) / (H * W)
    dclip = _native.rasterize_backward(dbary.to(dev), clip_d, tris_d, ids, bary)
    o_dclip = oracle.backward(dbary.numpy(), clip.numpy(), tris.numpy(), o_ids, o_bary, threads=8)
    err = np.abs(dclip.cpu().numpy() - o_dclip).max()
    print(f"{name}: ids={ok_ids} z={ok_z} bary={ok_b} mismatched_ids={(ids.cpu().numpy()!=o_ids).sum()} "
          f"bwd_maxerr={err:.3e} |grad|max={np.abs(o_dclip).max():.3e}", flush=True)
    if time_it:
        for shape in (0,):
            _native.lib().mr_debug_set_raster_probe(shape)
            for _ in range(3): _native.rasterize_forward(clip_d, tris_d, W, H)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            n = 10
            for _ in range(n): _native.rasterize_forward(clip_d, tris_d, W, H)
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
            B = clip.shape[0]
            print(f"   fwd tile_shape={shape}: {dt*1e3:.3f} ms  -> {B*H*W*20/dt/1e9:.1f} GB/s G-buffer, {B*H*W/dt/1e6:.0f} Mpix/s", flush=True)
        _native.lib().mr_debug_set_raster_probe(0)
        dbary_d = dbary.to(dev)
        for _ in range(3): _native.rasterize_backward(dbary_d, clip_d, tris_d, ids, bary)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): _native.rasterize_backward(dbary_d, clip_d, tris_d, ids, bary)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
        print(f"   bwd: {dt*1e3:.3f} ms -> {clip.shape[0]*H*W*28/dt/1e9:.1f} GB/s", flush=True)
    return ok_ids and ok_z and ok_b

print("lib version", _native.lib().mr_version(), torch.cuda.get_device_name(0), flush=True)
tri_clip = torch.tensor([[[-0.5, -0.5, 0.8, 1.0], [0.0, 0.5, 0.3, 1.0], [0.5, -0.5, 0.3, 1.0]]])
tri = torch.tensor([[0, 1, 2]], dtype=torch.int32)
ok = check("triangle 640x480", tri_clip, tri, 640, 480)
w = torch.tensor([0.2, 0.5, 2.0]).view(1, 3, 1)
ok &= check("persp triangle 640x480", tri_clip * w, tri, 640, 480)
w = torch.tensor([1.0, -0.5, 2.0]).view(1, 3, 1)
ok &= check("behind-eye vertex 100x75", tri_clip * w, tri, 100, 75)
job = synthetic.sphere_job(2, 64, 64, 6)
ok &= check("sphere K=6 64x64 B=2", job["clip"], job["triangles"], 64, 64)
job = synthetic.sphere_job(8, 256, 256, 50)
ok &= check("sphere 5k 256x256 B=8", job["clip"], job["triangles"], 256, 256, time_it=True)
job = synthetic.sphere_job(3, 300, 200, 50)
ok &= check("sphere 5k 300x200 B=3", job["clip"], job["triangles"], 300, 200)
B = int(os.environ.get("CHECK_B", "32"))
job = synthetic.sphere_job(B, 1024, 1024, 50)
ok &= check(f"sphere 5k 1024x1024 B={B}", job["clip"], job["triangles"], 1024, 1024, time_it=True)
print("ALL OK" if ok else "MISMATCH", flush=True)
sys.exit(0 if ok else 1)
