set -e
cd "$GRAFT_REPO_ROOT"
for flags in "$@"; do
  make -C pytorch_mesh_renderer_amd/csrc clean >/dev/null
  make -j8 -C pytorch_mesh_renderer_amd/csrc "$flags" all >/dev/null 2>&1
  echo "--- $flags"
  for a in 4 9 16; do python3 - $a <<'PY'
import sys, time, torch
sys.path.insert(0, '.')
from pytorch_mesh_renderer_amd import _native
from pytorch_mesh_renderer_amd.common import synthetic
A = int(sys.argv[1]); B, W, H = 32, 1024, 1024
dev = torch.device('cuda:0')
job = synthetic.sphere_job(B, W, H, 50)
clip, tris = job['clip'].to(dev), job['triangles'].to(dev)
attrs = torch.rand(B, clip.shape[1], A, device=dev); bg = torch.full((A,), -1.0, device=dev)
for _ in range(3): _native.rasterize_interpolate_forward(clip, attrs, tris, bg, W, H)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): _native.rasterize_interpolate_forward(clip, attrs, tris, bg, W, H)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
print("   A=%2d one-pass forward (records + setup + coarse + k_raster<INTERP>): %.3f ms  %.0f GB/s" % (A, dt * 1e3, B * W * H * (16 + 4 * A) / dt / 1e9))
PY
  done
done
