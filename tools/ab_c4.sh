# Same-box A/B of two versions of raster_forward.hip at configs[3]'s per-GPU shape (and configs[1]).
set -e
cd $GRAFT_REPO_ROOT
target=pytorch_mesh_renderer_amd/csrc/raster_forward.hip
for v in "$1" "$2" "$1" "$2"; do
  cp "$v" "$target"
  make -C pytorch_mesh_renderer_amd/csrc clean >/dev/null
  make -j8 -C pytorch_mesh_renderer_amd/csrc all >/dev/null 2>&1
  echo "--- $v"
  timeout -k 5 100 python tools/raster_bench.py --config c4 2>/dev/null | grep fwd
  timeout -k 5 100 python tools/raster_bench.py --config c2 2>/dev/null | grep fwd
  timeout -k 5 300 python bench.py --config c4 --cpu-sample 0 --steps 60 2>/dev/null | grep -o "ms_per_step[^,]*\|avg_kernel_ms\": [0-9.]*" | tr '\n' ' '; echo
done
