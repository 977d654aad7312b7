# Same-box A/B of two VERSIONS of one source file (interleaved old/new/old/new inside one gpurun call;
# box-to-box noise is ~5 %, run-to-run on one box ~1 %).
#   git show HEAD:pytorch_mesh_renderer_amd/csrc/raster_forward.hip > gpurun_in/old.hip
#   cp pytorch_mesh_renderer_amd/csrc/raster_forward.hip gpurun_in/new.hip
#   gpurun -- 'AB_BENCH=raster bash tools/ab_files.sh pytorch_mesh_renderer_amd/csrc/raster_forward.hip gpurun_in/old.hip gpurun_in/new.hip'
# The file is left at the NEW version on the box (nothing persists there anyway).
set -e
cd $GRAFT_REPO_ROOT
target=$1; old=$2; new=$3
for v in "$old" "$new" "$old" "$new"; do
  cp "$v" "$target"
  make -C pytorch_mesh_renderer_amd/csrc clean >/dev/null
  make -j8 -C pytorch_mesh_renderer_amd/csrc all >/dev/null 2>&1
  [ "$AB_BENCH" = bench ] || make -j8 -C pytorch_mesh_renderer_amd/csrc probes >/dev/null 2>&1
  echo "--- $v"
  case "$AB_BENCH" in
    shade) timeout -k 5 100 python tools/shade_bench.py 2>/dev/null | grep shade ;;
    soft)  timeout -k 5 200 python tools/soft_bench.py 2>/dev/null | grep config5 ;;
    bench) timeout -k 5 200 python bench.py --cpu-sample 0 --steps 200 2>/dev/null | grep -o "\"ms_per_step\": [0-9.]*\|avg_kernel_ms\": [0-9.]*" | sed 's/avg_kernel_ms": //; s/"ms_per_step": //' | tr '\n' ' '; echo " (step | fused fwd, gbuffer, shade bwd, l1 fwd)" ;;
    *)     for k in 0 32 40; do timeout -k 5 100 python tools/raster_bench.py --variant $k 2>/dev/null | grep variant; done ;;
  esac
done
