# A/B of compile-time variants inside ONE gpurun call (box-to-box noise is ~5 %).
#   gpurun -- 'bash tools/ab_probe.sh "EXTRA=-DMR_RASTER_WAVES=6" "RASTER_FWD_FLAGS=-fno-slp-vectorize"'
set -e
cd $GRAFT_REPO_ROOT
for flags in "$@"; do
  make -C pytorch_mesh_renderer_amd/csrc clean >/dev/null
  make -j8 -C pytorch_mesh_renderer_amd/csrc "$flags" all >/dev/null 2>&1
  make -j8 -C pytorch_mesh_renderer_amd/csrc "$flags" probes >/dev/null 2>&1
  echo "--- $flags"
  if [ "$AB_BENCH" = step ]; then timeout -k 5 200 python bench.py --cpu-sample 0 --steps 100 2>/dev/null | grep -o "\"ms_per_step\": [0-9.]*\|avg_kernel_ms\": [0-9.]*" | sed 's/avg_kernel_ms": //; s/"ms_per_step": //' | tr '\n' ' '; echo " (step | fused fwd, gbuffer, shade bwd, l1 fwd)"
  elif [ "$AB_BENCH" = bench ]; then timeout -k 5 200 python bench.py --cpu-sample 0 --steps 100 2>/dev/null | grep -o "ms_per_step[^,]*\|avg_kernel_ms[^}]*" | tr '\n' ' '; echo
  elif [ "$AB_BENCH" = kernels ]; then timeout -k 5 100 python tools/shade_bench.py 2>/dev/null | grep shade; timeout -k 5 100 python tools/raster_bench.py --config c3 --backward 2>/dev/null | grep "bwd\|fwd"
  elif [ "$AB_BENCH" = spec ]; then timeout -k 5 200 python tools/specular_bench.py 2>/dev/null | grep "specular=True"
  elif [ "$AB_BENCH" = rasterize ]; then timeout -k 5 200 python tools/rasterize_bench.py 2>/dev/null | grep rasterize
  elif [ "$AB_BENCH" = c4 ]; then timeout -k 5 300 python bench.py --config c4 --cpu-sample 0 --steps 60 2>/dev/null | grep -o "ms_per_step[^,]*\|avg_kernel_ms\": [0-9.]*" | tr '\n' ' '; echo
  elif [ "$AB_BENCH" = rows ]; then timeout -k 5 200 python tools/rasterize_bench.py 2>/dev/null | grep rasterize; timeout -k 5 200 python tools/specular_bench.py 2>/dev/null | grep "specular"; timeout -k 5 300 python bench.py --config c4 --cpu-sample 0 --steps 60 2>/dev/null | grep -o "ms_per_step[^,]*"; timeout -k 5 300 python bench.py --cpu-sample 0 --steps 100 2>/dev/null | grep -o "ms_per_step[^,]*"
  elif [ "$AB_BENCH" = rbwd ]; then timeout -k 5 200 python tools/raster_bench.py --config c3 --backward 2>/dev/null | grep bwd; timeout -k 5 200 python tools/raster_bench.py --config c4 --backward 2>/dev/null | grep bwd
  elif [ "$AB_BENCH" = gather ]; then bash tools/gather_time.sh
  elif [ "$AB_BENCH" = l1 ]; then timeout -k 5 100 python tools/l1_bench.py 2>/dev/null | grep -v amdgpu
  elif [ "$AB_BENCH" = soft ]; then timeout -k 5 200 python tools/soft_bench.py 2>/dev/null | grep config5
  elif [ "$AB_BENCH" = shade ]; then timeout -k 5 100 python tools/shade_bench.py 2>/dev/null | grep shade
  else for v in 0 32 40; do timeout -k 5 100 python tools/raster_bench.py --variant $v 2>/dev/null | grep variant; done; fi
done
