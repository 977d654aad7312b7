# Same-box comparison of compile-time variants of k_raster: time (tools/raster_bench.py, whole mr_rasterize_forward
# and the no-store probe) and HBM write traffic (rocprofv3 --pmc WRITE_SIZE) per flag set.
#   gpurun -- 'bash tools/ab_raster_flags.sh "EXTRA=" "EXTRA=-DMR_TILE_W=32" ...'
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/ab_flags
i=0
for flags in "$@"; do
  i=$((i+1))
  make -C pytorch_mesh_renderer_amd/csrc clean >/dev/null
  make -j8 -C pytorch_mesh_renderer_amd/csrc "$flags" all >/dev/null 2>&1
  make -j8 -C pytorch_mesh_renderer_amd/csrc "$flags" probes >/dev/null 2>&1
  echo "--- $flags"
  [ -n "$AB_TESTS" ] && { timeout -k 10 300 python -m pytest tests/test_raster_gpu.py -x -q 2>&1 | tail -1; }
  for v in 0 32; do timeout -k 5 100 python tools/raster_bench.py --variant $v 2>/dev/null | grep variant; done
  [ -n "$AB_C4" ] && timeout -k 5 100 python tools/raster_bench.py --config c4 --variant 0 2>/dev/null | grep variant
  OUT=gpurun_out/ab_flags/w$i
  rm -rf "$OUT"
  MR_NATIVE_LIB_PATH= timeout -k 10 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT" -o run -- \
      python3 tools/raster_bench.py --iters 5 > "$OUT.log" 2>&1 || { echo "pmc failed"; tail -3 "$OUT.log"; }
  python3 - "$OUT" <<'PY'
import csv, glob, sys
v = [float(r["Counter_Value"]) for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)
     for r in csv.DictReader(open(f)) if "k_raster<" in r["Kernel_Name"] and r["Counter_Name"] == "WRITE_SIZE"]
print("   k_raster WRITE_SIZE %.1f MB per launch (n=%d; G-buffer 671.1 MB)" % (sum(v) / max(len(v), 1) * 1024 / 1e6, len(v)))
PY
done
