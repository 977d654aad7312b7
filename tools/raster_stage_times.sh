# Kernel-level durations (rocprofv3 --kernel-trace --stats) of mr_rasterize_forward's kernels for each
# stage-timing probe of k_raster (probes build).  gpurun --timeout 900 -- 'bash tools/raster_stage_times.sh [config]'
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
CFG=${1:-c3}
OUT=gpurun_out/stage_times
rm -rf "$OUT" && mkdir -p "$OUT"
for v in ${PROBES:-0 64 32 8 16 3 1 2}; do
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/v$v" -o run -- \
      python3 tools/raster_bench.py --config $CFG --iters 10 --variant $v > "$OUT/v$v.log" 2>&1 || echo "variant $v failed"
done
python3 - <<'PY'
import csv, glob
for v in [int(x) for x in __import__("os").environ.get("PROBES", "0 64 32 8 16 3 1 2").split()]:
    for f in glob.glob("gpurun_out/stage_times/v%d/**/*kernel_stats.csv" % v, recursive=True):
        rows = [r for r in csv.DictReader(open(f)) if any(k in r["Name"] for k in ("k_raster", "k_setup", "k_coarse"))]
        print("probe %2d  " % v + "  ".join("%s %.1f us (n=%s)" % (r["Name"].split("(")[0].split("::")[-1][:22], float(r["AverageNs"]) / 1e3, r["Calls"]) for r in rows))
PY
