set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for mode in 1 0; do
  OUT=gpurun_out/prof_rasterize_$mode
  rm -rf "$OUT"
  MR_FUSED_INTERPOLATION=$mode timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o run -- python3 tools/rasterize_bench.py --attrs ${1:-9} > "$OUT.log" 2>&1 || tail -3 "$OUT.log"
  echo "--- MR_FUSED_INTERPOLATION=$mode: $(grep rasterize $OUT.log)"
  python3 - "$OUT" <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(f)))[:9]:
        print("   %-100s %4s calls  %8.1f us" % (r["Name"][:100], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
