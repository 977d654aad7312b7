# HBM write / fetch traffic of mr_rasterize_forward's kernels (tools/raster_bench.py), one rocprofv3 --pmc pass each.
#   gpurun -- 'bash tools/pmc_write_size.sh [config]'
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
CFG=${1:-c3}
OUT=gpurun_out/pmc_write
rm -rf "$OUT" && mkdir -p "$OUT"
for c in WRITE_SIZE FETCH_SIZE; do
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$OUT/$c" -o run -- \
      python3 tools/raster_bench.py --config $CFG --iters 5 > "$OUT/$c.log" 2>&1 || { echo "$c failed"; tail -3 "$OUT/$c.log"; }
done
python3 - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("gpurun_out/pmc_write/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_raster<" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    v = acc[k]
    print(f"k_raster {k:12s} {sum(v)/len(v):14.1f} KB per launch (n={len(v)})")
PY
