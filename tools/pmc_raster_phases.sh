# VALU / SALU / LDS instruction counts of k_raster per stage-timing probe (mr_set_raster_tile_shape).
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_phases
rm -rf "$OUT" && mkdir -p "$OUT"
for v in 0 1 3 16 8 32; do
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES --output-format csv \
      -d "$OUT/v$v" -o run -- python3 tools/raster_bench.py --iters 3 --variant $v > "$OUT/v$v.log" 2>&1 || echo "variant $v failed"
done
python3 - <<'PY'
import csv, glob, collections
for v in (0, 1, 3, 16, 8, 32):
    acc = collections.defaultdict(list)
    for f in glob.glob("gpurun_out/pmc_phases/v%d/**/*counter_collection.csv" % v, recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_raster<" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print("variant %2d " % v + "  ".join("%s=%.1fM" % (k[3:], sum(x) / len(x) / 1e6) for k, x in sorted(acc.items())))
PY
