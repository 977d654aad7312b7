# Vector / scalar / LDS instruction counts of k_raster per stage-timing probe (probes build), one --pmc pass each.
#   gpurun --timeout 900 -- 'bash tools/raster_stage_insts.sh [config]'
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
CFG=${1:-c3}
OUT=gpurun_out/stage_insts
rm -rf "$OUT" && mkdir -p "$OUT"
for v in ${PROBES:-0 32 8 16 3 1 2}; do
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH --output-format csv -d "$OUT/v$v" -o run -- \
      python3 tools/raster_bench.py --config $CFG --iters 3 --variant $v > "$OUT/v$v.log" 2>&1 || echo "variant $v failed"
done
python3 - <<'PY'
import csv, glob, collections
for v in [int(x) for x in __import__("os").environ.get("PROBES", "0 32 8 16 3 1 2").split()]:
    acc = collections.defaultdict(list)
    for f in glob.glob("gpurun_out/stage_insts/v%d/**/*counter_collection.csv" % v, recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_raster" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print("probe %2d  " % v + "  ".join("%s %.1f M" % (k.replace("SQ_INSTS_", ""), sum(x) / len(x) / 1e6) for k, x in sorted(acc.items())))
PY
