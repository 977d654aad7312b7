# Instruction counts, wait cycles and written bytes of k_raster per stage-timing probe (probes build), one --pmc pass per
# counter group (round 6: the wait / store counters DESIGN r5 named for configs[3]'s shape).
#   gpurun --timeout 1200 -- 'make -C pytorch_mesh_renderer_amd/csrc -j8 probes >/dev/null 2>&1; bash tools/raster_stage_insts.sh [config]'
# Probes (include/mesh_raster_debug.h): 0 normal | 32 no stores | 8 no depth loop | 16 no coverage and no depth loop (the empty
# walk + stores) | 3 bin + tile masks, no walk.  Output: gpurun_out/stage_insts_<config>/summary.txt
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
CFG=${1:-c3}
OUT=gpurun_out/stage_insts_$CFG
rm -rf "$OUT" && mkdir -p "$OUT"
g=0
for group in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH" \
             "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VMEM_WR SQ_BUSY_CYCLES" \
             "SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS" \
             "WRITE_SIZE"; do
  g=$((g + 1))
  for v in ${PROBES:-0 32 8 16 3}; do
    timeout -k 10 200 rocprofv3 --kernel-trace --pmc $group --output-format csv -d "$OUT/g$g/v$v" -o run -- \
        python3 tools/raster_bench.py --config $CFG --iters 3 --variant $v > "$OUT/g$g.v$v.log" 2>&1 || echo "group $g variant $v failed"
  done
done
for v in ${PROBES:-0 32 8 16 3}; do timeout -k 10 100 python3 tools/raster_bench.py --config $CFG --iters 20 --variant $v 2>/dev/null | grep variant; done > "$OUT/times.txt"
# ... and the kernels' own durations per probe (plain kernel trace, no counters)
for v in ${PROBES:-0 32 8 16 3}; do
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/t/v$v" -o run -- \
      python3 tools/raster_bench.py --config $CFG --iters 10 --variant $v > "$OUT/t.v$v.log" 2>&1 || echo "trace of variant $v failed"
done
STAGE_OUT=$OUT python3 - <<'PY' | tee "$OUT/summary.txt"
import csv, glob, collections, os
out = os.environ["STAGE_OUT"]
print(open(out + "/times.txt").read().rstrip())
for v in [int(x) for x in os.environ.get("PROBES", "0 32 8 16 3").split()]:
    for f in glob.glob(out + "/t/v%d/**/*kernel_stats.csv" % v, recursive=True):
        rows = [r for r in csv.DictReader(open(f)) if r["Name"].startswith(("void mr::", "mr::"))]
        print("probe %2d kernels (avg us): " % v + ", ".join("%s %.1f" % (r["Name"].split("(")[0].split("::")[-1][:28], float(r["AverageNs"]) / 1e3) for r in rows[:5]))
for v in [int(x) for x in os.environ.get("PROBES", "0 32 8 16 3").split()]:
    acc = collections.defaultdict(list)
    for f in glob.glob(out + "/g*/v%d/**/*counter_collection.csv" % v, recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_raster" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    def fmt(k, x):
        m = sum(x) / len(x)
        return "%s %.0f KB" % (k, m) if k.endswith("_SIZE") else "%s %.2f M" % (k.replace("SQ_", ""), m / 1e6)
    print("probe %2d  " % v + "  ".join(fmt(k, x) for k, x in sorted(acc.items())))
PY
find "$OUT" -name '*.csv' -delete; find "$OUT" -name '*.log' -size -2k -delete
