# SQ counters of k_raster alone (tools/raster_bench.py), one rocprofv3 --pmc pass per group.
#   gpurun --timeout 900 -- 'bash tools/pmc_raster.sh'
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_raster
rm -rf "$OUT" && mkdir -p "$OUT"
i=0
for group in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR" \
             "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" \
             "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
             "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY" \
             "SQ_INSTS_BRANCH SQ_IFETCH SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU" \
             "SQ_LEVEL_WAVES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_TRANS_F32 GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $group --output-format csv -d "$OUT/g$i" -o run -- \
      python3 tools/raster_bench.py --iters 5 > "$OUT/g$i.log" 2>&1 || { echo "group $i failed"; tail -3 "$OUT/g$i.log"; }
done
python3 - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("gpurun_out/pmc_raster/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_raster<" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    v = acc[k]
    print(f"{k:28s} {sum(v)/len(v):16.0f}  (n={len(v)})")
PY
