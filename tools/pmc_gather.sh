set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_gather
rm -rf "$OUT" && mkdir -p "$OUT"
i=0
for set in "TCC_HIT_sum TCC_MISS_sum TCC_EA_RDREQ_sum TCC_REQ_sum" \
           "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY" \
           "TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_ATOMIC_WITH_RET_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" \
           "GRBM_GUI_ACTIVE GRBM_COUNT SQ_INSTS_VALU SQ_INSTS_VMEM_RD"; do
  i=$((i + 1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$OUT/p$i" -o run -- \
      python3 bench.py --steps 3 --warmup 1 --cpu-sample 0 --extras 0 > "$OUT/p$i.log" 2>&1 || true
done
python3 - <<'PY' | tee gpurun_out/pmc_gather/summary.txt
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
keys = ("k_shade_gather", "k_corner_setup", "k_bwd_setup")
for f in glob.glob("gpurun_out/pmc_gather/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        for key in keys:
            if key in r["Kernel_Name"]:
                acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
for key, d in acc.items():
    print(key)
    for k, v in sorted(d.items()):
        print("   %-34s %12.0f  (%d launches)" % (k, sum(v) / len(v), len(v)))
PY
