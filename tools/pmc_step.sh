# SQ activity counters of the bench step's kernels, four --pmc passes over `bench.py --steps 3`.
#   gpurun -- 'bash tools/pmc_step.sh'   -> gpurun_out/pmc_step/summary.txt
# (copy the summary to profiles/rNN_pmc_step_summary.txt; MR_SHADE_BACKWARD_KERNEL=1 in the environment
#  profiles the rows kernel instead of the lane-accumulating one; PMC_CMD="python3 tools/soft_bench.py 3"
#  PMC_KEYS="k_soft_backward,k_soft_forward" PMC_OUT=gpurun_out/pmc_soft profiles another command's kernels)
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=${PMC_OUT:-gpurun_out/pmc_step}
CMD=${PMC_CMD:-python3 bench.py --steps 3 --warmup 1 --cpu-sample 0 --extras 0}   # --extras 0: the legs after the timed region launch the same kernels at other shapes
export PMC_OUT=$OUT
rm -rf "$OUT" && mkdir -p "$OUT"
i=0
for set in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" \
           "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
           "SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH" \
           "SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY"; do
  i=$((i + 1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$OUT/p$i" -o run -- \
      $CMD > "$OUT/p$i.log" 2>&1
done
python3 - <<'PY' | tee "$OUT/summary.txt"
import csv, glob, collections, os
acc = collections.defaultdict(lambda: collections.defaultdict(list))
keys = tuple(os.environ["PMC_KEYS"].split(",")) if os.environ.get("PMC_KEYS") else ("k_accumulate_lanes", "k_accumulate_rows", "k_raster<64, 0, true,", "k_raster<64, 0, false, 0,", "k_l1_forward", "k_shade_gather")
for f in glob.glob(os.environ["PMC_OUT"] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        for key in keys:
            if key in r["Kernel_Name"]:
                acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
for key, d in acc.items():
    print(key)
    for k, v in sorted(d.items()):
        print("   %-24s %10.2f M  (%d launches)" % (k, sum(v) / len(v) / 1e6, len(v)))
PY
