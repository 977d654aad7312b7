# SQ instruction counters of the SoftRas kernels (tools/soft_bench.py).
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_soft
rm -rf "$OUT" && mkdir -p "$OUT"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES --output-format csv \
    -d "$OUT/a" -o run -- python3 tools/soft_bench.py > "$OUT/a.log" 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_INSTS_SMEM --output-format csv \
    -d "$OUT/b" -o run -- python3 tools/soft_bench.py > "$OUT/b.log" 2>&1
python3 - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmc_soft/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        for key in ("k_soft_forward", "k_soft_backward", "k_soft_setup"):
            if key in r["Kernel_Name"]:
                acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
for key, d in acc.items():
    print(key, "  ".join("%s=%.1fM" % (k[3:], sum(v) / len(v) / 1e6) for k, v in sorted(d.items())))
PY
