set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/prof_c4
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_c4 -o run -- \
    python3 bench.py --config c4 --steps 30 --warmup 3 --cpu-sample 0 > gpurun_out/prof_c4.log 2>&1
f=$(find gpurun_out/prof_c4 -name '*kernel_stats.csv' | head -1)
python3 - "$f" <<PY
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:16]:
    print("  %-100s calls %5s avg %9.1f us" % (r["Name"][:100], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
