set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/prof_spec
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_spec -o run -- \
    python3 tools/specular_bench.py > gpurun_out/prof_spec.log 2>&1
cat gpurun_out/prof_spec.log | tail -3
f=$(find gpurun_out/prof_spec -name '*kernel_stats.csv' | head -1)
python3 - "$f" <<PY
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:24]:
    print("  %-100s calls %5s avg %9.1f us" % (r["Name"][:100], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
