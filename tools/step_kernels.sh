# Per-step kernel time accounting of the bench step (rocprofv3 --kernel-trace --stats over 20 steps).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_q -o run --output-format csv -- python3 bench.py --steps 20 --warmup 2 --cpu-sample 0 --extras 0 > gpurun_out/prof_q.log 2>&1
python3 - <<'PY'
import csv, glob
rows = list(csv.DictReader(open(glob.glob("gpurun_out/prof_q/*kernel_stats.csv")[0])))
steps = max(int(r["Calls"]) for r in rows if "k_l1_forward" in r["Name"])
tot = 0.0
for r in rows:
    per = int(r["Calls"]) / steps
    if per >= 0.4:
        us = float(r["AverageNs"]) / 1e3 * per
        tot += us
        print("%-76s x%.1f %8.1f us/step" % (r["Name"][:76], per, us))
print("total %.1f us/step over %d steps (20 with the shading epilogue, 22 without)" % (tot, steps))
PY
