# rocprofv3 kernel statistics of the non-headline configurations, one pass each (run ON the GPU box):
#   gpurun --timeout 900 -- 'bash tools/profile_configs.sh TAG'
# -> gpurun_out/prof_TAG/{c4,soft,spec}_kernel_stats.csv + the tools' own lines; copy what is to be judged
#    to profiles/ (tools/summarize_profiles is not needed: the stats CSV is already the summary).
set -e
TAG=${1:-r03}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_$TAG
rm -rf "$OUT" && mkdir -p "$OUT"
prof() {  # name, then the program (python3 first: no wrapper between rocprofv3 and the program)
  name=$1; shift
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/$name" -o run -- "$@" > "$OUT/$name.log" 2>&1 || true
  f=$(find "$OUT/$name" -name '*kernel_stats.csv' | head -1)
  [ -n "$f" ] && python3 - "$f" "$OUT/${name}_kernel_stats.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
with open(sys.argv[2], "w", newline="") as out:   # the 25 largest kernels, names cut to 160 characters
    w = csv.writer(out)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
    for r in rows[:25]:
        w.writerow([r["Name"][:160], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"]])
PY
  grep -h "ms/step\|ms_per_step\|specular=" "$OUT/$name.log" | cut -c1-300 > "$OUT/${name}_line.txt" || true
}
prof c4 python3 bench.py --config c4 --cpu-sample 0 --steps 30 --warmup 3
prof soft python3 tools/soft_bench.py
prof spec python3 tools/specular_bench.py
prof c3 python3 bench.py --cpu-sample 0 --steps 30 --warmup 3
ls "$OUT"
