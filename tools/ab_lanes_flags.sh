# Same-box comparison of compile-time variants of the shading backward's pixel pass (k_accumulate_lanes<ShadeLaneFn>)
# inside the benchmarked step: kernel time (bench.py's HIP events) and SQ instruction / activity counters per launch.
#   gpurun -- 'bash tools/ab_lanes_flags.sh "EXTRA=" "EXTRA=-DMR_...=1" ...'
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/ab_lanes
i=0
for flags in "$@"; do
  i=$((i+1))
  make -C pytorch_mesh_renderer_amd/csrc clean >/dev/null
  make -j8 -C pytorch_mesh_renderer_amd/csrc "$flags" all >/dev/null 2>&1
  echo "--- $flags"
  [ -n "$AB_TESTS" ] && { timeout -k 10 400 python -m pytest tests/test_render_gpu.py -x -q 2>&1 | tail -1; }
  timeout -k 5 200 python bench.py --cpu-sample 0 --extras 0 --steps 100 2>/dev/null | grep -o "\"ms_per_step\": [0-9.]*\|avg_kernel_ms\": [0-9.]*" | sed 's/avg_kernel_ms": //; s/"ms_per_step": //' | tr '\n' ' '; echo " (step | fused fwd, gbuffer, shade bwd, l1 fwd)"
  OUT=gpurun_out/ab_lanes/p$i
  rm -rf "$OUT"
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH --output-format csv -d "$OUT" -o run -- \
      python3 bench.py --steps 3 --warmup 1 --cpu-sample 0 --extras 0 > "$OUT.log" 2>&1 || { echo "pmc failed"; tail -3 "$OUT.log"; }
  python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_accumulate_lanes" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("   k_accumulate_lanes per launch: " + "  ".join("%s %.2f M" % (k.replace("SQ_", ""), sum(v) / len(v) / 1e6) for k, v in sorted(acc.items())))
PY
done
