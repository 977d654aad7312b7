# Same-box, same-library A/B of an ENVIRONMENT switch by the benchmarked step: alternating process runs.
#   gpurun -- 'RUNS=3 CFG=c3 bash tools/ab_env.sh MR_LEAN_PREPARED 0 1'
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
var=$1; shift
for i in $(seq ${RUNS:-3}); do
  for v in "$@"; do
    printf "%s=%s  " "$var" "$v"
    env "$var=$v" timeout -k 5 200 python bench.py --config ${CFG:-c3} --cpu-sample 0 --extras 0 --steps 100 2>/dev/null | grep -o "\"ms_per_step\": [0-9.]*\|avg_kernel_ms\": [0-9.]*" | sed 's/avg_kernel_ms": //; s/"ms_per_step": //' | tr '\n' ' '; echo " (step | fused fwd, gbuffer, in-step, shade bwd, l1 fwd)"
  done
done
