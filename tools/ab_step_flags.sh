# Same-box comparison of compile-time variants by the benchmarked STEP and its kernels' event times, several process runs per
# variant (the loss kernel and the fused forward read differently from process to process on one box: profiles/r06_shade_sched_strategies.txt).
#   gpurun -- 'RUNS=3 bash tools/ab_step_flags.sh "EXTRA=" "EXTRA=-DMR_L1_REVERSE=0" ...'
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for flags in "$@"; do
  make -C pytorch_mesh_renderer_amd/csrc clean >/dev/null
  make -j8 -C pytorch_mesh_renderer_amd/csrc "$flags" all >/dev/null 2>&1
  echo "--- $flags   (step | fused fwd, gbuffer burst, gbuffer in step, shade bwd, l1 fwd; ms)"
  for i in $(seq ${RUNS:-3}); do
    timeout -k 5 200 python bench.py --cpu-sample 0 --extras 0 --steps 100 2>/dev/null | grep -o "\"ms_per_step\": [0-9.]*\|avg_kernel_ms\": [0-9.]*" | sed 's/avg_kernel_ms": //; s/"ms_per_step": //' | tr '\n' ' '; echo
  done
done
