# Run-to-run spread of bench.py's kernel timings on ONE box: the fused forward kernel reads 0.186-0.188 ms in some process
# runs and 0.203-0.207 in others of the same tree (round 6: profiles/r06_b_bench_driver_like.json).  N plain runs, then N with
# the allocator's expandable segments.  (Result, profiles/r06_run_variance.txt: on one box fifteen runs -- five of them with a
# 24 GB block reserved first, a hook since removed -- all read 0.203-0.207: the spread is between boxes, not between runs.)
#   gpurun -- 'bash tools/run_variance.sh [N]'
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
N=${1:-5}
one() {
  timeout -k 10 200 python3 bench.py --steps 40 --warmup 5 --cpu-sample 0 --extras 0 2>/dev/null | python3 -c "
import json, sys
for line in sys.stdin:
    if line.startswith('{'):
        l = json.loads(line)
        print('   step %.4f | fwd %.4f gbuffer %.4f in-step %.4f bwd %.4f l1 %.4f' % (l['ms_per_step'], l['roofline']['avg_kernel_ms'], l['roofline_gbuffer']['avg_kernel_ms'], l['roofline_gbuffer_in_step']['avg_kernel_ms'], l['roofline_shade_backward']['avg_kernel_ms'], l['roofline_l1_forward']['avg_kernel_ms']))
"
}
echo "--- plain"; for i in $(seq $N); do one; done
echo "--- PYTORCH_HIP_ALLOC_CONF=expandable_segments:True"; for i in $(seq $N); do PYTORCH_HIP_ALLOC_CONF=expandable_segments:True one; done
