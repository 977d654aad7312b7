# Run ON the GPU box (through gpurun) from the repo root: kernel-trace statistics and HBM
# traffic counters of the bench command, one rocprofv3 pass each (PMC passes carry
# --kernel-trace only, as the pool requires).  Outputs under gpurun_out/prof_final/;
# tools/summarize_profiles.py turns them into the files committed under profiles/.
#   gpurun --timeout 900 -- 'bash tools/collect_profiles.sh [c4]'
# (with a configuration other than the default the outputs go to gpurun_out/prof_final_<config>/ and
#  `python tools/summarize_profiles.py <tag> <config>` files them under "configs" in kernel_traffic.json)
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
CFG=${1:-c3}
OUT=gpurun_out/prof_final
[ "$CFG" != c3 ] && OUT=gpurun_out/prof_final_$CFG
rm -rf "$OUT" && mkdir -p "$OUT"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o run -- \
    python3 bench.py --config $CFG --steps 20 --warmup 3 --cpu-sample 0 --extras 0 > "$OUT/bench_under_rocprof.log" 2>&1
for counter in WRITE_SIZE FETCH_SIZE; do
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc $counter --output-format csv -d "$OUT/pmc_$counter" -o run -- \
      python3 bench.py --config $CFG --steps 3 --warmup 1 --cpu-sample 0 --extras 0 > "$OUT/pmc_$counter.log" 2>&1
done
timeout -k 10 250 python3 bench.py --config $CFG $( [ "$CFG" != c3 ] && echo --cpu-sample 0 ) > "$OUT/bench.log" 2>&1
# round 6: the legs after the timed region -- configs[1], configs[3]'s share, configs[4], the specular
# step, rasterize() with nine attributes -- under the kernel trace too, so that every figure of the driver's line has a
# kernel-level breakdown of the same tag (summarize_profiles.py: <tag>_bench_extras_kernel_stats.csv)
if [ "$CFG" = c3 ]; then
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_extras" -o run -- \
      python3 bench.py --steps 20 --warmup 3 --cpu-sample 0 --extras 1 > "$OUT/bench_extras_under_rocprof.log" 2>&1
fi
grep '"metric"' "$OUT/bench.log" | cut -c1-200
find "$OUT" -name '*.csv' | sort
