# kernel-trace statistics of k_shade_gather inside the bench step (one rocprofv3 run)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_q -o run --output-format csv -- python3 bench.py --steps 10 --warmup 2 --cpu-sample 0 > gpurun_out/prof_q.log 2>&1
python3 -c "
import csv,glob
for r in csv.DictReader(open(glob.glob('gpurun_out/prof_q/*kernel_stats.csv')[0])):
    if 'k_shade_gather' in r['Name'] or 'k_l1_finish' in r['Name']: print(r['Name'][:50], r['Calls'], 'avg us', float(r['AverageNs'])/1e3)
"
