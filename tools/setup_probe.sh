# (Round 6 record: this probe needs a k_setup that honours -DMR_TIMING_SKIP_CORNER_STORE; the hook was removed with the experiment,
#  profiles/r06_lean_prepared_forward.txt.  Kept for the method: the upper half of k_setup's durations = its launches inside timed steps.)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for flags in "EXTRA=" "EXTRA=-DMR_TIMING_SKIP_CORNER_STORE=1"; do
  make -C pytorch_mesh_renderer_amd/csrc clean >/dev/null; make -j8 -C pytorch_mesh_renderer_amd/csrc "$flags" all >/dev/null 2>&1
  for cfg in c4 c3; do
    rm -rf gpurun_out/setup_probe; timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/setup_probe -o run -- python3 bench.py --config $cfg --steps 30 --warmup 3 --cpu-sample 0 --extras 0 > /dev/null 2>&1
    python3 - "$flags" $cfg <<'PY'
import csv, glob, sys
d = [ (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for f in glob.glob("gpurun_out/setup_probe/**/*kernel_trace.csv", recursive=True) for r in csv.DictReader(open(f)) if "k_setup" in r["Kernel_Name"]]
d.sort()
hi = d[len(d) // 2:]     # the upper half: the launches with the prepared block (the timed steps)
print("%-44s %s: k_setup n=%d, lower half median %.1f us, upper half median %.1f us" % (sys.argv[1], sys.argv[2], len(d), d[len(d) // 4], hi[len(hi) // 2]))
PY
  done
done
rm -rf gpurun_out/setup_probe
