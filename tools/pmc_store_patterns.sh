# Calibration of the WRITE_SIZE counter for k_raster's store shapes: the store-only micro-kernels of
# tools/ubench/store_patterns.hip -- among them k_raster's exact store instructions over the same 671 MB -- under
# rocprofv3 --pmc WRITE_SIZE, next to their event timings.  If the pattern that IS k_raster's reads 1.00 x its bytes, the
# kernel's 1.19 x is real surplus; if it reads 1.19 x, the counter over-counts that store shape.
#   gpurun -- 'bash tools/pmc_store_patterns.sh [B W H]'     -> gpurun_out/pmc_store/summary.txt
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_store
rm -rf "$OUT" && mkdir -p "$OUT"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o "$OUT/store_patterns" tools/ubench/store_patterns.hip
"$OUT/store_patterns" "$@" > "$OUT/timings.jsonl"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc" -o run -- "$OUT/store_patterns" "$@" > "$OUT/pmc.log" 2>&1
python3 - "$OUT" "$@" <<'PY' | tee "$OUT/summary.txt"
import csv, glob, json, re, sys, collections
out = sys.argv[1]
dims = [int(a) for a in sys.argv[2:5]] if len(sys.argv) > 4 else [32, 1024, 1024]
px = dims[0] * dims[1] * dims[2]
names = [json.loads(l)["pattern"] for l in open(out + "/timings.jsonl") if l.startswith("{")]
tbps = [json.loads(l)["TBps_mean"] for l in open(out + "/timings.jsonl") if l.startswith("{")]
acc = collections.defaultdict(list)
for f in glob.glob(out + "/pmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r"k_store<(?:\(Mode\))?(\d+)>", r["Kernel_Name"])
        if m and r["Counter_Name"] == "WRITE_SIZE":
            acc[int(m.group(1))].append(float(r["Counter_Value"]))
print("WRITE_SIZE calibration, %d x %d x %d (counter in KB per launch; algorithmic = bytes the kernel stores)" % tuple(dims))
for mode in sorted(acc):
    algorithmic = px * (16 if "without the depth" in names[mode] else 20)
    v = acc[mode]
    kb = sum(v) / len(v)
    print("  mode %2d %-62s WRITE_SIZE %12.0f KB = %.3f x algorithmic   (%.2f TB/s by events, n=%d)" % (
        mode, names[mode], kb, kb * 1024 / algorithmic, tbps[mode], len(v)))
PY
