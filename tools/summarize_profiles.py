"""Condense gpurun_out/prof_final (tools/collect_profiles.sh) into the files under profiles/.

    python tools/summarize_profiles.py <tag> [config]     # e.g. r03_b, r03_b c4

Writes profiles/<tag>_bench_kernel_stats.csv (rocprofv3 --stats, top kernels),
profiles/<tag>_bench.json (the bench lines of the same box) and rewrites
profiles/kernel_traffic.json (HBM bytes per launch of the forward kernels and of the shading backward from the PMC
passes; read by bench.py, which labels it as an offline value).
"""
import csv, glob, json, os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "prof_final")
tag = sys.argv[1]
config = sys.argv[2] if len(sys.argv) > 2 else "c3"   # bench.py --config the passes ran with
if config != "c3":
    SRC = SRC + "_" + config
prof = os.path.join(ROOT, "profiles")


def one(pattern):
    hits = glob.glob(os.path.join(SRC, pattern), recursive=True)
    if not hits:
        raise SystemExit("missing " + pattern)
    return hits[0]


# 1. kernel statistics
rows = list(csv.DictReader(open(one("stats/**/*kernel_stats.csv"))))
suffix = "" if config == "c3" else "_" + config
with open(os.path.join(prof, tag + suffix + "_bench_kernel_stats.csv"), "w", newline="") as f:
    w = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
    w.writeheader()
    for r in rows[:24]:
        w.writerow(r)

# 1b. the same with the legs after the timed region (round 6): every kernel of every leg, 48 rows
extras = glob.glob(os.path.join(SRC, "stats_extras/**/*kernel_stats.csv"), recursive=True)
if extras:
    rows_x = list(csv.DictReader(open(extras[0])))
    with open(os.path.join(prof, tag + "_bench_extras_kernel_stats.csv"), "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=list(rows_x[0].keys()))
        w.writeheader()
        for r in rows_x[:48]:
            w.writerow(r)

# 2. bench lines (plain and under rocprof)
lines = {}
for name in ("bench.log", "bench_under_rocprof.log", "bench_extras_under_rocprof.log"):
    if not os.path.exists(os.path.join(SRC, name)):
        continue
    for line in open(os.path.join(SRC, name)):
        if line.startswith('{"metric"'):
            lines[name[:-4]] = json.loads(line)
json.dump(lines, open(os.path.join(prof, tag + suffix + "_bench.json"), "w"), indent=1)

# 3. HBM traffic of the kernels of interest, per launch
# k_raster<R, PROBE, SHADE>: "true>" = with the shading epilogue (the step's forward), "false>" = the
# G-buffer kernel alone (bench.py runs 22 such steps after its timed region)
wanted = {"k_raster_shade": "k_raster<64, 0, true,", "k_raster": "k_raster<64, 0, false, 0",   # (SHADE = false, INTERP = 0; whatever template parameters follow)
          "k_shade_forward": "k_shade_forward(", "ShadeGradFn": ("ShadeFoldLaneFn", "ShadeLaneFn", "ShadeGradFn"),
          "k_l1_forward": ("k_l1_forward(", "k_l1_forward_regions("), "k_l1_backward": "k_l1_backward("}
raw = {k: {} for k in wanted}
for counter in ("WRITE_SIZE", "FETCH_SIZE"):
    acc = {k: [] for k in wanted}
    for r in csv.DictReader(open(one("pmc_%s/**/*counter_collection.csv" % counter))):
        if r["Counter_Name"] != counter:
            continue
        for k, needle in wanted.items():
            needles = needle if isinstance(needle, tuple) else (needle,)
            if any(n in r["Kernel_Name"] for n in needles):
                acc[k].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        if v:
            raw[k][counter] = sum(v) / len(v)
            raw[k]["launches_" + counter] = len(v)
bench = lines.get("bench", {})


def traffic(entry):
    # gfx950: WRITE_SIZE counts KB as is, FETCH_SIZE is doubled (MI355X_MICROARCH.md, HBM section)
    return int(round((entry.get("WRITE_SIZE", 0.0) + 2.0 * entry.get("FETCH_SIZE", 0.0)) * 1024))


cmd = "python3 bench.py --steps 3 --warmup 1 --cpu-sample 0" + ("" if config == "c3" else " --config " + config)
out = {
    "command": cmd,
    "tag": tag,
    "how": "rocprofv3 --kernel-trace --pmc WRITE_SIZE and --pmc FETCH_SIZE in separate passes over the command "
           "(tools/collect_profiles.sh), averaged over each kernel's dispatches; bytes = (WRITE_SIZE + "
           "2*FETCH_SIZE) * 1024 (FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950; WRITE_SIZE is "
           "exact for k_shade_forward's 16-B stores in the same run: 524288 KB = 537 MB)",
    "kernels": {
        "k_raster_shade": {"bytes_per_launch": traffic(raw["k_raster_shade"]),
                           "algorithmic_bytes": bench.get("roofline", {}).get("algorithmic_bytes")},
        "k_raster": {"bytes_per_launch": traffic(raw["k_raster"]),
                     "algorithmic_bytes": bench.get("roofline_gbuffer", {}).get("algorithmic_bytes")},
        "shade_backward": {"bytes_per_launch": traffic(raw["ShadeGradFn"]),
                           "algorithmic_bytes": bench.get("roofline_shade_backward", {}).get("algorithmic_bytes")},
        "l1_forward": {"bytes_per_launch": traffic(raw["k_l1_forward"]),
                       "algorithmic_bytes": bench.get("roofline_l1_forward", {}).get("algorithmic_bytes")},
    },
    "raw": raw,
}
# one file for every configuration: the default (configs[2]) at the top level, as bench.py has read it
# since round 1, the others under "configs"
path = os.path.join(prof, "kernel_traffic.json")
if config == "c3":
    merged = dict(out)
    if os.path.exists(path):
        merged["configs"] = json.load(open(path)).get("configs", {})
else:
    merged = json.load(open(path)) if os.path.exists(path) else {}
    merged.setdefault("configs", {})[config] = out
json.dump(merged, open(path, "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if k != "how"}, indent=1))
