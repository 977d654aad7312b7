# Every number DESIGN.md section 5 quotes, on one box.  gpurun --timeout 900 -- 'bash tools/measure_all.sh'
cd "$GRAFT_REPO_ROOT"
run() { echo "### $*"; timeout -k 10 300 "$@" 2>&1 | grep -v "amdgpu.ids" | tail -4; }
run python bench.py
run python bench.py --config c4 --cpu-sample 0 --steps 100
run python tools/raster_bench.py --config c2 --backward
run python tools/raster_bench.py --config c3 --backward
run python tools/raster_bench.py --config c4 --backward
run python tools/step_bench.py
run python tools/rasterize_bench.py
run python tools/specular_bench.py
run python tools/soft_bench.py
run python tools/graph_bench.py
run python tools/shade_bench.py
