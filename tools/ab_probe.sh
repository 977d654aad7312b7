set -e
cd $GRAFT_REPO_ROOT
for p in 1 2; do
  make -C pytorch_mesh_renderer_amd/csrc clean >/dev/null
  make -C pytorch_mesh_renderer_amd/csrc EXTRA=-DMR_PROBE_ROWS=$p >/dev/null 2>&1
  echo "--- probe $p"; timeout -k 5 120 python tools/shade_bench.py | tail -1
done
