set -e
cd $GRAFT_REPO_ROOT
for p in none 0 1 2; do
  make -C pytorch_mesh_renderer_amd/csrc clean >/dev/null
  if [ $p = none ]; then make -C pytorch_mesh_renderer_amd/csrc >/dev/null 2>&1; else make -C pytorch_mesh_renderer_amd/csrc EXTRA=-DMR_PROBE_FLUSH=$p >/dev/null 2>&1; fi
  echo "--- probe $p"; python tools/shade_bench.py | tail -1
done
