# Same-box A/B of two whole csrc TREES (a change that spans several files), interleaved old/new/old/new inside one
# gpurun call: each tree is built into its own library and bench.py loads it through MR_NATIVE_LIB_PATH.
#   mkdir -p gpurun_in/old_csrc && git archive HEAD pytorch_mesh_renderer_amd/csrc | tar -x -C gpurun_in/old_csrc --strip-components=2
#   gpurun -- 'bash tools/ab_trees.sh gpurun_in/old_csrc pytorch_mesh_renderer_amd/csrc [bench.py arguments]'
# (the trees must share the ABI the Python layer speaks; include/ is the repo's)
set -e
cd $GRAFT_REPO_ROOT
old=$1; new=$2; shift 2
for t in "$old" "$new"; do
  make -j8 -C "$t" OBJDIR=build_ab OUT=libmesh_raster_hip_ab.so COMMON_INC="-I$GRAFT_REPO_ROOT/include" all >/dev/null 2>&1 || { echo "build of $t failed"; exit 1; }
done
for t in "$old" "$new" "$old" "$new"; do
  echo "--- $t"
  MR_NATIVE_LIB_PATH="$GRAFT_REPO_ROOT/$t/libmesh_raster_hip_ab.so" timeout -k 5 200 python bench.py --cpu-sample 0 --extras 0 --steps 200 "$@" 2>/dev/null \
    | grep -o "\"ms_per_step\": [0-9.]*\|avg_kernel_ms\": [0-9.]*" | sed 's/avg_kernel_ms": //; s/"ms_per_step": //' | tr '\n' ' '
  echo " (step | fused fwd, gbuffer, shade bwd, l1 fwd)"
done
