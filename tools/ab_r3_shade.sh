# Round 3, same-box A/B of the shading backward's pixel kernels inside the bench step:
#   rows kernel (forced) in builds of run_accum.h's reduction loop, then the lane-accumulating kernel.
#   gpurun -- 'bash tools/ab_r3_shade.sh'
set -e
cd $GRAFT_REPO_ROOT
line() { grep -o "ms_per_step[^,]*\|avg_kernel_ms[^}]*" | tr '\n' ' '; echo; }
for round in 1 2; do
  for flags in "EXTRA=-DMR_ROWS_PIPELINE=0" "EXTRA=-DMR_ROWS_PIPELINE=1"; do
    make -C pytorch_mesh_renderer_amd/csrc clean >/dev/null
    make -j8 -C pytorch_mesh_renderer_amd/csrc "$flags" all >/dev/null 2>&1
    echo "--- $flags, rows kernel"
    MR_SHADE_BACKWARD_KERNEL=1 timeout -k 5 200 python bench.py --cpu-sample 0 --steps 100 2>/dev/null | line
    echo "--- $flags, lanes kernel"
    MR_SHADE_BACKWARD_KERNEL=0 timeout -k 5 200 python bench.py --cpu-sample 0 --steps 100 2>/dev/null | line
  done
done
