/*
 * mesh_raster_debug.h -- test and profiling hooks of libmesh_raster_hip.so.
 *
 * NOT part of the drop-in boundary (include/mesh_raster.h): nothing here has a counterpart in the
 * reference, product code (the pytorch_mesh_renderer_amd package, bench.py's timed step) never calls it.
 * All state is per calling thread.
 */
#ifndef MESH_RASTER_DEBUG_H_
#define MESH_RASTER_DEBUG_H_

#include "mesh_raster.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Region size of the forward raster kernel for the calling thread's next launches.  The kernel
 * walks 64x64-pixel regions per workgroup, or 32x32 when the launch is small (fewer than four
 * 64x64 regions per CU).  0 = that automatic choice (default); 32 / 64 force one so that the parity
 * tests and the fuzz can run BOTH instantiations on the same inputs.  Results are bit-identical
 * either way; the workspace query follows the setting, so set it before querying. */
int mr_debug_set_raster_region_edge(int edge);

/* Pixel kernel of mr_shade_backward / mr_shade_backward_l1 for the calling thread's next launches:
 * 0 = automatic (default: the lane-accumulating kernel without light_grads and outside the
 * deterministic mode; the rows kernel otherwise), 1 = always the rows kernel (all 36 sums are
 * formed, unwanted outputs are just not written), 2 = as 0.  For the parity tests, which compare
 * the two kernels on the same inputs. */
int mr_debug_set_shade_backward_kernel(int which);

/* Measurement: the calling thread's next forward calls launch their k_raster kernel n times back to back (1 <= n <= 64;
 * default 1) -- identical launches that rewrite the same outputs -- inside the ONE event pair of mr_time_next_kernel
 * (MR_TIMER_RASTER_FORWARD): bench.py times the G-buffer kernel this way, so that the ~5 us of stream idle time an event
 * pair costs is spread over n launches instead of deciding a 130 us figure.  Results are those of one launch. */
int mr_debug_set_raster_repeat(int n);

/* The functor of the most recent per-triangle accumulation pass (k_accumulate_rows / _lanes / _runs: the pixel pass
 * of every backward entry point) launched by any thread of this process, as the compiler spells the launcher's
 * instantiation, e.g. "... [Fn = mr::ShadeFoldLaneFn<1, true>]"; "" before the first one.  The parity tests use it to
 * make sure the specialised kernel they pin to the reference's gradients is the one that ran.  Static storage. */
const char *mr_debug_last_accumulate_kernel(void);

/* Stage-timing probes of k_raster.  Only a library built with -DMR_PROBES (make probes ->
 * libmesh_raster_hip_probes.so) contains the probe instantiations; the production library
 * returns MR_EINVAL for every value but 0, its kernel has no probe code at all.
 *   0 normal | 1 bin only | 2 empty tile walk + stores | 3 bin + tile masks | 8 no depth loop |
 *   16 no coverage loop | 32 no stores | 40 no depth loop and no stores |
 *   64 row-shaped store addresses (each store instruction one contiguous run; scrambled image)
 * The G-buffer is UNDEFINED while a probe is selected. */
int mr_debug_set_raster_probe(int probe);

/* The SoftRas kernels' nearest-point-on-a-segment evaluation on caller-given data: for i < n,
 * out[4 i .. 4 i + 3] = (nearest.x, nearest.y, t, squared distance) of device point points[2 i .. ] against
 * the segment seg_a[2 i ..] -> seg_b[2 i ..], computed by the same two device functions k_soft_setup /
 * k_soft_forward / k_soft_backward call (edge_setup, edge_nearest in soft.hip).  Exists so that the
 * vectors the reference's own test holds for point_to_segment_nearest
 * (src/soft_mesh_renderer/test_rasterize.py:9-44, rasterize.py:169-176) can be checked on the device.
 * Device pointers, asynchronous on `stream`. */
int mr_debug_soft_nearest(const float *points, const float *seg_a, const float *seg_b, int n, float *out, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* MESH_RASTER_DEBUG_H_ */
