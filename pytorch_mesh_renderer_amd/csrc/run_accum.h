// Per-triangle scatter-add of per-pixel quantities, the gfx950 way.
//
// Both backward passes of the path end in "for every covered pixel, add K numbers
// to the triangle that owns the pixel" (reference: the nine '+=' of
// rasterize_triangles.cpp:232-269, and the index_put_(accumulate=True) that
// autograd derives from the gather in src/mesh_renderer/rasterize.py:130-132).
// Doing that with one global atomic per number per pixel is ~20 G atomics/s on
// MI355X -- tens of milliseconds at 1024^2 x 32.  Instead:
//
//   1. each lane walks DOWN one pixel column (wave loads stay 64 consecutive
//      pixels = coalesced) and keeps kAcc running sums in registers for as long
//      as the triangle id does not change (runs are tens of pixels long);
//   2. a finished run goes into a workgroup-local LDS hash table keyed by triangle
//      id with ds_add_f32;
//   3. the table is drained once per workgroup with contiguous 48-byte global
//      float atomics into acc[image][triangle][kAcc].
//
// PixelFn supplies the per-pixel values:
//   struct Fn { __device__ bool operator()(size_t pix, int T, int &tri, float (&v)[12]) const; };
// returning false for pixels that contribute nothing.
#pragma once

#include "mr_internal.h"

namespace mr {

constexpr int kRunThreads = 256;
constexpr int kRunRowsPerWave = 32;                                  // pixels each lane walks
constexpr int kRunRegionH = kRunRowsPerWave * (kRunThreads / kWave);  // 128 rows / workgroup
constexpr int kRunSlots = 512;                                       // LDS hash slots
constexpr int kRunSlotsLog2 = 9;
constexpr int kAcc = 12;
constexpr int kRunMaxProbe = 16;

__device__ __forceinline__ int run_find_slot(int *keys, int tri) {
  unsigned h = ((unsigned)tri * 2654435761u) >> (32 - kRunSlotsLog2);
  for (int probe = 0; probe < kRunMaxProbe; ++probe) {
    const int old = atomicCAS(&keys[h], -1, tri);
    if (old == -1 || old == tri) return (int)h;
    h = (h + 1) & (kRunSlots - 1);
  }
  return -1;
}

__device__ __forceinline__ void run_flush(int *keys, float *vals, float *acc_img, int tri,
                                          float (&a)[kAcc]) {
  if (tri < 0) return;
  const int slot = run_find_slot(keys, tri);
  if (slot >= 0) {
#pragma unroll
    for (int k = 0; k < kAcc; ++k) atomicAdd(&vals[slot * kAcc + k], a[k]);
  } else {  // table saturated (very dense mesh): straight to HBM
#pragma unroll
    for (int k = 0; k < kAcc; ++k) atomicAdd(&acc_img[(size_t)tri * kAcc + k], a[k]);
  }
#pragma unroll
  for (int k = 0; k < kAcc; ++k) a[k] = 0.0f;
}

template <class PixelFn>
__global__ __launch_bounds__(kRunThreads) void k_accumulate_runs(
    PixelFn fn, int T, int W, int H, int regions_x, int regions_per_image, int n_regions,
    int regions_per_xcd, float *__restrict__ acc) {
  __shared__ int s_keys[kRunSlots];
  __shared__ float s_vals[kRunSlots * kAcc];

  const int region = xcd_contiguous_block((int)blockIdx.x, n_regions, regions_per_xcd);
  if (region < 0) return;
  const int img = region / regions_per_image;
  const int rr = region - img * regions_per_image;
  const int ry = rr / regions_x;
  const int rx = rr - ry * regions_x;

  const int tid = (int)threadIdx.x;
  for (int i = tid; i < kRunSlots; i += kRunThreads) s_keys[i] = -1;
  for (int i = tid; i < kRunSlots * kAcc; i += kRunThreads) s_vals[i] = 0.0f;
  __syncthreads();

  const int lane = tid & (kWave - 1);
  const int wave = tid >> 6;
  const int x = rx * kWave + lane;
  const int y_begin = ry * kRunRegionH + wave * kRunRowsPerWave;
  const int y_end = min(y_begin + kRunRowsPerWave, H);
  float *acc_img = acc + (size_t)img * T * kAcc;

  if (x < W) {
    float a[kAcc];
#pragma unroll
    for (int k = 0; k < kAcc; ++k) a[k] = 0.0f;
    int run_tri = -1;
    size_t pix = ((size_t)img * H + y_begin) * W + x;
    for (int y = y_begin; y < y_end; ++y, pix += W) {
      int tri;
      float v[kAcc];
      if (!fn(pix, T, tri, v)) continue;
      if (tri != run_tri) {
        run_flush(s_keys, s_vals, acc_img, run_tri, a);
        run_tri = tri;
      }
#pragma unroll
      for (int k = 0; k < kAcc; ++k) a[k] += v[k];
    }
    run_flush(s_keys, s_vals, acc_img, run_tri, a);
  }
  __syncthreads();

  // Drain: 16 lanes per slot (12 active) -> 48 contiguous bytes per triangle.
  for (int i = tid; i < kRunSlots * 16; i += kRunThreads) {
    const int slot = i >> 4, k = i & 15;
    const int tri = s_keys[slot];
    if (tri >= 0 && k < kAcc) atomicAdd(&acc_img[(size_t)tri * kAcc + k], s_vals[slot * kAcc + k]);
  }
}

template <class PixelFn>
inline int launch_accumulate_runs(const PixelFn &fn, int B, int T, int W, int H, float *acc,
                                  hipStream_t s) {
  const int regions_x = (W + kWave - 1) / kWave;
  const int regions_y = (H + kRunRegionH - 1) / kRunRegionH;
  const int per_image = regions_x * regions_y;
  const int n_regions = per_image * B;
  const int per_xcd = (n_regions + kXcds - 1) / kXcds;
  hipLaunchKernelGGL((k_accumulate_runs<PixelFn>), dim3((unsigned)(per_xcd * kXcds)),
                     dim3(kRunThreads), 0, s, fn, T, W, H, regions_x, per_image, n_regions,
                     per_xcd, acc);
  return check_launch();
}

struct F3 {
  float x, y, z;
};

}  // namespace mr
