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
// The functor supplies the per-pixel values:
//   struct Fn {
//     static constexpr int kN = ...;          // sums per triangle (<= 12)
//     struct Pixel {...}; struct Triangle {...};
//     __device__ bool load_pixel(size_t pix, int T, int &tri, Pixel &p) const;  // false: skip
//     __device__ void load_triangle(int img, int tri, Triangle &t) const;       // on run change
//     __device__ void accumulate(const Pixel &p, const Triangle &t, float (&a)[kN]) const;
//   };
// Per-triangle data (e.g. the adjugate) is fetched once per run, not per pixel.
#pragma once

#include "mr_internal.h"

namespace mr {

constexpr int kRunThreads = 256;
constexpr int kRunRowsPerWave = 32;                                  // pixels each lane walks
constexpr int kRunRegionH = kRunRowsPerWave * (kRunThreads / kWave);  // 128 rows / workgroup
constexpr int kRunSlots = 512;                                       // LDS hash slots
constexpr int kRunSlotsLog2 = 9;
constexpr int kAccStride = 12;  // floats per (image, triangle) row of acc[]: 48 B
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

template <int N>
__device__ __forceinline__ void run_flush(int *keys, float *vals, float *acc_img, int tri,
                                          float (&a)[N]) {
  if (tri < 0) return;
  const int slot = run_find_slot(keys, tri);
  if (slot >= 0) {
#pragma unroll
    for (int k = 0; k < N; ++k) atomicAdd(&vals[slot * N + k], a[k]);
  } else {  // table saturated (very dense mesh): straight to HBM
#pragma unroll
    for (int k = 0; k < N; ++k) atomicAdd(&acc_img[(size_t)tri * kAccStride + k], a[k]);
  }
#pragma unroll
  for (int k = 0; k < N; ++k) a[k] = 0.0f;
}

template <class Fn>
__global__ __launch_bounds__(kRunThreads) void k_accumulate_runs(
    Fn fn, int T, int W, int H, int regions_x, int regions_per_image, int n_regions,
    int regions_per_xcd, float *__restrict__ acc) {
  constexpr int N = Fn::kN;
  static_assert(N <= kAccStride, "accumulator row too small");
  __shared__ int s_keys[kRunSlots];
  __shared__ float s_vals[kRunSlots * N];

  const int region = xcd_contiguous_block((int)blockIdx.x, n_regions, regions_per_xcd);
  if (region < 0) return;
  const int img = region / regions_per_image;
  const int rr = region - img * regions_per_image;
  const int ry = rr / regions_x;
  const int rx = rr - ry * regions_x;

  const int tid = (int)threadIdx.x;
  for (int i = tid; i < kRunSlots; i += kRunThreads) s_keys[i] = -1;
  for (int i = tid; i < kRunSlots * N; i += kRunThreads) s_vals[i] = 0.0f;
  __syncthreads();

  const int lane = tid & (kWave - 1);
  const int wave = tid >> 6;
  const int x = rx * kWave + lane;
  const int y_begin = ry * kRunRegionH + wave * kRunRowsPerWave;
  const int y_end = min(y_begin + kRunRowsPerWave, H);
  float *acc_img = acc + (size_t)img * T * kAccStride;

  if (x < W) {
    float a[N];
#pragma unroll
    for (int k = 0; k < N; ++k) a[k] = 0.0f;
    int run_tri = -1;
    typename Fn::Triangle tri_data;
    size_t pix = ((size_t)img * H + y_begin) * W + x;
    for (int y = y_begin; y < y_end; ++y, pix += W) {
      int tri;
      typename Fn::Pixel p;
      if (!fn.load_pixel(pix, T, tri, p)) continue;
      if (tri != run_tri) {
        run_flush<N>(s_keys, s_vals, acc_img, run_tri, a);
        run_tri = tri;
        fn.load_triangle(img, tri, tri_data);
      }
      fn.accumulate(p, tri_data, a);
    }
    run_flush<N>(s_keys, s_vals, acc_img, run_tri, a);
  }
  __syncthreads();

  // Drain: 16 lanes per slot (N active) -> one contiguous <=48-byte row per triangle.
  for (int i = tid; i < kRunSlots * 16; i += kRunThreads) {
    const int slot = i >> 4, k = i & 15;
    const int tri = s_keys[slot];
    if (tri >= 0 && k < N) atomicAdd(&acc_img[(size_t)tri * kAccStride + k], s_vals[slot * N + k]);
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
