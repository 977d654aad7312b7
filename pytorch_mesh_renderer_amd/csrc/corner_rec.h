// Per-(image, triangle) corner attribute records shared by the fused shading kernels
// (shade.hip) and the SoftRas kernels (soft.hip).
#pragma once

#include "run_accum.h"

namespace mr {

struct Corners {  // the three corners' (normal, position, diffuse): 27 floats
  float c[3][9];
};

// Corners of one (image, triangle), gathered once by k_corner_setup so that the per-pixel
// kernels follow ONE pointer (id -> 128-byte record) instead of two (id -> vertex ids ->
// nine scattered 12-byte reads).  128 bytes, 128-byte aligned = one cache line.
struct alignas(128) CornerRec {
  float4 q[8];  // 27 floats used, row-major [corner][attribute]
};

__device__ __forceinline__ void load_corners(const CornerRec *__restrict__ rec, Corners &o) {
  float v[28];
#pragma unroll
  for (int q = 0; q < 7; ++q) {
    const float4 f = rec->q[q];
    v[4 * q] = f.x; v[4 * q + 1] = f.y; v[4 * q + 2] = f.z; v[4 * q + 3] = f.w;
  }
#pragma unroll
  for (int k = 0; k < 3; ++k)
#pragma unroll
    for (int a = 0; a < 9; ++a) o.c[k][a] = v[k * 9 + a];
}

// Fills CornerRec[B*T] from the three [B,V,3] attribute arrays (defined in shade.hip).
int launch_corner_setup(const float *normals, const float *positions, const float *diffuse,
                        const int32_t *tris, int B, int V, int T, CornerRec *out, hipStream_t s);

}  // namespace mr
