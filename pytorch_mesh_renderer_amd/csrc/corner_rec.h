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

// The record of (image b, triangle t): its corners' (normal, position, diffuse), 27 floats + 5 of
// padding.  A corner index outside [0, V) reads vertex 0 (such a triangle is never rasterized).
__device__ __forceinline__ void fill_corner_record(const F3 *__restrict__ normals, const F3 *__restrict__ positions,
                                                   const F3 *__restrict__ diffuse, const int32_t *__restrict__ tris,
                                                   int b, int t, int V, CornerRec *__restrict__ out) {
  float v[32];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    int vi = tris[3 * t + k];
    if ((unsigned)vi >= (unsigned)V) vi = 0;
    const size_t at = (size_t)b * V + vi;
    const F3 n = normals[at], p = positions[at], d = diffuse[at];
    v[k * 9 + 0] = n.x; v[k * 9 + 1] = n.y; v[k * 9 + 2] = n.z;
    v[k * 9 + 3] = p.x; v[k * 9 + 4] = p.y; v[k * 9 + 5] = p.z;
    v[k * 9 + 6] = d.x; v[k * 9 + 7] = d.y; v[k * 9 + 8] = d.z;
  }
#pragma unroll
  for (int i = 27; i < 32; ++i) v[i] = 0.f;
#pragma unroll
  for (int q = 0; q < 8; ++q) out->q[q] = make_float4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
}

// Fills CornerRec[B*T] from the three [B,V,3] attribute arrays (defined in shade.hip).
int launch_corner_setup(const float *normals, const float *positions, const float *diffuse,
                        const int32_t *tris, int B, int V, int T, CornerRec *out, hipStream_t s);

}  // namespace mr
