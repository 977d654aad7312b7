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
__device__ __forceinline__ void gather_corner_values(const F3 *__restrict__ normals, const F3 *__restrict__ positions,
                                                     const F3 *__restrict__ diffuse, const int32_t *__restrict__ tris,
                                                     int b, int t, int V, float (&v)[32]) {
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
}
__device__ __forceinline__ void fill_corner_record(const F3 *__restrict__ normals, const F3 *__restrict__ positions,
                                                   const F3 *__restrict__ diffuse, const int32_t *__restrict__ tris,
                                                   int b, int t, int V, CornerRec *__restrict__ out) {
  float v[32];
  gather_corner_values(normals, positions, diffuse, tris, b, t, V, v);
#pragma unroll
  for (int q = 0; q < 8; ++q) out->q[q] = make_float4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
}

// Round 4: the record of the shading backward's folded lane kernel (ShadeFoldLaneFn, shade.hip), written by
// k_bwd_setup from the CornerRec and the clip-space corners: the attributes in a DIFFERENCE basis --
// e0 = c0 - c2, e1 = c1 - c2, c2 -- so that a pixel interpolates with 18 multiply-adds instead of 27
// (at = c2 + b0 e0 + b1 e1: the rasterizer's barycentrics sum to 1) and the gradient through the
// barycentrics needs two dot products with e0 / e1 instead of three with the corners (the rasterizer's
// backward is invariant under a common shift of d L / d b: the brackets of cpp:202-269 sum to ~0 over the
// corners), and of the adjugate only the rows of corners 0 and 1.  37 floats in 160 bytes.
//   q[0..6]  e0[9] e1[9] c2[9] u0c[0] | q[7] u0c[1] u0c[2] u1c[0] u1c[1] | q[8] u1c[2] s[0] s[1] s[2] | q[9] 1/|det| - - -
struct alignas(32) FoldRec {
  float4 q[10];
};
struct FoldTriangle {
  float e0[9], e1[9], c2[9], u0[3], u1[3], s[3], inv;
};
__device__ __forceinline__ void load_fold_triangle(const FoldRec *__restrict__ rec, FoldTriangle &t) {
  float v[40];
#pragma unroll
  for (int q = 0; q < 10; ++q) {
    const float4 f = rec->q[q];
    v[4 * q] = f.x; v[4 * q + 1] = f.y; v[4 * q + 2] = f.z; v[4 * q + 3] = f.w;
  }
#pragma unroll
  for (int a = 0; a < 9; ++a) { t.e0[a] = v[a]; t.e1[a] = v[9 + a]; t.c2[a] = v[18 + a]; }
#pragma unroll
  for (int c = 0; c < 3; ++c) { t.u0[c] = v[27 + c]; t.u1[c] = v[30 + c]; t.s[c] = v[33 + c]; }
  t.inv = v[36];
}

// FoldRec from a corner record's 27 values c[corner * 9 + attribute], the sign-corrected adjugate u[9] (row i =
// edge i, column c = clip component x / y / w) and 1 / |det| (rasterize_triangles.cpp:180-198).
//
// pull (round 5; nullptr = the layout above): the image's clip-space transform rows x / y / w, pull[r * 4 + c'] =
// M[{0, 1, 3}[r]][c'] -- the record then carries the rasterizer's backward ALREADY pulled back to world space, for
// the kernel that wants the whole vertex gradient and nothing else (ShadeFoldLaneFn).  With g0 = dL/db0 - dL/db2,
// g1 = dL/db1 - dL/db2 the clip-space bracket of a pixel is q_c = (g0 (s_c b0 - u_0c) + g1 (s_c b1 - u_1c)) / |det|
// and its pull-back (M^T q)_c' = (g0 b0 + g1 b1) S_c' + g0 P0_c' + g1 P1_c' with the per-TRIANGLE vectors
//   S_c' = sum_r pull[r][c'] s_r / |det|,  P0_c' = -sum_r pull[r][c'] u_0r / |det|,  P1_c' = -sum_r pull[r][c'] u_1r / |det|:
// 11 multiply-adds per pixel instead of 24, and still one cancellation PER PIXEL between the b S and the P terms
// (nothing is summed over pixels before it).  Slots 27..35 = S, P0, P1; the record's tenth quad is unused.
__device__ __forceinline__ void store_fold_record(const float (&c)[32], const float (&u)[9], float inv_abs_det,
                                                  FoldRec *__restrict__ out, const float *__restrict__ pull = nullptr) {
  float v[40];
#pragma unroll
  for (int a = 0; a < 9; ++a) {
    v[a] = c[a] - c[18 + a];
    v[9 + a] = c[9 + a] - c[18 + a];
    v[18 + a] = c[18 + a];
  }
  float s3[3];
#pragma unroll
  for (int c3 = 0; c3 < 3; ++c3) s3[c3] = (u[c3] + u[3 + c3]) + u[6 + c3];  // cpp:187-198
  if (pull) {
#pragma unroll
    for (int cw = 0; cw < 3; ++cw) {
      const float p0 = pull[cw], p1 = pull[4 + cw], p2 = pull[8 + cw];   // column c' of the x, y, w rows
      v[27 + cw] = ((p0 * s3[0] + p1 * s3[1]) + p2 * s3[2]) * inv_abs_det;
      v[30 + cw] = -(((p0 * u[0] + p1 * u[1]) + p2 * u[2]) * inv_abs_det);
      v[33 + cw] = -(((p0 * u[3] + p1 * u[4]) + p2 * u[5]) * inv_abs_det);
    }
    v[36] = 0.f;
  } else {
#pragma unroll
    for (int k = 0; k < 6; ++k) v[27 + k] = u[k];
#pragma unroll
    for (int c3 = 0; c3 < 3; ++c3) v[33 + c3] = s3[c3];
    v[36] = inv_abs_det;
  }
  v[37] = 0.f; v[38] = 0.f; v[39] = 0.f;
#pragma unroll
  for (int q = 0; q < 10; ++q) out->q[q] = make_float4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
}
// the three transform rows a pulled record needs, as 12 floats (rows x, y, w of image b's [4][4] matrix)
__device__ __forceinline__ void load_pull_rows(const float *__restrict__ transforms, int b, float (&pull)[12]) {
  const float *m = transforms + (size_t)b * 16;
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c) pull[r * 4 + c] = m[(r == 2 ? 3 : r) * 4 + c];
}
struct FoldTriangleW {   // the pulled record: see store_fold_record
  float e0[9], e1[9], c2[9], S[3], P0[3], P1[3];
};
__device__ __forceinline__ void load_fold_triangle_w(const FoldRec *__restrict__ rec, FoldTriangleW &t) {
  float v[36];
#pragma unroll
  for (int q = 0; q < 9; ++q) {
    const float4 f = rec->q[q];
    v[4 * q] = f.x; v[4 * q + 1] = f.y; v[4 * q + 2] = f.z; v[4 * q + 3] = f.w;
  }
#pragma unroll
  for (int a = 0; a < 9; ++a) { t.e0[a] = v[a]; t.e1[a] = v[9 + a]; t.c2[a] = v[18 + a]; }
#pragma unroll
  for (int c = 0; c < 3; ++c) { t.S[c] = v[27 + c]; t.P0[c] = v[30 + c]; t.P1[c] = v[33 + c]; }
}

// The block mr_render_forward can prepare for the folded shading backward (include/mesh_raster.h,
// `backward_prepared`): FoldRec[B*T], then the compact accumulator rows [B*T][kFoldAccStride] floats.
constexpr int kFoldAccStride = 12;  // floats per (image, triangle) accumulator row of the folded variant: 48 bytes
inline size_t fold_prepared_recs_bytes(int B, int T) { return align_up((size_t)B * T * sizeof(FoldRec), 256); }
inline size_t fold_prepared_bytes(int B, int T) {
  return fold_prepared_recs_bytes(B, T) + align_up((size_t)B * T * kFoldAccStride * sizeof(float), 256);
}

// Fills CornerRec[B*T] from the three [B,V,3] attribute arrays (defined in shade.hip).
int launch_corner_setup(const float *normals, const float *positions, const float *diffuse,
                        const int32_t *tris, int B, int V, int T, CornerRec *out, hipStream_t s);

}  // namespace mr
