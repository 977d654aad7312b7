// Backward of the barycentric rasterizer for gfx950 (MI355X).
//
// Replaces rasterize_triangles_backward
// (reference: src/mesh_renderer/kernels/rasterize_triangles.cpp:131-273), a serial
// loop that recomputes a 3x3 adjugate per pixel and does nine '+=' into
// df_dvertices.  Here the per-pixel work is reduced to what actually depends on
// the pixel.  With U the sign-corrected adjugate, S_k its column sums and
// b / g the pixel's barycentrics / upstream gradient:
//
//   dL/dM[k][j] = sum_px sum_i g_i * ( -U[i][k] b_j + S_k b_i b_j ) / |det|
//               = ( S_k * C_j  -  sum_i U[i][k] * A_ij ) / |det|
//   A_ij = sum_px g_i b_j   (9 numbers per triangle)
//   C_j  = sum_px (g . b) b_j   (3 numbers per triangle)
//
// k_accumulate_runs (run_accum.h) streams the G-buffer once (28 B/px: dbary 12 + id 4 + bary 12).
//               Each lane walks DOWN one pixel column, so every wave load is 64
//               consecutive pixels (coalesced), and keeps the 12 sums in registers
//               while the triangle id does not change (runs of tens of pixels).
//               A finished run is added into a workgroup-local LDS hash table keyed
//               by triangle id (ds_add_f32); the table is drained with 48-byte
//               contiguous global float atomics into acc[B][T][12].
// k_finalize    one thread per (image, triangle) with a non-zero accumulator:
//               rebuilds U once and scatters the 9 vertex partials into dclip.
//
// The sums are associated differently from the reference's row-major serial loop
// (fp32, order-dependent there too), so parity is to 1e-4 abs, not bitwise.
#include "run_accum.h"

namespace mr {
namespace {

constexpr int kThreads = 256;
constexpr float kDegenerateCutoff = 0.9f;  // cpp:13

// Per-pixel values for the rasterizer backward: A_ij = g_i b_j, C_j = (g.b) b_j.
struct RasterGradFn {
  const F3 *__restrict__ dbary;
  const int32_t *__restrict__ ids;
  const F3 *__restrict__ bary;
  __device__ __forceinline__ bool operator()(size_t pix, int T, int &tri, float (&v)[kAcc]) const {
    const int t = ids[pix];
    const F3 b = bary[pix];
    if ((unsigned)t >= (unsigned)T) return false;                           // foreign id
    if (t == 0 && (b.x + b.y) + b.z < kDegenerateCutoff) return false;      // cpp:162
    const F3 g = dbary[pix];
    const float gb = g.x * b.x + g.y * b.y + g.z * b.z;
    v[0] = g.x * b.x; v[1] = g.x * b.y; v[2] = g.x * b.z;
    v[3] = g.y * b.x; v[4] = g.y * b.y; v[5] = g.y * b.z;
    v[6] = g.z * b.x; v[7] = g.z * b.y; v[8] = g.z * b.z;
    v[9] = gb * b.x; v[10] = gb * b.y; v[11] = gb * b.z;
    tri = t;
    return true;
  }
};

__global__ __launch_bounds__(kThreads) void k_finalize(
    const float *__restrict__ acc, const float4 *__restrict__ clip,
    const int32_t *__restrict__ tris, int B, int V, int T, float *__restrict__ dclip) {
  const long gid = (long)blockIdx.x * kThreads + threadIdx.x;
  if (gid >= (long)B * T) return;
  float a[kAcc];
  bool any = false;
#pragma unroll
  for (int k = 0; k < kAcc; ++k) {
    a[k] = acc[gid * kAcc + k];
    any |= (a[k] != 0.0f);  // NaN counts as touched
  }
  if (!any) return;  // no pixel of this triangle survived (or its gradient is 0)
  const int b = (int)(gid / T);
  const int t = (int)(gid - (long)b * T);
  const int vi[3] = {tris[3 * t], tris[3 * t + 1], tris[3 * t + 2]};
  float xs[3], ys[3], ws[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    if ((unsigned)vi[k] >= (unsigned)V) return;
    const float4 p = clip[(long)b * V + vi[k]];
    xs[k] = p.x; ys[k] = p.y; ws[k] = p.w;
  }
  const float a11 = xs[0], a12 = xs[1], a13 = xs[2];
  const float a21 = ys[0], a22 = ys[1], a23 = ys[2];
  const float a31 = ws[0], a32 = ws[1], a33 = ws[2];
  float u[9];
  u[0] = a22 * a33 - a32 * a23; u[1] = a13 * a32 - a33 * a12; u[2] = a12 * a23 - a22 * a13;
  u[3] = a23 * a31 - a33 * a21; u[4] = a11 * a33 - a31 * a13; u[5] = a13 * a21 - a23 * a11;
  u[6] = a21 * a32 - a31 * a22; u[7] = a12 * a31 - a32 * a11; u[8] = a11 * a22 - a21 * a12;
  const float det = a11 * u[0] + a12 * u[3] + a13 * u[6];
  if (det < 0.0f) {
#pragma unroll
    for (int k = 0; k < 9; ++k) u[k] = -u[k];
  }
  const float inv_abs_det = 1.0f / fabsf(det);
  const int out_col[3] = {0, 1, 3};  // x, y, w; the z column never receives gradient
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float colsum = u[c] + u[3 + c] + u[6 + c];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const float v = (colsum * a[9 + j] - (u[c] * a[j] + u[3 + c] * a[3 + j] + u[6 + c] * a[6 + j])) *
                      inv_abs_det;
      atomicAdd(&dclip[((long)b * V + vi[j]) * 4 + out_col[c]], v);
    }
  }
}

}  // namespace

size_t raster_backward_ws(int B, int V, int T, int W, int H) {
  (void)V; (void)W; (void)H;
  return align_up((size_t)B * T * kAcc * sizeof(float), 256);
}

int launch_raster_backward(const float *dbary, const float *clip, const int32_t *tris,
                           const int32_t *ids, const float *bary, int B, int V, int T, int W,
                           int H, float *dclip, void *ws, hipStream_t s) {
  float *acc = (float *)ws;
  if (B == 0 || V == 0) return MR_OK;
  if (hipMemsetAsync(dclip, 0, (size_t)B * V * 4 * sizeof(float), s) != hipSuccess) return check_launch();
  if (T == 0) return MR_OK;
  if (hipMemsetAsync(acc, 0, (size_t)B * T * kAcc * sizeof(float), s) != hipSuccess) return check_launch();
  RasterGradFn fn{(const F3 *)dbary, ids, (const F3 *)bary};
  int rc = launch_accumulate_runs(fn, B, T, W, H, acc, s);
  if (rc != MR_OK) return rc;
  const long nbt = (long)B * T;
  hipLaunchKernelGGL(k_finalize, dim3((unsigned)((nbt + kThreads - 1) / kThreads)), dim3(kThreads),
                     0, s, acc, (const float4 *)clip, tris, B, V, T, dclip);
  return check_launch();
}

}  // namespace mr
