// The per-pixel geometry of the specular term (render.py:304-341): shared by the specular shading kernels
// (shade_spec.hip) and by the rasterizer's norm epilogue (raster_forward.hip, k_raster<..., NORMS>), which forms the
// across-pixels norm of `rdc` in the same pass that writes the G-buffer.
#pragma once

namespace mr {
namespace spec {

constexpr float kPixelNormEps = 1e-12f;   // torch.nn.functional.normalize default eps
__device__ __forceinline__ float px_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float px_sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }

// Geometry of one pixel that does not depend on the light.
struct PixelFrame {
  float N[3], nn, inv_nn;     // normalised normal (render.py:201)
  float Cd[3], cn, inv_cn;    // direction to the camera (render.py:333-336)
};
__device__ __forceinline__ void pixel_frame(const float *at, const float *cam, PixelFrame &f) {
#pragma clang fp contract(fast)
  f.nn = px_sqrt(at[0] * at[0] + at[1] * at[1] + at[2] * at[2]);
  f.inv_nn = px_rcp(fmaxf(f.nn, kPixelNormEps));
  const float c[3] = {cam[0] - at[3], cam[1] - at[4], cam[2] - at[5]};
  f.cn = px_sqrt(c[0] * c[0] + c[1] * c[1] + c[2] * c[2]);
  f.inv_cn = px_rcp(fmaxf(f.cn, kPixelNormEps));
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    f.N[k] = at[k] * f.inv_nn;
    f.Cd[k] = c[k] * f.inv_cn;
  }
}

// One light at one pixel (render.py:304-341).
struct LightTerm {
  float D[3], vn, inv_vn, pre, ndl;   // direction to the light, N . D and its clamp
  float M[3], mn, inv_mn;             // mirror reflection direction
  float rdc;                          // M . Cd, BEFORE the across-pixels normalisation
};
__device__ __forceinline__ void light_term(const float *at, const PixelFrame &f, const float *lp, LightTerm &o) {
#pragma clang fp contract(fast)
  const float v[3] = {lp[0] - at[3], lp[1] - at[4], lp[2] - at[5]};
  o.vn = px_sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
  o.inv_vn = px_rcp(fmaxf(o.vn, kPixelNormEps));
#pragma unroll
  for (int k = 0; k < 3; ++k) o.D[k] = v[k] * o.inv_vn;
  o.pre = f.N[0] * o.D[0] + f.N[1] * o.D[1] + f.N[2] * o.D[2];
  o.ndl = fminf(fmaxf(o.pre, 0.0f), 1.0f);
  float m[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) m[k] = 2.0f * o.ndl * f.N[k] - o.D[k];
  o.mn = px_sqrt(m[0] * m[0] + m[1] * m[1] + m[2] * m[2]);
  o.inv_mn = px_rcp(fmaxf(o.mn, kPixelNormEps));
#pragma unroll
  for (int k = 0; k < 3; ++k) o.M[k] = m[k] * o.inv_mn;
  o.rdc = o.M[0] * f.Cd[0] + o.M[1] * f.Cd[1] + o.M[2] * f.Cd[2];
}

}  // namespace spec
}  // namespace mr
