// Forward Phong shading of one pixel from its G-buffer sample: shared by k_shade_forward
// (shade.hip) and the shading epilogue of k_raster (raster_forward.hip).
#pragma once

#include "corner_rec.h"

namespace mr {

constexpr float kNormEps = 1e-12f;  // torch.nn.functional.normalize default eps

struct Lights {
  const float *__restrict__ pos;  // [B,L,3]
  const float *__restrict__ col;  // [B,L,3]
  const float *__restrict__ amb;  // [B,3] or nullptr
  int L;
};

// 1-ulp hardware reciprocal / square root (v_rcp_f32, v_sqrt_f32): the IEEE-exact sequences
// are ~10 VALU ops each and these kernels are VALU-bound; the parity budget is 1e-4 absolute.
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float fast_sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
// 1 / max(sqrt(s), eps) of torch.nn.functional.normalize for a squared length s, as ONE transcendental
// (v_rsq_f32, 1 ulp) instead of v_sqrt + v_rcp: they issue at a third of the plain vector rate.
// s = 0 -> rsq = inf -> 1 / eps; s = inf -> 0; NaN propagates.  `norm > eps` is `s > eps^2`.
constexpr float kInvNormEps = 1.0f / kNormEps, kNormEpsSquared = kNormEps * kNormEps;
__device__ __forceinline__ float inv_norm(float s) { return fminf(__builtin_amdgcn_rsqf(s), kInvNormEps); }

// alpha = clamp(sum(2*bary), 0, 1); attr = alpha * interp + (1 - alpha) * (-1)
// (rasterize.py:137-150 with render.py:197's background of -1).
__device__ __forceinline__ void interpolate9(const Corners &cr, const F3 b, float &pre, float &alpha,
                                             float (&interp)[9], float (&attr)[9]) {
#pragma clang fp contract(fast)
  pre = (2.0f * b.x + 2.0f * b.y) + 2.0f * b.z;
  alpha = fminf(fmaxf(pre, 0.0f), 1.0f);
  const float one_m = 1.0f - alpha;
#pragma unroll
  for (int a = 0; a < 9; ++a) {
    interp[a] = (cr.c[0][a] * b.x + cr.c[1][a] * b.y) + cr.c[2][a] * b.z;
    attr[a] = alpha * interp[a] + one_m * -1.0f;
  }
}

constexpr int kMaxLights = 4;      // lights the kernels keep in registers / unrolled instantiations (1..4)
constexpr int kMaxLightsAny = 32;  // round 3: up to this many through a run-time loop (the reference takes any
                                   // count, src/mesh_renderer/render.py:304-323); LDS stage of the forward epilogue

// Wave-uniform read-only data read through the constant address space: scalar loads (lgkmcnt).
typedef const __attribute__((address_space(4))) float *ConstFloats;

// Where shade_attributes reads the image's light parameters from.
struct LightsInMemory {  // straight from the [B,L,3] / [B,3] arrays (wave-uniform addresses)
  const Lights &l;
  int img;
  __device__ __forceinline__ int count() const { return l.L; }
  __device__ __forceinline__ bool has_ambient() const { return l.amb != nullptr; }
  __device__ __forceinline__ float ambient(int k) const { return l.amb[(size_t)img * 3 + k]; }
  __device__ __forceinline__ float position(int i, int k) const { return l.pos[((size_t)img * l.L + i) * 3 + k]; }
  __device__ __forceinline__ float color(int i, int k) const { return l.col[((size_t)img * l.L + i) * 3 + k]; }
};

// Read once per workgroup through the constant address space (scalar loads) and kept in scalar
// registers: for k_raster's tile loop, where a vector load's wait would drain the G-buffer stores
// in flight (vmcnt is in order) and a scalar load per tile would expose its latency.
struct LightsInRegisters {
  int n;
  bool amb_on;
  float amb[3], pos[kMaxLights][3], col[kMaxLights][3];
  __device__ __forceinline__ void load(const Lights &l, int img) {
    n = l.L;
    amb_on = l.amb != nullptr;
#pragma unroll
    for (int k = 0; k < 3; ++k) amb[k] = amb_on ? ((ConstFloats)(uintptr_t)l.amb)[(size_t)img * 3 + k] : 0.0f;
#pragma unroll
    for (int i = 0; i < kMaxLights; ++i)
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const size_t at = ((size_t)img * l.L + min(i, l.L - 1)) * 3 + k;
        pos[i][k] = ((ConstFloats)(uintptr_t)l.pos)[at];
        col[i][k] = ((ConstFloats)(uintptr_t)l.col)[at];
      }
  }
  __device__ __forceinline__ int count() const { return n; }
  __device__ __forceinline__ bool has_ambient() const { return amb_on; }
  __device__ __forceinline__ float ambient(int k) const { return amb[k]; }
  __device__ __forceinline__ float position(int i, int k) const { return pos[i][k]; }
  __device__ __forceinline__ float color(int i, int k) const { return col[i][k]; }
};

// Staged once per workgroup in LDS ([0..2] ambient, then per light 3 position + 3 colour floats) and
// read per tile with wave-uniform (broadcast) ds_reads: no scalar registers held across the tile
// walk (they were spilled to VGPR lanes and read back with v_readlane), no per-tile memory latency.
struct LightsInLds {
  const float *s;  // LDS
  int n;
  bool amb_on;
  static constexpr int kFloats = 3 + 6 * kMaxLightsAny;
  // all threads of the workgroup call this (at least 3 + 6 L of them: 195 for 32 lights); the caller
  // provides the barrier before the first use
  __device__ __forceinline__ static void stage(const Lights &l, int img, float *lds, int tid) {
    if (tid < 3) lds[tid] = l.amb ? ((ConstFloats)(uintptr_t)l.amb)[(size_t)img * 3 + tid] : 0.0f;
    if (tid >= 3 && tid < 3 + 6 * l.L) {
      const int i = (tid - 3) / 6, k = (tid - 3) % 6;
      const float *src = k < 3 ? l.pos : l.col;
      lds[tid] = ((ConstFloats)(uintptr_t)src)[((size_t)img * l.L + i) * 3 + (k % 3)];
    }
  }
  __device__ __forceinline__ int count() const { return n; }
  __device__ __forceinline__ bool has_ambient() const { return amb_on; }
  __device__ __forceinline__ float ambient(int k) const { return s[k]; }
  __device__ __forceinline__ float position(int i, int k) const { return s[3 + 6 * i + k]; }
  __device__ __forceinline__ float color(int i, int k) const { return s[6 + 6 * i + k]; }
};

// Shading of one covered pixel from its blended attributes `at` (render.py:201-215, 298-323).
template <class LightSet>
__device__ __forceinline__ float4 shade_attributes(const float (&at)[9], const LightSet &lights) {
#pragma clang fp contract(fast)  // also inside raster_forward.hip, which is built with -ffp-contract=off
  const bool mask = (at[6] >= 0.0f) || (at[7] >= 0.0f) || (at[8] >= 0.0f);  // render.py:215
  if (!mask) return make_float4(0.f, 0.f, 0.f, 0.f);
  const float inv_nn = inv_norm(at[0] * at[0] + at[1] * at[1] + at[2] * at[2]);
  const float nx = at[0] * inv_nn, ny = at[1] * inv_nn, nz = at[2] * inv_nn;
  float r = 0.f, g = 0.f, bl = 0.f;
  if (lights.has_ambient()) {  // render.py:298-301
    r = lights.ambient(0) * at[6];
    g = lights.ambient(1) * at[7];
    bl = lights.ambient(2) * at[8];
  }
  auto add_light = [&](const int l) {  // render.py:304-323
    const float vx = lights.position(l, 0) - at[3], vy = lights.position(l, 1) - at[4],
                vz = lights.position(l, 2) - at[5];
    const float inv_vn = inv_norm(vx * vx + vy * vy + vz * vz);
    const float ndl = fminf(fmaxf(nx * (vx * inv_vn) + ny * (vy * inv_vn) + nz * (vz * inv_vn), 0.0f), 1.0f);
    r += at[6] * ndl * lights.color(l, 0);
    g += at[7] * ndl * lights.color(l, 1);
    bl += at[8] * ndl * lights.color(l, 2);
  };
#pragma unroll
  for (int l = 0; l < kMaxLights; ++l) {  // the first four unrolled, as before
    if (l >= lights.count()) break;
    add_light(l);
  }
#pragma unroll 1
  for (int l = kMaxLights; l < lights.count(); ++l) add_light(l);  // any further ones (round 3), same order
  return make_float4(r, g, bl, 1.0f);
}

__device__ __forceinline__ float4 shade_pixel(const Corners &cr, const F3 b, const Lights &lights, int img) {
  float pre, alpha, interp[9], at[9];
  interpolate9(cr, b, pre, alpha, interp, at);
  return shade_attributes(at, LightsInMemory{lights, img});
}

}  // namespace mr
