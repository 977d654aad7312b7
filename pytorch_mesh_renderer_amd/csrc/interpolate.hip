// Deferred attribute interpolation for gfx950 (MI355X), forward and backward.
//
// Replaces the eager-torch block of rasterize_clip_space
// (reference: src/mesh_renderer/rasterize.py:118-150): the [B*H*W,3,A] corner
// gather (3.6 GB at 1024^2 x 32, A=9), the multiply / sum, the alpha clamp and
// the background blend -- and the backward autograd derives from it
// (index_put_(accumulate=True) into the attributes, plus d/d barycentrics).
//
//   k_interp_forward   one thread per OUTPUT ELEMENT (pixel, attribute): stores are
//                      fully coalesced; the id / barycentric / corner loads of the
//                      A threads of a pixel hit the same L1 lines.
//   k_interp_dbary     one thread per pixel: dL/dbary (12 B/px written).
//   attribute grads    per-triangle run accumulation (run_accum.h) in chunks of 4
//                      attributes x 3 corners = 12 sums, then one thread per touched
//                      (image, triangle, chunk) scatters into dattrs[B,V,A].
#include "run_accum.h"

namespace mr {
namespace {

constexpr int kThreads = 256;
constexpr int kChunk = 4;  // attributes per accumulation pass (x3 corners = kAcc)
constexpr int kAcc = kChunk * 3;
constexpr int kAccStride = 12;  // floats per acc row

// alpha = clamp(sum(2*bary), 0, 1)  (rasterize.py:145-147)
__device__ __forceinline__ float coverage_alpha(const F3 b, float &pre_clamp) {
  pre_clamp = (2.0f * b.x + 2.0f * b.y) + 2.0f * b.z;
  return fminf(fmaxf(pre_clamp, 0.0f), 1.0f);
}

__global__ __launch_bounds__(kThreads) void k_interp_forward(
    const int32_t *__restrict__ ids, const F3 *__restrict__ bary, const float *__restrict__ attrs,
    const int32_t *__restrict__ tris, const float *__restrict__ background, size_t n_elems,
    size_t px_per_image, int V, int T, int A, float *__restrict__ out) {
  for (size_t e = (size_t)blockIdx.x * kThreads + threadIdx.x; e < n_elems;
       e += (size_t)gridDim.x * kThreads) {
    const size_t pix = e / (unsigned)A;
    const int a = (int)(e - pix * (unsigned)A);
    const int img = (int)(pix / px_per_image);
    int t = ids[pix];
    if ((unsigned)t >= (unsigned)T) t = 0;
    const F3 b = bary[pix];
    const float *va = attrs + (size_t)img * V * A + a;
    const int i0 = tris[3 * t], i1 = tris[3 * t + 1], i2 = tris[3 * t + 2];
    const float c0 = va[(size_t)i0 * A], c1 = va[(size_t)i1 * A], c2 = va[(size_t)i2 * A];
    const float value = (c0 * b.x + c1 * b.y) + c2 * b.z;  // rasterize.py:137-141
    float pre;
    const float alpha = coverage_alpha(b, pre);
    out[e] = alpha * value + (1.0f - alpha) * background[a];  // rasterize.py:149-150
  }
}

__global__ __launch_bounds__(kThreads) void k_interp_dbary(
    const float *__restrict__ dout, const int32_t *__restrict__ ids, const F3 *__restrict__ bary,
    const float *__restrict__ attrs, const int32_t *__restrict__ tris,
    const float *__restrict__ background, size_t n_px, size_t px_per_image, int V, int T, int A,
    F3 *__restrict__ dbary) {
  for (size_t pix = (size_t)blockIdx.x * kThreads + threadIdx.x; pix < n_px;
       pix += (size_t)gridDim.x * kThreads) {
    const int img = (int)(pix / px_per_image);
    int t = ids[pix];
    if ((unsigned)t >= (unsigned)T) t = 0;
    const F3 b = bary[pix];
    float pre;
    const float alpha = coverage_alpha(b, pre);
    const float *v0 = attrs + ((size_t)img * V + tris[3 * t]) * A;
    const float *v1 = attrs + ((size_t)img * V + tris[3 * t + 1]) * A;
    const float *v2 = attrs + ((size_t)img * V + tris[3 * t + 2]) * A;
    const float *g = dout + pix * A;
    float d0 = 0.0f, d1 = 0.0f, d2 = 0.0f, dalpha = 0.0f;
    for (int a = 0; a < A; ++a) {
      const float c0 = v0[a], c1 = v1[a], c2 = v2[a];
      const float go = g[a];
      const float gv = alpha * go;  // d/d(value)
      d0 += gv * c0;
      d1 += gv * c1;
      d2 += gv * c2;
      const float value = (c0 * b.x + c1 * b.y) + c2 * b.z;
      dalpha += go * (value - background[a]);
    }
    // torch.clamp passes the gradient where min <= x <= max (inclusive)
    const float dpre = (pre >= 0.0f && pre <= 1.0f) ? 2.0f * dalpha : 0.0f;
    F3 r;
    r.x = d0 + dpre;
    r.y = d1 + dpre;
    r.z = d2 + dpre;
    dbary[pix] = r;
  }
}

// Per-pixel values for the attribute scatter-add: alpha * dout[a] * b_k for the 4
// attributes of one chunk and the 3 corners.
struct AttrGradFn {
  static constexpr int kN = kChunk * 3;
  static constexpr int kStride = kAccStride;
  static constexpr int kSlots = 512;
  static constexpr int kMinWavesPerSimd = 5;
  const float *__restrict__ dout;
  const int32_t *__restrict__ ids;
  const F3 *__restrict__ bary;
  int A, a_begin;

  struct Pixel {
    F3 b;
    float gv[kChunk];
  };
  struct Triangle {};
  struct Raw {
    F3 b;
    int t;
    float g[kChunk];
  };
  using Image = NoImageSums;

  __device__ __forceinline__ void begin_image(int, Image &) const {}
  __device__ __forceinline__ void end_image(int, Image &) const {}
  __device__ __forceinline__ void fetch(int, int, int, size_t pix, Raw &r) const {
    r.b = bary[pix];
    r.t = ids[pix];
    const float *g = dout + pix * A + a_begin;
#pragma unroll
    for (int c = 0; c < kChunk; ++c) r.g[c] = (a_begin + c < A) ? g[c] : 0.0f;  // uniform test
  }
  __device__ __forceinline__ bool prepare(const Raw &r, int T, int &tri, Pixel &p) const {
    float pre;
    const float alpha = coverage_alpha(r.b, pre);
    if (!(alpha > 0.0f)) return false;  // background: every term is alpha * ... = 0
    if ((unsigned)r.t >= (unsigned)T) return false;
    p.b = r.b;
#pragma unroll
    for (int c = 0; c < kChunk; ++c) p.gv[c] = alpha * r.g[c];
    tri = r.t;
    return true;
  }
  __device__ __forceinline__ void load_triangle(int, int, Triangle &) const {}
  __device__ __forceinline__ void accumulate(const Pixel &p, const Triangle &,
                                             float (&acc)[kN], Image &) const {
#pragma unroll
    for (int c = 0; c < kChunk; ++c) {
      acc[c * 3 + 0] += p.gv[c] * p.b.x;
      acc[c * 3 + 1] += p.gv[c] * p.b.y;
      acc[c * 3 + 2] += p.gv[c] * p.b.z;
    }
  }
};

__global__ __launch_bounds__(kThreads) void k_attr_finalize(
    const float *__restrict__ acc, const int32_t *__restrict__ tris, int B, int V, int T, int A,
    int a_begin, float *__restrict__ dattrs) {
  const long gid = (long)blockIdx.x * kThreads + threadIdx.x;
  if (gid >= (long)B * T) return;
  float s[kAcc];
  bool any = false;
#pragma unroll
  for (int k = 0; k < kAcc; ++k) {
    s[k] = acc[gid * kAccStride + k];
    any |= (s[k] != 0.0f);
  }
  if (!any) return;
  const int b = (int)(gid / T);
  const int t = (int)(gid - (long)b * T);
#pragma unroll
  for (int corner = 0; corner < 3; ++corner) {
    const int vi = tris[3 * t + corner];
    if ((unsigned)vi >= (unsigned)V) continue;
    float *dst = dattrs + ((size_t)b * V + vi) * A + a_begin;
#pragma unroll
    for (int c = 0; c < kChunk; ++c) {
      if (a_begin + c < A) atomicAdd(&dst[c], s[c * 3 + corner]);
    }
  }
}

inline unsigned capped_blocks(size_t n) {
  const size_t want = (n + kThreads - 1) / kThreads;
  const size_t cap = 256u * 16u;  // 256 CUs x 16 blocks: grid-stride the rest
  return (unsigned)(want < cap ? (want ? want : 1) : cap);
}

}  // namespace

int launch_interp_forward(const int32_t *ids, const float *bary, const float *attrs,
                          const int32_t *tris, const float *bg, int B, int V, int T, int W,
                          int H, int A, float *out, hipStream_t s) {
  const size_t px_per_image = (size_t)W * H;
  const size_t n_elems = px_per_image * B * A;
  if (n_elems == 0) return MR_OK;
  hipLaunchKernelGGL(k_interp_forward, dim3(capped_blocks(n_elems)), dim3(kThreads), 0, s, ids,
                     (const F3 *)bary, attrs, tris, bg, n_elems, px_per_image, V, T, A, out);
  return check_launch();
}

size_t interp_backward_ws(int B, int V, int T, int W, int H, int A) {
  (void)V; (void)W; (void)H; (void)A;
  return align_up((size_t)B * T * kAccStride * sizeof(float), 256);
}

int launch_interp_backward(const float *dout, const int32_t *ids, const float *bary,
                           const float *attrs, const int32_t *tris, const float *bg, int B,
                           int V, int T, int W, int H, int A, float *dattrs, float *dbary,
                           void *ws, hipStream_t s) {
  const size_t px_per_image = (size_t)W * H;
  const size_t n_px = px_per_image * B;
  if ((size_t)B * V * A > 0 &&
      zero_async(dattrs, (size_t)B * V * A * sizeof(float), s) != hipSuccess)
    return check_launch();
  if (n_px == 0 || A == 0) return MR_OK;
  hipLaunchKernelGGL(k_interp_dbary, dim3(capped_blocks(n_px)), dim3(kThreads), 0, s, dout, ids,
                     (const F3 *)bary, attrs, tris, bg, n_px, px_per_image, V, T, A, (F3 *)dbary);
  int rc = check_launch();
  if (rc != MR_OK || T == 0 || V == 0) return rc;
  float *acc = (float *)ws;
  const long nbt = (long)B * T;
  for (int a_begin = 0; a_begin < A; a_begin += kChunk) {
    if (zero_async(acc, (size_t)nbt * kAccStride * sizeof(float), s) != hipSuccess)
      return check_launch();
    AttrGradFn fn{dout, ids, (const F3 *)bary, A, a_begin};
    rc = launch_accumulate_runs(fn, B, T, W, H, acc, s);
    if (rc != MR_OK) return rc;
    hipLaunchKernelGGL(k_attr_finalize, dim3((unsigned)((nbt + kThreads - 1) / kThreads)),
                       dim3(kThreads), 0, s, acc, tris, B, V, T, A, a_begin, dattrs);
    rc = check_launch();
    if (rc != MR_OK) return rc;
  }
  return MR_OK;
}

}  // namespace mr
