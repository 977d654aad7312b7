// Backward of rasterize() -- attribute interpolation AND the rasterizer underneath it -- in ONE pass
// over the G-buffer, for gfx950 (MI355X).
//
// Replaces, for mesh_renderer.rasterize / rasterize_clip_space with up to 16 attributes, the
// autograd graph the reference builds behind src/mesh_renderer/rasterize.py:118-150 (corner
// gather, multiply / sum, alpha clamp, background blend: index_put_(accumulate) into the
// attributes and d/d barycentrics) chained into rasterize_triangles.cpp:131-273.  The composed
// kernels of interpolate.hip + raster_backward.hip need ceil(A / 4) + 2 passes (4.4 ms at
// 1024^2 x 32, A = 9); here the per-triangle sums are the outer product of the 3 barycentrics
// with (A attribute gradients x alpha, 3 clip brackets), exactly the shape the row kernel of
// run_accum.h reduces: each pixel parks 3 + A + 3 factors, 3 (A + 3) sums per triangle.
//
//   k_attr_corner_setup<AP>   one thread per (image, triangle): the three corners' attributes
//                             as one record (the per-pixel kernels follow ONE pointer)
//   k_accumulate_rows<AttrRowsFn<AP>>   reads dL/dout (4 A B/px) + id + barycentrics (16 B/px)
//   k_attr_gather<AP>         one thread per (image, vertex): sums its incident triangles' rows
//                             (CSR adjacency from the host side): dattributes, dclip
//
// AP = A rounded up to 4, 8, 12 or 16 (template parameter; the tail attributes are zeros).
#include "run_accum.h"

namespace mr {
extern thread_local int g_deterministic;  // mr_set_deterministic (shade.hip)
extern thread_local int g_shade_backward_kernel;  // mr_debug_set_shade_backward_kernel (shade.hip): 1 = rows kernels
namespace {

constexpr int kThreads = 256;
constexpr float kDegenerateCutoff = 0.9f;  // rasterize_triangles.cpp:13

template <int AP>
struct AttrCorners {
  float c[3][AP];
};

template <int AP>
__global__ __launch_bounds__(kThreads) void k_attr_corner_setup(
    const float *__restrict__ attrs, const int32_t *__restrict__ tris, int B, int V, int T, int A,
    float *__restrict__ out) {
  const long gid = (long)blockIdx.x * kThreads + threadIdx.x;
  if (gid >= (long)B * T) return;
  const int b = (int)(gid / T);
  const int t = (int)(gid - (long)b * T);
  float4 *rec = (float4 *)(out + gid * (3 * AP));  // 3 * AP floats, 16-byte aligned (AP % 4 == 0)
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    int vi = tris[3 * t + k];
    if ((unsigned)vi >= (unsigned)V) vi = 0;
    const float *src = attrs + ((size_t)b * V + vi) * A;
    float v[AP];
#pragma unroll
    for (int a = 0; a < AP; ++a) v[a] = a < A ? src[a] : 0.0f;
#pragma unroll
    for (int q = 0; q < AP / 4; ++q) rec[k * (AP / 4) + q] = make_float4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
  }
}

// Forward interpolation from the corner records: one thread per PIXEL (blockIdx.y = image: no
// 64-bit divisions; the thread-per-element form spent 1.1 ms at 1024^2 x 32, A = 9, on index
// arithmetic), two load levels (id -> record) instead of three (id -> vertex ids -> attributes).
// A wavefront's 64 pixels form one contiguous run of 64 A floats in `out`: the lanes park their A
// values in LDS and the run leaves as A fully coalesced 256-byte stores (each lane storing its own
// A floats directly -- 64 scattered dwords per store instruction -- was 3x slower).  rasterize.py:137-150.
template <int AP>
__global__ __launch_bounds__(kThreads) void k_interp_forward_rec(
    const int32_t *__restrict__ ids, const F3 *__restrict__ bary, const float *__restrict__ corners,
    const float *__restrict__ background, unsigned px_per_image, int T, int A, float *__restrict__ out) {
  __shared__ float s_stage[kThreads / kWave][kWave * AP];
  const int img = (int)blockIdx.y;
  const size_t img_px = (size_t)img * px_per_image;
  const unsigned lane = threadIdx.x & (kWave - 1);
  float *stage = s_stage[threadIdx.x >> 6];
  float bg[AP];
#pragma unroll
  for (int a = 0; a < AP; ++a) bg[a] = a < A ? background[a] : 0.0f;
  for (unsigned p0 = blockIdx.x * kThreads; p0 < px_per_image; p0 += gridDim.x * kThreads) {
    const unsigned p = p0 + threadIdx.x;
    if (p - lane >= px_per_image) break;            // whole wavefront beyond the image (wave-uniform)
    const size_t pix = img_px + min(p, px_per_image - 1);  // lanes past the end recompute the last pixel
    int t = ids[pix];
    if ((unsigned)t >= (unsigned)T) t = 0;
    const F3 b = bary[pix];
    const float pre = (2.0f * b.x + 2.0f * b.y) + 2.0f * b.z;
    const float alpha = fminf(fmaxf(pre, 0.0f), 1.0f);
    const float one_m = 1.0f - alpha;
    float c[3 * AP];
    const float4 *src = (const float4 *)(corners + ((size_t)img * T + t) * (3 * AP));
#pragma unroll
    for (int q = 0; q < 3 * AP / 4; ++q) {
      const float4 f = src[q];
      c[4 * q] = f.x; c[4 * q + 1] = f.y; c[4 * q + 2] = f.z; c[4 * q + 3] = f.w;
    }
    // park the pixel's A values, then the wavefront writes its run of 64 A floats, coalesced
    float *mine = stage + lane * A;
#pragma unroll
    for (int a = 0; a < AP; ++a) {
      if (a < A) {  // wave-uniform
        const float value = (c[a] * b.x + c[AP + a] * b.y) + c[2 * AP + a] * b.z;
        mine[a] = alpha * value + one_m * bg[a];
      }
    }
    __builtin_amdgcn_wave_barrier();  // LDS executes one wavefront's operations in order
    const unsigned wave_first = p - lane;                                  // first pixel of this wavefront
    const unsigned n_here = min(64u, px_per_image - wave_first) * (unsigned)A;  // floats in its run
    float *run = out + (img_px + wave_first) * A;
    for (unsigned i = lane; i < n_here; i += 64u) run[i] = stage[i];
    __builtin_amdgcn_wave_barrier();
  }
}

template <int AP>
struct AttrRowsFn {
  static constexpr int kN = 3 * AP + 9;          // [corner][attribute] partials + 9 clip partials
  static constexpr int kStride = (kN + 3) & ~3;  // floats per acc row
  static constexpr int kRowsPerWave = MR_ROWS_PER_WAVE;  // see run_accum.h
  static constexpr int kSlots = 256;
  static constexpr int kMinWavesPerSimd = AP <= 8 ? 4 : 3;
  static constexpr bool kCountBackground = false;
  // parked per pixel: b[3] | y[AP] = alpha * dL/dout | q[3] = clip brackets
  static constexpr int kFactors = AP + 6;
  // multiple of 4 (ds_write_b128) and an ODD multiple of 4 (bank-conflict-free row stride)
  static constexpr int kFactorStride = AP == 4 ? 12 : (AP <= 12 ? 20 : 28);
  static_assert(kFactors <= kFactorStride && kN <= kWave, "row layout");
  __device__ static void factor_pair(int o, int &ia, int &ib) {
    if (o < 3 * AP) { ia = o / AP; ib = 3 + o % AP; }
    else { ia = (o - 3 * AP) / 3; ib = 3 + AP + (o - 3 * AP) % 3; }
  }
  const float *__restrict__ dout;       // [B,H,W,A]
  const int32_t *__restrict__ ids;
  const F3 *__restrict__ bary;
  const float *__restrict__ corners;    // [B,T,3*AP]
  const BwdRec *__restrict__ recs;
  const float *__restrict__ background; // [A]
  int A, T_;

  struct Pixel {
    F3 b;
    float g[AP];
    int tri;
  };
  struct Raw {
    F3 b;
    int t;
    float g[AP];
  };
  struct Triangle {
    AttrCorners<AP> cr;
    BwdTriangle bt;
  };
  struct Image {
    int n_bg;  // unused (kCountBackground = false)
  };

  __device__ __forceinline__ void begin_image(int, Image &) const {}
  __device__ __forceinline__ void end_strip(int, int, Image &) const {}
  __device__ __forceinline__ void fetch(int, int, int, size_t pix, Raw &r) const {
    r.b = bary[pix];
    r.t = ids[pix];
    const float *g = dout + pix * A;
#pragma unroll
    for (int a = 0; a < AP; ++a) r.g[a] = a < A ? g[a] : 0.0f;  // wave-uniform test
  }
  __device__ __forceinline__ bool prepare(const Raw &r, int T, int &tri, Pixel &p) const {
    const float pre = (2.0f * r.b.x + 2.0f * r.b.y) + 2.0f * r.b.z;
    // alpha == 0: every attribute term is alpha * ... = 0 and the rasterizer skips the pixel
    // (only triangle 0 can own a pixel with an all-zero barycentric sum, cpp:162)
    if (!(pre > 0.0f)) return false;
    if ((unsigned)r.t >= (unsigned)T) return false;
    p.b = r.b;
#pragma unroll
    for (int a = 0; a < AP; ++a) p.g[a] = r.g[a];
    p.tri = r.t;
    tri = r.t;
    return true;
  }
  __device__ __forceinline__ void load_triangle(int img, int tri, Triangle &t) const {
    const float4 *src = (const float4 *)(corners + ((size_t)img * T_ + tri) * (3 * AP));
#pragma unroll
    for (int q = 0; q < 3 * AP / 4; ++q) {
      const float4 f = src[q];
      const float v[4] = {f.x, f.y, f.z, f.w};
#pragma unroll
      for (int j = 0; j < 4; ++j) t.cr.c[(4 * q + j) / AP][(4 * q + j) % AP] = v[j];
    }
    load_bwd_triangle(recs + (size_t)img * T_ + tri, t.bt);
  }

  __device__ __forceinline__ void factors(const Pixel &p, const Triangle &t, float (&f)[kFactorStride],
                                          Image &) const {
    const float pre = (2.0f * p.b.x + 2.0f * p.b.y) + 2.0f * p.b.z;
    const float alpha = fminf(fmaxf(pre, 0.0f), 1.0f);  // rasterize.py:145-147
    float dalpha = 0.f, db[3] = {0.f, 0.f, 0.f};
    f[0] = p.b.x; f[1] = p.b.y; f[2] = p.b.z;
#pragma unroll
    for (int a = 0; a < AP; ++a) {
      const float go = p.g[a];
      const float gv = alpha * go;  // d/d(interpolated value), rasterize.py:149-150
      const float value = (t.cr.c[0][a] * p.b.x + t.cr.c[1][a] * p.b.y) + t.cr.c[2][a] * p.b.z;
      dalpha += go * (value - (a < A ? background[a] : 0.0f));
#pragma unroll
      for (int k = 0; k < 3; ++k) db[k] += gv * t.cr.c[k][a];
      f[3 + a] = gv;  // d/d attr[corner k][a] = b_k * gv: the product is formed in the reduction
    }
    // torch.clamp passes the gradient where min <= x <= max (inclusive)
    const float dpre = (pre >= 0.0f && pre <= 1.0f) ? 2.0f * dalpha : 0.0f;
    F3 dbary;
    dbary.x = db[0] + dpre; dbary.y = db[1] + dpre; dbary.z = db[2] + dpre;
    // rasterizer backward (cpp:162 skip rule, then cpp:202-269)
    const bool skip = p.tri == 0 && (p.b.x + p.b.y) + p.b.z < kDegenerateCutoff;
    float q[3];
    raster_pixel_q(p.b, dbary, t.bt, skip ? 0.f : t.bt.inv, q);
    f[3 + AP] = q[0]; f[4 + AP] = q[1]; f[5 + AP] = q[2];
#pragma unroll
    for (int k = kFactors; k < kFactorStride; ++k) f[k] = 0.f;
  }
};

// The same pass through k_accumulate_lanes (run_accum.h, round 3): the 3 AP + 9 sums stay in registers for as
// long as a lane stays on one triangle going down its column; only finished vertical runs go through LDS.
// Measured at 1024^2 x 32, 5k triangles: A = 4 392 -> 348 us, A = 8 ~-4 %.  (Less than the shading backward
// gained: SQ counters show this pass 53 % vector-busy at 3 waves per SIMD with ~40 branches and ~100 scalar
// instructions per row -- it is bound by the row loop's control flow and the triangle records' round trip,
// not by the reduction that the lane kernel removes.)
#ifndef MR_ATTR_LANES_PIPELINED
#define MR_ATTR_LANES_PIPELINED 0   // the two-rows-ahead row loop (run_accum.h): A = 4 1.11 -> 1.12 ms, A = 8 1.70 -> 1.78 (VGPRs 94 -> 102, 126 -> 140)
#endif
#ifndef MR_ATTR_LANES_MAX_AP
#define MR_ATTR_LANES_MAX_AP 8    // wider records stay on the rows kernel (AP = 12: 45 sums, 161 VGPRs, 17 KB of LDS
                                  // per wavefront: 601 us against the rows kernel's 612 at 1024^2 x 32 -- no gain)
#endif
template <int AP>
struct AttrLaneFn : AttrRowsFn<AP> {
  using Base = AttrRowsFn<AP>;
  static constexpr int kLaneRowsPerWave = 16;
  static constexpr bool kPipelinedRows = MR_ATTR_LANES_PIPELINED != 0;
  static constexpr int kMinWavesPerSimd = AP <= 4 ? 4 : 3;   // 94 / 126 / 161 VGPRs for AP = 4 / 8 / 12 (LDS: 11 / 13 / 17 KB per wavefront)
  __device__ static int column(int o) { return o; }
  __device__ __forceinline__ void accumulate(const typename Base::Pixel &p, const typename Base::Triangle &t,
                                             float (&a)[Base::kN], typename Base::Image &im) const {
    float f[Base::kFactorStride];
    Base::factors(p, t, f, im);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
#pragma unroll
      for (int c = 0; c < AP; ++c) a[k * AP + c] = fmaf(f[k], f[3 + c], a[k * AP + c]);
#pragma unroll
      for (int c = 0; c < 3; ++c) a[3 * AP + k * 3 + c] = fmaf(f[k], f[3 + AP + c], a[3 * AP + k * 3 + c]);
    }
  }
};

// ---------------------------------------------------------------------------------------------------------
// Round 4: the backward for a G-buffer the caller declares NORMALISED (MR_GBUFFER_NORMALISED: ids / bary are what
// mr_rasterize_forward / mr_rasterize_interpolate_forward wrote for these vertices -- rasterize() always).
// Every covered pixel's barycentrics sum to 1 within rounding, so alpha = clamp(2 sum b) is exactly 1 and lies
// outside the clamp's pass band: out = value, d value = dout, nothing flows through alpha (no `value`, no
// d alpha: 5 AP vector instructions per pixel gone).  The gradient through the barycentrics enters the
// rasterizer's backward only through differences (its brackets sum to ~0 over the corners, see ShadeFoldLaneFn
// in shade.hip), so the record holds the attributes as e0 = c0 - c2, e1 = c1 - c2 and two adjugate rows:
//   g0 = dout . e0, g1 = dout . e1;  q_c = (g0 (s_c b0 - u_0c) + g1 (s_c b1 - u_1c)) / |det|
//   sums: b_k dout[a] (3 A) and b_k q_c (9), kept in registers down the lane's vertical run (k_accumulate_lanes).
// Templated on the EXACT attribute count (A = 9 is 36 sums, not the 45 of its padded width).
constexpr int fold_rec_floats(int AP) { return 2 * AP + 12; }   // e0[AP] e1[AP] | u0[3] u1[3] s[3] 1/|det| - -

template <int AP>
__global__ __launch_bounds__(kThreads) void k_attr_fold_setup(const float *__restrict__ corners,
                                                              const BwdRec *__restrict__ recs, long nbt,
                                                              float *__restrict__ out) {
  const long gid = (long)blockIdx.x * kThreads + threadIdx.x;
  if (gid >= nbt) return;
  const float4 *src = (const float4 *)(corners + gid * (3 * AP));
  float c[3 * AP];
#pragma unroll
  for (int q = 0; q < 3 * AP / 4; ++q) {
    const float4 f = src[q];
    c[4 * q] = f.x; c[4 * q + 1] = f.y; c[4 * q + 2] = f.z; c[4 * q + 3] = f.w;
  }
  const BwdRec r = recs[gid];
  float v[fold_rec_floats(AP)];
#pragma unroll
  for (int a = 0; a < AP; ++a) {
    v[a] = c[a] - c[2 * AP + a];
    v[AP + a] = c[AP + a] - c[2 * AP + a];
  }
  float *t = v + 2 * AP;
  t[0] = r.a.x; t[1] = r.a.y; t[2] = r.a.z;          // adjugate row of corner 0 (clip components x, y, w)
  t[3] = r.a.w; t[4] = r.b.x; t[5] = r.b.y;          // ... of corner 1
  t[6] = r.c.y; t[7] = r.c.z; t[8] = r.c.w;          // column sums
  t[9] = r.d.x; t[10] = 0.f; t[11] = 0.f;            // 1 / |det|
  float4 *dst = (float4 *)(out + gid * fold_rec_floats(AP));
#pragma unroll
  for (int q = 0; q < fold_rec_floats(AP) / 4; ++q) dst[q] = make_float4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
}

template <int A>
struct AttrFoldLaneFn {
  static constexpr int AP = A <= 4 ? 4 : (A <= 8 ? 8 : (A <= 12 ? 12 : 16));
  static constexpr int kN = 3 * A + 9;
  static constexpr int kStride = AttrRowsFn<AP>::kStride;   // the gather reads rows of the padded layout
  static constexpr int kLaneRowsPerWave = 8;
  static constexpr int kMinWavesPerSimd = kN <= 21 ? 5 : (kN <= 36 ? 4 : 3);
  static constexpr bool kCountBackground = false;
  const float *__restrict__ dout;       // [B,H,W,A]
  const int32_t *__restrict__ ids;
  const F3 *__restrict__ bary;
  const float *__restrict__ fold;       // [B,T,fold_rec_floats(AP)]
  int T_, W, H;

  struct Pixel { F3 b; float g[A]; };
  struct Raw { F3 b; int t; float g[A]; };
  struct Triangle { float e0[A], e1[A], u0[3], u1[3], s[3], inv; };
  struct Image { int n_bg; };
  // sum o < 3 A: corner o / A, attribute o % A -> row column corner * AP + attribute; then the nine clip sums
  __device__ static int column(int o) { return o < 3 * A ? (o / A) * AP + o % A : 3 * AP + (o - 3 * A); }
  __device__ __forceinline__ void begin_image(int, Image &) const {}
  __device__ __forceinline__ void end_strip(int, int, Image &) const {}
  __device__ __forceinline__ void fetch(int img, int x, int y, size_t, Raw &r) const {
    // wave-uniform image bases + 32-bit offsets (W * H * A * 4 < 2^31 is checked by the launcher)
    const size_t img_px = (size_t)img * H * W;
    const unsigned pix = (unsigned)(y * W) + (unsigned)x;
    r.b = *(const F3 *)((const char *)(bary + img_px) + pix * 12u);
    r.t = *(const int32_t *)((const char *)(ids + img_px) + pix * 4u);
    const float *g = (const float *)((const char *)(dout + img_px * A) + pix * (unsigned)(A * 4));
#pragma unroll
    for (int a = 0; a < A; ++a) r.g[a] = g[a];
  }
  __device__ __forceinline__ bool prepare(const Raw &r, int T, int &tri, Pixel &p) const {
    const float pre = (2.0f * r.b.x + 2.0f * r.b.y) + 2.0f * r.b.z;
    if (!(pre > 0.0f)) return false;   // uncovered: alpha = 0, nothing reaches the attributes or the vertices
    if ((unsigned)r.t >= (unsigned)T) return false;
    p.b = r.b;
#pragma unroll
    for (int a = 0; a < A; ++a) p.g[a] = r.g[a];
    tri = r.t;
    return true;
  }
  __device__ __forceinline__ void load_triangle(int img, int tri, Triangle &t) const {
    const float4 *src = (const float4 *)(fold + ((size_t)img * T_ + tri) * fold_rec_floats(AP));
    float v[fold_rec_floats(AP)];
#pragma unroll
    for (int q = 0; q < fold_rec_floats(AP) / 4; ++q) {
      if (4 * q >= A && 4 * q < AP) continue;               // padding of e0
      if (4 * q >= AP + A && 4 * q < 2 * AP) continue;      // padding of e1
      const float4 f = src[q];
      v[4 * q] = f.x; v[4 * q + 1] = f.y; v[4 * q + 2] = f.z; v[4 * q + 3] = f.w;
    }
#pragma unroll
    for (int a = 0; a < A; ++a) { t.e0[a] = v[a]; t.e1[a] = v[AP + a]; }
#pragma unroll
    for (int c = 0; c < 3; ++c) { t.u0[c] = v[2 * AP + c]; t.u1[c] = v[2 * AP + 3 + c]; t.s[c] = v[2 * AP + 6 + c]; }
    t.inv = v[2 * AP + 9];
  }
  __device__ __forceinline__ void accumulate(const Pixel &p, const Triangle &t, float (&acc)[kN], Image &) const {
    float g0 = 0.f, g1 = 0.f;
#pragma unroll
    for (int a = 0; a < A; ++a) {
      g0 = fmaf(p.g[a], t.e0[a], g0);
      g1 = fmaf(p.g[a], t.e1[a], g1);
    }
    float q[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float w0 = t.s[c] * p.b.x - t.u0[c];
      const float w1 = t.s[c] * p.b.y - t.u1[c];
      q[c] = (g0 * w0 + g1 * w1) * t.inv;
    }
    const float b[3] = {p.b.x, p.b.y, p.b.z};
#pragma unroll
    for (int k = 0; k < 3; ++k) {
#pragma unroll
      for (int a = 0; a < A; ++a) acc[k * A + a] = fmaf(b[k], p.g[a], acc[k * A + a]);
#pragma unroll
      for (int c = 0; c < 3; ++c) acc[3 * A + k * 3 + c] = fmaf(b[k], q[c], acc[3 * A + k * 3 + c]);
    }
  }
};
#ifndef MR_ATTR_FOLD_MAX_A
#define MR_ATTR_FOLD_MAX_A 12   // wider: the rows kernel
#endif

// One thread per (image, vertex): sums the rows of the triangles incident to its vertex (CSR
// adjacency: entry = 3 * triangle + corner).  Every output is written exactly once, no atomics.
// DET (mr_set_deterministic): the rows hold 64-bit fixed-point sums (run_accum.h), converted here.
template <int AP, bool DET>
__global__ __launch_bounds__(kThreads) void k_attr_gather(
    const float *__restrict__ acc, const float *__restrict__ det_scale, const int32_t *__restrict__ offsets,
    const int32_t *__restrict__ entries, int B, int V, int T, int A, float *__restrict__ dattrs,
    float *__restrict__ dclip) {
  constexpr int STRIDE = AttrRowsFn<AP>::kStride;
  const long gid = (long)blockIdx.x * kThreads + threadIdx.x;
  if (gid >= (long)B * V) return;
  const int b = (int)(gid / V);
  const int v = (int)(gid - (long)b * V);
  float a[AP], c[3] = {0.f, 0.f, 0.f};
#pragma unroll
  for (int j = 0; j < AP; ++j) a[j] = 0.f;
  const int e1 = offsets[v + 1];
  for (int i = offsets[v]; i < e1; ++i) {
    const int e = entries[i];
    const int t = e / 3, k = e - 3 * t;
    const float *row = acc + ((size_t)b * T + t) * STRIDE;
    const long long *row_x = (const long long *)acc + ((size_t)b * T + t) * STRIDE;
#pragma unroll
    for (int j = 0; j < AP; ++j) a[j] += DET ? (float)row_x[k * AP + j] * det_scale[1] : row[k * AP + j];
#pragma unroll
    for (int j = 0; j < 3; ++j) c[j] += DET ? (float)row_x[3 * AP + k * 3 + j] * det_scale[1] : row[3 * AP + k * 3 + j];
  }
  if (DET && *det_overflow_flag(det_scale)) {  // a contribution outside the fixed-point range: NaN, not garbage
#pragma unroll
    for (int j = 0; j < AP; ++j) a[j] = __int_as_float(0x7fc00000);
    c[0] = c[1] = c[2] = __int_as_float(0x7fc00000);
  }
  float *dst = dattrs + (size_t)gid * A;
#pragma unroll
  for (int j = 0; j < AP; ++j)
    if (j < A) dst[j] = a[j];
  ((float4 *)dclip)[gid] = make_float4(c[0], c[1], 0.0f, c[2]);  // column z stays 0
}

inline int padded_attrs(int A) { return A <= 4 ? 4 : (A <= 8 ? 8 : (A <= 12 ? 12 : 16)); }
inline size_t acc_bytes(int B, int T, int AP) {  // 8 bytes per element: room for the deterministic mode's fixed point
  const int stride = (3 * AP + 9 + 3) & ~3;
  return align_up((size_t)B * T * stride * sizeof(long long), 256);
}
inline size_t corner_bytes(int B, int T, int AP) { return align_up((size_t)B * T * 3 * AP * sizeof(float), 256); }

template <int AP>
int setup_records(const float *attrs, const int32_t *tris, int B, int V, int T, int A, float *corners,
                  hipStream_t s) {
  const long nbt = (long)B * T;
  hipLaunchKernelGGL(k_attr_corner_setup<AP>, dim3((unsigned)((nbt + kThreads - 1) / kThreads)), dim3(kThreads),
                     0, s, attrs, tris, B, V, T, A, corners);
  return check_launch();
}

template <int AP>
int run_forward(const int32_t *ids, const float *bary, const float *attrs, const int32_t *tris, const float *bg,
                int B, int V, int T, int W, int H, int A, float *out, float *corners, hipStream_t s) {
  int rc = setup_records<AP>(attrs, tris, B, V, T, A, corners, s);
  if (rc != MR_OK) return rc;
  const unsigned px_per_image = (unsigned)W * (unsigned)H;  // W, H <= 65535
  const unsigned want = (px_per_image + kThreads - 1) / kThreads, cap = 2048u;
  hipLaunchKernelGGL(k_interp_forward_rec<AP>, dim3(want < cap ? want : cap, (unsigned)B), dim3(kThreads), 0, s,
                     ids, (const F3 *)bary, corners, bg, px_per_image, T, A, out);
  return check_launch();
}

template <int AP>
int run(const float *dout, const int32_t *ids, const float *bary, const float *clip, const float *attrs,
        const int32_t *tris, const float *bg, const int32_t *offsets, const int32_t *entries,
        const float *corner_records, int B, int V, int T, int W, int H, int A, float *dattrs, float *dclip,
        int gbuffer_flags, void *ws, hipStream_t s) {
  char *p = (char *)ws;
  float *acc = (float *)p;
  p += acc_bytes(B, T, AP);
  BwdRec *recs = (BwdRec *)p;
  p += align_up((size_t)B * T * sizeof(BwdRec), 256);
  float *corners = (float *)p;
  p += corner_bytes(B, T, AP);
  float *det_block = (float *)p;
  p += kDetBlockBytes;
  float *fold_recs = (float *)p;
  const bool det = g_deterministic != 0;
  if (zero_async(acc, (size_t)B * T * AttrRowsFn<AP>::kStride * (det ? sizeof(long long) : sizeof(float)), s) !=
      hipSuccess)
    return check_launch();
  int rc = MR_OK;
  if (det && (rc = launch_det_scale(dout, (size_t)B * H * W * A, 1.0f, det_block, s)) != MR_OK) return rc;
  rc = launch_bwd_setup(clip, tris, B, V, T, recs, s);
  if (rc != MR_OK) return rc;
  if (corner_records) {  // the forward's records (same inputs): skip the gather
    corners = const_cast<float *>(corner_records);
  } else {
    rc = setup_records<AP>(attrs, tris, B, V, T, A, corners, s);
    if (rc != MR_OK) return rc;
  }
  // a normalised G-buffer (rasterize()'s own): the folded lane kernel, templated on the exact attribute count
  const bool folded = (gbuffer_flags & MR_GBUFFER_NORMALISED) != 0 && !det && g_shade_backward_kernel != 1 &&
                      A <= MR_ATTR_FOLD_MAX_A && (size_t)W * H * A * sizeof(float) < (size_t)1 << 31;
  if (folded) {
    const long nbt = (long)B * T;
    hipLaunchKernelGGL(k_attr_fold_setup<AP>, dim3((unsigned)((nbt + kThreads - 1) / kThreads)), dim3(kThreads), 0, s, corners,
                       recs, nbt, fold_recs);
    if ((rc = check_launch()) != MR_OK) return rc;
    auto go = [&](auto tag) {
      constexpr int kA = decltype(tag)::value;
      if constexpr (kA >= 1 && kA <= MR_ATTR_FOLD_MAX_A && (kA + 3) / 4 * 4 == AP) {
        AttrFoldLaneFn<kA> fn{dout, ids, (const F3 *)bary, fold_recs, T, W, H};
        rc = launch_accumulate_lanes(fn, B, T, W, H, acc, s);
      }
    };
    switch (A - (AP - 4)) {   // A = AP - 3 .. AP
      case 1: go(std::integral_constant<int, AP - 3>{}); break;
      case 2: go(std::integral_constant<int, AP - 2>{}); break;
      case 3: go(std::integral_constant<int, AP - 1>{}); break;
      default: go(std::integral_constant<int, AP>{}); break;
    }
  } else
  if constexpr (AP <= MR_ATTR_LANES_MAX_AP) {
    if (!det && g_shade_backward_kernel != 1) {   // (debug switch: 1 = rows kernel)
      AttrLaneFn<AP> fn{{dout, ids, (const F3 *)bary, corners, recs, bg, A, T}};
      rc = launch_accumulate_lanes(fn, B, T, W, H, acc, s);
    } else {
      AttrRowsFn<AP> fn{dout, ids, (const F3 *)bary, corners, recs, bg, A, T};
      rc = launch_accumulate_rows(fn, B, T, W, H, acc, s, det ? det_block : nullptr);
    }
  } else {
    AttrRowsFn<AP> fn{dout, ids, (const F3 *)bary, corners, recs, bg, A, T};
    rc = launch_accumulate_rows(fn, B, T, W, H, acc, s, det ? det_block : nullptr);
  }
  if (rc != MR_OK) return rc;
  const long nbv = (long)B * V;
  const dim3 grid((unsigned)((nbv + kThreads - 1) / kThreads));
  if (det)
    hipLaunchKernelGGL((k_attr_gather<AP, true>), grid, dim3(kThreads), 0, s, acc, det_block, offsets, entries, B, V,
                       T, A, dattrs, dclip);
  else
    hipLaunchKernelGGL((k_attr_gather<AP, false>), grid, dim3(kThreads), 0, s, acc, det_block, offsets, entries, B, V,
                       T, A, dattrs, dclip);
  return check_launch();
}

}  // namespace

int interp_raster_max_attrs() { return 16; }

size_t interp_raster_backward_ws(int B, int V, int T, int W, int H, int A) {
  (void)V; (void)W; (void)H;
  const int AP = padded_attrs(A);
  return acc_bytes(B, T, AP) + align_up((size_t)B * T * sizeof(BwdRec), 256) + corner_bytes(B, T, AP) + kDetBlockBytes +
         align_up((size_t)B * T * fold_rec_floats(AP) * sizeof(float), 256);   // last: the folded kernel's records
}

size_t interp_records_bytes(int B, int T, int A) { return corner_bytes(B, T, padded_attrs(A)); }

// The per-(image, triangle) attribute records alone: [corner][AP] floats, AP = A rounded up to 4 / 8 / 12 / 16
// (what the rasterizer's INTERP epilogue reads, raster_forward.hip, and the fused backward reuses).
int launch_attr_records(const float *attrs, const int32_t *tris, int B, int V, int T, int A, void *records, hipStream_t s) {
  if ((size_t)B * T == 0) return MR_OK;
  float *corners = (float *)records;
  switch (padded_attrs(A)) {
    case 4: return setup_records<4>(attrs, tris, B, V, T, A, corners, s);
    case 8: return setup_records<8>(attrs, tris, B, V, T, A, corners, s);
    case 12: return setup_records<12>(attrs, tris, B, V, T, A, corners, s);
    default: return setup_records<16>(attrs, tris, B, V, T, A, corners, s);
  }
}

int launch_interp_forward_records(const int32_t *ids, const float *bary, const float *attrs, const int32_t *tris,
                                  const float *bg, int B, int V, int T, int W, int H, int A, float *out,
                                  void *records, hipStream_t s) {
  if ((size_t)B * W * H * A == 0) return MR_OK;
  float *corners = (float *)records;
  switch (padded_attrs(A)) {
    case 4: return run_forward<4>(ids, bary, attrs, tris, bg, B, V, T, W, H, A, out, corners, s);
    case 8: return run_forward<8>(ids, bary, attrs, tris, bg, B, V, T, W, H, A, out, corners, s);
    case 12: return run_forward<12>(ids, bary, attrs, tris, bg, B, V, T, W, H, A, out, corners, s);
    default: return run_forward<16>(ids, bary, attrs, tris, bg, B, V, T, W, H, A, out, corners, s);
  }
}

int launch_interp_raster_backward(const float *dout, const int32_t *ids, const float *bary, const float *clip,
                                  const float *attrs, const int32_t *tris, const float *bg,
                                  const int32_t *offsets, const int32_t *entries, const void *corner_records,
                                  int B, int V, int T, int W, int H, int A, float *dattrs, float *dclip,
                                  int gbuffer_flags, void *ws, hipStream_t s) {
  if (B == 0 || V == 0) return MR_OK;
  if (T == 0 || (size_t)W * H == 0 || A == 0) {  // nothing contributes: the outputs are zeros
    if ((size_t)A > 0 && zero_async(dattrs, (size_t)B * V * A * sizeof(float), s) != hipSuccess)
      return check_launch();
    if (zero_async(dclip, (size_t)B * V * 4 * sizeof(float), s) != hipSuccess) return check_launch();
    return MR_OK;
  }
  const float *cr = (const float *)corner_records;
  switch (padded_attrs(A)) {
    case 4: return run<4>(dout, ids, bary, clip, attrs, tris, bg, offsets, entries, cr, B, V, T, W, H, A, dattrs, dclip, gbuffer_flags, ws, s);
    case 8: return run<8>(dout, ids, bary, clip, attrs, tris, bg, offsets, entries, cr, B, V, T, W, H, A, dattrs, dclip, gbuffer_flags, ws, s);
    case 12: return run<12>(dout, ids, bary, clip, attrs, tris, bg, offsets, entries, cr, B, V, T, W, H, A, dattrs, dclip, gbuffer_flags, ws, s);
    default: return run<16>(dout, ids, bary, clip, attrs, tris, bg, offsets, entries, cr, B, V, T, W, H, A, dattrs, dclip, gbuffer_flags, ws, s);
  }
}

}  // namespace mr
