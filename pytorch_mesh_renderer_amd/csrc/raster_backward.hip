// Backward of the barycentric rasterizer for gfx950 (MI355X).
//
// Replaces rasterize_triangles_backward
// (reference: src/mesh_renderer/kernels/rasterize_triangles.cpp:131-273), a serial
// row-major loop that rebuilds a 3x3 adjugate per pixel and does nine '+=' into
// df_dvertices[V,4].
//
//   k_bwd_setup        one thread per (image, triangle): sign-corrected adjugate U,
//                      its column sums and 1/|det| as a 64-byte record (cpp:180-198).
//   k_accumulate_runs  (run_accum.h) streams the G-buffer once -- 28 B/px: dbary 12
//                      + id 4 + bary 12.  Each lane walks down a pixel column, fetches
//                      the triangle record only when the id changes, evaluates the
//                      nine partials of cpp:202-269 per pixel and keeps them in
//                      registers for the run; runs land in an LDS hash table and
//                      leave as contiguous float atomics into acc[B][T][9(+3)].
//   k_bwd_scatter      one thread per touched (image, triangle): nine atomics into
//                      dclip[B,V,4].
//
// Numerics.  d b_i / d M = (-U_ic b_j + S_c b_i b_j)/|det| is a small difference
// of large terms for small or sliver triangles, so the cancellation has to happen
// PER PIXEL, before any summation (summing g*b first and combining with U
// afterwards is ~sqrt(#pixels) less accurate -- measured 1.5e-4 vs the
// reference).  This file is therefore compiled with -ffp-contract=off and keeps
// the reference's association inside each per-pixel partial; only the order in
// which pixels are summed differs (fp32, order-dependent in the reference too),
// and the division by |det| is a multiplication by its fp32 reciprocal (<= 1.5 ulp
// per partial).  Parity bar: 1e-4 abs.
#include "run_accum.h"
#include "corner_rec.h"

namespace mr {

// Optional chores for the caller's accumulation pass, so that it needs no memset launches of its
// own (~5 us each): thread (image, triangle) clears the triangle's accumulator row (`zero_row_quads`
// float4 per row), and the grid clears `zero_tail_count` floats at `zero_tail` between its threads.
__global__ __launch_bounds__(256) void k_bwd_setup(
    const float4 *__restrict__ clip, const int32_t *__restrict__ tris, int B, int V, int T,
    BwdRec *__restrict__ recs, float4 *__restrict__ zero_rows, int zero_row_quads,
    float *__restrict__ zero_tail, int zero_tail_count, const CornerRec *__restrict__ corners,
    FoldRec *__restrict__ fold_recs, const float *__restrict__ pull_transforms) {
  const long gid = (long)blockIdx.x * 256 + threadIdx.x;
  for (long i = gid; i < zero_tail_count; i += (long)gridDim.x * 256) zero_tail[i] = 0.0f;
  {  // the workgroup's 256 rows are one contiguous range: cleared with coalesced 16-byte stores
    const long first = (long)blockIdx.x * 256 * zero_row_quads, end = (long)B * T * zero_row_quads;
    for (int q = 0; q < zero_row_quads; ++q) {
      const long i = first + (long)q * 256 + threadIdx.x;
      if (i < end) zero_rows[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  if (gid >= (long)B * T) return;
  const int b = (int)(gid / T);
  const int t = (int)(gid - (long)b * T);
  const int i0 = tris[3 * t], i1 = tris[3 * t + 1], i2 = tris[3 * t + 2];
  BwdRec r;
  r.a = r.b = r.c = r.d = make_float4(0.f, 0.f, 0.f, 0.f);
  if ((unsigned)i0 < (unsigned)V && (unsigned)i1 < (unsigned)V && (unsigned)i2 < (unsigned)V) {
    const float4 p0 = clip[(long)b * V + i0], p1 = clip[(long)b * V + i1], p2 = clip[(long)b * V + i2];
    const float a11 = p0.x, a12 = p1.x, a13 = p2.x;
    const float a21 = p0.y, a22 = p1.y, a23 = p2.y;
    const float a31 = p0.w, a32 = p1.w, a33 = p2.w;
    float u0 = a22 * a33 - a32 * a23, u1 = a13 * a32 - a33 * a12, u2 = a12 * a23 - a22 * a13;
    float u3 = a23 * a31 - a33 * a21, u4 = a11 * a33 - a31 * a13, u5 = a13 * a21 - a23 * a11;
    float u6 = a21 * a32 - a31 * a22, u7 = a12 * a31 - a32 * a11, u8 = a11 * a22 - a21 * a12;
    const float det = a11 * u0 + a12 * u3 + a13 * u6;
    if (det < 0.0f) {
      u0 = -u0; u1 = -u1; u2 = -u2; u3 = -u3; u4 = -u4; u5 = -u5; u6 = -u6; u7 = -u7; u8 = -u8;
    }
    r.a = make_float4(u0, u1, u2, u3);
    r.b = make_float4(u4, u5, u6, u7);
    r.c = make_float4(u8, (u0 + u3) + u6, (u1 + u4) + u7, (u2 + u5) + u8);  // cpp:187-198
    r.d = make_float4(1.0f / fabsf(det), 0.f, 0.f, 0.f);
  }
  recs[gid] = r;
  if (fold_recs) {  // (launch-uniform) the folded lane kernel's record, see corner_rec.h
    float c[32];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const float4 f = corners[gid].q[q];
      c[4 * q] = f.x; c[4 * q + 1] = f.y; c[4 * q + 2] = f.z; c[4 * q + 3] = f.w;
    }
    const float u[9] = {r.a.x, r.a.y, r.a.z, r.a.w, r.b.x, r.b.y, r.b.z, r.b.w, r.c.x};
    if (pull_transforms) {   // (launch-uniform) the pulled form, for ShadeFoldLaneFn
      float pull[12];
      load_pull_rows(pull_transforms, b, pull);
      store_fold_record(c, u, r.d.x, fold_recs + gid, pull);
    } else {
      store_fold_record(c, u, r.d.x, fold_recs + gid);
    }
  }
}

int launch_bwd_setup(const float *clip, const int32_t *tris, int B, int V, int T, BwdRec *recs,
                     hipStream_t s, void *zero_rows, size_t zero_row_bytes, float *zero_tail,
                     int zero_tail_count, const void *corners, void *fold_recs, const float *pull_transforms) {
  const long nbt = (long)B * T;
  if (nbt == 0) return MR_OK;
  if (zero_row_bytes % 16 != 0 || ((corners == nullptr) != (fold_recs == nullptr))) return MR_EINVAL;
  hipLaunchKernelGGL(k_bwd_setup, dim3((unsigned)((nbt + 255) / 256)), dim3(256), 0, s,
                     (const float4 *)clip, tris, B, V, T, recs, (float4 *)zero_rows,
                     zero_rows ? (int)(zero_row_bytes / 16) : 0, zero_tail, zero_tail ? zero_tail_count : 0,
                     (const CornerRec *)corners, (FoldRec *)fold_recs, pull_transforms);
  return check_launch();
}

namespace {

constexpr int kThreads = 256;
constexpr float kDegenerateCutoff = 0.9f;  // cpp:13
constexpr int kStride = 12;                // floats per acc row (9 used): 48 B

struct RasterGradFn {
  static constexpr int kN = 9;  // [corner j][component c] partials
  static constexpr int kStride = mr::kStride;
  static constexpr int kSlots = 512;
  static constexpr int kMinWavesPerSimd = 6;
  const F3 *__restrict__ dbary;
  const int32_t *__restrict__ ids;
  const F3 *__restrict__ bary;
  const BwdRec *__restrict__ recs;
  int T_;

  struct Pixel {
    F3 b, g;
  };
  struct Raw {
    F3 b, g;
    int t;
  };
  using Triangle = BwdTriangle;
  using Image = NoImageSums;

  __device__ __forceinline__ void begin_image(int, Image &) const {}
  __device__ __forceinline__ void end_image(int, Image &) const {}
  __device__ __forceinline__ void fetch(int, int, int, size_t pix, Raw &r) const {
    r.t = __builtin_nontemporal_load(&ids[pix]);
    r.b = load_streamed(&bary[pix]);
    r.g = load_streamed(&dbary[pix]);
  }
  __device__ __forceinline__ bool prepare(const Raw &r, int T, int &tri, Pixel &p) const {
    if ((unsigned)r.t >= (unsigned)T) return false;                             // foreign id
    if (r.t == 0 && (r.b.x + r.b.y) + r.b.z < kDegenerateCutoff) return false;  // cpp:162
    p.b = r.b;
    p.g = r.g;
    tri = r.t;
    return true;
  }
  __device__ __forceinline__ void load_triangle(int img, int tri, Triangle &t) const {
    load_bwd_triangle(recs + (size_t)img * T_ + tri, t);
  }
  __device__ __forceinline__ void accumulate(const Pixel &p, const Triangle &t, float (&acc)[kN],
                                             Image &) const {
    raster_pixel_partials(p.b, p.g, t, acc);
  }
};

// The pixel pass is the lane-accumulating kernel (round 3); the column-run kernel with its LDS hash table stays for the
// deterministic mode (launch_accumulate_runs_fixed) and interpolate.hip, the rows-kernel variant measured in round 3 is gone.

// Round 3: the same nine sums through k_accumulate_lanes (run_accum.h) -- each lane keeps them in
// registers down its vertical run (RasterGradFn's accumulate()), only finished runs go through LDS.
// Measured, whole mr_rasterize_backward call: 1024^2 x 32 / 5k triangles 0.310 -> 0.288 ms, 2048^2 x 8 /
// 50k 0.364 -> 0.366, 256^2 x 8 0.054 -> (with the launch-size-dependent strip height) see DESIGN.
// MR_RASTER_BWD_PIPELINED (second half of round 3): the row loop with the streamed planes two rows ahead and
// unconditional record loads (run_accum.h, kPipelinedRows): 0.282 -> 0.267 ms, 0.366 -> 0.343.
#ifndef MR_RASTER_BWD_PIPELINED
#define MR_RASTER_BWD_PIPELINED 1
#endif
#ifndef MR_RASTER_LANE_ROWS
#define MR_RASTER_LANE_ROWS 16
#endif
struct RasterLanesFn : RasterGradFn {
  static constexpr int kLaneRowsPerWave = MR_RASTER_LANE_ROWS;
  static constexpr bool kCountBackground = false;
  static constexpr bool kPipelinedRows = MR_RASTER_BWD_PIPELINED != 0;
  __device__ static int column(int o) { return o; }
  struct Image { int n_bg; };
  __device__ __forceinline__ void begin_image(int, Image &) const {}
  __device__ __forceinline__ void end_strip(int, int, Image &) const {}
  __device__ __forceinline__ void accumulate(const Pixel &p, const Triangle &t, float (&acc)[kN], Image &) const {
    raster_pixel_partials(p.b, p.g, t, acc);
  }
};

__global__ __launch_bounds__(kThreads) void k_bwd_scatter(
    const float *__restrict__ acc, const int32_t *__restrict__ tris, int B, int V, int T,
    float *__restrict__ dclip) {
  const long gid = (long)blockIdx.x * kThreads + threadIdx.x;
  if (gid >= (long)B * T) return;
  float a[9];
  bool any = false;
#pragma unroll
  for (int k = 0; k < 9; ++k) {
    a[k] = acc[gid * kStride + k];
    any |= (a[k] != 0.0f);  // NaN counts as touched
  }
  if (!any) return;
  const int b = (int)(gid / T);
  const int t = (int)(gid - (long)b * T);
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const int vi = tris[3 * t + j];
    if ((unsigned)vi >= (unsigned)V) continue;
    float *dst = dclip + ((long)b * V + vi) * 4;
    atomicAdd(&dst[0], a[j * 3 + 0]);  // x
    atomicAdd(&dst[1], a[j * 3 + 1]);  // y
    atomicAdd(&dst[3], a[j * 3 + 2]);  // w; the z column never receives gradient
  }
}

// 8 bytes per element: room for the deterministic mode's fixed-point accumulators
inline size_t acc_bytes(int B, int T) { return align_up((size_t)B * T * kStride * sizeof(long long), 256); }
inline size_t dclip_fixed_bytes(int B, int V) { return align_up((size_t)B * V * 4 * sizeof(long long), 256); }

// ---- deterministic mode (mr_set_deterministic): fixed point end to end -----------------------
__global__ __launch_bounds__(kThreads) void k_abs_max_f(const float *__restrict__ x, size_t n, int *__restrict__ max_bits) {
  int best = 0;
  for (size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += (size_t)gridDim.x * kThreads)
    best = max(best, __float_as_int(fabsf(x[i])));
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) best = max(best, __shfl_down(best, off));
  // one atomic per WORKGROUP (thousands of wavefronts on one address queue up behind each other)
  __shared__ int s_best[kThreads / kWave];
  if ((threadIdx.x & (kWave - 1)) == 0) s_best[threadIdx.x >> 6] = best;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < kThreads / kWave; ++w) best = max(best, s_best[w]);
    if (best != 0) atomicMax(max_bits, best);
  }
}

__global__ void k_det_scale_from_max(const int *__restrict__ max_bits, float *__restrict__ det_scale) {
  const float g = __int_as_float(max_bits[0]);
  int e = 0;
  if (g > 0.0f && g < INFINITY) (void)frexpf(g, &e);
  const int k = min(max(41 - e, -100), 100);
  det_scale[0] = ldexpf(1.0f, k);
  det_scale[1] = ldexpf(1.0f, -k);
}

// per-triangle fixed-point sums -> per-vertex fixed-point sums (integer atomics: order-free)
__global__ __launch_bounds__(kThreads) void k_bwd_scatter_fixed(
    const long long *__restrict__ acc, const int32_t *__restrict__ tris, int B, int V, int T,
    long long *__restrict__ dclip_fixed) {
  const long gid = (long)blockIdx.x * kThreads + threadIdx.x;
  if (gid >= (long)B * T) return;
  long long a[9];
  bool any = false;
#pragma unroll
  for (int k = 0; k < 9; ++k) {
    a[k] = acc[gid * kStride + k];
    any |= (a[k] != 0);
  }
  if (!any) return;
  const int b = (int)(gid / T);
  const int t = (int)(gid - (long)b * T);
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const int vi = tris[3 * t + j];
    if ((unsigned)vi >= (unsigned)V) continue;
    unsigned long long *dst = (unsigned long long *)dclip_fixed + ((long)b * V + vi) * 4;
    atomicAdd(&dst[0], (unsigned long long)a[j * 3 + 0]);
    atomicAdd(&dst[1], (unsigned long long)a[j * 3 + 1]);
    atomicAdd(&dst[3], (unsigned long long)a[j * 3 + 2]);
  }
}

__global__ __launch_bounds__(kThreads) void k_from_fixed(const long long *__restrict__ fixed,
                                                         const float *__restrict__ det_scale, long n,
                                                         float *__restrict__ out) {
  const long i = (long)blockIdx.x * kThreads + threadIdx.x;
  // a contribution did not fit the fixed-point range (run_accum.h, atomic_add_fixed): NaN, not garbage
  if (i < n) out[i] = *det_overflow_flag(det_scale) ? __int_as_float(0x7fc00000) : (float)fixed[i] * det_scale[1];
}

}  // namespace

extern thread_local int g_deterministic;

size_t raster_backward_ws(int B, int V, int T, int W, int H) {
  (void)W; (void)H;
  return acc_bytes(B, T) + align_up((size_t)B * T * sizeof(BwdRec), 256) + dclip_fixed_bytes(B, V) + 256;
}

int launch_raster_backward(const float *dbary, const float *clip, const int32_t *tris,
                           const int32_t *ids, const float *bary, int B, int V, int T, int W,
                           int H, float *dclip, void *ws, hipStream_t s) {
  if (B == 0 || V == 0) return MR_OK;
  if (zero_async(dclip, (size_t)B * V * 4 * sizeof(float), s) != hipSuccess) return check_launch();
  if (T == 0) return MR_OK;
  float *acc = (float *)ws;
  BwdRec *recs = (BwdRec *)((char *)ws + acc_bytes(B, T));
  const bool det = g_deterministic != 0;
  if (zero_async(acc, (size_t)B * T * kStride * (det ? sizeof(long long) : sizeof(float)), s) != hipSuccess)
    return check_launch();
  int rc = launch_bwd_setup(clip, tris, B, V, T, recs, s);
  if (rc != MR_OK) return rc;
  RasterGradFn fn{(const F3 *)dbary, ids, (const F3 *)bary, recs, T};
  if (det) {
    long long *dclip_fixed = (long long *)((char *)recs + align_up((size_t)B * T * sizeof(BwdRec), 256));
    float *det_scale = (float *)((char *)dclip_fixed + dclip_fixed_bytes(B, V));
    int *max_bits = (int *)(det_scale + 4);
    if (zero_async(dclip_fixed, dclip_fixed_bytes(B, V) + 256, s) != hipSuccess) return check_launch();
    const size_t n = (size_t)B * H * W * 3;
    const size_t want = (n + kThreads - 1) / kThreads;
    hipLaunchKernelGGL(k_abs_max_f, dim3((unsigned)(want < 2048 ? want : 2048)), dim3(kThreads), 0, s, dbary, n, max_bits);
    if ((rc = check_launch()) != MR_OK) return rc;
    hipLaunchKernelGGL(k_det_scale_from_max, dim3(1), dim3(1), 0, s, max_bits, det_scale);
    if ((rc = check_launch()) != MR_OK) return rc;
    {
      KernelTimer timer(MR_TIMER_RASTER_BACKWARD, s);
      rc = launch_accumulate_runs_fixed(fn, B, T, W, H, acc, det_scale, s);
    }
    if (rc != MR_OK) return rc;
    const long nbt = (long)B * T, nv4 = (long)B * V * 4;
    hipLaunchKernelGGL(k_bwd_scatter_fixed, dim3((unsigned)((nbt + kThreads - 1) / kThreads)), dim3(kThreads), 0, s,
                       (const long long *)acc, tris, B, V, T, dclip_fixed);
    if ((rc = check_launch()) != MR_OK) return rc;
    hipLaunchKernelGGL(k_from_fixed, dim3((unsigned)((nv4 + kThreads - 1) / kThreads)), dim3(kThreads), 0, s,
                       dclip_fixed, det_scale, nv4, dclip);
    return check_launch();
  }
  {
    KernelTimer timer(MR_TIMER_RASTER_BACKWARD, s);
    RasterLanesFn lanes{{(const F3 *)dbary, ids, (const F3 *)bary, recs, T}};
    rc = launch_accumulate_lanes(lanes, B, T, W, H, acc, s);
  }
  if (rc != MR_OK) return rc;
  const long nbt = (long)B * T;
  hipLaunchKernelGGL(k_bwd_scatter, dim3((unsigned)((nbt + kThreads - 1) / kThreads)), dim3(kThreads),
                     0, s, acc, tris, B, V, T, dclip);
  return check_launch();
}

}  // namespace mr
