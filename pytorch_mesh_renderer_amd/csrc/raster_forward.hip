// Forward G-buffer rasterizer for gfx950 (MI355X).
//
// Replaces rasterize_triangles_forward
// (reference: src/mesh_renderer/kernels/rasterize_triangles.cpp:302-419 and its
// helpers :19-98) with three kernels on one stream -- and, as mr_render_forward, the whole
// forward of render() (src/mesh_renderer/render.py:183-228): a clip-space transform in front
// (k_vertex_transform) and the Phong shading as the epilogue of k_raster's tile walk
// (template parameter SHADE, see RasterShade):
//
//   k_setup   one thread per (image, triangle): sign-corrected adjugate, clip z/w,
//             pixel bbox (binary64 projection as in cpp:361-366), packed into a
//             64-byte record; plus the binary64 pixel-centre tables (cpp:376-377);
//             for mr_render_forward also the shading's corner record (corner_rec.h).
//   k_coarse  one workgroup per (image, 256x256-pixel cell): id-ordered list of the
//             triangles whose bbox touches the cell, the cell's depth split, and -- second
//             level -- the same list cut down once more for each of the cell's 4 x 4 regions.
//   k_raster  one 256-thread workgroup per 64x64-pixel region (32x32 for small launches), two stages:
//     bin    each wavefront scans a quarter of the region's candidate list (no barrier
//            in the loop): bbox-vs-region test, then an EXACT trivial reject -- the
//            reference's own edge function evaluated at the region's most favourable
//            pixel centre; fp32 multiply/add are monotone, so a negative value there
//            proves every pixel of the region fails that edge.  Survivors are copied,
//            in triangle-id order, as 80-byte entries (adjugate, z, w, id, bbox) into
//            an LDS bin; 80 B = 20 banks keeps per-lane ds_read_b128 conflict-free.
//     tiles  each wavefront walks 16x4-pixel tiles, one pixel per lane.  Per tile:
//            (0) once per region, one thread per bin entry walks the tiles under the
//                entry's bbox, applies the same exact trivial reject per tile and sets
//                its bit in that tile's 256-bit LDS mask;
//            (1) coverage: for every surviving entry, read wave-uniformly from LDS,
//                all lanes evaluate the three edge functions and record "inside" as
//                one bit of a per-lane mask -- no divisions, no divergence;
//            (2) depth: each lane walks ITS OWN set bits in ascending id, re-reads that
//                entry from LDS and runs the reference's barycentric / z arithmetic.
//                The loop runs max-over-lanes(depth complexity) times, not once per
//                candidate, and every active lane does useful work.
//            Ascending id per pixel reproduces the reference's sequential z-buffer
//            (ties -> later id, NaN handling) with no ordering trick.  Each pixel is
//            written exactly once: (id, z, b0, b1, b2), 20 B/px, whole 128-B lines
//            per workgroup.  A bin that fills up is flushed and the walk resumes from
//            the stored state (rare: > 64 survivors in one quarter of the list).
//
// Exactness: this file is compiled with -ffp-contract=off; every float expression
// below is written in the reference's association order.
// fp32 '/' lowers to the IEEE-correct v_div_scale/v_div_fmas/v_div_fixup sequence.
#include <type_traits>

#include "mr_internal.h"
#include "shade_pixel.h"
#include "spec_pixel.h"

namespace mr {

thread_local int g_last_hip_error = 0;

namespace {

constexpr int kThreads = 256;

// x86 cvttss2si semantics for the reference's static_cast<int> (cpp:21,29):
// out-of-range and NaN convert to INT_MIN, which then clamps to 0.
__device__ __forceinline__ int cvt_trunc_x86(float f) {
  return (f >= -2147483648.0f && f < 2147483648.0f) ? (int)f : INT_MIN;
}
__device__ __forceinline__ int clampi(int v, int lo, int hi) {
  const int a = v > lo ? v : lo;
  return a < hi ? a : hi;
}
// std::max / std::min argument order matters for NaN (cpp:19-31).
__device__ __forceinline__ float max_std(float a, float b) { return (a < b) ? b : a; }
__device__ __forceinline__ float min_std(float a, float b) { return (b < a) ? b : a; }

// A proven lower bound of every depth value the rasterizer can compute for a "tame" triangle, or
// -inf for any other.  Tame: all three w in [2^-60, 2^60], every |z| zero or in [2^-60, 2^60].
// For an inside pixel the kernel evaluates (cpp:395-397), with rounded barycentrics b_i >= 0,
//     zz = fl( fl(fl(b0 z0 + b1 z1) + b2 z2) / fl(fl(b0 w0 + b1 w1) + b2 w2) ),
// every product and sum rounded once (no contraction).  With r_i = z_i / w_i, u = 2^-24 and
// rho = max|z_i| / min w_i:  sum b_i z_i = sum (b_i w_i) r_i >= min(r) * sum b_i w_i, the rounded
// numerator is off by at most 3u * sum b_i |z_i| <= 3u * rho * sum b_i w_i, the rounded denominator
// by a factor within 1 +- 3u, the division by another 1 +- u, and |min r| <= rho, hence
//     zz >= min(r) - 8.1 u rho            (plus < 2^-86 for products that underflow).
// The bound below leaves twice that margin and is rounded towards -inf.  It also follows that a
// tame triangle never produces a NaN or an infinite depth (its denominator is positive).
// Used ONLY to order and skip work (front-to-back classes of the bin, see k_raster): results
// stay bit-identical to the sequential z-buffer of the reference.
__device__ __forceinline__ float depth_lower_bound(float z0, float z1, float z2, float w0, float w1,
                                                   float w2) {
  const float wmin = fminf(fminf(w0, w1), w2), wmax = fmaxf(fmaxf(w0, w1), w2);
  const float zabs = fmaxf(fmaxf(fabsf(z0), fabsf(z1)), fabsf(z2));
  auto z_ok = [](float z) { const float a = fabsf(z); return a == 0.0f || (a >= 0x1p-60f && a <= 0x1p60f); };
  const bool tame = wmin >= 0x1p-60f && wmax <= 0x1p60f && z_ok(z0) && z_ok(z1) && z_ok(z2);
  if (!tame) return -INFINITY;  // also taken for NaNs: every comparison above is false for them
  const double rmin = fmin(fmin((double)z0 / (double)w0, (double)z1 / (double)w1), (double)z2 / (double)w2);
  const double bound = rmin - 0x1p-20 * ((double)zabs / (double)wmin) - 0x1p-80;
  float f = (float)bound;
  if ((double)f > bound) f = nextafterf(f, -INFINITY);
  return f;
}

// clip = transform (row-major 4x4 per image) applied to (vertex, 1): what render() feeds the
// rasterizer (camera_utils.transform_homogeneous, src/common/camera_utils.py:142-170), one thread per
// (image, vertex) instead of a concatenation, a batched GEMM and their autograd nodes.
__global__ __launch_bounds__(kThreads) void k_vertex_transform(
    const F3 *__restrict__ vertices, const float4 *__restrict__ transforms, int B, int V,
    float4 *__restrict__ clip) {
  const long gid = (long)blockIdx.x * kThreads + threadIdx.x;
  if (gid >= (long)B * V) return;
  const int b = (int)(gid / V);
  const F3 p = vertices[gid];
  float o[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const float4 m = transforms[(size_t)b * 4 + r];
    o[r] = ((m.x * p.x + m.y * p.y) + m.z * p.z) + m.w;
  }
  clip[gid] = make_float4(o[0], o[1], o[2], o[3]);
}

// The attribute arrays of render()'s shading, for launches that also want the per-triangle corner
// records (corner_rec.h) built on the side: `corners` == nullptr switches that off.
struct SetupAttributes {
  const F3 *__restrict__ normals, *__restrict__ positions, *__restrict__ diffuse;
  CornerRec *__restrict__ corners;
  // Round 4 (with `corners`): the caller will run the folded shading backward on these inputs
  // (mr_render_forward's `backward_prepared`): the thread also writes its triangle's FoldRec -- it has
  // the corner attributes and the sign-corrected adjugate in registers anyway -- and clears the triangle's
  // accumulator row, so that the backward needs no setup launch of its own (k_bwd_setup: 19 us at 1024^2 x 32).
  FoldRec *__restrict__ fold_recs;
  float4 *__restrict__ fold_acc;   // [B*T][kFoldAccStride / 4]
  const float *__restrict__ fold_transforms;   // [B,4,4] (with fold_recs): the records carry the pulled form (corner_rec.h)
  // Round 5 (with `positions`): the clip-space transform of render() happens HERE -- vertex thread gid writes clip_out[gid],
  // and a triangle's thread transforms its own three corners (k_vertex_transform's expression, the same bits) instead of
  // waiting for a launch of its own to have written them: one launch and one dependent round trip less per step.
  const float4 *__restrict__ xf = nullptr;     // [B][4] rows of the clip-space transforms, or nullptr: `clip` is an input
  float4 *__restrict__ clip_out = nullptr;     // [B,V,4]
};

// Heaviest regions first (round 3).  k_raster's workgroups cost anything between ~5 us (background)
// and ~30 us (sphere interior); dispatched in image order, the launch ended with a tail of heavy
// regions at falling occupancy.  k_coarse files every region of an XCD's range under a weight class
// -- by the length of its candidate list -- and k_raster's workgroups take the classes heaviest first,
// so the launch drains on the cheap ones.  Placement only: results do not depend on it.
constexpr int kWeightClasses = 8;
__device__ __forceinline__ int region_weight_class(int count) {  // 0 = heaviest; count < 0: the list overflowed
  if (count < 0 || count >= 256) return 0;
  if (count == 0) return kWeightClasses - 1;
  return min(kWeightClasses - 2, 8 - (32 - __builtin_clz((unsigned)count)));  // 128.. -> 1, 64.. -> 2, ..., 1..7 -> 6
}

__device__ __forceinline__ float4 clip_of(const F3 p, const float4 *__restrict__ rows) {   // k_vertex_transform's expression
  float o[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const float4 m = rows[r];
    o[r] = ((m.x * p.x + m.y * p.y) + m.z * p.z) + m.w;
  }
  return make_float4(o[0], o[1], o[2], o[3]);
}

__global__ __launch_bounds__(kThreads) void k_setup(
    const float4 *__restrict__ clip, const int32_t *__restrict__ tris, int B, int V, int T,
    int W, int H, TriRec *__restrict__ recs, TriBox *__restrict__ bbs,
    float *__restrict__ pxtab, float *__restrict__ pytab, const SetupAttributes attrs,
    int32_t *__restrict__ order_count, long n_main) {
  const long gid = (long)blockIdx.x * kThreads + threadIdx.x;
  if (gid < kXcds * kWeightClasses) order_count[gid] = 0;  // k_coarse counts into it (a later kernel on the stream)
  const long nbt = (long)B * T;
  const float hw = (float)(0.5 * (double)W);  // cpp:309
  const float hh = (float)(0.5 * (double)H);  // cpp:310
  if (gid >= n_main) {   // (n_main = max(B T, B V with attrs.xf): the threads behind fill the tables)
    // pixel-centre tables: binary64 expression, one rounding (cpp:376-377)
    const long k = gid - n_main;
    if (k < W) {
      pxtab[k] = (float)(((double)k + 0.5) / (double)hw - 1.0);
    } else if (k < (long)W + H) {
      const long r = k - W;
      pytab[r] = (float)(((double)r + 0.5) / (double)hh - 1.0);
    }
    return;
  }
  if (attrs.xf && gid < (long)B * V)   // launch-uniform pointer: this thread's vertex
    attrs.clip_out[gid] = clip_of(attrs.positions[gid], attrs.xf + (size_t)(gid / V) * 4);
  if (gid >= nbt) return;
  const int b = (int)(gid / T);
  const int t = (int)(gid - (long)b * T);
  float corner_values[32];
  if (attrs.corners) {
    gather_corner_values(attrs.normals, attrs.positions, attrs.diffuse, tris, b, t, V, corner_values);
#pragma unroll
    for (int q = 0; q < 8; ++q)
      attrs.corners[gid].q[q] = make_float4(corner_values[4 * q], corner_values[4 * q + 1], corner_values[4 * q + 2],
                                            corner_values[4 * q + 3]);
    if (attrs.fold_acc) {
#pragma unroll
      for (int q = 0; q < kFoldAccStride / 4; ++q) attrs.fold_acc[gid * (kFoldAccStride / 4) + q] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  const int i0 = tris[3 * t + 0], i1 = tris[3 * t + 1], i2 = tris[3 * t + 2];
  TriBox bb{0u, 0u, -INFINITY, 0u};
  if ((unsigned)i0 < (unsigned)V && (unsigned)i1 < (unsigned)V && (unsigned)i2 < (unsigned)V) {
    float4 p0, p1, p2;
    if (attrs.xf) {   // (the same bits clip_out gets: one expression, no contraction in this file)
      const float4 *rows = attrs.xf + (size_t)b * 4;
      p0 = clip_of(attrs.positions[(long)b * V + i0], rows);
      p1 = clip_of(attrs.positions[(long)b * V + i1], rows);
      p2 = clip_of(attrs.positions[(long)b * V + i2], rows);
    } else {
      p0 = clip[(long)b * V + i0];
      p1 = clip[(long)b * V + i1];
      p2 = clip[(long)b * V + i2];
    }
    const float w0 = p0.w, w1 = p1.w, w2 = p2.w;
    if (!(w0 < 0 && w1 < 0 && w2 < 0)) {  // cpp:339
      // rows of M: x, y, w; columns: the three corners (cpp:350-353)
      const float a11 = p0.x, a12 = p1.x, a13 = p2.x;
      const float a21 = p0.y, a22 = p1.y, a23 = p2.y;
      const float a31 = w0, a32 = w1, a33 = w2;
      float m0 = a22 * a33 - a32 * a23;
      float m1 = a13 * a32 - a33 * a12;
      float m2 = a12 * a23 - a22 * a13;
      float m3 = a23 * a31 - a33 * a21;
      float m4 = a11 * a33 - a31 * a13;
      float m5 = a13 * a21 - a23 * a11;
      float m6 = a21 * a32 - a31 * a22;
      float m7 = a12 * a31 - a32 * a11;
      float m8 = a11 * a22 - a21 * a12;
      const float det = a11 * m0 + a12 * m3 + a13 * m6;  // cpp:77
      if (det < 0.0f) {
        m0 = -m0; m1 = -m1; m2 = -m2; m3 = -m3; m4 = -m4;
        m5 = -m5; m6 = -m6; m7 = -m7; m8 = -m8;
      }
      int l = 0, r = W, bot = 0, top = H;  // cpp:356
      if (w0 > 0 && w1 > 0 && w2 > 0) {    // cpp:360
        // float divide, then binary64 add and multiply, rounded once (cpp:361-366)
        const float x0 = (float)(((double)(p0.x / w0) + 1.0) * (double)hw);
        const float x1 = (float)(((double)(p1.x / w1) + 1.0) * (double)hw);
        const float x2 = (float)(((double)(p2.x / w2) + 1.0) * (double)hw);
        const float y0 = (float)(((double)(p0.y / w0) + 1.0) * (double)hh);
        const float y1 = (float)(((double)(p1.y / w1) + 1.0) * (double)hh);
        const float y2 = (float)(((double)(p2.y / w2) + 1.0) * (double)hh);
        l = clampi(cvt_trunc_x86(floorf(min_std(min_std(x0, x1), x2))), 0, W);
        r = clampi(cvt_trunc_x86(ceilf(max_std(max_std(x0, x1), x2))), 0, W);
        bot = clampi(cvt_trunc_x86(floorf(min_std(min_std(y0, y1), y2))), 0, H);
        top = clampi(cvt_trunc_x86(ceilf(max_std(max_std(y0, y1), y2))), 0, H);
      }
      if (r > l && top > bot) {
        const uint2 box = pack_bbox(l, r, bot, top);
        bb.lr = box.x;
        bb.bt = box.y;
        bb.zlo = depth_lower_bound(p0.z, p1.z, p2.z, w0, w1, w2);
        TriRec rec;
        // edge i = (m[3i], m[3i+1], m[3i+2]); edges 0 and 1 interleaved for packed fp32 math
        rec.a = make_float4(m0, m3, m1, m4);
        rec.b = make_float4(m2, m5, m6, m7);
        // (z_k, w_k) pairs: the depth loop forms clip z and clip w with packed fp32 math
        rec.c = make_float4(m8, 0.0f, p0.z, w0);
        rec.d = make_float4(p1.z, w1, p2.z, w2);
        recs[gid] = rec;
        if (attrs.fold_recs) {  // (only triangles that can be drawn are ever looked up by a pixel)
          const float u[9] = {m0, m1, m2, m3, m4, m5, m6, m7, m8};
          float pull[12];
          load_pull_rows(attrs.fold_transforms, b, pull);
          store_fold_record(corner_values, u, 1.0f / fabsf(det), attrs.fold_recs + gid, pull);
        }
      }
    }
  }
  bbs[gid] = bb;
}

// ---------------------------------------------------------------------------------------
// Coarse binning: one 1024-thread workgroup per (image, 256x256-pixel cell) compacts, in
// triangle-id order, the ids whose bbox touches the cell.  The raster kernel's regions
// (4x4 per cell) then scan ~T * ((256 + d) / W)^2 ids instead of all T (d = triangle size).
// ---------------------------------------------------------------------------------------
constexpr int kCellRegions = 4;         // regions per cell edge (cell = 4 x 4 regions)
constexpr int kCoarseThreads = 1024;
// Second level: per raster REGION, the ids of its cell's list whose bbox touches the region (~60-100 of
// the cell's ~400 at 1024^2 / 5k triangles), still in triangle-id order.  k_raster's bin stage then
// scans one or two 64-id chunks per region instead of seven (0.225 -> 0.187 ms for the G-buffer
// kernel at 1024^2 x 32).  A region whose list would not fit kRegionListCap ids keeps reading its
// cell's list (count = -1).
constexpr int kRegionListCap = 512;
constexpr int kCellStash = 2048;  // hits of a cell kept in LDS between k_coarse's two levels (24 KB)

// Round 5: a level ABOVE the cells for large triangle counts.  k_coarse's workgroups each scan ALL T boxes for their
// cell -- at 50k triangles and 2048^2 x 8 that is 512 workgroups x 13 trips of 4096 boxes, 55 us.  A SUPER-CELL is
// kSuperCells x kSuperCells cells (1024 pixels at 64-pixel regions); one workgroup per (image, super-cell, chunk of
// 4096 triangles) tests its chunk against the super-cell once, in two passes -- k_coarse_top<false> counts the hits
// per chunk, k_coarse_top<true> writes them behind the earlier chunks' (so a super-cell's list is one contiguous,
// id-ordered array) -- and a cell's workgroup then scans its super-cell's list (~T / 4 + overlap at 2 x 2 super-cells)
// instead of all T.  Used when T >= kTopMinTriangles and the image has more than one super-cell; lists and results are
// the same either way (the cell lists are built from a superset of their hits, in id order).
constexpr int kSuperCells = 4;
constexpr int kTopUnroll = 4, kTopChunk = kCoarseThreads * kTopUnroll;
constexpr int kTopMinTriangles = 4 * kTopChunk;
template <bool WRITE>
__global__ __launch_bounds__(kCoarseThreads) void k_coarse_top(
    const TriBox *__restrict__ bbs, int T, int W, int H, int sc_x, int sc_per_image, int sc_size, int n_chunks,
    int32_t *__restrict__ counts, int32_t *__restrict__ top_ids) {
  static_assert(kTopUnroll * (kCoarseThreads / kWave) == kWave, "one lane per (sub-chunk, wavefront) count");
  __shared__ int s_counts[kWave];
  const int chunk = (int)blockIdx.x % n_chunks;
  const int rest = (int)blockIdx.x / n_chunks;
  const int sc = rest % sc_per_image, img = rest / sc_per_image;
  const int sy = sc / sc_x, sx = sc - sy * sc_x;
  const int X0 = sx * sc_size, Y0 = sy * sc_size;
  const int X1 = min(X0 + sc_size, W), Y1 = min(Y0 + sc_size, H);
  const int tid = (int)threadIdx.x, lane = tid & (kWave - 1), wave = tid >> 6;
  const TriBox *img_bbs = bbs + (size_t)img * T;
  unsigned long long m[kTopUnroll];
#pragma unroll
  for (int u = 0; u < kTopUnroll; ++u) {
    const int t = chunk * kTopChunk + u * kCoarseThreads + tid;
    const TriBox bb = (t < T) ? img_bbs[t] : TriBox{0u, 0u, 0.0f, 0u};
    const int l = (int)(bb.lr & 0xffffu), r = (int)(bb.lr >> 16);
    const int bt = (int)(bb.bt & 0xffffu), tp = (int)(bb.bt >> 16);
    m[u] = __ballot((l < X1) && (r > X0) && (bt < Y1) && (tp > Y0));   // empty bbox = all zeros
    if (lane == 0) s_counts[u * (kCoarseThreads / kWave) + wave] = __builtin_popcountll(m[u]);
  }
  __syncthreads();
  int incl = s_counts[lane];
  const int own = incl;
#pragma unroll
  for (int off = 1; off < kWave; off <<= 1) {
    const int up = __shfl_up(incl, off);
    if (lane >= off) incl += up;
  }
  int32_t *cnt = counts + ((size_t)img * sc_per_image + sc) * n_chunks;
  if (!WRITE) {
    if (tid == kWave - 1) cnt[chunk] = incl;
    return;
  }
  int base = 0;
  for (int c = 0; c < chunk; ++c) base += cnt[c];   // (workgroup-uniform scalar loads; a handful of chunks)
  const int excl = incl - own;
  int32_t *out = top_ids + ((size_t)img * sc_per_image + sc) * T;
#pragma unroll
  for (int u = 0; u < kTopUnroll; ++u) {
    const int offset = base + __shfl(excl, u * (kCoarseThreads / kWave) + wave);   // (every lane takes part in the shuffle)
    if ((m[u] >> lane) & 1ull) {
      const int pos = offset + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m[u] >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m[u], 0u));
      out[pos] = chunk * kTopChunk + u * kCoarseThreads + tid;
    }
  }
}

__global__ __launch_bounds__(kCoarseThreads) void k_coarse(
    const TriBox *__restrict__ bbs, int T, int W, int H, int cells_x, int cells_per_image, int cell_size,
    int32_t *__restrict__ cell_ids, int32_t *__restrict__ cell_count, float *__restrict__ cell_split,
    int regions_x, int regions_y, int32_t *__restrict__ region_ids, int32_t *__restrict__ region_count,
    int regions_per_xcd, int32_t *__restrict__ order_count, int32_t *__restrict__ order_list,
    const int32_t *__restrict__ top_ids, const int32_t *__restrict__ top_counts, int sc_x, int sc_per_image,
    int n_chunks) {
  static_assert(kCoarseThreads / kWave == kCellRegions * kCellRegions, "one wavefront per region of the cell");
  __shared__ float s_wave_lo[kCoarseThreads / kWave], s_wave_hi[kCoarseThreads / kWave];
  __shared__ uint2 s_hit_box[kCellStash];  // (lr, bt) of the cell's first kCellStash hits, for the second level
  __shared__ int32_t s_hit_id[kCellStash];
  __shared__ int s_order_slot[kCellRegions * kCellRegions], s_order_region[kCellRegions * kCellRegions];
  const int img = (int)blockIdx.x / cells_per_image;
  const int cell = (int)blockIdx.x - img * cells_per_image;
  const int cy = cell / cells_x, cx = cell - cy * cells_x;
  const int X0 = cx * cell_size, Y0 = cy * cell_size;
  const int X1 = min(X0 + cell_size, W), Y1 = min(Y0 + cell_size, H);
  const int tid = (int)threadIdx.x, lane = tid & (kWave - 1), wave = tid >> 6;
  const TriBox *img_bbs = bbs + (size_t)img * T;
  int32_t *out = cell_ids + ((size_t)img * cells_per_image + cell) * T;
  // the boxes to scan: all T, or (k_coarse_top ran) the id-ordered list of this cell's super-cell
  const int32_t *src = nullptr;
  int n_src = T;
  if (top_ids) {   // launch-uniform
    const int sc = (cy / kSuperCells) * sc_x + cx / kSuperCells;
    src = top_ids + ((size_t)img * sc_per_image + sc) * T;
    const int32_t *cnt = top_counts + ((size_t)img * sc_per_image + sc) * n_chunks;
    n_src = 0;
    for (int c = 0; c < n_chunks; ++c) n_src += cnt[c];
#ifdef MR_COARSE_TOP_DEBUG
    if (MR_COARSE_TOP_DEBUG == 1) { src = nullptr; n_src = T; }   // run the top kernels, ignore their lists
#endif
  }
  int n = 0;  // workgroup-uniform
  float lo = INFINITY, hi = -INFINITY;  // range of the depth bounds of this thread's hits
  constexpr int kUnroll = 4;
  static_assert(kUnroll * (kCoarseThreads / kWave) == kWave, "one lane per (sub-chunk, wavefront) count");
  // One barrier per kUnroll x 1024 triangles: every wavefront posts the hit counts of its kUnroll
  // sub-chunks, and after the barrier EVERY wavefront scans the 64 counts itself (lane = sub-chunk x 16
  // + wavefront, the id order of the hits) with six shuffles.  The counts are double-buffered: a
  // wavefront can only overwrite a buffer two trips later, i.e. after a barrier that everyone reached
  // after reading it.  (Two barriers per 1024 triangles before: 100 us at 50k triangles, 2048^2.)
  __shared__ int s_counts[2][kWave];
  int trip = 0;
  // (Measured, no gain: requesting the next trip's boxes a trip ahead; keeping the hits in LDS and
  // copying the list out at the end; skipping 64-triangle chunks by a union box -- see DESIGN.md 4.1.)
  for (int base = 0; base < n_src; base += kUnroll * kCoarseThreads, trip ^= 1) {
    TriBox bb[kUnroll];
    int tri[kUnroll];
#pragma unroll
    for (int u = 0; u < kUnroll; ++u) {
      const int k = base + u * kCoarseThreads + tid;
      tri[u] = src ? (k < n_src ? src[k] : -1) : k;
    }
#pragma unroll
    for (int u = 0; u < kUnroll; ++u) {
      const int t = tri[u];
      bb[u] = (t >= 0 && t < T) ? img_bbs[t] : TriBox{0u, 0u, 0.0f, 0u};
    }
    unsigned long long m[kUnroll];
#pragma unroll
    for (int u = 0; u < kUnroll; ++u) {
      const int l = (int)(bb[u].lr & 0xffffu), r = (int)(bb[u].lr >> 16);
      const int bt = (int)(bb[u].bt & 0xffffu), tp = (int)(bb[u].bt >> 16);
      const bool hit = (l < X1) && (r > X0) && (bt < Y1) && (tp > Y0);  // empty bbox = all zeros
      if (hit) {
        lo = fminf(lo, bb[u].zlo);  // -inf (a triangle that is not tame) sticks
        hi = fmaxf(hi, bb[u].zlo);
      }
      m[u] = __ballot(hit);
      if (lane == 0) s_counts[trip][u * (kCoarseThreads / kWave) + wave] = __builtin_popcountll(m[u]);
    }
    __syncthreads();
    int incl = s_counts[trip][lane];
    const int own = incl;
#pragma unroll
    for (int off = 1; off < kWave; off <<= 1) {
      const int up = __shfl_up(incl, off);
      if (lane >= off) incl += up;
    }
    const int excl = incl - own;
#pragma unroll
    for (int u = 0; u < kUnroll; ++u) {
      const int offset = n + __shfl(excl, u * (kCoarseThreads / kWave) + wave);
      if ((m[u] >> lane) & 1ull) {
        const int t = tri[u];
        const int pos = offset + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m[u] >> 32),
                                                                __builtin_amdgcn_mbcnt_lo((unsigned)m[u], 0u));
        out[pos] = t;
        if (pos < kCellStash) {
          s_hit_box[pos] = make_uint2(bb[u].lr, bb[u].bt);
          s_hit_id[pos] = t;
        }
      }
    }
    n += __shfl(incl, kWave - 1);
  }
  __syncthreads();  // the stash and the list are complete
  // The cell's depth split: midway between the smallest and the largest depth bound of its
  // triangles.  k_raster draws the triangles whose bound lies below it first ("near" class).
  // NaN = some triangle of the cell is not tame: its regions keep strict triangle-id order.
#pragma unroll
  for (int off = kWave / 2; off > 0; off >>= 1) {
    lo = fminf(lo, __shfl_xor(lo, off));
    hi = fmaxf(hi, __shfl_xor(hi, off));
  }
  if (lane == 0) {
    s_wave_lo[wave] = lo;
    s_wave_hi[wave] = hi;
  }
  __syncthreads();
  if (tid == 0) {
#pragma unroll
    for (int w = 0; w < kCoarseThreads / kWave; ++w) {
      lo = fminf(lo, s_wave_lo[w]);
      hi = fmaxf(hi, s_wave_hi[w]);
    }
    cell_count[(size_t)img * cells_per_image + cell] = n;
    cell_split[(size_t)img * cells_per_image + cell] = (n > 0 && lo > -INFINITY) ? 0.5f * lo + 0.5f * hi : NAN;
  }
  // Second level: wavefront w compacts the cell's list (just written; the barriers above order it)
  // for region (w % 4, w / 4) of the cell -- wave-local ballots, no further barriers.
  {
    const int edge = cell_size / kCellRegions;
    const int rx = cx * kCellRegions + wave % kCellRegions, ry = cy * kCellRegions + wave / kCellRegions;
    if (rx < regions_x && ry < regions_y) {  // wave-uniform
      const int RX0 = rx * edge, RY0 = ry * edge;
      const int RX1 = min(RX0 + edge, W), RY1 = min(RY0 + edge, H);
      const size_t region = ((size_t)img * regions_y + ry) * regions_x + rx;
      int32_t *rout = region_ids + region * kRegionListCap;
      int count = 0;
      for (int base = 0; base < n; base += kWave) {
        const int k = base + lane;
        int t = -1;
        uint2 box = make_uint2(0u, 0u);
        if (k < n) {
          if (k < kCellStash) {
            t = s_hit_id[k];
            box = s_hit_box[k];
          } else {
            t = out[k];
            const TriBox bb = img_bbs[t];
            box = make_uint2(bb.lr, bb.bt);
          }
        }
        const int l = (int)(box.x & 0xffffu), r = (int)(box.x >> 16);
        const int bt = (int)(box.y & 0xffffu), tp = (int)(box.y >> 16);
        const bool hit = (l < RX1) && (r > RX0) && (bt < RY1) && (tp > RY0);  // t < 0: the all-zero box misses
        const unsigned long long m = __ballot(hit);
        if (hit) {
          const int pos = count + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
          if (pos < kRegionListCap) rout[pos] = t;
        }
        count += __builtin_popcountll(m);
      }
      if (lane == 0) {
        region_count[region] = count <= kRegionListCap ? count : -1;
        // the region's (XCD, weight class) slot, see region_weight_class
        const int xcd = (int)(region / (size_t)regions_per_xcd);
        s_order_slot[wave] = xcd * kWeightClasses + region_weight_class(count <= kRegionListCap ? count : -1);
        s_order_region[wave] = (int)region;
      }
    } else if (lane == 0) {
      s_order_slot[wave] = -1;
    }
  }
  // File the cell's regions under their slots with ONE atomic per (cell, slot): one per region put up to
  // ~700 same-address atomics in a row on an XCD's busiest class (k_coarse 13 -> 50 us).
  __syncthreads();
  if (tid < kCellRegions * kCellRegions) {
    const int slot = s_order_slot[tid];
    int leader = tid, n = 0, rank = 0;
    for (int w = 0; w < kCellRegions * kCellRegions; ++w) {
      if (slot >= 0 && s_order_slot[w] == slot) {
        leader = min(leader, w);
        rank += w < tid ? 1 : 0;
        n += 1;
      }
    }
    int base = 0;
    if (slot >= 0 && leader == tid) base = atomicAdd(&order_count[slot], n);
    base = __shfl(base, leader);   // (lanes 0..15 of the first wavefront)
    if (slot >= 0) order_list[(size_t)slot * regions_per_xcd + base + rank] = s_order_region[tid];
  }
}

// Per-pixel running z-buffer state (registers).
struct PixelState {
  float z, b0, b1, b2;
  int id;
  int ent;   // SHADE: bin entry (slot) of the winner, for the epilogue's record lookup
};

// ---------------------------------------------------------------------------------------
// The raster kernel: LDS bin of full records, coverage / depth split.
// ---------------------------------------------------------------------------------------
constexpr int kEntryDw = 20;   // LDS entry: 20 dwords = 80 B (see file header)
constexpr int kSubCap = 64;    // entries one wavefront may add per round
constexpr int kWaves = kThreads / kWave;
constexpr int kBin2Cap = kSubCap * kWaves;
static_assert(kSubCap >= kWave, "a 64-triangle chunk must always fit an empty sub-bin (progress)");

// Entry layout (dwords): 0-3 a0 a1 b0 b1 | 4-7 c0 c1 a2 b2 | 8 c2 | 9 -tolerance of the conservative coverage
// test | 10-15 z0 w0 z1 w1 z2 w2 | 16 bbox clipped to the region, region-relative: l | bottom << 16 |
// 17 (w-1) | (h-1) << 16 | 18 id | 19 unused
// where edge_i(px, py) = (a_i * px + b_i * py) + c_i  (cpp:46).  (While the bin is being filled the id sits in
// dword 9 and 18-19 hold the entry's (chunk, rank in class) tag; the final placement writes the layout above.)
struct Entry {
  float4 q0, q1, q2, q3;
  uint4 tail;   // bbox (2), id, unused
};

typedef float v2f __attribute__((ext_vector_type(2)));  // packed fp32 (v_pk_mul_f32 / v_pk_add_f32)
typedef unsigned short v2u16 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned pk_sub_u16(unsigned a, unsigned b) {  // per-half a - b, wrapping
  return __builtin_bit_cast(unsigned, __builtin_bit_cast(v2u16, a) - __builtin_bit_cast(v2u16, b));
}
__device__ __forceinline__ unsigned pk_min_u16(unsigned a, unsigned b) {
  return __builtin_bit_cast(unsigned, __builtin_elementwise_min(__builtin_bit_cast(v2u16, a),
                                                                __builtin_bit_cast(v2u16, b)));
}

__device__ __forceinline__ Entry read_entry(const float *s_ent, int e) {
  const float *p = s_ent + __mul24(e, kEntryDw);
  Entry r;
  r.q0 = *(const float4 *)(p);
  r.q1 = *(const float4 *)(p + 4);
  r.q2 = *(const float4 *)(p + 8);
  r.q3 = *(const float4 *)(p + 12);
  r.tail = *(const uint4 *)(p + 16);
  return r;
}

// Exact trivial reject of one triangle against a rectangle of pixel centres
// [pxlo, pxhi] x [pylo, pyhi]: edge i is evaluated with the reference's own un-fused
// expression (cpp:46) at the centre that maximises it.  fl(a*px), fl(.+.) are monotone,
// so every pixel centre of the rectangle gets a value <= this one; a negative value
// means edge i rejects the whole rectangle.  NaNs compare false and never reject.
__device__ __forceinline__ bool rect_outside_edge(float a, float b, float c, float pxlo, float pxhi,
                                                  float pylo, float pyhi) {
  const float fx = (a >= 0.0f) ? pxhi : pxlo;
  const float fy = (b >= 0.0f) ? pyhi : pylo;
  const float e = (a * fx + b * fy) + c;
  return e < 0.0f;
}

__device__ __forceinline__ bool rect_outside_triangle(const float4 q0, const float4 q1, const float m8,
                                                      float pxlo, float pxhi, float pylo, float pyhi) {
  // bitwise |: all three edges are evaluated straight-line (no dependent branches)
  return (int)rect_outside_edge(q0.x, q0.z, q1.x, pxlo, pxhi, pylo, pyhi) |
         (int)rect_outside_edge(q0.y, q0.w, q1.y, pxlo, pxhi, pylo, pyhi) |
         (int)rect_outside_edge(q1.z, q1.w, m8, pxlo, pxhi, pylo, pyhi);
}

#ifndef MR_TILE_W
#define MR_TILE_W 16  // pixels per tile row (tile = MR_TILE_W x 64 / MR_TILE_W), see k_raster
#endif
#ifndef MR_RASTER_WAVES
#define MR_RASTER_WAVES 7  // register-allocation hint: yields 71 VGPRs (<= 72 = 7 waves per SIMD)
#endif
// b_i = e_i / s, correctly rounded, for three numerators over ONE denominator (cpp:385-387).
// fp32 '/' lowers to  v_div_scale x2, v_rcp, 2 FMAs refining the reciprocal, v_mul + 3 FMAs
// refining the quotient, v_div_fmas, v_div_fixup.  When the operands are far from the exponent
// range's ends -- s in [2^-20, 2^20], every numerator in [2^-60, ~s] -- v_div_scale scales
// nothing, v_div_fmas is a plain FMA and v_div_fixup returns its input, so the result is exactly
// the FMA chain below; the refined reciprocal depends on s alone and is computed once (22 VALU
// instructions and one v_rcp instead of 30 and three).  Anything else -- on-edge pixels with a
// zero numerator, extreme magnitudes -- takes the ordinary division.
__device__ __forceinline__ void div3_common_denominator(float n0, float n1, float n2, float s,
                                                        unsigned long long valid, float &q0, float &q1,
                                                        float &q2) {
  // plain = min(n) >= 2^-60 and s in [2^-20, 2^20].  One path for the whole wavefront (the ordinary division is
  // correct for every operand): lanes without a candidate (`valid` clear) must not force the slow one.  (Two
  // ballots combined on the scalar side: the ballot of one compound condition went through a 0 / 1 vector
  // register and a second compare.)
  const unsigned long long not_plain =
      __builtin_amdgcn_ballot_w64(!(__builtin_fminf(__builtin_fminf(n0, n1), n2) >= 0x1p-60f)) |
      __builtin_amdgcn_ballot_w64(__builtin_amdgcn_fmed3f(s, 0x1p-20f, 0x1p20f) != s);
  if ((not_plain & valid) == 0ull) {
    const float r0 = __builtin_amdgcn_rcpf(s);
    const float r = __builtin_fmaf(__builtin_fmaf(-s, r0, 1.0f), r0, r0);
    auto quotient = [&](const float n) {
      float q = n * r;
      q = __builtin_fmaf(__builtin_fmaf(-s, q, n), r, q);
      q = __builtin_fmaf(__builtin_fmaf(-s, q, n), r, q);
      return q;
    };
    q0 = quotient(n0); q1 = quotient(n1); q2 = quotient(n2);
  } else {
    q0 = n0 / s; q1 = n1 / s; q2 = n2 / s;
  }
}

// A hardware hazard LLVM does not cover on this target (found at the end of round 4 with the fuzzer: attribute 0 of
// lanes 12-15 of every 16 came out as the pixel's triangle id).  A MUBUF store of more than 64 bits reads its data
// registers over several cycles; a vector instruction that overwrites one of them in the very next issue slot
// corrupts what the last lanes store.  The hazard recognizer inserts the documented wait state only for stores
// WITHOUT a register soffset (GCNHazardRecognizer::createsVALUHazard); the tile walk's stores carry their tile
// offset there.  The data registers are kept alive through an `s_nop` behind the store -- an asm with side
// effects stays behind the store, and nothing can overwrite its inputs before it.
#ifndef MR_WIDE_STORE_NOPS
#define MR_WIDE_STORE_NOPS 1   // s_nop operand: 1 = two wait states; -1 = no protection (to reproduce the corruption)
#endif
typedef unsigned mr_v4u __attribute__((ext_vector_type(4)));
typedef unsigned mr_v3u __attribute__((ext_vector_type(3)));
template <int AUX>
__device__ __forceinline__ void store_b128_soffset(const mr_v4u d, const __amdgpu_buffer_rsrc_t rs, const unsigned voffset,
                                                   const int soffset) {
  __builtin_amdgcn_raw_buffer_store_b128(d, rs, voffset, soffset, AUX);
#if MR_WIDE_STORE_NOPS >= 0
  asm volatile("s_nop %4" ::"v"(d.x), "v"(d.y), "v"(d.z), "v"(d.w), "n"(MR_WIDE_STORE_NOPS));
#endif
}
template <int AUX>
__device__ __forceinline__ void store_b96_soffset(const mr_v3u d, const __amdgpu_buffer_rsrc_t rs, const unsigned voffset,
                                                  const int soffset) {
  __builtin_amdgcn_raw_buffer_store_b96(d, rs, voffset, soffset, AUX);
#if MR_WIDE_STORE_NOPS >= 0
  asm volatile("s_nop %3" ::"v"(d.x), "v"(d.y), "v"(d.z), "n"(MR_WIDE_STORE_NOPS));
#endif
}

// R = region edge in pixels: 64, or 32 for small launches (configs[1]: 8 x 256^2 is only 128
// regions of 64^2 -- half the CUs idle and >256 candidates per region; 512 regions of 32^2 fill
// the chip with one bin round each).
// PROBE: 0 in production.  Non-zero values (only instantiated with -DMR_PROBES, see
// mesh_raster_debug.h) switch stages off for stage timing and leave the outputs undefined.
// SHADE: render()'s deferred shading (shade_pixel.h) runs as the epilogue of a region's LAST bin round,
// on the pixel state the walk still holds in registers, and the RGBA image leaves next to the
// G-buffer: the separate k_shade_forward pass (16 B/px read again + a second dependent gather level,
// id -> corner record) disappears.
struct RasterShade {
  const CornerRec *__restrict__ corners;  // [B*T] (k_corner_setup)
  Lights lights;
  float *__restrict__ rgba;               // [B,H,W,4], image rows (row 0 = top)
  uint32_t *__restrict__ rgba8;           // nullptr, or [B,H,W] 8-bit RGBA frames of the same image (the
                                          // examples' `(image * 255.0).astype(np.uint8)`, loss.hip's to_u8)
  int keep_z;                             // 0: the caller does not want the depth plane -- it is then only
                                          // written between the bin rounds of a crowded region (as state)
  // INTERP (round 4): rasterize()'s attribute interpolation (src/mesh_renderer/rasterize.py:118-150) as the
  // epilogue instead of the shading -- k_interp_forward_rec's arithmetic on the pixel state in registers
  const float *__restrict__ attr_records;  // [B*T][3 * AP]: the corners' attributes, [corner][attribute] (interp_fused.hip)
  const float *__restrict__ background;    // [A]
  float *__restrict__ attr_out;            // [B,H,W,A], G-buffer row order (row 0 = bottom, like ids / bary)
  int A;
  // Round 4: nullptr, or [B][ceil(H / 64)][ceil(W / 64)] bytes out (64-pixel regions only): 1 = the region is a whole
  // 64 x 64 block without a single candidate triangle -- every pixel of it is background in the G-buffer and transparent
  // black in the image.  Consumers (the loss, the shading backward) skip such blocks without reading them.
  uint8_t *__restrict__ empty_map;
  // NORMS (round 5; with attr_records = [corner][normal 3 | position 3 | 2 of padding], A = 6, attr_out = nullptr): the
  // specular term's across-pixels norm (render.py:342-348) as the epilogue -- every covered pixel's squared
  // reflection . camera dot product per light, summed per region; shade_spec.hip's norm pass then does not run.
  const float *__restrict__ camera = nullptr;        // [B,3]
  float *__restrict__ norm_partials = nullptr;       // [regions][8]: sums for up to four lights, the covered-pixel count, padding
};

#ifndef MR_RASTER_STORE_AUX
#define MR_RASTER_STORE_AUX 2  // cache policy of the G-buffer / RGBA stores: 2 = nontemporal (written once, read by a
                                // later kernel from HBM anyway): kernel -5 %, step -2 % against 0 (same-box A/B)
#endif
#ifndef MR_RASTER_STORE_AUX_IDS
#define MR_RASTER_STORE_AUX_IDS 0  // the id plane alone through the caches: its 64-byte runs are the shortest of the tile's
                                   // stores.  Same-box A/B, two boxes: fused kernel 0.261 -> 0.250 / 0.2496 -> 0.2483 ms,
                                   // G-buffer kernel 0.189 -> 0.164 / 0.161 -> 0.160 (never slower; barycentrics or the
                                   // depth plane cached: slower or equal)
#endif
#ifndef MR_BARY_STAGE
#define MR_BARY_STAGE 0  // round 4 (measured, OFF: WRITE_SIZE unchanged at 1.19x, kernel +3 %): the barycentric plane leaves as 16-byte-per-lane stores after a transpose through
                         // LDS (see "staged barycentric store" in k_raster) instead of one 12-byte store per lane
#endif
#ifndef MR_RASTER_STORE_AUX_Z
#define MR_RASTER_STORE_AUX_Z MR_RASTER_STORE_AUX
#endif
#ifndef MR_EPI_DIFF_BASIS
#define MR_EPI_DIFF_BASIS 1   // the shading epilogue's LDS records in the difference basis (see the record loader)
#endif
#ifndef MR_RASTER_STORE_AUX_RGBA
#define MR_RASTER_STORE_AUX_RGBA MR_RASTER_STORE_AUX   // the image plane of the fused forward (its consumer, the loss, reads it back to front)
#endif
#ifndef MR_COARSE_TOP
#define MR_COARSE_TOP 1   // 0: k_coarse always scans all T boxes (A/B)
#endif
#ifndef MR_EPI_LDS_RECORDS
#define MR_EPI_LDS_RECORDS 1  // round 4: the shading epilogue reads its winners' corner records per lane from LDS
                              // (see "corner records in LDS" in k_raster) instead of one winner at a time through the scalar cache
#endif
#ifndef MR_RASTER_XREC
#define MR_RASTER_XREC 64         // extra record slots of the crowded-launch instantiation (0: no such instantiation).
                                  // configs[3], same box, fused forward: none 0.3169 / 0.3148 ms; 128 slots at four workgroups
                                  // per CU 0.3116 / 0.3109; 64 slots (144 in all) at five 0.3057; 128 at five (spills) 0.3135
#endif
#ifndef MR_RASTER_XREC_WAVES
#define MR_RASTER_XREC_WAVES 5
#endif
#ifndef MR_RASTER_XREC_DENSITY
#define MR_RASTER_XREC_DENSITY 32  // triangles per 64 x 64 pixels of image (T * 4096 / (W H)) from which it is used
#endif
#ifndef MR_RASTER_SHADE_WAVES
#define MR_RASTER_SHADE_WAVES 6  // measured against 7 (more spills) and 5: 0.336 / 0.352 / 0.347 ms at 1024^2 x 32
#endif
#ifndef MR_RASTER_INTERP_STORE_AUX
#define MR_RASTER_INTERP_STORE_AUX 0   // the interpolated image's stores: per-lane pieces of A floats -- through the caches (merged in L2)
#endif
#ifndef MR_RASTER_INTERP_STAGE
#define MR_RASTER_INTERP_STAGE 1   // see stage_slot in k_raster
#endif
#ifndef MR_SETUP_TRANSFORMS
#define MR_SETUP_TRANSFORMS 1   // render()'s clip-space transform inside k_setup (0: its own launch, k_vertex_transform)
#endif
#ifndef MR_RASTER_RGBA8_NT
#define MR_RASTER_RGBA8_NT 0   // the 8-bit frames' store policy: no measurable difference (same-box A/B, step and kernel)
#endif
#ifndef MR_RASTER_NORMS_WAVES
#define MR_RASTER_NORMS_WAVES 6   // whole call at 1024^2 x 32: 4 -> 0.254, 5 -> 0.233, 6 -> 0.228 ms (same box)
#endif
#ifndef MR_RASTER_INTERP_WAVES
#define MR_RASTER_INTERP_WAVES 5
#endif
// XREC (round 5, SHADE only): that many extra corner-record slots of LDS behind the bin, for launches whose regions are
// crowded (configs[3]: ~170 entries per 64 x 64 region against the 106 record slots the bin's unused top offers): the
// epilogue then reads its winners' records per lane from LDS there too instead of one winner at a time through the scalar
// cache.  64 slots = 7 KB more LDS: five workgroups per CU instead of six.  Chosen on the host (launch_k_raster_probe).
template <int R, int PROBE, bool SHADE, int INTERP = 0, int AX = 0, int XREC = 0, bool NORMS = false>
__global__ __launch_bounds__(kThreads, SHADE ? (XREC ? MR_RASTER_XREC_WAVES : MR_RASTER_SHADE_WAVES) : NORMS ? MR_RASTER_NORMS_WAVES : INTERP >= 12 ? 4 : INTERP ? MR_RASTER_INTERP_WAVES : MR_RASTER_WAVES) void k_raster(
    const TriRec *__restrict__ recs, const TriBox *__restrict__ bbs,
    const float *__restrict__ pxtab, const float *__restrict__ pytab, int T, int W, int H,
    int regions_x, int regions_per_image, int n_regions, int regions_per_xcd,
    const int32_t *__restrict__ cell_ids, const int32_t *__restrict__ cell_count,
    const float *__restrict__ cell_split, int cells_x,
    int cells_per_image, const int32_t *__restrict__ region_ids,
    const int32_t *__restrict__ region_count, const int32_t *__restrict__ order_count,
    const int32_t *__restrict__ order_list, int32_t *__restrict__ ids, float *__restrict__ bary,
    float *__restrict__ zbuf, const RasterShade shade) {
  static_assert(R == 64 || R == 32, "region edge");
  static_assert(XREC == 0 || (SHADE && XREC % 4 == 0), "extra record slots: the shading epilogue's");
  static_assert(!SHADE || PROBE == 0, "the shading epilogue has no timing probes");
  static_assert(INTERP == 0 || (!SHADE && PROBE == 0 && INTERP % 4 == 0 && INTERP <= 16), "one epilogue at a time");
  static_assert(AX == 0 || (INTERP > 0 && AX <= INTERP && AX > INTERP - 4), "a fixed attribute count belongs to its padded variant");
  static_assert(!NORMS || (INTERP == 8 && AX == 6), "the norm epilogue interpolates normals and positions");
  constexpr bool EPI = SHADE || INTERP > 0;   // an epilogue runs on a region's last bin round
  // A wavefront's tile is kTileW x kTileH pixels, one per lane.  16 x 4: every row of a tile's
  // G-buffer stores is a whole, aligned 64-byte sector (16 ids / depths) or three of them (16
  // barycentric triples); with 8 x 8 tiles two wavefronts shared each sector and the L2 had to
  // merge their halves (measured: the kernel ran FASTER with fewer workgroups in flight).
  constexpr int kTileW = MR_TILE_W, kTileH = kWave / kTileW;
  static_assert(kTileW == 8 || kTileW == 16 || kTileW == 32, "tile width");
  constexpr int kTilesX = R / kTileW, kTilesY = R / kTileH;
  constexpr int kTiles = kTilesX * kTilesY;  // tiles per region: 64 or 16
  constexpr int kMaskWords = kBin2Cap / 32;
  constexpr int kEntDw = kBin2Cap * kEntryDw + XREC * 28;   // the bin, then XREC more record slots (records grow down from the end)
  __shared__ __attribute__((aligned(16))) float s_ent[kEntDw];
  // LDS budget at R = 64: 20480 (entries) + 2048 (masks / bin bookkeeping) + 512 (pixel centres)
  // = 23040 B = 18 allocation granules of 1280 B, so that SEVEN workgroups fit the 160 KB of a CU.
  // The bin-stage bookkeeping is dead by the time the tile masks are built and shares their storage.
  constexpr int kRoundChunks = kThreads;  // chunks of 64 triangles scanned per round
  constexpr int kSharedDw = kTiles * kMaskWords > kRoundChunks + 3 * kWaves ? kTiles * kMaskWords
                                                                             : kRoundChunks + 3 * kWaves;
  // [0, 2R): pixel centres of the region's columns, then rows; [2R, 2R + kSharedDw): tile masks /
  // bin bookkeeping.  One array: a lane's pixel-centre and mask-word addresses then differ by a
  // constant and share one address register.
  __shared__ __attribute__((aligned(16))) unsigned s_misc[2 * R + kSharedDw];
  float (*s_pxy)[R] = (float (*)[R])s_misc;  // pixel centres of the region's columns / rows
  unsigned *s_shared = s_misc + 2 * R;
  unsigned (*s_tmask)[kMaskWords] = (unsigned (*)[kMaskWords])s_shared;  // per tile: entries touching it
  int *s_chunk_count = (int *)s_shared;   // [kRoundChunks]
  int *s_count = s_chunk_count + kRoundChunks, *s_stop = s_count + kWaves, *s_wave_total = s_stop + kWaves;

  // Hardware block b runs on XCD b % 8 (round-robin dispatch) and is the (b / 8)-th workgroup there:
  // it takes the (b / 8)-th region of that XCD's range in weight-class order, heaviest class first.
  int region;
  {
    const int xcd = (int)blockIdx.x % kXcds;
    int pos = (int)blockIdx.x / kXcds;
    if (xcd * regions_per_xcd + pos >= n_regions || pos >= regions_per_xcd) return;  // padding block (whole workgroup)
    const int32_t *counts = order_count + xcd * kWeightClasses;
    int cls = 0;
    // (Measured in round 6, no effect: half or all of the XCD's EMPTY regions -- store-only -- opening the order instead of
    //  closing it, so that the launch's first round of workgroups does not bin in lockstep with nothing in the store
    //  queue: whole call 0.193-0.195 / 0.194 / 0.196 ms, profiles/r06_raster_empty_first.txt.)
#pragma unroll
    for (int c = 0; c < kWeightClasses - 1; ++c) {
      const int n = counts[c];   // wave-uniform scalar loads
      if (cls == c && pos >= n) { pos -= n; cls = c + 1; }
    }
    region = order_list[(size_t)(xcd * kWeightClasses + cls) * regions_per_xcd + pos];
  }
  const int img = region / regions_per_image;
  const int rr = region - img * regions_per_image;
  const int ry = rr / regions_x;
  const int rx = rr - ry * regions_x;
  const int X0 = rx * R, Y0 = ry * R;
  const int X1 = min(X0 + R, W), Y1 = min(Y0 + R, H);

  const int tid = (int)threadIdx.x;
  const int lane = tid & (kWave - 1);
  // INTERP: the attribute count -- a compile-time constant in the instantiations for the usual counts (AX: 3, 9), where
  // the staging stores' shapes fold to straight-line code (with a run-time count they were ~20 branches per tile on
  // spilled lane masks), else the launch's
  const int attr_n = AX ? AX : shade.A;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const TriRec *img_recs = recs + (size_t)img * T;
  const TriBox *img_bbs = bbs + (size_t)img * T;
  const size_t img_px = (size_t)img * H * W;
  // this region's coarse cell: the id-ordered list of triangles whose bbox touches it
  const int cell = (ry / kCellRegions) * cells_x + (rx / kCellRegions);
  // ... or, where k_coarse's second level could fit it, the shorter list of those that touch this region
  const int region_n = region_count[region];  // workgroup-uniform
  const int32_t *cand = region_n >= 0 ? region_ids + (size_t)region * kRegionListCap
                                      : cell_ids + ((size_t)img * cells_per_image + cell) * T;
  const int n_cand = region_n >= 0 ? region_n : cell_count[(size_t)img * cells_per_image + cell];
  if (R == 64 && shade.empty_map && threadIdx.x == 0)
    shade.empty_map[region] = (n_cand == 0 && X1 - X0 == R && Y1 - Y0 == R) ? 1 : 0;
  // Front-to-back classes (see depth_lower_bound): when every triangle of the cell is tame, the
  // bin holds the candidates whose depth bound lies below the cell's split first ("near"), in id
  // order, then the others ("far"), in id order.  A tile whose pixels all hold a depth below the
  // split after its near words cannot be changed by any far candidate (each of those is at least
  // as deep as the split) and skips them.  Because the classes break the id order, ties are
  // resolved explicitly in that mode (equal depth -> larger id, as the sequential loop of
  // cpp:401-409 does); otherwise (NaN split) there is one class and the reference's own test.
  const float split_raw = cell_split[(size_t)img * cells_per_image + cell];
  const bool ordered = split_raw == split_raw;  // workgroup-uniform
  const float split = ordered ? split_raw : INFINITY;

  // pixel-centre extents of the region (tables are monotone in the pixel index)
  const float rpxlo = pxtab[X0], rpxhi = pxtab[X1 - 1], rpylo = pytab[Y0], rpyhi = pytab[Y1 - 1];
  // The region's pixel centres live in LDS: the tile loop must not issue global LOADS,
  // because vmcnt is in-order on gfx950 and a load's wait would also drain the previous
  // tile's G-buffer stores (measured: +0.1 ms at 1024^2 x 32).
  static_assert(2 * R <= kThreads, "one thread per table slot");
  if (tid < R) s_pxy[0][tid] = pxtab[min(X0 + tid, W - 1)];
  else if (tid < 2 * R) s_pxy[1][tid - R] = pytab[min(Y0 + tid - R, H - 1)];

  // ---- stage 2a: which bin entries touch which tile ------------------------------------
  // One thread per bin entry walks the tiles under the entry's bbox and applies the exact
  // trivial reject against each; survivors set their bit in the tile's 256-bit mask.  This
  // costs ~(entries x tiles-per-entry) lane-tests per region instead of (tiles x entries).
  auto build_tile_masks = [&](const int n_near, const int far_base, const int n_far) {
    for (int i = tid; i < kSharedDw; i += kThreads) s_shared[i] = 0u;
    // pixel-centre extents of tile column t: s_pxy[0][t kTileW] .. s_pxy[0][min(t kTileW + kTileW - 1, last)]
    const int last_x = X1 - 1 - X0, last_y = Y1 - 1 - Y0;
    __syncthreads();
    // Round 3: SEVERAL threads per entry.  A region's bin holds ~40 entries at 1024^2 / 5k triangles:
    // with one thread each, one wavefront of the four walked up to 64 tiles per entry serially while
    // the other three waited at the barrier -- 24 us of the kernel's 150 (stage probes, profiles/
    // r03_raster_stage_times.txt).  `per` threads (a power of two, as many as fit the workgroup, at
    // most one per tile row) now share an entry's tile rows.
    const int total = n_near + n_far;
    int per_log = 0;
    while (per_log < 4 && (total << (per_log + 1)) <= kThreads) ++per_log;   // workgroup-uniform
    const int idx = tid >> per_log, sub = tid & ((1 << per_log) - 1), per = 1 << per_log;
    if (idx < total) {
      const int e = idx < n_near ? idx : far_base + (idx - n_near);
      const float *p = s_ent + e * kEntryDw;
      const uint2 box = *(const uint2 *)(p + 16);
      const float4 q0 = *(const float4 *)(p), q1 = *(const float4 *)(p + 4);
      const float m8 = p[8];
      // tile range under the bbox (already clipped to the region; never empty)
      const int bl = (int)(box.x & 0xffffu), bb = (int)(box.x >> 16);
      const int tx0 = bl / kTileW, tx1 = (bl + (int)(box.y & 0xffffu)) / kTileW;
      const int ty0 = bb / kTileH, ty1 = (bb + (int)(box.y >> 16)) / kTileH;
      const unsigned bit = 1u << (e & 31);
      const int word = e >> 5;
      for (int ty = ty0 + sub; ty <= ty1; ty += per) {
        const float ylo = s_pxy[1][ty * kTileH], yhi = s_pxy[1][min(ty * kTileH + kTileH - 1, last_y)];
        for (int tx = tx0; tx <= tx1; ++tx) {
          if (!rect_outside_triangle(q0, q1, m8, s_pxy[0][tx * kTileW],
                                     s_pxy[0][min(tx * kTileW + kTileW - 1, last_x)], ylo, yhi))
            atomicOr(&s_tmask[ty * kTilesX + tx][word], bit);
        }
      }
    }
    __syncthreads();
  };

  // ---- stage 2b: the tile walk --------------------------------------------------------
  // Per-lane constants of the tile walk: lane (lx, ly) of a tile.  The G-buffer is
  // addressed through three raw buffer descriptors anchored at this REGION's first pixel (scalar
  // registers, built once): a tile's accesses are then  descriptor + scalar tile offset +
  // 32-bit per-lane offset, and the tile loop spends no vector instructions on address
  // arithmetic (16.7 M of the kernel's 131.6 M VALU instructions per launch at 1024^2 x 32 were
  // tile-loop overhead).  Offsets stay below 64 rows x 65535 px x 12 B = 50 MB.
  const int lx = lane % kTileW, ly = lane / kTileW;
  const unsigned lane_pix = (unsigned)(ly * W + lx);  // pixel offset inside a tile
  const unsigned lane_xy0 = (unsigned)lx | ((unsigned)ly << 16);
  const float *lane_px = &s_pxy[0][lx], *lane_py = &s_pxy[1][ly];
  const size_t region_pix = img_px + (size_t)Y0 * W + X0;
  constexpr int kRsrcWord3 = 0x00020000;  // raw 32-bit buffer on gfx9-family targets
  const __amdgpu_buffer_rsrc_t rs_ids = __builtin_amdgcn_make_buffer_rsrc(ids + region_pix, 0, 0x7fffffff, kRsrcWord3);
  const __amdgpu_buffer_rsrc_t rs_z = __builtin_amdgcn_make_buffer_rsrc(zbuf + region_pix, 0, 0x7fffffff, kRsrcWord3);
  const __amdgpu_buffer_rsrc_t rs_bary = __builtin_amdgcn_make_buffer_rsrc(bary + 3 * region_pix, 0, 0x7fffffff, kRsrcWord3);
  // RGBA rows are flipped (render.py:384-386: image row H-1-y shows G-buffer row y).  The descriptor
  // is anchored at the image row of the region's LAST G-buffer row -- the lowest address the region
  // writes; for a ragged top region that row lies above the image and only serves as an origin --
  // so that tile and lane offsets are non-negative: tile (tx, ty), lane (lx, ly) ->
  // ((R - kTileH - ty kTileH) W + tx kTileW) + ((kTileH - 1 - ly) W + lx) pixels.
  const __amdgpu_buffer_rsrc_t rs_rgba = __builtin_amdgcn_make_buffer_rsrc(
      SHADE ? shade.rgba + 4 * ((ptrdiff_t)img_px + ((ptrdiff_t)H - R - Y0) * W + X0) : nullptr, 0, 0x7fffffff,
      kRsrcWord3);
  const CornerRec *img_corners = SHADE ? shade.corners + (size_t)img * T : nullptr;
  const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(
      (INTERP && !NORMS) ? shade.attr_out + region_pix * (size_t)attr_n : nullptr, 0, 0x7fffffff, kRsrcWord3);
  const float *img_attr_records = INTERP ? shade.attr_records + (size_t)img * T * (3 * INTERP) : nullptr;
  const unsigned lane_out = INTERP ? lane_pix * (unsigned)attr_n * 4u : 0u;   // byte offset of the lane's pixel inside a tile
  int ent_init = 0;   // slot a pixel without a winner looks up (INTERP: the background's record, set per round)
  // INTERP: a pixel's A floats are A * 4 bytes apart from its neighbour's: stored per lane they reach memory as
  // 16-byte (or smaller) pieces at a stride of A * 4 bytes -- partial sectors from every store instruction (measured
  // at A = 9: 505 us for the kernel; with nontemporal stores 1.4 ms).  A tile's 64 x A floats are four rows of
  // 16 A contiguous floats: the lanes park their values pixel-major in an LDS slot of their wavefront
  // (write width chosen by A's alignment: conflict-free for odd A and A = 4, 12) and read the tile back as
  // 16-byte chunks, lane + 64 i: every store instruction then writes whole, contiguous 1 KB pieces of a row.
  // (The INTERP instantiations run at 4-5 waves per SIMD on their registers: the slot costs no occupancy.)
  float *stage_slot = nullptr;
  unsigned stage_goff[INTERP > 0 ? INTERP / 4 : 1] = {};
  __shared__ __attribute__((aligned(16))) float s_background[INTERP > 0 ? INTERP : 4];   // (INTERP only)
  // NORMS: per-lane sums of rdc^2 (four lights) and the lane's covered pixels, over every tile it walks
  [[maybe_unused]] float norm_acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
  [[maybe_unused]] float norm_covered = 0.0f;
  [[maybe_unused]] float norm_cam[3] = {0.0f, 0.0f, 0.0f}, norm_lp[4][3] = {};
  [[maybe_unused]] const int norm_lights = NORMS ? min(shade.lights.L, 4) : 0;
  if constexpr (NORMS) {
    if (tid < INTERP) s_background[tid] = -1.0f;   // render.py:197 (unused by the epilogue below: uncovered pixels are counted)
#pragma unroll
    for (int c = 0; c < 3; ++c) norm_cam[c] = shade.camera[(size_t)img * 3 + c];   // (wave-uniform: scalar loads)
#pragma unroll
    for (int l = 0; l < 4; ++l)
#pragma unroll
      for (int c = 0; c < 3; ++c)
        norm_lp[l][c] = l < norm_lights ? shade.lights.pos[((size_t)img * shade.lights.L + l) * 3 + c] : 0.0f;
  } else if constexpr (INTERP > 0) {
    __shared__ __attribute__((aligned(16))) float s_stage[kWaves * kWave * INTERP];
    stage_slot = s_stage + wave * (kWave * INTERP);
    // The background colour in LDS (the bin stage's barriers come before its first use): read per tile from memory it
    // was 9-12 VECTOR loads, and vmcnt retires in order -- their wait drained the previous tile's stores
    // (SQ_INSTS_VMEM_RD 13 per tile; the same trap as the corner records of the shading epilogue).
    if (tid < INTERP) s_background[tid] = tid < attr_n ? shade.background[tid] : 0.0f;
    const unsigned chunks_per_row = (unsigned)(kTileW * attr_n) / 4u;      // 16-byte chunks in one tile row: 4 A
#pragma unroll
    for (int i = 0; i < INTERP / 4; ++i) {
      const unsigned j = (unsigned)lane + 64u * i;                          // this lane's chunk in trip i
      const unsigned r = j / chunks_per_row;                                // tile row
      stage_goff[i] = r * (unsigned)(W * attr_n) * 4u + (j - r * chunks_per_row) * 16u;
    }
  }
  const float *s_lights_ptr = nullptr;
  if constexpr (SHADE) {  // (no LDS at all in the G-buffer-only instantiation: its 23040 B are exactly 18 granules)
    __shared__ float s_lights[(LightsInLds::kFloats + 31) / 32 * 32];  // 3 + 6 x 32 lights: 780 B, inside the 19th LDS granule
    LightsInLds::stage(shade.lights, img, s_lights, tid);  // (the bin stage's barriers come before any use)
    s_lights_ptr = s_lights;
  }
  const LightsInLds lights{s_lights_ptr, shade.lights.L, shade.lights.amb != nullptr};
  typedef float v3f __attribute__((ext_vector_type(3)));
  typedef unsigned v3u __attribute__((ext_vector_type(3)));
  // Empty regions (round 4): no candidate at all -- the cleared G-buffer (id 0, depth 1, barycentrics 0;
  // cpp:313-321) and, with the shading epilogue, transparent black.  The values do not depend on the lane,
  // so the lanes are laid along the region's ROWS: every store instruction covers 64 / R whole rows of the
  // region, one contiguous run each (256 B of ids / depths, 768 B of barycentrics, 1 KB of RGBA at R = 64)
  // instead of a tile's 64- / 192-byte pieces, and none of the bin / mask / walk machinery runs.
  if (PROBE == 0 && n_cand == 0 && X1 - X0 == R && Y1 - Y0 == R) {  // workgroup-uniform
    constexpr int kRowsPerInst = kWave / R;
    const int x = lane % R;
    for (int y = wave * kRowsPerInst + lane / R; y < R; y += kWaves * kRowsPerInst) {
      const unsigned pix = (unsigned)(y * W + x);
      __builtin_amdgcn_raw_buffer_store_b32(0u, rs_ids, pix * 4u, 0, MR_RASTER_STORE_AUX_IDS);
      if (!EPI || shade.keep_z)
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, 1.0f), rs_z, pix * 4u, 0, MR_RASTER_STORE_AUX_Z);
      __builtin_amdgcn_raw_buffer_store_b96(v3u{0u, 0u, 0u}, rs_bary, pix * 12u, 0, MR_RASTER_STORE_AUX);
      if constexpr (SHADE) {
        typedef unsigned v4u __attribute__((ext_vector_type(4)));
        __builtin_amdgcn_raw_buffer_store_b128(v4u{0u, 0u, 0u, 0u}, rs_rgba, (unsigned)((R - 1 - y) * W + x) * 16u, 0,
                                               MR_RASTER_STORE_AUX_RGBA);
        if (shade.rgba8) __builtin_nontemporal_store(0u, &shade.rgba8[img_px + (size_t)(H - 1 - (Y0 + y)) * W + X0 + x]);
      }
      if constexpr (INTERP > 0 && !NORMS) {   // (NORMS: an empty region adds nothing; its partial row was cleared by the launcher)
        // an uncovered pixel: id 0, barycentrics 0 -> alpha 0: 0 * (triangle 0's attributes, weighted by zeros) + 1 * background
        // (rasterize.py:137-150; the products are kept: a non-finite attribute of triangle 0 shows here as in the reference)
        for (int a = 0; a < attr_n; ++a) {
          const float value = (img_attr_records[a] * 0.0f + img_attr_records[INTERP + a] * 0.0f) + img_attr_records[2 * INTERP + a] * 0.0f;
          const float o = 0.0f * value + 1.0f * shade.background[a];
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, o), rs_out, pix * (unsigned)attr_n * 4u + 4u * a, 0,
                                                0);
        }
      }
    }
    return;
  }
  // Every wavefront of this kernel is fully populated (256-thread workgroups, padding workgroups
  // leave as a whole), so EXEC is all ones in wave-uniform code: the coverage loop restores it
  // with a constant after its v_cmpx chain.
  static_assert(kThreads % kWave == 0, "full wavefronts only");
  // lanes 0..7 fetch one of the tile's eight mask words each; a ballot of "non-zero" is then the
  // list of words worth visiting (most of a tile's 256 mask bits are zero: ~3.6 candidates)
  // (lanes 8.. of a 16-wide tile fetch words of the next tiles -- the last tile's reach a few
  // dwords past the masks, never past the workgroup's LDS -- and are masked out of the ballot)
  static_assert(kMaskWords == 8 && kTileW >= 8, "lx == lane for lanes 0..7: it indexes the tile's mask words");
  const unsigned *lane_word = (const unsigned *)lane_px + 2 * R;  // &s_tmask[0][lx]
  // `full`: the region lies wholly inside the image (all but the last column / row of regions of
  // an image whose size is not a multiple of R): no per-tile or per-lane bounds tests at all.
  // Staged barycentric store (round 4).  One 12-byte store per lane puts lane boundaries inside 32-byte
  // sectors (lane 2 covers bytes 24..35, ...): the memory pipeline forwards the nontemporal store in groups
  // of lanes, a sector shared by two groups goes to memory twice as two partial writes, and WRITE_SIZE
  // showed 796 MB leaving for a 671 MB G-buffer (profiles/kernel_traffic.json, r03: +1/3 of the 403 MB
  // barycentric plane) -- in a kernel that an empty, store-only walk already takes 134 of its 148 us.
  // A tile's 64 x 3 floats are four rows of 192 contiguous bytes: each lane parks its triple at
  // 12 * lane of a 768-byte LDS slot private to its wavefront (stride 3 dwords: conflict-free) and lanes
  // with lx < 12 read 16 bytes back at 192 ly + 16 lx = 12 * lane + 4 lx and store them at the same
  // offset of the tile: 48 lanes x 16 bytes, every lane inside one sector, every sector written whole.
  // The slots are the top 3 KB of the entry bin (no LDS beyond the 18 granules the seven workgroups per
  // CU leave): `stage` is set while the bin's entries end below them (~40 of 256 used at 1024^2 / 5k
  // triangles), a crowded bin keeps the per-lane store.  Only whole-region (`full`) walks stage.
  constexpr int kStageDw = 3 * kWave;                                   // per wavefront
  constexpr int kStageEntries = (kWaves * kStageDw + kEntryDw - 1) / kEntryDw;  // bin entries the slots overlay
  // Corner records in LDS (round 4).  The epilogue used to take a tile's winning triangles one at a time: the
  // record through the scalar cache, 27 multiply-adds with a scalar operand under the winner's lanes -- ~33 vector
  // instructions per winner, ~2.7 winners per covered tile, 89 of the epilogue's ~140 per tile.  A region's bin
  // holds ~40 entries of its 256: when the region needs ONE bin round and the entries end below slot
  // kRecordSlots, the unused top of the bin holds every entry's corner record (28 dwords, 16-byte aligned, four
  // banks apart) -- loaded once per region, one thread per entry -- the depth loop remembers the winner's SLOT
  // next to its id, and the epilogue is 7 per-lane ds_read_b128 and 27 multiply-adds per tile, whatever the
  // number of winners.  LDS reads count on lgkmcnt like the scalar loads did: the G-buffer stores stay undisturbed.
  constexpr int kRecordDw = INTERP > 0 ? 3 * INTERP : 28;  // (INTERP: the attribute record of interp_fused.hip, [corner][AP])
  constexpr int kRecordSlotsRaw = kEntDw / (kEntryDw + kRecordDw);   // 106 (SHADE): entries + records fit the bin; XREC = 128: 181
  constexpr int kRecordSlots = kRecordSlotsRaw < kBin2Cap ? kRecordSlotsRaw : kBin2Cap;
  auto record_of = [&](const int slot) -> const float * { return s_ent + (kEntDw - (slot + 1) * kRecordDw); };
  auto raster_pass = [&](auto fresh_tag, auto full_tag, auto stage_tag, auto recs_tag, const int far_word,
                         const bool last_round) {
    constexpr bool fresh = decltype(fresh_tag)::value, full = decltype(full_tag)::value;
    constexpr bool stage = MR_BARY_STAGE && decltype(stage_tag)::value && full && !(PROBE & 64);
    constexpr bool lds_recs = EPI && MR_EPI_LDS_RECORDS && decltype(recs_tag)::value;
    const unsigned near_words = (1u << far_word) - 1u;  // far_word == kMaskWords: every word
    // wavefront w walks tiles w, w + 4, ... (row-major tile numbering).  (Walking pairs of
    // horizontally adjacent tiles back to back, so that both halves of a 128-byte line come from
    // one wavefront, was measured: no difference.)
    {
    for (int tile = wave; tile < kTiles; tile += kWaves) {
      const int ty = tile / kTilesX, tx = tile % kTilesX;
      const int x0 = X0 + tx * kTileW, y0 = Y0 + ty * kTileH;
      if (!full && (x0 >= X1 || y0 >= Y1)) continue;  // wave-uniform
      const bool in_image = full || (lx < W - x0 && ly < H - y0);
      const int tile_pix = ty * kTileH * W + tx * kTileW;  // wave-uniform, relative to the region
      const float px = lane_px[tx * kTileW];
      const float py = lane_py[ty * kTileH];
      const unsigned my_word = lane_word[tile * kMaskWords];
      const v2f px2 = {px, px}, py2 = {py, py};
      // region-relative pixel coordinates of this lane, packed (x | y << 16)
      const unsigned lane_xy = lane_xy0 + ((unsigned)(tx * kTileW) | ((unsigned)(ty * kTileH) << 16));
      PixelState st;
      unsigned long long pass_mask, pass_mask2;  // scratch of the depth loop's compares
      if (fresh) {
        st.z = 1.0f; st.b0 = 0.0f; st.b1 = 0.0f; st.b2 = 0.0f;  // cpp:313-321
        st.id = -1;  // "nothing drawn yet" loses every tie; stored as 0
        st.ent = ent_init;
      } else if (in_image) {
        st.ent = 0;
        st.z = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_z, lane_pix * 4u, tile_pix * 4, 0));
        st.id = (int)__builtin_amdgcn_raw_buffer_load_b32(rs_ids, lane_pix * 4u, tile_pix * 4, 0);
        st.b0 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_bary, lane_pix * 12u, tile_pix * 12, 0));
        st.b1 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_bary, lane_pix * 12u + 4u, tile_pix * 12, 0));
        st.b2 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_bary, lane_pix * 12u + 8u, tile_pix * 12, 0));
      }
      const unsigned all_words = (PROBE & 16) ? 0u  // timing probe: no coverage, no depth
                                              : (unsigned)__ballot(my_word != 0u) & ((1u << kMaskWords) - 1u);
      // near words first; the far ones only if some pixel of the tile could still be won by them
      unsigned words = all_words & near_words, far_words = all_words & ~near_words;
      for (;;) {
      while (words) {
        const int w = __builtin_ctz(words);
        words &= words - 1u;
        unsigned todo = (unsigned)__builtin_amdgcn_readlane((int)my_word, w);
        const int ebase = w * 32;
        // (1) coverage: wave-uniform entry, all lanes; one mask bit per candidate.  (Requesting
        //     the next candidate's entry before evaluating the current one was measured slower:
        //     +5 VGPRs cost the seventh wave per SIMD, 0.371 -> 0.394 ms; two candidates per trip
        //     at 70 VGPRs changed nothing: the loop does not wait on LDS latency.)
        unsigned mine = 0u;
        // Conservative coverage (round 4).  The result of this loop only SELECTS the candidates a lane then
        // evaluates exactly in (2), so a superset will do: the three edges with fused multiply-adds (4 vector
        // instructions instead of the 8 of the reference's un-fused expression), compared with the entry's
        // -tolerance instead of 0, no "some value > 0" sum, no bbox test -- 8 vector instructions, 3 LDS reads and
        // 7 scalar ones per (candidate, tile) instead of 18 + 4 + 9 -- and (2) applies the exact tests (cpp:96-97) to
        // the reference's values, which it forms anyway.  |px|, |py| < 1 and M_i = |a_i| + |b_i| + |c_i|: the
        // un-fused value differs from the true one by at most 3u M_i, the fused one by 2u M_i (u = 2^-24), so an
        // edge whose reference value is >= 0 has a fused value >= -5u M_i > -2^-20 max(M); underflowing products
        // add less than 2^-146.  An entry whose coefficients could overflow (max M > 2^100, or a NaN) carries
        // -inf (or NaN) as its tolerance, and "not less than" lets every pixel -- and a NaN minimum -- through.
        // (Stage probes before: walk without this loop 31 us, with it 113, with the depth loop 138.)
        do {
          const int j = __builtin_ctz(todo);
          asm("s_bitset0_b32 %0, %1" : "+s"(todo) : "s"(j));   // todo &= todo - 1 in one scalar instruction
          const float *p = s_ent + (ebase + j) * kEntryDw;  // wave-uniform address
          const float4 q0 = *(const float4 *)(p), q1 = *(const float4 *)(p + 4);
          const float2 ct = *(const float2 *)(p + 8);   // c2, -tolerance
          const v2f e01 = __builtin_elementwise_fma(v2f{q0.x, q0.y}, px2,
                                                    __builtin_elementwise_fma(v2f{q0.z, q0.w}, py2, v2f{q1.x, q1.y}));
          const float e2 = __builtin_fmaf(q1.z, px, __builtin_fmaf(q1.w, py, ct.x));
          float emin;
          asm volatile(
              "v_min3_f32 %[emin], %[e0], %[e1], %[e2]\n\t"
              "v_cmpx_nlt_f32_e32 vcc, %[emin], %[ntol]\n\t"
              "v_lshl_or_b32 %[mine], 1, %[j], %[mine]\n\t"
              "s_mov_b64 exec, -1"
              : [mine] "+v"(mine), [emin] "=&v"(emin)
              : [e0] "v"(e01.x), [e1] "v"(e01.y), [e2] "v"(e2), [ntol] "v"(ct.y), [j] "s"(j)
              : "vcc");
        } while (todo);
        // (2) depth: every lane walks its own candidates in ascending id
        if (PROBE & 8) { st.id += (int)mine; continue; }  // timing probe: coverage only
        // The trip is straight-line code for the whole wavefront -- lanes that have run out of
        // candidates ride along on a valid dummy entry and are masked out of the final select --
        // because vector instructions cost the same whatever EXEC holds, and a divergent
        // `if (mine)` made the compiler keep two copies of the pixel state (10 moves per trip).
        for (;;) {
          const unsigned long long has = __ballot(mine != 0u);   // lanes with a candidate left
          if (!has) break;
          const unsigned bit = min((unsigned)(__ffs((int)mine) - 1), 31u);  // no candidate: slot 31
          mine &= mine - 1u;
          const int slot = ebase + (int)bit;
          const Entry t = read_entry(s_ent, slot);  // per-lane LDS address
          const v2f e01 = (v2f{t.q0.x, t.q0.y} * px2 + v2f{t.q0.z, t.q0.w} * py2) +
                          v2f{t.q1.x, t.q1.y};                         // same bits as in (1)
          float e0 = e01.x, e1 = e01.y;
          float e2 = (t.q1.z * px + t.q1.w * py) + t.q2.x;
          asm("" : "+v"(e0), "+v"(e1), "+v"(e2));  // scalar from here on: no re-packing moves
          const float s = (e0 + e1) + e2;                              // cpp:384
          // cpp:96-97 exactly, on the reference's own values (see (1): its conservative test lets a few pixels
          // next to an edge through)
          const unsigned dxy = pk_sub_u16(lane_xy, t.tail.x);   // inside the bbox (cpp:96)
          const unsigned long long valid = has & __builtin_amdgcn_ballot_w64(__builtin_fminf(__builtin_fminf(e0, e1), e2) >= 0.0f) &
                                           __builtin_amdgcn_ballot_w64(s > 0.0f) &
                                           __builtin_amdgcn_ballot_w64(pk_min_u16(dxy, t.tail.y) == dxy);
          float b0, b1, b2;
          div3_common_denominator(e0, e1, e2, s, valid, b0, b1, b2);   // cpp:385-387
          // cpp:395-396: cz = (b0 z0 + b1 z1) + b2 z2 and cw likewise, as (z, w) pairs
          const v2f zw = (v2f{t.q2.z, t.q2.w} * v2f{b0, b0} + v2f{t.q3.x, t.q3.y} * v2f{b1, b1}) +
                         v2f{t.q3.z, t.q3.w} * v2f{b2, b2};
          const float cz = zw.x, cw = zw.y;
          const float zz = cz / cw;                                    // cpp:397
          // cpp:401: the candidate loses if zz < -1 || zz > 1 || zz > zbuf; a NaN passes.  The
          // winner's five values replace the pixel state in place.  With front-to-back classes
          // the candidates do not arrive in id order, so the tie the sequential loop resolves
          // implicitly (equal depth: the later id overwrites) is tested explicitly; depths are
          // finite in that mode.
          // ONE test for both modes (round 4; a run-time `if (ordered)` around two variants made the compiler keep
          // two copies of the pixel state: ten moves and two branches per trip).  The candidate takes the pixel
          // unless zz > z, or zz == z and the pixel already holds a LATER triangle: in id order (one class) the
          // second clause never fires and a NaN passes, as in the reference; with front-to-back classes depths are
          // finite and the clause resolves the tie the sequential loop resolves implicitly.  ("Nothing drawn yet"
          // is id -1 in both modes, stored as 0.)
          asm volatile(
              "v_cmp_ngt_f32_e64 vcc, |%[zz]|, 1.0\n\t"
              "v_cmp_ngt_f32_e64 %[m], %[zz], %[z]\n\t"
              "s_and_b64 %[valid2], vcc, %[valid]\n\t"
              "s_and_b64 %[valid2], %[valid2], %[m]\n\t"
              "v_cmp_eq_f32_e64 vcc, %[zz], %[z]\n\t"
              "v_cmp_gt_i32_e64 %[m], %[id], %[tid]\n\t"
              "s_and_b64 vcc, vcc, %[m]\n\t"
              "s_andn2_b64 vcc, %[valid2], vcc\n\t"
              "v_cndmask_b32_e32 %[z], %[z], %[zz], vcc\n\t"
              "v_cndmask_b32_e32 %[id], %[id], %[tid], vcc\n\t"
              "v_cndmask_b32_e32 %[c0], %[c0], %[b0], vcc\n\t"
              "v_cndmask_b32_e32 %[c1], %[c1], %[b1], vcc\n\t"
              "v_cndmask_b32_e32 %[c2], %[c2], %[b2], vcc\n\t"
              "s_mov_b64 %[m], vcc"
              : [z] "+v"(st.z), [id] "+v"(st.id), [c0] "+v"(st.b0), [c1] "+v"(st.b1), [c2] "+v"(st.b2),
                [m] "=&s"(pass_mask), [valid2] "=&s"(pass_mask2)
              : [zz] "v"(zz), [tid] "v"(t.tail.z), [b0] "v"(b0), [b1] "v"(b1), [b2] "v"(b2), [valid] "s"(valid)
              : "vcc");
          if constexpr (lds_recs) {  // the winner's slot rides along with its id (pass_mask: the lanes that took this candidate)
            asm("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(st.ent) : "v"(slot), "s"(pass_mask));
          }
        }
      }
      if (far_words == 0u || !__ballot(in_image && !(st.z < split))) break;
      words = far_words;
      far_words = 0u;
      }
      if constexpr (NORMS) {
        if (last_round) {  // workgroup-uniform
          // a covered pixel (alpha = clamp(2 sum b) = 1: the barycentrics this kernel writes sum to 1): normal and
          // position interpolated from the winner's record, then every light's reflection . camera dot product
          // (spec_pixel.h, as shade_spec.hip's norm pass evaluates it); uncovered pixels are counted, see k_spec_norm_finish
          const float pre = (2.0f * st.b0 + 2.0f * st.b1) + 2.0f * st.b2;
          const bool live = in_image && pre > 0.0f;
          const float *rec;
          if constexpr (lds_recs) rec = record_of(st.ent);
          else rec = img_attr_records + (size_t)min((unsigned)max(st.id, 0), (unsigned)(T - 1)) * (3 * INTERP);
          const float4 c0a = *(const float4 *)(rec), c0b = *(const float4 *)(rec + 4);
          const float4 c1a = *(const float4 *)(rec + INTERP), c1b = *(const float4 *)(rec + INTERP + 4);
          const float4 c2a = *(const float4 *)(rec + 2 * INTERP), c2b = *(const float4 *)(rec + 2 * INTERP + 4);
          float at[6];
          {
#pragma clang fp contract(fast)
            const float alpha = fminf(fmaxf(pre, 0.0f), 1.0f), one_m = 1.0f - alpha;
            const float k0[6] = {c0a.x, c0a.y, c0a.z, c0a.w, c0b.x, c0b.y}, k1[6] = {c1a.x, c1a.y, c1a.z, c1a.w, c1b.x, c1b.y},
                        k2[6] = {c2a.x, c2a.y, c2a.z, c2a.w, c2b.x, c2b.y};
#pragma unroll
            for (int a = 0; a < 6; ++a) at[a] = alpha * ((k0[a] * st.b0 + k1[a] * st.b1) + k2[a] * st.b2) - one_m;
          }
          const float cam[3] = {norm_cam[0], norm_cam[1], norm_cam[2]};
          spec::PixelFrame f;
          spec::pixel_frame(at, cam, f);
#pragma unroll
          for (int l = 0; l < 4; ++l) {
            if (l < norm_lights) {   // wave-uniform
              const float lp[3] = {norm_lp[l][0], norm_lp[l][1], norm_lp[l][2]};
              spec::LightTerm lt;
              spec::light_term(at, f, lp, lt);
              norm_acc[l] += live ? lt.rdc * lt.rdc : 0.0f;
            }
          }
          norm_covered += live ? 1.0f : 0.0f;
        }
      } else if constexpr (INTERP > 0) {
        if (last_round) {  // workgroup-uniform
          constexpr int AP = INTERP;
          typedef unsigned v4u __attribute__((ext_vector_type(4)));
          typedef unsigned v2u __attribute__((ext_vector_type(2)));
          // the record: [corner][AP] floats, from LDS by the winner's slot, or -- a crowded region -- per lane from
          // memory by triangle id (no winner: triangle 0, as the reference)
          const float *rec;
          if constexpr (lds_recs) rec = record_of(st.ent);
          else rec = img_attr_records + (size_t)min((unsigned)max(st.id, 0), (unsigned)(T - 1)) * (3 * AP);
          const float pre = (2.0f * st.b0 + 2.0f * st.b1) + 2.0f * st.b2;
          const float alpha = fminf(fmaxf(pre, 0.0f), 1.0f);
          const float one_m = 1.0f - alpha;
          const int tile_out = tile_pix * attr_n * 4;
          constexpr bool staged = full && MR_RASTER_INTERP_STAGE;
          float *mine = stage_slot + lane * attr_n;
          // four attributes at a time: three 16-byte reads (one per corner), four results, straight to their
          // destination -- the whole record at once was 36 + 12 live registers at A = 9
#pragma unroll
          for (int q = 0; q < AP / 4; ++q) {
            if (4 * q >= attr_n) break;   // wave-uniform
            const float4 c0 = *(const float4 *)(rec + 4 * q), c1 = *(const float4 *)(rec + AP + 4 * q),
                         c2 = *(const float4 *)(rec + 2 * AP + 4 * q);
            const float4 bg4 = *(const float4 *)(s_background + 4 * q);   // (zeros beyond A)
            float o[4];
            {
#pragma clang fp contract(fast)
              const float k0[4] = {c0.x, c0.y, c0.z, c0.w}, k1[4] = {c1.x, c1.y, c1.z, c1.w}, k2[4] = {c2.x, c2.y, c2.z, c2.w};
              const float bg[4] = {bg4.x, bg4.y, bg4.z, bg4.w};
#pragma unroll
              for (int a = 0; a < 4; ++a) {
                const float value = (k0[a] * st.b0 + k1[a] * st.b1) + k2[a] * st.b2;
                o[a] = alpha * value + one_m * bg[a];
              }
            }
            const int left = attr_n - 4 * q;   // wave-uniform, >= 1
            if constexpr (staged) {
              if ((attr_n & 3) == 0) *(float4 *)(mine + 4 * q) = make_float4(o[0], o[1], o[2], o[3]);
              else if ((attr_n & 1) == 0) {
                *(float2 *)(mine + 4 * q) = make_float2(o[0], o[1]);
                if (left >= 4) *(float2 *)(mine + 4 * q + 2) = make_float2(o[2], o[3]);
              } else {
#pragma unroll
                for (int a = 0; a < 4; ++a)
                  if (a < left) mine[4 * q + a] = o[a];
              }
            } else if (in_image) {
              const unsigned u0 = __builtin_bit_cast(unsigned, o[0]), u1 = __builtin_bit_cast(unsigned, o[1]),
                             u2 = __builtin_bit_cast(unsigned, o[2]), u3 = __builtin_bit_cast(unsigned, o[3]);
              if (left >= 4) store_b128_soffset<MR_RASTER_INTERP_STORE_AUX>(v4u{u0, u1, u2, u3}, rs_out, lane_out + 16u * q, tile_out);
              else if (left == 3) store_b96_soffset<MR_RASTER_INTERP_STORE_AUX>(v3u{u0, u1, u2}, rs_out, lane_out + 16u * q, tile_out);
              else if (left == 2) __builtin_amdgcn_raw_buffer_store_b64(v2u{u0, u1}, rs_out, lane_out + 16u * q, tile_out, MR_RASTER_INTERP_STORE_AUX);
              else __builtin_amdgcn_raw_buffer_store_b32(u0, rs_out, lane_out + 16u * q, tile_out, MR_RASTER_INTERP_STORE_AUX);
            }
          }
          if constexpr (staged) {
            // (same wavefront: LDS operations complete in order) the tile back as 16-byte chunks, lane + 64 i
            const unsigned n_chunks = 16u * (unsigned)attr_n;   // 64 pixels x A floats / 4
#pragma unroll
            for (int i = 0; i < AP / 4; ++i) {
              const unsigned jc = (unsigned)lane + 64u * i;
              if (64u * i < n_chunks) {  // wave-uniform
                const float4 v = *(const float4 *)(stage_slot + 4u * jc);   // (beyond n_chunks: inside the slot, not stored)
                if (jc < n_chunks)
                  store_b128_soffset<MR_RASTER_STORE_AUX>(v4u{__builtin_bit_cast(unsigned, v.x), __builtin_bit_cast(unsigned, v.y),
                                                              __builtin_bit_cast(unsigned, v.z), __builtin_bit_cast(unsigned, v.w)},
                                                          rs_out, stage_goff[i], tile_out);
              }
            }
          }
        }
      }
      if (SHADE && last_round) {  // workgroup-uniform
        // the rule of k_shade_forward: a pixel is shaded iff alpha = clamp(2 sum(bary)) > 0
        const bool live = in_image && ((2.0f * st.b0 + 2.0f * st.b1) + 2.0f * st.b2) > 0.0f;
        float4 rgba = make_float4(0.f, 0.f, 0.f, 0.f);
        // The corner attributes come through the SCALAR cache, one winning triangle of the tile at
        // a time (a 64-pixel tile shows ~3 of them): scalar loads count on lgkmcnt, so their wait
        // leaves the G-buffer stores of the previous tile alone -- vector loads count on vmcnt
        // with the stores, in order, and cost +0.1 ms (measured: per-lane gather 0.420 ms, one
        // shared record 0.377, no loads 0.321).
        float interp[9];
#pragma unroll
        for (int a = 0; a < 9; ++a) interp[a] = 0.0f;
        if constexpr (lds_recs) {
          // (lanes without a winner hold slot 0 and read a record they do not use)
          const float4 *rec = (const float4 *)record_of(st.ent);
          float c[28];
#pragma unroll
          for (int q = 0; q < 7; ++q) {
            const float4 f = rec[q];
            c[4 * q] = f.x; c[4 * q + 1] = f.y; c[4 * q + 2] = f.z; c[4 * q + 3] = f.w;
          }
          const float bk[3] = {st.b0, st.b1, st.b2};
          {
#pragma clang fp contract(fast)
            if constexpr (MR_EPI_DIFF_BASIS) {
#pragma unroll
              for (int a = 0; a < 9; ++a) interp[a] = c[a] * bk[0] + (c[9 + a] * bk[1] + c[18 + a]);
            } else {
#pragma unroll
              for (int k = 0; k < 3; ++k)
#pragma unroll
                for (int a = 0; a < 9; ++a) interp[a] = c[k * 9 + a] * bk[k] + interp[a];
            }
          }
        }
        unsigned long long todo = lds_recs ? 0ull : __ballot(live);
        while (todo) {  // wave-uniform
          const int src = __builtin_ctzll(todo);
          const int t = __builtin_amdgcn_readlane(st.id, src);
          const bool mine = live && st.id == t;
          const unsigned tc = min((unsigned)max(t, 0), (unsigned)(T - 1));
          ConstFloats rec = (ConstFloats)(uintptr_t)(img_corners + tc);
          todo &= ~__ballot(mine);
          const float bk[3] = {st.b0, st.b1, st.b2};
          // the whole 128-byte record with one wait (explicit: left to itself the compiler turns a
          // 27-dword uniform read into per-lane vector loads)
          typedef float v16f __attribute__((ext_vector_type(16)));
          typedef float v8f __attribute__((ext_vector_type(8)));
          typedef float v4f __attribute__((ext_vector_type(4)));
          v16f r0;
          v8f r1;
          v4f r2;
          // (one asm block from the first load to the wait: the compiler does not track loads it did
          // not issue and must not be given a chance to reuse their destination registers early)
          asm volatile(
              "s_load_dwordx16 %0, %3, 0x0\n\t"
              "s_load_dwordx8 %1, %3, 0x40\n\t"
              "s_load_dwordx4 %2, %3, 0x60\n\t"
              "s_waitcnt lgkmcnt(0)"
              : "=&s"(r0), "=&s"(r1), "=&s"(r2)
              : "s"(rec));
          float c[28];
#pragma unroll
          for (int a = 0; a < 16; ++a) c[a] = r0[a];
#pragma unroll
          for (int a = 0; a < 8; ++a) c[16 + a] = r1[a];
#pragma unroll
          for (int a = 0; a < 4; ++a) c[24 + a] = r2[a];
          if (mine) {
#pragma clang fp contract(fast)
#pragma unroll
            for (int k = 0; k < 3; ++k)
#pragma unroll
              for (int a = 0; a < 9; ++a) interp[a] = c[k * 9 + a] * bk[k] + interp[a];
          }
        }
        // interpolate9's blend with the -1 background is the identity here: these barycentrics were
        // just normalised by this kernel (their sum is 1 up to three roundings), so alpha = clamp(2 sum
        // b, 0, 1) is exactly 1 and 1 * x + 0 * (-1) = x -- the same bits k_shade_forward produces
        if (live) rgba = shade_attributes(interp, lights);
        if (in_image) {
          typedef float v4f __attribute__((ext_vector_type(4)));
          typedef unsigned v4u __attribute__((ext_vector_type(4)));
          const unsigned lane_rgba = (unsigned)((kTileH - 1 - ly) * W + lx) * 16u;
          const int tile_rgba = ((R - kTileH - ty * kTileH) * W + tx * kTileW) * 16;
          store_b128_soffset<MR_RASTER_STORE_AUX_RGBA>(__builtin_bit_cast(v4u, v4f{rgba.x, rgba.y, rgba.z, rgba.w}), rs_rgba,
                                                       lane_rgba, tile_rgba);
          if (shade.rgba8) {  // workgroup-uniform: the 8-bit frame for the multi-GPU hand-over, 4 B/px
            auto u8 = [](float v) { return (unsigned)(fminf(fmaxf(v, 0.0f), 1.0f) * 255.0f); };  // NaN -> 0
            const unsigned packed = u8(rgba.x) | (u8(rgba.y) << 8) | (u8(rgba.z) << 16) | (u8(rgba.w) << 24);
            const int y = Y0 + ty * kTileH + ly, x = X0 + tx * kTileW + lx;
            // (through the caches, like the id plane -- MR_RASTER_STORE_AUX_IDS: a tile row is a 64-byte run, half a line,
            //  and the neighbouring tile's half arrives from another wavefront; nontemporal, the halves reached memory apart)
            if (MR_RASTER_RGBA8_NT) __builtin_nontemporal_store(packed, &shade.rgba8[img_px + (size_t)(H - 1 - y) * W + x]);
            else shade.rgba8[img_px + (size_t)(H - 1 - y) * W + x] = packed;
          }
        }
      }
      if (in_image && !((PROBE & 32) && st.z != 123.0f)) {  // 32: timing probe, no stores
        // 64: timing probe (R = 64 only) -- the same bytes, but tile (tx, ty) writes ROW ty * 4 + tx of the
        // region, lane = x: every store instruction covers one contiguous run (256 B of ids / depths,
        // 768 B of barycentrics) instead of four 64- / 192-byte runs.  The image comes out scrambled.
        const unsigned st_lane = ((PROBE & 64) && R == 64) ? (unsigned)lane : lane_pix;
        const int st_tile = ((PROBE & 64) && R == 64) ? (ty * kTileH + tx) * W : tile_pix;
        __builtin_amdgcn_raw_buffer_store_b32((unsigned)max(st.id, 0), rs_ids, st_lane * 4u, st_tile * 4,
                                              (PROBE & 64) ? MR_RASTER_STORE_AUX : MR_RASTER_STORE_AUX_IDS);
        if (!EPI || !last_round || shade.keep_z)  // workgroup-uniform
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, st.z), rs_z, st_lane * 4u, st_tile * 4, MR_RASTER_STORE_AUX_Z);
        if constexpr (stage) {
          typedef float v4f __attribute__((ext_vector_type(4)));
          typedef unsigned v4u __attribute__((ext_vector_type(4)));
          float *slot = s_ent + (kBin2Cap * kEntryDw - kWaves * kStageDw) + wave * kStageDw + 3 * lane;
          slot[0] = st.b0; slot[1] = st.b1; slot[2] = st.b2;
          // (same wavefront, LDS operations complete in order: no barrier between the two)
          const v4f run = *(const v4f *)(slot + lx);
          if (lx < kTileW * 3 / 4)
            store_b128_soffset<MR_RASTER_STORE_AUX>(__builtin_bit_cast(v4u, run), rs_bary, lane_pix * 12u + (unsigned)lx * 4u,
                                                    tile_pix * 12);
        } else {
          store_b96_soffset<MR_RASTER_STORE_AUX>(__builtin_bit_cast(v3u, v3f{st.b0, st.b1, st.b2}), rs_bary, st_lane * 12u,
                                                 st_tile * 12);
        }
      }
      // Later rounds re-LOAD the pixel state.  vmcnt is in order on gfx950, so the compiler's wait
      // for such a load also drains the G-buffer stores behind it; draining explicitly here -- on
      // this rare path only -- tells its dataflow that no load is pending where the two variants
      // of the walk merge, which keeps the first round's loop free of vmcnt waits (its stores
      // then retire under the next tiles' arithmetic).
      if (!fresh) __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
    }
    }
  };

  // ---- stage 1: rounds of (scan a slice of the list -> bin -> walk the tiles) -----------
  // A round covers up to kRoundChunks 64-triangle chunks.  Wavefront w takes chunks
  // w, w+4, w+8, ... so that survivors -- which cluster in id space for any mesh with
  // spatially coherent numbering -- spread evenly over the four 64-entry sub-bins; the
  // id order is restored at compaction from per-chunk survivor counts.
  bool first_pass = true;
  int round_base = 0;  // first list position of the round, workgroup-uniform, multiple of 64
  do {
    const int chunks_left = (n_cand - round_base + kWave - 1) / kWave;
    const int round_chunks = min(chunks_left, kRoundChunks);
    s_chunk_count[tid] = 0;  // kRoundChunks == kThreads
    __syncthreads();
    int count = 0, stop = round_chunks;  // stop: first chunk this wavefront could not take
    // One chunk per trip, and a candidate's RECORD is requested together with its box (round 3): a
    // region's list holds ~80 ids -- one chunk for each of two wavefronts -- and nearly all of them pass
    // the box test (the list was made by it, at region granularity), so the stage was a chain of three
    // dependent L2 round trips (id -> box -> record); now two.
    constexpr int kUnroll = 1;
    for (int c0 = wave; c0 < round_chunks && stop == round_chunks; c0 += kUnroll * kWaves) {
      TriBox bb[kUnroll];
      int tri[kUnroll];
      bool near[kUnroll];
      float4 rec_a[kUnroll], rec_b[kUnroll], rec_c[kUnroll], rec_d[kUnroll];
#pragma unroll
      for (int u = 0; u < kUnroll; ++u) {
        const int c = c0 + u * kWaves;
        const int k = round_base + c * kWave + lane;
        tri[u] = (c < round_chunks && k < n_cand) ? cand[k] : -1;
      }
#pragma unroll
      for (int u = 0; u < kUnroll; ++u) {
        const int t = max(tri[u], 0);   // (lanes without a candidate read triangle 0 and ignore it)
        bb[u] = img_bbs[t];
        const TriRec *rp = img_recs + t;
        rec_a[u] = rp->a; rec_b[u] = rp->b; rec_c[u] = rp->c; rec_d[u] = rp->d;
        if (tri[u] < 0) bb[u] = TriBox{0u, 0u, 0.0f, 0u};
      }
#pragma unroll
      for (int u = 0; u < kUnroll; ++u) {
        const int l = (int)(bb[u].lr & 0xffffu), r = (int)(bb[u].lr >> 16);
        const int bt = (int)(bb[u].bt & 0xffffu), tp = (int)(bb[u].bt >> 16);
        near[u] = (l < X1) && (r > X0) && (bt < Y1) && (tp > Y0);  // empty bbox = all zeros
      }
#pragma unroll
      for (int u = 0; u < kUnroll; ++u) {
        const int c = c0 + u * kWaves;
        if (c >= round_chunks || stop != round_chunks) continue;  // wave-uniform
        const int t = tri[u];
        bool pass = false;
        if (near[u]) pass = !rect_outside_triangle(rec_a[u], rec_b[u], rec_c[u].x, rpxlo, rpxhi, rpylo, rpyhi);
        const unsigned long long m = __ballot(pass);
        const int cnt = __builtin_popcountll(m);
        if (count + cnt > kSubCap) {  // this wavefront's sub-bin is full
          stop = c;
          continue;
        }
        // depth class of the survivor (0 = near, 1 = far) and its rank inside (chunk, class)
        const bool far = pass && !(bb[u].zlo < split);
        const unsigned long long m_far = __ballot(far);
        const int cnt_far = __builtin_popcountll(m_far);
        if (pass) {
          const int rank = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32),
                                                          __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
          const unsigned long long m_class = far ? m_far : (m & ~m_far);
          const unsigned class_rank = __builtin_amdgcn_mbcnt_hi((unsigned)(m_class >> 32),
                                                                __builtin_amdgcn_mbcnt_lo((unsigned)m_class, 0u)) |
                                      (far ? 0x80000000u : 0u);
          float *p = s_ent + (wave * kSubCap + count + rank) * kEntryDw;
          const float4 q2 = rec_c[u];
          *(float4 *)(p) = rec_a[u];
          *(float4 *)(p + 4) = rec_b[u];
          *(float4 *)(p + 8) = make_float4(q2.x, __builtin_bit_cast(float, t), q2.z, q2.w);
          *(float4 *)(p + 12) = rec_d[u];
          const int l = (int)(bb[u].lr & 0xffffu), r = (int)(bb[u].lr >> 16);
          const int bt = (int)(bb[u].bt & 0xffffu), tp = (int)(bb[u].bt >> 16);
          // bbox clipped to the region, region-relative: (l | b << 16), (w - 1 | h - 1 << 16)
          const int cl = max(l, X0), cb = max(bt, Y0);
          const unsigned rel = (unsigned)(cl - X0) | ((unsigned)(cb - Y0) << 16);
          const unsigned ext = (unsigned)(min(r, X1) - cl - 1) | ((unsigned)(min(tp, Y1) - cb - 1) << 16);
          *(uint4 *)(p + 16) = make_uint4(rel, ext, (unsigned)c, class_rank);
        }
        if (lane == 0) s_chunk_count[c] = (cnt - cnt_far) | (cnt_far << 16);  // near | far << 16
        count += cnt;
      }
    }
    if (lane == 0) {
      s_count[wave] = count;
      s_stop[wave] = stop;
    }
    __syncthreads();
    // chunks before the first one any wavefront had to refuse are complete
    int keep_chunks = round_chunks;
#pragma unroll
    for (int w = 0; w < kWaves; ++w) keep_chunks = min(keep_chunks, s_stop[w]);
    // exclusive prefix sums of the kept chunks' survivor counts, both classes at once in the
    // two halves of one register (thread t <-> chunk t; totals never exceed kBin2Cap)
    int n_near, n_far, far_base;
    {
      const int v = (tid < keep_chunks) ? s_chunk_count[tid] : 0;
      int incl = v;
#pragma unroll
      for (int off = 1; off < kWave; off <<= 1) {
        const int up = __shfl_up(incl, off);
        if (lane >= off) incl += up;
      }
      if (lane == kWave - 1) s_wave_total[wave] = incl;
      __syncthreads();
      int wave_off = 0, total = 0;
#pragma unroll
      for (int w = 0; w < kWaves; ++w) {
        const int tot = s_wave_total[w];
        if (w < wave) wave_off += tot;
        total += tot;
      }
      s_chunk_count[tid] = wave_off + incl - v;  // now: offsets of chunk tid inside the two classes
      n_near = total & 0xffff;
      n_far = total >> 16;
      // the far class starts on a mask-word boundary when the bin has room for the gap, so that
      // "first far word" is a clean cut for the tile walk
      far_base = (n_near + 31) & ~31;
      if (far_base + n_far > kBin2Cap) far_base = n_near;
    }
    // move every kept entry to its slot: (class, id) order (in place, via registers)
    Entry mine;
    unsigned my_chunk = 0u, my_rank = 0u;
    const int my_slot = (tid >> 6) * kSubCap + (tid & (kWave - 1));  // sub-bin of wave tid/64
    const bool have = (tid & (kWave - 1)) < s_count[tid >> 6];
    if (have) {
      mine = read_entry(s_ent, my_slot);
      const uint2 tag = *(const uint2 *)(s_ent + my_slot * kEntryDw + 18);
      my_chunk = tag.x;
      my_rank = tag.y;
    }
    __syncthreads();
    if (have && (int)my_chunk < keep_chunks) {
      const int offs = s_chunk_count[my_chunk];
      const int slot = (my_rank & 0x80000000u) ? far_base + (offs >> 16) + (int)(my_rank & 0xffffu)
                                               : (offs & 0xffff) + (int)my_rank;
      float *p = s_ent + slot * kEntryDw;
      *(float4 *)(p) = mine.q0;
      *(float4 *)(p + 4) = mine.q1;
      // -tolerance of the walk's conservative coverage test (see there): 2^-20 (|a_i| + |b_i| + |c_i|), the largest of the three edges
      const float m0 = (fabsf(mine.q0.x) + fabsf(mine.q0.z)) + fabsf(mine.q1.x);
      const float m1 = (fabsf(mine.q0.y) + fabsf(mine.q0.w)) + fabsf(mine.q1.y);
      const float m2 = (fabsf(mine.q1.z) + fabsf(mine.q1.w)) + fabsf(mine.q2.x);
      const float mmax = fmaxf(fmaxf(m0, m1), m2);   // (fmaxf drops a NaN operand: the three sums are tested one by one)
      // coefficients that could overflow, or a NaN in ANY of the three edges (every comparison with it is false):
      // the test lets every pixel through and the exact test of the depth trip decides
      const bool tame = (m0 <= 0x1p100f) & (m1 <= 0x1p100f) & (m2 <= 0x1p100f);
      const float ntol = tame ? -(0x1p-20f * mmax + 0x1p-140f) : -INFINITY;
      *(float4 *)(p + 8) = make_float4(mine.q2.x, ntol, mine.q2.z, mine.q2.w);
      *(float4 *)(p + 12) = mine.q3;
      *(uint4 *)(p + 16) = make_uint4(mine.tail.x, mine.tail.y, __builtin_bit_cast(unsigned, mine.q2.y), 0u);
    }
    const int next_base = round_base + keep_chunks * kWave;
    __syncthreads();
    // mask word that starts the far class (kMaskWords: no clean cut, the walk never skips)
    const int far_word = (ordered && n_far > 0 && (far_base & 31) == 0) ? far_base >> 5 : kMaskWords;
    const bool last_round = next_base >= n_cand;
    // the staging slots overlay the last kStageEntries entries of the bin (see "staged barycentric store")
    const bool stage = max(n_near, far_base + n_far) <= kBin2Cap - kStageEntries;
    // the epilogue's corner records fit the unused top of the bin (see "corner records in LDS"): ONE bin round
    // (the slots of an earlier round's winners would be gone) with every entry below slot kRecordSlots
    // (INTERP keeps one slot more: the record a pixel WITHOUT a winner looks up -- triangle 0's, as in the reference)
    const int top_slot = max(n_near, far_base + n_far);
    const bool recs = EPI && MR_EPI_LDS_RECORDS && first_pass && last_round && top_slot + (INTERP > 0 ? 1 : 0) <= kRecordSlots;
    if (recs) {  // workgroup-uniform; build_tile_masks' barriers order these stores before the walk
      const int total = n_near + n_far;
      if (INTERP > 0) ent_init = top_slot;
      if (tid < total + (INTERP > 0 ? 1 : 0)) {
        const int slot = tid < n_near ? tid : tid < total ? far_base + (tid - n_near) : top_slot;
        const int t = tid < total ? __builtin_bit_cast(int, s_ent[slot * kEntryDw + 18]) : 0;
        const unsigned tc = min((unsigned)max(t, 0), (unsigned)(T - 1));
        const float4 *src = SHADE ? (const float4 *)(img_corners + tc) : (const float4 *)(img_attr_records + (size_t)tc * kRecordDw);
        float4 *dst = (float4 *)(s_ent + (kEntDw - (slot + 1) * kRecordDw));
        float4 q[kRecordDw / 4];
#pragma unroll
        for (int i = 0; i < kRecordDw / 4; ++i) q[i] = src[i];
        if constexpr (SHADE && MR_EPI_DIFF_BASIS) {
          // the shading epilogue's record in the DIFFERENCE basis (as the backward's FoldRec): e0 = c0 - c2,
          // e1 = c1 - c2, c2 -- the barycentrics this kernel writes sum to 1 up to three roundings, so a pixel
          // interpolates with c2 + b0 e0 + b1 e1: 18 multiply-adds per tile instead of 27
          float v[28];
#pragma unroll
          for (int i = 0; i < 7; ++i) { v[4 * i] = q[i].x; v[4 * i + 1] = q[i].y; v[4 * i + 2] = q[i].z; v[4 * i + 3] = q[i].w; }
#pragma unroll
          for (int a = 0; a < 9; ++a) { v[a] -= v[18 + a]; v[9 + a] -= v[18 + a]; }
#pragma unroll
          for (int i = 0; i < 7; ++i) q[i] = make_float4(v[4 * i], v[4 * i + 1], v[4 * i + 2], v[4 * i + 3]);
        }
#pragma unroll
        for (int i = 0; i < kRecordDw / 4; ++i) dst[i] = q[i];
      }
    }
    auto walk = [&]() {
      constexpr std::false_type no{};
      constexpr std::true_type yes{};
      if (!first_pass) raster_pass(no, no, no, no, far_word, last_round);
      else if (X1 - X0 == R && Y1 - Y0 == R) {
        if (MR_BARY_STAGE && stage) raster_pass(yes, yes, yes, no, far_word, last_round);
        else if (recs) raster_pass(yes, yes, no, yes, far_word, last_round);
        else raster_pass(yes, yes, no, no, far_word, last_round);
      } else if (recs) raster_pass(yes, no, no, yes, far_word, last_round);
      else raster_pass(yes, no, no, no, far_word, last_round);
    };
    if constexpr (PROBE == 0 || PROBE >= 8) {
      build_tile_masks(n_near, far_base, n_far);
      walk();
    } else if constexpr (PROBE == 2) {
      build_tile_masks(0, 0, 0);   // timing probe: tile walk over an empty bin
      walk();
    } else if constexpr (PROBE == 3) {
      build_tile_masks(n_near, far_base, n_far);  // timing probe: bin + tile masks, no walk
    }
    first_pass = false;
    round_base = next_base;
    if (round_base < n_cand) __syncthreads();  // tiles done with the bin; orders the state stores
  } while (round_base < n_cand);
  if constexpr (NORMS) {   // the region's row of partial sums: fixed tree per wavefront, fixed order over the wavefronts
    __shared__ float s_norm[kWaves][5];
    float v[5] = {norm_acc[0], norm_acc[1], norm_acc[2], norm_acc[3], norm_covered};
#pragma unroll
    for (int k = 0; k < 5; ++k) {
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) v[k] += __shfl_down(v[k], off);
      if (lane == 0) s_norm[wave][k] = v[k];
    }
    __syncthreads();
    if (tid < 5) {
      float t = 0.0f;
      for (int w = 0; w < kWaves; ++w) t += s_norm[w][tid];
      shade.norm_partials[(size_t)region * 8 + tid] = t;
    }
  }
}

}  // namespace

// Region edge for a launch: 64 pixels, or 32 when 64-pixel regions would leave the chip short of
// workgroups (fewer than four per CU).  Pure function of the dimensions: the workspace query and
// the launcher must agree on the cell grid.
thread_local int g_raster_region_edge = 0;  // 0: automatic; 32 / 64: forced (mr_debug_set_raster_region_edge)
thread_local int g_raster_repeat = 1;       // k_raster launches per call (mr_debug_set_raster_repeat: measurement only)

static int region_edge(int B, int W, int H) {
  if (g_raster_region_edge != 0) return g_raster_region_edge;
  const long regions64 = (long)B * ((W + 63) / 64) * ((H + 63) / 64);
  return regions64 < 4L * 256 ? 32 : 64;
}

// super-cell lists + per-chunk counts of k_coarse_top (0 when the level is not used)
static bool coarse_top_used(int T, int W, int H, int cell) {
  const int sc = kSuperCells * cell;
  return MR_COARSE_TOP && T >= kTopMinTriangles && ((W + sc - 1) / sc) * ((H + sc - 1) / sc) > 1;
}
static size_t coarse_top_bytes(int B, int T, int W, int H, int cell) {
  if (!coarse_top_used(T, W, H, cell)) return 0;
  const int sc = kSuperCells * cell;
  const size_t n_sc = (size_t)((W + sc - 1) / sc) * ((H + sc - 1) / sc) * B;
  const size_t n_chunks = (size_t)(T + kTopChunk - 1) / kTopChunk;
  return align_up(n_sc * T * sizeof(int32_t), 256) + align_up(n_sc * n_chunks * sizeof(int32_t), 256);
}

size_t raster_forward_ws(int B, int V, int T, int W, int H) {
  (void)V;
  const size_t nbt = (size_t)B * T;
  const int cell = kCellRegions * region_edge(B, W, H);
  const size_t cells = (size_t)((W + cell - 1) / cell) * ((H + cell - 1) / cell) * B;
  const int edge = region_edge(B, W, H);
  const size_t regions = (size_t)((W + edge - 1) / edge) * ((H + edge - 1) / edge) * B;
  return align_up(nbt * sizeof(TriRec), 256) + align_up(nbt * sizeof(TriBox), 256) +
         align_up((size_t)W * sizeof(float), 256) + align_up((size_t)H * sizeof(float), 256) +
         align_up(cells * T * sizeof(int32_t), 256) + 2 * align_up(cells * sizeof(int32_t), 256) +
         align_up(regions * kRegionListCap * sizeof(int32_t), 256) + align_up(regions * sizeof(int32_t), 256) +
         256 + align_up((size_t)kXcds * kWeightClasses * ((regions + kXcds - 1) / kXcds) * sizeof(int32_t), 256) +
         coarse_top_bytes(B, T, W, H, cell);
}

#ifdef MR_PROBES
thread_local int g_raster_probe = 0;  // stage-timing probe of the NEXT launches on this thread (debug builds only)
#endif

namespace {
struct RasterArgs {
  const TriRec *recs; const TriBox *bbs; const float *pxtab, *pytab;
  int T, W, H, regions_x, per_image, n_regions, per_xcd;
  const int32_t *cell_ids, *cell_count;
  const float *cell_split;
  int cells_x, cells_per_image;
  const int32_t *region_ids, *region_count, *order_count, *order_list;
  int32_t *ids; float *bary, *z;
  RasterShade shade;  // rgba == nullptr: G-buffer only
};

template <int R, int PROBE, bool SHADE = false, int INTERP = 0, int AX = 0, int XREC = 0, bool NORMS = false>
void launch_k_raster(const RasterArgs &a, dim3 grid, hipStream_t s) {
  hipLaunchKernelGGL((k_raster<R, PROBE, SHADE, INTERP, AX, XREC, NORMS>), grid, dim3(kThreads), 0, s, a.recs, a.bbs, a.pxtab, a.pytab, a.T,
                     a.W, a.H, a.regions_x, a.per_image, a.n_regions, a.per_xcd, a.cell_ids, a.cell_count,
                     a.cell_split, a.cells_x, a.cells_per_image, a.region_ids, a.region_count, a.order_count,
                     a.order_list, a.ids, a.bary, a.z, a.shade);
}

template <int R>
void launch_k_raster_probe(const RasterArgs &a, dim3 grid, hipStream_t s) {
  if (a.shade.rgba) {
    // crowded launches (triangles per 64 x 64 pixels of image): the instantiation with extra record slots, see XREC
    if (MR_RASTER_XREC > 0 && R == 64 && (double)a.T * 4096.0 >= (double)MR_RASTER_XREC_DENSITY * a.W * a.H)
      return launch_k_raster<R, 0, true, 0, 0, (R == 64 ? MR_RASTER_XREC : 0)>(a, grid, s);
    return launch_k_raster<R, 0, true>(a, grid, s);
  }
  if (a.shade.norm_partials) return launch_k_raster<R, 0, false, 8, 6, 0, true>(a, grid, s);   // the specular norm as the epilogue
  if (a.shade.attr_out) {  // rasterize()'s interpolation as the epilogue, attribute count padded to 4 / 8 / 12 / 16
    if (a.shade.A == 9) return launch_k_raster<R, 0, false, 12, 9>(a, grid, s);   // (normal, position, colour: render()'s set)
    if (a.shade.A == 3) return launch_k_raster<R, 0, false, 4, 3>(a, grid, s);
    switch ((a.shade.A + 3) / 4) {
      case 1: return launch_k_raster<R, 0, false, 4>(a, grid, s);
      case 2: return launch_k_raster<R, 0, false, 8>(a, grid, s);
      case 3: return launch_k_raster<R, 0, false, 12>(a, grid, s);
      default: return launch_k_raster<R, 0, false, 16>(a, grid, s);
    }
  }
#ifdef MR_PROBES
  switch (g_raster_probe) {
    case 1: return launch_k_raster<R, 1>(a, grid, s);
    case 2: return launch_k_raster<R, 2>(a, grid, s);
    case 3: return launch_k_raster<R, 3>(a, grid, s);
    case 8: return launch_k_raster<R, 8>(a, grid, s);
    case 16: return launch_k_raster<R, 16>(a, grid, s);
    case 32: return launch_k_raster<R, 32>(a, grid, s);
    case 40: return launch_k_raster<R, 40>(a, grid, s);
    case 48: return launch_k_raster<R, 48>(a, grid, s);
    case 64: return launch_k_raster<R, 64>(a, grid, s);
    default: break;
  }
#endif
  launch_k_raster<R, 0>(a, grid, s);
}
}  // namespace

namespace {
int raster_forward(const float *clip, const int32_t *tris, int B, int V, int T, int W, int H, int32_t *ids,
                   float *bary, float *z, const RasterShade &shade, const SetupAttributes &attrs, void *ws,
                   hipStream_t s) {
  const size_t nbt = (size_t)B * T;
  char *p = (char *)ws;
  TriRec *recs = (TriRec *)p;
  p += align_up(nbt * sizeof(TriRec), 256);
  TriBox *bbs = (TriBox *)p;
  p += align_up(nbt * sizeof(TriBox), 256);
  float *pxtab = (float *)p;
  p += align_up((size_t)W * sizeof(float), 256);
  float *pytab = (float *)p;
  p += align_up((size_t)H * sizeof(float), 256);
  const int edge = region_edge(B, W, H), cell = kCellRegions * edge;
  const int cells_x = (W + cell - 1) / cell, cells_y = (H + cell - 1) / cell;
  const int cells_per_image = cells_x * cells_y;
  int32_t *cell_ids = (int32_t *)p;
  p += align_up((size_t)cells_per_image * B * T * sizeof(int32_t), 256);
  int32_t *cell_count = (int32_t *)p;
  p += align_up((size_t)cells_per_image * B * sizeof(int32_t), 256);
  float *cell_split = (float *)p;
  p += align_up((size_t)cells_per_image * B * sizeof(int32_t), 256);
  const int regions_x = (W + edge - 1) / edge;
  const int regions_y = (H + edge - 1) / edge;
  const int per_image = regions_x * regions_y;
  const int n_regions = per_image * B;
  int32_t *region_ids = (int32_t *)p;
  p += align_up((size_t)n_regions * kRegionListCap * sizeof(int32_t), 256);
  int32_t *region_count = (int32_t *)p;
  p += align_up((size_t)n_regions * sizeof(int32_t), 256);
  const int per_xcd = (n_regions + kXcds - 1) / kXcds;
  int32_t *order_count = (int32_t *)p;   // [kXcds][kWeightClasses]
  p += 256;
  int32_t *order_list = (int32_t *)p;    // [kXcds][kWeightClasses][per_xcd]
  p += align_up((size_t)kXcds * kWeightClasses * per_xcd * sizeof(int32_t), 256);
  const bool top = coarse_top_used(T, W, H, cell);
  const int sc_size = kSuperCells * cell, sc_x = (W + sc_size - 1) / sc_size, sc_per_image = sc_x * ((H + sc_size - 1) / sc_size);
  const int n_chunks = (T + kTopChunk - 1) / kTopChunk;
  int32_t *top_ids = top ? (int32_t *)p : nullptr;
  int32_t *top_counts = top ? (int32_t *)(p + align_up((size_t)sc_per_image * B * T * sizeof(int32_t), 256)) : nullptr;

  const long n_main = (attrs.xf && (long)B * V > (long)nbt) ? (long)B * V : (long)nbt;
  const long setup_threads = n_main + W + H;
  const unsigned setup_blocks = (unsigned)((setup_threads + kThreads - 1) / kThreads);
  hipLaunchKernelGGL(k_setup, dim3(setup_blocks), dim3(kThreads), 0, s, (const float4 *)clip, tris,
                     B, V, T, W, H, recs, bbs, pxtab, pytab, attrs, order_count, n_main);
  int rc = check_launch();
  if (rc != MR_OK) return rc;
  if (B == 0) return MR_OK;
  if (top) {
    const dim3 top_grid((unsigned)((size_t)B * sc_per_image * n_chunks));
    hipLaunchKernelGGL(k_coarse_top<false>, top_grid, dim3(kCoarseThreads), 0, s, bbs, T, W, H, sc_x, sc_per_image, sc_size,
                       n_chunks, top_counts, top_ids);
    hipLaunchKernelGGL(k_coarse_top<true>, top_grid, dim3(kCoarseThreads), 0, s, bbs, T, W, H, sc_x, sc_per_image, sc_size,
                       n_chunks, top_counts, top_ids);
    if ((rc = check_launch()) != MR_OK) return rc;
  }
  hipLaunchKernelGGL(k_coarse, dim3((unsigned)(cells_per_image * B)), dim3(kCoarseThreads), 0, s, bbs, T,
                     W, H, cells_x, cells_per_image, cell, cell_ids, cell_count, cell_split, regions_x, regions_y,
                     region_ids, region_count, per_xcd, order_count, order_list, top_ids, top_counts, sc_x, sc_per_image,
                     n_chunks);
  rc = check_launch();
  if (rc != MR_OK) return rc;

  const dim3 grid((unsigned)(per_xcd * kXcds));
  const RasterArgs args{recs, bbs, pxtab, pytab, T, W, H, regions_x, per_image, n_regions, per_xcd,
                        cell_ids, cell_count, cell_split, cells_x, cells_per_image, region_ids, region_count,
                        order_count, order_list, ids, bary, z, shade};
  {
    KernelTimer timer(MR_TIMER_RASTER_FORWARD, s);  // records only when a caller armed it
    // (mr_debug_set_raster_repeat: the same launch n times back to back inside ONE event pair -- the kernel rewrites
    //  the same outputs from the same lists -- so that the ~5 us an event pair costs is spread over n launches)
    for (int r = 0; r < g_raster_repeat; ++r) {
      if (edge == 32) launch_k_raster_probe<32>(args, grid, s);
      else launch_k_raster_probe<64>(args, grid, s);
    }
  }
  return check_launch();
}
}  // namespace

int launch_raster_forward(const float *clip, const int32_t *tris, int B, int V, int T, int W,
                          int H, int32_t *ids, float *bary, float *z, void *ws, hipStream_t s) {
  return raster_forward(clip, tris, B, V, T, W, H, ids, bary, z,
                        RasterShade{nullptr, Lights{nullptr, nullptr, nullptr, 0}, nullptr, nullptr, 1, nullptr, nullptr, nullptr, 0, nullptr},
                        SetupAttributes{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr}, ws, s);
}

// rasterize_clip_space() forward for up to 16 attributes in ONE pass over the pixels (round 4): the attribute
// records (interp_fused.hip: one [3][AP] record per (image, triangle)), then k_setup / k_coarse / k_raster with
// the interpolation as the tile walk's epilogue.  The depth plane is scratch (`z`: a [B,H,W] buffer the kernel
// may use as state between the bin rounds of a crowded region).
int launch_rasterize_interpolate_forward(const float *clip, const float *attrs, const int32_t *tris, const float *background,
                                         int B, int V, int T, int W, int H, int A, int32_t *ids, float *bary, float *z,
                                         float *out, void *records, void *ws, hipStream_t s) {
  if ((size_t)B * W * H == 0) return MR_OK;
  const int rc = launch_attr_records(attrs, tris, B, V, T, A, records, s);
  if (rc != MR_OK) return rc;
  return raster_forward(clip, tris, B, V, T, W, H, ids, bary, z,
                        RasterShade{nullptr, Lights{nullptr, nullptr, nullptr, 0}, nullptr, nullptr, 0, (const float *)records,
                                    background, out, A, nullptr},
                        SetupAttributes{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr}, ws, s);
}

// ---- the G-buffer AND the specular term's across-pixels norms in one pass over the pixels (round 5) --------------
// render() with a specular term used to rasterize, then run shade_spec.hip's norm pass over the G-buffer (150 us at
// 1024^2 x 32: a dependent G-buffer -> corner record gather per pixel) before the pass that writes the image.  The
// norm -- per (image, light) the sum over ALL pixels of (reflection . camera)^2, render.py:342-348 -- only needs each
// covered pixel's interpolated normal and position, which the tile walk has at hand (k_raster<..., NORMS>); every
// uncovered pixel carries the attributes -1 and the same value, added as count x value by k_spec_norm_finish.
namespace {
// [B*T][3][8]: corner k = normal (3), position (3), two zeros
__global__ __launch_bounds__(kThreads) void k_norm_records(const F3 *__restrict__ normals, const F3 *__restrict__ positions,
                                                           const int32_t *__restrict__ tris, int B, int V, int T,
                                                           float4 *__restrict__ out) {
  const long gid = (long)blockIdx.x * kThreads + threadIdx.x;
  if (gid >= (long)B * T) return;
  const int b = (int)(gid / T), t = (int)(gid - (long)b * T);
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    int vi = tris[3 * t + k];
    if ((unsigned)vi >= (unsigned)V) vi = 0;   // (as gather_corner_values: such a triangle is never drawn)
    const F3 n = normals[(size_t)b * V + vi], p = positions[(size_t)b * V + vi];
    out[gid * 6 + 2 * k] = make_float4(n.x, n.y, n.z, p.x);
    out[gid * 6 + 2 * k + 1] = make_float4(p.y, p.z, 0.0f, 0.0f);
  }
}

// One workgroup per image: norms2[image][l] = the regions' partial sums in a fixed order + (uncovered pixels) x the
// background's value.
__global__ __launch_bounds__(kThreads) void k_spec_norm_finish(const float *__restrict__ partials, int regions_per_image,
                                                               const float *__restrict__ light_pos, const float *__restrict__ camera,
                                                               int L, long pixels_per_image, float *__restrict__ norms2) {
  __shared__ float s_part[kThreads / kWave][5];
  const int img = (int)blockIdx.x, tid = (int)threadIdx.x, lane = tid & (kWave - 1), wave = tid >> 6;
  const float *mine = partials + (size_t)img * regions_per_image * 8;
  float v[5] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
  for (int i = tid; i < regions_per_image; i += kThreads) {
    const float4 a = *(const float4 *)(mine + (size_t)i * 8);
    v[0] += a.x; v[1] += a.y; v[2] += a.z; v[3] += a.w;
    v[4] += mine[(size_t)i * 8 + 4];
  }
#pragma unroll
  for (int k = 0; k < 5; ++k) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v[k] += __shfl_down(v[k], off);
    if (lane == 0) s_part[wave][k] = v[k];
  }
  __syncthreads();
  if (tid < L) {
    float sum = 0.0f, covered = 0.0f;
    for (int w = 0; w < kThreads / kWave; ++w) {
      sum += s_part[w][tid];
      covered += s_part[w][4];   // (whole numbers below 2^24 per region row: exact)
    }
    float at[6] = {-1.0f, -1.0f, -1.0f, -1.0f, -1.0f, -1.0f};   // the background of render.py:197
    const float cam[3] = {camera[img * 3], camera[img * 3 + 1], camera[img * 3 + 2]};
    const float *lp = light_pos + ((size_t)img * L + tid) * 3;
    const float lpos[3] = {lp[0], lp[1], lp[2]};
    spec::PixelFrame f;
    spec::pixel_frame(at, cam, f);
    spec::LightTerm lt;
    spec::light_term(at, f, lpos, lt);
    norms2[(size_t)img * L + tid] = sum + ((float)pixels_per_image - covered) * (lt.rdc * lt.rdc);
  }
}
}  // namespace

size_t rasterize_specular_norms_ws(int B, int V, int T, int W, int H) {
  const int edge = region_edge(B, W, H);
  const size_t regions = (size_t)((W + edge - 1) / edge) * ((H + edge - 1) / edge) * B;
  return align_up(raster_forward_ws(B, V, T, W, H), 256) + align_up((size_t)B * T * 24 * sizeof(float), 256) +
         align_up(regions * 8 * sizeof(float), 256);
}

int launch_rasterize_specular_norms(const float *clip, const int32_t *tris, const float *normals, const float *positions,
                                    const float *light_pos, const float *camera, int B, int V, int T, int W, int H, int L,
                                    int32_t *ids, float *bary, float *z, int want_z, float *norms2, void *ws, hipStream_t s) {
  if (B == 0) return MR_OK;
  if (L < 1 || L > 4 || T < 1 || V < 1) return MR_EINVAL;
  const int edge = region_edge(B, W, H);
  const int per_image = ((W + edge - 1) / edge) * ((H + edge - 1) / edge);
  float *records = (float *)((char *)ws + align_up(raster_forward_ws(B, V, T, W, H), 256));
  float *partials = (float *)((char *)records + align_up((size_t)B * T * 24 * sizeof(float), 256));
  if (zero_async(partials, (size_t)B * per_image * 8 * sizeof(float), s) != hipSuccess) return check_launch();
  const long nbt = (long)B * T;
  if ((size_t)W * H > 0) {
    hipLaunchKernelGGL(k_norm_records, dim3((unsigned)((nbt + kThreads - 1) / kThreads)), dim3(kThreads), 0, s,
                       (const F3 *)normals, (const F3 *)positions, tris, B, V, T, (float4 *)records);
    int rc = check_launch();
    if (rc != MR_OK) return rc;
    RasterShade shade{nullptr, Lights{light_pos, nullptr, nullptr, L}, nullptr, nullptr, want_z, records, nullptr, nullptr, 6,
                      nullptr};
    shade.camera = camera;
    shade.norm_partials = partials;
    rc = raster_forward(clip, tris, B, V, T, W, H, ids, bary, z, shade,
                        SetupAttributes{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr}, ws, s);
    if (rc != MR_OK) return rc;
  }
  hipLaunchKernelGGL(k_spec_norm_finish, dim3((unsigned)B), dim3(kThreads), 0, s, partials, per_image, light_pos, camera, L,
                     (long)W * H, norms2);
  return check_launch();
}

int launch_vertex_transform(const float *vertices, const float *transforms, int B, int V, float *clip,
                            hipStream_t s) {
  const long nbv = (long)B * V;
  if (nbv == 0) return MR_OK;
  hipLaunchKernelGGL(k_vertex_transform, dim3((unsigned)((nbv + kThreads - 1) / kThreads)), dim3(kThreads), 0, s,
                     (const F3 *)vertices, (const float4 *)transforms, B, V, (float4 *)clip);
  return check_launch();
}

// render()'s forward from world-space vertices to the image, four launches: the clip-space transform,
// the per-triangle setup (rasterizer records AND the shading's corner records), the coarse cell lists,
// and ONE pass over the pixels that leaves the G-buffer and the shaded RGBA (see RasterShade).
// `clip` (out) and `corner_records` (out, shade_forward_ws() bytes) are what the backward needs.
int launch_render_forward(const float *vertices, const float *transforms, const float *normals,
                          const float *diffuse, const int32_t *tris, const float *light_pos,
                          const float *light_col, const float *ambient, int B, int V, int T, int W, int H,
                          int L, float *clip, int32_t *ids, float *bary, float *z, int want_z, float *rgba,
                          uint8_t *rgba_u8, void *corner_records, void *backward_prepared, uint8_t *empty_regions, void *ws,
                          hipStream_t s) {
  if ((size_t)B * W * H == 0 || !MR_SETUP_TRANSFORMS)   // (no pixels: the setup kernel does not run; the clip-space vertices are still an output)
  {
    const int rc = launch_vertex_transform(vertices, transforms, B, V, clip, s);
    if (rc != MR_OK) return rc;
  }
  if (empty_regions && region_edge(B, W, H) != 64 && (size_t)B * W * H > 0) {   // 32-pixel regions (small launches): nothing is flagged
    if (zero_async(empty_regions, (size_t)B * ((H + 63) / 64) * ((W + 63) / 64), s) != hipSuccess) return check_launch();
  }
  if ((size_t)B * W * H == 0) return MR_OK;
  CornerRec *corners = (CornerRec *)corner_records;
  SetupAttributes setup{(const F3 *)normals, (const F3 *)vertices, (const F3 *)diffuse, corners,
                        (FoldRec *)backward_prepared,
                        backward_prepared ? (float4 *)((char *)backward_prepared + fold_prepared_recs_bytes(B, T))
                                          : nullptr,
                        transforms};
  if (MR_SETUP_TRANSFORMS) {   // the clip-space transform rides k_setup (SetupAttributes::xf)
    setup.xf = (const float4 *)transforms;
    setup.clip_out = (float4 *)clip;
  }
  return raster_forward(clip, tris, B, V, T, W, H, ids, bary, z,
                        RasterShade{corners, Lights{light_pos, light_col, ambient, L}, rgba, (uint32_t *)rgba_u8, want_z, nullptr,
                                    nullptr, nullptr, 0, region_edge(B, W, H) == 64 ? empty_regions : nullptr},
                        setup, ws, s);
}

}  // namespace mr
