// SoftRas renderer for gfx950 (MI355X): probabilistic coverage + softmax depth aggregation,
// forward and hand-derived backward.
//
// Replaces rasterize_batch (reference: src/soft_mesh_renderer/rasterize.py:212-424) and the
// autograd graph torch builds behind it.  The reference is a per-pixel Python loop over the
// triangles a bbox quadtree returns (~165 pixels/s); the quadtree is only an index, the
// candidate set of a pixel is "triangles whose blur-inflated NDC bbox contains the pixel
// centre, inclusive" (src/soft_mesh_renderer/quadtree.py:18-31), which is what the tile bins
// below reproduce.
//
//   k_soft_setup     one thread per (image, triangle): NDC corners, the analytic inverse of
//                    [[x],[y],[1]] (barycentric rows), signed area (rasterize.py:120-123,301:
//                    area >= 0 = back-facing or degenerate, culled), inflated bbox.  128-B record.
//   k_soft_forward   one 256-thread workgroup per 16x16-pixel tile, one pixel per thread.  The
//                    image's triangles are tested against the tile 256 at a time and compacted
//                    in id order into an LDS list; every thread then walks the list (record
//                    reads are wave-uniform): bbox containment, screen barycentrics, nearest
//                    point on the three edges (rasterize.py:144-176), inside / blur cull,
//                    perspective-corrected sample barycentrics, depth, diffuse Phong colour
//                    (rasterize.py:183-208), sigmoid coverage, and an online softmax over depth
//                    (rasterize.py:397-414).  Writes RGBA (row 0 = top, no flip) and the
//                    per-pixel (max logit, weight sum, prod(1-D)) the backward needs.
//   k_soft_backward  same walk; per (pixel, triangle) pair the forward is recomputed and
//                    back-propagated by hand to the 39 inputs of the triangle (clip xyzw,
//                    position, normal, diffuse of 3 corners) and to the lights; the 39 partials
//                    are summed over the wavefront's 64 pixels with a butterfly and leave as 39
//                    global float atomics per (wavefront, triangle).
//
// Parity bar: 1e-4 abs on RGBA and gradients.  At the reference's default gamma = 1e-4 one ulp
// of depth moves a logit by ~1e-3, so depth follows the reference's operation order; the 3x3
// inverse is analytic here (torch's is LU), which was measured to cost ~2.5e-5 on RGB.
#include "corner_rec.h"

namespace mr {
extern thread_local int g_deterministic;  // mr_set_deterministic (shade.hip)
namespace {

constexpr int kThreads = 256;
constexpr int kTile = 16;          // pixels per tile edge
constexpr int kListCap = 512;      // LDS candidate list (triangle ids) per pass
constexpr int kSoftStride = 28;    // floats per parked row in k_soft_backward (25 used; 28 = conflict-free b128)
constexpr int kMaxLights = 4;
constexpr float kEps = 1e-10f;     // rasterize.py:211
constexpr float kNormEps = 1e-12f;

// gfx950 has no scalar floating-point ALU: whatever a kernel computes from a record's fields alone -- the same
// value in all 64 lanes -- costs a vector instruction per wavefront and candidate.  The edges' unit
// directions and inverse lengths (a sqrt and two reciprocals per edge at a quarter of the vector rate) and the
// reciprocals of w are therefore formed once per triangle here, with the expressions the pixel kernels used.
struct alignas(64) SoftRec {
  float x[3], y[3], zn[3], w[3];   // NDC corners and clip w
  float minv[9];                   // rows = barycentric coefficients (a, b, c): bc_i = a x + b y + c
  float lo[2], hi[2];              // blur-inflated NDC bbox
  float valid;                     // 1 = front-facing, non-degenerate
  float nx[3], ny[3], ilen[3];     // edges 01, 12, 20: unit direction (rasterize.py:169-172) and 1 / length
  float iw[3];                     // 1 / w
  float pad[10];
};
static_assert(sizeof(SoftRec) == 192, "three 64-byte scalar loads");

struct SoftParams {
  float sigma, gamma, blur;
  float inv_sigma, inv_gamma, blur2;  // formed on the host: a uniform division inside a kernel is vector work per pair
};
inline SoftParams soft_params(float sigma, float gamma, float blur) {
  return SoftParams{sigma, gamma, blur, 1.0f / sigma, 1.0f / gamma, blur * blur};
}

// Wave-uniform part of point_to_segment_nearest (rasterize.py:169-172) for one edge a -> b: the unit
// direction n = ab / max(|ab|, 1e-12) and 1 / |ab|; formed once per (image, triangle) by k_soft_setup.
__device__ __forceinline__ void edge_setup(float abx, float aby, float &nx, float &ny, float &ilen) {
  const float len = sqrtf(abx * abx + aby * aby);
  const float il = 1.0f / fmaxf(len, kNormEps);
  nx = abx * il;
  ny = aby * il;
  ilen = 1.0f / len;
}

__global__ __launch_bounds__(kThreads) void k_soft_setup(
    const float4 *__restrict__ clip, const int32_t *__restrict__ tris, int B, int V, int T,
    float blur, SoftRec *__restrict__ recs) {
  const long gid = (long)blockIdx.x * kThreads + threadIdx.x;
  if (gid >= (long)B * T) return;
  const int b = (int)(gid / T);
  const int t = (int)(gid - (long)b * T);
  SoftRec r;
  r.valid = 0.f;
  for (int k = 0; k < (int)(sizeof(r.pad) / sizeof(float)); ++k) r.pad[k] = 0.f;
  const int vi[3] = {tris[3 * t], tris[3 * t + 1], tris[3 * t + 2]};
  bool ok = true;
#pragma unroll
  for (int k = 0; k < 3; ++k) ok = ok && (unsigned)vi[k] < (unsigned)V;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const float4 c = ok ? clip[(long)b * V + vi[k]] : make_float4(0.f, 0.f, 0.f, 1.f);
    r.x[k] = c.x / c.w;  // rasterize.py:287
    r.y[k] = c.y / c.w;
    r.zn[k] = c.z / c.w;
    r.w[k] = c.w;
  }
  // signed area, rasterize.py:120-123 with (p, v0, v1) = (V0, V1, V2)
  const float area = (r.x[0] - r.x[1]) * (r.y[2] - r.y[1]) - (r.y[0] - r.y[1]) * (r.x[2] - r.x[1]);
  // inverse of [[x0 x1 x2],[y0 y1 y2],[1 1 1]]: row i gives barycentric i
  const float det = r.x[0] * (r.y[1] - r.y[2]) - r.x[1] * (r.y[0] - r.y[2]) + r.x[2] * (r.y[0] - r.y[1]);
  const float inv = 1.0f / det;
  r.minv[0] = (r.y[1] - r.y[2]) * inv; r.minv[1] = (r.x[2] - r.x[1]) * inv; r.minv[2] = (r.x[1] * r.y[2] - r.x[2] * r.y[1]) * inv;
  r.minv[3] = (r.y[2] - r.y[0]) * inv; r.minv[4] = (r.x[0] - r.x[2]) * inv; r.minv[5] = (r.x[2] * r.y[0] - r.x[0] * r.y[2]) * inv;
  r.minv[6] = (r.y[0] - r.y[1]) * inv; r.minv[7] = (r.x[1] - r.x[0]) * inv; r.minv[8] = (r.x[0] * r.y[1] - r.x[1] * r.y[0]) * inv;
  r.lo[0] = fminf(fminf(r.x[0], r.x[1]), r.x[2]) - blur;  // rasterize.py:302-305
  r.lo[1] = fminf(fminf(r.y[0], r.y[1]), r.y[2]) - blur;
  r.hi[0] = fmaxf(fmaxf(r.x[0], r.x[1]), r.x[2]) + blur;
  r.hi[1] = fmaxf(fmaxf(r.y[0], r.y[1]), r.y[2]) + blur;
#pragma unroll
  for (int e = 0; e < 3; ++e) {
    const int a = e, b2 = (e + 1) % 3;
    const float abx = r.x[b2] - r.x[a], aby = r.y[b2] - r.y[a];
    edge_setup(abx, aby, r.nx[e], r.ny[e], r.ilen[e]);
    r.iw[e] = 1.0f / r.w[e];
  }
  // rasterize.py:331-336: area > 0 back-facing, == 0 degenerate; a singular matrix leaves area 0
  r.valid = (ok && !(area >= 0.0f) && det != 0.0f) ? 1.f : 0.f;
  recs[gid] = r;
}

// Everything the forward computes for one (pixel, triangle) pair that survives the culls.
struct Pair {
  float bc[3];          // screen barycentrics
  float t[3], d2[3];    // nearest-point parameter / squared distance per edge (01, 12, 20)
  int edge;             // argmin edge
  bool inside;
  float u[3], q[3], s1, sb[3];
  float z, dist2;
  float kd[3], pos[3], nraw[3], nn, N[3];
  float lum;
  float D, logit;
  float c[3];
};

// ML: the light count the kernel is compiled for (1, or kMaxLights for 2..4): the lights are wave-uniform,
// i.e. scalar registers, and those are what the pair math runs out of (k_soft_backward spilled 109 of them
// to vector lanes with four lights' worth held).
template <int ML>
struct LightSet {
  float pos[ML][3];
  float inten[ML];
  int L;
};

__device__ __forceinline__ void edge_nearest(float px, float py, float ax, float ay, float abx, float aby, float nx,
                                             float ny, float ilen, float &t, float &d2) {  // rasterize.py:169-176
  const float dpn = (px - ax) * nx + (py - ay) * ny;
  const float prx = dpn * nx, pry = dpn * ny;
  t = fminf(fmaxf((prx * nx + pry * ny) * ilen, 0.0f), 1.0f);
  const float qx = ax + t * abx - px, qy = ay + t * aby - py;
  d2 = qx * qx + qy * qy;
}

// mr_debug_soft_nearest (mesh_raster_debug.h): the two device functions above on caller-given points and
// segments, so that the reference's own vectors for point_to_segment_nearest (test_rasterize.py:9-44) can be
// checked against what the SoftRas kernels evaluate.  out[i] = (nearest x, nearest y, t, squared distance).
__global__ __launch_bounds__(kThreads) void k_debug_soft_nearest(const float2 *__restrict__ p, const float2 *__restrict__ a,
                                                                 const float2 *__restrict__ b, int n,
                                                                 float4 *__restrict__ out) {
  const int i = (int)(blockIdx.x * kThreads + threadIdx.x);
  if (i >= n) return;
  const float abx = b[i].x - a[i].x, aby = b[i].y - a[i].y;
  float nx, ny, ilen, t, d2;
  edge_setup(abx, aby, nx, ny, ilen);
  edge_nearest(p[i].x, p[i].y, a[i].x, a[i].y, abx, aby, nx, ny, ilen, t, d2);
  out[i] = make_float4(a[i].x + t * abx, a[i].y + t * aby, t, d2);
}

// Returns false when the pair is culled (bbox, blur radius, depth range).
template <int ML>
__device__ __forceinline__ bool eval_pair(const SoftRec &r, const CornerRec *__restrict__ corner_rec, Corners &cr,
                                          const LightSet<ML> &ls,
                                          const SoftParams &pr, float px, float py, Pair &o) {
  if (!(px <= r.hi[0] && px >= r.lo[0] && py <= r.hi[1] && py >= r.lo[1])) return false;  // quadtree.py:18-31
#pragma unroll
  for (int i = 0; i < 3; ++i) o.bc[i] = r.minv[3 * i] * px + r.minv[3 * i + 1] * py + r.minv[3 * i + 2];
  edge_nearest(px, py, r.x[0], r.y[0], r.x[1] - r.x[0], r.y[1] - r.y[0], r.nx[0], r.ny[0], r.ilen[0], o.t[0], o.d2[0]);
  edge_nearest(px, py, r.x[1], r.y[1], r.x[2] - r.x[1], r.y[2] - r.y[1], r.nx[1], r.ny[1], r.ilen[1], o.t[1], o.d2[1]);
  edge_nearest(px, py, r.x[2], r.y[2], r.x[0] - r.x[2], r.y[0] - r.y[2], r.nx[2], r.ny[2], r.ilen[2], o.t[2], o.d2[2]);
  o.edge = 0;
  o.dist2 = o.d2[0];
  if (o.d2[1] < o.dist2) { o.edge = 1; o.dist2 = o.d2[1]; }
  if (o.d2[2] < o.dist2) { o.edge = 2; o.dist2 = o.d2[2]; }
  o.inside = !(o.bc[0] < 0.f || o.bc[1] < 0.f || o.bc[2] < 0.f);
  if (!o.inside && o.dist2 > pr.blur2) return false;  // rasterize.py:354
  if (o.inside) {
    o.u[0] = o.bc[0]; o.u[1] = o.bc[1]; o.u[2] = o.bc[2];
  } else if (o.edge == 0) {
    o.u[0] = 1.f - o.t[0]; o.u[1] = o.t[0]; o.u[2] = 0.f;
  } else if (o.edge == 1) {
    o.u[0] = 0.f; o.u[1] = 1.f - o.t[1]; o.u[2] = o.t[1];
  } else {
    o.u[0] = o.t[2]; o.u[1] = 0.f; o.u[2] = 1.f - o.t[2];
  }
  o.s1 = 0.f;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    o.q[k] = o.u[k] * r.iw[k];
    o.s1 += fabsf(o.q[k]);
  }
  const float is1 = 1.0f / fmaxf(o.s1, kNormEps);  // F.normalize(p=1), rasterize.py:359-365
#pragma unroll
  for (int k = 0; k < 3; ++k) o.sb[k] = o.q[k] * is1;
  const float zz = o.sb[0] * r.zn[0] + o.sb[1] * r.zn[1] + o.sb[2] * r.zn[2];
  o.z = 0.5f - zz / 2.0f;  // rasterize.py:368-370
  if (o.z < 0.0f || o.z > 1.0f) return false;
#pragma unroll
  for (int c = 0; c < 3; ++c) {  // rasterize.py:194-196; corner record rows: normal, position, diffuse
    o.nraw[c] = o.sb[0] * cr.c[0][c] + o.sb[1] * cr.c[1][c] + o.sb[2] * cr.c[2][c];
    o.pos[c] = o.sb[0] * cr.c[0][3 + c] + o.sb[1] * cr.c[1][3 + c] + o.sb[2] * cr.c[2][3 + c];
    o.kd[c] = o.sb[0] * cr.c[0][6 + c] + o.sb[1] * cr.c[1][6 + c] + o.sb[2] * cr.c[2][6 + c];
  }
  o.nn = sqrtf(o.nraw[0] * o.nraw[0] + o.nraw[1] * o.nraw[1] + o.nraw[2] * o.nraw[2]);
  const float inn = 1.0f / fmaxf(o.nn, kNormEps);
#pragma unroll
  for (int c = 0; c < 3; ++c) o.N[c] = o.nraw[c] * inn;
  o.lum = 0.f;
#pragma unroll
  for (int l = 0; l < ML; ++l) {  // rasterize.py:197-206 (unrolled: static register indices)
    if (ML > 1 && l >= ls.L) break;
    const float vx = ls.pos[l][0] - o.pos[0], vy = ls.pos[l][1] - o.pos[1], vz = ls.pos[l][2] - o.pos[2];
    const float ivn = 1.0f / fmaxf(sqrtf(vx * vx + vy * vy + vz * vz), kNormEps);
    const float ndl = fminf(fmaxf((vx * o.N[0] + vy * o.N[1] + vz * o.N[2]) * ivn, 0.0f), 1.0f);
    o.lum += ndl * ls.inten[l];
  }
#pragma unroll
  for (int c = 0; c < 3; ++c) o.c[c] = o.kd[c] * o.lum;
  const float xarg = (o.inside ? o.dist2 : -o.dist2) * pr.inv_sigma;  // rasterize.py:388-389
  o.D = 1.0f / (1.0f + expf(-xarg));
  o.logit = o.z * pr.inv_gamma;  // rasterize.py:394
  return true;
}

template <int ML>
__device__ __forceinline__ void load_lights(const float *lpos, const float *lint, int img, int L, LightSet<ML> &ls) {
  ls.L = L;
  for (int l = 0; l < ML; ++l) {
    const bool have = l < L;
    ls.inten[l] = have ? lint[(size_t)img * L + l] : 0.f;
    for (int c = 0; c < 3; ++c) ls.pos[l][c] = have ? lpos[((size_t)img * L + l) * 3 + c] : 0.f;
  }
}

// Compacts, in id order, the triangles cand[base .. base+256) (the id-ordered list of the tile's
// coarse cell, n_cand entries) whose inflated bbox touches the tile's pixel-centre rectangle
// into s_list; returns the new length (workgroup-uniform).
__device__ __forceinline__ int bin_chunk(const SoftRec *img_recs, const int32_t *cand, int n_cand, int base,
                                         float tx0, float tx1, float ty0, float ty1, int *s_list,
                                         int *s_wave_count, int n) {
  const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int k = base + tid;
  bool hit = false;
  int t = 0;
  if (k < n_cand) {
    t = cand[k];
    const SoftRec *r = img_recs + t;
    hit = r->valid != 0.f && r->lo[0] <= tx1 && r->hi[0] >= tx0 && r->lo[1] <= ty1 && r->hi[1] >= ty0;
  }
  const unsigned long long m = __ballot(hit);
  if (lane == 0) s_wave_count[wave] = __builtin_popcountll(m);
  __syncthreads();
  int offset = n, total = 0;
#pragma unroll
  for (int w = 0; w < kThreads / 64; ++w) {
    const int c = s_wave_count[w];
    if (w < wave) offset += c;
    total += c;
  }
  if (hit)
    s_list[offset + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u))] = t;
  __syncthreads();
  return n + total;
}

// Coarse binning: one 1024-thread workgroup per (image, cell of kCellTiles x kCellTiles tiles)
// compacts, in triangle-id order, the front-facing triangles whose inflated bbox touches the
// cell.  A tile then scans its cell's list (~T * ((cell + triangle) / image)^2 ids) instead of
// all T -- at 512^2 with 5k triangles that is 12x less scanning per tile.
constexpr int kCellTiles = 8;                  // tiles per cell edge: 128 x 128 pixels
constexpr int kCoarseThreads = 1024;

__device__ __forceinline__ void cell_rect(int cx, int cy, int W, int H, float &x0, float &x1, float &y0,
                                          float &y1) {
  const int px0 = cx * kCellTiles * kTile, px1 = min(px0 + kCellTiles * kTile, W) - 1;
  const int py0 = cy * kCellTiles * kTile, py1 = min(py0 + kCellTiles * kTile, H) - 1;
  // the same pixel-centre expressions as the tiles' rectangles (tile_geometry): a tile's rectangle
  // lies inside its cell's, so the cell list is a superset of every tile list
  x0 = (float)(2.0 * (((double)px0 + 0.5) / (double)W) - 1.0);
  x1 = (float)(2.0 * (((double)px1 + 0.5) / (double)W) - 1.0);
  y1 = (float)(-2.0 * (((double)py0 + 0.5) / (double)H) + 1.0);  // y grows downwards in the image
  y0 = (float)(-2.0 * (((double)py1 + 0.5) / (double)H) + 1.0);
}

__global__ __launch_bounds__(kCoarseThreads) void k_soft_coarse(
    const SoftRec *__restrict__ recs, int T, int W, int H, int cells_x, int cells_per_image,
    int32_t *__restrict__ cell_ids, int32_t *__restrict__ cell_count) {
  __shared__ int s_wave_count[kCoarseThreads / 64];
  const int img = (int)blockIdx.x / cells_per_image;
  const int cell = (int)blockIdx.x - img * cells_per_image;
  const int cy = cell / cells_x, cx = cell - cy * cells_x;
  float x0, x1, y0, y1;
  cell_rect(cx, cy, W, H, x0, x1, y0, y1);
  const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const SoftRec *img_recs = recs + (size_t)img * T;
  int32_t *out = cell_ids + ((size_t)img * cells_per_image + cell) * T;
  int n = 0;  // workgroup-uniform
  for (int base = 0; base < T; base += kCoarseThreads) {
    const int t = base + tid;
    bool hit = false;
    if (t < T) {
      const SoftRec *r = img_recs + t;
      hit = r->valid != 0.f && r->lo[0] <= x1 && r->hi[0] >= x0 && r->lo[1] <= y1 && r->hi[1] >= y0;
    }
    const unsigned long long m = __ballot(hit);
    if (lane == 0) s_wave_count[wave] = __builtin_popcountll(m);
    __syncthreads();
    int offset = n, total = 0;
#pragma unroll
    for (int w = 0; w < kCoarseThreads / 64; ++w) {
      const int c = s_wave_count[w];
      if (w < wave) offset += c;
      total += c;
    }
    if (hit)
      out[offset + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u))] = t;
    n += total;
    __syncthreads();
  }
  if (tid == 0) cell_count[(size_t)img * cells_per_image + cell] = n;
}

struct TileGeom {
  int img, x, y, cell, tile;  // tile: logical tile index over the whole batch
  bool in_image;
  float px, py, tx0, tx1, ty0, ty1;
};

__device__ __forceinline__ bool tile_geometry(int W, int H, int tiles_x, int tiles_per_image, int n_tiles,
                                              int tiles_per_xcd, TileGeom &g) {
  const int tile = xcd_contiguous_block((int)blockIdx.x, n_tiles, tiles_per_xcd);
  if (tile < 0) return false;
  g.tile = tile;
  g.img = tile / tiles_per_image;
  const int rr = tile - g.img * tiles_per_image;
  const int ty = rr / tiles_x, tx = rr - ty * tiles_x;
  g.cell = (ty / kCellTiles) * ((tiles_x + kCellTiles - 1) / kCellTiles) + tx / kCellTiles;
  const int tid = (int)threadIdx.x;
  // a wavefront = an 8 x 8 quadrant of the tile, not a 16 x 4 strip: a triangle (~8 px across + the blur
  // margin at config 5) then touches ~10 % fewer wavefronts, and each of them pays the full pair math
  const int wv = tid >> 6, ln = tid & 63;
  g.x = tx * kTile + (wv & 1) * 8 + (ln & 7);
  g.y = ty * kTile + (wv >> 1) * 8 + (ln >> 3);
  g.in_image = g.x < W && g.y < H;
  // pixel centres as the reference computes them: double arithmetic, then float32 (rasterize.py:315-317)
  g.px = (float)(2.0 * (((double)g.x + 0.5) / (double)W) - 1.0);
  g.py = (float)(-2.0 * (((double)g.y + 0.5) / (double)H) + 1.0);
  const int x0 = tx * kTile, x1 = min(x0 + kTile, W) - 1, y0 = ty * kTile, y1 = min(y0 + kTile, H) - 1;
  g.tx0 = (float)(2.0 * (((double)x0 + 0.5) / (double)W) - 1.0);
  g.tx1 = (float)(2.0 * (((double)x1 + 0.5) / (double)W) - 1.0);
  g.ty1 = (float)(-2.0 * (((double)y0 + 0.5) / (double)H) + 1.0);  // y grows downwards in the image
  g.ty0 = (float)(-2.0 * (((double)y1 + 0.5) / (double)H) + 1.0);
  return true;
}

template <int ML>
__global__ __launch_bounds__(kThreads) void k_soft_forward(
    const SoftRec *__restrict__ recs, const CornerRec *__restrict__ corners,
    const float *__restrict__ lpos, const float *__restrict__ lint, int T, int W, int H, int L,
    SoftParams pr, int tiles_x, int tiles_per_image, int n_tiles, int tiles_per_xcd,
    const int32_t *__restrict__ cell_ids, const int32_t *__restrict__ cell_count, int cells_per_image,
    float4 *__restrict__ rgba, float4 *__restrict__ aux) {
  __shared__ int s_list[kListCap];
  __shared__ int s_wave_count[kThreads / 64];
  TileGeom g;
  if (!tile_geometry(W, H, tiles_x, tiles_per_image, n_tiles, tiles_per_xcd, g)) return;
  const SoftRec *img_recs = recs + (size_t)g.img * T;
  const CornerRec *img_corners = corners + (size_t)g.img * T;
  LightSet<ML> ls;
  load_lights(lpos, lint, g.img, L, ls);

  float m = kEps * pr.inv_gamma;  // running max logit; the reference's floor (rasterize.py:397)
  float sw = 0.f, acc[3] = {0.f, 0.f, 0.f}, prod = 1.f;
  int n = 0;
  const int32_t *cand = cell_ids + ((size_t)g.img * cells_per_image + g.cell) * T;
  const int n_cand = cell_count[(size_t)g.img * cells_per_image + g.cell];
  for (int base = 0; base < n_cand; base += kThreads) {
    n = bin_chunk(img_recs, cand, n_cand, base, g.tx0, g.tx1, g.ty0, g.ty1, s_list, s_wave_count, n);
    if (n + kThreads > kListCap || base + kThreads >= n_cand) {  // list (nearly) full or last chunk: walk it
      for (int k = 0; k < n; ++k) {
        const int t = __builtin_amdgcn_readfirstlane(s_list[k]);  // workgroup-uniform
        const SoftRec r = img_recs[t];
        Corners cr;
        load_corners(img_corners + t, cr);
        Pair p;
        if (g.in_image && eval_pair(r, img_corners + t, cr, ls, pr, g.px, g.py, p)) {
          if (p.logit > m) {  // online softmax: rescale what has been summed so far
            const float sc = expf(m - p.logit);
            sw *= sc; acc[0] *= sc; acc[1] *= sc; acc[2] *= sc;
            m = p.logit;
          }
          const float wgt = p.D * expf(p.logit - m);
          sw += wgt;
          acc[0] += wgt * p.c[0]; acc[1] += wgt * p.c[1]; acc[2] += wgt * p.c[2];
          prod *= (1.0f - p.D);
        }
      }
      n = 0;
      __syncthreads();
    }
  }
  if (g.in_image) {
    const float bg = fmaxf(expf(kEps * pr.inv_gamma - m), kEps);  // rasterize.py:401
    const float S = sw + bg;
    const size_t pix = ((size_t)g.img * H + g.y) * W + g.x;
    rgba[pix] = make_float4(acc[0] / S, acc[1] / S, acc[2] / S, 1.0f - prod);
    aux[pix] = make_float4(m, S, prod, 0.f);
  }
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  return v;
}

constexpr int kLightRow = 16;  // floats per wavefront row of light sums (4 x kMaxLights)
#ifndef MR_SOFT_BWD_WAVES
#define MR_SOFT_BWD_WAVES 4   // round 3: 135 -> 128 VGPRs, four waves per SIMD instead of three: step 1.09 -> 1.01 ms
#endif
// DET (mr_set_deterministic, round 3): the 39 sums of a (wavefront, triangle) leave as 64-bit fixed-point
// integer atomics into int64 copies of the four outputs (det_fixed: dclip [B,V,4], then dnormals,
// dpositions, ddiffuse [B,V,3] each) instead of float atomics into the outputs; k_soft_from_fixed converts.
template <bool DET, int ML>
__global__ __launch_bounds__(kThreads, MR_SOFT_BWD_WAVES) void k_soft_backward(
    const SoftRec *__restrict__ recs, const CornerRec *__restrict__ corners,
    const float *__restrict__ lpos, const float *__restrict__ lint, const int32_t *__restrict__ tris,
    int V, int T, int W, int H, int L, SoftParams pr, int tiles_x, int tiles_per_image, int n_tiles,
    int tiles_per_xcd, const int32_t *__restrict__ cell_ids, const int32_t *__restrict__ cell_count,
    int cells_per_image, const float4 *__restrict__ drgba, const float4 *__restrict__ rgba,
    const float4 *__restrict__ aux, float *__restrict__ dclip, float *__restrict__ dnormals,
    float *__restrict__ dpositions, float *__restrict__ ddiffuse, float *__restrict__ light_rows,
    long long *__restrict__ det_fixed, const float *__restrict__ det_scale, int B) {
  __shared__ int s_list[kListCap];
  __shared__ int s_wave_count[kThreads / 64];
  TileGeom g;
  if (!tile_geometry(W, H, tiles_x, tiles_per_image, n_tiles, tiles_per_xcd, g)) return;
  const SoftRec *img_recs = recs + (size_t)g.img * T;
  const CornerRec *img_corners = corners + (size_t)g.img * T;
  LightSet<ML> ls;
  load_lights(lpos, lint, g.img, L, ls);
  const int lane = (int)threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);

  // Reduction of a triangle's 39 partials over the wavefront's 64 pixels (see the end of the
  // candidate loop): every pixel lane parks kSoftFactors values in an LDS row, then lane o < 39
  // sums factor[fa] * factor[fb] over the 64 rows.  27 of the 39 partials are outer products
  // sb[corner] x (d/d normal, position, diffuse): they are parked as 3 + 9 factors and
  // multiplied in the reduction; the 12 clip partials are parked as they are and multiplied by
  // the constant 1 in slot 24.  (Before: 39 wave butterflies of 6 ds_bpermute + 6 adds each and
  // 39 single-lane atomics per (wavefront, triangle).)
  __shared__ __attribute__((aligned(16))) float s_stage[kThreads / 64][64 * kSoftStride];
  float *stage = s_stage[wave];
  const int o_corner = lane / 13, o_comp = lane - 13 * o_corner;   // output o = corner * 13 + comp
  const int fa = (o_comp < 4) ? 12 + 4 * min(o_corner, 2) + o_comp : min(o_corner, 2);
  const int fb = (o_comp < 4) ? 24 : 3 + (o_comp - 4);
  const float *col_a = stage + fa, *col_b = stage + fb;
  // where lane o's sum goes: dclip[.., 4] for comp 0..3, then dnormals / dpositions / ddiffuse [.., 3]
  float *out_base = o_comp < 4 ? dclip : (o_comp < 7 ? dnormals : (o_comp < 10 ? dpositions : ddiffuse));
  const int out_stride = o_comp < 4 ? 4 : 3;
  const int out_off = o_comp < 4 ? o_comp : (o_comp - 4) % 3;
  // DET: the same element inside the int64 copy (the four arrays back to back)
  const size_t bv = (size_t)B * V;
  long long *fixed_base = !DET ? nullptr
                        : o_comp < 4 ? det_fixed : det_fixed + bv * 4 + (size_t)(o_comp < 7 ? 0 : (o_comp < 10 ? 1 : 2)) * bv * 3;
  const float to_fixed = DET ? det_scale[0] : 0.0f;

  float4 go = make_float4(0.f, 0.f, 0.f, 0.f), out = go, ax = make_float4(0.f, 1.f, 1.f, 0.f);
  if (g.in_image) {
    const size_t pix = ((size_t)g.img * H + g.y) * W + g.x;
    go = drgba[pix];
    out = rgba[pix];
    ax = aux[pix];
  }
  const float m = ax.x, inv_S = 1.0f / ax.y, prod = ax.z;
  float g_lp[ML][3], g_li[ML];
  for (int l = 0; l < ML; ++l) { g_li[l] = 0.f; g_lp[l][0] = g_lp[l][1] = g_lp[l][2] = 0.f; }

  int n = 0;
  const int32_t *cand = cell_ids + ((size_t)g.img * cells_per_image + g.cell) * T;
  const int n_cand = cell_count[(size_t)g.img * cells_per_image + g.cell];
  for (int base = 0; base < n_cand; base += kThreads) {
    n = bin_chunk(img_recs, cand, n_cand, base, g.tx0, g.tx1, g.ty0, g.ty1, s_list, s_wave_count, n);
    if (n + kThreads > kListCap || base + kThreads >= n_cand) {
      for (int k = 0; k < n; ++k) {
        const int t = __builtin_amdgcn_readfirstlane(s_list[k]);  // workgroup-uniform
        const SoftRec r = img_recs[t];
        Corners cr;
        load_corners(img_corners + t, cr);
        Pair p;
        const bool live = g.in_image && eval_pair(r, img_corners + t, cr, ls, pr, g.px, g.py, p);
        if (!__ballot(live)) continue;  // no pixel of this wavefront touches the triangle
        // this lane's corner vertex id for the commit at the end (latency hidden by the math)
        const int my_vertex = (lane < 39) ? tris[3 * t + o_corner] : 0;
        // gradient of this pixel's output w.r.t. the triangle's 39 inputs, as parked factors:
        // f[0..2] sb[corner] | f[3..11] d/d (normal, position, diffuse) before the sb factor |
        // f[12 + 4 corner + c] d/d clip xyzw of the corner | f[24] = 1
        float f[kSoftStride];
#pragma unroll
        for (int k = 0; k < kSoftStride; ++k) f[k] = 0.f;
        f[24] = 1.0f;
        if (live) {
          const float e = expf(p.logit - m);
          const float wgt = p.D * e;
          // rgb = sum_i w_i c_i / S with S = sum_i w_i + bg  (rasterize.py:397-410)
          const float g_w = (go.x * (p.c[0] - out.x) + go.y * (p.c[1] - out.y) + go.z * (p.c[2] - out.z)) * inv_S;
          const float g_c[3] = {go.x * wgt * inv_S, go.y * wgt * inv_S, go.z * wgt * inv_S};
          // alpha = 1 - prod(1 - D): d alpha / d x_i = prod * D_i  (x = +-d2/sigma, D = sigmoid(x))
          const float g_x = g_w * e * p.D * (1.0f - p.D) + go.w * prod * p.D;
          const float g_d2 = (p.inside ? g_x : -g_x) * pr.inv_sigma;
          const float g_z = g_w * wgt * pr.inv_gamma;
          // ---- colour (rasterize.py:183-208) ----
          float g_sb[3] = {0.f, 0.f, 0.f};
          float g_kd[3], g_pos[3] = {0.f, 0.f, 0.f}, g_N[3] = {0.f, 0.f, 0.f};
          float g_lum = 0.f;
#pragma unroll
          for (int c = 0; c < 3; ++c) {
            g_kd[c] = g_c[c] * p.lum;
            g_lum += g_c[c] * p.kd[c];
          }
#pragma unroll
          for (int l = 0; l < ML; ++l) {
            if (ML > 1 && l >= ls.L) break;
            const float v[3] = {ls.pos[l][0] - p.pos[0], ls.pos[l][1] - p.pos[1], ls.pos[l][2] - p.pos[2]};
            const float vn = sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
            const float ivn = 1.0f / fmaxf(vn, kNormEps);
            const float D3[3] = {v[0] * ivn, v[1] * ivn, v[2] * ivn};
            const float pre = D3[0] * p.N[0] + D3[1] * p.N[1] + D3[2] * p.N[2];
            const float ndl = fminf(fmaxf(pre, 0.0f), 1.0f);
            g_li[l] += g_lum * ndl;
            if (pre >= 0.0f && pre <= 1.0f) {  // clamp passes the gradient inclusively
              const float g_pre = g_lum * ls.inten[l];
              float dd = 0.f, gD[3];
#pragma unroll
              for (int c = 0; c < 3; ++c) { g_N[c] += g_pre * D3[c]; gD[c] = g_pre * p.N[c]; dd += D3[c] * gD[c]; }
#pragma unroll
              for (int c = 0; c < 3; ++c) {
                const float gvv = (vn > kNormEps ? (gD[c] - D3[c] * dd) : gD[c]) * ivn;
                g_lp[l][c] += gvv;
                g_pos[c] -= gvv;
              }
            }
          }
          float g_nraw[3];
          {
            const float inn = 1.0f / fmaxf(p.nn, kNormEps);
            const float nd = p.N[0] * g_N[0] + p.N[1] * g_N[1] + p.N[2] * g_N[2];
#pragma unroll
            for (int c = 0; c < 3; ++c) g_nraw[c] = (p.nn > kNormEps ? (g_N[c] - p.N[c] * nd) : g_N[c]) * inn;
          }
#pragma unroll
          for (int c = 0; c < 3; ++c) {
            f[3 + c] = g_nraw[c];
            f[6 + c] = g_pos[c];
            f[9 + c] = g_kd[c];
          }
#pragma unroll
          for (int a = 0; a < 3; ++a) {
            f[a] = p.sb[a];
#pragma unroll
            for (int c = 0; c < 3; ++c)
              g_sb[a] += g_nraw[c] * cr.c[a][c] + g_pos[c] * cr.c[a][3 + c] + g_kd[c] * cr.c[a][6 + c];
          }
          // ---- depth: z = 0.5 - (sb . zn) / 2 ----
          float g_zn[3];
#pragma unroll
          for (int a = 0; a < 3; ++a) {
            g_sb[a] += -0.5f * g_z * r.zn[a];
            g_zn[a] = -0.5f * g_z * p.sb[a];
          }
          // ---- sb = q / sum|q|, q = u / w ----
          float g_w4[3], g_u[3];
          {
            const float is1 = 1.0f / fmaxf(p.s1, kNormEps);
            const float dot = g_sb[0] * p.sb[0] + g_sb[1] * p.sb[1] + g_sb[2] * p.sb[2];
#pragma unroll
            for (int a = 0; a < 3; ++a) {
              const float sgn = p.q[a] > 0.f ? 1.f : (p.q[a] < 0.f ? -1.f : 0.f);
              const float g_q = p.s1 > kNormEps ? (g_sb[a] - sgn * dot) * is1 : g_sb[a] * is1;
              g_u[a] = g_q * r.iw[a];
              g_w4[a] = -g_q * p.q[a] * r.iw[a];
            }
          }
          // ---- u -> screen geometry ----
          float g_x2[3] = {0.f, 0.f, 0.f}, g_y2[3] = {0.f, 0.f, 0.f};  // d/d NDC x_k, y_k
          float g_t = 0.f;                                              // d/d t of the nearest edge
          if (p.inside) {
            // bc = Minv p, Minv = inverse(M2d):  dL/dM[r][k] = -(Minv^T g_bc)[r] * bc[k], rows x and y
            float mg[2];
#pragma unroll
            for (int rr = 0; rr < 2; ++rr)
              mg[rr] = r.minv[rr] * g_u[0] + r.minv[3 + rr] * g_u[1] + r.minv[6 + rr] * g_u[2];
#pragma unroll
            for (int a = 0; a < 3; ++a) {
              g_x2[a] -= mg[0] * p.bc[a];
              g_y2[a] -= mg[1] * p.bc[a];
            }
          } else if (p.edge == 0) {
            g_t = g_u[1] - g_u[0];
          } else if (p.edge == 1) {
            g_t = g_u[2] - g_u[1];
          } else {
            g_t = g_u[0] - g_u[2];
          }
          // ---- squared distance and t of the nearest edge (rasterize.py:169-176) ----
          {
            // selects, not r.x[p.edge]: a dynamically indexed register array lives in scratch memory
            const int ia = p.edge, ib = (ia == 0) ? 1 : (ia == 1 ? 2 : 0);
            const float ax2 = ia == 0 ? r.x[0] : (ia == 1 ? r.x[1] : r.x[2]);
            const float ay2 = ia == 0 ? r.y[0] : (ia == 1 ? r.y[1] : r.y[2]);
            const float bx = ia == 0 ? r.x[1] : (ia == 1 ? r.x[2] : r.x[0]);
            const float by = ia == 0 ? r.y[1] : (ia == 1 ? r.y[2] : r.y[0]);
            const float abx = bx - ax2, aby = by - ay2;
            const float iL2 = 1.0f / (abx * abx + aby * aby);
            const float tt = ia == 0 ? p.t[0] : (ia == 1 ? p.t[1] : p.t[2]);
            const float dvx = ax2 + tt * abx - g.px, dvy = ay2 + tt * aby - g.py;
            const float gxx = 2.0f * dvx * g_d2, gxy = 2.0f * dvy * g_d2;  // d/d nearest point
            float gax = gxx, gay = gxy;                                   // direct dependence on a
            float gabx = tt * gxx, gaby = tt * gxy;
            const float g_tt_total = g_t + gxx * abx + gxy * aby;
            // unclamped t = ((p - a) . ab) / |ab|^2; the clamp passes the gradient on [0, 1]
            const float num = (g.px - ax2) * abx + (g.py - ay2) * aby;
            const float traw = num * iL2;
            if (traw >= 0.0f && traw <= 1.0f) {
              const float g_num = g_tt_total * iL2;
              const float g_L2 = -g_tt_total * traw * iL2;
              gax -= g_num * abx; gay -= g_num * aby;
              gabx += g_num * (g.px - ax2) + 2.0f * g_L2 * abx;
              gaby += g_num * (g.py - ay2) + 2.0f * g_L2 * aby;
            }
            // ab = b - a
            const float fax = gax - gabx, fay = gay - gaby;
#pragma unroll
            for (int a = 0; a < 3; ++a) {
              if (a == ia) { g_x2[a] += fax; g_y2[a] += fay; }
              if (a == ib) { g_x2[a] += gabx; g_y2[a] += gaby; }
            }
          }
          // ---- NDC -> clip: x = cx / w, y = cy / w, zn = cz / w ----
#pragma unroll
          for (int a = 0; a < 3; ++a) {
            const float iw = r.iw[a];
            f[12 + 4 * a + 0] = g_x2[a] * iw;
            f[12 + 4 * a + 1] = g_y2[a] * iw;
            f[12 + 4 * a + 2] = g_zn[a] * iw;
            f[12 + 4 * a + 3] = g_w4[a] - (g_x2[a] * r.x[a] + g_y2[a] * r.y[a] + g_zn[a] * r.zn[a]) * iw;
          }
        }
        // park the row (zeros for pixels the triangle does not touch), then lane o sums its
        // product over the 64 rows and commits it with one atomic
        {
          float4 *row = (float4 *)(stage + lane * kSoftStride);
#pragma unroll
          for (int q = 0; q < kSoftStride / 4; ++q)
            row[q] = make_float4(f[4 * q], f[4 * q + 1], f[4 * q + 2], f[4 * q + 3]);
        }
        __builtin_amdgcn_wave_barrier();  // LDS executes one wavefront's operations in order
        float sum = 0.0f;
#pragma unroll
        for (int pb = 0; pb < 64; pb += 8) {
          float ra[8], rb[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            ra[j] = col_a[(pb + j) * kSoftStride];
            rb[j] = col_b[(pb + j) * kSoftStride];
          }
#pragma unroll
          for (int j = 0; j < 8; ++j) sum = fmaf(ra[j], rb[j], sum);
        }
        __builtin_amdgcn_wave_barrier();  // the next candidate overwrites the rows
        if (lane < 39 && sum != 0.0f && (unsigned)my_vertex < (unsigned)V) {
          const size_t at = ((size_t)g.img * V + my_vertex) * out_stride + out_off;
          if (DET) atomic_add_fixed(fixed_base + at, sum, to_fixed, det_overflow_flag(det_scale));
          else atomicAdd(out_base + at, sum);
        }
      }
      n = 0;
      __syncthreads();
    }
  }
  // The light gradients leave as ONE row per wavefront -- [3L position sums | L intensity sums] -- and
  // k_soft_light_sum adds an image's rows in a fixed order.  (As float atomics on the image's 4L
  // addresses they were 0.7 ms of this kernel's 1.26 at 512^2 x 16: 4096 wavefronts per image queued
  // up on one cache line.)
  float *row = light_rows + ((size_t)g.tile * (kThreads / 64) + wave) * kLightRow;
#pragma unroll
  for (int l = 0; l < ML; ++l) {
    if (l >= L) break;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float s = wave_sum(g_lp[l][c]);
      if (lane == 0) row[l * 3 + c] = s;
    }
    const float s = wave_sum(g_li[l]);
    if (lane == 0) row[3 * L + l] = s;
  }
}

// One workgroup per image: the fixed-order sum of its wavefronts' rows -> dlpos [B,L,3], dlint [B,L].
constexpr int kLightSumThreads = 1024;
__global__ __launch_bounds__(kLightSumThreads) void k_soft_light_sum(const float *__restrict__ rows, int per_image, int L,
                                                                     float *__restrict__ dlpos, float *__restrict__ dlint) {
  constexpr int kParts = kLightSumThreads / kLightRow;
  __shared__ float s_part[kParts][kLightRow + 1];
  const int img = (int)blockIdx.x, slot = (int)threadIdx.x % kLightRow, part = (int)threadIdx.x / kLightRow;
  const float *mine = rows + (size_t)img * per_image * kLightRow;
  float v = 0.0f;
  if (slot < 4 * L) {
    constexpr int kInFlight = 8;
    for (int i = part; i < per_image; i += kParts * kInFlight) {
      float x[kInFlight];
#pragma unroll
      for (int u = 0; u < kInFlight; ++u) {
        const int k = i + u * kParts;
        x[u] = k < per_image ? mine[(size_t)k * kLightRow + slot] : 0.0f;
      }
#pragma unroll
      for (int u = 0; u < kInFlight; ++u) v += x[u];
    }
  }
  s_part[part][slot] = v;
  __syncthreads();
  if (part == 0 && slot < 4 * L) {
    float t = 0.0f;
    for (int k = 0; k < kParts; ++k) t += s_part[k][slot];
    if (slot < 3 * L) dlpos[(size_t)img * 3 * L + slot] = t;
    else dlint[(size_t)img * L + (slot - 3 * L)] = t;
  }
}

// DET: the int64 sums back to float, array by array (dclip [B,V,4] | dnormals | dpositions | ddiffuse [B,V,3])
__global__ __launch_bounds__(kThreads) void k_soft_from_fixed(const long long *__restrict__ fixed,
                                                             const float *__restrict__ det_scale, size_t bv,
                                                             float *__restrict__ dclip, float *__restrict__ dnormals,
                                                             float *__restrict__ dpositions, float *__restrict__ ddiffuse) {
  const size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x;
  if (i >= bv * 13) return;
  const float v = *det_overflow_flag(det_scale) ? __int_as_float(0x7fc00000) : (float)fixed[i] * det_scale[1];
  if (i < bv * 4) dclip[i] = v;
  else if (i < bv * 7) dnormals[i - bv * 4] = v;
  else if (i < bv * 10) dpositions[i - bv * 7] = v;
  else ddiffuse[i - bv * 10] = v;
}
inline size_t soft_fixed_bytes(int B, int V) { return align_up((size_t)B * V * 13 * sizeof(long long), 256); }

inline size_t soft_rec_bytes(int B, int T) { return align_up((size_t)B * T * sizeof(SoftRec), 256); }

struct TileGrid {
  int tiles_x, per_image, n_tiles, per_xcd;
};
inline TileGrid tile_grid(int B, int W, int H) {
  TileGrid g;
  g.tiles_x = (W + kTile - 1) / kTile;
  g.per_image = g.tiles_x * ((H + kTile - 1) / kTile);
  g.n_tiles = g.per_image * B;
  g.per_xcd = (g.n_tiles + kXcds - 1) / kXcds;
  return g;
}

}  // namespace

int soft_max_lights() { return kMaxLights; }

struct CellGrid {
  int cells_x, per_image;
};
inline CellGrid cell_grid(int W, int H) {
  const int span = kCellTiles * kTile;
  CellGrid c;
  c.cells_x = (W + span - 1) / span;
  c.per_image = c.cells_x * ((H + span - 1) / span);
  return c;
}
inline size_t cell_ids_bytes(int B, int T, int W, int H) {
  return align_up((size_t)B * cell_grid(W, H).per_image * T * sizeof(int32_t), 256);
}
inline size_t cell_count_bytes(int B, int W, int H) {
  return align_up((size_t)B * cell_grid(W, H).per_image * sizeof(int32_t), 256);
}

static size_t soft_light_rows_bytes(int B, int W, int H) {
  return align_up((size_t)tile_grid(B, W, H).n_tiles * (kThreads / 64) * kLightRow * sizeof(float), 256);
}

size_t soft_ws(int B, int V, int T, int W, int H) {
  return soft_rec_bytes(B, T) + align_up((size_t)B * T * sizeof(CornerRec), 256) + cell_ids_bytes(B, T, W, H) +
         cell_count_bytes(B, W, H) + soft_light_rows_bytes(B, W, H) + soft_fixed_bytes(B, V) + kDetBlockBytes;
}

// the head of the workspace: what soft_prepare() leaves there and both passes read
size_t soft_prepared_bytes(int B, int V, int T, int W, int H) {
  (void)V;
  return soft_rec_bytes(B, T) + align_up((size_t)B * T * sizeof(CornerRec), 256) + cell_ids_bytes(B, T, W, H) +
         cell_count_bytes(B, W, H);
}

static void soft_prepared_layout(void *head, int B, int T, int W, int H, SoftRec *&recs, CornerRec *&corners,
                                 int32_t *&cell_ids, int32_t *&cell_count) {
  char *p = (char *)head;
  recs = (SoftRec *)p;
  p += soft_rec_bytes(B, T);
  corners = (CornerRec *)p;
  p += align_up((size_t)B * T * sizeof(CornerRec), 256);
  cell_ids = (int32_t *)p;
  p += cell_ids_bytes(B, T, W, H);
  cell_count = (int32_t *)p;
}

// records, corner attributes and the coarse cell lists: the part the forward and the backward share
static int soft_prepare(const float *clip, const float *positions, const float *normals, const float *diffuse,
                        const int32_t *tris, int B, int V, int T, int W, int H, float blur, void *ws,
                        SoftRec *&recs, CornerRec *&corners, int32_t *&cell_ids, int32_t *&cell_count,
                        hipStream_t s) {
  soft_prepared_layout(ws, B, T, W, H, recs, corners, cell_ids, cell_count);
  const CellGrid cg = cell_grid(W, H);
  const long nbt = (long)B * T;
  if (nbt > 0) {
    hipLaunchKernelGGL(k_soft_setup, dim3((unsigned)((nbt + kThreads - 1) / kThreads)), dim3(kThreads), 0, s,
                       (const float4 *)clip, tris, B, V, T, blur, recs);
    int rc = check_launch();
    if (rc != MR_OK) return rc;
    rc = launch_corner_setup(normals, positions, diffuse, tris, B, V, T, corners, s);
    if (rc != MR_OK) return rc;
  }
  hipLaunchKernelGGL(k_soft_coarse, dim3((unsigned)(cg.per_image * B)), dim3(kCoarseThreads), 0, s, recs, T, W, H,
                     cg.cells_x, cg.per_image, cell_ids, cell_count);
  return check_launch();
}

int launch_debug_soft_nearest(const float *p, const float *a, const float *b, int n, float *out, hipStream_t s) {
  if (n <= 0) return MR_OK;
  hipLaunchKernelGGL(k_debug_soft_nearest, dim3((unsigned)((n + kThreads - 1) / kThreads)), dim3(kThreads), 0, s,
                     (const float2 *)p, (const float2 *)a, (const float2 *)b, n, (float4 *)out);
  return check_launch();
}

int launch_soft_forward(const float *clip, const float *positions, const float *normals,
                        const float *diffuse, const int32_t *tris, const float *lpos,
                        const float *lint, int B, int V, int T, int W, int H, int L, float sigma,
                        float gamma, float blur, float *rgba, float *aux, void *ws, hipStream_t s) {
  if ((size_t)B * W * H == 0) return MR_OK;
  SoftRec *recs;
  CornerRec *corners;
  int32_t *cell_ids, *cell_count;
  const int rc0 = soft_prepare(clip, positions, normals, diffuse, tris, B, V, T, W, H, blur, ws, recs, corners,
                               cell_ids, cell_count, s);
  if (rc0 != MR_OK) return rc0;
  const TileGrid tg = tile_grid(B, W, H);
  const SoftParams pr = soft_params(sigma, gamma, blur);
  if (L == 1)
    hipLaunchKernelGGL(k_soft_forward<1>, dim3((unsigned)(tg.per_xcd * kXcds)), dim3(kThreads), 0, s, recs,
                       corners, lpos, lint, T, W, H, L, pr, tg.tiles_x, tg.per_image, tg.n_tiles, tg.per_xcd,
                       cell_ids, cell_count, cell_grid(W, H).per_image, (float4 *)rgba, (float4 *)aux);
  else
    hipLaunchKernelGGL(k_soft_forward<kMaxLights>, dim3((unsigned)(tg.per_xcd * kXcds)), dim3(kThreads), 0, s, recs,
                       corners, lpos, lint, T, W, H, L, pr, tg.tiles_x, tg.per_image, tg.n_tiles, tg.per_xcd,
                       cell_ids, cell_count, cell_grid(W, H).per_image, (float4 *)rgba, (float4 *)aux);
  return check_launch();
}

int launch_soft_backward(const float *drgba, const float *rgba, const float *aux, const float *clip,
                         const float *positions, const float *normals, const float *diffuse,
                         const int32_t *tris, const float *lpos, const float *lint, int B, int V, int T,
                         int W, int H, int L, float sigma, float gamma, float blur, float *dclip,
                         float *dpositions, float *dnormals, float *ddiffuse, float *dlpos, float *dlint,
                         const void *prepared, void *ws, hipStream_t s) {
  if (B == 0) return MR_OK;
  const size_t v3 = (size_t)B * V * 3 * sizeof(float), v4 = (size_t)B * V * 4 * sizeof(float);
  const size_t l3 = (size_t)B * L * 3 * sizeof(float), l1 = (size_t)B * L * sizeof(float);
  // A caller that lays the six outputs out back to back in this order (_native.py does) gets ONE memset
  // instead of six launch-bound ones (~5 us each: 30 us of a 1.1 ms step).
  if ((char *)dpositions == (char *)dclip + v4 && (char *)dnormals == (char *)dpositions + v3 &&
      (char *)ddiffuse == (char *)dnormals + v3 && (char *)dlpos == (char *)ddiffuse + v3 &&
      (char *)dlint == (char *)dlpos + l3) {
    if (zero_async(dclip, v4 + 3 * v3 + l3 + l1, s) != hipSuccess) return check_launch();
  } else {
    if (V > 0) {
      if (zero_async(dclip, v4, s) != hipSuccess) return check_launch();
      if (zero_async(dpositions, v3, s) != hipSuccess) return check_launch();
      if (zero_async(dnormals, v3, s) != hipSuccess) return check_launch();
      if (zero_async(ddiffuse, v3, s) != hipSuccess) return check_launch();
    }
    if (zero_async(dlpos, l3, s) != hipSuccess) return check_launch();
    if (zero_async(dlint, l1, s) != hipSuccess) return check_launch();
  }
  if (T == 0 || V == 0 || (size_t)W * H == 0) return MR_OK;
  SoftRec *recs;
  CornerRec *corners;
  int32_t *cell_ids, *cell_count;
  if (prepared) {
    // the forward pass of the same inputs left these (mr_soft_prepared_bytes at the head of ITS workspace):
    // three launches (~30 us of a 1 ms step) are not repeated
    soft_prepared_layout(const_cast<void *>(prepared), B, T, W, H, recs, corners, cell_ids, cell_count);
  } else {
    const int rc = soft_prepare(clip, positions, normals, diffuse, tris, B, V, T, W, H, blur, ws, recs, corners,
                                cell_ids, cell_count, s);
    if (rc != MR_OK) return rc;
  }
  const TileGrid tg = tile_grid(B, W, H);
  const SoftParams pr = soft_params(sigma, gamma, blur);
  float *light_rows = (float *)((char *)ws + soft_prepared_bytes(B, V, T, W, H));
  long long *det_fixed = (long long *)((char *)light_rows + soft_light_rows_bytes(B, W, H));
  float *det_block = (float *)((char *)det_fixed + soft_fixed_bytes(B, V));
#define MR_SOFT_BWD(DET_, ML_)                                                                              \
  hipLaunchKernelGGL((k_soft_backward<DET_, ML_>), dim3((unsigned)(tg.per_xcd * kXcds)), dim3(kThreads), 0, s, \
                     recs, corners, lpos, lint, tris, V, T, W, H, L, pr, tg.tiles_x, tg.per_image, tg.n_tiles,  \
                     tg.per_xcd, cell_ids, cell_count, cell_grid(W, H).per_image, (const float4 *)drgba,       \
                     (const float4 *)rgba, (const float4 *)aux, dclip, dnormals, dpositions, ddiffuse,         \
                     light_rows, det_fixed, det_block, B)
  const bool det = g_deterministic != 0;
  if (det) {
    // the scale: a contribution carries up to 1 / sigma (or 1 / gamma) over the upstream gradient before the
    // geometry factors; the fixed point's 2^21 of headroom takes those
    if (zero_async(det_fixed, (size_t)B * V * 13 * sizeof(long long), s) != hipSuccess) return check_launch();
    const float gain = 1.0f / fminf(fminf(sigma, gamma), 1.0f);
    const int rcd = launch_det_scale(drgba, (size_t)B * H * W * 4, gain, det_block, s);
    if (rcd != MR_OK) return rcd;
    if (L == 1) MR_SOFT_BWD(true, 1);
    else MR_SOFT_BWD(true, kMaxLights);
  } else {
    if (L == 1) MR_SOFT_BWD(false, 1);
    else MR_SOFT_BWD(false, kMaxLights);
  }
#undef MR_SOFT_BWD
  const int rc2 = check_launch();
  if (rc2 != MR_OK) return rc2;
  if (det) {
    const size_t n = (size_t)B * V * 13;
    hipLaunchKernelGGL(k_soft_from_fixed, dim3((unsigned)((n + kThreads - 1) / kThreads)), dim3(kThreads), 0, s,
                       det_fixed, det_block, (size_t)B * V, dclip, dnormals, dpositions, ddiffuse);
    const int rc3 = check_launch();
    if (rc3 != MR_OK) return rc3;
  }
  hipLaunchKernelGGL(k_soft_light_sum, dim3((unsigned)B), dim3(kLightSumThreads), 0, s, light_rows,
                     tg.per_image * (kThreads / 64), L, dlpos, dlint);
  return check_launch();
}

}  // namespace mr
