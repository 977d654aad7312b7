// Vertex normals of a triangle mesh on gfx950: the scatter-add of
// compute_vertex_normals (reference: src/common/meshes.py:3-35) as a per-vertex GATHER.
//
// The reference adds, for every triangle and each of its corners o (with a, b the next two
// corners), the area-weighted face normal (a - o) x (b - o) to vertex o with index_add_ and
// normalises the sums (eps 1e-6).  Here eight lanes own one (image, vertex): they walk the
// vertex's incident (triangle, corner) pairs in the CSR adjacency that the renderer already keeps
// per triangle array, evaluate each corner's own cross product (the reference's three expressions
// differ in rounding; each corner uses its own) and sum in a fixed order: no atomics, every
// output written once, bitwise reproducible.  The backward is the same walk with the three
// cross products' derivatives gathered at the vertex.
#include "mr_internal.h"

namespace mr {
namespace {

constexpr int kThreads = 256;
constexpr float kNormalEps = 1e-6f;  // torch.nn.functional.normalize(..., eps=1e-6), meshes.py:34

struct V3 {
  float x, y, z;
};
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ V3 cross(V3 a, V3 b) {
  return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
__device__ __forceinline__ float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }

__device__ __forceinline__ bool load_triangle(const int32_t *tris, int t, int V, int (&idx)[3]) {
  idx[0] = tris[3 * t];
  idx[1] = tris[3 * t + 1];
  idx[2] = tris[3 * t + 2];
  return (unsigned)idx[0] < (unsigned)V && (unsigned)idx[1] < (unsigned)V && (unsigned)idx[2] < (unsigned)V;
}

// Eight lanes per (image, vertex), one incident (triangle, corner) pair each (a second trip for valence
// > 8): the walk is three dependent loads per pair (entry -> triangle -> corners), and with one thread per
// vertex those chains ran one after the other at under one wavefront per SIMD -- 21 + 5 + 29 us for 41k
// vertices (SoftRas config 5), all of it latency.  The partial sums meet in a fixed butterfly over the
// eight lanes: still no atomics, every output written once, bitwise reproducible.
constexpr int kLanesPerVertex = 8;

__device__ __forceinline__ V3 sum_over_vertex_lanes(V3 s) {
#pragma unroll
  for (int m = 1; m < kLanesPerVertex; m <<= 1) {
    s.x += __shfl_xor(s.x, m, kLanesPerVertex);
    s.y += __shfl_xor(s.y, m, kLanesPerVertex);
    s.z += __shfl_xor(s.z, m, kLanesPerVertex);
  }
  return s;
}

__global__ __launch_bounds__(kThreads) void k_vertex_normals(
    const V3 *__restrict__ vertices, const int32_t *__restrict__ tris, const int32_t *__restrict__ offsets,
    const int32_t *__restrict__ entries, int B, int V, V3 *__restrict__ sums, V3 *__restrict__ normals) {
  const long tid = (long)blockIdx.x * kThreads + threadIdx.x;
  const long gid = tid / kLanesPerVertex;
  const int sub = (int)(tid % kLanesPerVertex);
  const bool have = gid < (long)B * V;   // (whole groups of eight: the butterfly below stays inside one)
  const long g = have ? gid : 0;
  const int b = (int)(g / V), v = (int)(g - (long)b * V);
  const V3 *vb = vertices + (size_t)b * V;
  V3 s{0.f, 0.f, 0.f};
  const int e1 = have ? offsets[v + 1] : 0;
  for (int i = offsets[v] + sub; i < e1; i += kLanesPerVertex) {
    const int e = entries[i], t = e / 3, k = e - 3 * t;
    int idx[3];
    if (!load_triangle(tris, t, V, idx)) continue;  // the reference would index out of bounds
    const V3 o = vb[idx[k]], a = vb[idx[(k + 1) % 3]], c = vb[idx[(k + 2) % 3]];
    s = s + cross(a - o, c - o);  // meshes.py:24-33, corner k's expression
  }
  s = sum_over_vertex_lanes(s);
  if (!have || sub != 0) return;
  sums[gid] = s;
  const float inv = 1.0f / fmaxf(sqrtf(dot(s, s)), kNormalEps);
  normals[gid] = {s.x * inv, s.y * inv, s.z * inv};
}

// d normalize(s) / d s applied to dn:  (dn - n (n . dn)) / |s|   (|s| > eps), dn / eps otherwise.
__device__ __forceinline__ V3 normalize_backward(V3 s, V3 dn) {
  const float len = sqrtf(dot(s, s));
  if (len > kNormalEps) {
    const float inv = 1.0f / len;
    const V3 nrm{s.x * inv, s.y * inv, s.z * inv};
    const float nd = dot(nrm, dn);
    return {(dn.x - nrm.x * nd) * inv, (dn.y - nrm.y * nd) * inv, (dn.z - nrm.z * nd) * inv};
  }
  const float inv = 1.0f / kNormalEps;
  return {dn.x * inv, dn.y * inv, dn.z * inv};
}

// For c_j = (v_{j+1} - v_j) x (v_{j+2} - v_j) with upstream g_j (the gradient of the sum at v_j):
//   d/d v_{j+1} = (v_{j+2} - v_j) x g_j,   d/d v_{j+2} = g_j x (v_{j+1} - v_j),   d/d v_j = -(both).
// The vertex at corner k of a triangle collects its share of all three c_j.  g_j is formed on the spot
// from the corner's dnormal and sum (a launch of its own before: 5 us for 40 flops per vertex).
__global__ __launch_bounds__(kThreads) void k_vertex_normals_backward(
    const V3 *__restrict__ vertices, const V3 *__restrict__ dnormals, const V3 *__restrict__ sums,
    const int32_t *__restrict__ tris, const int32_t *__restrict__ offsets, const int32_t *__restrict__ entries,
    int B, int V, V3 *__restrict__ dvertices) {
  const long tid = (long)blockIdx.x * kThreads + threadIdx.x;
  const long gid = tid / kLanesPerVertex;
  const int sub = (int)(tid % kLanesPerVertex);
  const bool have = gid < (long)B * V;
  const long g = have ? gid : 0;
  const int b = (int)(g / V), v = (int)(g - (long)b * V);
  const V3 *vb = vertices + (size_t)b * V;
  const V3 *nb = dnormals + (size_t)b * V;
  const V3 *sb = sums + (size_t)b * V;
  V3 d{0.f, 0.f, 0.f};
  const int e1 = have ? offsets[v + 1] : 0;
  for (int i = offsets[v] + sub; i < e1; i += kLanesPerVertex) {
    const int e = entries[i], t = e / 3, k = e - 3 * t;
    int idx[3];
    if (!load_triangle(tris, t, V, idx)) continue;
    const int k1 = (k + 1) % 3, k2 = (k + 2) % 3;
    const V3 p0 = vb[idx[k]], p1 = vb[idx[k1]], p2 = vb[idx[k2]];
    const V3 g0 = normalize_backward(sb[idx[k]], nb[idx[k]]);
    const V3 g1 = normalize_backward(sb[idx[k1]], nb[idx[k1]]);
    const V3 g2 = normalize_backward(sb[idx[k2]], nb[idx[k2]]);
    // j = k: this vertex is the cross product's origin
    const V3 da = cross(p2 - p0, g0), db = cross(g0, p1 - p0);
    d = d - (da + db);
    // j = k2 (its "next" corner is k): d/d v_{j+1} = (v_{j+2} - v_j) x g_j with j+2 = k1
    d = d + cross(p1 - p2, g2);
    // j = k1 (its "next-next" corner is k): d/d v_{j+2} = g_j x (v_{j+1} - v_j) with j+1 = k2
    d = d + cross(g1, p2 - p1);
  }
  d = sum_over_vertex_lanes(d);
  if (have && sub == 0) dvertices[gid] = d;
}

}  // namespace

int launch_vertex_normals(const float *vertices, const int32_t *tris, const int32_t *offsets,
                          const int32_t *entries, int B, int V, float *sums, float *normals, hipStream_t s) {
  const long n = (long)B * V;
  if (n == 0) return MR_OK;
  hipLaunchKernelGGL(k_vertex_normals, dim3((unsigned)((n * kLanesPerVertex + kThreads - 1) / kThreads)), dim3(kThreads), 0, s,
                     (const V3 *)vertices, tris, offsets, entries, B, V, (V3 *)sums, (V3 *)normals);
  return check_launch();
}

int launch_vertex_normals_backward(const float *dnormals, const float *vertices, const float *sums,
                                   const int32_t *tris, const int32_t *offsets, const int32_t *entries,
                                   int B, int V, float *dvertices, hipStream_t s) {
  const long n = (long)B * V;
  if (n == 0) return MR_OK;
  const dim3 grid((unsigned)((n * kLanesPerVertex + kThreads - 1) / kThreads));
  hipLaunchKernelGGL(k_vertex_normals_backward, grid, dim3(kThreads), 0, s, (const V3 *)vertices,
                     (const V3 *)dnormals, (const V3 *)sums, tris, offsets, entries, B, V, (V3 *)dvertices);
  return check_launch();
}

// ---- clip-space transforms: perspective . look_at (round 3) ----------------------------------
// Replaces, for cameras that live on the device (the optimisation examples optimise them there:
// src/examples/example4.py:54-58), look_at (src/common/camera_utils.py:45-96), perspective
// (:99-139) and their product -- ~30 tiny launches and as many autograd nodes per step, +0.45 ms on a
// 1.2 ms SoftRas step -- by ONE launch each way: a thread per image builds the 4x4, the backward is
// derived by hand to eye, center and up (the reference's look_at calls numpy on the eye and is not
// differentiable w.r.t. it at all; fov / near / far stay non-differentiable here -- a request for
// their gradient takes the torch expression).
namespace {

struct Camera {  // what both directions recompute
  V3 f, s, u;           // unit forward, unit side, camera up
  float f_len, s_len;   // |center - eye|, |f x up|
  float p00, p11, p22, p23;
};

__device__ __forceinline__ Camera camera_frame(V3 eye, V3 center, V3 up, float fov_y, float near_clip, float far_clip,
                                               float aspect) {
  Camera c;
  const V3 fr = center - eye;
  c.f_len = sqrtf(dot(fr, fr));
  const float fi = 1.0f / c.f_len;
  c.f = {fr.x * fi, fr.y * fi, fr.z * fi};
  const V3 sr = cross(c.f, up);
  c.s_len = sqrtf(dot(sr, sr));
  const float si = 1.0f / c.s_len;
  c.s = {sr.x * si, sr.y * si, sr.z * si};
  c.u = cross(c.s, c.f);
  const float focal = 1.0f / tanf(fov_y * (3.14159265358979323846f / 360.0f));  // camera_utils.py:127
  const float range = far_clip - near_clip;
  c.p00 = focal / aspect;
  c.p11 = focal;
  c.p22 = -(far_clip + near_clip) / range;
  c.p23 = -2.0f * (far_clip * near_clip / range);
  return c;
}

constexpr float kCameraDegenerate = 1e-6f;  // camera_utils.py:7

__global__ __launch_bounds__(kThreads) void k_camera_transforms(
    const V3 *__restrict__ eye, const V3 *__restrict__ center, const V3 *__restrict__ up,
    const float *__restrict__ fov_y, const float *__restrict__ near_clip, const float *__restrict__ far_clip,
    float aspect, int B, float *__restrict__ transforms, int *__restrict__ degenerate) {
  const int b = (int)(blockIdx.x * kThreads + threadIdx.x);
  if (b >= B) return;
  const V3 e = eye[b];
  const Camera c = camera_frame(e, center[b], up[b], fov_y[b], near_clip[b], far_clip[b], aspect);
  // the reference's two assertions (camera_utils.py:68-69, 74-76): bit 0 eye ~ center, bit 1 up ~ gaze
  const int flags = (!(c.f_len > kCameraDegenerate) ? 1 : 0) | (!(c.s_len > kCameraDegenerate) ? 2 : 0);
  if (flags) atomicOr(degenerate, flags);
  // view = R T: rows (s, -s.e), (u, -u.e), (-f, f.e), (0 0 0 1); clip = P view
  const float v0[4] = {c.s.x, c.s.y, c.s.z, -dot(c.s, e)};
  const float v1[4] = {c.u.x, c.u.y, c.u.z, -dot(c.u, e)};
  const float v2[4] = {-c.f.x, -c.f.y, -c.f.z, dot(c.f, e)};
  float *m = transforms + (size_t)b * 16;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    m[j] = c.p00 * v0[j];
    m[4 + j] = c.p11 * v1[j];
    m[8 + j] = c.p22 * v2[j] + (j == 3 ? c.p23 : 0.0f);
    m[12 + j] = -v2[j];
  }
}

__global__ __launch_bounds__(kThreads) void k_camera_transforms_backward(
    const float *__restrict__ dtransforms, const V3 *__restrict__ eye, const V3 *__restrict__ center,
    const V3 *__restrict__ up, const float *__restrict__ fov_y, const float *__restrict__ near_clip,
    const float *__restrict__ far_clip, float aspect, int B, V3 *__restrict__ deye, V3 *__restrict__ dcenter,
    V3 *__restrict__ dup) {
  const int b = (int)(blockIdx.x * kThreads + threadIdx.x);
  if (b >= B) return;
  const V3 e = eye[b], upv = up[b];
  const Camera c = camera_frame(e, center[b], upv, fov_y[b], near_clip[b], far_clip[b], aspect);
  const float *g = dtransforms + (size_t)b * 16;
  // d view = P^T d clip (the view's last row is constant)
  float dv[3][4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    dv[0][j] = c.p00 * g[j];
    dv[1][j] = c.p11 * g[4 + j];
    dv[2][j] = c.p22 * g[8 + j] - g[12 + j];
  }
  auto axpy = [](V3 a, float k, V3 x) { return V3{a.x + k * x.x, a.y + k * x.y, a.z + k * x.z}; };
  // rows (s, -s.e), (u, -u.e), (-f, f.e)
  V3 gs = axpy(V3{dv[0][0], dv[0][1], dv[0][2]}, -dv[0][3], e);
  V3 gu = axpy(V3{dv[1][0], dv[1][1], dv[1][2]}, -dv[1][3], e);
  V3 gf = axpy(V3{-dv[2][0], -dv[2][1], -dv[2][2]}, dv[2][3], e);
  V3 ge = axpy(axpy(axpy(V3{0.f, 0.f, 0.f}, -dv[0][3], c.s), -dv[1][3], c.u), dv[2][3], c.f);
  // u = s x f
  gs = gs + cross(c.f, gu);
  gf = gf + cross(gu, c.s);
  // s = s_raw / |s_raw|, s_raw = f x up
  const V3 gsr = axpy(gs, -dot(c.s, gs), c.s);
  const float si = 1.0f / c.s_len;
  const V3 gs_raw = {gsr.x * si, gsr.y * si, gsr.z * si};
  gf = gf + cross(upv, gs_raw);
  const V3 gup = cross(gs_raw, c.f);
  // f = f_raw / |f_raw|, f_raw = center - eye
  const V3 gfr = axpy(gf, -dot(c.f, gf), c.f);
  const float fi = 1.0f / c.f_len;
  const V3 gf_raw = {gfr.x * fi, gfr.y * fi, gfr.z * fi};
  deye[b] = ge - gf_raw;
  dcenter[b] = gf_raw;
  dup[b] = gup;
}

}  // namespace

int launch_camera_transforms(const float *eye, const float *center, const float *up, const float *fov_y,
                             const float *near_clip, const float *far_clip, float aspect, int B, float *transforms,
                             int *degenerate, hipStream_t s) {
  if (B == 0) return MR_OK;
  if (zero_async(degenerate, sizeof(int), s) != hipSuccess) return check_launch();
  hipLaunchKernelGGL(k_camera_transforms, dim3((unsigned)((B + kThreads - 1) / kThreads)), dim3(kThreads), 0, s,
                     (const V3 *)eye, (const V3 *)center, (const V3 *)up, fov_y, near_clip, far_clip, aspect, B,
                     transforms, degenerate);
  return check_launch();
}

int launch_camera_transforms_backward(const float *dtransforms, const float *eye, const float *center,
                                      const float *up, const float *fov_y, const float *near_clip,
                                      const float *far_clip, float aspect, int B, float *deye, float *dcenter,
                                      float *dup, hipStream_t s) {
  if (B == 0) return MR_OK;
  hipLaunchKernelGGL(k_camera_transforms_backward, dim3((unsigned)((B + kThreads - 1) / kThreads)), dim3(kThreads),
                     0, s, dtransforms, (const V3 *)eye, (const V3 *)center, (const V3 *)up, fov_y, near_clip,
                     far_clip, aspect, B, (V3 *)deye, (V3 *)dcenter, (V3 *)dup);
  return check_launch();
}

}  // namespace mr
