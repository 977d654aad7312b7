// Vertex normals of a triangle mesh on gfx950: the scatter-add of
// compute_vertex_normals (reference: src/common/meshes.py:3-35) as a per-vertex GATHER.
//
// The reference adds, for every triangle and each of its corners o (with a, b the next two
// corners), the area-weighted face normal (a - o) x (b - o) to vertex o with index_add_ and
// normalises the sums (eps 1e-6).  Here one thread owns one (image, vertex): it walks the
// vertex's incident (triangle, corner) pairs in the CSR adjacency that the renderer already keeps
// per triangle array, evaluates the corner's own cross product (the reference's three expressions
// differ in rounding; each corner uses its own) and sums in adjacency order: no atomics, every
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

__global__ __launch_bounds__(kThreads) void k_vertex_normals(
    const V3 *__restrict__ vertices, const int32_t *__restrict__ tris, const int32_t *__restrict__ offsets,
    const int32_t *__restrict__ entries, int B, int V, V3 *__restrict__ sums, V3 *__restrict__ normals) {
  const long gid = (long)blockIdx.x * kThreads + threadIdx.x;
  if (gid >= (long)B * V) return;
  const int b = (int)(gid / V), v = (int)(gid - (long)b * V);
  const V3 *vb = vertices + (size_t)b * V;
  V3 s{0.f, 0.f, 0.f};
  const int e1 = offsets[v + 1];
  for (int i = offsets[v]; i < e1; ++i) {
    const int e = entries[i], t = e / 3, k = e - 3 * t;
    int idx[3];
    if (!load_triangle(tris, t, V, idx)) continue;  // the reference would index out of bounds
    const V3 o = vb[idx[k]], a = vb[idx[(k + 1) % 3]], c = vb[idx[(k + 2) % 3]];
    s = s + cross(a - o, c - o);  // meshes.py:24-33, corner k's expression
  }
  sums[gid] = s;
  const float inv = 1.0f / fmaxf(sqrtf(dot(s, s)), kNormalEps);
  normals[gid] = {s.x * inv, s.y * inv, s.z * inv};
}

// d normalize(s) / d s applied to dn:  (dn - n (n . dn)) / |s|   (|s| > eps), dn / eps otherwise.
__global__ __launch_bounds__(kThreads) void k_vertex_normals_dsums(const V3 *__restrict__ dnormals,
                                                                   const V3 *__restrict__ sums, long n,
                                                                   V3 *__restrict__ dsums) {
  const long gid = (long)blockIdx.x * kThreads + threadIdx.x;
  if (gid >= n) return;
  const V3 s = sums[gid], dn = dnormals[gid];
  const float len = sqrtf(dot(s, s));
  if (len > kNormalEps) {
    const float inv = 1.0f / len;
    const V3 nrm{s.x * inv, s.y * inv, s.z * inv};
    const float nd = dot(nrm, dn);
    dsums[gid] = {(dn.x - nrm.x * nd) * inv, (dn.y - nrm.y * nd) * inv, (dn.z - nrm.z * nd) * inv};
  } else {
    const float inv = 1.0f / kNormalEps;
    dsums[gid] = {dn.x * inv, dn.y * inv, dn.z * inv};
  }
}

// For c_j = (v_{j+1} - v_j) x (v_{j+2} - v_j) with upstream g_j (the gradient of the sum at v_j):
//   d/d v_{j+1} = (v_{j+2} - v_j) x g_j,   d/d v_{j+2} = g_j x (v_{j+1} - v_j),   d/d v_j = -(both).
// The vertex at corner k of a triangle collects its share of all three c_j.
__global__ __launch_bounds__(kThreads) void k_vertex_normals_backward(
    const V3 *__restrict__ vertices, const V3 *__restrict__ dsums, const int32_t *__restrict__ tris,
    const int32_t *__restrict__ offsets, const int32_t *__restrict__ entries, int B, int V,
    V3 *__restrict__ dvertices) {
  const long gid = (long)blockIdx.x * kThreads + threadIdx.x;
  if (gid >= (long)B * V) return;
  const int b = (int)(gid / V), v = (int)(gid - (long)b * V);
  const V3 *vb = vertices + (size_t)b * V;
  const V3 *gb = dsums + (size_t)b * V;
  V3 d{0.f, 0.f, 0.f};
  const int e1 = offsets[v + 1];
  for (int i = offsets[v]; i < e1; ++i) {
    const int e = entries[i], t = e / 3, k = e - 3 * t;
    int idx[3];
    if (!load_triangle(tris, t, V, idx)) continue;
    const int k1 = (k + 1) % 3, k2 = (k + 2) % 3;
    const V3 p0 = vb[idx[k]], p1 = vb[idx[k1]], p2 = vb[idx[k2]];
    const V3 g0 = gb[idx[k]], g1 = gb[idx[k1]], g2 = gb[idx[k2]];
    // j = k: this vertex is the cross product's origin
    const V3 da = cross(p2 - p0, g0), db = cross(g0, p1 - p0);
    d = d - (da + db);
    // j = k2 (its "next" corner is k): d/d v_{j+1} = (v_{j+2} - v_j) x g_j with j+2 = k1
    d = d + cross(p1 - p2, g2);
    // j = k1 (its "next-next" corner is k): d/d v_{j+2} = g_j x (v_{j+1} - v_j) with j+1 = k2
    d = d + cross(g1, p2 - p1);
  }
  dvertices[gid] = d;
}

}  // namespace

int launch_vertex_normals(const float *vertices, const int32_t *tris, const int32_t *offsets,
                          const int32_t *entries, int B, int V, float *sums, float *normals, hipStream_t s) {
  const long n = (long)B * V;
  if (n == 0) return MR_OK;
  hipLaunchKernelGGL(k_vertex_normals, dim3((unsigned)((n + kThreads - 1) / kThreads)), dim3(kThreads), 0, s,
                     (const V3 *)vertices, tris, offsets, entries, B, V, (V3 *)sums, (V3 *)normals);
  return check_launch();
}

int launch_vertex_normals_backward(const float *dnormals, const float *vertices, const float *sums,
                                   const int32_t *tris, const int32_t *offsets, const int32_t *entries,
                                   int B, int V, float *dsums, float *dvertices, hipStream_t s) {
  const long n = (long)B * V;
  if (n == 0) return MR_OK;
  const dim3 grid((unsigned)((n + kThreads - 1) / kThreads));
  hipLaunchKernelGGL(k_vertex_normals_dsums, grid, dim3(kThreads), 0, s, (const V3 *)dnormals, (const V3 *)sums, n,
                     (V3 *)dsums);
  int rc = check_launch();
  if (rc != MR_OK) return rc;
  hipLaunchKernelGGL(k_vertex_normals_backward, grid, dim3(kThreads), 0, s, (const V3 *)vertices,
                     (const V3 *)dsums, tris, offsets, entries, B, V, (V3 *)dvertices);
  return check_launch();
}

}  // namespace mr
