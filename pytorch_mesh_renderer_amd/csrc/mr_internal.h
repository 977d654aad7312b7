// Shared device helpers and launch plumbing for the gfx950 rasterizer kernels.
// Everything here is CDNA4-only: wave64, __builtin_amdgcn_* intrinsics.
#pragma once

#include <hip/hip_runtime.h>
#include <limits.h>
#include <stdint.h>

#include "mesh_raster.h"

namespace mr {

constexpr int kWave = 64;       // gfx950 wavefront
constexpr int kXcds = 8;        // MI355X: 8 XCDs, each with a private L2

// Per-(image, triangle) setup record produced by k_setup and consumed by the
// raster kernel through wave-uniform (scalar) loads.  64 bytes, 64-B aligned.
//   a = (m0 m1 m2 m3)  b = (m4 m5 m6 m7)  c = (m8 z0 z1 z2)  d = (w0 w1 w2 -)
// m[] is the sign-corrected adjugate of [[x],[y],[w]] (rows = edge functions).
struct alignas(64) TriRec {
  float4 a, b, c, d;
};

// Per-(image, triangle) bbox record, 16 bytes: pixel bbox [l, r) x [bot, top) packed as 4 x u16
// (an empty / culled triangle is all zeros so that no region ever selects it) and a lower bound
// of the depth values the triangle can produce (-inf when none is proven: raster_forward.hip).
struct alignas(16) TriBox {
  unsigned lr, bt;
  float zlo;
  unsigned pad;
};

__device__ __forceinline__ uint2 pack_bbox(int l, int r, int bot, int top) {
  return make_uint2((unsigned)l | ((unsigned)r << 16), (unsigned)bot | ((unsigned)top << 16));
}

__device__ __forceinline__ int lane_id() {
  return (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
}

__device__ __forceinline__ float readlane_f(float v, int lane) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}

// Remap the hardware block id so that each XCD (blocks b, b+8, b+16, ... land
// on the same XCD under round-robin dispatch) owns one CONTIGUOUS range of the
// logical work list.  Placement only affects speed (which L2 holds which
// image's triangle records), never correctness.  Returns -1 for padding blocks.
__device__ __forceinline__ int xcd_contiguous_block(int hw_block, int n_logical, int per_xcd) {
  const int logical = (hw_block % kXcds) * per_xcd + hw_block / kXcds;
  return logical < n_logical ? logical : -1;
}

// ---- host side -------------------------------------------------------------
extern thread_local int g_last_hip_error;

inline int check_launch() {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    g_last_hip_error = (int)e;
    return MR_ELAUNCH;
  }
  return MR_OK;
}

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// Zero-fill as a KERNEL, never hipMemsetAsync.  Round 5: a memset NODE of a captured HIP graph replayed after eager work
// had run on the same device wrote another value than the one captured (ROCm 7.2: the 8-byte empty-block map came out as
// 0xC6 bytes on the second replay, tests/test_render_gpu.py::test_capture_step_replays_the_references_loop); kernel
// nodes carry their arguments by value.  Any alignment, any size; 16-byte stores over the aligned body.
static __global__ __launch_bounds__(256) void k_zero_bytes(unsigned char *__restrict__ p, size_t head, size_t n16, size_t tail) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  uint4 *body = (uint4 *)(p + head);
  for (size_t k = i; k < n16; k += (size_t)gridDim.x * 256) body[k] = make_uint4(0u, 0u, 0u, 0u);
  if (i < head) p[i] = 0;
  if (i < tail) p[head + n16 * 16 + i] = 0;
}
inline hipError_t zero_async(void *ptr, size_t bytes, hipStream_t s) {
  if (bytes == 0 || ptr == nullptr) return hipSuccess;
  unsigned char *p = (unsigned char *)ptr;
  size_t head = (size_t)((16 - ((uintptr_t)p & 15)) & 15);
  if (head > bytes) head = bytes;
  const size_t n16 = (bytes - head) / 16, tail = bytes - head - n16 * 16;
  size_t blocks = (n16 + 255) / 256;
  if (blocks < 1) blocks = 1;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(k_zero_bytes, dim3((unsigned)blocks), dim3(256), 0, s, p, head, n16, tail);
  return hipPeekAtLastError();
}

// mr_debug_last_accumulate_kernel (mesh_raster_debug.h): the functor of the most recent per-triangle accumulation
// pass launched by ANY thread of the process (autograd launches backward passes on a thread of its own), as the
// compiler spells it -- e.g. "... [Fn = mr::ShadeFoldLaneFn<1, true>]".  The parity tests read it to make sure the
// specialised kernel they mean to pin to the reference is the one that ran.  One relaxed pointer store per launch.
extern const char *volatile g_last_accumulate_kernel;

// mr_time_next_kernel (mesh_raster.h): per-thread, one-shot pairs of HIP events recorded on the
// launch stream immediately around one named kernel.  A caller arms a pair, the next launch of
// that kernel ON THE ARMING THREAD consumes it; nothing is recorded otherwise.
struct KernelTimerSlot {
  hipEvent_t start = nullptr, stop = nullptr;
};
extern thread_local KernelTimerSlot g_kernel_timers[MR_TIMER_COUNT];

class KernelTimer {
 public:
  KernelTimer(int which, hipStream_t s) : stream_(s) {
    KernelTimerSlot &slot = g_kernel_timers[which];
    stop_ = slot.stop;
    if (slot.start) (void)hipEventRecord(slot.start, s);
    slot = KernelTimerSlot{};
  }
  ~KernelTimer() {
    if (stop_) (void)hipEventRecord(stop_, stream_);
  }
  KernelTimer(const KernelTimer &) = delete;
  KernelTimer &operator=(const KernelTimer &) = delete;

 private:
  hipStream_t stream_;
  hipEvent_t stop_;
};

// Host launchers (one per .hip file)
int launch_raster_forward(const float *clip, const int32_t *tris, int B, int V, int T, int W,
                          int H, int32_t *ids, float *bary, float *z, void *ws, hipStream_t s);
size_t raster_forward_ws(int B, int V, int T, int W, int H);
int launch_raster_backward(const float *dbary, const float *clip, const int32_t *tris,
                           const int32_t *ids, const float *bary, int B, int V, int T, int W,
                           int H, float *dclip, void *ws, hipStream_t s);
size_t raster_backward_ws(int B, int V, int T, int W, int H);
int launch_interp_forward(const int32_t *ids, const float *bary, const float *attrs,
                          const int32_t *tris, const float *bg, int B, int V, int T, int W,
                          int H, int A, float *out, hipStream_t s);
int launch_interp_backward(const float *dout, const int32_t *ids, const float *bary,
                           const float *attrs, const int32_t *tris, const float *bg, int B,
                           int V, int T, int W, int H, int A, float *dattrs, float *dbary,
                           void *ws, hipStream_t s);
size_t interp_backward_ws(int B, int V, int T, int W, int H, int A);
int shade_max_lights();
int shade_light_gradient_max_lights();
int launch_shade_forward(const int32_t *ids, const float *bary, const float *normals,
                         const float *positions, const float *diffuse, const int32_t *tris,
                         const float *light_pos, const float *light_col, const float *ambient,
                         int B, int V, int T, int W, int H, int L, float *rgba, void *ws,
                         hipStream_t s);
size_t shade_forward_ws(int B, int V, int T, int W, int H);
size_t shade_backward_ws(int B, int V, int T, int W, int H);
int launch_shade_backward(const float *drgba, const uint8_t *signs, const float *sign_upstream,
                          const int32_t *ids, const float *bary,
                          const float *clip, const float *normals, const float *positions,
                          const float *diffuse, const int32_t *tris, const float *light_pos,
                          const float *light_col, const float *ambient, int B, int V, int T, int W,
                          int H, int L, float *dclip, float *dnormals, float *dpositions,
                          float *ddiffuse, float *light_grads, const void *corner_records,
                          const int32_t *vertex_offsets, const int32_t *vertex_entries,
                          const float *transforms, int gbuffer_flags, void *prepared, const uint8_t *empty_regions, void *ws,
                          hipStream_t s);
size_t shade_backward_prepared_bytes(int B, int T);

int interp_raster_max_attrs();
size_t interp_raster_backward_ws(int B, int V, int T, int W, int H, int A);
size_t interp_records_bytes(int B, int T, int A);
int launch_interp_forward_records(const int32_t *ids, const float *bary, const float *attrs, const int32_t *tris,
                                  const float *bg, int B, int V, int T, int W, int H, int A, float *out,
                                  void *records, hipStream_t s);
int launch_interp_raster_backward(const float *dout, const int32_t *ids, const float *bary, const float *clip,
                                  const float *attrs, const int32_t *tris, const float *bg,
                                  const int32_t *offsets, const int32_t *entries, const void *corner_records,
                                  int B, int V, int T, int W, int H, int A, float *dattrs, float *dclip,
                                  int gbuffer_flags, void *ws, hipStream_t s);
int launch_vertex_transform(const float *vertices, const float *transforms, int B, int V, float *clip,
                            hipStream_t s);
int launch_attr_records(const float *attrs, const int32_t *tris, int B, int V, int T, int A, void *records, hipStream_t s);
int launch_rasterize_interpolate_forward(const float *clip, const float *attrs, const int32_t *tris, const float *background,
                                         int B, int V, int T, int W, int H, int A, int32_t *ids, float *bary, float *z,
                                         float *out, void *records, void *ws, hipStream_t s);
int launch_render_forward(const float *vertices, const float *transforms, const float *normals,
                          const float *diffuse, const int32_t *tris, const float *light_pos,
                          const float *light_col, const float *ambient, int B, int V, int T, int W, int H,
                          int L, float *clip, int32_t *ids, float *bary, float *z, int want_z, float *rgba,
                          uint8_t *rgba_u8, void *corner_records, void *backward_prepared, uint8_t *empty_regions, void *ws,
                          hipStream_t s);
// the G-buffer and the specular term's across-pixels norms (norms2 [B,L], L <= 4) in one pass (raster_forward.hip)
size_t rasterize_specular_norms_ws(int B, int V, int T, int W, int H);
int launch_rasterize_specular_norms(const float *clip, const int32_t *tris, const float *normals, const float *positions,
                                    const float *light_pos, const float *camera, int B, int V, int T, int W, int H, int L,
                                    int32_t *ids, float *bary, float *z, int want_z, float *norms2, void *ws, hipStream_t s);
size_t shade_specular_forward_ws(int B, int V, int T, int W, int H);
int launch_shade_specular_forward(const int32_t *ids, const float *bary, const float *normals,
                                  const float *positions, const float *diffuse, const float *specular,
                                  const int32_t *tris, const float *light_pos, const float *light_col,
                                  const float *ambient, const float *camera, const float *shininess,
                                  int shininess_per_vertex, int B, int V, int T, int W, int H, int L,
                                  float *rgba, float *norms2, int norms2_given,
                                  void *ws, hipStream_t s);
size_t shade_specular_backward_ws(int B, int V, int T, int W, int H);
size_t shade_specular_backward_l1_ws(int B, int V, int T, int W, int H);
// signs / sign_upstream: the upstream gradient as mean|image - target|'s sign codes instead of drgba (drgba null)
int launch_shade_specular_backward(const float *drgba, const uint8_t *signs, const float *sign_upstream,
                                   const int32_t *ids, const float *bary,
                                   const float *clip, const float *normals, const float *positions,
                                   const float *diffuse, const float *specular, const int32_t *tris,
                                   const float *light_pos, const float *light_col, const float *ambient,
                                   const float *camera, const float *shininess, int shininess_per_vertex,
                                   const float *norms2, int B, int V, int T, int W, int H, int L, float *dclip,
                                   float *dnormals, float *dpositions, float *ddiffuse, float *dspecular,
                                   float *dshininess, float *light_grads, const int32_t *vertex_offsets,
                                   const int32_t *vertex_entries, const float *transforms, int gbuffer_flags,
                                   int grads_wanted, void *ws, hipStream_t s);
int launch_l1_forward(const float *a, const float *b, size_t n, float *out, uint8_t *signs, float *partials,
                      hipStream_t s);
int launch_l1_backward(const uint8_t *signs, size_t n, const float *upstream, float *da, hipStream_t s);
int launch_image_empty_regions(const float *image, int B, int H, int W, uint8_t *map, hipStream_t s);
int launch_l1_forward_regions(const float *a, const float *b, int B, int H, int W, const uint8_t *empty_a,
                              const uint8_t *empty_b, float *out, uint8_t *signs, float *partials, hipStream_t s);
int launch_export_u8(const float *in, size_t n, uint8_t *out, hipStream_t s);
int launch_vertex_normals(const float *vertices, const int32_t *tris, const int32_t *offsets,
                          const int32_t *entries, int B, int V, float *sums, float *normals, hipStream_t s);
int launch_vertex_normals_backward(const float *dnormals, const float *vertices, const float *sums,
                                   const int32_t *tris, const int32_t *offsets, const int32_t *entries,
                                   int B, int V, float *dvertices, hipStream_t s);
int launch_camera_transforms(const float *eye, const float *center, const float *up, const float *fov_y,
                             const float *near_clip, const float *far_clip, float aspect, int B, float *transforms,
                             int *degenerate, hipStream_t s);
int launch_camera_transforms_backward(const float *dtransforms, const float *eye, const float *center,
                                      const float *up, const float *fov_y, const float *near_clip,
                                      const float *far_clip, float aspect, int B, float *deye, float *dcenter,
                                      float *dup, hipStream_t s);
int launch_tone_map(const float *image, int B, size_t per_image, float gamma, int *max_bits, float *out,
                    uint8_t *out_u8, hipStream_t s);
int soft_max_lights();
int launch_debug_soft_nearest(const float *p, const float *a, const float *b, int n, float *out, hipStream_t s);
size_t soft_ws(int B, int V, int T, int W, int H);
size_t soft_prepared_bytes(int B, int V, int T, int W, int H);
int launch_soft_forward(const float *clip, const float *positions, const float *normals,
                        const float *diffuse, const int32_t *tris, const float *lpos,
                        const float *lint, int B, int V, int T, int W, int H, int L, float sigma,
                        float gamma, float blur, float *rgba, float *aux, void *ws, hipStream_t s);
int launch_soft_backward(const float *drgba, const float *rgba, const float *aux, const float *clip,
                         const float *positions, const float *normals, const float *diffuse,
                         const int32_t *tris, const float *lpos, const float *lint, int B, int V, int T,
                         int W, int H, int L, float sigma, float gamma, float blur, float *dclip,
                         float *dpositions, float *dnormals, float *ddiffuse, float *dlpos, float *dlint,
                         const void *prepared, void *ws, hipStream_t s);

}  // namespace mr
