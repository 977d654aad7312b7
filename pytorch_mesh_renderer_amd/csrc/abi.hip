// extern "C" surface of libmesh_raster_hip.so (see include/mesh_raster.h).
// Argument validation lives here; kernels and launch geometry live in the
// per-stage .hip files.  No allocation, no synchronisation, no exceptions.
#include "mr_internal.h"
#include "mesh_raster_debug.h"

namespace {

inline bool bad_dims(int B, int V, int T, int W, int H) {
  return B < 0 || V < 0 || T < 0 || W < 1 || H < 1 || W > 65535 || H > 65535;
}

inline int check_ws(const void *ws, size_t have, size_t need) {
  if (need == 0) return MR_OK;
  if (ws == nullptr || have < need) return MR_EWORKSPACE;
  if (((uintptr_t)ws & 255u) != 0) return MR_EWORKSPACE;
  return MR_OK;
}

}  // namespace

namespace mr {
extern thread_local int g_raster_region_edge;
extern thread_local int g_raster_repeat;
#ifdef MR_PROBES
extern thread_local int g_raster_probe;
#endif
thread_local KernelTimerSlot g_kernel_timers[MR_TIMER_COUNT];
extern thread_local int g_deterministic;
extern thread_local int g_shade_backward_kernel;
const char *volatile g_last_accumulate_kernel = "";
}

extern "C" {

int mr_version(void) { return 354; /* + mr_rasterize_specular_norms_forward, norms2_given; 353: mr_shade_specular_backward_l1 (352: dclip optional, backward_prepared / prepared, mr_debug_soft_nearest, two's-complement sign codes) */ }

int mr_last_hip_error(void) { return mr::g_last_hip_error; }

int mr_set_deterministic(int on) {
  const int before = mr::g_deterministic;
  mr::g_deterministic = on ? 1 : 0;
  return before;
}

int mr_time_next_kernel(int which, void *start_event, void *stop_event) {
  if (which < 0 || which >= MR_TIMER_COUNT) return MR_EINVAL;
  if ((start_event == nullptr) != (stop_event == nullptr)) return MR_EINVAL;
  mr::g_kernel_timers[which].start = (hipEvent_t)start_event;
  mr::g_kernel_timers[which].stop = (hipEvent_t)stop_event;
  return MR_OK;
}

// ---- mesh_raster_debug.h ----------------------------------------------------------------
int mr_debug_set_raster_region_edge(int edge) {
  if (edge != 0 && edge != 32 && edge != 64) return MR_EINVAL;
  mr::g_raster_region_edge = edge;
  return MR_OK;
}

int mr_debug_set_shade_backward_kernel(int which) {
  if (which < 0 || which > 2) return MR_EINVAL;
  mr::g_shade_backward_kernel = which;
  return MR_OK;
}

int mr_debug_set_raster_repeat(int n) {
  if (n < 1 || n > 64) return MR_EINVAL;
  mr::g_raster_repeat = n;
  return MR_OK;
}

const char *mr_debug_last_accumulate_kernel(void) { return mr::g_last_accumulate_kernel; }

int mr_debug_set_raster_probe(int probe) {
#ifdef MR_PROBES
  static const int kProbes[] = {0, 1, 2, 3, 8, 16, 32, 40, 48, 64};
  for (int v : kProbes) {
    if (v == probe) {
      mr::g_raster_probe = probe;
      return MR_OK;
    }
  }
  return MR_EINVAL;
#else
  return probe == 0 ? MR_OK : MR_EINVAL;  // production build: no probe code in the kernel
#endif
}

int mr_debug_soft_nearest(const float *points, const float *seg_a, const float *seg_b, int n, float *out, void *stream) {
  if (n < 0 || (n > 0 && (!points || !seg_a || !seg_b || !out))) return MR_EINVAL;
  return mr::launch_debug_soft_nearest(points, seg_a, seg_b, n, out, (hipStream_t)stream);
}

size_t mr_rasterize_forward_workspace_bytes(int B, int V, int T, int W, int H) {
  if (bad_dims(B, V, T, W, H)) return 0;
  return mr::raster_forward_ws(B, V, T, W, H);
}

int mr_rasterize_forward(const float *clip, const int32_t *triangles, int B, int V, int T, int W,
                         int H, int32_t *ids, float *bary, float *z, void *workspace,
                         size_t workspace_bytes, void *stream) {
  if (bad_dims(B, V, T, W, H)) return MR_EINVAL;
  if (B == 0) return MR_OK;
  if (!ids || !bary || !z) return MR_EINVAL;
  if ((V > 0 && !clip) || (T > 0 && !triangles)) return MR_EINVAL;
  const int rc = check_ws(workspace, workspace_bytes, mr::raster_forward_ws(B, V, T, W, H));
  if (rc != MR_OK) return rc;
  return mr::launch_raster_forward(clip, triangles, B, V, T, W, H, ids, bary, z, workspace,
                                   (hipStream_t)stream);
}

size_t mr_rasterize_backward_workspace_bytes(int B, int V, int T, int W, int H) {
  if (bad_dims(B, V, T, W, H)) return 0;
  return mr::raster_backward_ws(B, V, T, W, H);
}

int mr_rasterize_backward(const float *dbary, const float *clip, const int32_t *triangles,
                          const int32_t *ids, const float *bary, int B, int V, int T, int W,
                          int H, float *dclip, void *workspace, size_t workspace_bytes,
                          void *stream) {
  if (bad_dims(B, V, T, W, H)) return MR_EINVAL;
  if (B == 0 || V == 0) return MR_OK;
  if (!dbary || !clip || !ids || !bary || !dclip || (T > 0 && !triangles)) return MR_EINVAL;
  const int rc = check_ws(workspace, workspace_bytes, mr::raster_backward_ws(B, V, T, W, H));
  if (rc != MR_OK) return rc;
  return mr::launch_raster_backward(dbary, clip, triangles, ids, bary, B, V, T, W, H, dclip,
                                    workspace, (hipStream_t)stream);
}

int mr_interpolate_forward(const int32_t *ids, const float *bary, const float *attrs,
                           const int32_t *triangles, const float *background, int B, int V,
                           int T, int W, int H, int A, float *out, void *stream) {
  // the reference gathers triangle 0's corners for empty pixels, so T = 0 is an error there too
  if (bad_dims(B, V, T, W, H) || A < 0 || T < 1 || V < 1) return MR_EINVAL;
  if (B == 0 || A == 0) return MR_OK;
  if (!ids || !bary || !attrs || !triangles || !background || !out) return MR_EINVAL;
  return mr::launch_interp_forward(ids, bary, attrs, triangles, background, B, V, T, W, H, A, out,
                                   (hipStream_t)stream);
}

size_t mr_interpolate_backward_workspace_bytes(int B, int V, int T, int W, int H, int A) {
  if (bad_dims(B, V, T, W, H) || A < 0) return 0;
  return mr::interp_backward_ws(B, V, T, W, H, A);
}

int mr_interpolate_backward(const float *dout, const int32_t *ids, const float *bary,
                            const float *attrs, const int32_t *triangles,
                            const float *background, int B, int V, int T, int W, int H, int A,
                            float *dattrs, float *dbary, void *workspace,
                            size_t workspace_bytes, void *stream) {
  if (bad_dims(B, V, T, W, H) || A < 0 || T < 1 || V < 1) return MR_EINVAL;
  if (B == 0) return MR_OK;
  if (!dout || !ids || !bary || !attrs || !triangles || !background || !dattrs || !dbary)
    return MR_EINVAL;
  const int rc = check_ws(workspace, workspace_bytes, mr::interp_backward_ws(B, V, T, W, H, A));
  if (rc != MR_OK) return rc;
  return mr::launch_interp_backward(dout, ids, bary, attrs, triangles, background, B, V, T, W, H,
                                    A, dattrs, dbary, workspace, (hipStream_t)stream);
}

int mr_interpolate_raster_max_attributes(void) { return mr::interp_raster_max_attrs(); }

size_t mr_interpolate_raster_backward_workspace_bytes(int B, int V, int T, int W, int H, int A) {
  if (bad_dims(B, V, T, W, H) || A < 0 || A > mr::interp_raster_max_attrs()) return 0;
  return mr::interp_raster_backward_ws(B, V, T, W, H, A);
}

size_t mr_interpolate_records_bytes(int B, int T, int A) {
  if (B < 0 || T < 0 || A < 1 || A > mr::interp_raster_max_attrs()) return 0;
  return mr::interp_records_bytes(B, T, A);
}

int mr_interpolate_forward_records(const int32_t *ids, const float *bary, const float *attrs,
                                   const int32_t *triangles, const float *background, int B, int V,
                                   int T, int W, int H, int A, float *out, void *records,
                                   size_t records_bytes, void *stream) {
  if (bad_dims(B, V, T, W, H) || A < 1 || A > mr::interp_raster_max_attrs() || T < 1 || V < 1)
    return MR_EINVAL;
  if (B == 0) return MR_OK;
  if (!ids || !bary || !attrs || !triangles || !background || !out) return MR_EINVAL;
  const int rc = check_ws(records, records_bytes, mr::interp_records_bytes(B, T, A));
  if (rc != MR_OK) return rc;
  return mr::launch_interp_forward_records(ids, bary, attrs, triangles, background, B, V, T, W, H, A, out,
                                           records, (hipStream_t)stream);
}

int mr_rasterize_interpolate_forward(const float *clip, const float *attrs, const int32_t *triangles,
                                     const float *background, int B, int V, int T, int W, int H, int A,
                                     int32_t *ids, float *bary, float *z, float *out, void *records,
                                     size_t records_bytes, void *workspace, size_t workspace_bytes, void *stream) {
  if (bad_dims(B, V, T, W, H) || A < 1 || A > mr::interp_raster_max_attrs() || T < 1 || V < 1) return MR_EINVAL;
  if (B == 0) return MR_OK;
  if (!clip || !attrs || !triangles || !background || !ids || !bary || !z || !out || ((uintptr_t)clip & 15u) ||
      ((uintptr_t)records & 255u))
    return MR_EINVAL;
  int rc = check_ws(records, records_bytes, mr::interp_records_bytes(B, T, A));
  if (rc != MR_OK) return rc;
  rc = check_ws(workspace, workspace_bytes, mr::raster_forward_ws(B, V, T, W, H));
  if (rc != MR_OK) return rc;
  return mr::launch_rasterize_interpolate_forward(clip, attrs, triangles, background, B, V, T, W, H, A, ids, bary, z, out,
                                                  records, workspace, (hipStream_t)stream);
}

int mr_interpolate_raster_backward(const float *dout, const int32_t *ids, const float *bary,
                                   const float *clip, const float *attributes,
                                   const int32_t *triangles, const float *background,
                                   const int32_t *vertex_offsets, const int32_t *vertex_entries,
                                   const void *corner_records, int B, int V, int T, int W, int H, int A,
                                   float *dattributes, float *dclip, int gbuffer_flags, void *workspace,
                                   size_t workspace_bytes, void *stream) {
  if (bad_dims(B, V, T, W, H) || A < 0 || A > mr::interp_raster_max_attrs() || (gbuffer_flags & ~MR_GBUFFER_NORMALISED))
    return MR_EINVAL;
  if (B == 0 || V == 0) return MR_OK;
  if (!dclip || (A > 0 && !dattributes)) return MR_EINVAL;
  if (((uintptr_t)dclip & 15u) != 0) return MR_EINVAL;
  if (T > 0 && A > 0 && (size_t)W * H > 0) {
    if (!dout || !ids || !bary || !clip || !attributes || !triangles || !background || !vertex_offsets ||
        !vertex_entries)
      return MR_EINVAL;
    const int rc = check_ws(workspace, workspace_bytes, mr::interp_raster_backward_ws(B, V, T, W, H, A));
    if (rc != MR_OK) return rc;
  }
  if (((uintptr_t)corner_records & 15u) != 0) return MR_EINVAL;
  return mr::launch_interp_raster_backward(dout, ids, bary, clip, attributes, triangles, background,
                                           vertex_offsets, vertex_entries, corner_records, B, V, T, W, H, A,
                                           dattributes, dclip, gbuffer_flags, workspace, (hipStream_t)stream);
}

int mr_shade_max_lights(void) { return mr::shade_max_lights(); }
int mr_shade_fast_lights(void) { return mr::shade_light_gradient_max_lights(); }

int mr_shade_forward(const int32_t *ids, const float *bary, const float *normals,
                     const float *positions, const float *diffuse, const int32_t *triangles,
                     const float *light_positions, const float *light_intensities,
                     const float *ambient, int B, int V, int T, int W, int H, int L, float *rgba,
                     void *workspace, size_t workspace_bytes, void *stream) {
  if (bad_dims(B, V, T, W, H) || T < 1 || V < 1 || L < 1 || L > mr::shade_max_lights())
    return MR_EINVAL;
  if (B == 0) return MR_OK;
  if (!ids || !bary || !normals || !positions || !diffuse || !triangles || !light_positions ||
      !light_intensities || !rgba)
    return MR_EINVAL;
  const int rc = check_ws(workspace, workspace_bytes, mr::shade_forward_ws(B, V, T, W, H));
  if (rc != MR_OK) return rc;
  return mr::launch_shade_forward(ids, bary, normals, positions, diffuse, triangles, light_positions,
                                  light_intensities, ambient, B, V, T, W, H, L, rgba, workspace,
                                  (hipStream_t)stream);
}

int mr_vertex_transform(const float *vertices, const float *transforms, int B, int V, float *clip,
                        void *stream) {
  if (B < 0 || V < 0) return MR_EINVAL;
  if ((size_t)B * V == 0) return MR_OK;
  if (!vertices || !transforms || !clip || ((uintptr_t)clip & 15u) || ((uintptr_t)transforms & 15u))
    return MR_EINVAL;
  return mr::launch_vertex_transform(vertices, transforms, B, V, clip, (hipStream_t)stream);
}

int mr_render_forward(const float *vertices, const float *transforms, const float *normals,
                      const float *diffuse, const int32_t *triangles, const float *light_positions,
                      const float *light_intensities, const float *ambient, int B, int V, int T, int W,
                      int H, int L, float *clip, int32_t *ids, float *bary, float *z, int want_z,
                      float *rgba, uint8_t *rgba_u8, void *corner_records, void *backward_prepared,
                      uint8_t *empty_regions, void *workspace, size_t workspace_bytes, void *stream) {
  if (bad_dims(B, V, T, W, H) || T < 1 || V < 1 || L < 1 || L > mr::shade_max_lights())
    return MR_EINVAL;
  if (B == 0) return MR_OK;
  if (!vertices || !transforms || !normals || !diffuse || !triangles || !light_positions ||
      !light_intensities || !clip || !ids || !bary || !z || !rgba || !corner_records ||
      ((uintptr_t)corner_records & 127u) || ((uintptr_t)clip & 15u) || ((uintptr_t)transforms & 15u) ||
      ((uintptr_t)rgba_u8 & 3u) || ((uintptr_t)backward_prepared & 255u))
    return MR_EINVAL;
  const int rc = check_ws(workspace, workspace_bytes, mr::raster_forward_ws(B, V, T, W, H));
  if (rc != MR_OK) return rc;
  return mr::launch_render_forward(vertices, transforms, normals, diffuse, triangles, light_positions,
                                   light_intensities, ambient, B, V, T, W, H, L, clip, ids, bary, z,
                                   want_z, rgba, rgba_u8, corner_records, backward_prepared, empty_regions, workspace,
                                   (hipStream_t)stream);
}

size_t mr_shade_forward_workspace_bytes(int B, int V, int T, int W, int H) {
  if (bad_dims(B, V, T, W, H)) return 0;
  return mr::shade_forward_ws(B, V, T, W, H);
}

size_t mr_shade_backward_prepared_bytes(int B, int T) {
  if (B < 0 || T < 0) return 0;
  return mr::shade_backward_prepared_bytes(B, T);
}

size_t mr_shade_backward_workspace_bytes(int B, int V, int T, int W, int H) {
  if (bad_dims(B, V, T, W, H)) return 0;
  return mr::shade_backward_ws(B, V, T, W, H);
}

int mr_shade_backward(const float *drgba, const int32_t *ids, const float *bary, const float *clip,
                      const float *normals, const float *positions, const float *diffuse,
                      const int32_t *triangles, const float *light_positions,
                      const float *light_intensities, const float *ambient, int B, int V, int T,
                      int W, int H, int L, float *dclip, float *dnormals, float *dpositions,
                      float *ddiffuse, float *light_grads, const void *corner_records,
                      const int32_t *vertex_offsets, const int32_t *vertex_entries,
                      const float *transforms, int gbuffer_flags, void *prepared, const uint8_t *empty_regions,
                      void *workspace, size_t workspace_bytes, void *stream) {
  if (bad_dims(B, V, T, W, H) || T < 1 || V < 1 || L < 1 || L > mr::shade_max_lights() ||
      (gbuffer_flags & ~MR_GBUFFER_NORMALISED))
    return MR_EINVAL;
  if (B == 0) return MR_OK;
  if (!drgba || !ids || !bary || !clip || !normals || !positions || !diffuse || !triangles ||
      !light_positions || !light_intensities || (!dclip && !transforms) || !dpositions)
    return MR_EINVAL;
  if ((!dnormals || !ddiffuse) && !vertex_offsets) return MR_EINVAL;  /* only the per-vertex gather can leave outputs out */
  if (((uintptr_t)corner_records & 127u) != 0) return MR_EINVAL;
  if ((vertex_offsets == nullptr) != (vertex_entries == nullptr)) return MR_EINVAL;
  if (vertex_offsets && ((uintptr_t)dclip & 15u) != 0) return MR_EINVAL;
  if (transforms && (!vertex_offsets || ((uintptr_t)transforms & 3u) != 0)) return MR_EINVAL;
  if (((uintptr_t)prepared & 255u) != 0) return MR_EINVAL;
  const int rc = check_ws(workspace, workspace_bytes, mr::shade_backward_ws(B, V, T, W, H));
  if (rc != MR_OK) return rc;
  return mr::launch_shade_backward(drgba, nullptr, nullptr, ids, bary, clip, normals, positions, diffuse,
                                   triangles, light_positions, light_intensities, ambient, B, V, T, W, H, L,
                                   dclip, dnormals, dpositions, ddiffuse, light_grads, corner_records,
                                   vertex_offsets, vertex_entries, transforms, gbuffer_flags, prepared, empty_regions, workspace,
                                   (hipStream_t)stream);
}

size_t mr_shade_backward_l1_workspace_bytes(int B, int V, int T, int W, int H) {
  if (bad_dims(B, V, T, W, H)) return 0;
  return mr::shade_backward_ws(B, V, T, W, H) + 256;
}

int mr_shade_backward_l1(const uint8_t *signs, const float *upstream, const int32_t *ids,
                         const float *bary, const float *clip, const float *normals,
                         const float *positions, const float *diffuse, const int32_t *triangles,
                         const float *light_positions, const float *light_intensities,
                         const float *ambient, int B, int V, int T, int W, int H, int L, float *dclip,
                         float *dnormals, float *dpositions, float *ddiffuse, float *light_grads,
                         const void *corner_records, const int32_t *vertex_offsets,
                         const int32_t *vertex_entries, const float *transforms, int gbuffer_flags,
                         void *prepared, const uint8_t *empty_regions, void *workspace, size_t workspace_bytes,
                         void *stream) {
  if (bad_dims(B, V, T, W, H) || T < 1 || V < 1 || L < 1 || L > mr::shade_max_lights() ||
      (gbuffer_flags & ~MR_GBUFFER_NORMALISED))
    return MR_EINVAL;
  if (B == 0) return MR_OK;
  if (!signs || !upstream || !ids || !bary || !clip || !normals || !positions || !diffuse || !triangles ||
      !light_positions || !light_intensities || (!dclip && !transforms) || !dpositions)
    return MR_EINVAL;
  if ((!dnormals || !ddiffuse) && !vertex_offsets) return MR_EINVAL;  /* only the per-vertex gather can leave outputs out */
  if (((uintptr_t)corner_records & 127u) != 0) return MR_EINVAL;
  if ((vertex_offsets == nullptr) != (vertex_entries == nullptr)) return MR_EINVAL;
  if (vertex_offsets && ((uintptr_t)dclip & 15u) != 0) return MR_EINVAL;
  if (transforms && (!vertex_offsets || ((uintptr_t)transforms & 3u) != 0)) return MR_EINVAL;
  if (((uintptr_t)prepared & 255u) != 0) return MR_EINVAL;
  const int rc = check_ws(workspace, workspace_bytes, mr::shade_backward_ws(B, V, T, W, H) + 256);
  if (rc != MR_OK) return rc;
  return mr::launch_shade_backward(nullptr, signs, upstream, ids, bary, clip, normals, positions, diffuse,
                                   triangles, light_positions, light_intensities, ambient, B, V, T, W, H, L,
                                   dclip, dnormals, dpositions, ddiffuse, light_grads, corner_records,
                                   vertex_offsets, vertex_entries, transforms, gbuffer_flags, prepared, empty_regions, workspace,
                                   (hipStream_t)stream);
}

size_t mr_shade_specular_forward_workspace_bytes(int B, int V, int T, int W, int H) {
  if (bad_dims(B, V, T, W, H)) return 0;
  return mr::shade_specular_forward_ws(B, V, T, W, H);
}

int mr_shade_specular_forward(const int32_t *ids, const float *bary, const float *normals,
                              const float *positions, const float *diffuse, const float *specular,
                              const int32_t *triangles, const float *light_positions,
                              const float *light_intensities, const float *ambient,
                              const float *camera_position, const float *shininess,
                              int shininess_per_vertex, int B, int V, int T, int W, int H, int L,
                              float *rgba, float *norms2, int norms2_given, void *workspace, size_t workspace_bytes,
                              void *stream) {
  if (bad_dims(B, V, T, W, H) || T < 1 || V < 1 || L < 1 || L > mr::shade_light_gradient_max_lights())
    return MR_EINVAL;
  if (norms2_given != 0 && norms2_given != 1) return MR_EINVAL;
  if (B == 0) return MR_OK;
  if (!ids || !bary || !normals || !positions || !diffuse || !specular || !triangles ||
      !light_positions || !light_intensities || !camera_position || !shininess || !rgba || !norms2)
    return MR_EINVAL;
  const int rc = check_ws(workspace, workspace_bytes, mr::shade_specular_forward_ws(B, V, T, W, H));
  if (rc != MR_OK) return rc;
  return mr::launch_shade_specular_forward(ids, bary, normals, positions, diffuse, specular, triangles,
                                           light_positions, light_intensities, ambient,
                                           camera_position, shininess, shininess_per_vertex, B, V, T, W,
                                           H, L, rgba, norms2, norms2_given,
                                           workspace, (hipStream_t)stream);
}

size_t mr_rasterize_specular_norms_workspace_bytes(int B, int V, int T, int W, int H) {
  if (bad_dims(B, V, T, W, H)) return 0;
  return mr::rasterize_specular_norms_ws(B, V, T, W, H);
}

int mr_rasterize_specular_norms_forward(const float *clip, const int32_t *triangles, const float *normals,
                                        const float *positions, const float *light_positions,
                                        const float *camera_position, int B, int V, int T, int W, int H, int L,
                                        int32_t *ids, float *bary, float *z, int want_z, float *norms2,
                                        void *workspace, size_t workspace_bytes, void *stream) {
  if (bad_dims(B, V, T, W, H) || T < 1 || V < 1 || L < 1 || L > mr::shade_light_gradient_max_lights())
    return MR_EINVAL;
  if (B == 0) return MR_OK;
  if (!clip || !triangles || !normals || !positions || !light_positions || !camera_position || !ids || !bary || !z ||
      !norms2)
    return MR_EINVAL;
  const int rc = check_ws(workspace, workspace_bytes, mr::rasterize_specular_norms_ws(B, V, T, W, H));
  if (rc != MR_OK) return rc;
  return mr::launch_rasterize_specular_norms(clip, triangles, normals, positions, light_positions, camera_position, B, V, T,
                                             W, H, L, ids, bary, z, want_z != 0, norms2, workspace, (hipStream_t)stream);
}

size_t mr_shade_specular_backward_workspace_bytes(int B, int V, int T, int W, int H) {
  if (bad_dims(B, V, T, W, H)) return 0;
  return mr::shade_specular_backward_ws(B, V, T, W, H);
}

int mr_shade_specular_backward(const float *drgba, const int32_t *ids, const float *bary,
                               const float *clip, const float *normals, const float *positions,
                               const float *diffuse, const float *specular,
                               const int32_t *triangles, const float *light_positions,
                               const float *light_intensities, const float *ambient,
                               const float *camera_position, const float *shininess,
                               int shininess_per_vertex, const float *norms2, int B, int V, int T,
                               int W, int H, int L, float *dclip, float *dnormals, float *dpositions,
                               float *ddiffuse, float *dspecular, float *dshininess,
                               float *light_grads, const int32_t *vertex_offsets,
                               const int32_t *vertex_entries, const float *transforms, int gbuffer_flags,
                               int grads_wanted, void *workspace, size_t workspace_bytes, void *stream) {
  if (bad_dims(B, V, T, W, H) || T < 1 || V < 1 || L < 1 || L > mr::shade_light_gradient_max_lights())
    return MR_EINVAL;
  if ((gbuffer_flags & ~MR_GBUFFER_NORMALISED) != 0 || (grads_wanted & ~MR_GRAD_ALL) != 0) return MR_EINVAL;
  if ((vertex_offsets == nullptr) != (vertex_entries == nullptr)) return MR_EINVAL;
  if (vertex_offsets && ((uintptr_t)dclip & 15u) != 0) return MR_EINVAL;
  if (B == 0) return MR_OK;
  if (!drgba || !ids || !bary || !clip || !normals || !positions || !diffuse || !specular ||
      !triangles || !light_positions || !light_intensities || !camera_position || !shininess ||
      !norms2 || !dclip || !dnormals || !dpositions || !ddiffuse || !dspecular || !light_grads ||
      (shininess_per_vertex && !dshininess))
    return MR_EINVAL;
  const int rc = check_ws(workspace, workspace_bytes, mr::shade_specular_backward_ws(B, V, T, W, H));
  if (rc != MR_OK) return rc;
  return mr::launch_shade_specular_backward(drgba, nullptr, nullptr, ids, bary, clip, normals, positions, diffuse,
                                            specular, triangles, light_positions, light_intensities,
                                            ambient, camera_position, shininess, shininess_per_vertex,
                                            norms2, B, V, T, W, H, L, dclip, dnormals, dpositions,
                                            ddiffuse, dspecular, dshininess, light_grads, vertex_offsets,
                                            vertex_entries, transforms, gbuffer_flags, grads_wanted, workspace,
                                            (hipStream_t)stream);
}

size_t mr_shade_specular_backward_l1_workspace_bytes(int B, int V, int T, int W, int H) {
  if (bad_dims(B, V, T, W, H)) return 0;
  return mr::shade_specular_backward_l1_ws(B, V, T, W, H);
}

int mr_shade_specular_backward_l1(const uint8_t *signs, const float *upstream, const int32_t *ids,
                                  const float *bary,
                               const float *clip, const float *normals, const float *positions,
                               const float *diffuse, const float *specular,
                               const int32_t *triangles, const float *light_positions,
                               const float *light_intensities, const float *ambient,
                               const float *camera_position, const float *shininess,
                               int shininess_per_vertex, const float *norms2, int B, int V, int T,
                               int W, int H, int L, float *dclip, float *dnormals, float *dpositions,
                               float *ddiffuse, float *dspecular, float *dshininess,
                               float *light_grads, const int32_t *vertex_offsets,
                               const int32_t *vertex_entries, const float *transforms, int gbuffer_flags,
                               int grads_wanted, void *workspace, size_t workspace_bytes, void *stream) {
  if (bad_dims(B, V, T, W, H) || T < 1 || V < 1 || L < 1 || L > mr::shade_light_gradient_max_lights())
    return MR_EINVAL;
  if ((gbuffer_flags & ~MR_GBUFFER_NORMALISED) != 0 || (grads_wanted & ~MR_GRAD_ALL) != 0) return MR_EINVAL;
  if ((vertex_offsets == nullptr) != (vertex_entries == nullptr)) return MR_EINVAL;
  if (vertex_offsets && ((uintptr_t)dclip & 15u) != 0) return MR_EINVAL;
  if (B == 0) return MR_OK;
  if (!signs || !upstream || !ids || !bary || !clip || !normals || !positions || !diffuse || !specular ||
      !triangles || !light_positions || !light_intensities || !camera_position || !shininess ||
      !norms2 || !dclip || !dnormals || !dpositions || !ddiffuse || !dspecular || !light_grads ||
      (shininess_per_vertex && !dshininess))
    return MR_EINVAL;
  const int rc = check_ws(workspace, workspace_bytes, mr::shade_specular_backward_l1_ws(B, V, T, W, H));
  if (rc != MR_OK) return rc;
  return mr::launch_shade_specular_backward(nullptr, signs, upstream, ids, bary, clip, normals, positions, diffuse,
                                            specular, triangles, light_positions, light_intensities,
                                            ambient, camera_position, shininess, shininess_per_vertex,
                                            norms2, B, V, T, W, H, L, dclip, dnormals, dpositions,
                                            ddiffuse, dspecular, dshininess, light_grads, vertex_offsets,
                                            vertex_entries, transforms, gbuffer_flags, grads_wanted, workspace,
                                            (hipStream_t)stream);
}

int mr_soft_max_lights(void) { return mr::soft_max_lights(); }

size_t mr_soft_workspace_bytes(int B, int V, int T, int W, int H) {
  if (bad_dims(B, V, T, W, H)) return 0;
  return mr::soft_ws(B, V, T, W, H);
}

size_t mr_soft_prepared_bytes(int B, int V, int T, int W, int H) {
  if (bad_dims(B, V, T, W, H)) return 0;
  return mr::soft_prepared_bytes(B, V, T, W, H);
}

int mr_soft_forward(const float *clip, const float *positions, const float *normals,
                    const float *diffuse, const int32_t *triangles, const float *light_positions,
                    const float *light_intensities, int B, int V, int T, int W, int H, int L,
                    float sigma, float gamma, float blur, float *rgba, float *aux, void *workspace,
                    size_t workspace_bytes, void *stream) {
  if (bad_dims(B, V, T, W, H) || L < 1 || L > mr::soft_max_lights() || !(sigma > 0.f) || !(gamma > 0.f))
    return MR_EINVAL;
  if (B == 0) return MR_OK;
  if (!rgba || !aux || !light_positions || !light_intensities) return MR_EINVAL;
  if (T > 0 && (!clip || !positions || !normals || !diffuse || !triangles)) return MR_EINVAL;
  const int rc = check_ws(workspace, workspace_bytes, mr::soft_prepared_bytes(B, V, T, W, H));  // the head is all it uses
  if (rc != MR_OK) return rc;
  return mr::launch_soft_forward(clip, positions, normals, diffuse, triangles, light_positions,
                                 light_intensities, B, V, T, W, H, L, sigma, gamma, blur, rgba, aux,
                                 workspace, (hipStream_t)stream);
}

int mr_soft_backward(const float *drgba, const float *rgba, const float *aux, const float *clip,
                     const float *positions, const float *normals, const float *diffuse,
                     const int32_t *triangles, const float *light_positions,
                     const float *light_intensities, int B, int V, int T, int W, int H, int L,
                     float sigma, float gamma, float blur, float *dclip, float *dpositions,
                     float *dnormals, float *ddiffuse, float *dlight_positions,
                     float *dlight_intensities, const void *prepared, void *workspace, size_t workspace_bytes,
                     void *stream) {
  if (bad_dims(B, V, T, W, H) || L < 1 || L > mr::soft_max_lights() || !(sigma > 0.f) || !(gamma > 0.f))
    return MR_EINVAL;
  if (B == 0) return MR_OK;
  if (!drgba || !rgba || !aux || !light_positions || !light_intensities || !dlight_positions ||
      !dlight_intensities)
    return MR_EINVAL;
  if (V > 0 && (!clip || !positions || !normals || !diffuse || !dclip || !dpositions || !dnormals || !ddiffuse))
    return MR_EINVAL;
  if (T > 0 && !triangles) return MR_EINVAL;
  const int rc = check_ws(workspace, workspace_bytes, mr::soft_ws(B, V, T, W, H));
  if (rc != MR_OK) return rc;
  return mr::launch_soft_backward(drgba, rgba, aux, clip, positions, normals, diffuse, triangles,
                                  light_positions, light_intensities, B, V, T, W, H, L, sigma, gamma,
                                  blur, dclip, dpositions, dnormals, ddiffuse, dlight_positions,
                                  dlight_intensities, prepared, workspace, (hipStream_t)stream);
}

int mr_l1_loss_partials(void) { return MR_L1_PARTIALS; }

int mr_l1_loss_forward(const float *a, const float *b, size_t n, float *loss, uint8_t *signs,
                       float *partials, void *stream) {
  if (!loss || (n > 0 && (!a || !b || !partials))) return MR_EINVAL;
  if ((((uintptr_t)a | (uintptr_t)b) & 15u) != 0) return MR_EINVAL;
  return mr::launch_l1_forward(a, b, n, loss, signs, partials, (hipStream_t)stream);
}

size_t mr_empty_regions_bytes(int B, int W, int H) {
  if (B < 0 || W < 0 || H < 0) return 0;
  return (size_t)B * ((H + 63) / 64) * ((W + 63) / 64);
}

int mr_image_empty_regions(const float *image, int B, int H, int W, uint8_t *map, void *stream) {
  if (B < 0 || H < 0 || W < 0 || H > 65535 || W > 65535) return MR_EINVAL;
  if ((size_t)B * H * W == 0) return MR_OK;
  if (!image || !map || ((uintptr_t)image & 15u)) return MR_EINVAL;
  return mr::launch_image_empty_regions(image, B, H, W, map, (hipStream_t)stream);
}

int mr_l1_loss_forward_regions(const float *a, const float *b, int B, int H, int W, const uint8_t *empty_a,
                               const uint8_t *empty_b, float *loss, uint8_t *signs, float *partials, void *stream) {
  if (B < 0 || H < 0 || W < 0 || H > 65535 || W > 65535 || !loss) return MR_EINVAL;
  if ((size_t)B * H * W > 0 && (!a || !b || !partials || !empty_a || !empty_b)) return MR_EINVAL;
  if ((((uintptr_t)a | (uintptr_t)b) & 15u) != 0) return MR_EINVAL;
  return mr::launch_l1_forward_regions(a, b, B, H, W, empty_a, empty_b, loss, signs, partials, (hipStream_t)stream);
}

int mr_l1_loss_backward(const uint8_t *signs, size_t n, const float *upstream, float *da, void *stream) {
  if (n > 0 && (!signs || !upstream || !da)) return MR_EINVAL;
  if (((uintptr_t)da & 15u) != 0) return MR_EINVAL;
  return mr::launch_l1_backward(signs, n, upstream, da, (hipStream_t)stream);
}

int mr_camera_transforms(const float *eye, const float *center, const float *up, const float *fov_y,
                         const float *near_clip, const float *far_clip, float aspect, int B, float *transforms,
                         int32_t *degenerate, void *stream) {
  if (B < 0 || !(aspect > 0.0f)) return MR_EINVAL;
  if (B == 0) return MR_OK;
  if (!eye || !center || !up || !fov_y || !near_clip || !far_clip || !transforms || !degenerate) return MR_EINVAL;
  return mr::launch_camera_transforms(eye, center, up, fov_y, near_clip, far_clip, aspect, B, transforms,
                                      (int *)degenerate, (hipStream_t)stream);
}

int mr_camera_transforms_backward(const float *dtransforms, const float *eye, const float *center,
                                  const float *up, const float *fov_y, const float *near_clip,
                                  const float *far_clip, float aspect, int B, float *deye, float *dcenter,
                                  float *dup, void *stream) {
  if (B < 0 || !(aspect > 0.0f)) return MR_EINVAL;
  if (B == 0) return MR_OK;
  if (!dtransforms || !eye || !center || !up || !fov_y || !near_clip || !far_clip || !deye || !dcenter || !dup)
    return MR_EINVAL;
  return mr::launch_camera_transforms_backward(dtransforms, eye, center, up, fov_y, near_clip, far_clip, aspect, B,
                                               deye, dcenter, dup, (hipStream_t)stream);
}

int mr_vertex_normals_forward(const float *vertices, const int32_t *triangles,
                              const int32_t *vertex_offsets, const int32_t *vertex_entries, int B, int V,
                              int T, float *sums, float *normals, void *stream) {
  if (B < 0 || V < 0 || T < 0) return MR_EINVAL;
  if (B == 0 || V == 0) return MR_OK;
  if (!vertices || !vertex_offsets || !sums || !normals || (T > 0 && (!triangles || !vertex_entries)))
    return MR_EINVAL;
  return mr::launch_vertex_normals(vertices, triangles, vertex_offsets, vertex_entries, B, V, sums, normals,
                                   (hipStream_t)stream);
}

int mr_vertex_normals_backward(const float *dnormals, const float *vertices, const float *sums,
                               const int32_t *triangles, const int32_t *vertex_offsets,
                               const int32_t *vertex_entries, int B, int V, int T, float *dvertices,
                               void *stream) {
  if (B < 0 || V < 0 || T < 0) return MR_EINVAL;
  if (B == 0 || V == 0) return MR_OK;
  if (!dnormals || !vertices || !sums || !vertex_offsets || !dvertices ||
      (T > 0 && (!triangles || !vertex_entries)))
    return MR_EINVAL;
  return mr::launch_vertex_normals_backward(dnormals, vertices, sums, triangles, vertex_offsets,
                                            vertex_entries, B, V, dvertices, (hipStream_t)stream);
}

int mr_tone_map(const float *image, int B, size_t elements_per_image, float gamma, int32_t *max_scratch,
                float *out_f32, uint8_t *out_u8, void *stream) {
  if (B < 0) return MR_EINVAL;
  if (B == 0 || elements_per_image == 0) return MR_OK;
  if (!image || !max_scratch || ((out_f32 == nullptr) == (out_u8 == nullptr))) return MR_EINVAL;
  return mr::launch_tone_map(image, B, elements_per_image, gamma, (int *)max_scratch, out_f32, out_u8,
                             (hipStream_t)stream);
}

int mr_export_u8(const float *image, size_t n, uint8_t *out, void *stream) {
  if (n > 0 && (!image || !out)) return MR_EINVAL;
  if (((uintptr_t)image & 15u) != 0 || ((uintptr_t)out & 3u) != 0) return MR_EINVAL;
  return mr::launch_export_u8(image, n, out, (hipStream_t)stream);
}

}  // extern "C"
