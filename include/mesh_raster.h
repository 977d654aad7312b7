/*
 * mesh_raster.h -- C ABI of the MI355X (gfx950) differentiable mesh rasterizer.
 *
 * Drop-in boundary for the reference's one native module, the pybind11
 * extension `rasterize_triangles_cpp`
 *   (src/mesh_renderer/kernels/rasterize_triangles.cpp:421-424), whose only
 * caller is BarycentricRasterizer (src/mesh_renderer/rasterize_triangles_ext.py
 * :39-40 forward, :56-61 backward).  The interpolation / shading entry points
 * replace the eager-torch bodies of rasterize_clip_space
 * (src/mesh_renderer/rasterize.py:112-150) and phong_shader
 * (src/mesh_renderer/render.py:287-386).
 *
 * Conventions for every entry point
 *   - all data pointers are DEVICE pointers (HBM resident), contiguous, C order;
 *   - the caller owns every buffer, outputs included; nothing is allocated or
 *     freed here, so a call sequence can be captured into a hipGraph;
 *   - work is enqueued on `stream` (a hipStream_t passed as void*; NULL = the
 *     legacy default stream) and the call returns without synchronising;
 *   - `workspace` is scratch the call may overwrite; its minimum size comes
 *     from the matching *_workspace_bytes() query, 256-byte aligned;
 *   - return value: MR_OK or a negative MR_E* code; no exceptions cross the ABI;
 *   - batched: B independent images share one `triangles` array; the
 *     reference's per-image call is B = 1.
 *
 * Numerical contract: mr_rasterize_forward reproduces the reference's ids, z
 * and barycentrics BIT FOR BIT (un-fused binary32 arithmetic in the
 * reference's evaluation order, binary64 pixel centres / bbox projection;
 * SURVEY.md Appendix A).  Gradients agree within 1e-4 abs (summation order
 * differs: LDS/atomic accumulation instead of a serial row-major loop).
 */
#ifndef MESH_RASTER_H_
#define MESH_RASTER_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MR_OK 0
#define MR_EINVAL (-1)     /* null pointer, negative count, W/H out of [1, 65535] */
#define MR_EWORKSPACE (-2) /* workspace missing, too small or misaligned */
#define MR_ELAUNCH (-3)    /* a HIP launch / memset failed (see mr_last_hip_error) */

/* Library / ABI version: major*10000 + minor*100 + patch. */
int mr_version(void);
/* hipError_t of the most recent failed launch in this process, 0 if none. */
int mr_last_hip_error(void);

/* ---- rasterize_triangles_cpp.forward -------------------------------------
 * Replaces rasterize_triangles_forward (rasterize_triangles.cpp:302-419).
 *   clip      [B,V,4] f32  clip-space XYZW
 *   triangles [T,3]   i32  vertex ids (a triangle with an id outside [0,V) is
 *                          skipped; the reference would read out of bounds)
 *   ids       [B,H,W]   i32 out  winning triangle id, 0 where nothing was drawn
 *   bary      [B,H,W,3] f32 out  perspective-correct barycentrics, 0 where empty
 *   z         [B,H,W]   f32 out  NDC depth of the winner, 1.0 where empty
 * Row 0 is the BOTTOM scanline (NDC y = -1), as in the reference. */
size_t mr_rasterize_forward_workspace_bytes(int B, int V, int T, int W, int H);
int mr_rasterize_forward(const float *clip, const int32_t *triangles,
                         int B, int V, int T, int W, int H,
                         int32_t *ids, float *bary, float *z,
                         void *workspace, size_t workspace_bytes, void *stream);

/* ---- rasterize_triangles_cpp.backward ------------------------------------
 * Replaces rasterize_triangles_backward (rasterize_triangles.cpp:131-273).
 *   dbary [B,H,W,3] f32   dL/d(bary)
 *   ids, bary             the forward's outputs
 *   dclip [B,V,4] f32 out dL/d(clip); zeroed here; column 2 (z) stays 0 */
size_t mr_rasterize_backward_workspace_bytes(int B, int V, int T, int W, int H);
int mr_rasterize_backward(const float *dbary, const float *clip,
                          const int32_t *triangles, const int32_t *ids,
                          const float *bary, int B, int V, int T, int W, int H,
                          float *dclip, void *workspace, size_t workspace_bytes,
                          void *stream);

/* ---- deferred attribute interpolation -------------------------------------
 * Replaces the gather / multiply / sum / alpha / background-blend block of
 * rasterize_clip_space (src/mesh_renderer/rasterize.py:118-150).
 *   attrs      [B,V,A] f32
 *   background [A]     f32
 *   out        [B,H,W,A] f32 out (not flipped) */
int mr_interpolate_forward(const int32_t *ids, const float *bary,
                           const float *attrs, const int32_t *triangles,
                           const float *background, int B, int V, int T,
                           int W, int H, int A, float *out, void *stream);

/* Its autograd backward (index_put_(accumulate) + d/d bary).
 *   dout   [B,H,W,A] f32
 *   dattrs [B,V,A]   f32 out, zeroed here
 *   dbary  [B,H,W,3] f32 out */
size_t mr_interpolate_backward_workspace_bytes(int B, int V, int T, int W, int H, int A);
int mr_interpolate_backward(const float *dout, const int32_t *ids,
                            const float *bary, const float *attrs,
                            const int32_t *triangles, const float *background,
                            int B, int V, int T, int W, int H, int A,
                            float *dattrs, float *dbary, void *workspace,
                            size_t workspace_bytes, void *stream);

/* Backward of mr_interpolate_forward AND of the rasterizer underneath it in ONE pass over the
 * G-buffer, for 1 <= A <= mr_interpolate_raster_max_attributes() (replaces the pair
 * mr_interpolate_backward + mr_rasterize_backward, which needs ceil(A / 4) + 2 passes):
 *   dout            [B,H,W,A] f32   dL/d(out of mr_interpolate_forward)
 *   clip            [B,V,4]   f32   the clip-space vertices the G-buffer was made from
 *   vertex_offsets, vertex_entries  CSR vertex -> (triangle, corner) adjacency of `triangles`
 *                                   (see mr_shade_backward)
 *   dattributes     [B,V,A]   f32 out, dclip [B,V,4] f32 out (16-byte aligned; column z stays 0);
 *                                   both are written completely, no pre-zeroing needed
 *   corner_records  NULL, or the `records` buffer mr_interpolate_forward_records filled for the
 *                   SAME inputs: reused instead of rebuilt
 *   gbuffer_flags   MR_GBUFFER_NORMALISED (see mr_shade_backward): ids / bary are the rasterizer's own
 *                   output for these vertices -- alpha is exactly 1 on every covered pixel, so up to 12
 *                   attributes the pixel pass keeps the 3 A + 9 products in registers down each lane's
 *                   vertical run over difference-basis records (round 4); 0: any G-buffer, general kernels
 * mr_interpolate_forward_records is mr_interpolate_forward for 1 <= A <= that maximum, through
 * per-(image, triangle) corner records (two dependent load levels instead of three; `records`:
 * mr_interpolate_records_bytes(B, T, A) bytes, 256-byte aligned, caller-owned). */
int mr_interpolate_raster_max_attributes(void);
size_t mr_interpolate_records_bytes(int B, int T, int A);

/* rasterize_clip_space()'s forward (src/mesh_renderer/rasterize.py:66-152) for 1 <= A <=
 * mr_interpolate_raster_max_attributes() attributes in ONE pass over the pixels: mr_rasterize_forward with
 * mr_interpolate_forward_records' arithmetic as the epilogue of the tile walk, on the pixel state still in
 * registers (the G-buffer is not read back; the winners' attribute records come from LDS).  Outputs: ids and
 * bary exactly as mr_rasterize_forward writes them, `out` [B,H,W,A] exactly mr_interpolate_forward_records'
 * expression (G-buffer row order), `records` as that call fills them (for mr_interpolate_raster_backward).
 *   z          [B,H,W] f32: scratch -- written only where a crowded region needs it as state between its bin
 *              rounds; undefined afterwards
 *   workspace  mr_rasterize_forward_workspace_bytes() bytes */
int mr_rasterize_interpolate_forward(const float *clip, const float *attrs, const int32_t *triangles,
                                     const float *background, int B, int V, int T, int W, int H, int A,
                                     int32_t *ids, float *bary, float *z, float *out, void *records,
                                     size_t records_bytes, void *workspace, size_t workspace_bytes, void *stream);
int mr_interpolate_forward_records(const int32_t *ids, const float *bary, const float *attrs,
                                   const int32_t *triangles, const float *background, int B, int V,
                                   int T, int W, int H, int A, float *out, void *records,
                                   size_t records_bytes, void *stream);
size_t mr_interpolate_raster_backward_workspace_bytes(int B, int V, int T, int W, int H, int A);
int mr_interpolate_raster_backward(const float *dout, const int32_t *ids, const float *bary,
                                   const float *clip, const float *attributes,
                                   const int32_t *triangles, const float *background,
                                   const int32_t *vertex_offsets, const int32_t *vertex_entries,
                                   const void *corner_records, int B, int V, int T, int W, int H,
                                   int A, float *dattributes, float *dclip, int gbuffer_flags,
                                   void *workspace, size_t workspace_bytes, void *stream);

/* ---- fused deferred shading (diffuse + ambient Phong) ---------------------------
 * Replaces, for render() without specular terms, attribute interpolation
 * (src/mesh_renderer/rasterize.py:118-150), the unpack / normalise / mask block of
 * render() (src/mesh_renderer/render.py:199-215) and phong_shader's ambient +
 * diffuse terms, alpha mask and vertical flip (render.py:287-323, 373-386).
 *   ids, bary                 the G-buffer of mr_rasterize_forward
 *   normals, positions, diffuse  [B,V,3] f32 per-vertex attributes (positions =
 *                             world-space vertices)
 *   light_positions, light_intensities [B,L,3] f32, 1 <= L <= mr_shade_max_lights() (32: the
 *                             reference takes any count, render.py:304-323; up to mr_shade_fast_lights()
 *                             = 4 of them are kept in registers, further ones go through a run-time
 *                             loop).  Two things stay at mr_shade_fast_lights() per call: the specular
 *                             entry points, and light_grads of mr_shade_backward[_l1] -- a caller
 *                             with more lights asks for the light gradients four lights at a time
 *                             (each light's gradient depends on that light alone; see
 *                             pytorch_mesh_renderer_amd/_native.py, shade_backward)
 *   ambient                   [B,3] f32 or NULL
 *   rgba                      [B,H,W,4] f32 out; row 0 is the TOP scanline; alpha is
 *                             1 on covered pixels with a non-negative diffuse colour */
int mr_shade_max_lights(void);
int mr_shade_fast_lights(void);
size_t mr_shade_forward_workspace_bytes(int B, int V, int T, int W, int H);
int mr_shade_forward(const int32_t *ids, const float *bary, const float *normals,
                     const float *positions, const float *diffuse,
                     const int32_t *triangles, const float *light_positions,
                     const float *light_intensities, const float *ambient,
                     int B, int V, int T, int W, int H, int L, float *rgba,
                     void *workspace, size_t workspace_bytes, void *stream);

/* clip[b,v] = transforms[b] . (vertices[b,v], 1): the clip-space transform render() applies before
 * rasterizing (camera_utils.transform_homogeneous, src/common/camera_utils.py:142-170), one thread
 * per vertex; each row is evaluated as ((m0 x + m1 y) + m2 z) + m3 without contraction.
 *   vertices [B,V,3] f32, transforms [B,4,4] f32 row-major (16-byte aligned), clip [B,V,4] f32 out
 *   (16-byte aligned). */
int mr_vertex_transform(const float *vertices, const float *transforms, int B, int V, float *clip,
                        void *stream);

/* render()'s forward from world-space vertices to the image (src/mesh_renderer/render.py:183-228
 * without a specular term): the clip-space transform of camera_utils.transform_homogeneous
 * (src/common/camera_utils.py:142-170), mr_rasterize_forward and mr_shade_forward, the last two in
 * ONE pass over the pixels: the shading runs as the epilogue of the rasterizer's tile walk, on the
 * pixel state still held in registers, so the G-buffer is not read back (and the id ->
 * corner-record gather has one dependent level less).  What FusedPhongRenderer calls.  Outputs:
 * the G-buffer of mr_rasterize_forward run on `clip` (ids, bary, z; bit-identical) and the image
 * of mr_shade_forward (rgba, within the shading's 1e-4 budget).
 *   vertices        [B,V,3] f32  world-space positions (also the shading's `positions`)
 *   transforms      [B,4,4] f32  row-major clip-space transforms, 16-byte aligned
 *   clip            [B,V,4] f32 out  transforms[b] . (vertices[b,v], 1), evaluated per row as
 *                   ((m0 x + m1 y) + m2 z) + m3 without contraction; 16-byte aligned.  The
 *                   backward entry points take it as their `clip` argument.
 *   rgba_u8         NULL, or [B,H,W,4] u8 out (4-byte aligned): the same image as 8-bit frames, exactly
 *                   what mr_export_u8 would make of `rgba` -- for callers that hand frames over or
 *                   write them to files and would otherwise read the float image back (4 B/px of
 *                   stores instead of a 20 B/px pass)
 *   z, want_z       z is always a [B,H,W] buffer; with want_z == 0 the caller declares that it will
 *                   not read it (render() does not): the depth plane is then written only where a
 *                   crowded region needs it as state between its bin rounds, 4 B/px of stores less,
 *                   and its contents after the call are undefined.
 *   corner_records  out, mr_shade_forward_workspace_bytes() bytes, 128-byte aligned: the gathered
 *                   per-triangle attribute records; may be handed to mr_shade_backward.
 *   backward_prepared  NULL, or out: mr_shade_backward_prepared_bytes(B, T) bytes, 256-byte aligned.  For a
 *                   caller that will differentiate the image to the world-space vertices ONLY (the usual
 *                   optimisation loop): the per-triangle setup kernel -- which holds the corner attributes
 *                   and the sign-corrected adjugate in registers anyway -- also writes the records of the
 *                   folded shading backward and clears its accumulator rows.  Handed to
 *                   mr_shade_backward[_l1] as `prepared`, the backward then needs no setup launch.
 *   empty_regions   NULL, or out: mr_empty_regions_bytes(B, W, H) bytes, [B][ceil(H/64)][ceil(W/64)]: 1 = the
 *                   64 x 64-pixel block (G-buffer rows 64 by .. 64 by + 63, i.e. image rows H - 1 - y) is whole and
 *                   holds no candidate triangle: background in the G-buffer, transparent black in the image.
 *                   mr_l1_loss_forward_regions and mr_shade_backward[_l1] skip such blocks without reading them.
 *                   (Launches small enough for 32-pixel regions flag nothing.)
 *   workspace       mr_rasterize_forward_workspace_bytes() bytes */
size_t mr_empty_regions_bytes(int B, int W, int H);
int mr_render_forward(const float *vertices, const float *transforms, const float *normals,
                      const float *diffuse, const int32_t *triangles,
                      const float *light_positions, const float *light_intensities,
                      const float *ambient, int B, int V, int T, int W, int H, int L,
                      float *clip, int32_t *ids, float *bary, float *z, int want_z, float *rgba,
                      uint8_t *rgba_u8, void *corner_records, void *backward_prepared, uint8_t *empty_regions,
                      void *workspace, size_t workspace_bytes, void *stream);

/* Backward of mr_shade_forward AND of the rasterizer underneath it, in one pass
 * over the G-buffer (reads 32 B/px).  All outputs are zeroed here.
 *   drgba        [B,H,W,4] f32  dL/d(rgba); the alpha channel's gradient is ignored
 *   clip         [B,V,4]   f32  the clip-space vertices the G-buffer was made from
 *   dclip        [B,V,4]   f32 out  through the barycentrics (column z stays 0)
 *   dnormals, dpositions, ddiffuse [B,V,3] f32 out  through the interpolated attributes.
 *                               dnormals and / or ddiffuse may be NULL (with the vertex adjacency):
 *                               that gradient is not wanted -- autograd's needs_input_grad -- and its
 *                               nine sums per triangle are not formed; without light_grads the pixel
 *                               pass then keeps the 18 or 27 remaining products in registers down each
 *                               lane's vertical run (k_accumulate_lanes) instead of reducing 36 per row
 *   light_grads  [B, 6L+3] f32 out  per image: d light_positions (L x 3),
 *                               d light_intensities (L x 3), d ambient (3; 0 if NULL).
 *                               NULL: not wanted -- an instantiation without the nine per-lane
 *                               accumulators runs (the pixel pass 4 % faster)
 *   corner_records  NULL, or the first mr_shade_forward_workspace_bytes() bytes of the workspace
 *                   mr_shade_forward ran with for the SAME inputs (128-byte aligned): its gathered
 *                   per-triangle attribute records are then reused instead of rebuilt.
 *   vertex_offsets, vertex_entries  both NULL, or the CSR vertex -> (triangle, corner) adjacency of
 *                   `triangles`: offsets [V+1] i32, entries [offsets[V]] i32 with entry = 3 *
 *                   triangle + corner, grouped by vertex (corners whose index is outside [0, V)
 *                   left out).  With it the per-triangle sums are GATHERED per vertex (no atomics,
 *                   fixed summation order); without it they are scattered with float atomics.
 *   transforms      NULL, or the [B,4,4] row-major clip-space transforms that `clip` was made with
 *                   (clip = transforms . (positions, 1), as mr_render_forward does): dpositions then
 *                   also receives the clip-space gradient pulled back through that product, i.e.
 *                   it becomes the whole gradient w.r.t. the world-space vertices; dclip is still
 *                   written (a caller that differentiates the transforms needs it).  Requires the
 *                   vertex adjacency.  With transforms, dclip may be NULL: the clip-space gradient is
 *                   not wanted on its own (the caller differentiates to the vertices, not to the
 *                   cameras).  Where the lane kernel has the variant (no light_grads, dnormals and
 *                   ddiffuse NULL, MR_GBUFFER_NORMALISED, not deterministic) the pull-back is then applied
 *                   per pixel and the pixel pass keeps 9 sums per triangle instead of 18; elsewhere the
 *                   clip gradient goes to the workspace.
 *   prepared        NULL, or the block mr_render_forward filled as `backward_prepared` for the SAME
 *                   inputs (256-byte aligned).  Used -- the setup launch skipped -- when the call is one
 *                   the folded kernel serves (dclip, dnormals, ddiffuse, light_grads NULL; transforms and
 *                   the adjacency given; MR_GBUFFER_NORMALISED; not deterministic), ignored otherwise.  The
 *                   block's accumulator rows are clear on entry and DIRTY afterwards: a block serves ONE
 *                   backward call that uses it (differentiating a second time: pass NULL -- the call then
 *                   runs its own setup kernel -- or clear the rows, the block's last
 *                   B * T * 48 bytes rounded up to 256).
 *   empty_regions   NULL, or mr_render_forward's `empty_regions` for the SAME G-buffer: the difference-basis lane
 *                   kernels leave a strip that lies inside an empty block without reading it.
 * dclip, dnormals, dpositions, ddiffuse, light_grads laid out back to back in that order are
 * zeroed with a single memset (none at all with the vertex adjacency: every output is written
 * exactly once). */
size_t mr_shade_backward_prepared_bytes(int B, int T);
size_t mr_shade_backward_workspace_bytes(int B, int V, int T, int W, int H);
int mr_shade_backward(const float *drgba, const int32_t *ids, const float *bary,
                      const float *clip, const float *normals, const float *positions,
                      const float *diffuse, const int32_t *triangles,
                      const float *light_positions, const float *light_intensities,
                      const float *ambient, int B, int V, int T, int W, int H, int L,
                      float *dclip, float *dnormals, float *dpositions,
                      float *ddiffuse, float *light_grads, const void *corner_records,
                      const int32_t *vertex_offsets, const int32_t *vertex_entries,
                      const float *transforms, int gbuffer_flags, void *prepared, const uint8_t *empty_regions,
                      void *workspace, size_t workspace_bytes,
                      void *stream);
/* gbuffer_flags, bit 0 = MR_GBUFFER_NORMALISED: the caller vouches that ids / bary are what
 * mr_rasterize_forward (or mr_render_forward) wrote for these very vertices -- every covered pixel's
 * barycentrics sum to 1 within rounding.  The coverage alpha = clamp(2 * sum) (rasterize.py:137-150) is then
 * exactly 1 and outside the clamp's pass band, and the pixel pass leaves out the blend with the background
 * and the d / d alpha terms: the same bits with ~20 % fewer vector instructions per pixel.  0 = any
 * G-buffer (e.g. one the caller edited). */
#define MR_GBUFFER_NORMALISED 1

/* mr_shade_backward for an upstream gradient that is the backward of mr_l1_loss_forward(rgba,
 * target): instead of the [B,H,W,4] float image mr_l1_loss_backward would write (16 B/px) it takes
 * the loss's packed sign codes (`signs`, one byte per pixel in image row order, exactly as
 * mr_l1_loss_forward wrote them for an image of B*H*W*4 elements) and `upstream`, the device scalar
 * d L / d loss.  Same outputs as mr_shade_backward(drgba = upstream * sign / (B*H*W*4)); the
 * workspace must be mr_shade_backward_l1_workspace_bytes() long. */
size_t mr_shade_backward_l1_workspace_bytes(int B, int V, int T, int W, int H);
int mr_shade_backward_l1(const uint8_t *signs, const float *upstream, const int32_t *ids,
                         const float *bary, const float *clip, const float *normals,
                         const float *positions, const float *diffuse, const int32_t *triangles,
                         const float *light_positions, const float *light_intensities,
                         const float *ambient, int B, int V, int T, int W, int H, int L,
                         float *dclip, float *dnormals, float *dpositions, float *ddiffuse,
                         float *light_grads, const void *corner_records,
                         const int32_t *vertex_offsets, const int32_t *vertex_entries,
                         const float *transforms, int gbuffer_flags, void *prepared, const uint8_t *empty_regions,
                         void *workspace, size_t workspace_bytes, void *stream);

/* ---- fused deferred shading with the specular term --------------------------------
 * The same replacement as mr_shade_forward / mr_shade_backward for render() calls that pass
 * specular_colors and shininess_coefficients (src/mesh_renderer/render.py:157-181, 199-228;
 * phong_shader's specular term :326-372, including its L2 normalisation of the reflection .
 * camera dot product ACROSS ALL PIXELS of an image, :342-348 -- one extra pass over the
 * G-buffer each way).  Additional arguments:
 *   specular         [B,V,3] f32  per-vertex specular colours
 *   camera_position  [B,3]   f32  world-space eye
 *   shininess        [B] f32 one exponent per image, or, with shininess_per_vertex != 0,
 *                    [B,V] f32 one per vertex (interpolated as a 13th attribute, render.py
 *                                 :171-181, 222-224)
 *   norms2           [B,L]   f32  forward: out, sum over all pixels of (reflection . camera)^2;
 *                                 backward: in, the forward's values
 *   dspecular        [B,V,3] f32 out
 *   dshininess       [B,V]   f32 out  d per-vertex shininess; used (and required) only with
 *                                 shininess_per_vertex != 0
 *   light_grads      [B, 6L+7] f32 out  per image: d light_positions (L x 3), d light_intensities
 *                                 (L x 3), d ambient (3; 0 if NULL), d camera_position (3),
 *                                 d per-image shininess (1; 0 with per-vertex exponents)
 *   vertex_offsets, vertex_entries (backward)  both NULL, or the CSR vertex adjacency of `triangles`
 *                                 exactly as for mr_shade_backward: the per-triangle sums are then
 *                                 gathered per vertex (no atomics, every output written once, fixed
 *                                 order) instead of scattered with float atomics; required in the
 *                                 deterministic mode
 *   transforms, gbuffer_flags, grads_wanted (backward, round 4)
 *                                 grads_wanted: MR_GRAD_* bits of the gradients the caller will read; an
 *                                 output whose bit is clear still has to be a valid buffer and holds
 *                                 unspecified values afterwards (MR_GRAD_ALL: everything, as before).
 *                                 When only MR_GRAD_POSITIONS / MR_GRAD_CLIP are wanted -- render()
 *                                 differentiated to the vertices alone -- and gbuffer_flags has
 *                                 MR_GBUFFER_NORMALISED (see mr_shade_backward) and the deterministic mode is
 *                                 off, the pixel pass keeps its 18 sums per triangle in registers down each
 *                                 lane's vertical run instead of reducing 45 through LDS per pixel
 *                                 (0.64 -> 0.3x ms at 1024^2 x 32).  transforms ([B,4,4], clip = M (position,
 *                                 1)) or NULL: with them and MR_GRAD_CLIP clear the pull-back M^T dclip is
 *                                 folded into dpositions, which is then the whole gradient w.r.t. the
 *                                 world-space vertices (9 sums).
 * Only pixels that pass render()'s mask (render.py:215) evaluate the power; the reference's
 * autograd multiplies the masked pixels' zero gradient by pow(0, -1) = inf of the background
 * exponent -1 and returns NaN for every per-vertex-shininess call with a background pixel. */
#define MR_GRAD_NORMALS 1
#define MR_GRAD_POSITIONS 2
#define MR_GRAD_DIFFUSE 4
#define MR_GRAD_SPECULAR 8
#define MR_GRAD_SHININESS 16   /* per-vertex (dshininess) or per-image (light_grads' last column) */
#define MR_GRAD_LIGHTS 32      /* light_grads: light positions / intensities, ambient, camera position */
#define MR_GRAD_CLIP 64
#define MR_GRAD_ALL 127
size_t mr_shade_specular_forward_workspace_bytes(int B, int V, int T, int W, int H);
int mr_shade_specular_forward(const int32_t *ids, const float *bary, const float *normals,
                              const float *positions, const float *diffuse, const float *specular,
                              const int32_t *triangles, const float *light_positions,
                              const float *light_intensities, const float *ambient,
                              const float *camera_position, const float *shininess,
                              int shininess_per_vertex, int B, int V, int T, int W, int H, int L,
                              float *rgba, float *norms2, int norms2_given,
                              void *workspace, size_t workspace_bytes, void *stream);
/* norms2_given = 1: norms2 is an INPUT -- what mr_rasterize_specular_norms_forward wrote for the same G-buffer, normals,
 * positions, lights and camera -- and the norm pass over the G-buffer does not run (0: norms2 is an output, as before).
 *
 * mr_rasterize_specular_norms_forward (round 5): mr_rasterize_forward on `clip` AND those norms in one pass over the
 * pixels.  The reference L2-normalises the reflection . camera dot product across ALL pixels of an image per light
 * (render.py:342-348); that sum only needs every covered pixel's interpolated normal and position, which the
 * rasterizer's tile walk holds when it has settled the pixel, and the uncovered pixels all carry the background's
 * attributes -1 (render.py:197) and one common value.  ids / bary / z exactly as mr_rasterize_forward writes them (z is
 * required as a buffer; want_z = 0: its contents are unspecified afterwards); norms2 [B,L] f32 out, 1 <= L <= 4. */
size_t mr_rasterize_specular_norms_workspace_bytes(int B, int V, int T, int W, int H);
int mr_rasterize_specular_norms_forward(const float *clip, const int32_t *triangles, const float *normals,
                                        const float *positions, const float *light_positions,
                                        const float *camera_position, int B, int V, int T, int W, int H, int L,
                                        int32_t *ids, float *bary, float *z, int want_z, float *norms2,
                                        void *workspace, size_t workspace_bytes, void *stream);
size_t mr_shade_specular_backward_workspace_bytes(int B, int V, int T, int W, int H);
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
                               int grads_wanted, void *workspace, size_t workspace_bytes, void *stream);
/* mr_shade_specular_backward for an upstream gradient that is the backward of mr_l1_loss_forward(rgba, target)
 * -- the reference's optimisation loop, mesh_renderer_test.py:238-262, on the specular renderer (render.py:224-246):
 * `signs` / `upstream` as in mr_shade_backward_l1, in place of drgba.  Same outputs as
 * mr_shade_specular_backward(drgba = upstream * sign / (B*H*W*4)).  The one-pass vertex-gradient kernel (one or two
 * lights, transforms given, only MR_GRAD_POSITIONS wanted, MR_GBUFFER_NORMALISED, adjacency given, deterministic mode
 * off) reads the codes directly, 1 B/px instead of 16; every other case writes the dense image into the workspace
 * first and runs mr_shade_specular_backward's kernels.  The workspace must be
 * mr_shade_specular_backward_l1_workspace_bytes() long. */
size_t mr_shade_specular_backward_l1_workspace_bytes(int B, int V, int T, int W, int H);
int mr_shade_specular_backward_l1(const uint8_t *signs, const float *upstream, const int32_t *ids,
                                  const float *bary, const float *clip, const float *normals,
                                  const float *positions, const float *diffuse, const float *specular,
                                  const int32_t *triangles, const float *light_positions,
                                  const float *light_intensities, const float *ambient,
                                  const float *camera_position, const float *shininess,
                                  int shininess_per_vertex, const float *norms2, int B, int V, int T,
                                  int W, int H, int L, float *dclip, float *dnormals, float *dpositions,
                                  float *ddiffuse, float *dspecular, float *dshininess,
                                  float *light_grads, const int32_t *vertex_offsets,
                                  const int32_t *vertex_entries, const float *transforms, int gbuffer_flags,
                                  int grads_wanted, void *workspace, size_t workspace_bytes, void *stream);

/* ---- SoftRas renderer ---------------------------------------------------------------
 * Replaces rasterize_batch / rasterize of the reference's second renderer
 * (src/soft_mesh_renderer/rasterize.py:14-110, 212-424) and the autograd graph behind it.
 *   clip               [B,V,4] f32   clip-space vertices
 *   positions, normals, diffuse [B,V,3] f32 (positions = world-space vertices)
 *   triangles          [T,3] i32     counter-clockwise = front-facing (back faces are culled)
 *   light_positions    [B,L,3] f32, light_intensities [B,L] f32 (scalar), 1 <= L <= mr_soft_max_lights()
 *   sigma, gamma, blur the reference's sigma_val, gamma_val, blur_radius (NDC units)
 *   rgba               [B,H,W,4] f32 out; row 0 is the TOP scanline; alpha = silhouette
 *   aux                [B,H,W,4] f32 out; per-pixel softmax state kept for the backward */
int mr_soft_max_lights(void);
size_t mr_soft_workspace_bytes(int B, int V, int T, int W, int H);
int mr_soft_forward(const float *clip, const float *positions, const float *normals,
                    const float *diffuse, const int32_t *triangles,
                    const float *light_positions, const float *light_intensities,
                    int B, int V, int T, int W, int H, int L,
                    float sigma, float gamma, float blur, float *rgba, float *aux,
                    void *workspace, size_t workspace_bytes, void *stream);
/* mr_soft_forward leaves its per-triangle records and candidate lists in the first
 * mr_soft_prepared_bytes() of its workspace.  A caller that keeps that workspace untouched until the
 * backward pass of the SAME inputs and sizes hands it over as `prepared` and the backward does not
 * rebuild them (three launches); prepared = NULL rebuilds them in `workspace`.  The forward itself is
 * content with a workspace of mr_soft_prepared_bytes(); the backward wants mr_soft_workspace_bytes(). */
size_t mr_soft_prepared_bytes(int B, int V, int T, int W, int H);
/* All gradient outputs are zeroed here.  drgba [B,H,W,4]; dclip [B,V,4];
 * dpositions / dnormals / ddiffuse [B,V,3]; dlight_positions [B,L,3]; dlight_intensities [B,L]. */
int mr_soft_backward(const float *drgba, const float *rgba, const float *aux,
                     const float *clip, const float *positions, const float *normals,
                     const float *diffuse, const int32_t *triangles,
                     const float *light_positions, const float *light_intensities,
                     int B, int V, int T, int W, int H, int L,
                     float sigma, float gamma, float blur,
                     float *dclip, float *dpositions, float *dnormals, float *ddiffuse,
                     float *dlight_positions, float *dlight_intensities, const void *prepared,
                     void *workspace, size_t workspace_bytes, void *stream);

/* ---- mean-absolute-error image loss --------------------------------------------------
 * The loss the reference's optimisation tests and examples use,
 * torch.mean(torch.abs(render - target)) (src/mesh_renderer/mesh_renderer_test.py:250),
 * as one streaming pass each way.  a, b: n floats (16-byte aligned); loss: 1 float out;
 * signs: (n + 3) / 4 bytes out, or NULL when no gradient is wanted -- byte i holds
 * sign(a - b) of elements 4i .. 4i+3 as 2-bit two's-complement codes (0: zero, 1: +1, 3: -1), so that the
 * backward pass does not read the images again; upstream: 1 float (dL/dloss, read on the
 * device); da: n floats out (16-byte aligned) = upstream * sign(a - b) / n.
 * partials: MR_L1_PARTIALS floats of scratch: the workgroups' partial sums, added in a fixed
 * order by one wavefront -- the loss value is bit-identical from run to run. */
#define MR_L1_PARTIALS 2048
/* The same loss for [B,H,W,4] images whose EMPTY 64 x 64 blocks are known (round 4): empty_a / empty_b are
 * mr_empty_regions_bytes(B, W, H) byte maps as mr_render_forward writes them for its image and
 * mr_image_empty_regions computes them for any image (once per target).  Where both say 1 the block is neither read
 * nor compared -- |0 - 0| = 0, sign codes 0 -- which for an object over an empty background is the share of the
 * frame it leaves free.  Same loss value up to the grouping of the sum (still a fixed order), same sign codes. */
int mr_image_empty_regions(const float *image, int B, int H, int W, uint8_t *map, void *stream);
int mr_l1_loss_forward_regions(const float *a, const float *b, int B, int H, int W, const uint8_t *empty_a,
                               const uint8_t *empty_b, float *loss, uint8_t *signs, float *partials, void *stream);
int mr_l1_loss_partials(void); /* MR_L1_PARTIALS of the library that was loaded (a binding without the header asks) */
int mr_l1_loss_forward(const float *a, const float *b, size_t n, float *loss, uint8_t *signs,
                       float *partials, void *stream);
int mr_l1_loss_backward(const uint8_t *signs, size_t n, const float *upstream, float *da,
                        void *stream);

/* ---- 8-bit frame export ---------------------------------------------------------------
 * The conversion the reference's examples apply on the host before writing PNG / GIF frames,
 * (image * 255.0).astype(np.uint8) (src/examples/example1.py:52, example5.py:81), on the device:
 * out[i] = (uint8) trunc(clamp(image[i], 0, 1) * 255), NaN -> 0.  image: n floats (16-byte
 * aligned); out: n bytes (4-byte aligned).  Used for frames that leave the GPU (display, files,
 * the multi-GPU hand-over): a quarter of the fp32 bytes. */
int mr_export_u8(const float *image, size_t n, uint8_t *out, void *stream);

/* ---- vertex normals ---------------------------------------------------------------------
 * Replaces compute_vertex_normals (src/common/meshes.py:3-35): for every triangle corner the
 * area-weighted face normal (next - this) x (next_next - this) is added to the corner's vertex and
 * the sums are normalised (eps 1e-6).  Gather form over the CSR vertex -> (triangle, corner)
 * adjacency of `triangles` (vertex_offsets [V+1], vertex_entries, entry = 3 * triangle + corner,
 * see mr_shade_backward): no atomics, fixed summation order.  Triangles with a vertex id outside
 * [0, V) are skipped.
 *   vertices [B,V,3] f32;  sums [B,V,3] f32 out (the un-normalised sums, input of the backward);
 *   normals [B,V,3] f32 out;  backward: dnormals in, dvertices [B,V,3] out. */
int mr_vertex_normals_forward(const float *vertices, const int32_t *triangles,
                              const int32_t *vertex_offsets, const int32_t *vertex_entries, int B, int V,
                              int T, float *sums, float *normals, void *stream);
int mr_vertex_normals_backward(const float *dnormals, const float *vertices, const float *sums,
                               const int32_t *triangles, const int32_t *vertex_offsets,
                               const int32_t *vertex_entries, int B, int V, int T, float *dvertices,
                               void *stream);

/* ---- clip-space transforms --------------------------------------------------------------
 * perspective(aspect, fov_y, near, far) . look_at(eye, center, up) per image, the product render() and
 * rasterize() apply to the vertices (src/common/camera_utils.py:45-139; src/mesh_renderer/render.py
 * :187-193), as one launch on cameras that live on the device, and its backward to eye / center / up
 * (one launch; fov_y, near, far are not differentiated).
 *   eye, center, up [B,3] f32; fov_y (degrees), near_clip, far_clip [B] f32; aspect = width / height
 *   transforms [B,4,4] f32 out, row-major
 *   degenerate  1 int32 out (device): bit 0 = some |center - eye| <= 1e-6, bit 1 = some
 *               |forward x up| <= 1e-6 -- the two conditions the reference asserts on
 *               (camera_utils.py:68-69, 74-76); the caller decides when to read it back
 *   dtransforms [B,4,4] f32; deye, dcenter, dup [B,3] f32 out */
int mr_camera_transforms(const float *eye, const float *center, const float *up, const float *fov_y,
                         const float *near_clip, const float *far_clip, float aspect, int B, float *transforms,
                         int32_t *degenerate, void *stream);
int mr_camera_transforms_backward(const float *dtransforms, const float *eye, const float *center,
                                  const float *up, const float *fov_y, const float *near_clip,
                                  const float *far_clip, float aspect, int B, float *deye, float *dcenter,
                                  float *dup, void *stream);

/* ---- tone_mapper ------------------------------------------------------------------------
 * Replaces tone_mapper (src/mesh_renderer/render.py:389-419): per image,
 *   out = clamp(image^gamma / max(image^gamma), 0, 1)     (torch.pow / torch.max / torch.clamp
 * semantics, NaNs included).  image: B images of `elements_per_image` floats each; max_scratch: B
 * ints (device, overwritten: the per-image maxima as float bits); exactly one of out_f32 (same
 * shape as image) and out_u8 (8-bit frames, the examples' `(x * 255).astype(np.uint8)` applied to
 * the tone-mapped value) is non-NULL. */
int mr_tone_map(const float *image, int B, size_t elements_per_image, float gamma, int32_t *max_scratch,
                float *out_f32, uint8_t *out_u8, void *stream);

/* ---- deterministic mode ------------------------------------------------------------------
 * The reference accumulates its gradients sequentially (rasterize_triangles.cpp:156-157, 232-269) and
 * is bit-reproducible; the default kernels here end in float atomics, whose sums depend on the order
 * in which wavefronts commit (last-bit differences from run to run).  mr_set_deterministic(1) makes
 * the calling thread's later mr_shade_backward / mr_shade_backward_l1 / mr_rasterize_backward calls
 * accumulate in 64-bit FIXED POINT with integer atomics instead -- integer addition is associative,
 * so the result is independent of that order -- followed by fixed-order per-vertex gathers.  The
 * fixed-point scale is a power of two derived on the device from the largest upstream gradient g:
 * contributions below g * 2^-42 are rounded away, a per-triangle total may reach g * 2^21.  About
 * 10 % slower (8-byte atomics, one extra pass to find g when the upstream is a dense image).
 * mr_shade_backward needs the vertex adjacency in this mode (MR_EINVAL without).  Round 3 extends the
 * mode to mr_interpolate_raster_backward and mr_shade_specular_backward (the latter needs the
 * adjacency too).  A contribution that does not fit the fixed-point range (NaN, infinite, beyond
 * 2^63 after scaling) raises a flag and the outputs of that call are NaN instead of a finite wrong
 * number.  mr_soft_backward is covered as well (fixed-point integer atomics into 64-bit copies of its
 * four vertex outputs, scaled for the 1 / sigma and 1 / gamma its contributions carry; its light
 * gradients are fixed-order sums in either mode).  Not covered, float atomics remain: the composed
 * interpolation backward (mr_interpolate_backward, the path for more than 16 attributes).  mr_l1_loss_forward is always deterministic.  Returns the previous setting. */
int mr_set_deterministic(int on);

/* ---- kernel timing (measurement, no reference counterpart) ----------------------------
 * Arms ONE measurement: the next launch of the named kernel made BY THE CALLING THREAD records
 * hipEvent `start_event` immediately before and `stop_event` immediately after that kernel on
 * the call's stream, then the slot clears itself (one-shot, thread-local: a forgotten or a
 * concurrent caller's timer cannot attach to somebody else's launch).  bench.py's roofline legs
 * time single kernels inside a full step this way.  Results are unaffected.  NULL, NULL disarms. */
#define MR_TIMER_RASTER_FORWARD 0  /* k_raster: the forward G-buffer kernel of mr_rasterize_forward   */
#define MR_TIMER_SHADE_BACKWARD 1  /* the pixel pass of mr_shade_backward (fused shading backward)    */
#define MR_TIMER_SHADE_FORWARD 2   /* the pixel pass of mr_shade_forward                              */
#define MR_TIMER_RASTER_BACKWARD 3 /* the pixel pass of mr_rasterize_backward                         */
#define MR_TIMER_L1_FORWARD 4      /* the streaming pass of mr_l1_loss_forward                        */
#define MR_TIMER_COUNT 5
int mr_time_next_kernel(int which, void *start_event, void *stop_event);

#ifdef __cplusplus
}
#endif
#endif /* MESH_RASTER_H_ */
