// Fused deferred shading for gfx950 (MI355X): attribute interpolation + Phong,
// forward and backward, straight from the G-buffer.
//
// Replaces, for the diffuse + ambient path of mesh_renderer.render():
//   * the corner gather / multiply / sum / alpha / background blend of
//     rasterize_clip_space  (reference: src/mesh_renderer/rasterize.py:118-150),
//   * the attribute unpacking, normalisation and mask of render()
//     (src/mesh_renderer/render.py:199-215),
//   * phong_shader's ambient and diffuse terms, the alpha mask and the vertical flip
//     (src/mesh_renderer/render.py:287-323, 373-386),
// and the whole autograd graph the reference builds behind them.  In the reference
// these are ~40 eager ops over [B, L, H*W, 3] tensors (3.6 GB of gathered corners at
// 1024^2 x 32); here the forward reads 16 B/px (id + barycentrics) and writes 16 B/px
// (RGBA), and the backward reads 32 B/px (dRGBA + id + barycentrics).
//
//   k_shade_forward    one thread per pixel; the triangle's corner attributes come from one
//                      128-byte record per (image, triangle) (corner_rec.h).  render() itself
//                      no longer launches it: the same shading (shade_pixel.h) runs as the
//                      epilogue of k_raster's tile walk (raster_forward.hip, mr_render_forward).
//   ShadeGradFn        per-pixel backward evaluated inside k_accumulate_rows (run_accum.h):
//                      recomputes the pixel's shading, back-propagates to the interpolated
//                      normal / position / diffuse colour, to the barycentrics and -- through
//                      rasterize_triangles.cpp:202-269 -- to the clip-space corners; the 36
//                      sums per triangle (27 attribute + 9 clip) are products of 15 per-pixel
//                      factors formed in the row reduction, light / ambient gradients are
//                      per-lane sums reduced once per wave.  The upstream gradient is a dense
//                      image or, for the L1 loss, its 1 B/px sign codes.
//   k_shade_gather     sixteen lanes per (image, vertex): sums the rows of the incident
//                      triangles over the CSR adjacency (no atomics) and, given the clip-space
//                      transforms, pulls d clip back onto the world-space vertices.
//   k_shade_scatter    without an adjacency: one thread per touched (image, triangle),
//                      atomics into dnormals / dpositions / ddiffuse [B,V,3] and dclip [B,V,4].
#include "shade_pixel.h"

namespace mr {
namespace {

constexpr int kThreads = 256;
constexpr float kDegenerateCutoff = 0.9f;  // rasterize_triangles.cpp:13

__global__ __launch_bounds__(kThreads) void k_corner_setup(
    const F3 *__restrict__ normals, const F3 *__restrict__ positions, const F3 *__restrict__ diffuse,
    const int32_t *__restrict__ tris, int B, int V, int T, CornerRec *__restrict__ out) {
  const long gid = (long)blockIdx.x * kThreads + threadIdx.x;
  if (gid >= (long)B * T) return;
  const int b = (int)(gid / T);
  const int t = (int)(gid - (long)b * T);
  fill_corner_record(normals, positions, diffuse, tris, b, t, V, out + gid);
}

// One workgroup = 256 consecutive pixels of a row segment x kRows consecutive rows; each
// thread shades kRows pixels of one column.  All G-buffer loads of the kRows pixels are
// issued first, then all corner-record loads, then the arithmetic: the kernel is bound
// by load latency (two dependent levels: id -> corner record), not by bandwidth or VALU.
#ifndef MR_SHADE_ROWS
#define MR_SHADE_ROWS 2  // measured: 1 -> 0.290, 2 -> 0.264, 3 -> 0.282, 4 -> 0.293 ms (C3)
#endif
#ifndef MR_SHADE_FWD_WAVES
#define MR_SHADE_FWD_WAVES 1
#endif
constexpr int kShadeRows = MR_SHADE_ROWS;

__global__ __launch_bounds__(kThreads, MR_SHADE_FWD_WAVES) void k_shade_forward(
    const int32_t *__restrict__ ids, const F3 *__restrict__ bary,
    const CornerRec *__restrict__ corners, Lights lights, int B, int T, int W, int H,
    int x_blocks, int y_blocks, float4 *__restrict__ out) {
  const int blk = (int)blockIdx.x;
  const int img = blk / (x_blocks * y_blocks);
  const int rem = blk - img * (x_blocks * y_blocks);
  const int yb = rem / x_blocks, xb = rem - yb * x_blocks;
  const int x = xb * kThreads + (int)threadIdx.x;
  if (x >= W) return;
  const int y0 = yb * kShadeRows;
  F3 b[kShadeRows];
  int t[kShadeRows];
  bool live[kShadeRows];
#pragma unroll
  for (int r = 0; r < kShadeRows; ++r) {
    const int y = min(y0 + r, H - 1);
    const size_t pix = ((size_t)img * H + y) * W + x;
    b[r] = load_streamed(&bary[pix]);
    t[r] = __builtin_nontemporal_load(&ids[pix]);
  }
  Corners cr[kShadeRows];
#pragma unroll
  for (int r = 0; r < kShadeRows; ++r) {
    // alpha == 0: every attribute is the -1 background, mask = 0 -> transparent black
    live[r] = ((2.0f * b[r].x + 2.0f * b[r].y) + 2.0f * b[r].z) > 0.0f;
    if ((unsigned)t[r] >= (unsigned)T) t[r] = 0;
    if (live[r]) load_corners(corners + (size_t)img * T + t[r], cr[r]);
  }
#pragma unroll
  for (int r = 0; r < kShadeRows; ++r) {
    const int y = y0 + r;
    if (y >= H) break;
    const float4 rgba = live[r] ? shade_pixel(cr[r], b[r], lights, img) : make_float4(0.f, 0.f, 0.f, 0.f);
    // render.py:384-386: the image is flipped vertically (G-buffer row 0 is the bottom)
    typedef float v4f __attribute__((ext_vector_type(4)));
    __builtin_nontemporal_store(v4f{rgba.x, rgba.y, rgba.z, rgba.w}, (v4f *)&out[((size_t)img * H + (H - 1 - y)) * W + x]);
  }
}

// SIGNS: the upstream gradient is the backward of mean|image - target| (loss.hip): it arrives as
// the 2-bit sign codes that loss's forward packed (1 B/px, image rows) and one device scalar
// instead of a [B,H,W,4] float image (16 B/px that k_l1_backward would first have to write).
// LG: the caller wants the light / ambient gradients (false: light_grads == nullptr; their nine
// per-lane accumulators and ~25 instructions per row are not compiled in: the kernel -4 %).
// L = 1..4: that many lights, kept in registers and unrolled.  L = 0 (round 3): any count up to
// kMaxLightsAny through a run-time loop, each light read through the scalar cache per pixel row (no
// per-lane copies); without the light gradients (LG) -- their 6 L per-lane sums do not scale: the
// caller forms them four lights at a time (pytorch_mesh_renderer_amd/_native.py, shade_backward).
template <int L, bool SIGNS, bool LG>
struct ShadeGradFn {
  static_assert(L >= 0 && L <= kMaxLights && (L > 0 || !LG), "lights in registers: 1..4; L = 0 = run-time count, no light gradients");
  static constexpr int LA = L > 0 ? L : 1;  // array extents
  static constexpr int kN = 36;       // 27 attribute partials [corner][attr] + 9 clip partials
  static constexpr int kStride = 36;
  // see run_accum.h.  With the light gradients every strip ends in 6L + 3 atomics on the image's
  // one set of sums: 8-row strips put four times as many of them on the same addresses (dense
  // upstream, 1024^2 x 32: 8 -> 0.84, 16 -> 0.56, 32 -> 0.56 ms)
  static constexpr int kRowsPerWave = LG ? 16 : MR_ROWS_PER_WAVE;
  // per-pixel factors parked in LDS: b[3] | y[9] = alpha * d/d attr | q[3] = clip brackets
  static constexpr int kFactors = 15;
  static constexpr int kFactorStride = 20;
  // sum o = corner * 9 + attr      -> b[corner] * y[attr]
  //     o = 27 + corner * 3 + comp -> b[corner] * q[comp]
  __device__ static void factor_pair(int o, int &ia, int &ib) {
    if (o < 27) { ia = o / 9; ib = 3 + o % 9; }
    else { ia = (o - 27) / 3; ib = 12 + (o - 27) % 3; }
  }
  static constexpr int kSlots = 256;
#ifndef MR_SHADE_WAVES
#define MR_SHADE_WAVES 3
#endif
  static constexpr int kMinWavesPerSimd = MR_SHADE_WAVES;
  static constexpr bool kCountBackground = false;
  const float4 *__restrict__ drgba;   // [B,H,W,4], image rows (flipped w.r.t. the G-buffer); !SIGNS
  const uint8_t *__restrict__ signs;  // [B,H,W] bytes: sign codes of the 4 channels; SIGNS
  const float *__restrict__ sign_upstream;  // SIGNS: d loss / d mean, one device float ...
  float sign_inv_n;                         // ... and 1 / element count of the mean
  const int32_t *__restrict__ ids;
  const F3 *__restrict__ bary;
  const CornerRec *__restrict__ corners;
  const BwdRec *__restrict__ recs;
  Lights lights;
  float *__restrict__ light_rows;     // LG: [strips][L*6 + 3] per strip: dpos (L x 3), dcol (L x 3), dambient (3)
  int T_, W, H;
  const float *__restrict__ transforms = nullptr;  // [B,4,4] clip = M (position, 1): ShadeLaneFn<..., FOLD> only
  // mr_render_forward's empty_regions ([B][ceil(H / 64)][ceil(W / 64)], 1 = a whole 64 x 64 block of background) or nullptr:
  // the difference-basis lane kernels leave a strip that lies in such a block without reading it (skip_strip)
  const uint8_t *__restrict__ empty_map = nullptr;
  __device__ __forceinline__ bool strip_is_empty(int img, int rx, int y_begin, int y_end) const {
    if (!empty_map) return false;
    const int by0 = y_begin >> 6, by1 = (y_end - 1) >> 6;   // (a strip is 64 pixels wide: block column = strip column)
    if (by0 != by1) return false;
    const int blocks_x = (W + 63) >> 6, blocks_y = (H + 63) >> 6;
    return empty_map[((size_t)img * blocks_y + by0) * blocks_x + rx] != 0;
  }

  struct Pixel {
    F3 b, g;
    int tri;
  };
  struct Raw {
    F3 b;
    int t;
    float4 g;       // !SIGNS
    unsigned code;  // SIGNS
  };
  struct Triangle {
    Corners cr;
    BwdTriangle bt;
  };
  struct Image {
    int n_bg;                              // unused (kCountBackground = false)
    int img;                               // L = 0: where this image's lights start
    float g_scale;                         // SIGNS: upstream * 1 / n
    float lp[LA][3], li[LA][3], amb[3];    // this image's lights (loaded once per lane; L > 0)
    float dpos[LA][3], dcol[LA][3], damb[3];  // per-lane partial sums
    float pull[3][3];                      // FOLD: pull[r][c] = M[{0, 1, 3}[r]][c], the clip x / y / w rows' position columns
  };

  __device__ __forceinline__ void begin_image(int img, Image &im) const {
    im.g_scale = SIGNS ? sign_upstream[0] * sign_inv_n : 0.f;
    im.img = img;
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int c = 0; c < 3; ++c)   // (wave-uniform scalar loads; dead code unless a FOLD functor reads them)
        im.pull[r][c] = transforms ? transforms[(size_t)img * 16 + (r == 2 ? 3 : r) * 4 + c] : 0.f;
#pragma unroll
    for (int l = 0; l < L; ++l) {
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        im.lp[l][c] = lights.pos[((size_t)img * L + l) * 3 + c];
        im.li[l][c] = lights.col[((size_t)img * L + l) * 3 + c];
        im.dpos[l][c] = 0.f;
        im.dcol[l][c] = 0.f;
      }
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      im.amb[c] = lights.amb ? lights.amb[(size_t)img * 3 + c] : 0.f;
      im.damb[c] = 0.f;
    }
  }

  __device__ __forceinline__ void fetch(int img, int x, int y, size_t pix, Raw &r) const {
    // Round 4: wave-uniform image bases (scalar registers) + 32-bit per-lane offsets inside the image --
    // global_load's saddr form -- instead of a 64-bit address per plane and row (12 of the row loop's
    // ~340 vector instructions were 64-bit multiply-adds forming them).  W * H < 2^27 pixels per image:
    // the largest byte offset, 16 W H, stays below 2^31.
    (void)pix;
    const size_t img_px = (size_t)img * H * W;                         // wave-uniform
    const unsigned gpix = (unsigned)(y * W) + (unsigned)x;             // G-buffer row y
    const unsigned ipix = (unsigned)((H - 1 - y) * W) + (unsigned)x;   // image row (un-flipped)
    const char *bary_img = (const char *)(bary + img_px), *ids_img = (const char *)(ids + img_px);
    r.b = load_streamed((const F3 *)(bary_img + gpix * 12u));
    r.t = __builtin_nontemporal_load((const int32_t *)(ids_img + gpix * 4u));
    if (SIGNS) r.code = __builtin_nontemporal_load(signs + img_px + ipix);
    else r.g = *(const float4 *)((const char *)(drgba + img_px) + ipix * 16u);
  }
  __device__ __forceinline__ bool prepare(const Raw &r, int T, int &tri, Pixel &p) const {
    const float pre = (2.0f * r.b.x + 2.0f * r.b.y) + 2.0f * r.b.z;
    if (!(pre > 0.0f)) return false;  // background: mask = 0, no gradient anywhere
    if ((unsigned)r.t >= (unsigned)T) return false;
    p.b = r.b;
    if (SIGNS) {  // 2-bit two's-complement codes 0, +1, -1 (loss.hip: sign_code), times the scale in factors():
                  // one v_bfe_i32 and one conversion per channel
      p.g.x = (float)(int)__builtin_amdgcn_sbfe(r.code, 0u, 2u);
      p.g.y = (float)(int)__builtin_amdgcn_sbfe(r.code, 2u, 2u);
      p.g.z = (float)(int)__builtin_amdgcn_sbfe(r.code, 4u, 2u);
    } else {
      p.g.x = r.g.x; p.g.y = r.g.y; p.g.z = r.g.z;  // d/d alpha is dropped: the mask is not differentiable
    }
    p.tri = r.t;
    tri = r.t;
    return true;
  }

  __device__ __forceinline__ void load_triangle(int img, int tri, Triangle &t) const {
    load_corners(corners + (size_t)img * T_ + tri, t.cr);
    load_bwd_triangle(recs + (size_t)img * T_ + tri, t.bt);
  }

  // The shading's backward at one pixel: from the interpolated attributes at[9] = (normal, position, diffuse
  // colour) and the upstream gradient g of the pixel's RGB to dat[9] = d L / d at (render.py:199-228, 287-323
  // differentiated by hand); the light / ambient gradients go to the per-lane sums of `im` (LG).
  // UNSCALED (sign-coded upstream only): g stays the bare sign (-1, 0, +1); every output is linear in g, so the
  // caller's consumer multiplies the finished sums by the loss's scale once per vertex instead of once per pixel.
  template <bool UNSCALED = false>
  __device__ __forceinline__ void attribute_gradients(const float (&at)[9], const F3 &pg, Image &im,
                                                      float (&dat_out)[9]) const {
    // render.py:215 mask: where() sends no gradient to a masked pixel.  All 36 outputs are
    // linear in g, so a masked pixel simply runs with g = 0 (every output must be assigned).
    const bool mask = (at[6] >= 0.0f) || (at[7] >= 0.0f) || (at[8] >= 0.0f);

    const float gs = (SIGNS && !UNSCALED) ? im.g_scale : 1.0f;
    const float g[3] = {mask ? pg.x * gs : 0.f, mask ? pg.y * gs : 0.f, mask ? pg.z * gs : 0.f};
    const float nn2 = at[0] * at[0] + at[1] * at[1] + at[2] * at[2];
    const float inv_nn = inv_norm(nn2);
    const float N[3] = {at[0] * inv_nn, at[1] * inv_nn, at[2] * inv_nn};
    float dN[3] = {0.f, 0.f, 0.f}, dP[3] = {0.f, 0.f, 0.f};
    float nd = 0.f;   // N . dN, kept as a running scalar (round 5): dN = sum_l t_l D_l, so N . dN = sum_l t_l (N . D_l)
    float dKd[3] = {g[0] * im.amb[0], g[1] * im.amb[1], g[2] * im.amb[2]};
#pragma unroll
    for (int c = 0; c < 3; ++c) if (LG) im.damb[c] += g[c] * at[6 + c];
    const int n_lights = L > 0 ? L : lights.L;  // (L > 0: a constant, the loop unrolls)
    ConstFloats all_pos = (ConstFloats)(uintptr_t)(lights.pos + (size_t)im.img * lights.L * 3);
    ConstFloats all_col = (ConstFloats)(uintptr_t)(lights.col + (size_t)im.img * lights.L * 3);
#pragma unroll
    for (int l = 0; l < n_lights; ++l) {
      float lpos[3], lcol[3];
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        lpos[c] = L > 0 ? im.lp[L > 0 ? l : 0][c] : all_pos[3 * l + c];   // L = 0: wave-uniform scalar loads
        lcol[c] = L > 0 ? im.li[L > 0 ? l : 0][c] : all_col[3 * l + c];
      }
      const float v[3] = {lpos[0] - at[3], lpos[1] - at[4], lpos[2] - at[5]};
      const float vn2 = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
      const float inv_vn = inv_norm(vn2);
      const float D[3] = {v[0] * inv_vn, v[1] * inv_vn, v[2] * inv_vn};
      const float pre_l = N[0] * D[0] + N[1] * D[1] + N[2] * D[2];
      const float ndl = fminf(fmaxf(pre_l, 0.0f), 1.0f);
      float t_l = 0.f;  // d/d ndl
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        dKd[c] += g[c] * ndl * lcol[c];
        if (LG) im.dcol[L > 0 ? l : 0][c] += g[c] * at[6 + c] * ndl;
        t_l += g[c] * at[6 + c] * lcol[c];
      }
      if (pre_l >= 0.0f && pre_l <= 1.0f) {  // torch.clamp passes the gradient inclusively
        // d ndl / d N = D and d ndl / d D = N, so both projections' dot products are t_l (N . D) = t_l pre_l:
        // the backward of v / max(|v|, eps) is (dD - D (D . dD)) / |v| = t_l (N - D pre_l) / |v|
        const float tp = t_l * pre_l;
        nd += tp;
        const float k = t_l * inv_vn;
        const bool unit = vn2 > kNormEpsSquared;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          dN[c] += t_l * D[c];
          const float dv = unit ? k * (N[c] - D[c] * pre_l) : k * N[c];
          if (LG) im.dpos[L > 0 ? l : 0][c] += dv;
          dP[c] -= dv;
        }
      }
    }
    float dat[9];
    {
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        dat[c] = (nn2 > kNormEpsSquared ? (dN[c] - N[c] * nd) : dN[c]) * inv_nn;
        dat[3 + c] = dP[c];
        dat[6 + c] = dKd[c];
      }
    }
#pragma unroll
    for (int a = 0; a < 9; ++a) dat_out[a] = dat[a];
  }

  __device__ __forceinline__ void factors(const Pixel &p, const Triangle &t, float (&f)[kFactorStride],
                                          Image &im) const {
    factors_of<false>(p, t, f, im);
  }
  // OPAQUE: the caller vouches that every covered pixel's barycentrics sum to more than 1/2, as those
  // mr_rasterize_forward writes do (they are normalised: the sum is 1 to a few ulp).  Then alpha = clamp(2 sum)
  // is exactly 1 and sits outside the clamp's pass band: attr = 1 * interp + 0 * (-1) = interp, d/d attr =
  // 1 * dat, and nothing flows through alpha -- the same bits as the general path with ~45 of its ~235 vector
  // instructions per pixel gone (the blend, the d/d alpha dot product, its reciprocal).
  template <bool OPAQUE>
  __device__ __forceinline__ void factors_of(const Pixel &p, const Triangle &t, float (&f)[kFactorStride],
                                             Image &im) const {
    float pre, alpha, interp[9], at[9];
    if (OPAQUE) {
      pre = 2.0f;
      alpha = 1.0f;
#pragma unroll
      for (int a = 0; a < 9; ++a) at[a] = (t.cr.c[0][a] * p.b.x + t.cr.c[1][a] * p.b.y) + t.cr.c[2][a] * p.b.z;
    } else {
      interpolate9(t.cr, p.b, pre, alpha, interp, at);
    }
    float dat[9];
    attribute_gradients(at, p.g, im, dat);
    // interpolation backward (rasterize.py:137-150)
    // d/d alpha = sum_a dat[a] * (interp[a] + 1) with interp[a] + 1 = (at[a] + 1) / alpha
    // (alpha > 0 on every valid pixel): `interp` need not stay live across the shading math.
    float dalpha_a = 0.f, db[3] = {0.f, 0.f, 0.f};
    f[0] = p.b.x; f[1] = p.b.y; f[2] = p.b.z;
#pragma unroll
    for (int a = 0; a < 9; ++a) {
      const float di = OPAQUE ? dat[a] : alpha * dat[a];
      if (!OPAQUE) dalpha_a += dat[a] * (at[a] + 1.0f);  // background is -1
#pragma unroll
      for (int k = 0; k < 3; ++k) db[k] += di * t.cr.c[k][a];
      f[3 + a] = di;  // d/d attr[corner k][a] = b_k * di: the product is formed in the reduction
    }
    const float dpre = (!OPAQUE && pre >= 0.0f && pre <= 1.0f) ? 2.0f * dalpha_a * fast_rcp(alpha) : 0.0f;
    F3 dbary;
    dbary.x = db[0] + dpre; dbary.y = db[1] + dpre; dbary.z = db[2] + dpre;
    // rasterizer backward (cpp:162 skip rule, then cpp:202-269)
    const bool skip = p.tri == 0 && (p.b.x + p.b.y) + p.b.z < kDegenerateCutoff;
    float q[3];
    raster_pixel_q(p.b, dbary, t.bt, skip ? 0.f : t.bt.inv, q);
    f[12] = q[0]; f[13] = q[1]; f[14] = q[2];
#pragma unroll
    for (int k = kFactors; k < kFactorStride; ++k) f[k] = 0.f;
  }

  // the strip's row of light sums (k_sum_strip_rows adds an image's rows up, in a fixed order)
  __device__ __forceinline__ void end_strip(int img, int strip, Image &im) const {
    (void)img;
    if (!LG) return;
    float *dst = light_rows + (size_t)strip * (L * 6 + 3);
    const int lane = lane_id();
    auto reduce_add = [&](float v, int slot) {
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);  // fixed tree: deterministic
      if (lane == 0) dst[slot] = v;
    };
#pragma unroll
    for (int l = 0; l < L; ++l) {
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        reduce_add(im.dpos[l][c], l * 3 + c);
        reduce_add(im.dcol[l][c], L * 3 + l * 3 + c);
      }
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) reduce_add(lights.amb ? im.damb[c] : 0.0f, L * 6 + c);
  }
};

// The same backward for k_accumulate_lanes (run_accum.h): the lane multiplies its pixel's factors out
// and keeps the products in registers down its vertical run.  GROUPS selects the attribute gradients
// the caller wants (bit 0 normals, bit 1 positions, bit 2 diffuse colours; the nine clip-space sums
// always): render() differentiates to whatever requires grad -- in the optimisation loops of the
// reference's tests and examples the vertices alone (9 + 9 sums) or vertices and normals (27).
// Factors nobody multiplies are dead code to the compiler.
#ifndef MR_LANE_ROWS
#define MR_LANE_ROWS 16
#endif
#ifndef MR_SHADE_LANES_FOLD
#define MR_SHADE_LANES_FOLD 1   // see ShadeLaneFn<..., FOLD>
#endif
#ifndef MR_LANE_ROWS_LG
#define MR_LANE_ROWS_LG 16
#endif
#ifndef MR_SHADE_LANES_LG
#define MR_SHADE_LANES_LG 1
#endif
#ifndef MR_LANE_WAVES
#define MR_LANE_WAVES 4
#endif
// FOLD (round 4): the caller has the clip-space transforms M (clip = M (position, 1)) and does not want the
// clip-space gradient on its own -- render() differentiated to the vertices, not to the cameras.  The pull-back
// d position += M^T d clip is linear, so it is applied per PIXEL to the three clip brackets q and added to the
// position attribute's gradient before the outer product with the barycentrics: 9 sums per triangle fewer to
// keep in registers, to restart, to park, to merge and to commit (18 -> 9 for vertex gradients alone), for nine
// multiply-adds with scalar operands; the gather then finds zeros in the clip columns.
template <int L, bool SIGNS, bool LG, int GROUPS, bool OPAQUE = false, bool FOLD = false>
struct ShadeLaneFn : ShadeGradFn<L, SIGNS, LG> {
  using Base = ShadeGradFn<L, SIGNS, LG>;
  static_assert(GROUPS >= 0 && GROUPS < 8, "attribute groups: normals | positions | diffuse");
  static_assert(!FOLD || (GROUPS & 2), "folding the clip gradient needs the position group");
  static constexpr int kGroups = (GROUPS & 1) + ((GROUPS >> 1) & 1) + ((GROUPS >> 2) & 1);
  static constexpr int kN = 9 * kGroups + (FOLD ? 0 : 9);
  static constexpr int kStride = 36;  // the rows of acc keep ShadeGradFn's layout: the gather reads it
  static constexpr int kLaneRowsPerWave = LG ? MR_LANE_ROWS_LG : MR_LANE_ROWS;
  // 36 accumulators: 137-145 VGPRs; with light gradients 6 L + 3 more per-lane sums ride along
  static constexpr int kMinWavesPerSimd = LG ? 3 : (kN > 27 ? 3 : MR_LANE_WAVES);   // LG: 134-161 VGPRs
  // the gi-th selected group
  __host__ __device__ static constexpr int group(int gi) {
    int g = 0;
    for (int seen = 0; g < 3; ++g) {
      if ((GROUPS >> g) & 1) {
        if (seen == gi) break;
        ++seen;
      }
    }
    return g;
  }
  // sum o = (gi * 3 + corner) * 3 + c  -> b[corner] * y[group * 3 + c],  acc column corner * 9 + group * 3 + c
  //     o = 9 * kGroups + corner * 3 + c -> b[corner] * q[c],            acc column 27 + corner * 3 + c
  __device__ static int column(int o) {
    if (o >= 9 * kGroups) return 27 + (o - 9 * kGroups);
    const int gi = o / 9, k = (o % 9) / 3, c = o % 3;
    const int g = gi == 0 ? group(0) : gi == 1 ? group(1) : group(2);
    return k * 9 + g * 3 + c;
  }
  __device__ __forceinline__ void accumulate(const typename Base::Pixel &p, const typename Base::Triangle &t,
                                             float (&a)[kN], typename Base::Image &im) const {
    float f[Base::kFactorStride];
    Base::template factors_of<OPAQUE>(p, t, f, im);
    if (FOLD) {
#pragma unroll
      for (int c = 0; c < 3; ++c)
        f[3 + 3 + c] += (im.pull[0][c] * f[12] + im.pull[1][c] * f[13]) + im.pull[2][c] * f[14];
    }
#pragma unroll
    for (int gi = 0; gi < kGroups; ++gi)
#pragma unroll
      for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int c = 0; c < 3; ++c) a[(gi * 3 + k) * 3 + c] = fmaf(f[k], f[3 + group(gi) * 3 + c], a[(gi * 3 + k) * 3 + c]);
    if (!FOLD) {
#pragma unroll
      for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int c = 0; c < 3; ++c) a[9 * kGroups + k * 3 + c] = fmaf(f[k], f[12 + c], a[9 * kGroups + k * 3 + c]);
    }
  }
};

// The folded variant's functor proper (round 4): vertex gradients only, G-buffer normalised, no light
// gradients -- the benchmark's and every vertex-optimisation loop's backward.  Per pixel and lane:
//   at = c2 + b0 e0 + b1 e1                       (FoldRec's difference basis: 18 multiply-adds, not 27)
//   dat = attribute_gradients(at, g)              (the shading's own backward, shared with every other variant)
//   g0 = dat . e0, g1 = dat . e1                  (d L / d b0 - d L / d b2 and d L / d b1 - d L / d b2: 18, not 27)
//   q_c = (g0 (s_c b0 - u_0c) + g1 (s_c b1 - u_1c)) / |det|     (cpp:202-269 with the common shift d L / d b2
//                                                  taken out: its brackets sum to s_c (sum b - 1) ~ 0 over the corners)
//   y_c = dat[3 + c] + (M^T q)_c                  (position attribute + clip-space pull-back, see FOLD above);
//         round 5: M^T is applied to the per-TRIANGLE coefficients instead (the record's S, P0, P1:
//         (M^T q) = (g0 b0 + g1 b1) S + g0 P0 + g1 P1), 11 multiply-adds per pixel instead of 24
//   a[k][c] += b_k y_c                            (9 sums per triangle)
// ~50 vector instructions per pixel row fewer than ShadeLaneFn<L, SIGNS, false, 2, true, true>.
#ifndef MR_SHADE_FOLD_DIFF
#define MR_SHADE_FOLD_DIFF 1
#endif
#ifndef MR_FOLD_LANE_ROWS
#define MR_FOLD_LANE_ROWS 8
#endif
#ifndef MR_FOLD_LANE_WAVES
#define MR_FOLD_LANE_WAVES MR_LANE_WAVES
#endif
#ifndef MR_SHADE_USE_PREPARED
#define MR_SHADE_USE_PREPARED 1   // 0: ignore mr_render_forward's prepared block (A/B)
#endif
template <int L, bool SIGNS>
struct ShadeFoldLaneFn : ShadeGradFn<L, SIGNS, false> {
  using Base = ShadeGradFn<L, SIGNS, false>;
  static constexpr bool kSkipsStrips = true;
  __device__ __forceinline__ bool skip_strip(int img, int rx, int y_begin, int y_end) const {
    return Base::strip_is_empty(img, rx, y_begin, y_end);
  }
  static constexpr int kN = 9;
  static constexpr int kStride = kFoldAccStride;  // COMPACT rows: [corner][c] + 3 of padding (k_shade_gather_fold reads them)
  static constexpr int kLaneRowsPerWave = MR_FOLD_LANE_ROWS;   // 8: 0.2323 -> 0.2281 ms against 16 (32: 0.2507), same box
  static constexpr int kMinWavesPerSimd = MR_FOLD_LANE_WAVES;
  const FoldRec *__restrict__ fold_recs;   // in the PULLED form (store_fold_record with the image's transform rows)
  using Triangle = FoldTriangleW;
  // SIGNS: the sums are formed from the bare sign codes; k_shade_gather_fold multiplies by the loss's scale
  static constexpr bool kUnscaled = SIGNS;
  __device__ static int column(int o) { return o; }  // sum o = corner * 3 + c
  __device__ __forceinline__ void load_triangle(int img, int tri, Triangle &t) const {
    load_fold_triangle_w(fold_recs + (size_t)img * this->T_ + tri, t);
  }
  __device__ __forceinline__ void accumulate(const typename Base::Pixel &p, const Triangle &t, float (&a)[kN],
                                             typename Base::Image &im) const {
    float at[9], dat[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) at[k] = fmaf(p.b.x, t.e0[k], fmaf(p.b.y, t.e1[k], t.c2[k]));
    Base::template attribute_gradients<kUnscaled>(at, p.g, im, dat);
    float g0 = 0.f, g1 = 0.f;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      g0 = fmaf(dat[k], t.e0[k], g0);
      g1 = fmaf(dat[k], t.e1[k], g1);
    }
    // the rasterizer's backward, pulled back to world space per TRIANGLE (corner_rec.h: store_fold_record):
    //   y = d L / d position attribute + (g0 b0 + g1 b1) S + g0 P0 + g1 P1
    const float h = fmaf(g0, p.b.x, g1 * p.b.y);
    float y[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) y[c] = fmaf(h, t.S[c], fmaf(g0, t.P0[c], fmaf(g1, t.P1[c], dat[3 + c])));
    const float b[3] = {p.b.x, p.b.y, p.b.z};
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
      for (int c = 0; c < 3; ++c) a[k * 3 + c] = fmaf(b[k], y[c], a[k * 3 + c]);
  }
};

// The same difference-basis pixel pass for the other gradients a caller may want next to the vertices' (round 4):
// GROUPS selects the attribute groups (bit 0 normals, bit 1 positions, bit 2 diffuse colours) exactly as in
// ShadeLaneFn, FOLD the clip-space pull-back (dclip == nullptr with transforms; needs the position group).  Rows of
// acc keep ShadeGradFn's 36-float layout (the generic gather reads them): 9 sums per selected group, + 9 clip
// sums unless folded.  Needs the G-buffer normalised (alpha = 1) and no light gradients.
// LG (round 5): the light / ambient gradients ride along as per-lane image sums (6 L + 3 of them, one or two lights),
// exactly as in ShadeLaneFn<..., LG = true>: the strips are then MR_LANE_ROWS_LG rows tall (k_sum_strip_rows' geometry).
template <int L, bool SIGNS, int GROUPS, bool FOLD, bool LG = false>
struct ShadeDiffLaneFn : ShadeGradFn<L, SIGNS, LG> {
  using Base = ShadeGradFn<L, SIGNS, LG>;
  static constexpr bool kSkipsStrips = true;
  __device__ __forceinline__ bool skip_strip(int img, int rx, int y_begin, int y_end) const {
    return Base::strip_is_empty(img, rx, y_begin, y_end);
  }
  static_assert(GROUPS >= 1 && GROUPS < 8 && (!FOLD || (GROUPS & 2)), "attribute groups: normals | positions | diffuse");
  static constexpr int kGroups = (GROUPS & 1) + ((GROUPS >> 1) & 1) + ((GROUPS >> 2) & 1);
  static constexpr int kN = 9 * kGroups + (FOLD ? 0 : 9);
  static constexpr int kStride = 36;
  static constexpr int kLaneRowsPerWave = LG ? MR_LANE_ROWS_LG : (kN <= 18 ? MR_FOLD_LANE_ROWS : MR_LANE_ROWS);
  static constexpr int kMinWavesPerSimd = (LG || kN > 27) ? 3 : MR_LANE_WAVES;
  const FoldRec *__restrict__ fold_recs;   // FOLD: in the pulled form (store_fold_record with the image's transform rows)
  using Triangle = std::conditional_t<FOLD, FoldTriangleW, FoldTriangle>;
  __host__ __device__ static constexpr int group(int gi) {
    int g = 0;
    for (int seen = 0; g < 3; ++g) {
      if ((GROUPS >> g) & 1) {
        if (seen == gi) break;
        ++seen;
      }
    }
    return g;
  }
  __device__ static int column(int o) {
    if (o >= 9 * kGroups) return 27 + (o - 9 * kGroups);
    const int gi = o / 9, k = (o % 9) / 3, c = o % 3;
    const int g = gi == 0 ? group(0) : gi == 1 ? group(1) : group(2);
    return k * 9 + g * 3 + c;
  }
  __device__ __forceinline__ void load_triangle(int img, int tri, Triangle &t) const {
    if constexpr (FOLD) load_fold_triangle_w(fold_recs + (size_t)img * this->T_ + tri, t);
    else load_fold_triangle(fold_recs + (size_t)img * this->T_ + tri, t);
  }
  __device__ __forceinline__ void accumulate(const typename Base::Pixel &p, const Triangle &t, float (&a)[kN],
                                             typename Base::Image &im) const {
    float at[9], dat[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) at[k] = fmaf(p.b.x, t.e0[k], fmaf(p.b.y, t.e1[k], t.c2[k]));
    Base::attribute_gradients(at, p.g, im, dat);
    float g0 = 0.f, g1 = 0.f;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      g0 = fmaf(dat[k], t.e0[k], g0);
      g1 = fmaf(dat[k], t.e1[k], g1);
    }
    [[maybe_unused]] float q[3];
    if constexpr (FOLD) {   // the pulled record (round 5, as ShadeFoldLaneFn): (M^T q) = (g0 b0 + g1 b1) S + g0 P0 + g1 P1
      const float h = fmaf(g0, p.b.x, g1 * p.b.y);
#pragma unroll
      for (int c = 0; c < 3; ++c) dat[3 + c] = fmaf(h, t.S[c], fmaf(g0, t.P0[c], fmaf(g1, t.P1[c], dat[3 + c])));
    } else {
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float w0 = t.s[c] * p.b.x - t.u0[c];
        const float w1 = t.s[c] * p.b.y - t.u1[c];
        q[c] = (g0 * w0 + g1 * w1) * t.inv;
      }
    }
    const float b[3] = {p.b.x, p.b.y, p.b.z};
#pragma unroll
    for (int gi = 0; gi < kGroups; ++gi)
#pragma unroll
      for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int c = 0; c < 3; ++c) a[(gi * 3 + k) * 3 + c] = fmaf(b[k], dat[group(gi) * 3 + c], a[(gi * 3 + k) * 3 + c]);
    if (!FOLD) {
#pragma unroll
      for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int c = 0; c < 3; ++c) a[9 * kGroups + k * 3 + c] = fmaf(b[k], q[c], a[9 * kGroups + k * 3 + c]);
    }
  }
};

__global__ __launch_bounds__(kThreads) void k_shade_scatter(
    const float *__restrict__ acc, const int32_t *__restrict__ tris, int B, int V, int T,
    float *__restrict__ dnormals, float *__restrict__ dpositions, float *__restrict__ ddiffuse,
    float *__restrict__ dclip) {
  const long gid = (long)blockIdx.x * kThreads + threadIdx.x;
  if (gid >= (long)B * T) return;
  const float4 *row = (const float4 *)(acc + gid * 36);  // 144-byte rows, 16-byte aligned
  float a[36];
  bool any = false;
#pragma unroll
  for (int q = 0; q < 9; ++q) {
    const float4 v = row[q];
    a[4 * q] = v.x; a[4 * q + 1] = v.y; a[4 * q + 2] = v.z; a[4 * q + 3] = v.w;
    any |= (v.x != 0.f) | (v.y != 0.f) | (v.z != 0.f) | (v.w != 0.f);
  }
  if (!any) return;
  const int b = (int)(gid / T);
  const int t = (int)(gid - (long)b * T);
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const int vi = tris[3 * t + k];
    if ((unsigned)vi >= (unsigned)V) continue;
    const size_t v3 = ((size_t)b * V + vi) * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      atomicAdd(&dnormals[v3 + c], a[k * 9 + c]);
      atomicAdd(&dpositions[v3 + c], a[k * 9 + 3 + c]);
      atomicAdd(&ddiffuse[v3 + c], a[k * 9 + 6 + c]);
    }
    float *dc = dclip + ((size_t)b * V + vi) * 4;
    atomicAdd(&dc[0], a[27 + k * 3 + 0]);
    atomicAdd(&dc[1], a[27 + k * 3 + 1]);
    atomicAdd(&dc[3], a[27 + k * 3 + 2]);
  }
}

// Vertex-centric alternative to k_shade_scatter: the rows of the triangles incident to a vertex
// are summed per (image, vertex) (CSR adjacency built once per triangle array on the host side:
// entry e = 3 * triangle + corner, grouped by vertex).  No atomics, every output is written
// exactly once, the summation order is fixed.  Sixteen lanes per (image, vertex), one per output
// float (9 attribute gradients, 3 clip gradients, the clip z column's 0, 3 idle); the incident
// corners are fetched kChunk at a time, all loads of a chunk in flight together: the kernel is a
// chain of dependent load round trips (offsets -> entries -> rows), and as long as its longest
// chain -- the poles of a UV sphere have 50-100 incident triangles.
#ifndef MR_GATHER_CHUNK
#define MR_GATHER_CHUNK 8   // measured at 32 x 2502 vertices: 8 -> 31.6, 16 -> 33.1, 32 -> 41.8, 64 -> 74.9 us
#endif
template <bool DET>
__global__ __launch_bounds__(kThreads) void k_shade_gather(
    const float *__restrict__ acc, const float *__restrict__ det_scale, const int32_t *__restrict__ offsets,
    const int32_t *__restrict__ entries, int B, int V, int T, float *__restrict__ dnormals,
    float *__restrict__ dpositions, float *__restrict__ ddiffuse, float *__restrict__ dclip,
    const float *__restrict__ transforms) {
  const long tid = (long)blockIdx.x * kThreads + threadIdx.x;
  const long gid = tid >> 4;   // (image, vertex)
  const int j = (int)(tid & 15);
  if (gid >= (long)B * V || j > 12) return;
  const int b = (int)(gid / V);
  const int v = (int)(gid - (long)b * V);
  // outputs the caller does not want (nullptr): their sums were not formed either
  if ((j < 3 && !dnormals) || (j >= 6 && j < 9 && !ddiffuse)) return;
  float sum = 0.f;
  if (j < 12 && (dclip || j < 9)) {   // dclip == nullptr: not wanted (and its columns were not accumulated)
    const int e1 = offsets[v + 1];
    constexpr int kChunk = MR_GATHER_CHUNK;
    // this lane's float inside a triangle's 36-float row is  col0 + k * colk  for corner k
    const unsigned col0 = j < 9 ? (unsigned)j : 27u + (unsigned)(j - 9), colk = j < 9 ? 9u : 3u;
    const float *acc_f = acc + (size_t)b * T * 36;                                  // 32-bit offsets from here:
    const long long *acc_x = (const long long *)acc + (size_t)b * T * 36;          // T * 36 < 2^31
    for (int i = offsets[v]; i < e1; i += kChunk) {
      int e[kChunk];
      float val[kChunk];
#pragma unroll
      for (int u = 0; u < kChunk; ++u) e[u] = i + u < e1 ? entries[i + u] : -1;
#pragma unroll
      for (int u = 0; u < kChunk; ++u) {
        const unsigned t = (unsigned)e[u] / 3u, k = (unsigned)e[u] - 3u * t;
        const unsigned at = t * 36u + col0 + k * colk;
        // DET: fixed-point rows (8-byte elements) back to float, then the same fixed-order sums
        val[u] = e[u] < 0 ? 0.f : DET ? (float)acc_x[at] * det_scale[1] : acc_f[at];
      }
#pragma unroll
      for (int u = 0; u < kChunk; ++u) sum += val[u];
    }
  }
  if (transforms) {
    // clip = transforms[b] . (position, 1): the clip-space gradient (lanes 9, 10, 11 of this vertex:
    // x, y, w; the z column is 0) is pulled back onto the position gradient of lanes 3..5
    const int base = (int)(threadIdx.x & (kWave - 1)) & ~15;
    const float dx = __shfl(sum, base + 9), dy = __shfl(sum, base + 10), dw = __shfl(sum, base + 11);
    if (j >= 3 && j < 6) {
      const float *m = transforms + (size_t)b * 16 + (j - 3);
      sum += (m[0] * dx + m[4] * dy) + m[12] * dw;
    }
  }
  if (DET && *det_overflow_flag(det_scale)) sum = __int_as_float(0x7fc00000);  // see atomic_add_fixed
  if (!dclip && j >= 9) return;
  float *out = j < 3 ? dnormals + gid * 3 + j
             : j < 6 ? dpositions + gid * 3 + (j - 3)
             : j < 9 ? ddiffuse + gid * 3 + (j - 6)
             : j < 11 ? dclip + gid * 4 + (j - 9)
             : j == 11 ? dclip + gid * 4 + 3 : dclip + gid * 4 + 2;  // j == 12: column z stays 0
  *out = sum;
}

// The folded variant's per-vertex gather: its accumulator rows hold nine floats -- [corner][c], the whole
// gradient of the corner's world-space position -- in 48 bytes, so a vertex needs three sums, not thirteen:
// four lanes per (image, vertex) (the fourth idles), 320 k threads instead of 1.3 M at 32 x 2502 vertices, a third
// of the bytes.  Same fixed summation order as k_shade_gather (the adjacency's).
// CLEAR (not used, see the launch): every accumulator float is read by exactly one lane of one vertex -- entry
// e = 3 t + k belongs to vertex triangles[t][k] alone -- which stores a zero behind its read: the rows are clear
// again when the kernel ends.
// scale_src / scale_mul: the sums were formed from bare sign codes (ShadeFoldLaneFn<L, true>): every output is
// multiplied by scale_src[0] * scale_mul (the loss's upstream gradient over its element count); nullptr: by 1.
template <bool CLEAR>
__global__ __launch_bounds__(kThreads) void k_shade_gather_fold(
    float *__restrict__ acc, const int32_t *__restrict__ offsets, const int32_t *__restrict__ entries, int B, int V,
    int T, float *__restrict__ dpositions, const float *__restrict__ scale_src, float scale_mul) {
  const long tid = (long)blockIdx.x * kThreads + threadIdx.x;
  const long gid = tid >> 2;   // (image, vertex)
  const int c = (int)(tid & 3);
  if (gid >= (long)B * V || c == 3) return;
  const int b = (int)(gid / V);
  const int v = (int)(gid - (long)b * V);
  float *acc_f = acc + (size_t)b * T * kFoldAccStride;
  constexpr int kChunk = MR_GATHER_CHUNK;
  float sum = 0.f;
  const int e1 = offsets[v + 1];
  for (int i = offsets[v]; i < e1; i += kChunk) {
    int e[kChunk];
    float val[kChunk];
#pragma unroll
    for (int u = 0; u < kChunk; ++u) e[u] = i + u < e1 ? entries[i + u] : -1;
#pragma unroll
    for (int u = 0; u < kChunk; ++u) {
      const unsigned t = (unsigned)e[u] / 3u, k = (unsigned)e[u] - 3u * t;
      val[u] = e[u] < 0 ? 0.f : acc_f[t * (unsigned)kFoldAccStride + k * 3u + (unsigned)c];
    }
    if (CLEAR) {
#pragma unroll
      for (int u = 0; u < kChunk; ++u) {
        const unsigned t = (unsigned)e[u] / 3u, k = (unsigned)e[u] - 3u * t;
        if (e[u] >= 0) acc_f[t * (unsigned)kFoldAccStride + k * 3u + (unsigned)c] = 0.f;
      }
    }
#pragma unroll
    for (int u = 0; u < kChunk; ++u) sum += val[u];
  }
  dpositions[gid * 3 + c] = scale_src ? sum * (scale_src[0] * scale_mul) : sum;
}

inline unsigned capped_blocks(size_t n) {
  const size_t want = (n + kThreads - 1) / kThreads;
  const size_t cap = 256u * 32u;
  return (unsigned)(want < cap ? (want ? want : 1) : cap);
}

// 8 bytes per element: room for the deterministic mode's fixed-point accumulators
inline size_t shade_acc_bytes(int B, int T) { return align_up((size_t)B * T * 36 * sizeof(long long), 256); }
constexpr size_t kDetMiscBytes = 512;  // det_scale (2 floats), max bits (1 int)
// one row of light sums per strip of the pixel pass (the variant with light gradients walks 16-row strips)
struct LightGradLaneGeometry { static constexpr int kLaneRowsPerWave = MR_LANE_ROWS_LG; };  // = ShadeLaneFn<..., LG = true, ...>'s strips
inline int light_strips_per_image(int B, int W, int H, bool lanes) {
  return lanes ? lanes_strips_per_image<LightGradLaneGeometry>(B, W, H) : strips_per_image<ShadeGradFn<1, true, true>>(W, H);
}
inline size_t light_rows_bytes(int B, int W, int H) {
  const int strips = max(light_strips_per_image(B, W, H, true), light_strips_per_image(B, W, H, false));
  return align_up((size_t)B * strips * (kMaxLights * 6 + 3) * sizeof(float), 256);
}

// ---- deterministic mode helpers -------------------------------------------------------------
// largest |x| of an array as float bits (non-negative floats order like integers; a NaN sorts on top)
__global__ __launch_bounds__(kThreads) void k_abs_max(const float4 *__restrict__ x, size_t n4, int *__restrict__ max_bits) {
  int best = 0;
  for (size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x; i < n4; i += (size_t)gridDim.x * kThreads) {
    const float4 v = x[i];
    best = max(max(best, __float_as_int(fabsf(v.x))), max(__float_as_int(fabsf(v.y)), __float_as_int(fabsf(v.z))));
  }
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

// (2^k, 2^-k) with k such that the largest upstream gradient maps to about 2^41
__global__ void k_det_scale(const int *__restrict__ max_bits, const float *__restrict__ sign_upstream,
                            float sign_inv_n, float *__restrict__ det_scale) {
  const float g = sign_upstream ? fabsf(sign_upstream[0] * sign_inv_n) : __int_as_float(max_bits[0]);
  int e = 0;
  if (g > 0.0f && g < INFINITY) (void)frexpf(g, &e);  // g = m * 2^e, m in [0.5, 1)
  const int k = min(max(41 - e, -100), 100);
  det_scale[0] = ldexpf(1.0f, k);
  det_scale[1] = ldexpf(1.0f, -k);
}

inline size_t corner_bytes(int B, int T) { return align_up((size_t)B * T * sizeof(CornerRec), 256); }

}  // namespace

int shade_max_lights() { return kMaxLightsAny; }
int shade_light_gradient_max_lights() { return kMaxLights; }

size_t shade_forward_ws(int B, int V, int T, int W, int H) {
  (void)V; (void)W; (void)H;
  return corner_bytes(B, T);
}

int launch_corner_setup(const float *normals, const float *positions, const float *diffuse,
                               const int32_t *tris, int B, int V, int T, CornerRec *out, hipStream_t s) {
  const long nbt = (long)B * T;
  hipLaunchKernelGGL(k_corner_setup, dim3((unsigned)((nbt + kThreads - 1) / kThreads)), dim3(kThreads), 0,
                     s, (const F3 *)normals, (const F3 *)positions, (const F3 *)diffuse, tris, B, V, T, out);
  return check_launch();
}

int launch_shade_forward(const int32_t *ids, const float *bary, const float *normals,
                         const float *positions, const float *diffuse, const int32_t *tris,
                         const float *light_pos, const float *light_col, const float *ambient,
                         int B, int V, int T, int W, int H, int L, float *rgba, void *ws,
                         hipStream_t s) {
  const size_t n_px = (size_t)B * W * H;
  if (n_px == 0) return MR_OK;
  CornerRec *corners = (CornerRec *)ws;
  int rc = launch_corner_setup(normals, positions, diffuse, tris, B, V, T, corners, s);
  if (rc != MR_OK) return rc;
  Lights lights{light_pos, light_col, ambient, L};
  const int x_blocks = (W + kThreads - 1) / kThreads, y_blocks = (H + kShadeRows - 1) / kShadeRows;
  {
    KernelTimer timer(MR_TIMER_SHADE_FORWARD, s);
    hipLaunchKernelGGL(k_shade_forward, dim3((unsigned)(x_blocks * y_blocks * B)), dim3(kThreads), 0, s,
                       ids, (const F3 *)bary, corners, lights, B, T, W, H, x_blocks, y_blocks,
                       (float4 *)rgba);
  }
  return check_launch();
}

size_t shade_backward_prepared_bytes(int B, int T) { return fold_prepared_bytes(B, T); }

size_t shade_backward_ws(int B, int V, int T, int W, int H) {
  return shade_acc_bytes(B, T) + align_up((size_t)B * T * sizeof(BwdRec), 256) + corner_bytes(B, T) +
         kDetMiscBytes + light_rows_bytes(B, W, H) + align_up((size_t)B * T * sizeof(FoldRec), 256) +
         align_up((size_t)B * V * 4 * sizeof(float), 256);  // last two: the folded kernel's records, dclip scratch
}

thread_local int g_deterministic = 0;  // mr_set_deterministic
// mr_debug_set_shade_backward_kernel: 0 = automatic (lane-accumulating kernel where it exists: no
// light gradients, not every attribute gradient wanted, not deterministic), 1 = always the rows
// kernel, 2 = the lane-accumulating kernel wherever it is instantiated
thread_local int g_shade_backward_kernel = 0;

int launch_shade_backward(const float *drgba, const uint8_t *signs, const float *sign_upstream,
                          const int32_t *ids, const float *bary,
                          const float *clip, const float *normals, const float *positions,
                          const float *diffuse, const int32_t *tris, const float *light_pos,
                          const float *light_col, const float *ambient, int B, int V, int T, int W,
                          int H, int L, float *dclip, float *dnormals, float *dpositions,
                          float *ddiffuse, float *light_grads, const void *corner_records,
                          const int32_t *vertex_offsets, const int32_t *vertex_entries,
                          const float *transforms, int gbuffer_flags, void *prepared, const uint8_t *empty_regions, void *ws,
                          hipStream_t s) {
  if (B == 0) return MR_OK;
  if (transforms && !(vertex_offsets && vertex_entries)) return MR_EINVAL;  // the gather applies them
  if (!dclip && !transforms) return MR_EINVAL;  // without the pull-back the clip-space gradient IS the vertex gradient
  const size_t v3 = (size_t)B * V * 3 * sizeof(float), v4 = (size_t)B * V * 4 * sizeof(float);
  const size_t lg = light_grads ? (size_t)B * (L * 6 + 3) * sizeof(float) : 0;  // nullptr: not wanted
  const bool det = g_deterministic != 0;
  // dclip == nullptr (with transforms): the caller wants the gradient of the world-space positions only.  Where
  // the lane kernel has the variant, the pull-back through the transforms is folded into the pixel pass
  // (ShadeLaneFn<..., FOLD>) and no clip-space sums exist at all; elsewhere the clip gradient goes to scratch.
  const int groups_wanted = (dnormals ? 1 : 0) | 2 | (ddiffuse ? 4 : 0);   // (6 -- positions + diffuse -- has no lane kernel)
  const bool lg_lanes = light_grads && MR_SHADE_LANES_LG && L >= 1 && L <= 2 && corner_records != nullptr;   // ShadeDiffLaneFn<..., LG>
  const bool fold_any = !dclip && transforms && MR_SHADE_LANES_FOLD && (!light_grads || lg_lanes) && !det && groups_wanted != 6 &&
                        (gbuffer_flags & MR_GBUFFER_NORMALISED) != 0 && g_shade_backward_kernel != 1 && T > 0 && V > 0 &&
                        (corner_records != nullptr || prepared != nullptr) && MR_SHADE_FOLD_DIFF;   // (any attribute groups)
  const bool fold = fold_any || (!dclip && transforms && MR_SHADE_LANES_FOLD && !light_grads && !det && !dnormals && !ddiffuse &&
                    (gbuffer_flags & MR_GBUFFER_NORMALISED) != 0 && g_shade_backward_kernel != 1 && T > 0 && V > 0);
  if (!dclip && !fold)
    dclip = (float *)((char *)ws + shade_backward_ws(B, V, T, W, H) - align_up((size_t)B * V * 4 * sizeof(float), 256));
  const float sign_inv_n = 1.0f / (float)((size_t)B * H * W * 4);  // the L1 mean runs over the whole image
  // With the vertex adjacency the gather writes every vertex output exactly once, and k_bwd_setup
  // clears the accumulator rows and light_grads on the side: no memset launches at all (two of
  // ~6 us each before).  (Not in the deterministic mode: its fixed-point side buffers are cleared
  // the plain way.)
  const bool fused_clear = vertex_offsets && vertex_entries && !det && T > 0 && V > 0;
  // Otherwise the outputs are zeroed here.  A caller that lays them out back to back (dclip,
  // dnormals, dpositions, ddiffuse, light_grads -- _native.py does) gets ONE memset instead of five
  // launch-bound ones.
  if (fused_clear) {   // (always the case with `fold`: transforms imply the adjacency, and it excludes det)
  } else if (dnormals && ddiffuse && (char *)dnormals == (char *)dclip + v4 && (char *)dpositions == (char *)dnormals + v3 &&
             (char *)ddiffuse == (char *)dpositions + v3 && (char *)light_grads == (char *)ddiffuse + v3) {
    if (zero_async(dclip, v4 + 3 * v3 + lg, s) != hipSuccess) return check_launch();
  } else {
    if (V > 0) {
      if (zero_async(dclip, v4, s) != hipSuccess) return check_launch();
      if (dnormals && zero_async(dnormals, v3, s) != hipSuccess) return check_launch();
      if (zero_async(dpositions, v3, s) != hipSuccess) return check_launch();
      if (ddiffuse && zero_async(ddiffuse, v3, s) != hipSuccess) return check_launch();
    }
    if (light_grads && zero_async(light_grads, lg, s) != hipSuccess) return check_launch();
  }
  if (T == 0 || V == 0) return MR_OK;
  float *acc = (float *)ws;
  BwdRec *recs = (BwdRec *)((char *)ws + shade_acc_bytes(B, T));
  CornerRec *corners = (CornerRec *)((char *)recs + align_up((size_t)B * T * sizeof(BwdRec), 256));
  float *det_scale = (float *)((char *)corners + corner_bytes(B, T));
  int *max_bits = (int *)(det_scale + 4);
  float *light_rows = (float *)((char *)det_scale + kDetMiscBytes);
  FoldRec *fold_recs = (FoldRec *)((char *)light_rows + light_rows_bytes(B, W, H));
  const bool fold_diff = fold && MR_SHADE_FOLD_DIFF && (corner_records != nullptr || prepared != nullptr) && !dnormals && !ddiffuse &&
                         !light_grads;   // (with light gradients: ShadeDiffLaneFn<..., LG>, below)
  // the difference-basis pixel pass for every other lane-kernel case on a normalised G-buffer: normals / diffuse
  // wanted, and / or the clip-space gradient wanted on its own, and (round 5, one or two lights) the light gradients
  const bool diff_general = !fold_diff && MR_SHADE_FOLD_DIFF && corner_records != nullptr &&
                            (!light_grads || (MR_SHADE_LANES_LG && L >= 1 && L <= 2)) && !det &&
                            (gbuffer_flags & MR_GBUFFER_NORMALISED) != 0 && g_shade_backward_kernel != 1 &&
                            vertex_offsets && vertex_entries && T > 0 && V > 0 && groups_wanted != 6;
  // `prepared` (mr_render_forward's backward_prepared: FoldRec[B*T] + cleared compact accumulator rows): the folded
  // kernel's setup launch is skipped, and the gather leaves the rows cleared again for the next backward call
  const bool use_prepared = fold_diff && prepared != nullptr && MR_SHADE_USE_PREPARED;
  if (fold && !fold_diff && !diff_general && !dclip) return MR_EINVAL;   // (cannot happen: every folding path is one of the two)
  if (use_prepared) {
    fold_recs = (FoldRec *)prepared;
    acc = (float *)((char *)prepared + fold_prepared_recs_bytes(B, T));
  }
  if (det && !(vertex_offsets && vertex_entries)) return MR_EINVAL;  // the scatter path is atomics only
  const size_t acc_bytes = (size_t)B * T * 36 * (det ? sizeof(long long) : sizeof(float));
  if (!fused_clear && zero_async(acc, acc_bytes, s) != hipSuccess) return check_launch();
  int rc = MR_OK;
  if (det) {
    if (zero_async(det_scale, kDetMiscBytes, s) != hipSuccess) return check_launch();
    if (!signs) {
      const size_t n4 = (size_t)B * H * W;
      const unsigned blocks = capped_blocks(n4) < 2048u ? capped_blocks(n4) : 2048u;
      hipLaunchKernelGGL(k_abs_max, dim3(blocks), dim3(kThreads), 0, s, (const float4 *)drgba, n4, max_bits);
      if ((rc = check_launch()) != MR_OK) return rc;
    }
    hipLaunchKernelGGL(k_det_scale, dim3(1), dim3(1), 0, s, max_bits, sign_upstream, sign_inv_n, det_scale);
    if ((rc = check_launch()) != MR_OK) return rc;
  }
  if (use_prepared) rc = MR_OK;
  else
  rc = fused_clear ? launch_bwd_setup(clip, tris, B, V, T, recs, s, acc, (fold_diff ? kFoldAccStride : 36) * sizeof(float), light_grads,
                                      light_grads ? B * (L * 6 + 3) : 0, (fold_diff || diff_general) ? corner_records : nullptr,
                                      (fold_diff || diff_general) ? fold_recs : nullptr,
                                      fold ? transforms : nullptr)   // the folded kernels read the pulled form
                   : launch_bwd_setup(clip, tris, B, V, T, recs, s);
  if (rc != MR_OK) return rc;
  if (corner_records) {  // the forward's records (same inputs): skip the gather
    corners = (CornerRec *)corner_records;
  } else {
    rc = launch_corner_setup(normals, positions, diffuse, tris, B, V, T, corners, s);
    if (rc != MR_OK) return rc;
  }
  Lights lights{light_pos, light_col, ambient, L};
  if (L > kMaxLights && light_grads) return MR_EINVAL;  // light gradients: four lights per call (see ShadeGradFn)
  // Attribute gradients the caller wants (nullptr: not wanted).  The scatter path (no adjacency)
  // writes all of them.
  if ((!dnormals || !ddiffuse) && !(vertex_offsets && vertex_entries)) return MR_EINVAL;
  const int groups = (dnormals ? 1 : 0) | 2 | (ddiffuse ? 4 : 0);
  // with light gradients: one or two lights (6 L + 3 more per-lane sums; three and four stay on the rows kernel)
  const bool lanes_exist = (!light_grads || (MR_SHADE_LANES_LG && L <= 2)) && !det && groups != 6 &&
                           (groups != 7 || 1);
  const bool use_lanes = lanes_exist && g_shade_backward_kernel != 1;
  const bool opaque = (gbuffer_flags & MR_GBUFFER_NORMALISED) != 0;  // (the lane kernels only)
#define MR_SHADE_LANES_O(NL, G, OPQ)                                                            \
  {                                                                                             \
    KernelTimer timer(MR_TIMER_SHADE_BACKWARD, s);                                              \
    if (signs) {                                                                                \
      ShadeLaneFn<NL, true, false, G, OPQ> fn{{nullptr, signs, sign_upstream, sign_inv_n, ids, (const F3 *)bary, \
                                               corners, recs, lights, nullptr, T, W, H}};       \
      rc = launch_accumulate_lanes(fn, B, T, W, H, acc, s);                                     \
    } else {                                                                                    \
      ShadeLaneFn<NL, false, false, G, OPQ> fn{{(const float4 *)drgba, nullptr, nullptr, 0.0f, ids, (const F3 *)bary, \
                                                corners, recs, lights, nullptr, T, W, H}};      \
      rc = launch_accumulate_lanes(fn, B, T, W, H, acc, s);                                     \
    }                                                                                           \
  }
#define MR_SHADE_LANES_LGV(NL, G)                                                               \
  {                                                                                             \
    KernelTimer timer(MR_TIMER_SHADE_BACKWARD, s);                                              \
    if (signs) {                                                                                \
      ShadeLaneFn<NL, true, true, G, false> fn{{nullptr, signs, sign_upstream, sign_inv_n, ids, (const F3 *)bary, \
                                                corners, recs, lights, light_rows, T, W, H}};   \
      rc = launch_accumulate_lanes(fn, B, T, W, H, acc, s);                                     \
    } else {                                                                                    \
      ShadeLaneFn<NL, false, true, G, false> fn{{(const float4 *)drgba, nullptr, nullptr, 0.0f, ids, (const F3 *)bary, \
                                                 corners, recs, lights, light_rows, T, W, H}};  \
      rc = launch_accumulate_lanes(fn, B, T, W, H, acc, s);                                     \
    }                                                                                           \
  }
#define MR_SHADE_LANES_FOLDED(NL)                                                               \
  if (fold_diff) {                                                                              \
    KernelTimer timer(MR_TIMER_SHADE_BACKWARD, s);                                              \
    if (signs) {                                                                                \
      ShadeFoldLaneFn<NL, true> fn{{nullptr, signs, sign_upstream, sign_inv_n, ids, (const F3 *)bary, corners, recs, \
                                    lights, nullptr, T, W, H, transforms, empty_regions}, fold_recs};          \
      rc = launch_accumulate_lanes(fn, B, T, W, H, acc, s);                                     \
    } else {                                                                                    \
      ShadeFoldLaneFn<NL, false> fn{{(const float4 *)drgba, nullptr, nullptr, 0.0f, ids, (const F3 *)bary, corners,  \
                                     recs, lights, nullptr, T, W, H, transforms, empty_regions}, fold_recs};   \
      rc = launch_accumulate_lanes(fn, B, T, W, H, acc, s);                                     \
    }                                                                                           \
  } else {                                                                                      \
    KernelTimer timer(MR_TIMER_SHADE_BACKWARD, s);                                              \
    if (signs) {                                                                                \
      ShadeLaneFn<NL, true, false, 2, true, true> fn{{nullptr, signs, sign_upstream, sign_inv_n, ids, (const F3 *)bary, \
                                                      corners, recs, lights, nullptr, T, W, H, transforms}};            \
      rc = launch_accumulate_lanes(fn, B, T, W, H, acc, s);                                     \
    } else {                                                                                    \
      ShadeLaneFn<NL, false, false, 2, true, true> fn{{(const float4 *)drgba, nullptr, nullptr, 0.0f, ids, (const F3 *)bary, \
                                                       corners, recs, lights, nullptr, T, W, H, transforms}};                \
      rc = launch_accumulate_lanes(fn, B, T, W, H, acc, s);                                     \
    }                                                                                           \
  }
#define MR_SHADE_LANES(NL, G)                                                                   \
  if (fold) MR_SHADE_LANES_FOLDED(NL) else if (opaque) MR_SHADE_LANES_O(NL, G, true) else MR_SHADE_LANES_O(NL, G, false)
#define MR_SHADE_LANES_LIGHTS(NL, G)   /* one or two lights: the variant with light gradients exists */ \
  if (light_grads) MR_SHADE_LANES_LGV(NL, G) else MR_SHADE_LANES(NL, G)
#define MR_SHADE_LANES_G(NL)                                                                    \
  if (groups == 2) MR_SHADE_LANES(NL, 2) else if (groups == 3) MR_SHADE_LANES(NL, 3) else MR_SHADE_LANES(NL, 7)
#define MR_SHADE_LANES_GL(NL)                                                                   \
  if (groups == 2) MR_SHADE_LANES_LIGHTS(NL, 2) else if (groups == 3) MR_SHADE_LANES_LIGHTS(NL, 3) else MR_SHADE_LANES_LIGHTS(NL, 7)
  if (diff_general && use_lanes) {
    const bool folded = dclip == nullptr;   // (implies transforms)
#define MR_SHADE_DIFF_LG(NL, G, F, LGV)                                                         \
  {                                                                                             \
    KernelTimer timer(MR_TIMER_SHADE_BACKWARD, s);                                              \
    if (signs) {                                                                                \
      ShadeDiffLaneFn<NL, true, G, F, LGV> fn{{nullptr, signs, sign_upstream, sign_inv_n, ids, (const F3 *)bary, corners, recs, \
                                               lights, LGV ? light_rows : nullptr, T, W, H, transforms, empty_regions}, fold_recs};    \
      rc = launch_accumulate_lanes(fn, B, T, W, H, acc, s);                                     \
    } else {                                                                                    \
      ShadeDiffLaneFn<NL, false, G, F, LGV> fn{{(const float4 *)drgba, nullptr, nullptr, 0.0f, ids, (const F3 *)bary, corners,  \
                                                recs, lights, LGV ? light_rows : nullptr, T, W, H, transforms, empty_regions}, fold_recs};                         \
      rc = launch_accumulate_lanes(fn, B, T, W, H, acc, s);                                     \
    }                                                                                           \
  }
#define MR_SHADE_DIFF(NL, G, F) MR_SHADE_DIFF_LG(NL, G, F, false)
#define MR_SHADE_DIFF_GV(NL, LGV)                                                               \
  if (folded) {                                                                                 \
    if (groups == 2) MR_SHADE_DIFF_LG(NL, 2, true, LGV) else if (groups == 3) MR_SHADE_DIFF_LG(NL, 3, true, LGV) else MR_SHADE_DIFF_LG(NL, 7, true, LGV) \
  } else {                                                                                      \
    if (groups == 2) MR_SHADE_DIFF_LG(NL, 2, false, LGV) else if (groups == 3) MR_SHADE_DIFF_LG(NL, 3, false, LGV) else MR_SHADE_DIFF_LG(NL, 7, false, LGV) \
  }
#define MR_SHADE_DIFF_G(NL) MR_SHADE_DIFF_GV(NL, false)
#define MR_SHADE_DIFF_GL(NL)   /* one or two lights: the variant with light gradients exists */ \
  if (light_grads) { MR_SHADE_DIFF_GV(NL, true) } else { MR_SHADE_DIFF_GV(NL, false) }
    switch (L) {
      case 1: MR_SHADE_DIFF_GL(1); break;
      case 2: MR_SHADE_DIFF_GL(2); break;
      case 3: MR_SHADE_DIFF_G(3); break;
      case 4: MR_SHADE_DIFF_G(4); break;
      default: MR_SHADE_DIFF_G(0); break;
    }
#undef MR_SHADE_DIFF_GL
#undef MR_SHADE_DIFF_G
#undef MR_SHADE_DIFF_GV
#undef MR_SHADE_DIFF
#undef MR_SHADE_DIFF_LG
  } else if (use_lanes) {
    switch (L) {
      case 1: MR_SHADE_LANES_GL(1); break;
      case 2: MR_SHADE_LANES_GL(2); break;
      case 3: MR_SHADE_LANES_G(3); break;
      case 4: MR_SHADE_LANES_G(4); break;
      default: MR_SHADE_LANES_G(0); break;  // 5..kMaxLightsAny lights: run-time loop
    }
  } else {
#define MR_SHADE_BWD(NL)                                                                        \
  {                                                                                             \
    KernelTimer timer(MR_TIMER_SHADE_BACKWARD, s);                                              \
    if (signs && light_grads) {                                                                 \
      ShadeGradFn<NL, true, true> fn{nullptr, signs, sign_upstream, sign_inv_n, ids, (const F3 *)bary, corners, \
                                     recs, lights, light_rows, T, W, H};                        \
      rc = launch_accumulate_rows(fn, B, T, W, H, acc, s, det ? det_scale : nullptr);           \
    } else if (signs) {                                                                         \
      ShadeGradFn<NL, true, false> fn{nullptr, signs, sign_upstream, sign_inv_n, ids, (const F3 *)bary, corners, \
                                      recs, lights, nullptr, T, W, H};                          \
      rc = launch_accumulate_rows(fn, B, T, W, H, acc, s, det ? det_scale : nullptr);           \
    } else if (light_grads) {                                                                   \
      ShadeGradFn<NL, false, true> fn{(const float4 *)drgba, nullptr, nullptr, 0.0f, ids, (const F3 *)bary, \
                                      corners, recs, lights, light_rows, T, W, H};              \
      rc = launch_accumulate_rows(fn, B, T, W, H, acc, s, det ? det_scale : nullptr);           \
    } else {                                                                                    \
      ShadeGradFn<NL, false, false> fn{(const float4 *)drgba, nullptr, nullptr, 0.0f, ids, (const F3 *)bary, \
                                       corners, recs, lights, nullptr, T, W, H};                \
      rc = launch_accumulate_rows(fn, B, T, W, H, acc, s, det ? det_scale : nullptr);           \
    }                                                                                           \
  }
  switch (L) {
    case 1: MR_SHADE_BWD(1); break;
    case 2: MR_SHADE_BWD(2); break;
    case 3: MR_SHADE_BWD(3); break;
    case 4: MR_SHADE_BWD(4); break;
    default: {  // 5..kMaxLightsAny lights: run-time loop, no light gradients (rejected above)
      KernelTimer timer(MR_TIMER_SHADE_BACKWARD, s);
      if (signs) {
        ShadeGradFn<0, true, false> fn{nullptr, signs, sign_upstream, sign_inv_n, ids, (const F3 *)bary, corners,
                                       recs, lights, nullptr, T, W, H};
        rc = launch_accumulate_rows(fn, B, T, W, H, acc, s, det ? det_scale : nullptr);
      } else {
        ShadeGradFn<0, false, false> fn{(const float4 *)drgba, nullptr, nullptr, 0.0f, ids, (const F3 *)bary,
                                        corners, recs, lights, nullptr, T, W, H};
        rc = launch_accumulate_rows(fn, B, T, W, H, acc, s, det ? det_scale : nullptr);
      }
    } break;
  }
  }
#undef MR_SHADE_BWD
#undef MR_SHADE_LANES_G
#undef MR_SHADE_LANES
#undef MR_SHADE_LANES_FOLDED
  if (rc != MR_OK) return rc;
  if (light_grads) {  // the strips' rows of light sums -> [B][6L + 3], fixed order (every element is written)
    rc = launch_sum_strip_rows(light_rows, B, light_strips_per_image(B, W, H, use_lanes), L * 6 + 3, light_grads, s);
    if (rc != MR_OK) return rc;
  }
  if (fold_diff) {  // compact rows, position gradient only
    const long nbv = (long)B * V * 4;
    // (a `prepared` block's rows are left dirty: it serves ONE backward call.  A gather that zeroes what it reads --
    //  k_shade_gather_fold<true> -- would let a block serve any number of calls, but its scattered 4-byte stores
    //  took the gather from 12.7 to 22.3 us; the host side falls back to this call's own setup kernel instead when
    //  a retained graph is differentiated a second time.)
    hipLaunchKernelGGL(k_shade_gather_fold<false>, dim3((unsigned)((nbv + kThreads - 1) / kThreads)), dim3(kThreads), 0, s,
                       acc, vertex_offsets, vertex_entries, B, V, T, dpositions, signs ? sign_upstream : nullptr, sign_inv_n);
    return check_launch();
  }
  if (vertex_offsets && vertex_entries) {
    const long nbv = (long)B * V * 16;  // sixteen lanes per vertex
    const dim3 grid((unsigned)((nbv + kThreads - 1) / kThreads));
    if (det) {
      hipLaunchKernelGGL(k_shade_gather<true>, grid, dim3(kThreads), 0, s, acc, det_scale, vertex_offsets,
                         vertex_entries, B, V, T, dnormals, dpositions, ddiffuse, dclip, transforms);
    } else {
      hipLaunchKernelGGL(k_shade_gather<false>, grid, dim3(kThreads), 0, s, acc, det_scale, vertex_offsets,
                         vertex_entries, B, V, T, dnormals, dpositions, ddiffuse, dclip, transforms);
    }
    return check_launch();
  }
  const long nbt = (long)B * T;
  hipLaunchKernelGGL(k_shade_scatter, dim3((unsigned)((nbt + kThreads - 1) / kThreads)),
                     dim3(kThreads), 0, s, acc, tris, B, V, T, dnormals, dpositions, ddiffuse, dclip);
  return check_launch();
}

}  // namespace mr
