// Fused deferred shading WITH the specular term, for gfx950 (MI355X).
//
// Same job as shade.hip (attribute interpolation + Phong straight from the G-buffer, forward
// and backward) for render() calls that pass specular_colors / shininess_coefficients
// (reference: src/mesh_renderer/render.py:157-181, 199-228 and phong_shader :326-372).
// The reference's specular term has a global coupling: the per-pixel "reflection . camera
// direction" dot product is L2-normalised ACROSS ALL PIXELS of an image, background included
// (render.py:342-348, dim=2 is the pixel axis), before it is clamped and raised to the
// shininess.  That costs one extra pass over the G-buffer each way:
//
//   forward   k_spec_pixels<L, kNorms>   sum over all pixels of rdc^2 per (image, light)
//             k_spec_pixels<L, kShade>   RGBA, using those norms
//   backward  k_spec_pixels<L, kGsum>    G = sum_p (dLoss/d rn_p) rdc_p per (image, light)
//             k_accumulate_rows<SpecGradFn<L>>   everything else: the pixel's shading is
//                                        recomputed and back-propagated to 12 interpolated
//                                        attributes (normal, position, diffuse, specular), the
//                                        barycentrics, the clip-space corners (cpp:202-269), the
//                                        lights and the camera position; 45 sums per triangle
//                                        (3 barycentrics x 15 factors) through the same
//                                        parked-factor reduction as the diffuse kernel
//             k_spec_scatter             45 sums per touched triangle -> vertex arrays
//
// Through the norm, EVERY pixel (masked ones and the background too) receives
// d rdc_p = d rn_p / norm - rdc_p G / norm^3; background pixels all carry the same attributes
// (-1), so their contribution to the light / camera gradients is evaluated once per lane and
// multiplied by the number of background pixels the lane has seen.
//
// Shininess: one exponent per image (float, 0-D or [B]) or, PV = true, per vertex ([B,V]: a 13th
// interpolated attribute, render.py:171-181, 222-224).  Both are differentiated (torch.pow's
// exponent rule: result * ln(base), 0 where base == 0 and exponent >= 0).  Only pixels that pass
// the render.py:215 mask evaluate the power: the reference multiplies the masked pixels' zero
// upstream gradient with pow(0, -1) = inf of the background's exponent -1 and returns NaN
// gradients for every per-vertex-shininess call with a background pixel; here those pixels
// contribute exactly what the mask says, nothing.  1..4 lights.
// fp32 with FMA contraction and 1-ulp v_rcp / v_sqrt / v_exp / v_log: parity budget 1e-4.
#include "corner_rec.h"
#include "spec_pixel.h"

#ifndef MR_SPEC_NT
#define MR_SPEC_NT 1  // nontemporal access to the streamed planes
#endif
#ifndef MR_SPEC_LANES
#define MR_SPEC_LANES 1  // 0: the backward's pixel pass always runs the rows kernel (A/B)
#endif

namespace mr {
extern thread_local int g_deterministic;  // mr_set_deterministic (shade.hip)
namespace {

constexpr int kThreads = 256;
constexpr float kNormEps = 1e-12f;         // torch.nn.functional.normalize default eps
constexpr float kDegenerateCutoff = 0.9f;  // rasterize_triangles.cpp:13
constexpr int kAttrMax = 13;               // normal, position, diffuse, specular (+ shininess)
constexpr int attr_count(bool pv) { return pv ? 13 : 12; }
constexpr float kLn2 = 0.6931471805599453f;

__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float fast_sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }

// The triangle's corner attributes, row-major [corner][attribute]: 36 floats in 9 float4, or 39
// (+1 pad) in 10 with the per-vertex shininess.
template <int A>
struct alignas(16) SpecCornerRec {
  static constexpr int kQuads = (3 * A + 3) / 4;
  float4 q[kQuads];
};
template <int A>
struct SpecCorners {
  float c[3][A];
};

template <int A>
__device__ __forceinline__ void load_spec_corners(const SpecCornerRec<A> *__restrict__ rec, SpecCorners<A> &o) {
#pragma unroll
  for (int q = 0; q < SpecCornerRec<A>::kQuads; ++q) {
    const float4 f = rec->q[q];
    const float v[4] = {f.x, f.y, f.z, f.w};
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (4 * q + j < 3 * A) o.c[(4 * q + j) / A][(4 * q + j) % A] = v[j];
  }
}

template <int A>
__global__ __launch_bounds__(kThreads) void k_spec_corner_setup(
    const F3 *__restrict__ normals, const F3 *__restrict__ positions, const F3 *__restrict__ diffuse,
    const F3 *__restrict__ specular, const float *__restrict__ shininess_v, const int32_t *__restrict__ tris,
    int B, int V, int T, SpecCornerRec<A> *__restrict__ out) {
  const long gid = (long)blockIdx.x * kThreads + threadIdx.x;
  if (gid >= (long)B * T) return;
  const int b = (int)(gid / T);
  const int t = (int)(gid - (long)b * T);
  float v[4 * SpecCornerRec<A>::kQuads];
#pragma unroll
  for (int k = 3 * A; k < 4 * SpecCornerRec<A>::kQuads; ++k) v[k] = 0.0f;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    int vi = tris[3 * t + k];
    if ((unsigned)vi >= (unsigned)V) vi = 0;
    const size_t at = (size_t)b * V + vi;
    const F3 n = normals[at], p = positions[at], d = diffuse[at], sp = specular[at];
    v[k * A + 0] = n.x; v[k * A + 1] = n.y; v[k * A + 2] = n.z;
    v[k * A + 3] = p.x; v[k * A + 4] = p.y; v[k * A + 5] = p.z;
    v[k * A + 6] = d.x; v[k * A + 7] = d.y; v[k * A + 8] = d.z;
    v[k * A + 9] = sp.x; v[k * A + 10] = sp.y; v[k * A + 11] = sp.z;
    if (A == 13) v[k * A + 12] = shininess_v[at];
  }
#pragma unroll
  for (int q = 0; q < SpecCornerRec<A>::kQuads; ++q)
    out[gid].q[q] = make_float4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
}

// Pointers to the per-image scene parameters (all wave-uniform once indexed by the image).
struct SpecSceneIn {
  const float *__restrict__ light_pos;   // [B,L,3]
  const float *__restrict__ light_col;   // [B,L,3]
  const float *__restrict__ ambient;     // [B,3] or nullptr
  const float *__restrict__ camera;      // [B,3]
  const float *__restrict__ shininess;   // [B]; null with per-vertex exponents (they ride the corner records)
  const float *__restrict__ norms2;      // [B,L]  sum over pixels of rdc^2 (null in the norm pass)
  const float *__restrict__ gsum;        // [B,L]  G (null outside the final backward pass)
};

template <int L>
struct SpecScene {
  float lp[L][3], li[L][3], amb[3], cam[3], shin;
  float inv_norm[L];   // 1 / max(sqrt(norms2), eps)
  float gcoef[L];      // G / norm^3, or 0 where the norm sits on its eps clamp
};

template <int L>
__device__ __forceinline__ void load_scene(const SpecSceneIn &in, int img, SpecScene<L> &sc) {
#pragma unroll
  for (int l = 0; l < L; ++l) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      sc.lp[l][c] = in.light_pos[((size_t)img * L + l) * 3 + c];
      sc.li[l][c] = in.light_col[((size_t)img * L + l) * 3 + c];
    }
    float inv = 0.0f, gc = 0.0f;
    if (in.norms2) {
      const float nrm = fast_sqrt(in.norms2[(size_t)img * L + l]);
      inv = fast_rcp(fmaxf(nrm, kNormEps));
      if (in.gsum && nrm > kNormEps) gc = in.gsum[(size_t)img * L + l] * inv * inv * inv;
    }
    sc.inv_norm[l] = inv;
    sc.gcoef[l] = gc;
  }
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    sc.amb[c] = in.ambient ? in.ambient[(size_t)img * 3 + c] : 0.0f;
    sc.cam[c] = in.camera[(size_t)img * 3 + c];
  }
  sc.shin = in.shininess ? in.shininess[img] : 0.0f;
}

// alpha = clamp(sum(2*bary), 0, 1); attr = alpha * interp + (1 - alpha) * (-1)
// (rasterize.py:137-150 with render.py:197's background of -1).
template <int A>
__device__ __forceinline__ void interpolate_attrs(const SpecCorners<A> &cr, const F3 b, float &pre, float &alpha,
                                                  float (&at)[A]) {
  pre = (2.0f * b.x + 2.0f * b.y) + 2.0f * b.z;
  alpha = fminf(fmaxf(pre, 0.0f), 1.0f);
  const float one_m = 1.0f - alpha;
#pragma unroll
  for (int a = 0; a < A; ++a) {
    const float interp = (cr.c[0][a] * b.x + cr.c[1][a] * b.y) + cr.c[2][a] * b.z;
    at[a] = alpha * interp - one_m;
  }
}

// (PixelFrame / LightTerm: the light-independent geometry of a pixel and one light's term, spec_pixel.h)
using spec::LightTerm;
using spec::PixelFrame;
using spec::light_term;
using spec::pixel_frame;

// rn -> clamp -> where(ndl != 0) -> pow (render.py:342-366): value, d value / d rn and
// d value / d shininess.  torch.pow(x, y) for x in [0, 1]: pow(x, 0) = 1, pow(0, y > 0) = 0,
// pow(0, y < 0) = inf; v_log_f32(0) = -inf makes v_exp_f32 give the 0 / inf.
struct SpecTerm {
  float spec, dspec_drn, dspec_dshin;
};
__device__ __forceinline__ void specularity(float rdc, float inv_norm, float ndl, float shin, SpecTerm &o) {
  const float rn = rdc * inv_norm;
  const float rc = fminf(fmaxf(rn, 0.0f), 1.0f);
  const bool lit = ndl != 0.0f;
  const float rw = lit ? rc : 0.0f;
  const float lg = __builtin_amdgcn_logf(rw);  // log2
  o.spec = shin == 0.0f ? 1.0f : __builtin_amdgcn_exp2f(shin * lg);
  // torch: d pow / d base = y * base^(y-1), 0 where y == 0; clamp passes the gradient on [0, 1] inclusively
  const float dpow = shin == 0.0f ? 0.0f : shin * (shin == 1.0f ? 1.0f : __builtin_amdgcn_exp2f((shin - 1.0f) * lg));
  o.dspec_drn = (lit && rn >= 0.0f && rn <= 1.0f) ? dpow : 0.0f;
  // torch: d pow / d exponent = result * ln(base), 0 where base == 0 and exponent >= 0
  o.dspec_dshin = (rw == 0.0f && shin >= 0.0f) ? 0.0f : o.spec * (lg * kLn2);
}

enum SpecPass { kNorms = 0, kShade = 1, kGsum = 2 };

// One thread per pixel.  kNorms: per-(image, light) sum of rdc^2 over ALL pixels.  kShade: RGBA.
// kGsum: per-(image, light) sum of (dLoss / d rn) * rdc.  The two sums leave as one partial per
// (workgroup, light) -- sums_out[workgroup][L] -- and k_spec_sum adds an image's partials in a fixed
// order.  (They used to be float atomics on the image's L addresses: 4096 workgroups per image queued
// up on the same cache line, 1.45 ms for the norm pass at 1024^2 x 32 against 0.26 ms for the shading
// pass that does more arithmetic.)
#ifndef MR_SPEC_ROWS
#define MR_SPEC_ROWS 4   // rows per workgroup, in batches whose G-buffer loads are all in flight before the first pixel's math
#endif
#ifndef MR_SPEC_CACHE_RECORD
#define MR_SPEC_CACHE_RECORD 0
#endif
#ifndef MR_SPEC_BATCH
#define MR_SPEC_BATCH 4
#endif
constexpr int kSpecRows = MR_SPEC_ROWS, kSpecBatch = MR_SPEC_BATCH;
static_assert(kSpecRows % kSpecBatch == 0, "whole batches");
template <int L, int PASS, bool PV>
__global__ __launch_bounds__(kThreads) void k_spec_pixels(
    const int32_t *__restrict__ ids, const F3 *__restrict__ bary,
    const SpecCornerRec<attr_count(PV)> *__restrict__ corners, SpecSceneIn scene_in, int T, int W, int H, int x_blocks,
    int y_blocks, const float4 *__restrict__ drgba, float4 *__restrict__ rgba_out, float *__restrict__ sums_out) {
  const int blk = (int)blockIdx.x;
  const int img = blk / (x_blocks * y_blocks);
  const int rem = blk - img * (x_blocks * y_blocks);
  const int yb = rem / x_blocks, xb = rem - yb * x_blocks;
  const int x = xb * kThreads + (int)threadIdx.x;
  constexpr int A = attr_count(PV);
  SpecScene<L> sc;
  load_scene(scene_in, img, sc);
  float part[L];
#pragma unroll
  for (int l = 0; l < L; ++l) part[l] = 0.0f;
  // One thread per pixel and workgroup launched 1.3 dependent round trips (G-buffer -> corner record) per
  // 256 pixels with nothing else in flight: 3.1 TB/s for the read-only norm pass.  All rows' G-buffer loads go
  // out first.
  // (MR_SPEC_CACHE_RECORD: the thread keeps its triangle's 144-byte corner record while it stays on that
  // triangle going down a 16-row column instead of gathering it per pixel.  Measured slower -- norm pass 153
  // -> 195 us, the other two +5..20 -- the gather is not what binds these passes.)
  SpecCorners<A> cr;
  int cached_t = -1;
#pragma unroll 1
  for (int r0 = 0; r0 < kSpecRows; r0 += kSpecBatch) {
  F3 b_row[kSpecBatch];
  int t_row[kSpecBatch];
  float4 g_row[kSpecBatch];
#pragma unroll
  for (int r = 0; r < kSpecBatch; ++r) {
    const int y = yb * kSpecRows + r0 + r;
    b_row[r] = F3{0.f, 0.f, 0.f};
    t_row[r] = -1;
    g_row[r] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (x < W && y < H) {
      const size_t pix = ((size_t)img * H + y) * W + x;
      b_row[r] = MR_SPEC_NT ? load_streamed(&bary[pix]) : bary[pix];
      t_row[r] = MR_SPEC_NT ? __builtin_nontemporal_load(&ids[pix]) : ids[pix];
      if (PASS == kGsum) {
        const size_t out_pix = ((size_t)img * H + (H - 1 - y)) * W + x;
        g_row[r] = MR_SPEC_NT ? load_streamed(&drgba[out_pix]) : drgba[out_pix];
      }
    }
  }
#pragma unroll
  for (int r = 0; r < kSpecBatch; ++r) {
    const int y = yb * kSpecRows + r0 + r;
    if (!(x < W && y < H)) continue;
    const F3 b = b_row[r];
    const int t = t_row[r];
    float at[A], pre = 0.0f, alpha = 0.0f;
    const bool live = ((2.0f * b.x + 2.0f * b.y) + 2.0f * b.z) > 0.0f && (unsigned)t < (unsigned)T;
    if (live) {
      if (!MR_SPEC_CACHE_RECORD || t != cached_t) {
        load_spec_corners(corners + (size_t)img * T + t, cr);
        cached_t = t;
      }
      interpolate_attrs(cr, b, pre, alpha, at);
    } else {
#pragma unroll
      for (int a = 0; a < A; ++a) at[a] = -1.0f;  // the background of render.py:197
    }
    const bool mask = (at[6] >= 0.0f) || (at[7] >= 0.0f) || (at[8] >= 0.0f);  // render.py:215
    const size_t out_pix = ((size_t)img * H + (H - 1 - y)) * W + x;            // render.py:384-386 flip
    if (PASS == kNorms || mask) {
      PixelFrame f;
      pixel_frame(at, sc.cam, f);
      float rgb[3] = {sc.amb[0] * at[6], sc.amb[1] * at[7], sc.amb[2] * at[8]};
      const float g[3] = {g_row[r].x, g_row[r].y, g_row[r].z};
#pragma unroll
      for (int l = 0; l < L; ++l) {
        LightTerm lt;
        light_term(at, f, sc.lp[l], lt);
        if (PASS == kNorms) {
          part[l] += lt.rdc * lt.rdc;
        } else {
          SpecTerm st;  // only masked-in pixels get here
          specularity(lt.rdc, sc.inv_norm[l], lt.ndl, PV ? at[A - 1] : sc.shin, st);
          if (PASS == kShade) {
#pragma unroll
            for (int c = 0; c < 3; ++c) rgb[c] += (at[6 + c] * lt.ndl + at[9 + c] * st.spec) * sc.li[l][c];
          } else {
            const float dspec = (g[0] * at[9] * sc.li[l][0] + g[1] * at[10] * sc.li[l][1]) + g[2] * at[11] * sc.li[l][2];
            part[l] += dspec * st.dspec_drn * lt.rdc;
          }
        }
      }
      if (PASS == kShade) {
        if (MR_SPEC_NT) store_streamed(&rgba_out[out_pix], make_float4(rgb[0], rgb[1], rgb[2], 1.0f));
        else rgba_out[out_pix] = make_float4(rgb[0], rgb[1], rgb[2], 1.0f);
      }
    } else if (PASS == kShade) {
      if (MR_SPEC_NT) store_streamed(&rgba_out[out_pix], make_float4(0.f, 0.f, 0.f, 0.f));
      else rgba_out[out_pix] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  }
  if (PASS != kShade) {  // workgroup-uniform
    __shared__ float s_part[kThreads / kWave][L];
    const int lane = (int)threadIdx.x & (kWave - 1), wave = (int)threadIdx.x >> 6;
#pragma unroll
    for (int l = 0; l < L; ++l) {
      float v = part[l];
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
      if (lane == 0) s_part[wave][l] = v;
    }
    __syncthreads();
    if ((int)threadIdx.x < L) {
      float v = 0.0f;
#pragma unroll
      for (int w = 0; w < kThreads / kWave; ++w) v += s_part[w][threadIdx.x];
      sums_out[(size_t)blk * L + threadIdx.x] = v;
    }
  }
}

// One workgroup per image: out[image][l] = the sum of its per-workgroup partials, fixed order.
__global__ __launch_bounds__(kThreads) void k_spec_sum(const float *__restrict__ partials, int per_image, int L,
                                                       float *__restrict__ out) {
  __shared__ float s_part[kThreads / kWave];
  const int img = (int)blockIdx.x, tid = (int)threadIdx.x;
  const float *mine = partials + (size_t)img * per_image * L;
  for (int l = 0; l < L; ++l) {
    float v = 0.0f;
    for (int i = tid; i < per_image; i += kThreads) v += mine[(size_t)i * L + l];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
    if ((tid & (kWave - 1)) == 0) s_part[tid >> 6] = v;
    __syncthreads();
    if (tid == 0) {
      float t = 0.0f;
      for (int w = 0; w < kThreads / kWave; ++w) t += s_part[w];
      out[(size_t)img * L + l] = t;
    }
    __syncthreads();
  }
}

// ---- the final backward pass, inside k_accumulate_rows (run_accum.h) -----------------------
// SIGNS (round 5, the coupled lane kernel only): the upstream gradient is the backward of mean|image - target|
// (loss.hip) and arrives as that loss's packed sign codes, one byte per pixel instead of sixteen; the pass is linear
// in the upstream gradient, so the sums are formed from the bare codes and the gather multiplies by the loss's scale.
template <int L, bool PV, bool SIGNS = false>
struct SpecGradFn {
  static constexpr int kA = attr_count(PV);
  static constexpr int kN = 3 * kA + 9;  // attribute partials [corner][attr] + 9 clip partials: 45 / 48
  static constexpr int kStride = 48;
  static constexpr int kRowsPerWave = 16;  // see run_accum.h
  static constexpr int kSlots = 256;
  static constexpr int kMinWavesPerSimd = 2;
  static constexpr bool kCountBackground = true;
  // parked per pixel: b[3] | y[kA] = alpha * d/d attr | q[3] = clip brackets
  static constexpr int kFactors = kA + 6;
  static constexpr int kFactorStride = 20;
  __device__ static void factor_pair(int o, int &ia, int &ib) {
    if (o < 3 * kA) { ia = o / kA; ib = 3 + o % kA; }
    else { ia = (o - 3 * kA) / 3; ib = 3 + kA + (o - 3 * kA) % 3; }
  }
  static constexpr int kLightRow = L * 6 + 7;
  const float4 *__restrict__ drgba;   // [B,H,W,4], image rows (flipped w.r.t. the G-buffer)
  const int32_t *__restrict__ ids;
  const F3 *__restrict__ bary;
  const SpecCornerRec<kA> *__restrict__ corners;
  const BwdRec *__restrict__ recs;
  SpecSceneIn scene_in;
  // [B][L*6 + 7]: dpos (L x 3), dcol (L x 3), dambient (3), dcamera (3), d per-image shininess (1)
  float *__restrict__ light_rows;   // [strips][kLightRow]: every strip's row of image-wide sums
  int T_, W, H;
  const uint8_t *__restrict__ signs;  // SIGNS: [B,H,W] bytes, image rows like drgba

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
    SpecCorners<kA> cr;
    BwdTriangle bt;
  };
  struct Image {
    SpecScene<L> sc;
    float dpos[L][3], dcol[L][3], damb[3], dcam[3], dshin;  // per-lane partial sums
    int n_bg;                                         // background pixels this lane has seen
  };

  __device__ __forceinline__ void begin_image(int img, Image &im) const {
    load_scene(scene_in, img, im.sc);
#pragma unroll
    for (int l = 0; l < L; ++l)
#pragma unroll
      for (int c = 0; c < 3; ++c) { im.dpos[l][c] = 0.f; im.dcol[l][c] = 0.f; }
#pragma unroll
    for (int c = 0; c < 3; ++c) { im.damb[c] = 0.f; im.dcam[c] = 0.f; }
    im.dshin = 0.f;
    im.n_bg = 0;
  }

  __device__ __forceinline__ void fetch(int img, int x, int y, size_t pix, Raw &r) const {
    r.b = MR_SPEC_NT ? load_streamed(&bary[pix]) : bary[pix];
    r.t = MR_SPEC_NT ? __builtin_nontemporal_load(&ids[pix]) : ids[pix];
    const size_t image_pix = ((size_t)img * H + (H - 1 - y)) * W + x;  // un-flip
    if (SIGNS) {
      r.code = MR_SPEC_NT ? __builtin_nontemporal_load(&signs[image_pix]) : signs[image_pix];
    } else {
      r.g = MR_SPEC_NT ? load_streamed(&drgba[image_pix]) : drgba[image_pix];
    }
  }
  __device__ __forceinline__ bool prepare(const Raw &r, int T, int &tri, Pixel &p) const {
    const float pre = (2.0f * r.b.x + 2.0f * r.b.y) + 2.0f * r.b.z;
    if (!(pre > 0.0f)) return false;  // background: counted by the kernel, handled in end_image
    if ((unsigned)r.t >= (unsigned)T) return false;
    p.b = r.b;
    if (SIGNS) {  // 2-bit two's-complement codes 0, +1, -1 (loss.hip: sign_code)
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
    load_spec_corners(corners + (size_t)img * T_ + tri, t.cr);
    load_bwd_triangle(recs + (size_t)img * T_ + tri, t.bt);
  }

  // Back-propagates (g = dLoss/d rgb of this pixel, plus the norm coupling) to the interpolated
  // attributes `dat`, accumulating light / ambient / camera gradients in `im` scaled by `weight`.
  // `shaded` = the pixel passed the render.py:215 mask; the others have g = 0 and only feel the norm.
  // SUMS = false (the lane kernel below, which serves callers that want no image-wide gradient): those sums are
  // not formed.
  template <bool SUMS = true, class Im = Image>
  __device__ __forceinline__ void shade_backward(const float (&at)[kA], const float (&g)[3], bool shaded,
                                                 float weight, Im &im, float (&dat)[kA]) const {
    const SpecScene<L> &sc = im.sc;
    float dshin = 0.f;
    PixelFrame f;
    pixel_frame(at, sc.cam, f);
    float dN[3] = {0.f, 0.f, 0.f}, dP[3] = {0.f, 0.f, 0.f}, dCd[3] = {0.f, 0.f, 0.f};
    float dKd[3] = {g[0] * sc.amb[0], g[1] * sc.amb[1], g[2] * sc.amb[2]};
    float dKs[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < 3; ++c)
      if constexpr (SUMS) im.damb[c] += weight * g[c] * at[6 + c];
#pragma unroll
    for (int l = 0; l < L; ++l) {
      LightTerm lt;
      light_term(at, f, sc.lp[l], lt);
      SpecTerm st{0.f, 0.f, 0.f};
      if (shaded) specularity(lt.rdc, sc.inv_norm[l], lt.ndl, PV ? at[kA - 1] : sc.shin, st);
      float t_l = 0.f, dspec = 0.f;  // d/d ndl of the diffuse term, d/d spec
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        dKd[c] += g[c] * lt.ndl * sc.li[l][c];
        dKs[c] += g[c] * st.spec * sc.li[l][c];
        if constexpr (SUMS) im.dcol[l][c] += weight * g[c] * (at[6 + c] * lt.ndl + at[9 + c] * st.spec);
        t_l += g[c] * at[6 + c] * sc.li[l][c];
        dspec += g[c] * at[9 + c] * sc.li[l][c];
      }
      dshin += dspec * st.dspec_dshin;
      // rn = rdc / norm with norm = |rdc over all pixels|: d rdc = d rn / norm - rdc G / norm^3
      const float d_rdc = dspec * st.dspec_drn * sc.inv_norm[l] - lt.rdc * sc.gcoef[l];
      float dM[3], dot_m = 0.f;
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        dM[k] = d_rdc * f.Cd[k];
        dCd[k] += d_rdc * lt.M[k];
        dot_m += lt.M[k] * dM[k];
      }
      // backward of m / max(|m|, eps), m = 2 ndl N - D
      float dm[3], n_dot_dm = 0.f;
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        dm[k] = (lt.mn > kNormEps ? (dM[k] - lt.M[k] * dot_m) : dM[k]) * lt.inv_mn;
        n_dot_dm += f.N[k] * dm[k];
      }
      const float d_ndl = t_l + 2.0f * n_dot_dm;
      float dD[3];
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        dN[k] += 2.0f * lt.ndl * dm[k];
        dD[k] = -dm[k];
      }
      if (lt.pre >= 0.0f && lt.pre <= 1.0f) {  // torch.clamp passes the gradient inclusively
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          dN[k] += d_ndl * lt.D[k];
          dD[k] += d_ndl * f.N[k];
        }
      }
      float dd = 0.f;
#pragma unroll
      for (int k = 0; k < 3; ++k) dd += lt.D[k] * dD[k];
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const float dv = (lt.vn > kNormEps ? (dD[k] - lt.D[k] * dd) : dD[k]) * lt.inv_vn;
        if constexpr (SUMS) im.dpos[l][k] += weight * dv;
        dP[k] -= dv;
      }
    }
    {  // Cd = c / max(|c|, eps), c = camera - P
      float dc_dot = 0.f;
#pragma unroll
      for (int k = 0; k < 3; ++k) dc_dot += f.Cd[k] * dCd[k];
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const float dc = (f.cn > kNormEps ? (dCd[k] - f.Cd[k] * dc_dot) : dCd[k]) * f.inv_cn;
        if constexpr (SUMS) im.dcam[k] += weight * dc;
        dP[k] -= dc;
      }
    }
    {
      const float nd = f.N[0] * dN[0] + f.N[1] * dN[1] + f.N[2] * dN[2];
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        dat[c] = (f.nn > kNormEps ? (dN[c] - f.N[c] * nd) : dN[c]) * f.inv_nn;
        dat[3 + c] = dP[c];
        dat[6 + c] = dKd[c];
        dat[9 + c] = dKs[c];
      }
      if (PV) dat[kA - 1] = dshin;
      else if constexpr (SUMS) im.dshin += weight * dshin;
    }
  }

  __device__ __forceinline__ void factors(const Pixel &p, const Triangle &t, float (&f)[kFactorStride],
                                          Image &im) const {
    float pre, alpha, at[kA];
    interpolate_attrs(t.cr, p.b, pre, alpha, at);
    // render.py:215 mask: where() sends no rgb gradient to a masked pixel (the norm coupling
    // still reaches it)
    const bool mask = (at[6] >= 0.0f) || (at[7] >= 0.0f) || (at[8] >= 0.0f);
    const float g[3] = {mask ? p.g.x : 0.f, mask ? p.g.y : 0.f, mask ? p.g.z : 0.f};
    float dat[kA];
    shade_backward(at, g, mask, 1.0f, im, dat);
    // interpolation backward (rasterize.py:137-150); interp[a] + 1 = (at[a] + 1) / alpha
    float dalpha_a = 0.f, db[3] = {0.f, 0.f, 0.f};
    f[0] = p.b.x; f[1] = p.b.y; f[2] = p.b.z;
#pragma unroll
    for (int a = 0; a < kA; ++a) {
      const float di = alpha * dat[a];
      dalpha_a += dat[a] * (at[a] + 1.0f);
#pragma unroll
      for (int k = 0; k < 3; ++k) db[k] += di * t.cr.c[k][a];
      f[3 + a] = di;
    }
    const float dpre = (pre >= 0.0f && pre <= 1.0f) ? 2.0f * dalpha_a * fast_rcp(alpha) : 0.0f;
    F3 dbary;
    dbary.x = db[0] + dpre; dbary.y = db[1] + dpre; dbary.z = db[2] + dpre;
    // rasterizer backward (cpp:162 skip rule, then cpp:202-269)
    const bool skip = p.tri == 0 && (p.b.x + p.b.y) + p.b.z < kDegenerateCutoff;
    float q[3];
    raster_pixel_q(p.b, dbary, t.bt, skip ? 0.f : t.bt.inv, q);
    f[3 + kA] = q[0]; f[4 + kA] = q[1]; f[5 + kA] = q[2];
#pragma unroll
    for (int k = kFactors; k < kFactorStride; ++k) f[k] = 0.f;
  }

  // the strip's row of light / camera / shininess sums (k_sum_strip_rows adds an image's rows up)
  __device__ __forceinline__ void end_strip(int img, int strip, Image &im) const {
    (void)img;
    if (im.n_bg > 0) {  // all background pixels carry the attributes -1: evaluate once, weight by the count
      float at[kA], dat[kA];
#pragma unroll
      for (int a = 0; a < kA; ++a) at[a] = -1.0f;
      const float g[3] = {0.f, 0.f, 0.f};
      shade_backward(at, g, false, (float)im.n_bg, im, dat);
    }
    float *dst = light_rows + (size_t)strip * kLightRow;
    const int lane = lane_id();
    auto reduce_add = [&](float v, int slot) {
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
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
    for (int c = 0; c < 3; ++c) {
      reduce_add(im.damb[c], L * 6 + c);
      reduce_add(im.dcam[c], L * 6 + 3 + c);
    }
    reduce_add(PV ? 0.0f : im.dshin, L * 6 + 6);
  }
};

// ---- the same pass for callers that want the VERTEX gradients only (round 4) ----------------------------------
// render() differentiated to the vertices alone -- the optimisation loops of the reference's tests -- needs, of the
// 45 sums per triangle above, the nine of the position attribute and the nine clip-space ones, and none of the
// image-wide sums (lights, camera, per-image shininess).  That fits k_accumulate_lanes (run_accum.h: the sums stay
// in registers down a lane's vertical run; the rows kernel spends one LDS reduction step per pixel), in the
// difference basis of shade.hip's ShadeFoldLaneFn: for a G-buffer whose barycentrics sum to 1 (MR_GBUFFER_NORMALISED:
// alpha = 1, nothing flows through it)
//   at = c2 + b0 e0 + b1 e1,  dat = shade_backward(at, g),  g0 = dat . e0,  g1 = dat . e1,
//   q_c = (g0 (s_c b0 - u_0c) + g1 (s_c b1 - u_1c)) / |det|          (cpp:202-269 less the common shift d L / d b2)
//   a[k][c] += b_k dat[3 + c]   and   a[9 + k][c] += b_k q_c,
// or, FOLD (the caller has the transforms M with clip = M (position, 1) and wants no clip gradient of its own):
//   a[k][c] += b_k (dat[3 + c] + (M^T q)_c), nine sums.
// Record per (image, triangle): e0[A] e1[A] c2[A] | u0[3] u1[3] s[3] 1/|det|  (3 A + 10 floats, 192 / 208 bytes).
template <int A>
struct alignas(16) SpecFoldRec {
  static constexpr int kFloats = 3 * A + 10;
  static constexpr int kQuads = (kFloats + 3) / 4;
  float4 q[kQuads];
};
template <int A>
struct SpecFoldTriangle {
  float e0[A], e1[A], c2[A], u0[3], u1[3], s[3], inv;
};
template <int A>
__device__ __forceinline__ void load_spec_fold_triangle(const SpecFoldRec<A> *__restrict__ rec, SpecFoldTriangle<A> &t) {
  float v[4 * SpecFoldRec<A>::kQuads];
#pragma unroll
  for (int q = 0; q < SpecFoldRec<A>::kQuads; ++q) {
    const float4 f = rec->q[q];
    v[4 * q] = f.x; v[4 * q + 1] = f.y; v[4 * q + 2] = f.z; v[4 * q + 3] = f.w;
  }
#pragma unroll
  for (int a = 0; a < A; ++a) { t.e0[a] = v[a]; t.e1[a] = v[A + a]; t.c2[a] = v[2 * A + a]; }
#pragma unroll
  for (int c = 0; c < 3; ++c) { t.u0[c] = v[3 * A + c]; t.u1[c] = v[3 * A + 3 + c]; t.s[c] = v[3 * A + 6 + c]; }
  t.inv = v[3 * A + 9];
}

// One thread per (image, triangle): SpecFoldRec from the corner record and the clip-space corners (the sign-corrected
// adjugate, its column sums and 1 / |det| exactly as k_bwd_setup forms them, rasterize_triangles.cpp:180-198).
// pull_transforms ([B,4,4] or nullptr, round 5): the last ten slots carry the rasterizer's backward pulled back to world
// space per triangle -- S[3], P0[3], P1[3] as corner_rec.h's store_fold_record forms them -- instead of u0, u1, s, 1/|det|.
template <int A>
__global__ __launch_bounds__(kThreads) void k_spec_fold_setup(const float4 *__restrict__ clip, const int32_t *__restrict__ tris,
                                                              const SpecCornerRec<A> *__restrict__ corners, int B, int V,
                                                              int T, SpecFoldRec<A> *__restrict__ out,
                                                              const float *__restrict__ pull_transforms) {
  const long gid = (long)blockIdx.x * kThreads + threadIdx.x;
  if (gid >= (long)B * T) return;
  const int b = (int)(gid / T);
  const int t = (int)(gid - (long)b * T);
  SpecCorners<A> cr;
  load_spec_corners(corners + gid, cr);
  float v[4 * SpecFoldRec<A>::kQuads];
#pragma unroll
  for (int a = 0; a < A; ++a) {
    v[a] = cr.c[0][a] - cr.c[2][a];
    v[A + a] = cr.c[1][a] - cr.c[2][a];
    v[2 * A + a] = cr.c[2][a];
  }
#pragma unroll
  for (int k = 3 * A; k < 4 * SpecFoldRec<A>::kQuads; ++k) v[k] = 0.0f;   // a triangle with a foreign corner is never drawn
  const int i0 = tris[3 * t], i1 = tris[3 * t + 1], i2 = tris[3 * t + 2];
  if ((unsigned)i0 < (unsigned)V && (unsigned)i1 < (unsigned)V && (unsigned)i2 < (unsigned)V) {
#pragma clang fp contract(off)   // the adjugate's differences of products round as in k_bwd_setup (raster_backward.hip)
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
    const float s0 = (u0 + u3) + u6, s1 = (u1 + u4) + u7, s2 = (u2 + u5) + u8;   // cpp:187-198
    const float inv = 1.0f / fabsf(det);
    if (pull_transforms) {
      float pull[12];
      load_pull_rows(pull_transforms, b, pull);
#pragma unroll
      for (int cw = 0; cw < 3; ++cw) {
        const float q0 = pull[cw], q1 = pull[4 + cw], q2 = pull[8 + cw];
        v[3 * A + cw] = ((q0 * s0 + q1 * s1) + q2 * s2) * inv;
        v[3 * A + 3 + cw] = -(((q0 * u0 + q1 * u1) + q2 * u2) * inv);
        v[3 * A + 6 + cw] = -(((q0 * u3 + q1 * u4) + q2 * u5) * inv);
      }
      v[3 * A + 9] = 0.0f;
    } else {
      v[3 * A + 0] = u0; v[3 * A + 1] = u1; v[3 * A + 2] = u2;
      v[3 * A + 3] = u3; v[3 * A + 4] = u4; v[3 * A + 5] = u5;
      v[3 * A + 6] = s0; v[3 * A + 7] = s1; v[3 * A + 8] = s2;
      v[3 * A + 9] = inv;
    }
  }
#pragma unroll
  for (int q = 0; q < SpecFoldRec<A>::kQuads; ++q)
    out[gid].q[q] = make_float4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
}

#ifndef MR_SPEC_LANE_ROWS
#define MR_SPEC_LANE_ROWS 8
#endif
#ifndef MR_SPEC_LANE_WAVES
#define MR_SPEC_LANE_WAVES 3
#endif
template <int L, bool PV, bool FOLD>
struct SpecFoldLaneFn : SpecGradFn<L, PV> {
  using Base = SpecGradFn<L, PV>;
  static constexpr int kA = Base::kA;
  static constexpr int kN = FOLD ? 9 : 18;
  static constexpr int kStride = Base::kStride;   // the rows of acc keep SpecGradFn's layout: k_spec_gather reads them
  static constexpr int kLaneRowsPerWave = MR_SPEC_LANE_ROWS;
  static constexpr int kMinWavesPerSimd = L <= 2 ? MR_SPEC_LANE_WAVES : 2;   // (three or four lights at 168 registers: spills)
  static constexpr bool kCountBackground = false;   // (background pixels only feed the image-wide sums)
  const SpecFoldRec<kA> *__restrict__ fold_recs;
  const float *__restrict__ transforms;   // FOLD: [B,4,4]
  using Triangle = SpecFoldTriangle<kA>;
  struct Image {
    SpecScene<L> sc;
    float pull[3][3];   // FOLD: pull[r][c] = M[{0, 1, 3}[r]][c], the clip x / y / w rows' position columns
    int n_bg;           // unused
  };
  // sum o = corner * 3 + c -> position attribute, acc column corner * kA + 3 + c;  o = 9 + corner * 3 + c -> clip
  __device__ static int column(int o) {
    return o < 9 ? (o / 3) * kA + 3 + o % 3 : 3 * kA + (o - 9);
  }
  __device__ __forceinline__ void begin_image(int img, Image &im) const {
    load_scene(this->scene_in, img, im.sc);
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int c = 0; c < 3; ++c) im.pull[r][c] = FOLD ? transforms[(size_t)img * 16 + (r == 2 ? 3 : r) * 4 + c] : 0.f;
    im.n_bg = 0;
  }
  __device__ __forceinline__ void load_triangle(int img, int tri, Triangle &t) const {
    load_spec_fold_triangle(fold_recs + (size_t)img * this->T_ + tri, t);
  }
  __device__ __forceinline__ void accumulate(const typename Base::Pixel &p, const Triangle &t, float (&a)[kN],
                                             Image &im) const {
    float at[kA], dat[kA];
#pragma unroll
    for (int k = 0; k < kA; ++k) at[k] = fmaf(p.b.x, t.e0[k], fmaf(p.b.y, t.e1[k], t.c2[k]));
    const bool mask = (at[6] >= 0.0f) || (at[7] >= 0.0f) || (at[8] >= 0.0f);   // render.py:215
    const float g[3] = {mask ? p.g.x : 0.f, mask ? p.g.y : 0.f, mask ? p.g.z : 0.f};
    Base::template shade_backward<false>(at, g, mask, 1.0f, im, dat);
    float g0 = 0.f, g1 = 0.f;
#pragma unroll
    for (int k = 0; k < kA; ++k) {
      g0 = fmaf(dat[k], t.e0[k], g0);
      g1 = fmaf(dat[k], t.e1[k], g1);
    }
    float q[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float w0 = t.s[c] * p.b.x - t.u0[c];
      const float w1 = t.s[c] * p.b.y - t.u1[c];
      q[c] = (g0 * w0 + g1 * w1) * t.inv;
    }
    float y[3];
#pragma unroll
    for (int c = 0; c < 3; ++c)
      y[c] = FOLD ? dat[3 + c] + ((im.pull[0][c] * q[0] + im.pull[1][c] * q[1]) + im.pull[2][c] * q[2]) : dat[3 + c];
    const float b[3] = {p.b.x, p.b.y, p.b.z};
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
      for (int c = 0; c < 3; ++c) a[k * 3 + c] = fmaf(b[k], y[c], a[k * 3 + c]);
    if (!FOLD) {
#pragma unroll
      for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int c = 0; c < 3; ++c) a[9 + k * 3 + c] = fmaf(b[k], q[c], a[9 + k * 3 + c]);
    }
  }
  __device__ __forceinline__ void end_strip(int, int, Image &) const {}
};

// ---- round 5: the vertex-only backward WITHOUT the separate G pass ----------------------------------------------
// The across-pixels norm couples every pixel's rdc to the image-wide sum G = sum_p (dL / d rn_p) rdc_p:
//   d rdc_p = a_p / norm - rdc_p G / norm^3,      a_p = dL / d rn_p.
// Everything downstream of d rdc_p is LINEAR in it, with a per-pixel direction J_p (the gradient of rdc_p w.r.t. the
// vertices, through the attributes and through the barycentrics): the vertex gradient is
//   sum_p [ y_diffuse,p + (a_p / norm) J_p ]  -  (G / norm^3) sum_p rdc_p J_p.
// k_spec_pixels<kGsum> used to find G in a pass of its own (0.25 ms at 1024^2 x 32: the whole shading recomputed)
// before the pixel pass could start.  Here ONE pass evaluates J_p once per pixel and light and keeps BOTH sums --
// S1 = sum_p (y_diffuse + (a_p / norm) J_p) and S2_l = sum_p rdc_p J_p, 9 + 9 L per triangle -- next to the per-lane
// partial sums of G; the per-vertex gather forms S1 - sum_l (G_l / norm_l^3) S2_l.  One or two lights, the clip-space
// pull-back folded into the record (pulled SpecFoldRec); three or four lights keep the two-pass scheme.
#ifndef MR_SPEC_COUPLED
#define MR_SPEC_COUPLED 1
#endif
template <int A>
struct SpecPulledTriangle {
  float e0[A], e1[A], c2[A], S[3], P0[3], P1[3];
};
template <int L, bool PV, bool SIGNS = false>
struct SpecCoupledLaneFn : SpecGradFn<L, PV, SIGNS> {
  using Base = SpecGradFn<L, PV, SIGNS>;
  static_assert(L == 1 || L == 2, "the second light's sums take the normal columns of the accumulator rows");
  static constexpr int kA = Base::kA;
  static constexpr int kN = 9 + 9 * L;
  static constexpr int kStride = Base::kStride;
  static constexpr int kLaneRowsPerWave = MR_SPEC_LANE_ROWS;
  static constexpr int kMinWavesPerSimd = L == 1 ? MR_SPEC_LANE_WAVES : 2;
  static constexpr bool kCountBackground = false;   // (background pixels do not depend on the vertices)
  const SpecFoldRec<kA> *__restrict__ fold_recs;    // in the pulled form
  float *__restrict__ g_rows;                       // [strips][L]: every strip's partial sums of G
  using Triangle = SpecPulledTriangle<kA>;
  struct Image {
    SpecScene<L> sc;
    float gpart[L];
    int n_bg;   // unused
  };
  // S1 -> the position columns; S2 of light 0 -> the clip columns; S2 of light 1 -> the normal columns
  __device__ static int column(int o) {
    if (o < 9) return (o / 3) * kA + 3 + o % 3;
    if (o < 18) return 3 * kA + (o - 9);
    return ((o - 18) / 3) * kA + (o - 18) % 3;
  }
  __device__ __forceinline__ void begin_image(int img, Image &im) const {
    load_scene(this->scene_in, img, im.sc);   // (scene_in.gsum is null: no coupling coefficient in this pass)
#pragma unroll
    for (int l = 0; l < L; ++l) im.gpart[l] = 0.f;
    im.n_bg = 0;
  }
  __device__ __forceinline__ void load_triangle(int img, int tri, Triangle &t) const {
    const SpecFoldRec<kA> *rec = fold_recs + (size_t)img * this->T_ + tri;
    float v[4 * SpecFoldRec<kA>::kQuads];
#pragma unroll
    for (int q = 0; q < SpecFoldRec<kA>::kQuads; ++q) {
      const float4 f = rec->q[q];
      v[4 * q] = f.x; v[4 * q + 1] = f.y; v[4 * q + 2] = f.z; v[4 * q + 3] = f.w;
    }
#pragma unroll
    for (int a = 0; a < kA; ++a) { t.e0[a] = v[a]; t.e1[a] = v[kA + a]; t.c2[a] = v[2 * kA + a]; }
#pragma unroll
    for (int c = 0; c < 3; ++c) { t.S[c] = v[3 * kA + c]; t.P0[c] = v[3 * kA + 3 + c]; t.P1[c] = v[3 * kA + 6 + c]; }
  }
  // y = (position attribute's gradient) + the barycentric path pulled back to world space, for attribute gradients d[n]
  template <int N_>
  __device__ __forceinline__ void pulled(const float (&d)[N_], const typename Base::Pixel &p, const Triangle &t,
                                         float (&y)[3]) const {
    float g0 = 0.f, g1 = 0.f;
#pragma unroll
    for (int k = 0; k < N_; ++k) {
      g0 = fmaf(d[k], t.e0[k], g0);
      g1 = fmaf(d[k], t.e1[k], g1);
    }
    const float h = fmaf(g0, p.b.x, g1 * p.b.y);
#pragma unroll
    for (int c = 0; c < 3; ++c) y[c] = fmaf(h, t.S[c], fmaf(g0, t.P0[c], fmaf(g1, t.P1[c], d[3 + c])));
  }
  __device__ __forceinline__ void accumulate(const typename Base::Pixel &p, const Triangle &t, float (&a)[kN],
                                             Image &im) const {
    float at[kA];
#pragma unroll
    for (int k = 0; k < kA; ++k) at[k] = fmaf(p.b.x, t.e0[k], fmaf(p.b.y, t.e1[k], t.c2[k]));
    const bool shaded = (at[6] >= 0.0f) || (at[7] >= 0.0f) || (at[8] >= 0.0f);   // render.py:215
    const float g[3] = {shaded ? p.g.x : 0.f, shaded ? p.g.y : 0.f, shaded ? p.g.z : 0.f};
    const SpecScene<L> &sc = im.sc;
    PixelFrame f;
    pixel_frame(at, sc.cam, f);
    float dN[3] = {0.f, 0.f, 0.f}, dP[3] = {0.f, 0.f, 0.f}, nd_d = 0.f, dshin = 0.f;
    float dKd[3] = {g[0] * sc.amb[0], g[1] * sc.amb[1], g[2] * sc.amb[2]};
    float dKs[3] = {0.f, 0.f, 0.f};
    float y1[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int l = 0; l < L; ++l) {
      LightTerm lt;
      light_term(at, f, sc.lp[l], lt);
      SpecTerm st{0.f, 0.f, 0.f};
      if (shaded) specularity(lt.rdc, sc.inv_norm[l], lt.ndl, PV ? at[kA - 1] : sc.shin, st);
      float t_l = 0.f, dspec = 0.f;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        dKd[c] += g[c] * lt.ndl * sc.li[l][c];
        dKs[c] += g[c] * st.spec * sc.li[l][c];
        t_l += g[c] * at[6 + c] * sc.li[l][c];
        dspec += g[c] * at[9 + c] * sc.li[l][c];
      }
      dshin += dspec * st.dspec_dshin;
      const float aprime = dspec * st.dspec_drn;   // dL / d rn
      im.gpart[l] += aprime * lt.rdc;
      const bool pass = lt.pre >= 0.0f && lt.pre <= 1.0f;   // torch.clamp passes the gradient inclusively
      // the diffuse term's part (d rdc = 0): d ndl = t_l
      if (pass) {
        const float k = t_l * lt.inv_vn;
        nd_d += t_l * lt.pre;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          dN[c] += t_l * lt.D[c];
          dP[c] -= lt.vn > kNormEps ? k * (f.N[c] - lt.D[c] * lt.pre) : k * f.N[c];
        }
      }
      // J: the same chain for d rdc = 1 (SpecGradFn::shade_backward with dM = Cd, dCd = M, whose projections' dot
      // products are both M . Cd = rdc)
      float dJ[6];
      {
        float dm[3], n_dot_dm = 0.f;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          dm[k] = (lt.mn > kNormEps ? (f.Cd[k] - lt.M[k] * lt.rdc) : f.Cd[k]) * lt.inv_mn;
          n_dot_dm += f.N[k] * dm[k];
        }
        float dNj[3], dD[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          dNj[k] = 2.0f * lt.ndl * dm[k];
          dD[k] = -dm[k];
        }
        if (pass) {
          const float d_ndl = 2.0f * n_dot_dm;
#pragma unroll
          for (int k = 0; k < 3; ++k) {
            dNj[k] += d_ndl * lt.D[k];
            dD[k] += d_ndl * f.N[k];
          }
        }
        float dd = 0.f, nd = 0.f;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          dd += lt.D[k] * dD[k];
          nd += f.N[k] * dNj[k];
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          const float dv = (lt.vn > kNormEps ? (dD[k] - lt.D[k] * dd) : dD[k]) * lt.inv_vn;
          const float dc = (f.cn > kNormEps ? (lt.M[k] - f.Cd[k] * lt.rdc) : lt.M[k]) * f.inv_cn;
          dJ[k] = (f.nn > kNormEps ? (dNj[k] - f.N[k] * nd) : dNj[k]) * f.inv_nn;
          dJ[3 + k] = -(dv + dc);
        }
      }
      float yj[3];
      pulled(dJ, p, t, yj);
      const float a1 = aprime * sc.inv_norm[l];
      const float b[3] = {p.b.x, p.b.y, p.b.z};
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        y1[c] = fmaf(a1, yj[c], y1[c]);
        const float y2 = lt.rdc * yj[c];
#pragma unroll
        for (int k = 0; k < 3; ++k) a[9 + 9 * l + k * 3 + c] = fmaf(b[k], y2, a[9 + 9 * l + k * 3 + c]);
      }
    }
    float dat[kA];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      dat[c] = (f.nn > kNormEps ? (dN[c] - f.N[c] * nd_d) : dN[c]) * f.inv_nn;
      dat[3 + c] = dP[c];
      dat[6 + c] = dKd[c];
      dat[9 + c] = dKs[c];
    }
    if (PV) dat[kA - 1] = dshin;
    float yd[3];
    pulled(dat, p, t, yd);
    const float b[3] = {p.b.x, p.b.y, p.b.z};
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
      for (int c = 0; c < 3; ++c) a[k * 3 + c] = fmaf(b[k], yd[c] + y1[c], a[k * 3 + c]);
  }
  __device__ __forceinline__ void end_strip(int, int strip, Image &im) const {
    const int lane = lane_id();
#pragma unroll
    for (int l = 0; l < L; ++l) {
      float v = im.gpart[l];
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);   // fixed tree
      if (lane == 0) g_rows[(size_t)strip * L + l] = v;
    }
  }
};

// The coupled pass's per-vertex gather: four lanes per (image, vertex), d positions = S1 - sum_l (G_l / norm_l^3) S2_l
// over the incident triangles' rows (CSR adjacency, fixed order, every output written once).
template <int A, int L>
__global__ __launch_bounds__(kThreads) void k_spec_gather_coupled(
    const float *__restrict__ acc, const int32_t *__restrict__ offsets, const int32_t *__restrict__ entries,
    const float *__restrict__ gsum, const float *__restrict__ norms2, int B, int V, int T,
    float *__restrict__ dpositions, const float *__restrict__ scale_src, float scale_mul) {
  const long tid = (long)blockIdx.x * kThreads + threadIdx.x;
  const long gid = tid >> 2;   // (image, vertex)
  const int c = (int)(tid & 3);
  if (gid >= (long)B * V || c == 3) return;
  const int b = (int)(gid / V);
  const int v = (int)(gid - (long)b * V);
  float gc[L];
#pragma unroll
  for (int l = 0; l < L; ++l) {
    const float nrm = fast_sqrt(norms2[(size_t)b * L + l]);
    const float inv = fast_rcp(fmaxf(nrm, kNormEps));
    gc[l] = nrm > kNormEps ? gsum[(size_t)b * L + l] * inv * inv * inv : 0.0f;   // (as load_scene forms gcoef)
  }
  const float *acc_f = acc + (size_t)b * T * 48;
  float s1 = 0.f, s2[L];
#pragma unroll
  for (int l = 0; l < L; ++l) s2[l] = 0.f;
  const int e1 = offsets[v + 1];
  constexpr int kChunk = 4;
  for (int i = offsets[v]; i < e1; i += kChunk) {
    int e[kChunk];
#pragma unroll
    for (int u = 0; u < kChunk; ++u) e[u] = i + u < e1 ? entries[i + u] : -1;
#pragma unroll
    for (int u = 0; u < kChunk; ++u) {
      if (e[u] < 0) continue;
      const unsigned t = (unsigned)e[u] / 3u, k = (unsigned)e[u] - 3u * t;
      const float *row = acc_f + (size_t)t * 48u;
      s1 += row[k * A + 3u + (unsigned)c];
      s2[0] += row[3u * A + k * 3u + (unsigned)c];
      if (L > 1) s2[L > 1 ? 1 : 0] += row[k * A + (unsigned)c];
    }
  }
  float out = s1;
#pragma unroll
  for (int l = 0; l < L; ++l) out -= gc[l] * s2[l];
  // (sign-coded upstream: S1 and G were formed from the bare codes, both linear in the loss's scale)
  dpositions[gid * 3 + c] = scale_src ? out * (scale_src[0] * scale_mul) : out;
}

template <int A>
__global__ __launch_bounds__(kThreads) void k_spec_scatter(
    const float *__restrict__ acc, const int32_t *__restrict__ tris, int B, int V, int T,
    float *__restrict__ dnormals, float *__restrict__ dpositions, float *__restrict__ ddiffuse,
    float *__restrict__ dspecular, float *__restrict__ dshininess, float *__restrict__ dclip) {
  const long gid = (long)blockIdx.x * kThreads + threadIdx.x;
  if (gid >= (long)B * T) return;
  const float4 *row = (const float4 *)(acc + gid * 48);  // 192-byte rows, 16-byte aligned
  float a[48];
  bool any = false;
#pragma unroll
  for (int q = 0; q < 12; ++q) {
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
      atomicAdd(&dnormals[v3 + c], a[k * A + c]);
      atomicAdd(&dpositions[v3 + c], a[k * A + 3 + c]);
      atomicAdd(&ddiffuse[v3 + c], a[k * A + 6 + c]);
      atomicAdd(&dspecular[v3 + c], a[k * A + 9 + c]);
    }
    if (A == 13) atomicAdd(&dshininess[(size_t)b * V + vi], a[k * A + 12]);
    float *dc = dclip + ((size_t)b * V + vi) * 4;
    atomicAdd(&dc[0], a[3 * A + k * 3 + 0]);
    atomicAdd(&dc[1], a[3 * A + k * 3 + 1]);
    atomicAdd(&dc[3], a[3 * A + k * 3 + 2]);
  }
}

// Vertex-centric alternative (round 3; the diffuse path has had it since round 2): sixteen lanes per
// (image, vertex), one per output float -- A attribute gradients, then the three clip gradients -- sum the
// rows of the incident triangles over the CSR adjacency (entry = 3 * triangle + corner).  No atomics,
// every output written exactly once, fixed order; DET: the rows hold 64-bit fixed point (run_accum.h).
template <int A, bool DET>
__global__ __launch_bounds__(kThreads) void k_spec_gather(
    const float *__restrict__ acc, const float *__restrict__ det_scale, const int32_t *__restrict__ offsets,
    const int32_t *__restrict__ entries, int B, int V, int T, float *__restrict__ dnormals,
    float *__restrict__ dpositions, float *__restrict__ ddiffuse, float *__restrict__ dspecular,
    float *__restrict__ dshininess, float *__restrict__ dclip) {
  static_assert(A + 3 <= 16, "one lane per output float");
  const long tid = (long)blockIdx.x * kThreads + threadIdx.x;
  const long gid = tid >> 4;   // (image, vertex)
  const int j = (int)(tid & 15);
  if (gid >= (long)B * V || j > A + 2) return;
  const int b = (int)(gid / V);
  const int v = (int)(gid - (long)b * V);
  float sum = 0.f;
  {
    const unsigned col0 = j < A ? (unsigned)j : 3u * A + (unsigned)(j - A), colk = j < A ? (unsigned)A : 3u;
    const float *acc_f = acc + (size_t)b * T * 48;
    const long long *acc_x = (const long long *)acc + (size_t)b * T * 48;
    const int e1 = offsets[v + 1];
    constexpr int kChunk = 8;
    for (int i = offsets[v]; i < e1; i += kChunk) {
      int e[kChunk];
      float val[kChunk];
#pragma unroll
      for (int u = 0; u < kChunk; ++u) e[u] = i + u < e1 ? entries[i + u] : -1;
#pragma unroll
      for (int u = 0; u < kChunk; ++u) {
        const unsigned t = (unsigned)e[u] / 3u, k = (unsigned)e[u] - 3u * t;
        const unsigned at = t * 48u + col0 + k * colk;
        val[u] = e[u] < 0 ? 0.f : DET ? (float)acc_x[at] * det_scale[1] : acc_f[at];
      }
#pragma unroll
      for (int u = 0; u < kChunk; ++u) sum += val[u];
    }
  }
  if (DET && *det_overflow_flag(det_scale)) sum = __int_as_float(0x7fc00000);  // see atomic_add_fixed
  float *out = j < 3 ? dnormals + gid * 3 + j
             : j < 6 ? dpositions + gid * 3 + (j - 3)
             : j < 9 ? ddiffuse + gid * 3 + (j - 6)
             : j < 12 ? dspecular + gid * 3 + (j - 9)
             : j < A ? dshininess + gid
             : j < A + 2 ? dclip + gid * 4 + (j - A)
             : dclip + gid * 4 + 3;
  *out = sum;
  if (j == A + 2) dclip[gid * 4 + 2] = 0.0f;  // the clip z column receives no gradient (with 13 attributes all 16 lanes are taken)
}

inline size_t spec_corner_bytes(int B, int T) {
  return align_up((size_t)B * T * sizeof(SpecCornerRec<kAttrMax>), 256);
}
// 8 bytes per element: room for the deterministic mode's fixed-point accumulators
inline size_t spec_acc_bytes(int B, int T) { return align_up((size_t)B * T * 48 * sizeof(long long), 256); }
inline size_t spec_sums_bytes(int B) { return align_up((size_t)B * 4 * sizeof(float), 256); }
// k_spec_pixels' per-workgroup partial sums (kNorms, kGsum): [workgroups][L <= 4]
inline int spec_blocks_per_image(int W, int H) {
  return ((W + kThreads - 1) / kThreads) * ((H + kSpecRows - 1) / kSpecRows);
}
inline size_t spec_partials_bytes(int B, int W, int H) {
  return align_up((size_t)B * spec_blocks_per_image(W, H) * 4 * sizeof(float), 256);
}

template <bool PV>
int launch_spec_corner_setup(const float *normals, const float *positions, const float *diffuse,
                             const float *specular, const float *shininess_v, const int32_t *tris, int B, int V,
                             int T, void *out, hipStream_t s) {
  constexpr int A = attr_count(PV);
  const long nbt = (long)B * T;
  hipLaunchKernelGGL((k_spec_corner_setup<A>), dim3((unsigned)((nbt + kThreads - 1) / kThreads)), dim3(kThreads),
                     0, s, (const F3 *)normals, (const F3 *)positions, (const F3 *)diffuse, (const F3 *)specular,
                     shininess_v, tris, B, V, T, (SpecCornerRec<A> *)out);
  return check_launch();
}

// sums: [B,L] out (kNorms, kGsum), through `partials` (spec_partials_bytes() of scratch)
template <int PASS, bool PV>
int launch_spec_pixels(int L, const int32_t *ids, const float *bary, const void *corners,
                       const SpecSceneIn &scene, int B, int T, int W, int H, const float *drgba, float *rgba,
                       float *sums, float *partials, hipStream_t s) {
  const int x_blocks = (W + kThreads - 1) / kThreads;
  const int y_blocks = (H + kSpecRows - 1) / kSpecRows;
  const dim3 grid((unsigned)((size_t)x_blocks * y_blocks * B)), block(kThreads);
#define MR_SPEC_PIXELS(NL)                                                                        \
  hipLaunchKernelGGL((k_spec_pixels<NL, PASS, PV>), grid, block, 0, s, ids, (const F3 *)bary,     \
                     (const SpecCornerRec<attr_count(PV)> *)corners, scene, T, W, H, x_blocks,    \
                     y_blocks, (const float4 *)drgba, (float4 *)rgba, partials)
  switch (L) {
    case 1: MR_SPEC_PIXELS(1); break;
    case 2: MR_SPEC_PIXELS(2); break;
    case 3: MR_SPEC_PIXELS(3); break;
    case 4: MR_SPEC_PIXELS(4); break;
    default: return MR_EINVAL;
  }
#undef MR_SPEC_PIXELS
  const int rc = check_launch();
  if (rc != MR_OK || PASS == kShade) return rc;
  hipLaunchKernelGGL(k_spec_sum, dim3((unsigned)B), dim3(kThreads), 0, s, partials, spec_blocks_per_image(W, H), L,
                     sums);
  return check_launch();
}

template <bool PV>
int spec_forward(const int32_t *ids, const float *bary, const float *normals, const float *positions,
                 const float *diffuse, const float *specular, const int32_t *tris, const float *light_pos,
                 const float *light_col, const float *ambient, const float *camera, const float *shininess,
                 int B, int V, int T, int W, int H, int L, float *rgba, float *norms2, int norms2_given, void *ws,
                 hipStream_t s) {
  int rc = launch_spec_corner_setup<PV>(normals, positions, diffuse, specular, PV ? shininess : nullptr, tris, B, V,
                                        T, ws, s);
  if (rc != MR_OK) return rc;
  float *partials = (float *)((char *)ws + spec_corner_bytes(B, T));
  SpecSceneIn scene{light_pos, light_col, ambient, camera, PV ? nullptr : shininess, nullptr, nullptr};
  if (!norms2_given) {   // (given: the rasterizer formed them in its own pass, mr_rasterize_specular_norms_forward)
    rc = launch_spec_pixels<kNorms, PV>(L, ids, bary, ws, scene, B, T, W, H, nullptr, nullptr, norms2, partials, s);
    if (rc != MR_OK) return rc;
  }
  scene.norms2 = norms2;
  return launch_spec_pixels<kShade, PV>(L, ids, bary, ws, scene, B, T, W, H, nullptr, rgba, nullptr, nullptr, s);
}

// one row of light / camera / shininess sums per strip of the backward's pixel pass
inline size_t spec_light_rows_bytes(int B, int W, int H) {
  // (also the coupled lane pass's G rows: two floats per 64-column strip of at least four rows)
  const size_t rows_kernel = (size_t)B * strips_per_image<SpecGradFn<1, false>>(W, H) * kSumRowSlots;
  const size_t lanes_g = (size_t)B * ((W + 63) / 64) * ((H + 3) / 4) * 2;
  return align_up((rows_kernel > lanes_g ? rows_kernel : lanes_g) * sizeof(float), 256);
}

template <bool PV>
int spec_backward(const float *drgba, const uint8_t *signs, const float *sign_upstream, const int32_t *ids, const float *bary, const float *clip,
                  const float *normals, const float *positions, const float *diffuse, const float *specular,
                  const int32_t *tris, const float *light_pos, const float *light_col, const float *ambient,
                  const float *camera, const float *shininess, const float *norms2, int B, int V, int T, int W,
                  int H, int L, float *dclip, float *dnormals, float *dpositions, float *ddiffuse,
                  float *dspecular, float *dshininess, float *light_grads, const int32_t *vertex_offsets,
                  const int32_t *vertex_entries, const float *transforms, int gbuffer_flags, int grads_wanted,
                  void *ws, hipStream_t s) {
  constexpr int A = attr_count(PV);
  const bool det = g_deterministic != 0;
  // the lane kernel (SpecFoldLaneFn): vertex gradients only, a normalised G-buffer, float atomics
  const bool lanes = MR_SPEC_LANES && !det && (gbuffer_flags & MR_GBUFFER_NORMALISED) != 0 &&
                     (grads_wanted & ~(MR_GRAD_POSITIONS | MR_GRAD_CLIP)) == 0;
  const bool fold = lanes && transforms && (grads_wanted & MR_GRAD_CLIP) == 0;
  // one pass instead of G pass + pixel pass (SpecCoupledLaneFn): one or two lights, folded, per-vertex gather
  const bool coupled = MR_SPEC_COUPLED && fold && L <= 2 && vertex_offsets && vertex_entries;
  if (det && !(vertex_offsets && vertex_entries)) return MR_EINVAL;  // the scatter path is float atomics only
  const size_t n_image = (size_t)B * H * W * 4;
  const float sign_inv_n = 1.0f / (float)n_image;
  if (signs && !coupled) {  // the other pixel kernels take the dense image: the loss's backward, into the workspace's tail
    float *dense = (float *)((char *)ws + align_up(shade_specular_backward_ws(B, V, T, W, H), 256));
    const int rc_l1 = launch_l1_backward(signs, n_image, sign_upstream, dense, s);
    if (rc_l1 != MR_OK) return rc_l1;
    drgba = dense;
    signs = nullptr;
  }
  char *p = (char *)ws;
  float *acc = (float *)p;
  p += spec_acc_bytes(B, T);
  BwdRec *recs = (BwdRec *)p;
  p += align_up((size_t)B * T * sizeof(BwdRec), 256);
  void *corners = p;
  p += spec_corner_bytes(B, T);
  float *gsum = (float *)p;
  p += spec_sums_bytes(B);
  float *partials = (float *)p;
  p += spec_partials_bytes(B, W, H);
  float *light_rows = (float *)p;
  p += spec_light_rows_bytes(B, W, H);
  float *det_block = (float *)p;
  p += kDetBlockBytes;
  SpecFoldRec<A> *fold_recs = (SpecFoldRec<A> *)p;
  if (zero_async(acc, (size_t)B * T * 48 * (det ? sizeof(long long) : sizeof(float)), s) != hipSuccess)
    return check_launch();
  int rc = MR_OK;
  if (det && (rc = launch_det_scale(drgba, (size_t)B * H * W * 4, 1.0f, det_block, s)) != MR_OK) return rc;
  if (!lanes) {
    rc = launch_bwd_setup(clip, tris, B, V, T, recs, s);
    if (rc != MR_OK) return rc;
  }
  rc = launch_spec_corner_setup<PV>(normals, positions, diffuse, specular, PV ? shininess : nullptr, tris, B, V, T,
                                    corners, s);
  if (rc != MR_OK) return rc;
  if (lanes) {
    const long nbt = (long)B * T;
    hipLaunchKernelGGL((k_spec_fold_setup<A>), dim3((unsigned)((nbt + kThreads - 1) / kThreads)), dim3(kThreads), 0, s,
                       (const float4 *)clip, tris, (const SpecCornerRec<A> *)corners, B, V, T, fold_recs,
                       coupled ? transforms : nullptr);
    if ((rc = check_launch()) != MR_OK) return rc;
  }
  SpecSceneIn scene{light_pos, light_col, ambient, camera, PV ? nullptr : shininess, norms2, nullptr};
  if (coupled) {
#define MR_SPEC_BWD_COUPLED(NL, SG)                                                                                  \
    {                                                                                                                \
      SpecCoupledLaneFn<NL, PV, SG> fn{{(const float4 *)drgba, ids, (const F3 *)bary, nullptr, nullptr, scene, nullptr, T, W, H, \
                                        signs}, fold_recs, light_rows};                                              \
      rc = launch_accumulate_lanes(fn, B, T, W, H, acc, s);                                                          \
      if (rc == MR_OK)                                                                                               \
        rc = launch_sum_strip_rows(light_rows, B, lanes_strips_per_image<SpecCoupledLaneFn<NL, PV, SG>>(B, W, H), NL, gsum, s); \
      if (rc == MR_OK) {                                                                                             \
        const long nbv4 = (long)B * V * 4;                                                                           \
        hipLaunchKernelGGL((k_spec_gather_coupled<A, NL>), dim3((unsigned)((nbv4 + kThreads - 1) / kThreads)), dim3(kThreads), 0, s, \
                           acc, vertex_offsets, vertex_entries, gsum, norms2, B, V, T, dpositions,                   \
                           SG ? sign_upstream : nullptr, sign_inv_n);                                                \
        rc = check_launch();                                                                                         \
      }                                                                                                              \
    }
    if (signs) {
      if (L == 1) MR_SPEC_BWD_COUPLED(1, true) else MR_SPEC_BWD_COUPLED(2, true)
    } else {
      if (L == 1) MR_SPEC_BWD_COUPLED(1, false) else MR_SPEC_BWD_COUPLED(2, false)
    }
#undef MR_SPEC_BWD_COUPLED
    return rc;
  }
  rc = launch_spec_pixels<kGsum, PV>(L, ids, bary, corners, scene, B, T, W, H, drgba, nullptr, gsum, partials, s);
  if (rc != MR_OK) return rc;
  scene.gsum = gsum;
#define MR_SPEC_BWD(NL)                                                                               \
  {                                                                                                   \
    SpecGradFn<NL, PV> fn{(const float4 *)drgba, ids, (const F3 *)bary, (const SpecCornerRec<A> *)corners, \
                          recs, scene, light_rows, T, W, H};                                          \
    rc = launch_accumulate_rows(fn, B, T, W, H, acc, s, det ? det_block : nullptr);                   \
  }
#define MR_SPEC_BWD_LANES(NL, FOLDED)                                                                    \
  {                                                                                                      \
    SpecFoldLaneFn<NL, PV, FOLDED> fn{{(const float4 *)drgba, ids, (const F3 *)bary, nullptr, nullptr, scene, \
                                       nullptr, T, W, H}, fold_recs, transforms};                        \
    rc = launch_accumulate_lanes(fn, B, T, W, H, acc, s);                                                \
  }
  if (lanes) {   // (the image-wide sums are not wanted: light_grads stays as the caller's launcher cleared it)
    switch (L * 2 + (fold ? 1 : 0)) {
      case 2: MR_SPEC_BWD_LANES(1, false); break;
      case 3: MR_SPEC_BWD_LANES(1, true); break;
      case 4: MR_SPEC_BWD_LANES(2, false); break;
      case 5: MR_SPEC_BWD_LANES(2, true); break;
      case 6: MR_SPEC_BWD_LANES(3, false); break;
      case 7: MR_SPEC_BWD_LANES(3, true); break;
      case 8: MR_SPEC_BWD_LANES(4, false); break;
      case 9: MR_SPEC_BWD_LANES(4, true); break;
      default: return MR_EINVAL;
    }
  } else {
    switch (L) {
      case 1: MR_SPEC_BWD(1); break;
      case 2: MR_SPEC_BWD(2); break;
      case 3: MR_SPEC_BWD(3); break;
      case 4: MR_SPEC_BWD(4); break;
      default: return MR_EINVAL;
    }
  }
#undef MR_SPEC_BWD
#undef MR_SPEC_BWD_LANES
  if (rc != MR_OK) return rc;
  if (!lanes) {
    rc = launch_sum_strip_rows(light_rows, B, strips_per_image<SpecGradFn<1, PV>>(W, H), L * 6 + 7, light_grads, s);
    if (rc != MR_OK) return rc;
  }
  if (vertex_offsets && vertex_entries) {
    const long nbv = (long)B * V * 16;  // sixteen lanes per vertex
    const dim3 grid((unsigned)((nbv + kThreads - 1) / kThreads));
    if (det)
      hipLaunchKernelGGL((k_spec_gather<A, true>), grid, dim3(kThreads), 0, s, acc, det_block, vertex_offsets,
                         vertex_entries, B, V, T, dnormals, dpositions, ddiffuse, dspecular, dshininess, dclip);
    else
      hipLaunchKernelGGL((k_spec_gather<A, false>), grid, dim3(kThreads), 0, s, acc, det_block, vertex_offsets,
                         vertex_entries, B, V, T, dnormals, dpositions, ddiffuse, dspecular, dshininess, dclip);
    return check_launch();
  }
  const long nbt = (long)B * T;
  hipLaunchKernelGGL((k_spec_scatter<A>), dim3((unsigned)((nbt + kThreads - 1) / kThreads)), dim3(kThreads), 0, s,
                     acc, tris, B, V, T, dnormals, dpositions, ddiffuse, dspecular, dshininess, dclip);
  return check_launch();
}

}  // namespace


size_t shade_specular_forward_ws(int B, int V, int T, int W, int H) {
  (void)V;
  return spec_corner_bytes(B, T) + spec_partials_bytes(B, W, H);
}

int launch_shade_specular_forward(const int32_t *ids, const float *bary, const float *normals,
                                  const float *positions, const float *diffuse, const float *specular,
                                  const int32_t *tris, const float *light_pos, const float *light_col,
                                  const float *ambient, const float *camera, const float *shininess,
                                  int shininess_per_vertex, int B, int V, int T, int W, int H, int L,
                                  float *rgba, float *norms2, int norms2_given, void *ws, hipStream_t s) {
  if ((size_t)B * W * H == 0) return MR_OK;
  return shininess_per_vertex
             ? spec_forward<true>(ids, bary, normals, positions, diffuse, specular, tris, light_pos, light_col,
                                  ambient, camera, shininess, B, V, T, W, H, L, rgba, norms2, norms2_given, ws, s)
             : spec_forward<false>(ids, bary, normals, positions, diffuse, specular, tris, light_pos, light_col,
                                   ambient, camera, shininess, B, V, T, W, H, L, rgba, norms2, norms2_given, ws, s);
}

size_t shade_specular_backward_ws(int B, int V, int T, int W, int H) {
  (void)V;
  return spec_acc_bytes(B, T) + align_up((size_t)B * T * sizeof(BwdRec), 256) + spec_corner_bytes(B, T) +
         spec_sums_bytes(B) + spec_partials_bytes(B, W, H) + spec_light_rows_bytes(B, W, H) + kDetBlockBytes +
         align_up((size_t)B * T * sizeof(SpecFoldRec<kAttrMax>), 256);
}

size_t shade_specular_backward_l1_ws(int B, int V, int T, int W, int H) {
  // + the dense upstream image for the pixel kernels that do not read sign codes
  return align_up(shade_specular_backward_ws(B, V, T, W, H), 256) + align_up((size_t)B * H * W * 4 * sizeof(float), 256);
}

int launch_shade_specular_backward(const float *drgba, const uint8_t *signs, const float *sign_upstream,
                                   const int32_t *ids, const float *bary, const float *clip, const float *normals, const float *positions,
                                   const float *diffuse, const float *specular, const int32_t *tris,
                                   const float *light_pos, const float *light_col, const float *ambient,
                                   const float *camera, const float *shininess, int shininess_per_vertex,
                                   const float *norms2, int B, int V, int T, int W, int H, int L, float *dclip,
                                   float *dnormals, float *dpositions, float *ddiffuse, float *dspecular,
                                   float *dshininess, float *light_grads, const int32_t *vertex_offsets,
                                   const int32_t *vertex_entries, const float *transforms, int gbuffer_flags,
                                   int grads_wanted, void *ws, hipStream_t s) {
  if (B == 0) return MR_OK;
  const size_t v3 = (size_t)B * V * 3 * sizeof(float);
  const bool gathered = vertex_offsets && vertex_entries && T > 0 && (size_t)W * H > 0;  // every output written once
  if (V > 0 && !gathered) {
    if (zero_async(dclip, (size_t)B * V * 4 * sizeof(float), s) != hipSuccess) return check_launch();
    if (zero_async(dnormals, v3, s) != hipSuccess) return check_launch();
    if (zero_async(dpositions, v3, s) != hipSuccess) return check_launch();
    if (zero_async(ddiffuse, v3, s) != hipSuccess) return check_launch();
    if (zero_async(dspecular, v3, s) != hipSuccess) return check_launch();
    if (shininess_per_vertex &&
        zero_async(dshininess, (size_t)B * V * sizeof(float), s) != hipSuccess)
      return check_launch();
  }
  if (zero_async(light_grads, (size_t)B * (L * 6 + 7) * sizeof(float), s) != hipSuccess)
    return check_launch();
  if (T == 0 || V == 0 || (size_t)W * H == 0) return MR_OK;
  return shininess_per_vertex
             ? spec_backward<true>(drgba, signs, sign_upstream, ids, bary, clip, normals, positions, diffuse, specular, tris, light_pos,
                                   light_col, ambient, camera, shininess, norms2, B, V, T, W, H, L, dclip,
                                   dnormals, dpositions, ddiffuse, dspecular, dshininess, light_grads,
                                   vertex_offsets, vertex_entries, transforms, gbuffer_flags, grads_wanted, ws, s)
             : spec_backward<false>(drgba, signs, sign_upstream, ids, bary, clip, normals, positions, diffuse, specular, tris, light_pos,
                                    light_col, ambient, camera, shininess, norms2, B, V, T, W, H, L, dclip,
                                    dnormals, dpositions, ddiffuse, dspecular, dshininess, light_grads,
                                    vertex_offsets, vertex_entries, transforms, gbuffer_flags, grads_wanted, ws, s);
}

}  // namespace mr
