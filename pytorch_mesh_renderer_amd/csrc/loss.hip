// Fused mean-absolute-error image loss for gfx950 (MI355X).
//
// The reference's optimisation tests and examples all drive the renderer with
// `torch.mean(torch.abs(render - target))` (src/mesh_renderer/mesh_renderer_test.py:250,
// src/examples/example5.py:70-92).  In eager torch that is five full passes over the
// [B,H,W,4] image (sub, abs, mean; sign, mul) -- 1.05 ms at 1024^2 x 32, more than the
// rasterizer.  Here: one streaming pass forward (reads 2 x 16 B/px, writes 1 B/px: the four
// signs of a pixel's channels as 2-bit codes) and one backward that never re-reads the images
// (reads 1 B/px, writes 16 B/px).  Both are HBM-bound; 49 B/px in total instead of 80.
#include "mr_internal.h"

#ifndef MR_L1_REGIONS_NT_A
#define MR_L1_REGIONS_NT_A 1   // the image stream of k_l1_forward_regions nontemporal too
#endif
#ifndef MR_L1_REVERSE
#define MR_L1_REVERSE 1
#endif
#ifndef MR_L1_BLOCKS
#define MR_L1_BLOCKS MR_L1_PARTIALS
#endif
static_assert(MR_L1_BLOCKS <= MR_L1_PARTIALS, "one partial sum per workgroup: the caller's scratch holds MR_L1_PARTIALS floats");
#ifndef MR_L1_UNROLL
#define MR_L1_UNROLL 2  // measured alone at 1024^2 x 32: 1 -> 0.216, 2 -> 0.211, 4 -> 0.248, 8 -> 0.241 ms
#endif

namespace mr {
namespace {

constexpr int kThreads = 256;

__device__ __forceinline__ float4 nt_load(const float4 *p) {
  typedef float v4f __attribute__((ext_vector_type(4)));
  const v4f v = __builtin_nontemporal_load((const v4f *)p);
  return make_float4(v.x, v.y, v.z, v.w);
}

// 2-bit sign code of d, two's complement: 0 -> 0, 1 -> +1, 3 -> -1 (NaN -> 0, like (0 < d) - (d < 0)); a
// reader gets the value with one signed bit-field extract (v_bfe_i32) and a conversion.
__device__ __forceinline__ unsigned sign_code(float d) { return d > 0.f ? 1u : (d < 0.f ? 3u : 0u); }
__device__ __forceinline__ float sign_value(unsigned code) {
  return (code & 2u) ? -1.f : ((code & 1u) ? 1.f : 0.f);
}

__global__ __launch_bounds__(kThreads) void k_l1_forward(const float4 *__restrict__ a,
                                                         const float4 *__restrict__ b, size_t n4,
                                                         const float *__restrict__ a_tail,
                                                         const float *__restrict__ b_tail, int n_tail,
                                                         float inv_n, float *__restrict__ partials,
                                                         uint8_t *__restrict__ signs) {
  float s = 0.f;
  // kUnroll pixels per trip, a grid stride apart, all 2 * kUnroll loads issued before the first use:
  // the pass is bound by bytes in flight (8 waves x 2 loads per SIMD left HBM at 4.4 TB/s).
  // Back to front (MR_L1_REVERSE): the tail of `a` is what the producer (the renderer) wrote last, so
  // part of it is still in the 256 MB Infinity Cache.
  constexpr int kUnroll = MR_L1_UNROLL;
  const size_t stride = (size_t)gridDim.x * kThreads;
  for (size_t j = (size_t)blockIdx.x * kThreads + threadIdx.x; j < n4; j += kUnroll * stride) {
    float4 x[kUnroll], y[kUnroll];
    size_t idx[kUnroll];
#pragma unroll
    for (int u = 0; u < kUnroll; ++u) {
      const size_t ju = j + u * stride;
      const size_t jc = ju < n4 ? ju : j;   // out of range: re-read the first (adds nothing below)
      idx[u] = MR_L1_REVERSE ? n4 - 1 - jc : jc;
      x[u] = nt_load(&a[idx[u]]);
      y[u] = nt_load(&b[idx[u]]);
    }
#pragma unroll
    for (int u = 0; u < kUnroll; ++u) {
      if (u > 0 && j + u * stride >= n4) break;
      const float d0 = x[u].x - y[u].x, d1 = x[u].y - y[u].y, d2 = x[u].z - y[u].z, d3 = x[u].w - y[u].w;
      s += (fabsf(d0) + fabsf(d1)) + (fabsf(d2) + fabsf(d3));
      if (signs) {
        const uint8_t code =
            (uint8_t)(sign_code(d0) | (sign_code(d1) << 2) | (sign_code(d2) << 4) | (sign_code(d3) << 6));
        __builtin_nontemporal_store(code, &signs[idx[u]]);
      }
    }
  }
  if (blockIdx.x == 0 && threadIdx.x == 0 && n_tail > 0) {  // the last n % 4 elements
    unsigned code = 0u;
    for (int k = 0; k < n_tail; ++k) {
      const float d = a_tail[k] - b_tail[k];
      s += fabsf(d);
      code |= sign_code(d) << (2 * k);
    }
    if (signs) signs[n4] = (uint8_t)code;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off);
  __shared__ float s_part[kThreads / kWave];
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
  if (lane == 0) s_part[wave] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
    for (int w = 0; w < kThreads / kWave; ++w) t += s_part[w];
    partials[blockIdx.x] = t * inv_n;  // summed in a fixed order by k_l1_finish: reproducible bits
  }
}

// (Round 4, measured and dropped: the workgroup that ARRIVES LAST adding the partial sums inside k_l1_forward --
//  write-through partials, a drained store, one agent-scope add to an arrival counter per workgroup, agent-scope
//  loads by the last one.  One launch less, but the pass's tail grows by more than the launch it saves: same-box
//  A/B 0.190 + finish -> 0.2005 ms for the kernel, step 0.7241 -> 0.7322 ms.)
// One workgroup adds the workgroups' partial sums in a fixed order (no float atomics: the loss
// value is bit-identical from run to run).  1024 threads, two independent loads each: a single
// wavefront looping over 2048 partials was 32 dependent load round trips (9 us).
constexpr int kFinishThreads = 1024;
__global__ __launch_bounds__(kFinishThreads) void k_l1_finish(const float *__restrict__ partials, int n,
                                                              float *__restrict__ out) {
  const int t = (int)threadIdx.x;
  float s = 0.f;
  for (int i = t; i < n; i += kFinishThreads) s += partials[i];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off);
  __shared__ float s_wave[kFinishThreads / kWave];
  if ((t & (kWave - 1)) == 0) s_wave[t >> 6] = s;
  __syncthreads();
  if (t < kWave) {
    float v = t < kFinishThreads / kWave ? s_wave[t] : 0.f;
#pragma unroll
    for (int off = 8; off > 0; off >>= 1) v += __shfl_down(v, off);
    if (t == 0) out[0] = v;
  }
}

__global__ __launch_bounds__(kThreads) void k_l1_backward(const uint8_t *__restrict__ signs, size_t n4,
                                                          int n_tail, const float *__restrict__ upstream,
                                                          float inv_n, float4 *__restrict__ da,
                                                          float *__restrict__ da_tail) {
  const float g = upstream[0] * inv_n;  // d loss / d mean, read on the device: no host sync
  for (size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x; i < n4; i += (size_t)gridDim.x * kThreads) {
    const unsigned c = signs[i];
    da[i] = make_float4(g * sign_value(c), g * sign_value(c >> 2), g * sign_value(c >> 4),
                        g * sign_value(c >> 6));
  }
  if (blockIdx.x == 0 && (int)threadIdx.x < n_tail)
    da_tail[threadIdx.x] = g * sign_value((unsigned)signs[n4] >> (2 * threadIdx.x));
}

// 8-bit frame export: what the reference's examples do on the host before writing PNG / GIF
// frames, `(image * 255.0).astype(np.uint8)` (src/examples/example1.py:52, example5.py:81), with the
// value clamped to [0, 1] first (numpy's cast of an out-of-range float is undefined).  One pass,
// 16 B read + 4 B written per RGBA pixel; NaN exports as 0.
__device__ __forceinline__ unsigned to_u8(float v) {
  const float c = fminf(fmaxf(v, 0.0f), 1.0f);  // fmaxf(NaN, 0) = 0
  return (unsigned)(c * 255.0f);                 // truncation, like astype(np.uint8)
}

__global__ __launch_bounds__(kThreads) void k_export_u8(const float4 *__restrict__ in, size_t n4,
                                                        const float *__restrict__ in_tail, int n_tail,
                                                        uint32_t *__restrict__ out, uint8_t *__restrict__ out_tail) {
  for (size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x; i < n4; i += (size_t)gridDim.x * kThreads) {
    const float4 v = in[i];
    out[i] = to_u8(v.x) | (to_u8(v.y) << 8) | (to_u8(v.z) << 16) | (to_u8(v.w) << 24);
  }
  if (blockIdx.x == 0 && (int)threadIdx.x < n_tail) out_tail[threadIdx.x] = (uint8_t)to_u8(in_tail[threadIdx.x]);
}

// ---- tone_mapper (src/mesh_renderer/render.py:389-419): out = clamp(image^gamma / max, 0, 1) with
// max taken per image over image^gamma.  Two streaming passes: (1) per-image maximum of the powers
// as order-preserving integer keys (tone_key below): one integer atomicMax per workgroup, which
// also reproduces torch.max's NaN propagation; (2) the
// powers are recomputed (cheaper than 4 B/element of scratch traffic), scaled with an IEEE
// division as torch does, clamped, and stored as fp32 or straight as 8-bit frames.
// FAST: gamma is finite.  For a finite positive base the power is exp2(gamma * log2(v)) on the 1-ulp
// hardware v_log_f32 / v_exp_f32 (relative error <= (0.69 |gamma log2 v| + 1) 2^-23: below 4e-7 for the
// values that survive the division by the image's maximum); zero with a positive exponent is zero;
// everything else (negative, infinite, NaN bases; zero with gamma <= 0) takes powf and its special
// cases, which is torch.pow's table.  (powf for every element made both passes instruction-bound:
// 0.78 + 0.60 ms for 32 x 1024^2 x 4 where the data moves in 0.1 + 0.22.)
template <bool FAST>
__device__ __forceinline__ float tone_power(float v, float gamma) {
  float p;
  if (FAST && v > 0.0f && v < INFINITY) p = __builtin_amdgcn_exp2f(gamma * __builtin_amdgcn_logf(v));
  else if (FAST && v == 0.0f && gamma > 0.0f) p = 0.0f;
  else p = powf(v, gamma);
  return p != p ? __int_as_float(0x7fc00000) : p;  // canonical NaN: positive as an integer
}

// The maximum is taken on integer keys that order like the floats they stand for: non-negative
// values and NaN (canonical, positive) keep their bit pattern -- NaN sorts on top, which is
// torch.max's propagation --, negative values (a negative base with an odd integer gamma: (-2)^3 =
// -8) map below all of them in their own order, -0.0 counts as +0.0.  (Round 2 masked the sign bit
// off instead: -8 became +8 and could win the maximum.)
__device__ __forceinline__ int tone_key(float p) {
  const int u = __float_as_int(p);
  return u >= 0 ? u : (u == INT_MIN ? 0 : u ^ 0x7fffffff);
}
__device__ __forceinline__ float tone_value(int key) { return __int_as_float(key >= 0 ? key : key ^ 0x7fffffff); }

// VEC: the image's element count is a multiple of four (every image then starts 16-byte aligned):
// 16-byte accesses.  One integer atomicMax per WORKGROUP, ~2048 workgroups in all: with one per
// wavefront of a 512-workgroup grid, 65k atomics queued up on the one cache line that holds all the
// images' maxima.
constexpr int kToneBlocks = 64;
template <bool FAST, bool VEC>
__global__ __launch_bounds__(kThreads) void k_tone_max(const float *__restrict__ image, size_t per_image,
                                                       float gamma, int *__restrict__ max_bits) {
  const float *img = image + (size_t)blockIdx.y * per_image;
  int best = INT_MIN;  // below every key
  auto take = [&](float v) { best = max(best, tone_key(tone_power<FAST>(v, gamma))); };
  if (VEC) {
    // four independent 16-byte loads in flight per lane (one per trip kept the pass at 4.1 TB/s: with eight
    // wavefronts per SIMD that is too few bytes in flight for the HBM round trip), nontemporal: read once
    const float4 *img4 = (const float4 *)img;
    const size_t n4 = per_image / 4, stride = (size_t)gridDim.x * kThreads;
    size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x;
    for (; i + 3 * stride < n4; i += 4 * stride) {
      float4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = nt_load(&img4[i + u * stride]);
#pragma unroll
      for (int u = 0; u < 4; ++u) { take(v[u].x); take(v[u].y); take(v[u].z); take(v[u].w); }
    }
    for (; i < n4; i += stride) {
      const float4 v = img4[i];
      take(v.x); take(v.y); take(v.z); take(v.w);
    }
  } else {
    for (size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x; i < per_image; i += (size_t)gridDim.x * kThreads)
      take(img[i]);
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) best = max(best, __shfl_down(best, off));
  __shared__ int s_best[kThreads / kWave];
  if ((threadIdx.x & (kWave - 1)) == 0) s_best[threadIdx.x >> 6] = best;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < kThreads / kWave; ++w) best = max(best, s_best[w]);
    if (best != INT_MIN) atomicMax(&max_bits[blockIdx.y], best);
  }
}

template <bool U8, bool FAST, bool VEC>
__global__ __launch_bounds__(kThreads) void k_tone_map(const float *__restrict__ image, size_t per_image,
                                                       float gamma, const int *__restrict__ max_bits,
                                                       float *__restrict__ out, uint8_t *__restrict__ out_u8) {
  const size_t base = (size_t)blockIdx.y * per_image;
  const float image_max = tone_value(max_bits[blockIdx.y]);
  auto scaled = [&](float v) { return tone_power<FAST>(v, gamma) / image_max; };  // IEEE division, as torch does
  auto clamped = [](float x) { return x != x ? x : fminf(fmaxf(x, 0.0f), 1.0f); };  // torch.clamp: NaN stays NaN
  if (VEC) {
    const float4 *in4 = (const float4 *)(image + base);
    const size_t n4 = per_image / 4, stride = (size_t)gridDim.x * kThreads;
    auto emit = [&](const size_t i, const float4 v) {
      const float a = scaled(v.x), b = scaled(v.y), c = scaled(v.z), d = scaled(v.w);
      if (U8) ((uint32_t *)(out_u8 + base))[i] = to_u8(a) | (to_u8(b) << 8) | (to_u8(c) << 16) | (to_u8(d) << 24);
      else ((float4 *)(out + base))[i] = make_float4(clamped(a), clamped(b), clamped(c), clamped(d));
    };
    size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x;
    for (; i + 3 * stride < n4; i += 4 * stride) {   // four loads in flight per lane, as in k_tone_max
      float4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = nt_load(&in4[i + u * stride]);
#pragma unroll
      for (int u = 0; u < 4; ++u) emit(i + u * stride, v[u]);
    }
    for (; i < n4; i += stride) emit(i, in4[i]);
  } else {
    for (size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x; i < per_image; i += (size_t)gridDim.x * kThreads) {
      const float x = scaled(image[base + i]);
      if (U8) out_u8[base + i] = (uint8_t)to_u8(x);
      else out[base + i] = clamped(x);
    }
  }
}

// ---- round 4: the loss over an image with KNOWN empty blocks ---------------------------------------------------
// A rendered image of an object over an empty background is, region by region, exactly zero where the rasterizer
// found no candidate triangle (mr_render_forward's `empty_regions`: 64 x 64-pixel blocks of the G-buffer, i.e. image
// rows H - 1 - y); a target made the same way has its own such blocks (mr_image_empty_regions, computed once per
// target).  Where BOTH are empty |a - b| = 0 and every sign code is 0: the block is neither read (2 x 16 B/px)
// nor compared -- a quarter of the benchmark's pixels.  One 256-thread workgroup walks blocks (grid stride); a
// block's row is 64 pixels = 1 KB = one wave-instruction of 16-byte loads per image.
constexpr int kBlockEdge = 64;

__global__ __launch_bounds__(kThreads) void k_image_empty_regions(const float4 *__restrict__ image, int B, int H, int W,
                                                                 int blocks_x, int blocks_y, uint8_t *__restrict__ map) {
  __shared__ int s_any;
  const int lane_x = (int)threadIdx.x & (kBlockEdge - 1), row0 = (int)threadIdx.x / kBlockEdge;   // 64 columns x 4 rows per trip
  for (int blk = (int)blockIdx.x; blk < B * blocks_x * blocks_y; blk += (int)gridDim.x) {
    const int img = blk / (blocks_x * blocks_y), rem = blk - img * blocks_x * blocks_y;
    const int by = rem / blocks_x, bx = rem - by * blocks_x;
    if (threadIdx.x == 0) s_any = 0;
    __syncthreads();
    const bool whole = bx * kBlockEdge + kBlockEdge <= W && by * kBlockEdge + kBlockEdge <= H;
    unsigned bits = 0u;
    if (whole) {
      for (int r = row0; r < kBlockEdge; r += kThreads / kBlockEdge) {
        const int yi = H - 1 - (by * kBlockEdge + r);   // image row of G-buffer row by * 64 + r
        const float4 v = nt_load(&image[((size_t)img * H + yi) * W + bx * kBlockEdge + lane_x]);
        bits |= __float_as_uint(v.x) | __float_as_uint(v.y) | __float_as_uint(v.z) | __float_as_uint(v.w);
      }
      bits &= 0x7fffffffu;   // (-0.0 is zero; a NaN has bits set)
    }
    if (__ballot(bits != 0u) != 0ull && (threadIdx.x & (kWave - 1)) == 0) atomicOr(&s_any, 1);
    __syncthreads();
    if (threadIdx.x == 0) map[blk] = (whole && s_any == 0) ? 1 : 0;
    __syncthreads();
  }
}

// mean |a - b| over [B,H,W,4] images with their empty-block maps: the flat kernel's streaming order (whole image rows,
// back to front, consecutive workgroups on consecutive rows), and a 64-pixel segment of a row whose block is empty on
// both sides is not loaded (its sign codes are zeros).  A first version walked block by block (1 KB row segments
// 16 KB apart): it skipped a quarter of the benchmark's pixels and was no faster than the flat pass (176 vs 182 us).
__global__ __launch_bounds__(kThreads) void k_l1_forward_regions(
    const float4 *__restrict__ a, const float4 *__restrict__ b, int B, int H, int W, int blocks_x, int blocks_y,
    const uint8_t *__restrict__ empty_a, const uint8_t *__restrict__ empty_b, float inv_n, float *__restrict__ partials,
    uint8_t *__restrict__ signs) {
  constexpr int kInFlight = MR_L1_UNROLL;
  float s = 0.f;
  const long n_rows = (long)B * H;
  const int lane_in_wave = (int)threadIdx.x & (kWave - 1);
  // (Also measured at the end of round 4, both slower in the benchmark's step: the flat kernel's own walk with one
  //  test per wavefront and load -- equal to the flat kernel with nothing to skip, but skipping scattered 1 KB
  //  segments saved only a third of their share: 0.1852 -> 0.1874 ms in the step -- and four rows a grid stride
  //  apart per workgroup, +8 %.)
  // A row's skip flags: lane i of every wavefront holds "block i of the row's band is empty on both sides"; their
  // ballot is the row's 64-bit skip mask (rows of more than 64 blocks -- W > 4096 -- skip nothing).  Requested ONE
  // ROW AHEAD: read per element in front of the image loads (the first version) the map bytes were a dependent
  // memory round trip per row, and the pass with nothing to skip ran 4.5 % behind the flat kernel.
  auto request = [&](const long k, unsigned &flag) {
    flag = 0u;
    if (k < n_rows && lane_in_wave < blocks_x && blocks_x <= kWave) {
      const long row = MR_L1_REVERSE ? n_rows - 1 - k : k;
      const int img = (int)(row / H), yi = (int)(row - (long)img * H);
      const size_t at = ((size_t)img * blocks_y + (H - 1 - yi) / kBlockEdge) * blocks_x + lane_in_wave;   // (rows counted from the image's LAST one)
      flag = (unsigned)empty_a[at] & (unsigned)empty_b[at];
    }
  };
  unsigned flag;
  request((long)blockIdx.x, flag);
  for (long k = (long)blockIdx.x; k < n_rows; k += (long)gridDim.x) {
    const long row = MR_L1_REVERSE ? n_rows - 1 - k : k;   // workgroup-uniform
    const unsigned long long skip = __ballot(flag != 0u);
    request(k + (long)gridDim.x, flag);
    const size_t base = (size_t)row * W;
    for (int x0 = (int)threadIdx.x; x0 < W; x0 += kThreads * kInFlight) {
      float4 va[kInFlight], vb[kInFlight];
#pragma unroll
      for (int u = 0; u < kInFlight; ++u) {
        const int x = x0 + u * kThreads;
        va[u] = vb[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        // (a wavefront's 64 pixels lie in one block: x0 is the thread index plus a multiple of 256)
        const int blk = __builtin_amdgcn_readfirstlane(x) / kBlockEdge;
        const bool live = blk >= kWave || ((skip >> blk) & 1ull) == 0ull;   // wave-uniform
        if (live && x < W) {
          va[u] = MR_L1_REGIONS_NT_A ? nt_load(&a[base + x]) : a[base + x];
          vb[u] = nt_load(&b[base + x]);
        }
      }
#pragma unroll
      for (int u = 0; u < kInFlight; ++u) {
        const int x = x0 + u * kThreads;
        if (x >= W) break;
        const float d0 = va[u].x - vb[u].x, d1 = va[u].y - vb[u].y, d2 = va[u].z - vb[u].z, d3 = va[u].w - vb[u].w;
        s += (fabsf(d0) + fabsf(d1)) + (fabsf(d2) + fabsf(d3));
        if (signs) {
          const uint8_t code =
              (uint8_t)(sign_code(d0) | (sign_code(d1) << 2) | (sign_code(d2) << 4) | (sign_code(d3) << 6));
          __builtin_nontemporal_store(code, &signs[base + x]);
        }
      }
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off);
  __shared__ float s_part[kThreads / kWave];
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
  if (lane == 0) s_part[wave] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
    for (int w = 0; w < kThreads / kWave; ++w) t += s_part[w];
    partials[blockIdx.x] = t * inv_n;  // summed in a fixed order by k_l1_finish: reproducible bits
  }
}

inline unsigned blocks_for(size_t n4) {
  const size_t want = (n4 + kThreads - 1) / kThreads;
  return (unsigned)(want < MR_L1_BLOCKS ? (want ? want : 1) : MR_L1_BLOCKS);
}

}  // namespace

int launch_l1_forward(const float *a, const float *b, size_t n, float *out, uint8_t *signs, float *partials,
                      hipStream_t s) {
  if (n == 0) {
    if (zero_async(out, sizeof(float), s) != hipSuccess) return check_launch();
    return MR_OK;
  }
  const size_t n4 = n / 4;
  const unsigned blocks = blocks_for(n4);
  {
    KernelTimer timer(MR_TIMER_L1_FORWARD, s);  // records only when a caller armed it
    hipLaunchKernelGGL(k_l1_forward, dim3(blocks), dim3(kThreads), 0, s, (const float4 *)a,
                       (const float4 *)b, n4, a + 4 * n4, b + 4 * n4, (int)(n - 4 * n4), 1.0f / (float)n, partials,
                       signs);
  }
  int rc = check_launch();
  if (rc != MR_OK) return rc;
  hipLaunchKernelGGL(k_l1_finish, dim3(1), dim3(kFinishThreads), 0, s, partials, (int)blocks, out);
  return check_launch();
}

int launch_image_empty_regions(const float *image, int B, int H, int W, uint8_t *map, hipStream_t s) {
  const int bx = (W + kBlockEdge - 1) / kBlockEdge, by = (H + kBlockEdge - 1) / kBlockEdge;
  const long n = (long)B * bx * by;
  if (n == 0) return MR_OK;
  hipLaunchKernelGGL(k_image_empty_regions, dim3((unsigned)(n < 4096 ? n : 4096)), dim3(kThreads), 0, s, (const float4 *)image, B,
                     H, W, bx, by, map);
  return check_launch();
}

// mean |a - b| over [B,H,W,4] images whose empty 64 x 64 blocks are known (see k_l1_forward_regions); same outputs as
// launch_l1_forward (the sum's grouping differs: whole rows per workgroup instead of a flat stride -- still a fixed order).
int launch_l1_forward_regions(const float *a, const float *b, int B, int H, int W, const uint8_t *empty_a,
                              const uint8_t *empty_b, float *out, uint8_t *signs, float *partials, hipStream_t s) {
  const size_t n = (size_t)B * H * W * 4;
  if (n == 0) {
    if (zero_async(out, sizeof(float), s) != hipSuccess) return check_launch();
    return MR_OK;
  }
  const int bx = (W + kBlockEdge - 1) / kBlockEdge, by = (H + kBlockEdge - 1) / kBlockEdge;
  if (W < kThreads)   // a row is less than one trip of the workgroup: the flat pass (reads everything) serves it better
    return launch_l1_forward(a, b, n, out, signs, partials, s);
  const long n_rows = (long)B * H;
  const unsigned blocks = (unsigned)(n_rows < MR_L1_BLOCKS ? n_rows : MR_L1_BLOCKS);
  {
    KernelTimer timer(MR_TIMER_L1_FORWARD, s);
    hipLaunchKernelGGL(k_l1_forward_regions, dim3(blocks), dim3(kThreads), 0, s, (const float4 *)a, (const float4 *)b, B, H, W,
                       bx, by, empty_a, empty_b, 1.0f / (float)n, partials, signs);
  }
  int rc = check_launch();
  if (rc != MR_OK) return rc;
  hipLaunchKernelGGL(k_l1_finish, dim3(1), dim3(kFinishThreads), 0, s, partials, (int)blocks, out);
  return check_launch();
}

int launch_l1_backward(const uint8_t *signs, size_t n, const float *upstream, float *da, hipStream_t s) {
  if (n == 0) return MR_OK;
  const size_t n4 = n / 4;
  hipLaunchKernelGGL(k_l1_backward, dim3(blocks_for(n4)), dim3(kThreads), 0, s, signs, n4,
                     (int)(n - 4 * n4), upstream, 1.0f / (float)n, (float4 *)da, da + 4 * n4);
  return check_launch();
}

int launch_tone_map(const float *image, int B, size_t per_image, float gamma, int *max_bits, float *out,
                    uint8_t *out_u8, hipStream_t s) {
  if (B == 0 || per_image == 0) return MR_OK;
  if (hipMemsetD32Async((hipDeviceptr_t)max_bits, INT_MIN, (size_t)B, s) != hipSuccess) return check_launch();  // below every key
  const bool vec = per_image % 4 == 0 && ((uintptr_t)image & 15u) == 0 && ((uintptr_t)out & 15u) == 0 &&
                   ((uintptr_t)out_u8 & 3u) == 0;
  const bool fast = gamma == gamma && gamma - gamma == 0.0f;  // finite
  const size_t want = ((vec ? per_image / 4 : per_image) + kThreads - 1) / kThreads;
  // ~2048 workgroups in all (a full chip of 256-thread workgroups), at least 8 and at most 512 per image
  const size_t per = B >= 256 ? 8 : (size_t)(kToneBlocks * 32 / B < 512 ? kToneBlocks * 32 / B : 512);
  const dim3 grid((unsigned)(want < per ? want : per), (unsigned)B);
#define MR_TONE(FAST, VEC)                                                                                 \
  {                                                                                                        \
    hipLaunchKernelGGL((k_tone_max<FAST, VEC>), grid, dim3(kThreads), 0, s, image, per_image, gamma, max_bits); \
    const int rc = check_launch();                                                                         \
    if (rc != MR_OK) return rc;                                                                            \
    if (out_u8) hipLaunchKernelGGL((k_tone_map<true, FAST, VEC>), grid, dim3(kThreads), 0, s, image, per_image, gamma, \
                                   max_bits, out, out_u8);                                                 \
    else hipLaunchKernelGGL((k_tone_map<false, FAST, VEC>), grid, dim3(kThreads), 0, s, image, per_image, gamma,      \
                            max_bits, out, out_u8);                                                        \
  }
  if (fast && vec) MR_TONE(true, true)
  else if (fast) MR_TONE(true, false)
  else if (vec) MR_TONE(false, true)
  else MR_TONE(false, false)
#undef MR_TONE
  return check_launch();
}

int launch_export_u8(const float *in, size_t n, uint8_t *out, hipStream_t s) {
  if (n == 0) return MR_OK;
  const size_t n4 = n / 4;
  hipLaunchKernelGGL(k_export_u8, dim3(blocks_for(n4)), dim3(kThreads), 0, s, (const float4 *)in, n4,
                     in + 4 * n4, (int)(n - 4 * n4), (uint32_t *)out, out + 4 * n4);
  return check_launch();
}

}  // namespace mr
