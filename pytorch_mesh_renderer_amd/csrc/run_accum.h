// Per-triangle scatter-add of per-pixel quantities, the gfx950 way.
//
// Every backward pass of the path ends in "for each covered pixel, add K numbers to
// the triangle that owns the pixel" (reference: the nine '+=' of
// rasterize_triangles.cpp:232-269, and the index_put_(accumulate=True) that
// autograd derives from the gather in src/mesh_renderer/rasterize.py:130-132).
// Doing that with one global atomic per number per pixel is ~20 G atomics/s on
// MI355X -- tens of milliseconds at 1024^2 x 32.  Instead:
//
//   1. each lane walks DOWN one pixel column (wave loads stay 64 consecutive
//      pixels = coalesced) and keeps the K running sums in registers for as long as
//      the triangle id does not change (runs are tens of pixels long);
//   2. a finished run goes into a workgroup-local LDS hash table keyed by triangle
//      id with ds_add_f32;
//   3. the table is drained once per workgroup with contiguous global float atomics
//      into acc[image][triangle][kStride].
//
// The functor supplies the per-pixel work:
//   struct Fn {
//     static constexpr int kN      = ...;   // sums per triangle
//     static constexpr int kStride = ...;   // floats per acc row (>= kN)
//     static constexpr int kSlots  = ...;   // LDS hash slots, power of two
//     static constexpr int kMinWavesPerSimd = ...;  // occupancy the register allocator must allow
//     struct Raw {...}; struct Pixel {...}; struct Triangle {...}; struct Image {...};
//     __device__ void begin_image(int img, Image &) const;
//     __device__ void fetch(int img, int x, int y, size_t pix, Raw &) const;   // loads only
//     __device__ bool prepare(const Raw &, int T, int &tri, Pixel &) const;    // false: skip
//     __device__ void load_triangle(int img, int tri, Triangle &) const;   // on run change
//     __device__ void accumulate(const Pixel &, const Triangle &, float (&a)[kN], Image &) const;
//     __device__ void end_image(int img, Image &) const;   // per-lane image-wide sums (k_accumulate_runs)
//     __device__ void end_strip(int img, int strip, Image &) const;   // the same for k_accumulate_rows
//   };
// fetch() must not branch on loaded data: the kernel issues the NEXT pixel's fetch before
// it evaluates the current one, so two pixels' loads are in flight per lane.
// Per-triangle data (e.g. the adjugate) is fetched once per run, not per pixel.
#pragma once

#include <type_traits>

#include "mr_internal.h"

namespace mr {

constexpr int kRunThreads = 256;
#ifndef MR_RUN_ROWS_PER_WAVE
#define MR_RUN_ROWS_PER_WAVE 32
#endif
constexpr int kRunRowsPerWave = MR_RUN_ROWS_PER_WAVE;                 // pixels each lane walks (k_accumulate_runs)
constexpr int kRunRegionH = kRunRowsPerWave * (kRunThreads / kWave);  // 128 rows / workgroup
constexpr int kRunMaxProbe = 16;

// Deterministic mode (mr_set_deterministic): sums are accumulated in 64-bit FIXED POINT with
// integer atomics.  Integer addition is associative (also through two's-complement wrap-around),
// so the result no longer depends on the order in which lanes and wavefronts commit: bit-identical
// from run to run, where float atomics differ in the last bits.  det_scale[0] = 2^k converts to
// fixed point (a power of two: exact), det_scale[1] = 2^-k back; k is derived on the device from the
// largest upstream gradient g so that g maps to about 2^41: values down to g * 2^-42 are resolved and
// a triangle's total may reach g * 2^21 before the 64-bit range ends.
// A contribution that does not fit -- NaN, infinite, or beyond +-2^63 after scaling (1 / det of a
// sliver triangle times a large upstream gradient) -- is not converted (the conversion of an
// out-of-range float is garbage of arbitrary sign): it raises the launch's overflow flag instead,
// det_overflow_flag(det_scale), and the pass that converts the sums back to float writes NaN
// everywhere when the flag is set -- the float path's answer to such inputs, spread over the whole
// output, instead of a finite wrong number.  (Sums of many in-range contributions still wrap
// silently beyond 2^63: the scale leaves 2^21 of headroom over the largest upstream gradient.)
__device__ __forceinline__ int *det_overflow_flag(const float *det_scale) { return (int *)det_scale + 8; }
__device__ __forceinline__ void atomic_add_fixed(long long *p, float v, float to_fixed, int *overflow) {
  const float x = v * to_fixed;
  if (!(fabsf(x) < 9.0e18f)) {
    atomicOr(overflow, 1);
    return;
  }
  atomicAdd((unsigned long long *)p, (unsigned long long)__float2ll_rn(x));
}

// The deterministic mode's 512-byte side block (zeroed per launch): float [0] = 2^k, [1] = 2^-k,
// int [4] = bits of the largest |upstream gradient|, int [8] = overflow flag (above).
constexpr size_t kDetBlockBytes = 512;
static __global__ __launch_bounds__(256) void k_det_abs_max(const float *__restrict__ x, size_t n, int *__restrict__ max_bits) {
  int best = 0;  // non-negative floats order like integers; a NaN sorts on top
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
    best = max(best, __float_as_int(fabsf(x[i])));
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) best = max(best, __shfl_down(best, off));
  __shared__ int s_best[4];  // one atomic per WORKGROUP (thousands on one address queue up)
  if ((threadIdx.x & 63) == 0) s_best[threadIdx.x >> 6] = best;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 4; ++w) best = max(best, s_best[w]);
    if (best != 0) atomicMax(max_bits, best);
  }
}
// (2^k, 2^-k) with k such that `gain` times the largest upstream gradient maps to about 2^41
static __global__ void k_det_scale_from_bits(const int *__restrict__ max_bits, float gain, float *__restrict__ det_scale) {
  const float g = __int_as_float(max_bits[0]) * gain;
  int e = 0;
  if (g > 0.0f && g < INFINITY) (void)frexpf(g, &e);  // g = m * 2^e, m in [0.5, 1)
  const int k = min(max(41 - e, -100), 100);
  det_scale[0] = ldexpf(1.0f, k);
  det_scale[1] = ldexpf(1.0f, -k);
}
// Zeroes the block and fills its scale pair from the n floats at x (the upstream gradient image).
// gain: the largest factor a contribution may carry over the upstream gradient beyond the 2^21 of
// headroom the scale leaves (1 for the rasterizer / shading passes; 1 / min(sigma, gamma) for SoftRas).
inline int launch_det_scale(const float *x, size_t n, float gain, float *det_block, hipStream_t s) {
  if (zero_async(det_block, kDetBlockBytes, s) != hipSuccess) return check_launch();
  const size_t want = (n + 255) / 256;
  hipLaunchKernelGGL(k_det_abs_max, dim3((unsigned)(want < 2048 ? (want ? want : 1) : 2048)), dim3(256), 0, s, x, n,
                     (int *)det_block + 4);
  int rc = check_launch();
  if (rc != MR_OK) return rc;
  hipLaunchKernelGGL(k_det_scale_from_bits, dim3(1), dim3(1), 0, s, (const int *)det_block + 4, gain, det_block);
  return check_launch();
}

template <int SLOTS>
__device__ __forceinline__ int run_find_slot(int *keys, int tri) {
  static_assert((SLOTS & (SLOTS - 1)) == 0, "power of two");
  unsigned h = (((unsigned)tri * 2654435761u) >> 12) & (SLOTS - 1);
  for (int probe = 0; probe < kRunMaxProbe; ++probe) {
    const int old = atomicCAS(&keys[h], -1, tri);
    if (old == -1 || old == tri) return (int)h;
    h = (h + 1) & (SLOTS - 1);
  }
  return -1;
}

// VAL = float (float atomics) or long long (deterministic mode: fixed point, `scale` = 2^k)
template <int N, int STRIDE, int SLOTS, class VAL>
__device__ __forceinline__ void run_flush(int *keys, VAL *vals, VAL *acc_img, int tri,
                                          float (&a)[N], float scale, int *overflow) {
  if (tri < 0) return;
  const int slot = run_find_slot<SLOTS>(keys, tri);
  VAL *dst = slot >= 0 ? &vals[slot * N] : &acc_img[(size_t)tri * STRIDE];  // saturated table: straight to HBM
#pragma unroll
  for (int k = 0; k < N; ++k) {
    if constexpr (sizeof(VAL) == 8) atomic_add_fixed((long long *)&dst[k], a[k], scale, overflow);
    else atomicAdd(&dst[k], a[k]);
  }
#pragma unroll
  for (int k = 0; k < N; ++k) a[k] = 0.0f;
}

template <class Fn, bool DET>
__global__ __launch_bounds__(kRunThreads, DET ? 1 : Fn::kMinWavesPerSimd) void k_accumulate_runs(
    Fn fn, int T, int W, int H, int regions_x, int regions_per_image, int n_regions,
    int regions_per_xcd, float *__restrict__ acc, const float *__restrict__ det_scale) {
  constexpr int N = Fn::kN, STRIDE = Fn::kStride, SLOTS = Fn::kSlots;
  static_assert(N <= STRIDE, "accumulator row too small");
  using VAL = typename std::conditional<DET, long long, float>::type;
  __shared__ int s_keys[SLOTS];
  __shared__ VAL s_vals[SLOTS * N];
  const float scale = DET ? det_scale[0] : 0.0f;
  int *overflow = DET ? det_overflow_flag(det_scale) : nullptr;

  const int region = xcd_contiguous_block((int)blockIdx.x, n_regions, regions_per_xcd);
  if (region < 0) return;
  const int img = region / regions_per_image;
  const int rr = region - img * regions_per_image;
  const int ry = rr / regions_x;
  const int rx = rr - ry * regions_x;

  const int tid = (int)threadIdx.x;
  for (int i = tid; i < SLOTS; i += kRunThreads) s_keys[i] = -1;
  for (int i = tid; i < SLOTS * N; i += kRunThreads) s_vals[i] = (VAL)0;
  __syncthreads();

  const int lane = tid & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform (SGPR)
  const int x = rx * kWave + lane;
  const int y_begin = ry * kRunRegionH + wave * kRunRowsPerWave;
  const int y_end = min(y_begin + kRunRowsPerWave, H);
  VAL *acc_img = (VAL *)acc + (size_t)img * T * STRIDE;

  typename Fn::Image image_sums;
  fn.begin_image(img, image_sums);
  if (x < W) {
    float a[N];
#pragma unroll
    for (int k = 0; k < N; ++k) a[k] = 0.0f;
    int run_tri = -1;
    typename Fn::Triangle tri_data;
    size_t pix = ((size_t)img * H + y_begin) * W + x;
    typename Fn::Raw raw_next;
    if (y_begin < y_end) fn.fetch(img, x, y_begin, pix, raw_next);
    for (int y = y_begin; y < y_end; ++y, pix += W) {
      const typename Fn::Raw raw = raw_next;
      if (y + 1 < y_end) fn.fetch(img, x, y + 1, pix + W, raw_next);  // software prefetch
      int tri;
      typename Fn::Pixel p;
      if (!fn.prepare(raw, T, tri, p)) continue;
      if (tri != run_tri) {
        run_flush<N, STRIDE, SLOTS, VAL>(s_keys, s_vals, acc_img, run_tri, a, scale, overflow);
        run_tri = tri;
        fn.load_triangle(img, tri, tri_data);
      }
      fn.accumulate(p, tri_data, a, image_sums);
    }
    run_flush<N, STRIDE, SLOTS, VAL>(s_keys, s_vals, acc_img, run_tri, a, scale, overflow);
  }
  fn.end_image(img, image_sums);  // every lane takes part (wave-level reduction inside)
  __syncthreads();

  // Drain: consecutive lanes take consecutive floats of a slot -> one contiguous
  // <= 4*kN-byte row of global float atomics per triangle.
  constexpr int kLanesPerSlot = N <= 16 ? 16 : (N <= 32 ? 32 : 64);
  for (int i = tid; i < SLOTS * kLanesPerSlot; i += kRunThreads) {
    const int slot = i / kLanesPerSlot, k = i % kLanesPerSlot;
    const int tri = s_keys[slot];
    if (tri >= 0 && k < N) {
      if constexpr (DET) atomicAdd((unsigned long long *)&acc_img[(size_t)tri * STRIDE + k],
                                   (unsigned long long)s_vals[slot * N + k]);
      else atomicAdd(&acc_img[(size_t)tri * STRIDE + k], s_vals[slot * N + k]);
    }
  }
}

// ---------------------------------------------------------------------------------------
// Row-transposed variant for functors whose N sums per triangle are PRODUCTS of a few
// per-pixel factors (the fused shading backward: 36 sums = 3 barycentrics x 12 gradients).
//
// With 36 register accumulators per lane the run-based kernel above sits at 2-3 waves per
// SIMD, and every row on which ANY lane finishes a run costs 36 ds_add_f32 wave-instructions
// (~26 LDS cycles each on gfx950, whatever the number of active lanes): 0.8 ms of 1.17 ms at
// 1024^2 x 32.  Here no sum lives in a pixel lane's registers:
//   1. each lane evaluates its pixel's kFactors factors (NOT the N products) and parks them in
//      an LDS row (kFactorStride / 4 ds_write_b128; a 20-float stride is bank-conflict-free);
//   2. lanes 0..N-1 then walk the 64 parked pixels left to right -- uniform control flow,
//      the pixel's triangle id comes from v_readlane -- lane o accumulating
//      factor[ia(o)] * factor[ib(o)] with one FMA per pixel; where the id changes the N-lane
//      partial leaves as ONE contiguous global float atomic (144 B for N = 36) into
//      acc[image][triangle][kStride].
// Forming the products in the reduction instead of per pixel saves N multiplies and
// (N - kFactors) / 4 LDS writes per pixel and, above all, N live registers in the pixel math.
// Triangle data is still fetched once per vertical run and kept in registers.
//
// Extra functor members: kCountBackground (if true, Image has an int n_bg that counts the pixels
// prepare() rejected), kFactors, kFactorStride (multiple of 4, >= kFactors),
//   static void factor_pair(int o, int &ia, int &ib);            // 0 <= o < kN
//   void factors(const Pixel &, const Triangle &, float (&f)[kFactorStride], Image &) const;
#ifndef MR_ROWS_MERGE_SLOTS
#define MR_ROWS_MERGE_SLOTS 16  // per-wavefront merge table of k_accumulate_rows (0: commit every segment)
#endif
// 8-byte LDS reads for the transposed reduction, written as inline assembly: left to itself hipcc
// pairs neighbouring 8-byte reads into ds_read2_b64, which moves 128 B/clk where ds_read_b64 moves
// 256 B/clk (MI355X_MICROARCH.md, LDS table) -- exactly the factor the factor-major layout is for.
// Each statement waits for its own reads (lgkmcnt(0)): outputs are valid when it ends.
typedef float lds_v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned lds_address(const float *p) {
  return (unsigned)(size_t)(const __attribute__((address_space(3))) float *)p;
}
__device__ __forceinline__ void lds_read_pairs4(unsigned addr_a, unsigned addr_b, lds_v2f (&a)[4], lds_v2f (&b)[4]) {
  asm volatile(
      "ds_read_b64 %0, %8\n\tds_read_b64 %4, %9\n\t"
      "ds_read_b64 %1, %8 offset:8\n\tds_read_b64 %5, %9 offset:8\n\t"
      "ds_read_b64 %2, %8 offset:16\n\tds_read_b64 %6, %9 offset:16\n\t"
      "ds_read_b64 %3, %8 offset:24\n\tds_read_b64 %7, %9 offset:24\n\t"
      "s_waitcnt lgkmcnt(0)"
      : "=&v"(a[0]), "=&v"(a[1]), "=&v"(a[2]), "=&v"(a[3]), "=&v"(b[0]), "=&v"(b[1]), "=&v"(b[2]), "=&v"(b[3])
      : "v"(addr_a), "v"(addr_b) : "memory");
}

// (Round 3, measured and dropped: software-pipelining this loop.  Hand-issued reads in inline assembly
// -- the next group's reads in flight under the FMAs -- made the register allocator copy registers
// whose loads were still in flight; the same with compiler-visible 16-byte reads, rolling or
// double-buffered, came out with eight v_mov per group and an lgkmcnt(0) at the loop-carried copies:
// 0.381 -> 0.416 ms.  What did help is not reducing per pixel at all: k_accumulate_lanes below.)
// Fn::kRowsPerWave: rows a wavefront of k_accumulate_rows walks down its 64-pixel-wide strip.  Short
// strips balance the load (background rows cost almost nothing, silhouette strips a lot) better than
// long ones amortise the merge table: measured at 1024^2 x 32, shading backward: 4 -> 0.417,
// 8 -> 0.405, 16 -> 0.412, 32 -> 0.440, 64 -> 0.502 ms; the specular backward (2 waves per SIMD, a
// per-image epilogue per wavefront): 8 -> 4.39, 16 -> 3.89, 32 -> 4.01 ms per step.
#ifndef MR_ROWS_PER_WAVE
#define MR_ROWS_PER_WAVE 8
#endif
// One wavefront per workgroup: the wavefronts of k_accumulate_rows share nothing (each has its own
// staging rows and merge table), and single-wave workgroups are placed and retired one by one --
// 256 threads -> 0.405, 128 -> 0.391, 64 -> 0.382 ms (shading backward, 1024^2 x 32).
#ifndef MR_ROWS_THREADS
#define MR_ROWS_THREADS 64
#endif
constexpr int kRowsThreads = MR_ROWS_THREADS;

template <class Fn, bool DET>
__global__ __launch_bounds__(kRowsThreads, Fn::kMinWavesPerSimd) void k_accumulate_rows(
    Fn fn, int T, int W, int H, int regions_x, int regions_per_image, int n_regions,
    int regions_per_xcd, float *__restrict__ acc, const float *__restrict__ det_scale) {
  constexpr int N = Fn::kN, STRIDE = Fn::kStride, F = Fn::kFactorStride;
  static_assert(F % 4 == 0 && Fn::kFactors <= F && N <= kWave && N <= STRIDE, "row layout");
  // Parked factors, FACTOR-major: row f holds factor f of the wavefront's 64 pixels, so that a
  // reducing lane fetches TWO consecutive pixels of its factor with one ds_read_b64 (256 B/clk
  // instead of the 128 B/clk of ds_read_b32: the reduction is bound by LDS reads).  Row stride 74
  // floats: consecutive rows start 10 banks apart (mod 64: 32 distinct even offsets), pairs never
  // collide; 6 spare columns behind a row absorb the last batch's over-read.
  constexpr int kRowStride = 74;
  static_assert(Fn::kFactors <= 32, "bank layout of the parked rows: 32 distinct even bank offsets");
  __shared__ __attribute__((aligned(16))) float s_stage[kRowsThreads / kWave][Fn::kFactors * kRowStride];
  // Per-wavefront merge table: a triangle's segments of consecutive rows are summed here, in LDS,
  // and leave as ONE N-lane global atomic per (wavefront, triangle) instead of one per (row,
  // segment).  Global float atomics run at one wave-instruction per ~50 ns per CU whatever their
  // width (MI355X_MICROARCH.md): at 1024^2 x 32 the ~3.5 commits per 64-pixel row were 7000 of them
  // per CU -- as long as the kernel itself.  The table is private to its wavefront (LDS executes a
  // wavefront's operations in order: plain read-add-write, no LDS atomics); lane i of `merge_keys`
  // holds the triangle id of slot i, so a lookup is one compare + ballot.
  // A slot is a whole 64-float row (lanes >= N carry along a copy of lane N-1's sum): reads and
  // writes of a slot then need no lane mask.
  constexpr int kMergeSlots = MR_ROWS_MERGE_SLOTS;
  static_assert(kMergeSlots <= 32, "slot lookups keep the low 32 bits of the ballot");
  __shared__ float s_merge[kRowsThreads / kWave][kMergeSlots > 0 ? kMergeSlots * kWave : 1];

  const int region = xcd_contiguous_block((int)blockIdx.x, n_regions, regions_per_xcd);
  if (region < 0) return;
  const int img = region / regions_per_image;
  const int rr = region - img * regions_per_image;
  const int ry = rr / regions_x;
  const int rx = rr - ry * regions_x;

  const int tid = (int)threadIdx.x;
  const int lane = tid & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform (SGPR)
  const int x = rx * kWave + lane;
  const bool in_range = x < W;
  const int xc = in_range ? x : W - 1;
  constexpr int kRowsPerWave = Fn::kRowsPerWave, kRowsRegionH = kRowsPerWave * (kRowsThreads / kWave);
  const int y_begin = ry * kRowsRegionH + wave * kRowsPerWave;
  const int y_end = min(y_begin + kRowsPerWave, H);
  float *acc_img = acc + (size_t)img * T * STRIDE;
  long long *acc_fixed = (long long *)acc + (size_t)img * T * STRIDE;  // DET: 8-byte elements
  const float to_fixed = DET ? det_scale[0] : 0.0f;
  float *stage = s_stage[wave];
  int ia, ib;  // the two factors whose product this lane sums (lanes >= N idle along)
  Fn::factor_pair(min(lane, N - 1), ia, ib);
  const unsigned row_a = lds_address(stage + ia * kRowStride), row_b = lds_address(stage + ib * kRowStride);
  typedef lds_v2f v2f;

  float *merge = s_merge[wave];
  int merge_keys = -1;   // lane i < kMergeSlots: triangle id held by slot i (-1: free)
  int merge_count = 0;   // slots in use, wave-uniform
  auto commit = [&](const int t, const float v) {  // one contiguous N-lane atomic into the triangle's row
    if (lane < N) {
      if (DET) atomic_add_fixed(&acc_fixed[(size_t)t * STRIDE + lane], v, to_fixed, det_overflow_flag(det_scale));
      else atomicAdd(&acc_img[(size_t)t * STRIDE + lane], v);
    }
  };
  auto flush_merge_table = [&]() {
#pragma unroll 1
    for (int slot = 0; slot < merge_count; ++slot)
      commit(__builtin_amdgcn_readlane(merge_keys, slot), merge[slot * kWave + lane]);
    merge_count = 0;
    merge_keys = -1;
  };

  typename Fn::Image image_sums;
  fn.begin_image(img, image_sums);
  int run_tri = -1;
  typename Fn::Triangle tri_data;
  size_t pix = ((size_t)img * H + y_begin) * W + xc;
  typename Fn::Raw raw_next;
  if (y_begin < y_end) fn.fetch(img, xc, y_begin, pix, raw_next);
  for (int y = y_begin; y < y_end; ++y, pix += W) {  // wave-uniform trip count
    const typename Fn::Raw raw = raw_next;
    if (y + 1 < y_end) fn.fetch(img, xc, y + 1, pix + W, raw_next);  // software prefetch
    int tri = -1;
    typename Fn::Pixel p;
    const bool valid = fn.prepare(raw, T, tri, p) && in_range;
    if (Fn::kCountBackground && in_range && !valid) image_sums.n_bg += 1;  // see Fn::end_image
    if (!__ballot(valid)) continue;  // nothing in this row segment (background)
    if (valid && tri != run_tri) {
      run_tri = tri;
      fn.load_triangle(img, tri, tri_data);
    }
    // Every lane parks its factors; lanes without a covered pixel park zeros, so that a segment
    // may simply run on to the next head: what lies between adds nothing.
    {
      float f[F];
      if (valid) {
        fn.factors(p, tri_data, f, image_sums);
      } else {
#pragma unroll
        for (int k = 0; k < Fn::kFactors; ++k) f[k] = 0.0f;
      }
#pragma unroll
      for (int k = 0; k < Fn::kFactors; ++k) stage[k * kRowStride + lane] = f[k];
    }
    const int my_tri = valid ? tri : -1;
    // segment heads along x: a valid pixel whose left neighbour holds another id (or none)
    const int left_tri = __shfl_up(my_tri, 1);
    const bool head = valid && (lane == 0 || left_tri != my_tri);
    unsigned long long heads = __ballot(head);
    const unsigned long long valids = __ballot(valid);
    __builtin_amdgcn_wave_barrier();  // LDS executes one wavefront's operations in order
    // ONE pass over the row's pixels in groups of eight (four 8-byte reads per factor): a group
    // without a head is eight FMAs straight; only at a head (~3.5 per row) the running sum is
    // closed -- into the merge table -- and restarted.  No per-segment loops, no masks, no tails.
    int cur_t = -1;           // triangle of the running segment, wave-uniform
    int cur_slot = -1;        // its merge-table slot, claimed when the segment starts (-1: none) ...
    float merged = 0.0f;      // ... and that slot's value so far, requested at the same time
    float sum = 0.0f, sum2 = 0.0f;  // two chains (even / odd pixels): a dependent FMA issues every ~6.6 clocks, an independent one every 4
    auto close_segment = [&]() {
      if (cur_slot >= 0) merge[cur_slot * kWave + lane] = merged + (sum + sum2);
      else if (cur_t >= 0) commit(cur_t, sum + sum2);  // no table, or more segments in one row than it has slots
    };
    auto open_segment = [&](const int pixel) {
      close_segment();
      cur_t = __builtin_amdgcn_readlane(my_tri, pixel);
      const unsigned hit = kMergeSlots > 0 ? (unsigned)__ballot(merge_keys == cur_t) : 0u;
      merged = 0.0f;
      if (hit) {
        cur_slot = __builtin_ctz(hit);
        merged = merge[cur_slot * kWave + lane];
      } else if (merge_count < kMergeSlots) {  // (the table was emptied before the pass if it was short of room)
        cur_slot = merge_count;
        if (lane == merge_count) merge_keys = cur_t;
        merge_count += 1;
      } else {
        cur_slot = -1;
      }
      sum = 0.0f;
      sum2 = 0.0f;
    };
    // each head may claim a slot of the merge table: if they might not all fit, everything in the
    // table leaves now (once per row at most, instead of a check per segment)
    if (kMergeSlots > 0 && merge_count + (int)__builtin_popcountll(heads) > kMergeSlots) flush_merge_table();
    const int first_group = heads ? (int)(__builtin_ctzll(heads) >> 3) : 8;
    const int last_group = valids ? (63 - (int)__builtin_clzll(valids)) >> 3 : -1;
    for (int g = first_group; g <= last_group; ++g) {
      v2f ra[4], rb[4];
      lds_read_pairs4(row_a + 32u * g, row_b + 32u * g, ra, rb);
      const unsigned hg = (unsigned)(heads >> (8 * g)) & 0xffu;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (hg & (1u << (2 * j))) open_segment(8 * g + 2 * j);
        sum = fmaf(ra[j].x, rb[j].x, sum);
        if (hg & (2u << (2 * j))) open_segment(8 * g + 2 * j + 1);
        sum2 = fmaf(ra[j].y, rb[j].y, sum2);
      }
    }
    close_segment();
    __builtin_amdgcn_wave_barrier();
  }
  flush_merge_table();
  fn.end_strip(img, region, image_sums);
}

// ---------------------------------------------------------------------------------------
// Lane-accumulating variant (round 3) for functors with few sums per triangle (N <= ~27).
//
// The rows kernel above spends one reduction step per PIXEL (64 per row: ~230 issued instructions
// and eight exposed LDS round trips per 64-pixel row, as much as the pixel math itself).  Here every
// lane adds its pixel's N products to N REGISTER accumulators for as long as it stays on one
// triangle walking down its column -- N FMAs per row -- and only a FINISHED vertical run (the
// triangle under the lane changed, or the strip ended) goes through LDS: the lane parks its N sums,
// and lanes 0..N-1 walk the finished lanes of this row -- one add per finished lane, ~7 per row at
// 1024^2 / 5k triangles instead of 64 -- merging neighbours that finished the same triangle and
// committing through the same per-wavefront merge table as the rows kernel.
// Measured at 1024^2 x 32, 5k triangles, shading backward with 18 sums (vertex gradients only):
// rows kernel 0.381 ms -> 0.275 ms; per launch 179.5 -> ~150 M vector, 87 -> 52 M scalar, 34 -> 16 M
// branch and 32.6 -> 6.8 M LDS instructions.  What was measured on the way: requesting the new
// triangles' records BEFORE the flush instead of after it -0.02 ms (their L2 round trip flies under
// the flush's LDS work); a dense pass for flushes of 20 or more lanes (the strip's end) -0.005; strips
// of 16 rows (8: +0.004, 32: +0.02, 64: +0.03 ms); 16 merge slots (8: +0.005, 32: +0.07 -- LDS per
// wavefront); merging with ds_add_f32 instead of read + add + store: +0.08 ms (an LDS float atomic
// costs ~26 LDS cycles whatever it does); restarting the accumulators with in-place selects
// instead of a branch (the compiler copied all of them out and back: 75 v_mov per row) -0.01.
//
// Functor interface as for k_accumulate_runs (accumulate() adds the pixel's N products to a[]) plus
//   static int column(int o);          // float of the triangle's acc row that sum o belongs to
//   static constexpr int kLaneRowsPerWave;
#ifndef MR_LANES_DENSE
#define MR_LANES_DENSE 20   // finished lanes from which a flush walks all 64 parked rows (65: never)
#endif
#ifndef MR_LANES_MERGE_SLOTS
#define MR_LANES_MERGE_SLOTS 16
#endif

constexpr int lanes_park_stride(int n) {  // multiple of 4 with an odd number of quads: per-lane b128 accesses are conflict-free
  int s = (n + 3) / 4;
  if (s % 2 == 0) s += 1;
  return s * 4;
}

// Fn::kPipelinedRows (optional member, default false): see the pipelined row loop in k_accumulate_lanes
template <class Fn, class = void>
struct LanesPipelined { static constexpr bool value = false; };
template <class Fn>
struct LanesPipelined<Fn, decltype((void)Fn::kPipelinedRows)> { static constexpr bool value = Fn::kPipelinedRows; };
template <class Fn>
constexpr bool lanes_pipelined() { return LanesPipelined<Fn>::value; }
// Fn::kPipelinedConditionalRecords (optional, default false): the pipelined loop (re)loads a lane's triangle
// records only when its triangle changed -- for functors whose records are large (the shading's 11 x 16 bytes
// per lane: unconditional loads would be 176 B per pixel of L2 traffic)
template <class Fn, class = void>
struct LanesPipelinedConditional { static constexpr bool value = false; };
template <class Fn>
struct LanesPipelinedConditional<Fn, decltype((void)Fn::kPipelinedConditionalRecords)> {
  static constexpr bool value = Fn::kPipelinedConditionalRecords;
};

// Fn::kSkipsStrips (optional member, default false): the functor can tell from a side table that a whole strip has
// nothing to contribute -- fn.skip_strip(img, strip column, first row, end row) -- and the wavefront leaves at once
template <class Fn, class = void>
struct LanesSkipStrips { static constexpr bool value = false; };
template <class Fn>
struct LanesSkipStrips<Fn, decltype((void)Fn::kSkipsStrips)> { static constexpr bool value = Fn::kSkipsStrips; };

template <class Fn, bool DET>
__global__ __launch_bounds__(kWave, Fn::kMinWavesPerSimd) void k_accumulate_lanes(
    Fn fn, int T, int W, int H, int regions_x, int regions_per_image, int n_regions,
    int regions_per_xcd, int rows_per_wave, float *__restrict__ acc, const float *__restrict__ det_scale) {
  constexpr int N = Fn::kN, STRIDE = Fn::kStride, P = lanes_park_stride(N);
  static_assert(N <= kWave && N <= STRIDE, "one reduction lane per sum");
  __shared__ __attribute__((aligned(16))) float s_park[kWave * P];
  constexpr int kMergeSlots = MR_LANES_MERGE_SLOTS;
  static_assert(kMergeSlots >= 1 && kMergeSlots <= 32, "lane i of merge_keys holds slot i's triangle; slot lookups keep the low 32 bits of the ballot");
  // (round 6: [slot][sum] -- only lanes 0..N-1 hold a sum; lanes >= N idle along on lane N-1's address with its value --
  //  0.6 KB instead of 4 KB per wavefront at N = 9: LDS no longer caps the kernel at five wavefronts per SIMD)
  constexpr int kMergeStride = N;
  __shared__ float s_merge[kMergeSlots * kMergeStride];

  const int region = xcd_contiguous_block((int)blockIdx.x, n_regions, regions_per_xcd);
  if (region < 0) return;
  const int img = region / regions_per_image;
  const int rr = region - img * regions_per_image;
  const int ry = rr / regions_x;
  const int rx = rr - ry * regions_x;

  const int lane = (int)threadIdx.x;
  const int x = rx * kWave + lane;
  const bool in_range = x < W;
  const int xc = in_range ? x : W - 1;
  const int y_begin = ry * rows_per_wave;
  const int y_end = min(y_begin + rows_per_wave, H);
  if constexpr (LanesSkipStrips<Fn>::value) {
    if (fn.skip_strip(img, rx, y_begin, y_end)) return;   // wave-uniform
  }
  float *acc_img = acc + (size_t)img * T * STRIDE;
  long long *acc_fixed = (long long *)acc + (size_t)img * T * STRIDE;  // DET: 8-byte elements
  const float to_fixed = DET ? det_scale[0] : 0.0f;
  const int red = min(lane, N - 1);     // the sum this lane reduces (lanes >= N idle along on a copy)
  const int col = Fn::column(red);      // ... and its float inside the triangle's acc row

  int merge_keys = -1;   // lane i < kMergeSlots: triangle id held by slot i (-1: free)
  int merge_count = 0;   // slots in use, wave-uniform
  auto commit = [&](const int t, const float v) {
    if (lane < N) {
      if (DET) atomic_add_fixed(&acc_fixed[(size_t)t * STRIDE + col], v, to_fixed, det_overflow_flag(det_scale));
      else atomicAdd(&acc_img[(size_t)t * STRIDE + col], v);
    }
  };
  auto flush_merge_table = [&]() {
#pragma unroll 1
    for (int slot = 0; slot < merge_count; ++slot)
      commit(__builtin_amdgcn_readlane(merge_keys, slot), s_merge[slot * kMergeStride + red]);
    merge_count = 0;
    merge_keys = -1;
  };

  typename Fn::Image image_sums;
  fn.begin_image(img, image_sums);
  float a[N];
#pragma unroll
  for (int k = 0; k < N; ++k) a[k] = 0.0f;
  int run_tri = -1;    // triangle the lane's sums belong to (-1: the sums are zero)
  int data_tri = -1;   // triangle whose records the lane holds
  typename Fn::Triangle tri_data;

  // Finished runs: park, reduce over the finished lanes, restart.
  //   sparse (a few lanes finished, the usual row): lanes 0..N-1 visit the finished lanes four at a
  //          time -- their four reads in flight together --, one add each;
  //   dense  (kDense or more, e.g. the end of the strip, where every lane finishes): every lane parks
  //          -- the unfinished ones zeros -- and the pass walks all 64 rows in groups of eight, opening
  //          a segment where the triangle changes (a ballot tells where), like the rows kernel.
  // A segment's sum goes into its triangle's merge-table slot as read + add + store (ds_add_f32 instead: measured,
  // 0.305 -> 0.368 ms, an LDS float atomic costs ~26 LDS cycles whatever it does).
  // (Measured in round 4 and dropped: DEFERRING the pass -- a finished run only parks, the lane remembers its
  //  triangle in a register, and the pass runs when a lane needs its slot again and at the end of the strip:
  //  ~3 passes per 8-row strip instead of one per row.  Shading backward per launch 115.8 -> 112.3 M vector, 57.1
  //  -> 50.0 M scalar, 19.5 -> 16.3 M branch, 6.8 -> 5.7 M LDS instructions, and 0.2213 -> 0.2224 ms: the pass is
  //  not where this kernel's time goes.)
  constexpr int kDense = MR_LANES_DENSE;
  auto flush = [&](const bool fin) __attribute__((always_inline)) {
    const unsigned long long finm = __ballot(fin);
    if (!finm) return;
    const bool dense = (int)__builtin_popcountll(finm) >= kDense;  // wave-uniform
    // (Measured in round 6 and dropped: the FEW finished lanes of an ordinary row -- ~3 of 64 -- committing their own N
    //  sums straight to the accumulator rows, N no-return global atomics under the finished lanes' mask, instead of the
    //  walk through LDS below.  Scalar instructions per launch 57.0 -> 34.7 M, vector 119.1 -> 117.8 M -- and the kernel
    //  0.213 -> 1.17 ms, same box, twice: ~3.7 M atomic INSTRUCTIONS per launch instead of ~0.4 M; a float atomic
    //  instruction costs the wavefront ~0.3 us whatever its mask.  profiles/r06_lanes_sparse_atomics.txt)
    {
      float4 *dst = (float4 *)(s_park + lane * P);
      auto quad = [&](const int q) {
        return make_float4(a[4 * q], 4 * q + 1 < N ? a[4 * q + 1] : 0.f, 4 * q + 2 < N ? a[4 * q + 2] : 0.f,
                           4 * q + 3 < N ? a[4 * q + 3] : 0.f);
      };
      if (dense) {  // wave-uniform: every lane parks
#pragma unroll
        for (int q = 0; q < (N + 3) / 4; ++q) {
          const float4 v = quad(q);
          dst[q] = make_float4(fin ? v.x : 0.f, fin ? v.y : 0.f, fin ? v.z : 0.f, fin ? v.w : 0.f);
        }
      } else if (fin) {
#pragma unroll
        for (int q = 0; q < (N + 3) / 4; ++q) dst[q] = quad(q);
      }
      // restart as selects on the accumulators in place (inside the branch above the compiler
      // copied all N of them out and back: 75 vector instructions per row)
#pragma unroll
      for (int k = 0; k < N; ++k) a[k] = fin ? 0.0f : a[k];
    }
    const int old = fin ? run_tri : -1;
    if (fin) run_tri = -1;
    __builtin_amdgcn_wave_barrier();  // LDS executes one wavefront's operations in order
    int cur_t = -1, cur_slot = -1;    // triangle of the running segment and its merge-table slot
    float sum = 0.0f, sum2 = 0.0f, merged = 0.0f;
    auto close_segment = [&]() {
      if (cur_t < 0) return;
      float *slot = &s_merge[cur_slot * kMergeStride + red];
      *slot = merged + (sum + sum2);
    };
    auto open_segment = [&](const int t) {
      close_segment();
      cur_t = t;
      const unsigned hit = (unsigned)__ballot(merge_keys == t);
      merged = 0.0f;
      if (hit) {
        cur_slot = __builtin_ctz(hit);
        merged = s_merge[cur_slot * kMergeStride + red];
      } else {
        if (merge_count == kMergeSlots) flush_merge_table();  // full: everything leaves, then slot 0
        cur_slot = merge_count;
        if (lane == merge_count) merge_keys = t;
        merge_count += 1;
      }
      sum = 0.0f;
      sum2 = 0.0f;
    };
    const float *src = s_park + red;
    if (dense) {
      const int left = __shfl_up(old, 1);
      const unsigned long long heads = __ballot(fin && (lane == 0 || left != old));
      const int g0 = (int)(__builtin_ctzll(finm) >> 3), g1 = (63 - (int)__builtin_clzll(finm)) >> 3;
      for (int g = g0; g <= g1; ++g) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = src[(8 * g + j) * P];
        const unsigned hg = (unsigned)(heads >> (8 * g)) & 0xffu;
        if (hg == 0u) {
          sum += (v[0] + v[1]) + (v[2] + v[3]);
          sum2 += (v[4] + v[5]) + (v[6] + v[7]);
        } else {
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            if (hg & (1u << j)) open_segment(__builtin_amdgcn_readlane(old, 8 * g + j));
            sum += v[j];
          }
        }
      }
    } else {
      unsigned long long m = finm;
      int left = (int)__builtin_popcountll(finm);
      while (left > 0) {  // four finished lanes per trip: their reads are in flight together
        int l[4], t[4];
        float v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {  // (behind the last one: ffs(0) - 1 = -1 -> lane 63, read and ignored)
          l[i] = (__builtin_ffsll((long long)m) - 1) & 63;
          m &= ~(1ull << l[i]);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = src[l[i] * P];
#pragma unroll
        for (int i = 0; i < 4; ++i) t[i] = __builtin_amdgcn_readlane(old, l[i]);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          if (i < left) {
            if (t[i] != cur_t) open_segment(t[i]);
            sum += v[i];
          }
        }
        left -= 4;
      }
    }
    close_segment();
    __builtin_amdgcn_wave_barrier();
  };

  size_t pix = ((size_t)img * H + y_begin) * W + xc;
  if constexpr (lanes_pipelined<Fn>()) {
    // Streamed planes TWO rows ahead (functors whose pixel math is light).  vmcnt retires loads in issue
    // order: in the plain loop below the wait for a row's triangle records (an L2 round trip) also drains the
    // next row's G-buffer / upstream loads issued just before them (an HBM round trip), so every row costs
    // one HBM latency (~1.9 us per row and wavefront at 1024^2 x 32, 66 % of the wave-cycles waiting).
    // Here a row's streamed values are requested two iterations before they are used, AFTER that
    // iteration's record loads, into the register set whose row has just been unpacked by prepare():
    // two sets used alternately by an iteration pair -- copying a pending load's destination would make
    // the compiler wait for it.
    typename Fn::Raw raw_a, raw_b;   // raw_a: rows y_begin, + 2, ...; raw_b: rows y_begin + 1, + 3, ...
    if (y_begin < y_end) fn.fetch(img, xc, y_begin, pix, raw_a);
    if (y_begin + 1 < y_end) fn.fetch(img, xc, y_begin + 1, pix + W, raw_b);
    auto row = [&](const int y, typename Fn::Raw &raw) __attribute__((always_inline)) {
      int tri = -1;
      typename Fn::Pixel p;
      const bool valid = fn.prepare(raw, T, tri, p) && in_range;
      if (Fn::kCountBackground && in_range && !valid) image_sums.n_bg += 1;
      const bool any = __ballot(valid) != 0ull;
      // EVERY lane (re)loads its row's records, with no branch around the loads: behind `if (tri !=
      // data_tri)` hipcc loads into temporaries and copies them into tri_data inside the branch -- an
      // s_waitcnt right behind the loads, the whole L2 round trip exposed.  Unconditional loads land in
      // tri_data's registers and are waited for where accumulate() first uses them, after the refill below
      // has been issued and the flush has run.  (Lanes without a pixel re-read the record they hold.)
      if constexpr (LanesPipelinedConditional<Fn>::value) {
        if (valid && tri != data_tri) {
          data_tri = tri;
          fn.load_triangle(img, tri, tri_data);
        }
      } else {
        data_tri = valid ? tri : max(data_tri, 0);
        fn.load_triangle(img, data_tri, tri_data);
      }
      if (y + 2 < y_end) fn.fetch(img, xc, y + 2, pix + 2 * (size_t)W, raw);  // refill: row y + 2
      pix += W;
      if (!any) return;  // nothing in this row segment (background); open runs stay open
      // (requesting the refill after the flush instead -- straight-line code away from accumulate()'s wait --
      //  measured slower: 0.268 -> 0.275 ms; hipcc's vmcnt placement around the flush's branches and atomics
      //  is conservative either way)
      flush(valid && run_tri >= 0 && tri != run_tri);
      if (valid) {
        run_tri = tri;
        fn.accumulate(p, tri_data, a, image_sums);
      }
    };
    for (int y = y_begin; y < y_end; y += 2) {  // wave-uniform trip count
      row(y, raw_a);
      if (y + 1 < y_end) row(y + 1, raw_b);
    }
  } else {
  // (Measured in round 4 and dropped: requesting the next row into the SAME registers right after prepare() has
  //  unpacked the current one -- it removes the 11 v_mov per row that copy `raw_next` into `raw`, but the loads then
  //  issue behind the previous row's arrival: shading backward 0.263 -> 0.291 ms.)
  typename Fn::Raw raw_next;
  if (y_begin < y_end) fn.fetch(img, xc, y_begin, pix, raw_next);
  for (int y = y_begin; y < y_end; ++y, pix += W) {  // wave-uniform trip count
    const typename Fn::Raw raw = raw_next;
    if (y + 1 < y_end) fn.fetch(img, xc, y + 1, pix + W, raw_next);  // software prefetch
    int tri = -1;
    typename Fn::Pixel p;
    const bool valid = fn.prepare(raw, T, tri, p) && in_range;
    if (Fn::kCountBackground && in_range && !valid) image_sums.n_bg += 1;
    if (!__ballot(valid)) continue;  // nothing in this row segment (background); open runs stay open
    // the new triangles' records are requested BEFORE the finished runs are reduced: the flush does
    // not touch them, so their L2 round trip (11 loads per lane) flies under its LDS work
    if (valid && tri != data_tri) {
      data_tri = tri;
      fn.load_triangle(img, tri, tri_data);
    }
    flush(valid && run_tri >= 0 && tri != run_tri);
    if (valid) {
      run_tri = tri;
      fn.accumulate(p, tri_data, a, image_sums);
    }
  }
  }
  flush(run_tri >= 0);
  flush_merge_table();
  fn.end_strip(img, region, image_sums);
}

// Image-wide sums of the rows kernel's functors (light / camera gradients): every strip -- one per
// workgroup, `region` above -- leaves ONE row of them, and this kernel adds an image's rows in a fixed
// order.  (They used to be float atomics on the image's one row: every strip of an image queued up
// on the same cache line; and the deterministic mode needed a fixed-point copy of the row.)
// One workgroup per image; thread = (slot, part): 32 slots x 32 parts, eight loads in flight per thread
// (with 8 parts and one load at a time this was a 128-deep chain of load latencies, 100 us).
constexpr int kSumRowSlots = 32, kSumRowThreads = 1024;
static __global__ __launch_bounds__(kSumRowThreads) void k_sum_strip_rows(const float *__restrict__ rows, int per_image,
                                                                          int row, float *__restrict__ out) {
  constexpr int kParts = kSumRowThreads / kSumRowSlots;
  __shared__ float s_part[kParts][kSumRowSlots + 1];
  const int img = (int)blockIdx.x, slot = (int)threadIdx.x % kSumRowSlots, part = (int)threadIdx.x / kSumRowSlots;
  const float *mine = rows + (size_t)img * per_image * row;
  float v = 0.0f;
  if (slot < row) {
    constexpr int kInFlight = 8;
    for (int i = part; i < per_image; i += kParts * kInFlight) {
      float x[kInFlight];
#pragma unroll
      for (int u = 0; u < kInFlight; ++u) {
        const int k = i + u * kParts;
        x[u] = k < per_image ? mine[(size_t)k * row + slot] : 0.0f;
      }
#pragma unroll
      for (int u = 0; u < kInFlight; ++u) v += x[u];
    }
  }
  s_part[part][slot] = v;
  __syncthreads();
  if (part == 0 && slot < row) {
    float t = 0.0f;
    for (int k = 0; k < kParts; ++k) t += s_part[k][slot];
    out[(size_t)img * row + slot] = t;
  }
}

// strips (= workgroups = `region`s) of k_accumulate_rows<Fn> per image
template <class Fn>
inline int strips_per_image(int W, int H) {
  constexpr int kRowsRegionH = Fn::kRowsPerWave * (kRowsThreads / kWave);
  return ((W + kWave - 1) / kWave) * ((H + kRowsRegionH - 1) / kRowsRegionH);
}
inline int launch_sum_strip_rows(const float *rows, int B, int per_image, int row, float *out, hipStream_t s) {
  if (row > kSumRowSlots) return MR_EINVAL;
  hipLaunchKernelGGL(k_sum_strip_rows, dim3((unsigned)B), dim3(kSumRowThreads), 0, s, rows, per_image, row, out);
  return check_launch();
}

// det_scale: nullptr = float atomics; else the device pair (2^k, 2^-k) of the deterministic mode and
// `acc` holds 8-byte fixed-point elements.
template <class Fn>
inline int launch_accumulate_rows(const Fn &fn, int B, int T, int W, int H, float *acc,
                                  hipStream_t s, const float *det_scale = nullptr) {
  g_last_accumulate_kernel = __PRETTY_FUNCTION__;
  const int regions_x = (W + kWave - 1) / kWave;
  constexpr int kRowsRegionH = Fn::kRowsPerWave * (kRowsThreads / kWave);
  const int regions_y = (H + kRowsRegionH - 1) / kRowsRegionH;
  const int per_image = regions_x * regions_y;
  const int n_regions = per_image * B;
  const int per_xcd = (n_regions + kXcds - 1) / kXcds;
  if (det_scale)
    hipLaunchKernelGGL((k_accumulate_rows<Fn, true>), dim3((unsigned)(per_xcd * kXcds)),
                       dim3(kRowsThreads), 0, s, fn, T, W, H, regions_x, per_image, n_regions,
                       per_xcd, acc, det_scale);
  else
    hipLaunchKernelGGL((k_accumulate_rows<Fn, false>), dim3((unsigned)(per_xcd * kXcds)),
                       dim3(kRowsThreads), 0, s, fn, T, W, H, regions_x, per_image, n_regions,
                       per_xcd, acc, det_scale);
  return check_launch();
}

// Rows a wavefront of k_accumulate_lanes walks down its 64-pixel-wide strip: Fn::kLaneRowsPerWave
// (16 at 1024^2 x 32: 8 -> +1 %, 32 -> +7 %), halved while the launch would leave wavefront slots of
// the chip empty (256^2 x 8 in 16-row strips is 512 wavefronts for 4096 slots: the rasterizer
// backward took 0.082 ms there against the rows kernel's 0.054).
template <class Fn>
inline int lanes_rows_per_wave(int B, int W, int H) {
  int rows = Fn::kLaneRowsPerWave;
  const long columns = (long)B * ((W + kWave - 1) / kWave);
  while (rows > 4 && columns * ((H + rows - 1) / rows) < 8192) rows /= 2;
  return rows;
}
template <class Fn>
inline int lanes_strips_per_image(int B, int W, int H) {
  const int rows = lanes_rows_per_wave<Fn>(B, W, H);
  return ((W + kWave - 1) / kWave) * ((H + rows - 1) / rows);
}
template <class Fn>
inline int launch_accumulate_lanes(const Fn &fn, int B, int T, int W, int H, float *acc,
                                   hipStream_t s, const float *det_scale = nullptr) {
  g_last_accumulate_kernel = __PRETTY_FUNCTION__;
  const int rows = lanes_rows_per_wave<Fn>(B, W, H);
  const int regions_x = (W + kWave - 1) / kWave;
  const int regions_y = (H + rows - 1) / rows;
  const int per_image = regions_x * regions_y;
  const int n_regions = per_image * B;
  const int per_xcd = (n_regions + kXcds - 1) / kXcds;
  if (det_scale) return MR_EINVAL;  // the deterministic mode stays on the rows / runs kernels (not instantiated here)
  hipLaunchKernelGGL((k_accumulate_lanes<Fn, false>), dim3((unsigned)(per_xcd * kXcds)), dim3(kWave), 0, s,
                     fn, T, W, H, regions_x, per_image, n_regions, per_xcd, rows, acc, det_scale);
  return check_launch();
}

template <class Fn>
inline int launch_accumulate_runs(const Fn &fn, int B, int T, int W, int H, float *acc,
                                  hipStream_t s) {
  g_last_accumulate_kernel = __PRETTY_FUNCTION__;
  const int regions_x = (W + kWave - 1) / kWave;
  const int regions_y = (H + kRunRegionH - 1) / kRunRegionH;
  const int per_image = regions_x * regions_y;
  const int n_regions = per_image * B;
  const int per_xcd = (n_regions + kXcds - 1) / kXcds;
  hipLaunchKernelGGL((k_accumulate_runs<Fn, false>), dim3((unsigned)(per_xcd * kXcds)),
                     dim3(kRunThreads), 0, s, fn, T, W, H, regions_x, per_image, n_regions,
                     per_xcd, acc, (const float *)nullptr);
  return check_launch();
}

// Deterministic mode: `acc` holds 8-byte fixed-point elements, det_scale = device (2^k, 2^-k).
template <class Fn>
inline int launch_accumulate_runs_fixed(const Fn &fn, int B, int T, int W, int H, float *acc,
                                        const float *det_scale, hipStream_t s) {
  g_last_accumulate_kernel = __PRETTY_FUNCTION__;
  const int regions_x = (W + kWave - 1) / kWave;
  const int regions_y = (H + kRunRegionH - 1) / kRunRegionH;
  const int per_image = regions_x * regions_y;
  const int n_regions = per_image * B;
  const int per_xcd = (n_regions + kXcds - 1) / kXcds;
  hipLaunchKernelGGL((k_accumulate_runs<Fn, true>), dim3((unsigned)(per_xcd * kXcds)),
                     dim3(kRunThreads), 0, s, fn, T, W, H, regions_x, per_image, n_regions,
                     per_xcd, acc, det_scale);
  return check_launch();
}

struct F3 {
  float x, y, z;
};
// A 12-byte element of a plane that is streamed once and never reused: nontemporal (the streaming
// kernels of this package ran 1-5 % faster with it, same-box A/Bs).
__device__ __forceinline__ float4 load_streamed(const float4 *p) {
  typedef float v4f __attribute__((ext_vector_type(4)));
  const v4f v = __builtin_nontemporal_load((const v4f *)p);
  return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void store_streamed(float4 *p, const float4 v) {
  typedef float v4f __attribute__((ext_vector_type(4)));
  __builtin_nontemporal_store(v4f{v.x, v.y, v.z, v.w}, (v4f *)p);
}
__device__ __forceinline__ F3 load_streamed(const F3 *p) {
  F3 v;
  v.x = __builtin_nontemporal_load(&p->x);
  v.y = __builtin_nontemporal_load(&p->y);
  v.z = __builtin_nontemporal_load(&p->z);
  return v;
}

struct NoImageSums {};

// ---- the rasterizer's own backward, shared by raster_backward.hip and shade.hip ----
// Record per (image, triangle): sign-corrected adjugate u[9] (row i = edge i, column
// c = clip component x/y/w), its column sums and 1/|det| (rasterize_triangles.cpp
// :180-198).  64 bytes.
struct alignas(64) BwdRec {
  float4 a, b, c, d;  // a=(u0..u3) b=(u4..u7) c=(u8,S0,S1,S2) d=(1/|det|,-,-,-)
};

struct BwdTriangle {
  float u[9], s[3], inv;
};

__device__ __forceinline__ void load_bwd_triangle(const BwdRec *r, BwdTriangle &t) {
  const float4 a = r->a, b = r->b, c = r->c, d = r->d;
  t.u[0] = a.x; t.u[1] = a.y; t.u[2] = a.z; t.u[3] = a.w;
  t.u[4] = b.x; t.u[5] = b.y; t.u[6] = b.z; t.u[7] = b.w;
  t.u[8] = c.x; t.s[0] = c.y; t.s[1] = c.z; t.s[2] = c.w;
  t.inv = d.x;
}

// The nine partials of rasterize_triangles.cpp:202-269 for one pixel, added into
// acc[j*3 + c] (corner j, clip component c = x, y, w).  The reference evaluates
//   sum_i g_i * (-u_ic * b_j + s_c * b_i * b_j) / |det|
// per (j, c); here b_j is factored out of the sum PER PIXEL (30 VALU ops instead of 150):
//   b_j * [ sum_i g_i * (s_c * b_i - u_ic) ] / |det|.
// The differences (s_c b_i - u_ic) are still formed per i and per pixel, as in the reference,
// so the cancellation between them (severe for small / sliver triangles) rounds the same
// way; only sums over PIXELS must never be factored (measured: 70x the error).
// raster_pixel_q returns the bracket q_c (c = x, y, w) -- the row kernel multiplies by b_j in
// its reduction; `inv` is 1/|det|, or 0 to switch the pixel off (cpp:162).
__device__ __forceinline__ void raster_pixel_q(const F3 bary, const F3 g, const BwdTriangle &t,
                                               const float inv, float (&q)[3]) {
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float w0 = t.s[c] * bary.x - t.u[0 + c];
    const float w1 = t.s[c] * bary.y - t.u[3 + c];
    const float w2 = t.s[c] * bary.z - t.u[6 + c];
    q[c] = ((g.x * w0 + g.y * w1) + g.z * w2) * inv;
  }
}

__device__ __forceinline__ void raster_pixel_partials(const F3 bary, const F3 g, const BwdTriangle &t,
                                                      float *acc) {
  float q[3];
  raster_pixel_q(bary, g, t, t.inv, q);
  const float b[3] = {bary.x, bary.y, bary.z};
#pragma unroll
  for (int c = 0; c < 3; ++c) {
#pragma unroll
    for (int j = 0; j < 3; ++j) acc[j * 3 + c] += b[j] * q[c];
  }
}

// One thread per (image, triangle): fills BwdRec from clip-space vertices.
// zero_rows / zero_tail: see k_bwd_setup (raster_backward.hip); zero_row_bytes is a multiple of 16.
// corners + fold_recs (both or neither): also write the folded lane kernel's FoldRec[B*T] (corner_rec.h) from
// the CornerRec[B*T] of the same inputs; pull_transforms ([B,4,4], optional): in the pulled form (store_fold_record).
int launch_bwd_setup(const float *clip, const int32_t *tris, int B, int V, int T, BwdRec *recs,
                     hipStream_t s, void *zero_rows = nullptr, size_t zero_row_bytes = 0,
                     float *zero_tail = nullptr, int zero_tail_count = 0, const void *corners = nullptr,
                     void *fold_recs = nullptr, const float *pull_transforms = nullptr);

}  // namespace mr
