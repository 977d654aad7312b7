// Store-pattern micro-benchmark for the G-buffer write of k_raster (raster_forward.hip) on gfx950.
//
// Question: is the 4.0-4.8 TB/s the forward kernel writes at the ceiling of the
// part, or of the kernel's store PATTERN (one wavefront = one 16x4-pixel tile: 64-byte runs of ids
// and depths, 192-byte runs of barycentrics, 4 KB apart, 12-byte-per-lane stores)?  Every variant
// writes the same 20 B/px G-buffer (ids i32, z f32, barycentrics 3 x f32; B x H x W = 32 x 1024^2,
// 671 MB per launch) from the same grid (one 256-thread workgroup per 64x64-pixel region) and does
// nothing else: what differs is which bytes one store instruction of one wavefront covers.
//
//   hipcc --offload-arch=gfx950 -O3 -o store_patterns store_patterns.hip && ./store_patterns
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

typedef float v3f __attribute__((ext_vector_type(3)));
typedef float v4f __attribute__((ext_vector_type(4)));

enum Mode {
  kTileNt = 0,      // k_raster today: wavefront w walks tiles w, w+4, ... (a 16-pixel-wide column of tiles), nontemporal
  kTileCached,      // the same, plain stores
  kTileRowNt,       // 16x4 tiles, but a wavefront walks the four tiles of a tile ROW back to back
  kRowNt,           // wavefront = 64x1 pixels: 256-B runs of ids / z, one 768-B run of barycentrics (12 B per lane)
  kRowX4Nt,         // the same, barycentrics as 48 lanes x 16 B (aligned dwordx4)
  kRowCached,       // kRowNt with plain stores
  kRow4Nt,          // wavefront = 64x4 pixels, four 256-B / 768-B runs per plane issued back to back (what LDS staging of 4 tiles gives)
  kLinearNt,        // reference: every plane written as one linear stream, 16 B per lane
  kLinearCached,
  kTileRaster,      // round 6 calibration: k_raster's OWN store instructions -- raw buffer stores with the tile offset in soffset,
                    // ids through the caches (aux 0), depth and barycentrics nontemporal (aux 2), 16x4 tiles, column walk
  kTileRasterNoZ,   // the same without the depth plane (render()'s forward does not write it): 16 B/px
  kTile32Raster,    // k_raster's store instructions on 32x2 tiles (-DMR_TILE_W=32): every run a whole number of 128-byte lines
  kTileRasterCached,   // kTileRaster with every plane through the caches (aux 0)
  kModes
};
static const char *kNames[kModes] = {"tile16x4 column walk, nt (k_raster today)", "tile16x4 column walk, cached",
                                     "tile16x4 row walk, nt", "row64x1, nt, bary b96", "row64x1, nt, bary 48 x b128",
                                     "row64x1, cached, bary b96", "rows 64x4 back to back, nt", "linear planes, nt, b128",
                                     "linear planes, cached, b128", "k_raster's store mix (buffer stores, ids cached, z + bary nt)",
                                     "k_raster's store mix without the depth plane",
                                     "k_raster's store mix on 32x2 tiles", "k_raster's stores, all planes cached"};

template <bool NT, class T>
__device__ __forceinline__ void put(T *p, T v) {
  if (NT) __builtin_nontemporal_store(v, p);
  else *p = v;
}

template <int MODE>
__global__ __launch_bounds__(256) void k_store(int32_t *__restrict__ ids, float *__restrict__ z, float *__restrict__ bary,
                                                int W, int H, int regions_x, int regions_per_image) {
  const int region = (int)blockIdx.x;
  const int img = region / regions_per_image, rr = region % regions_per_image;
  const int X0 = (rr % regions_x) * 64, Y0 = (rr / regions_x) * 64;
  const int lane = (int)threadIdx.x & 63, wave = (int)threadIdx.x >> 6;
  const size_t img_px = (size_t)img * W * H;
  const float fv = (float)(lane + region) * 1e-3f;
  if constexpr (MODE == kTileNt || MODE == kTileCached || MODE == kTileRowNt) {
    constexpr bool NT = MODE != kTileCached;
    const int lx = lane & 15, ly = lane >> 4;
    for (int i = 0; i < 16; ++i) {
      const int tile = MODE == kTileRowNt ? (wave * 4 + (i >> 2) * 16 + (i & 3)) : (wave + 4 * i);  // row-major, 4 tiles across
      const int ty = tile >> 2, tx = tile & 3;
      const size_t pix = img_px + (size_t)(Y0 + ty * 4 + ly) * W + X0 + tx * 16 + lx;
      put<NT>(&ids[pix], (int32_t)(tile + lane));
      put<NT>(&z[pix], fv);
      put<NT>((v3f *)(bary + 3 * pix), v3f{fv, fv + 1.0f, fv + 2.0f});
    }
  } else if constexpr (MODE == kTile32Raster) {
    typedef unsigned v3u __attribute__((ext_vector_type(3)));
    constexpr int kRsrcWord3 = 0x00020000;
    const size_t region_pix = img_px + (size_t)Y0 * W + X0;
    const __amdgpu_buffer_rsrc_t rs_ids = __builtin_amdgcn_make_buffer_rsrc(ids + region_pix, 0, 0x7fffffff, kRsrcWord3);
    const __amdgpu_buffer_rsrc_t rs_z = __builtin_amdgcn_make_buffer_rsrc(z + region_pix, 0, 0x7fffffff, kRsrcWord3);
    const __amdgpu_buffer_rsrc_t rs_bary = __builtin_amdgcn_make_buffer_rsrc(bary + 3 * region_pix, 0, 0x7fffffff, kRsrcWord3);
    const int lx = lane & 31, ly = lane >> 5;
    const unsigned lane_pix = (unsigned)(ly * W + lx);
    for (int i = 0; i < 16; ++i) {   // 2 x 32 tiles of 32 x 2 pixels per region: wavefront w walks tiles w, w + 4, ... (a 32-pixel-wide column pair)
      const int tile = wave + 4 * i, ty = tile >> 1, tx = tile & 1;
      const int tile_pix = ty * 2 * W + tx * 32;
      __builtin_amdgcn_raw_buffer_store_b32((unsigned)(tile + lane), rs_ids, lane_pix * 4u, tile_pix * 4, 0);
      __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, fv), rs_z, lane_pix * 4u, tile_pix * 4, 2);
      __builtin_amdgcn_raw_buffer_store_b96(v3u{__builtin_bit_cast(unsigned, fv), __builtin_bit_cast(unsigned, fv + 1.0f),
                                                __builtin_bit_cast(unsigned, fv + 2.0f)}, rs_bary, lane_pix * 12u, tile_pix * 12, 2);
      asm volatile("s_nop 1" ::: "memory");
    }
  } else if constexpr (MODE == kTileRaster || MODE == kTileRasterNoZ || MODE == kTileRasterCached) {
    typedef unsigned v3u __attribute__((ext_vector_type(3)));
    constexpr int kRsrcWord3 = 0x00020000;  // raw 32-bit buffer on gfx9-family targets (raster_forward.hip)
    const size_t region_pix = img_px + (size_t)Y0 * W + X0;
    const __amdgpu_buffer_rsrc_t rs_ids = __builtin_amdgcn_make_buffer_rsrc(ids + region_pix, 0, 0x7fffffff, kRsrcWord3);
    const __amdgpu_buffer_rsrc_t rs_z = __builtin_amdgcn_make_buffer_rsrc(z + region_pix, 0, 0x7fffffff, kRsrcWord3);
    const __amdgpu_buffer_rsrc_t rs_bary = __builtin_amdgcn_make_buffer_rsrc(bary + 3 * region_pix, 0, 0x7fffffff, kRsrcWord3);
    const int lx = lane & 15, ly = lane >> 4;
    const unsigned lane_pix = (unsigned)(ly * W + lx);
    for (int i = 0; i < 16; ++i) {
      const int tile = wave + 4 * i, ty = tile >> 2, tx = tile & 3;
      const int tile_pix = ty * 4 * W + tx * 16;   // wave-uniform: the stores' soffset
      __builtin_amdgcn_raw_buffer_store_b32((unsigned)(tile + lane), rs_ids, lane_pix * 4u, tile_pix * 4, 0);
      constexpr int kAux = MODE == kTileRasterCached ? 0 : 2;
      if (MODE != kTileRasterNoZ) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, fv), rs_z, lane_pix * 4u, tile_pix * 4, kAux);
      __builtin_amdgcn_raw_buffer_store_b96(v3u{__builtin_bit_cast(unsigned, fv), __builtin_bit_cast(unsigned, fv + 1.0f),
                                                __builtin_bit_cast(unsigned, fv + 2.0f)}, rs_bary, lane_pix * 12u, tile_pix * 12, kAux);
      asm volatile("s_nop 1" ::: "memory");   // (raster_forward.hip's store_b96_soffset: the wide-store hazard guard)
    }
  } else if constexpr (MODE == kRowNt || MODE == kRowCached || MODE == kRowX4Nt) {
    constexpr bool NT = MODE != kRowCached;
    for (int i = 0; i < 16; ++i) {
      const int y = Y0 + wave + 4 * i;
      const size_t pix = img_px + (size_t)y * W + X0 + lane;
      put<NT>(&ids[pix], (int32_t)(i + lane));
      put<NT>(&z[pix], fv);
      if constexpr (MODE == kRowX4Nt) {
        if (lane < 48) put<NT>((v4f *)(bary + 3 * (pix - lane)) + lane, v4f{fv, fv + 1.0f, fv + 2.0f, fv});
      } else {
        put<NT>((v3f *)(bary + 3 * pix), v3f{fv, fv + 1.0f, fv + 2.0f});
      }
    }
  } else if constexpr (MODE == kRow4Nt) {
    for (int i = 0; i < 4; ++i) {
      const int y0 = Y0 + (wave + 4 * i) * 4;
#pragma unroll
      for (int r = 0; r < 4; ++r) put<true>(&ids[img_px + (size_t)(y0 + r) * W + X0 + lane], (int32_t)(i + lane));
#pragma unroll
      for (int r = 0; r < 4; ++r) put<true>(&z[img_px + (size_t)(y0 + r) * W + X0 + lane], fv);
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (lane < 48) put<true>((v4f *)(bary + 3 * (img_px + (size_t)(y0 + r) * W + X0)) + lane, v4f{fv, fv + 1.0f, fv + 2.0f, fv});
    }
  } else {
    constexpr bool NT = MODE == kLinearNt;
    // the region's share of every plane as one contiguous chunk: 4096 px = 16 KB of ids / z, 48 KB of barycentrics
    const size_t first = (size_t)region * 4096;
    for (int i = 0; i < 4; ++i) {
      const size_t q = first / 4 + (size_t)i * 256 + threadIdx.x;  // float4 index
      put<NT>((v4f *)ids + q, v4f{fv, fv, fv, fv});
      put<NT>((v4f *)z + q, v4f{fv, fv, fv, fv});
    }
    for (int i = 0; i < 12; ++i) put<NT>((v4f *)bary + first * 3 / 4 + (size_t)i * 256 + threadIdx.x, v4f{fv, fv, fv, fv});
  }
}

template <int MODE>
static void run(int32_t *ids, float *z, float *bary, int B, int W, int H) {
  const int regions_x = W / 64, per_image = regions_x * (H / 64);
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k_store<MODE>, dim3(per_image * B), dim3(256), 0, 0, ids, z, bary, W, H, regions_x, per_image);
  CHECK(hipDeviceSynchronize());
  const int reps = 20;
  float best = 1e30f, total = 0.f;
  for (int i = 0; i < reps; ++i) {
    CHECK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(k_store<MODE>, dim3(per_image * B), dim3(256), 0, 0, ids, z, bary, W, H, regions_x, per_image);
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    best = ms < best ? ms : best;
    total += ms;
  }
  const double bytes = (MODE == kTileRasterNoZ ? 16.0 : 20.0) * B * W * H;
  printf("{\"pattern\": \"%s\", \"bytes\": %.0f, \"ms_mean\": %.4f, \"ms_best\": %.4f, \"TBps_mean\": %.3f, \"TBps_best\": %.3f}\n",
         kNames[MODE], bytes, total / reps, best, bytes / (total / reps * 1e-3) / 1e12, bytes / (best * 1e-3) / 1e12);
}

int main(int argc, char **argv) {
  // default: configs[2]'s planes (32 x 1024^2); `store_patterns 8 2048 2048`: configs[3]'s per-GPU share, same pixels
  const int B = argc > 3 ? atoi(argv[1]) : 32, W = argc > 3 ? atoi(argv[2]) : 1024, H = argc > 3 ? atoi(argv[3]) : 1024;
  const size_t px = (size_t)B * W * H;
  int32_t *ids;
  float *z, *bary;
  CHECK(hipMalloc(&ids, px * 4));
  CHECK(hipMalloc(&z, px * 4));
  CHECK(hipMalloc(&bary, px * 12));
  run<kTileNt>(ids, z, bary, B, W, H);
  run<kTileCached>(ids, z, bary, B, W, H);
  run<kTileRowNt>(ids, z, bary, B, W, H);
  run<kRowNt>(ids, z, bary, B, W, H);
  run<kRowX4Nt>(ids, z, bary, B, W, H);
  run<kRowCached>(ids, z, bary, B, W, H);
  run<kRow4Nt>(ids, z, bary, B, W, H);
  run<kLinearNt>(ids, z, bary, B, W, H);
  run<kLinearCached>(ids, z, bary, B, W, H);
  run<kTileRaster>(ids, z, bary, B, W, H);
  run<kTileRasterNoZ>(ids, z, bary, B, W, H);
  run<kTile32Raster>(ids, z, bary, B, W, H);
  run<kTileRasterCached>(ids, z, bary, B, W, H);
  return 0;
}
