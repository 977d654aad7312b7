// Forward G-buffer rasterizer for gfx950 (MI355X).
//
// Replaces rasterize_triangles_forward
// (reference: src/mesh_renderer/kernels/rasterize_triangles.cpp:302-419 and its
// helpers :19-98) with three kernels on one stream:
//
//   k_setup   one thread per (image, triangle): sign-corrected adjugate, clip z/w,
//             pixel bbox (binary64 projection as in cpp:361-366), packed into a
//             64-byte record; plus the binary64 pixel-centre tables (cpp:376-377).
//   k_raster  one 256-thread workgroup per 64x64-pixel region.  The workgroup
//             compacts, IN TRIANGLE-ID ORDER, the triangles whose bbox touches the
//             region into an LDS bin; each of its 4 wavefronts then walks 8x8 (or
//             16x4 / 32x2) pixel tiles, one pixel per lane: the tile's candidates
//             are picked from the LDS bin by a 64-wide bbox test + ballot, their
//             records arrive through wave-uniform scalar loads (SGPR operands), and
//             every lane runs the reference's exact edge / barycentric / z test.
//             Walking candidates in ascending id reproduces the reference's
//             sequential z-buffer semantics (ties -> later id, NaN handling)
//             without any ordering trick.  Each pixel is written exactly once:
//             (id, z, b0, b1, b2), 20 B/px, whole 128-B lines per workgroup.
//
// Exactness: this file is compiled with -ffp-contract=off; every float expression
// below is written in the reference's association order.
// fp32 '/' lowers to the IEEE-correct v_div_scale/v_div_fmas/v_div_fixup sequence.
#include "mr_internal.h"

namespace mr {

thread_local int g_last_hip_error = 0;

namespace {

constexpr int kRegionW = 64;   // pixels per workgroup region
constexpr int kRegionH = 64;
constexpr int kBinCap = 1024;  // LDS bin capacity (triangles per pass)
constexpr int kThreads = 256;

// x86 cvttss2si semantics for the reference's static_cast<int> (cpp:21,29):
// out-of-range and NaN convert to INT_MIN, which then clamps to 0.
__device__ __forceinline__ int cvt_trunc_x86(float f) {
  return (f >= -2147483648.0f && f < 2147483648.0f) ? (int)f : INT_MIN;
}
__device__ __forceinline__ int clampi(int v, int lo, int hi) {
  const int a = v > lo ? v : lo;
  return a < hi ? a : hi;
}
// std::max / std::min argument order matters for NaN (cpp:19-31).
__device__ __forceinline__ float max_std(float a, float b) { return (a < b) ? b : a; }
__device__ __forceinline__ float min_std(float a, float b) { return (b < a) ? b : a; }

__global__ __launch_bounds__(kThreads) void k_setup(
    const float4 *__restrict__ clip, const int32_t *__restrict__ tris, int B, int V, int T,
    int W, int H, TriRec *__restrict__ recs, uint2 *__restrict__ bbs,
    float *__restrict__ pxtab, float *__restrict__ pytab) {
  const long gid = (long)blockIdx.x * kThreads + threadIdx.x;
  const long nbt = (long)B * T;
  const float hw = (float)(0.5 * (double)W);  // cpp:309
  const float hh = (float)(0.5 * (double)H);  // cpp:310
  if (gid >= nbt) {
    // pixel-centre tables: binary64 expression, one rounding (cpp:376-377)
    const long k = gid - nbt;
    if (k < W) {
      pxtab[k] = (float)(((double)k + 0.5) / (double)hw - 1.0);
    } else if (k < (long)W + H) {
      const long r = k - W;
      pytab[r] = (float)(((double)r + 0.5) / (double)hh - 1.0);
    }
    return;
  }
  const int b = (int)(gid / T);
  const int t = (int)(gid - (long)b * T);
  const int i0 = tris[3 * t + 0], i1 = tris[3 * t + 1], i2 = tris[3 * t + 2];
  uint2 bb = make_uint2(0u, 0u);
  if ((unsigned)i0 < (unsigned)V && (unsigned)i1 < (unsigned)V && (unsigned)i2 < (unsigned)V) {
    const float4 p0 = clip[(long)b * V + i0];
    const float4 p1 = clip[(long)b * V + i1];
    const float4 p2 = clip[(long)b * V + i2];
    const float w0 = p0.w, w1 = p1.w, w2 = p2.w;
    if (!(w0 < 0 && w1 < 0 && w2 < 0)) {  // cpp:339
      // rows of M: x, y, w; columns: the three corners (cpp:350-353)
      const float a11 = p0.x, a12 = p1.x, a13 = p2.x;
      const float a21 = p0.y, a22 = p1.y, a23 = p2.y;
      const float a31 = w0, a32 = w1, a33 = w2;
      float m0 = a22 * a33 - a32 * a23;
      float m1 = a13 * a32 - a33 * a12;
      float m2 = a12 * a23 - a22 * a13;
      float m3 = a23 * a31 - a33 * a21;
      float m4 = a11 * a33 - a31 * a13;
      float m5 = a13 * a21 - a23 * a11;
      float m6 = a21 * a32 - a31 * a22;
      float m7 = a12 * a31 - a32 * a11;
      float m8 = a11 * a22 - a21 * a12;
      const float det = a11 * m0 + a12 * m3 + a13 * m6;  // cpp:77
      if (det < 0.0f) {
        m0 = -m0; m1 = -m1; m2 = -m2; m3 = -m3; m4 = -m4;
        m5 = -m5; m6 = -m6; m7 = -m7; m8 = -m8;
      }
      int l = 0, r = W, bot = 0, top = H;  // cpp:356
      if (w0 > 0 && w1 > 0 && w2 > 0) {    // cpp:360
        // float divide, then binary64 add and multiply, rounded once (cpp:361-366)
        const float x0 = (float)(((double)(p0.x / w0) + 1.0) * (double)hw);
        const float x1 = (float)(((double)(p1.x / w1) + 1.0) * (double)hw);
        const float x2 = (float)(((double)(p2.x / w2) + 1.0) * (double)hw);
        const float y0 = (float)(((double)(p0.y / w0) + 1.0) * (double)hh);
        const float y1 = (float)(((double)(p1.y / w1) + 1.0) * (double)hh);
        const float y2 = (float)(((double)(p2.y / w2) + 1.0) * (double)hh);
        l = clampi(cvt_trunc_x86(floorf(min_std(min_std(x0, x1), x2))), 0, W);
        r = clampi(cvt_trunc_x86(ceilf(max_std(max_std(x0, x1), x2))), 0, W);
        bot = clampi(cvt_trunc_x86(floorf(min_std(min_std(y0, y1), y2))), 0, H);
        top = clampi(cvt_trunc_x86(ceilf(max_std(max_std(y0, y1), y2))), 0, H);
      }
      if (r > l && top > bot) {
        bb = pack_bbox(l, r, bot, top);
        TriRec rec;
        rec.a = make_float4(m0, m1, m2, m3);
        rec.b = make_float4(m4, m5, m6, m7);
        rec.c = make_float4(m8, p0.z, p1.z, p2.z);
        rec.d = make_float4(w0, w1, w2, 0.0f);
        recs[gid] = rec;
      }
    }
  }
  bbs[gid] = bb;
}

// Per-pixel running z-buffer state (registers).
struct PixelState {
  float z, b0, b1, b2;
  int id;
};

// One candidate triangle against one pixel: exactly the body of cpp:376-409.
// m*, z*, w* are wave-uniform (SGPR) values; px, py, st are per lane.
__device__ __forceinline__ void shade_candidate(
    const float m0, const float m1, const float m2, const float m3, const float m4,
    const float m5, const float m6, const float m7, const float m8, const float z0,
    const float z1, const float z2, const float w0, const float w1, const float w2,
    const int tri, const bool in_bbox, const float px, const float py, PixelState &st) {
  const float e0 = (m0 * px + m1 * py) + m2;  // cpp:46
  const float e1 = (m3 * px + m4 * py) + m5;
  const float e2 = (m6 * px + m7 * py) + m8;
  const float s = (e0 + e1) + e2;  // cpp:384
  // cpp:96-97.  With all three >= 0 (hence no NaN), "some edge > 0" is the same
  // predicate as s > 0: a sum of non-negative floats is zero only if all are.
  const bool inside = in_bbox && (e0 >= 0.0f) && (e1 >= 0.0f) && (e2 >= 0.0f) && (s > 0.0f);
  if (inside) {
    const float b0 = e0 / s, b1 = e1 / s, b2 = e2 / s;  // cpp:385-387
    const float cz = (b0 * z0 + b1 * z1) + b2 * z2;     // cpp:395
    const float cw = (b0 * w0 + b1 * w1) + b2 * w2;     // cpp:396
    const float zz = cz / cw;                           // cpp:397
    if (!(zz < -1.0f || zz > 1.0f || zz > st.z)) {      // cpp:401
      st.z = zz;
      st.id = tri;
      st.b0 = b0;
      st.b1 = b1;
      st.b2 = b2;
    }
  }
}

// TW x TH = 64: pixel tile walked by one wavefront, one pixel per lane.
template <int TW, int TH>
__global__ __launch_bounds__(kThreads) void k_raster(
    const TriRec *__restrict__ recs, const uint2 *__restrict__ bbs,
    const float *__restrict__ pxtab, const float *__restrict__ pytab, int T, int W, int H,
    int regions_x, int regions_per_image, int n_regions, int regions_per_xcd,
    int32_t *__restrict__ ids, float *__restrict__ bary, float *__restrict__ zbuf) {
  static_assert(TW * TH == kWave, "one pixel per lane");
  static_assert(kRegionW % TW == 0 && kRegionH % TH == 0, "tiles must pave the region");
  __shared__ int s_tri[kBinCap];
  __shared__ uint2 s_bb[kBinCap];
  __shared__ int s_wave_count[kThreads / kWave];

  const int region = xcd_contiguous_block((int)blockIdx.x, n_regions, regions_per_xcd);
  if (region < 0) return;  // padding block (whole workgroup)
  const int img = region / regions_per_image;
  const int rr = region - img * regions_per_image;
  const int ry = rr / regions_x;
  const int rx = rr - ry * regions_x;
  const int X0 = rx * kRegionW, Y0 = ry * kRegionH;
  const int X1 = min(X0 + kRegionW, W), Y1 = min(Y0 + kRegionH, H);

  const int tid = (int)threadIdx.x;
  const int lane = tid & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const TriRec *img_recs = recs + (size_t)img * T;
  const uint2 *img_bbs = bbs + (size_t)img * T;
  const size_t img_px = (size_t)img * H * W;

  int n_bin = 0;        // workgroup-uniform
  bool first_pass = true;

  // Rasterize every tile of the region against the current LDS bin.
  auto raster_pass = [&](const int n, const bool fresh) {
    constexpr int kTilesX = kRegionW / TW, kTilesY = kRegionH / TH;
    for (int tile = wave; tile < kTilesX * kTilesY; tile += kThreads / kWave) {
      const int ty = tile / kTilesX, tx = tile - ty * kTilesX;
      const int x0 = X0 + tx * TW, y0 = Y0 + ty * TH;
      if (x0 >= X1 || y0 >= Y1) continue;  // wave-uniform
      const int x1 = min(x0 + TW, X1), y1 = min(y0 + TH, Y1);
      const int ix = x0 + (lane % TW), iy = y0 + (lane / TW);
      const bool in_image = ix < W && iy < H;
      const size_t pix = img_px + (size_t)iy * W + ix;
      const float px = pxtab[min(ix, W - 1)];
      const float py = pytab[min(iy, H - 1)];
      PixelState st;
      if (fresh) {
        st.z = 1.0f; st.b0 = 0.0f; st.b1 = 0.0f; st.b2 = 0.0f; st.id = 0;  // cpp:313-321
      } else if (in_image) {
        // bin overflowed earlier: resume from what this very lane stored
        st.z = zbuf[pix]; st.id = ids[pix];
        st.b0 = bary[3 * pix]; st.b1 = bary[3 * pix + 1]; st.b2 = bary[3 * pix + 2];
      }
      for (int base = 0; base < n; base += kWave) {
        const int k = base + lane;
        bool hit = false;
        int my_tri = 0;
        uint2 my_bb = make_uint2(0u, 0u);
        if (k < n) {
          my_tri = s_tri[k];
          my_bb = s_bb[k];
          const int l = (int)(my_bb.x & 0xffffu), r = (int)(my_bb.x >> 16);
          const int bt = (int)(my_bb.y & 0xffffu), tp = (int)(my_bb.y >> 16);
          hit = (l < x1) && (r > x0) && (bt < y1) && (tp > y0);
        }
        unsigned long long todo = __ballot(hit);
        while (todo) {  // ascending lane == ascending triangle id
          const int j = __builtin_ctzll(todo);
          todo &= todo - 1;
          const int tri = __builtin_amdgcn_readlane(my_tri, j);
          const unsigned bbx = (unsigned)__builtin_amdgcn_readlane((int)my_bb.x, j);
          const unsigned bby = (unsigned)__builtin_amdgcn_readlane((int)my_bb.y, j);
          const int l = (int)(bbx & 0xffffu), wdt = (int)(bbx >> 16) - l;
          const int bt = (int)(bby & 0xffffu), hgt = (int)(bby >> 16) - bt;
          const bool in_bbox =
              ((unsigned)(ix - l) < (unsigned)wdt) && ((unsigned)(iy - bt) < (unsigned)hgt);
          const TriRec *rp = img_recs + tri;  // wave-uniform address -> scalar loads
          const float4 ra = rp->a, rb = rp->b, rc = rp->c, rd = rp->d;
          shade_candidate(ra.x, ra.y, ra.z, ra.w, rb.x, rb.y, rb.z, rb.w, rc.x, rc.y, rc.z,
                          rc.w, rd.x, rd.y, rd.z, tri, in_bbox, px, py, st);
        }
      }
      if (in_image) {
        ids[pix] = st.id;
        zbuf[pix] = st.z;
        bary[3 * pix + 0] = st.b0;
        bary[3 * pix + 1] = st.b1;
        bary[3 * pix + 2] = st.b2;
      }
    }
  };

  // Bin the image's triangles against this region, 256 at a time, keeping id order.
  for (int base = 0; base < T; base += kThreads) {
    const int t = base + tid;
    bool hit = false;
    uint2 bb = make_uint2(0u, 0u);
    if (t < T) {
      bb = img_bbs[t];
      const int l = (int)(bb.x & 0xffffu), r = (int)(bb.x >> 16);
      const int bt = (int)(bb.y & 0xffffu), tp = (int)(bb.y >> 16);
      hit = (l < X1) && (r > X0) && (bt < Y1) && (tp > Y0);  // empty bbox = all zeros
    }
    const unsigned long long m = __ballot(hit);
    if (lane == 0) s_wave_count[wave] = __builtin_popcountll(m);
    __syncthreads();
    int offset = n_bin, total = 0;
#pragma unroll
    for (int w = 0; w < kThreads / kWave; ++w) {
      const int c = s_wave_count[w];
      if (w < wave) offset += c;
      total += c;
    }
    if (hit) {
      const int pos = offset + (int)__builtin_amdgcn_mbcnt_hi(
                                   (unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
      s_tri[pos] = t;
      s_bb[pos] = bb;
    }
    n_bin += total;
    __syncthreads();
    if (n_bin + kThreads > kBinCap && base + kThreads < T) {
      // bin (nearly) full with triangles still to come: flush it
      raster_pass(n_bin, first_pass);
      first_pass = false;
      n_bin = 0;
      __syncthreads();  // tiles done reading the bin; also orders the state stores
    }
  }
  raster_pass(n_bin, first_pass);
}

}  // namespace

size_t raster_forward_ws(int B, int V, int T, int W, int H) {
  (void)V;
  const size_t nbt = (size_t)B * T;
  return align_up(nbt * sizeof(TriRec), 256) + align_up(nbt * sizeof(uint2), 256) +
         align_up((size_t)W * sizeof(float), 256) + align_up((size_t)H * sizeof(float), 256);
}

int g_raster_tile_shape = 0;  // 0: 8x8, 1: 16x4, 2: 32x2 (mr_set_raster_tile_shape)
hipEvent_t g_raster_ev_start = nullptr, g_raster_ev_stop = nullptr;  // mr_set_raster_profile_events

int launch_raster_forward(const float *clip, const int32_t *tris, int B, int V, int T, int W,
                          int H, int32_t *ids, float *bary, float *z, void *ws, hipStream_t s) {
  const size_t nbt = (size_t)B * T;
  char *p = (char *)ws;
  TriRec *recs = (TriRec *)p;
  p += align_up(nbt * sizeof(TriRec), 256);
  uint2 *bbs = (uint2 *)p;
  p += align_up(nbt * sizeof(uint2), 256);
  float *pxtab = (float *)p;
  p += align_up((size_t)W * sizeof(float), 256);
  float *pytab = (float *)p;

  const long setup_threads = (long)nbt + W + H;
  const unsigned setup_blocks = (unsigned)((setup_threads + kThreads - 1) / kThreads);
  hipLaunchKernelGGL(k_setup, dim3(setup_blocks), dim3(kThreads), 0, s, (const float4 *)clip, tris,
                     B, V, T, W, H, recs, bbs, pxtab, pytab);
  int rc = check_launch();
  if (rc != MR_OK) return rc;
  if (B == 0) return MR_OK;

  const int regions_x = (W + kRegionW - 1) / kRegionW;
  const int regions_y = (H + kRegionH - 1) / kRegionH;
  const int per_image = regions_x * regions_y;
  const int n_regions = per_image * B;
  const int per_xcd = (n_regions + kXcds - 1) / kXcds;
  const dim3 grid((unsigned)(per_xcd * kXcds)), block(kThreads);
#define MR_LAUNCH_RASTER(TW, TH)                                                              \
  hipLaunchKernelGGL((k_raster<TW, TH>), grid, block, 0, s, recs, bbs, pxtab, pytab, T, W, H, \
                     regions_x, per_image, n_regions, per_xcd, ids, bary, z)
  if (g_raster_ev_start) (void)hipEventRecord(g_raster_ev_start, s);
  switch (g_raster_tile_shape) {
    case 1: MR_LAUNCH_RASTER(16, 4); break;
    case 2: MR_LAUNCH_RASTER(32, 2); break;
    default: MR_LAUNCH_RASTER(8, 8); break;
  }
#undef MR_LAUNCH_RASTER
  if (g_raster_ev_stop) (void)hipEventRecord(g_raster_ev_stop, s);
  return check_launch();
}

}  // namespace mr
