// Instruction-issue micro-benchmarks for gfx950: wave-instructions per clock per CU for the
// instruction classes k_raster is made of, at 1..8 waves per SIMD.  Standalone:
//   hipcc --offload-arch=gfx950 -O2 -o issue_rates issue_rates.hip && ./issue_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

constexpr int kIters = 20000;

// each body = 16 instructions on 8 independent register sets
#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

template <int KIND>
__global__ __launch_bounds__(256) void k_bench(float *out, unsigned long long *cycles, int iters) {
  float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  float b0 = 1.0001f, b1 = 0.9999f;
  typedef float v2f __attribute__((ext_vector_type(2)));
  v2f p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, pb = {b0, b1};
  __shared__ float s_buf[2048];
  s_buf[threadIdx.x] = a0; s_buf[threadIdx.x + 256] = a1; s_buf[threadIdx.x + 512] = a2; s_buf[threadIdx.x + 768] = a3;
  s_buf[threadIdx.x + 1024] = a4; s_buf[threadIdx.x + 1280] = a5; s_buf[threadIdx.x + 1536] = a6; s_buf[threadIdx.x + 1792] = a7;
  __syncthreads();
  int s0 = blockIdx.x, s1 = 3, s2 = 5, s3 = 7;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    if constexpr (KIND == 0) {  // v_fma_f32 x16
      asm volatile(
          "v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
          "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
          "v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
          "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
          : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0), "v"(b1));
    } else if constexpr (KIND == 1) {  // v_mul_f32 / v_add_f32 alternating x16
      asm volatile(
          "v_mul_f32 %0, %0, %8\n v_add_f32 %1, %1, %9\n v_mul_f32 %2, %2, %8\n v_add_f32 %3, %3, %9\n"
          "v_mul_f32 %4, %4, %8\n v_add_f32 %5, %5, %9\n v_mul_f32 %6, %6, %8\n v_add_f32 %7, %7, %9\n"
          "v_mul_f32 %0, %0, %8\n v_add_f32 %1, %1, %9\n v_mul_f32 %2, %2, %8\n v_add_f32 %3, %3, %9\n"
          "v_mul_f32 %4, %4, %8\n v_add_f32 %5, %5, %9\n v_mul_f32 %6, %6, %8\n v_add_f32 %7, %7, %9\n"
          : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0), "v"(b1));
    } else if constexpr (KIND == 2) {  // v_pk_fma_f32 x16 (4 register pairs)
      asm volatile(
          "v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
          "v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
          "v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
          "v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
          : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pb), "v"(pb));
    } else if constexpr (KIND == 3) {  // v_pk_mul_f32 / v_pk_add_f32 x16
      asm volatile(
          "v_pk_mul_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %5\n v_pk_mul_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %5\n"
          "v_pk_mul_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %5\n v_pk_mul_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %5\n"
          "v_pk_mul_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %5\n v_pk_mul_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %5\n"
          "v_pk_mul_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %5\n v_pk_mul_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %5\n"
          : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pb), "v"(pb));
    } else if constexpr (KIND == 4) {  // SALU x16
      asm volatile(
          "s_add_u32 %0, %0, %1\n s_xor_b32 %1, %1, %2\n s_add_u32 %2, %2, %3\n s_lshl_b32 %3, %3, 1\n"
          "s_add_u32 %0, %0, %1\n s_xor_b32 %1, %1, %2\n s_add_u32 %2, %2, %3\n s_and_b32 %3, %3, %0\n"
          "s_add_u32 %0, %0, %1\n s_xor_b32 %1, %1, %2\n s_add_u32 %2, %2, %3\n s_lshl_b32 %3, %3, 1\n"
          "s_add_u32 %0, %0, %1\n s_xor_b32 %1, %1, %2\n s_add_u32 %2, %2, %3\n s_and_b32 %3, %3, %0\n"
          : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) :: "scc");
    } else if constexpr (KIND == 5) {  // 16 VALU + 16 SALU interleaved
      asm volatile(
          "v_fma_f32 %0, %0, %12, %13\n s_add_u32 %8, %8, %9\n v_fma_f32 %1, %1, %12, %13\n s_xor_b32 %9, %9, %10\n"
          "v_fma_f32 %2, %2, %12, %13\n s_add_u32 %10, %10, %11\n v_fma_f32 %3, %3, %12, %13\n s_lshl_b32 %11, %11, 1\n"
          "v_fma_f32 %4, %4, %12, %13\n s_add_u32 %8, %8, %9\n v_fma_f32 %5, %5, %12, %13\n s_xor_b32 %9, %9, %10\n"
          "v_fma_f32 %6, %6, %12, %13\n s_add_u32 %10, %10, %11\n v_fma_f32 %7, %7, %12, %13\n s_and_b32 %11, %11, %8\n"
          "v_fma_f32 %0, %0, %12, %13\n s_add_u32 %8, %8, %9\n v_fma_f32 %1, %1, %12, %13\n s_xor_b32 %9, %9, %10\n"
          "v_fma_f32 %2, %2, %12, %13\n s_add_u32 %10, %10, %11\n v_fma_f32 %3, %3, %12, %13\n s_lshl_b32 %11, %11, 1\n"
          "v_fma_f32 %4, %4, %12, %13\n s_add_u32 %8, %8, %9\n v_fma_f32 %5, %5, %12, %13\n s_xor_b32 %9, %9, %10\n"
          "v_fma_f32 %6, %6, %12, %13\n s_add_u32 %10, %10, %11\n v_fma_f32 %7, %7, %12, %13\n s_and_b32 %11, %11, %8\n"
          : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7),
            "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3)
          : "v"(b0), "v"(b1) : "scc");
    } else if constexpr (KIND == 6) {  // v_rcp_f32 x16
      asm volatile(
          "v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n"
          "v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7\n"
          "v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n"
          "v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7\n"
          : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    } else if constexpr (KIND == 7) {  // v_div_scale / v_div_fmas / v_div_fixup mix x16 (approx. cost class)
      asm volatile(
          "v_div_scale_f32 %0, vcc, %0, %8, %0\n v_div_fixup_f32 %1, %1, %8, %9\n v_div_scale_f32 %2, vcc, %2, %8, %2\n v_div_fixup_f32 %3, %3, %8, %9\n"
          "v_div_scale_f32 %4, vcc, %4, %8, %4\n v_div_fixup_f32 %5, %5, %8, %9\n v_div_scale_f32 %6, vcc, %6, %8, %6\n v_div_fixup_f32 %7, %7, %8, %9\n"
          "v_div_scale_f32 %0, vcc, %0, %8, %0\n v_div_fixup_f32 %1, %1, %8, %9\n v_div_scale_f32 %2, vcc, %2, %8, %2\n v_div_fixup_f32 %3, %3, %8, %9\n"
          "v_div_scale_f32 %4, vcc, %4, %8, %4\n v_div_fixup_f32 %5, %5, %8, %9\n v_div_scale_f32 %6, vcc, %6, %8, %6\n v_div_fixup_f32 %7, %7, %8, %9\n"
          : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0), "v"(b1) : "vcc");
    } else if constexpr (KIND == 8) {  // v_cmp + v_cndmask x16
      asm volatile(
          "v_cmp_ge_f32 vcc, %0, %8\n v_cndmask_b32 %1, %1, %9, vcc\n v_cmp_ge_f32 vcc, %2, %8\n v_cndmask_b32 %3, %3, %9, vcc\n"
          "v_cmp_ge_f32 vcc, %4, %8\n v_cndmask_b32 %5, %5, %9, vcc\n v_cmp_ge_f32 vcc, %6, %8\n v_cndmask_b32 %7, %7, %9, vcc\n"
          "v_cmp_ge_f32 vcc, %0, %8\n v_cndmask_b32 %1, %1, %9, vcc\n v_cmp_ge_f32 vcc, %2, %8\n v_cndmask_b32 %3, %3, %9, vcc\n"
          "v_cmp_ge_f32 vcc, %4, %8\n v_cndmask_b32 %5, %5, %9, vcc\n v_cmp_ge_f32 vcc, %6, %8\n v_cndmask_b32 %7, %7, %9, vcc\n"
          : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0), "v"(b1) : "vcc");
    } else if constexpr (KIND == 9) {  // ds_read_b128, wave-uniform address (broadcast) x16
      float4 r0, r1, r2, r3;
      unsigned addr = (unsigned)(it & 15) * 16u;
      asm volatile(
          "ds_read_b128 %0, %4\n ds_read_b128 %1, %4 offset:16\n ds_read_b128 %2, %4 offset:32\n ds_read_b128 %3, %4 offset:48\n"
          "s_waitcnt lgkmcnt(0)\n"
          "ds_read_b128 %0, %4 offset:64\n ds_read_b128 %1, %4 offset:80\n ds_read_b128 %2, %4 offset:96\n ds_read_b128 %3, %4 offset:112\n"
          "s_waitcnt lgkmcnt(0)\n"
          "ds_read_b128 %0, %4 offset:128\n ds_read_b128 %1, %4 offset:144\n ds_read_b128 %2, %4 offset:160\n ds_read_b128 %3, %4 offset:176\n"
          "s_waitcnt lgkmcnt(0)\n"
          "ds_read_b128 %0, %4 offset:192\n ds_read_b128 %1, %4 offset:208\n ds_read_b128 %2, %4 offset:224\n ds_read_b128 %3, %4 offset:240\n"
          "s_waitcnt lgkmcnt(0)\n"
          : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3) : "v"(addr) : "memory");
      a0 += r0.x + r1.y + r2.z + r3.w;
    } else if constexpr (KIND == 10) {  // ds_read_b128 per-lane addresses, 80-B stride (k_raster's depth-stage read)
      float4 r0, r1, r2, r3;
      unsigned addr = ((threadIdx.x * 7u + it) & 63u) * 80u;
      asm volatile(
          "ds_read_b128 %0, %4\n ds_read_b128 %1, %4 offset:16\n ds_read_b128 %2, %4 offset:32\n ds_read_b128 %3, %4 offset:48\n"
          "s_waitcnt lgkmcnt(0)\n"
          "ds_read_b128 %0, %4 offset:64\n ds_read_b128 %1, %4 offset:80\n ds_read_b128 %2, %4 offset:96\n ds_read_b128 %3, %4 offset:112\n"
          "s_waitcnt lgkmcnt(0)\n"
          "ds_read_b128 %0, %4 offset:128\n ds_read_b128 %1, %4 offset:144\n ds_read_b128 %2, %4 offset:160\n ds_read_b128 %3, %4 offset:176\n"
          "s_waitcnt lgkmcnt(0)\n"
          "ds_read_b128 %0, %4 offset:192\n ds_read_b128 %1, %4 offset:208\n ds_read_b128 %2, %4 offset:224\n ds_read_b128 %3, %4 offset:240\n"
          "s_waitcnt lgkmcnt(0)\n"
          : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3) : "v"(addr & 0xfffu) : "memory");
      a0 += r0.x + r1.y + r2.z + r3.w;
    } else if constexpr (KIND == 11) {  // dependent chain: v_fma_f32 x16 on ONE accumulator (latency)
      asm volatile(
          "v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n"
          "v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n"
          "v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n"
          "v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n"
          : "+v"(a0) : "v"(b0), "v"(b1));
    } else if constexpr (KIND == 12) {  // integer VALU: v_and_b32 / v_or_b32 / v_lshlrev x16
      asm volatile(
          "v_and_b32 %0, %0, %8\n v_or_b32 %1, %1, %9\n v_add_u32 %2, %2, %8\n v_xor_b32 %3, %3, %9\n"
          "v_and_b32 %4, %4, %8\n v_or_b32 %5, %5, %9\n v_add_u32 %6, %6, %8\n v_xor_b32 %7, %7, %9\n"
          "v_and_b32 %0, %0, %8\n v_or_b32 %1, %1, %9\n v_add_u32 %2, %2, %8\n v_xor_b32 %3, %3, %9\n"
          "v_and_b32 %4, %4, %8\n v_or_b32 %5, %5, %9\n v_add_u32 %6, %6, %8\n v_xor_b32 %7, %7, %9\n"
          : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0), "v"(b1));
    } else if constexpr (KIND == 13) {  // VALU with an SGPR operand + s_cbranch-free scalar compare (v_fma with sgpr)
      asm volatile(
          "v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
          "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
          "v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
          "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
          : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "s"(s1), "v"(b1));
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * 256 + threadIdx.x] = p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y + a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (float)(s0 + s1 + s2 + s3);
  if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

template <int KIND>
void run(const char *name, int insts_per_iter, float *out, unsigned long long *cyc) {
  const int wpss[] = {1, 2, 4, 7, 8};
  printf("%-34s", name);
  for (int wps : wpss) {
    const int blocks = 256 * wps;  // 256-thread block = 1 wave per SIMD
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k_bench<KIND>, dim3(blocks), dim3(256), 0, 0, out, cyc, 100);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(k_bench<KIND>, dim3(blocks), dim3(256), 0, 0, out, cyc, kIters);
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipDeviceSynchronize());
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> h(blocks);
    CHECK(hipMemcpy(h.data(), cyc, blocks * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    double mean = 0;
    for (auto v : h) mean += (double)v;
    mean /= blocks;
    // per SIMD: wps waves x kIters x insts each, in `mean` cycles (s_memtime ticks = shader cycles)
    const double per_simd = (double)wps * kIters * insts_per_iter / mean;
    printf("  w%d: %.3f/clk (wall %.3f/ns, %.2f ms)", wps, per_simd, (double)wps * kIters * insts_per_iter / (ms * 1e6), ms);
  }
  printf("\n");
}

int main() {
  setvbuf(stdout, nullptr, _IONBF, 0);
  float *out; unsigned long long *cyc;
  CHECK(hipMalloc(&out, 256 * 8 * 256 * sizeof(float)));
  CHECK(hipMalloc(&cyc, 256 * 8 * sizeof(unsigned long long)));
  printf("wave-instructions per clock per SIMD (4 SIMDs per CU), by waves per SIMD\n");
  run<0>("v_fma_f32", 16, out, cyc);
  run<1>("v_mul_f32/v_add_f32", 16, out, cyc);
  run<2>("v_pk_fma_f32", 16, out, cyc);
  run<3>("v_pk_mul_f32/v_pk_add_f32", 16, out, cyc);
  run<4>("SALU (s_add/s_xor/s_lshl)", 16, out, cyc);
  run<5>("v_fma + SALU 1:1 (32 insts)", 32, out, cyc);
  run<6>("v_rcp_f32", 16, out, cyc);
  run<7>("v_div_scale/v_div_fixup", 16, out, cyc);
  run<8>("v_cmp/v_cndmask", 16, out, cyc);
  run<9>("ds_read_b128 broadcast", 16, out, cyc);
  run<10>("ds_read_b128 80B-stride per lane", 16, out, cyc);
  run<11>("v_fma_f32 dependent chain", 16, out, cyc);
  run<12>("int VALU and/or/add/xor", 16, out, cyc);
  run<13>("v_fma_f32 with SGPR operand", 16, out, cyc);
  return 0;
}
