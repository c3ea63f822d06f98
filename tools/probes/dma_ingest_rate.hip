// Probe: what ONE CU ingests into LDS by DMA (global_load_lds_dwordx4, 1 KiB per wave-instruction) as a function of
//   * the contiguous row segment a K-tile row brings (128 B = BK 64 of 16-bit operands, 64 B = BK 32),
//   * where the bytes come from (the XCD's L2, the Infinity Cache, HBM),
//   * how many KiB the loader waves keep in flight (counted s_waitcnt vmcnt),
//   * how many waves issue.
// One 1-workgroup-per-CU grid (140 KiB of LDS requested), every workgroup walks its OWN region of `region` bytes
// `passes` times: region 64 KiB x 256 workgroups = 2 MiB per XCD (L2), 512 KiB = 128 MiB total (Infinity Cache),
// 16 MiB = 4 GiB total (HBM).  Rows are `seg` bytes of a matrix with a 1024-byte pitch (K = 512, 16 bit).
// hipcc --offload-arch=gfx950 -O2 tools/probes/dma_ingest_rate.hip -o /tmp/dma && /tmp/dma
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;

template <int SEG, int DEPTH>  // DEPTH = DMAs (KiB) each wave keeps in flight
__global__ __launch_bounds__(256, 1) void ingest(const char* src, long region, int passes, int nwaves, long long* cycles) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (wave >= nwaves) return;
  constexpr int LPR = SEG / 16;            // lanes per row
  constexpr int RPI = 64 / LPR;            // rows per wave-instruction
  constexpr long PITCH = 1024;
  // the region as [rows][PITCH]: a K-tile column kt covers bytes kt*SEG.. of every row; walk rows fastest (a tile), then kt
  const long rows = region / PITCH;        // rows of this workgroup's region
  const char* base = src + (long)blockIdx.x * region;
  const int r_in = lane / LPR, c_in = lane % LPR;
  const long long t0 = clock64();
  long issued = 0;
  for (int ps = 0; ps < passes; ++ps) {
    for (long kt = 0; kt < PITCH / SEG; ++kt) {
      for (long r0 = wave * RPI; r0 + RPI <= rows; r0 += (long)nwaves * RPI) {
        const char* g = base + (r0 + r_in) * PITCH + kt * SEG + c_in * 16;
        char* l = smem + (size_t)wave * 32768 + (issued % 32) * 1024;  // (ring of 32 KiB per wave)
        __builtin_amdgcn_global_load_lds((glb_void*)g, (lds_void*)l, 16, 0, 0);
        ++issued;
        if (issued >= DEPTH) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DEPTH - 1) : "memory");
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const long long t1 = clock64();
  if (lane == 0) {
    cycles[(blockIdx.x * 4 + wave) * 2] = t1 - t0;
    cycles[(blockIdx.x * 4 + wave) * 2 + 1] = issued;
  }
}

template <int SEG, int DEPTH> static void run(const char* d, long region, int nwaves, const char* what, long long* dc) {
  const int G = 256;
  int passes = (int)(((long)32 << 20) / region);  // 32 MiB per workgroup and launch
  if (passes < 2) passes = 2;
  hipFuncSetAttribute(reinterpret_cast<const void*>(ingest<SEG, DEPTH>), hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((ingest<SEG, DEPTH>), dim3(G), dim3(256), 140 * 1024, 0, d, region, 1, nwaves, dc);  // warm
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((ingest<SEG, DEPTH>), dim3(G), dim3(256), 140 * 1024, 0, d, region, passes, nwaves, dc);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  const double bytes = (double)G * passes * (double)region;
  printf("%-22s seg %3d B  %d waves x %2d KiB in flight: %7.1f GB/s per CU  (%6.2f TB/s chip, %.3f ms)\n", what, SEG, nwaves, DEPTH,
         bytes / (ms * 1e-3) / 1e9 / G, bytes / (ms * 1e-3) / 1e12, ms);
}


// LEAN variant: the address form the GEMM kernels use - a wave-uniform 64-bit base (SGPRs) that walks the region plus a
// constant 32-bit lane offset - so a DMA costs ~4 scalar instructions; up to 8 issuing waves (512 threads).
__device__ __forceinline__ void glds16(const void* sbase, uint32_t voff, uint32_t lds_dst) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}
template <int DEPTH>
__global__ __launch_bounds__(512, 1) void ingest_lean(const char* src, long region, int passes, int nwaves) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (wave >= nwaves) return;
  const uint32_t lds0 = (uint32_t)reinterpret_cast<uintptr_t>((lds_void*)smem) + wave * 16384;
  // 8 rows x 128 B per DMA, rows of a 1024-byte-pitch matrix: lane offset constant, base walks 8 rows (8 KiB) per DMA
  const uint32_t voff = (uint32_t)((lane >> 3) * 1024 + (lane & 7) * 16);
  const char* base = src + (long)blockIdx.x * region;
  const long per_pass = region / 8192;  // 8-row groups (x 8 column tiles of 128 B)
  long issued = 0;
  for (int ps = 0; ps < passes; ++ps)
    for (int kt = 0; kt < 8; ++kt)
      for (long gidx = wave; gidx < per_pass; gidx += nwaves) {
        glds16(base + gidx * 8192 + kt * 128, voff, lds0 + (uint32_t)(issued & 15) * 1024);
        ++issued;
        if (issued >= DEPTH) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DEPTH - 1) : "memory");
      }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
template <int DEPTH> static void run_lean(const char* d, long region, int nwaves, const char* what) {
  const int G = 256;
  int passes = (int)(((long)32 << 20) / region);
  if (passes < 2) passes = 2;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(ingest_lean<DEPTH>), hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((ingest_lean<DEPTH>), dim3(G), dim3(512), 140 * 1024, 0, d, region, 1, nwaves);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL((ingest_lean<DEPTH>), dim3(G), dim3(512), 140 * 1024, 0, d, region, passes, nwaves);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  const double bytes = (double)G * passes * (double)region;
  printf("%-22s LEAN 128 B  %d waves x %2d KiB in flight: %7.1f GB/s per CU  (%6.2f TB/s chip, %.3f ms)\n", what, nwaves, DEPTH,
         bytes / (ms * 1e-3) / 1e9 / G, bytes / (ms * 1e-3) / 1e12, ms);
}

// MIXED variant: does issuing DMAs cost a COMPUTING wave its matrix time?  4 waves (one per SIMD), each iteration =
// NDMA LDS-DMAs (L2-resident region) + 32 back-to-back v_mfma_f32_16x16x32_bf16 (512 matrix cycles).
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
template <int NDMA>
__global__ __launch_bounds__(256, 1) void ingest_mixed(const char* src, long region, int iters, float* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint32_t lds0 = (uint32_t)reinterpret_cast<uintptr_t>((lds_void*)smem) + wave * 16384;
  const uint32_t voff = (uint32_t)((lane >> 3) * 1024 + (lane & 7) * 16);
  const char* base = src + (long)blockIdx.x * region;
  const long groups = region / 8192;
  bf16x8_t a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(float)(lane + e); b[e] = (__bf16)(float)(lane - e); }
  f32x4_t acc[16];
  for (int i = 0; i < 16; ++i) acc[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  long issued = wave;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int d = 0; d < NDMA; ++d) {
      const long gidx = issued % groups;
      glds16(base + gidx * 8192 + ((issued / groups) & 7) * 128, voff, lds0 + (uint32_t)(issued & 15) * 1024);
      issued += 4;
    }
    if (NDMA > 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA * 3) : "memory");  // (three iterations' DMAs stay in flight)
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += acc[i][0];
  if (s == 12345.678f) sink[0] = s;
}
template <int NDMA> static void run_mixed(const char* d, float* sink) {
  const int G = 256, iters = 20000;
  const long region = 64l << 10;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(ingest_mixed<NDMA>), hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((ingest_mixed<NDMA>), dim3(G), dim3(256), 140 * 1024, 0, d, region, 100, sink);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL((ingest_mixed<NDMA>), dim3(G), dim3(256), 140 * 1024, 0, d, region, iters, sink);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  const double flops = (double)G * 4 * iters * 32 * 2.0 * 16 * 16 * 32;
  const double bytes = (double)G * 4 * iters * NDMA * 1024.0;
  printf("MIXED 4 waves, %d DMAs + 32 MFMAs per iteration: %7.1f ns per iteration, %7.0f TF/s, %6.1f GB/s per CU ingested\n", NDMA,
         ms * 1e6 / iters, flops / (ms * 1e-3) / 1e12, bytes / (ms * 1e-3) / 1e9 / G);
}

int main() {
  const long total = 4l << 30;
  char* d;
  if (hipMalloc(&d, total) != hipSuccess) { printf("alloc failed\n"); return 1; }
  hipMemset(d, 1, total);
  long long* dc;
  hipMalloc(&dc, 256 * 4 * 2 * 8);
  struct { long region; const char* what; } src[3] = {{64l << 10, "L2 (64 KiB / WG)"}, {512l << 10, "Inf. Cache (512 KiB)"}, {16l << 20, "HBM (16 MiB / WG)"}};
  for (auto& s : src) {
    run<128, 8>(d, s.region, 4, s.what, dc);
    run<64, 8>(d, s.region, 4, s.what, dc);
    run<128, 16>(d, s.region, 4, s.what, dc);
    run<64, 16>(d, s.region, 4, s.what, dc);
    run<128, 24>(d, s.region, 4, s.what, dc);
    run<64, 24>(d, s.region, 4, s.what, dc);
    run<128, 28>(d, s.region, 4, s.what, dc);
    run<64, 28>(d, s.region, 4, s.what, dc);
    run<128, 32>(d, s.region, 2, s.what, dc);
    run<64, 32>(d, s.region, 2, s.what, dc);
  }
  float* sink;
  (void)hipMalloc(&sink, 64);
  run_mixed<0>(d, sink);
  run_mixed<1>(d, sink);
  run_mixed<2>(d, sink);
  run_mixed<4>(d, sink);
  run_mixed<8>(d, sink);
  for (auto& sr : src) {
    run_lean<8>(d, sr.region, 2, sr.what);
    run_lean<8>(d, sr.region, 4, sr.what);
    run_lean<8>(d, sr.region, 8, sr.what);
    run_lean<12>(d, sr.region, 8, sr.what);
    run_lean<16>(d, sr.region, 4, sr.what);
  }
  return 0;
}
