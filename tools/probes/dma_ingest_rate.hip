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
  return 0;
}
