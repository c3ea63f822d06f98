// gemm_wide: the GEMM family on a 256 x 256 output tile with 128 x 128 WAVE tiles and a hand-scheduled assembly main loop
// (DESIGN section 3 "128x128 wave tiles").  Four waves, one per SIMD, each owning the whole 512-register file:
// 256 accumulators in a[0:255] updated in place (hipcc cannot hold this tile - it copies accumulators once all AGPRs are taken,
// profiles/r03/negative_result_gemm256w_4waves_128x128_agpr.txt), 0.25 LDS fragment reads per MFMA, 64 KiB of operands per
// 8.4 MFLOP K-step = 32 B per MFMA clock asked of the CU's memory path (128x128 tiles: 64, the 256x128 ring kernel: 48).
// The feed is the vendor library's (MT256x256x64 MIWT8_8, profiles/r04/hipblaslt_kernels.txt): global -> VGPR -> LDS, the
// request one K-step ahead of the LDS write, the LDS write one K-step ahead of the fragment reads; one barrier per K-step.
// The loop text is generated (tools/gen_wide_loop.py -> gemm_wide_loop.inc, one asm statement); this file is the work walk,
// the per-tile address set-up and the epilogue (gemm_common.hpp's register-direct epilogue, one 64 x 64 piece at a time, read
// out of the accumulator file).  Serves pm_gemm (nn.Linear / 1x1: attention.py:415-442, openaimodel3d.py:154-190) where the
// grid keeps whole rounds of 256-wide tiles (prefer_wide below); the other kernels of gemm.hip keep the rest.
#include "gemm_common.hpp"
#include "gemm_wide_loop.inc"

namespace pm {

constexpr int WIDE_BM = 256, WIDE_BN = 256;
constexpr int WIDE_LDS = 128 * 1024;  // A buffers at 0 / 32 KiB, W buffers at 64 / 96 KiB

template <int IDX> __device__ __forceinline__ float agpr_read() {
  float x;
  asm volatile("v_accvgpr_read_b32 %0, a[%1]" : "=v"(x) : "n"(IDX) : "memory");
  return x;
}
// accumulator block (row block PI*4 + i, column block PJ*4 + j) of the wave's 8 x 8 -> acc[i][j] of one 64 x 64 piece
template <int PI, int PJ, int I> __device__ __forceinline__ void agpr_piece_row(f32x4 (&acc)[4][4]) {
#define PM_WIDE_RD(j) \
  acc[I][j] = f32x4{agpr_read<(((PI * 4 + I) * 8) + PJ * 4 + j) * 4 + 0>(), agpr_read<(((PI * 4 + I) * 8) + PJ * 4 + j) * 4 + 1>(), \
                    agpr_read<(((PI * 4 + I) * 8) + PJ * 4 + j) * 4 + 2>(), agpr_read<(((PI * 4 + I) * 8) + PJ * 4 + j) * 4 + 3>()}
  PM_WIDE_RD(0);
  PM_WIDE_RD(1);
  PM_WIDE_RD(2);
  PM_WIDE_RD(3);
#undef PM_WIDE_RD
}
template <int PI, int PJ> __device__ __forceinline__ void agpr_piece(f32x4 (&acc)[4][4]) {
  agpr_piece_row<PI, PJ, 0>(acc);
  agpr_piece_row<PI, PJ, 1>(acc);
  agpr_piece_row<PI, PJ, 2>(acc);
  agpr_piece_row<PI, PJ, 3>(acc);
}

// MODE (diagnostics, timing only): 1 = read the accumulators, no epilogue; 2 = the whole epilogue with every store masked off
template <typename T, int PI, int PJ, int MODE = 0>
__device__ __forceinline__ void wide_piece_epilogue(const GemmParams& p, const float (&bv)[4][4], int m_w, int n_w, int fr, int fq) {
  f32x4 acc[4][4];
  agpr_piece<PI, PJ>(acc);
  const int m_p = m_w + PI * 64, n_p = n_w + PJ * 64;
  if constexpr (MODE == 1) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) asm volatile("" ::"v"(acc[i][j]));
  } else if constexpr (MODE == 2) {
    GemmParams q = p;
    q.M = 0;
    epilogue_regs<T>(q, acc, bv, m_p, n_p, 0, 0, fr, fq, m_p >> 6, 0, false);
  } else if constexpr (MODE == 3) {
    epilogue_regs<T>(p, acc, bv, m_p, n_p, 0, 0, fr, fq, m_p >> 6, 0, false);  // (diagnostics: the generic epilogue on every flavour)
  } else if constexpr (MODE == 4) {
    epilogue_lean16<T, true>(p, acc, bv, m_p, n_p, fr, fq);  // (diagnostics: the lean arithmetic without its stores)
  } else {
    if (epilogue_lean16_ok(p, m_p, n_p))
      epilogue_lean16<T>(p, acc, bv, m_p, n_p, fr, fq);
    else
      epilogue_regs<T>(p, acc, bv, m_p, n_p, 0, 0, fr, fq, m_p >> 6, 0, false);
  }
}

#ifdef PM_DIAG
// diagnostics (variant 7): per workgroup {prologue, loop, epilogue} shader cycles, the loop's 100-MHz ticks, K-steps
__device__ unsigned long long* g_wide_stamps = nullptr;
#endif

template <typename T, int V>
__global__ __launch_bounds__(256) void gemm_wide_kernel(const GemmParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // (the dynamic region starts behind whatever static LDS hipcc gives the kernel - it promotes private arrays of the epilogue
  // into LDS - so every address of the loop is relative to it: low 32 bits of the generic address = the LDS byte offset)
  const uint32_t lds0 = (uint32_t)reinterpret_cast<uintptr_t>(smem);
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int G = gridDim.x;
  const int nwork = p.mtiles * p.ntiles;
  // work walk of the ring kernels (gemm.hip): each XCD sweeps a contiguous id range; ids in supertile order (8 row tiles x all
  // column tiles), so tiles that run together share A / W panels in one L2
  const bool xcd_walk = (G & 7) == 0 && nwork > G;
  const int per_xcd = (nwork + 7) >> 3, gx = G >> 3;
  const int slot_id = xcd_remap(blockIdx.x, G);
  const int nk_all = p.K / BK;

  const int wm = wave >> 1, wn = wave & 1;
  const int fr = lane & 15, fq = lane >> 4;
  const int r8 = lane >> 3;
  const int lc = (lane & 7) ^ r8;  // logical 16-byte k-chunk this lane stages (LDS position lane & 7 of row r8: chunk ^ (row & 7))
  // LDS addresses (bytes): staging writes / fragment reads of the two k-substeps
  const uint32_t lwa = lds0 + (uint32_t)(8 * wave * 128 + lane * 16), lww = 65536u + lwa;
  const uint32_t slot0 = (uint32_t)((fq ^ (fr & 7)) << 4), slot1 = (uint32_t)(((4 + fq) ^ (fr & 7)) << 4);
  const uint32_t lra0 = lds0 + (uint32_t)(wm * 16384 + fr * 128) + slot0, lra1 = lds0 + (uint32_t)(wm * 16384 + fr * 128) + slot1;
  const uint32_t lrw0 = lds0 + 65536u + (uint32_t)(wn * 16384 + fr * 128) + slot0;
  const uint32_t lrw1 = lds0 + 65536u + (uint32_t)(wn * 16384 + fr * 128) + slot1;

  // whole-tensor buffer descriptors: a lane's row offset goes in voffset, the K-step's byte offset in soffset
  const int64_t a_bytes = ((int64_t)(p.M - 1) * p.lda + p.K) * 2;
  const int64_t w_bytes = ((int64_t)(p.N - 1) * p.ldw + (p.kwrap ? p.kwrap : p.K)) * 2;
  const __amdgpu_buffer_rsrc_t adesc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.A), (short)0, (int)a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t wdesc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.Wt), (short)0, (int)w_bytes, 0x00020000);

#ifdef PM_DIAG
  unsigned long long st_pro = 0, st_loop = 0, st_epi = 0, st_real = 0, st_steps = 0;
#endif
  for (int round = 0;; ++round) {
    int w;
    if (!xcd_walk) {
      w = round * G + slot_id;
      if (w >= nwork) break;
    } else {
      const int xcd = blockIdx.x & 7, local = round * gx + (blockIdx.x >> 3);
      w = xcd * per_xcd + local;
      if (local >= per_xcd || w >= nwork) break;
    }
    constexpr int GM = 8;
    const int grp = w / (GM * p.ntiles);
    const int first_m = grp * GM;
    const int gm = (p.mtiles - first_m < GM) ? p.mtiles - first_m : GM;
    const int rin = w - grp * GM * p.ntiles;
    const int nt = rin / gm;
    const int mt = first_m + (rin - nt * gm);
    const int m0 = mt * WIDE_BM, n0 = nt * WIDE_BN;

    // (r8 / lc re-defined per tile for the same reason as fr_e / fq_e below: nothing of the per-lane offset arithmetic may be
    // hoisted above the tile loop - it would have to live across the loop statement in the compiler's 64 registers)
    int r8_t = r8, lc_t = lc;
    asm volatile("" : "+v"(r8_t), "+v"(lc_t));
    uint32_t ao[8], bo[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int pr = 32 * j + 8 * wave + r8_t;  // row of the A tile / LDS row of the W tile
      int m = m0 + pr;
      if (m > p.M - 1) m = p.M - 1;
      ao[j] = (uint32_t)(((int64_t)m * p.lda + lc_t * 8) * 2);
      int n = n0 + cperm(pr, p.act == PM_ACT_GEGLU || p.natural);  // LDS row -> output column (epilogue_regs)
      if (n > p.N - 1) n = p.N - 1;
      bo[j] = (uint32_t)(((int64_t)n * p.ldw + lc_t * 8) * 2);
    }
    uint32_t kld = 0, nk = (uint32_t)nk_all;
    const uint32_t kmax = (uint32_t)(nk_all - 1) * 128u;
#define PM_WIDE_OPERANDS                                                                                                       \
  [kld] "+s"(kld), [nk] "+s"(nk)                                                                                               \
      : [ao0] "v"(ao[0]), [ao1] "v"(ao[1]), [ao2] "v"(ao[2]), [ao3] "v"(ao[3]), [ao4] "v"(ao[4]), [ao5] "v"(ao[5]),               \
        [ao6] "v"(ao[6]), [ao7] "v"(ao[7]), [bo0] "v"(bo[0]), [bo1] "v"(bo[1]), [bo2] "v"(bo[2]), [bo3] "v"(bo[3]),               \
        [bo4] "v"(bo[4]), [bo5] "v"(bo[5]), [bo6] "v"(bo[6]), [bo7] "v"(bo[7]), [lwa] "v"(lwa), [lww] "v"(lww), [lra0] "v"(lra0), \
        [lra1] "v"(lra1), [lrw0] "v"(lrw0), [lrw1] "v"(lrw1), [adesc] "s"(adesc), [wdesc] "s"(wdesc), [kmax] "s"(kmax)            \
      : PM_WIDE_CLOBBERS
    if constexpr (!std::is_same<T, bf16>::value) {
      asm volatile(PM_WIDE_LOOP_DENSE_F16_V0 : PM_WIDE_OPERANDS);
    } else if constexpr (V == 0) {
      asm volatile(PM_WIDE_LOOP_DENSE_BF16_V0 : PM_WIDE_OPERANDS);
    }
#ifdef PM_DIAG
    else if constexpr (V == 1) { asm volatile(PM_WIDE_LOOP_DENSE_BF16_V1 : PM_WIDE_OPERANDS); }
    else if constexpr (V == 3) { asm volatile(PM_WIDE_LOOP_DENSE_BF16_V3 : PM_WIDE_OPERANDS); }
    else if constexpr (V == 4) { asm volatile(PM_WIDE_LOOP_DENSE_BF16_V4 : PM_WIDE_OPERANDS); }
    else if constexpr (V == 6) { asm volatile(PM_WIDE_LOOP_DENSE_BF16_V6 : PM_WIDE_OPERANDS); }
    unsigned long long t_a = 0, t_b = 0, t_pro = 0, r_a = 0, r_b = 0;
    if constexpr (V >= 7) {
      t_a = __builtin_amdgcn_s_memtime();
      asm volatile(PM_WIDE_LOOP_DENSE_BF16_V7 : [tpro] "=&s"(t_pro), PM_WIDE_OPERANDS);
      r_a = __builtin_amdgcn_s_memrealtime();
      t_b = __builtin_amdgcn_s_memtime();
    }
#endif
#undef PM_WIDE_OPERANDS
    // ---- epilogue: the wave's 128 x 128 block as four 64 x 64 pieces out of the accumulator file -----------------------
    // (lane coordinates re-defined HERE: left loop-invariant, hipcc hoists the epilogue's whole per-lane address arithmetic
    // above the tile loop and has to keep it alive across the loop statement, where it owns 64 registers - 62 spills)
    int fr_e = fr, fq_e = fq;
    asm volatile("" : "+v"(fr_e), "+v"(fq_e));
    const int m_w = m0 + wm * 128, n_w = n0 + wn * 128;
    float bv0[4][4], bv1[4][4];  // both column halves' bias requested before the first accumulator is read: one round trip
    load_bias_regs(p, bv0, n_w, 0, fq_e);
    load_bias_regs(p, bv1, n_w + 64, 0, fq_e);
    constexpr int EMODE = (V == 8) ? 1 : (V == 9) ? 2 : (V == 10) ? 3 : (V == 11) ? 4 : 0;
    wide_piece_epilogue<T, 0, 0, EMODE>(p, bv0, m_w, n_w, fr_e, fq_e);
    wide_piece_epilogue<T, 0, 1, EMODE>(p, bv1, m_w, n_w, fr_e, fq_e);
    wide_piece_epilogue<T, 1, 0, EMODE>(p, bv0, m_w, n_w, fr_e, fq_e);
    wide_piece_epilogue<T, 1, 1, EMODE>(p, bv1, m_w, n_w, fr_e, fq_e);
#ifdef PM_DIAG
    if constexpr (V >= 7) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const unsigned long long t_c = __builtin_amdgcn_s_memtime();
      st_pro += t_pro - t_a;
      st_loop += t_b - t_pro;
      st_epi += t_c - t_b;
      st_real = r_a;  // (last tile's end: the host divides the span by the sum of cycles)
      st_steps += (unsigned long long)nk_all;
    }
#endif
    // (the next tile's prologue overwrites LDS buffer 0: every wave is past the last K-step's barrier, behind which no wave
    // reads anything it still needs - tools/gen_wide_loop.py)
  }
#ifdef PM_DIAG
  if constexpr (V >= 7) {
    if (g_wide_stamps != nullptr && tid == 0) {
      unsigned long long* d = g_wide_stamps + (size_t)blockIdx.x * 8;
      d[0] = st_pro; d[1] = st_loop; d[2] = st_epi; d[3] = st_real; d[4] = st_steps;
      d[5] = __builtin_amdgcn_s_memtime(); d[6] = __builtin_amdgcn_s_memrealtime();
    }
  }
#endif
}

static int g_wide = 1;  // PANDORA_GEMM_WIDE (diagnostics build): 0 = never, 1 = by prefer_wide(), 2 = wherever legal

bool gemm_wide_legal(const GemmParams& p, int flags) {
  if (flags & (PM_FLAG_A_F32 | PM_FLAG_A_LO)) return false;
  if (p.splits != 1 || p.kwrap != 0) return false;
  if (p.K % BK || p.K < 2 * BK) return false;
  if ((p.lda & 7) || (p.ldw & 7)) return false;
  return true;
}

static int wide_mode() {
  static const bool init = [] {
    const char* e = diag_env("PANDORA_GEMM_WIDE");
    if (e) g_wide = atoi(e);
    return true;
  }();
  (void)init;
  return g_wide;
}

bool gemm_wide_wanted(const GemmParams& p, int flags, int num_cus) {
  const int mode = wide_mode();
  if (mode == 0 || !gemm_wide_legal(p, flags)) return false;
  if (mode == 2) return true;
  // whole rounds of 256 x 256 tiles and a K loop long enough to amortise the un-overlapped prologue / epilogue of a tile
  // (in-kernel stamps: ~6 000 + ~16 500 cycles per tile against ~2 370 per K-step - profiles/r06/wide_kernel_stamps.txt; measured
  // per shape against the other kernels in profiles/r06/wide_probe.txt: ahead from K = 5120, 31 % at K = 11520, level at K = 1280)
  const int64_t mt = (p.M + WIDE_BM - 1) / WIDE_BM, nt = (p.N + WIDE_BN - 1) / WIDE_BN;
  const int64_t tiles = mt * nt;
  const double rounds = (double)tiles / (double)(((tiles + num_cus - 1) / num_cus) * num_cus);
  const double fill = ((double)p.M / (double)(mt * WIDE_BM)) * ((double)p.N / (double)(nt * WIDE_BN));
  return p.K >= 40 * BK && tiles >= num_cus && rounds * fill >= 0.85;
}

static int g_wide_variant = 0;  // diagnostics build: which generated loop variant bf16 launches run (tools/gen_wide_loop.py VARIANTS)

template <typename T, int V> static int launch_wide_v(const GemmParams& q, int grid, hipStream_t stream) {
  static bool attr_set[16] = {false};
  int dev = 0;
  (void)hipGetDevice(&dev);
  dev = (dev >= 0 && dev < 16) ? dev : 0;
  if (!attr_set[dev]) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_wide_kernel<T, V>), hipFuncAttributeMaxDynamicSharedMemorySize, WIDE_LDS);
    attr_set[dev] = true;
  }
  hipLaunchKernelGGL((gemm_wide_kernel<T, V>), dim3(grid), dim3(256), WIDE_LDS, stream, q);
  return check_launch();
}

template <typename T> int launch_gemm_wide(const GemmParams& p, int num_cus, hipStream_t stream) {
  GemmParams q = p;
  q.mtiles = (p.M + WIDE_BM - 1) / WIDE_BM;
  q.ntiles = (p.N + WIDE_BN - 1) / WIDE_BN;
  const int64_t nwork = (int64_t)q.mtiles * q.ntiles;
  const int grid = (int)(nwork < num_cus ? nwork : num_cus);
#ifdef PM_DIAG
  if constexpr (std::is_same<T, bf16>::value) {
    switch (g_wide_variant) {
      case 1: return launch_wide_v<T, 1>(q, grid, stream);
      case 3: return launch_wide_v<T, 3>(q, grid, stream);
      case 4: return launch_wide_v<T, 4>(q, grid, stream);
      case 6: return launch_wide_v<T, 6>(q, grid, stream);
      case 7: return launch_wide_v<T, 7>(q, grid, stream);
      case 8: return launch_wide_v<T, 8>(q, grid, stream);
      case 9: return launch_wide_v<T, 9>(q, grid, stream);
      case 10: return launch_wide_v<T, 10>(q, grid, stream);
      case 11: return launch_wide_v<T, 11>(q, grid, stream);
      default: break;
    }
  }
#endif
  return launch_wide_v<T, 0>(q, grid, stream);
}
template int launch_gemm_wide<f16>(const GemmParams&, int, hipStream_t);
template int launch_gemm_wide<bf16>(const GemmParams&, int, hipStream_t);

}  // namespace pm

#ifdef PM_DIAG
// diagnostics: 0 = never, 1 = by the rule, 2 = wherever legal (A/B runs in one process, tools/wide_probe.py)
extern "C" void pm_debug_wide_stamps(void* buf) {  // device buffer of 64 bytes x grid size (variant 7), or NULL
  unsigned long long* b = reinterpret_cast<unsigned long long*>(buf);
  (void)hipMemcpyToSymbol(HIP_SYMBOL(pm::g_wide_stamps), &b, sizeof(b));
}
extern "C" void pm_debug_gemm_wide(int mode) {  // mode + 16 * loop variant
  (void)pm::wide_mode();
  pm::g_wide = mode & 15;
  pm::g_wide_variant = mode >> 4;
}
#endif
