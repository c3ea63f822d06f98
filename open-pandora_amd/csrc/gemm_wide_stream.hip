// gemm_wide_stream: gemm_wide's tile (256 x 256, four waves of 128 x 128, accumulators fixed in a[0:255], csrc/gemm_wide.hip) as ONE
// assembly statement per workgroup - prologue, K loop, tile boundary and epilogue (tools/gen_wide_stream.py ->
// gemm_wide_stream.inc).  gemm_wide's in-kernel stamps (profiles/r06/wide_kernel_stamps.txt) put its per-tile fixed cost at ~18 000-
// 23 000 cycles (prologue behind the previous tile's store drain + the C++ epilogue) against ~2 370 per K-step: 43 % of a K = 640
// tile.  Here the K stream runs ACROSS tile boundaries (the last three K-steps of a tile request the next tile's first ones), the
// stores of a tile drain under the next tile's loop (they are younger than its first requests in the in-order vmcnt queue), the
// bias is the accumulators' initial value and the conversion + stores are straight-line assembly.  The C++ below only builds the
// workgroup's tile list (as a table of four scalars per tile in LDS, read by the statement with one ds_read per tile) and the
// lane constants: between the statement's first and last instruction the compiler touches nothing.
// Serves pm_gemm's 16-bit projection flavours without residual / statistics: plain or + bias ("lin"), GEGLU (attention.py:415-442);
// M % 256 == 0, N % 256 == 0, K >= 256.  Everything else: the kernels of gemm.hip / gemm_wide.hip.
#include "gemm_common.hpp"
#include "gemm_wide_stream.inc"

namespace pm {

constexpr int WS_LDS = 128 * 1024 + 4 * 1024;  // the loop's two 64-KiB buffers + one 1-KiB tile table per wave
constexpr int WS_MAX_TILES = 64;               // tiles per workgroup (table entries)

template <typename T, bool GEGLU>
__global__ __launch_bounds__(256) void gemm_wide_stream_kernel(const GemmParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const uint32_t lds0 = (uint32_t)reinterpret_cast<uintptr_t>(smem);
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int G = gridDim.x;
  const int nwork = p.mtiles * p.ntiles;
  const bool xcd_walk = (G & 7) == 0 && nwork > G;
  const int per_xcd = (nwork + 7) >> 3, gx = G >> 3;
  const int slot_id = xcd_remap(blockIdx.x, G);
  // ---- this workgroup's tiles, in the ring kernels' walk (gemm.hip): {A tile offset, W tile offset, C offset of this wave's
  // 128 x 128 block, bias offset of its columns}, bytes, one 16-byte table entry per tile
  u32x4* const table = reinterpret_cast<u32x4*>(smem + 128 * 1024 + wave * 1024);
  int ntl = 0;
  for (int round = 0; round < WS_MAX_TILES; ++round) {
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
    const int64_t m0 = (int64_t)mt * 256, n0 = (int64_t)nt * 256;
    u32x4 ent;
    ent[0] = (uint32_t)(m0 * p.lda * 2);
    ent[1] = (uint32_t)(n0 * p.ldw * 2);
    ent[2] = (uint32_t)(((m0 + wm * 128) * p.ldc + (GEGLU ? (n0 >> 1) + wn * 64 : n0 + wn * 128)) * 2);
    ent[3] = (uint32_t)((n0 + wn * 128) * 4);
    if (lane == 0) table[round] = ent;
    ++ntl;
  }
  if (ntl == 0) return;  // (wave-uniform: the whole workgroup)
  // ---- lane constants
  const int fr = lane & 15, fq = lane >> 4;
  const int r8 = lane >> 3;
  const int lc = (lane & 7) ^ r8;
  const int row = 8 * wave + r8;  // this lane's row inside every 32-row group of the staging pieces
  const bool nat = GEGLU;         // (W rows in natural order for GEGLU, interleaved by cperm otherwise: epilogue_regs' layout)
  const uint32_t aob = (uint32_t)(((int64_t)row * p.lda + lc * 8) * 2);
  const uint32_t bob0 = (uint32_t)(((int64_t)cperm(row, nat) * p.ldw + lc * 8) * 2);
  const uint32_t bob1 = (uint32_t)(((int64_t)cperm(32 + row, nat) * p.ldw + lc * 8) * 2);
  const uint32_t lwa = lds0 + (uint32_t)(8 * wave * 128 + lane * 16), lww = 65536u + lwa;
  const uint32_t slot0 = (uint32_t)((fq ^ (fr & 7)) << 4), slot1 = (uint32_t)(((4 + fq) ^ (fr & 7)) << 4);
  const uint32_t lra0 = lds0 + (uint32_t)(wm * 16384 + fr * 128) + slot0, lra1 = lds0 + (uint32_t)(wm * 16384 + fr * 128) + slot1;
  const uint32_t lrw0 = lds0 + 65536u + (uint32_t)(wn * 16384 + fr * 128) + slot0;
  const uint32_t lrw1 = lds0 + 65536u + (uint32_t)(wn * 16384 + fr * 128) + slot1;
  const uint32_t cvo = (uint32_t)(((int64_t)fr * p.ldc + (GEGLU ? 4 : 8) * fq) * 2);
  const uint32_t bvo = (uint32_t)(fq * (GEGLU ? 16 : 32));
  // ---- whole-tensor buffer descriptors (a NULL bias: zero records, every load returns 0)
  const int nout = GEGLU ? (p.N >> 1) : p.N;
  const int64_t a_bytes = ((int64_t)(p.M - 1) * p.lda + p.K) * 2, w_bytes = ((int64_t)(p.N - 1) * p.ldw + p.K) * 2;
  const int64_t c_bytes = ((int64_t)(p.M - 1) * p.ldc + nout) * 2;
  const __amdgpu_buffer_rsrc_t adesc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.A), (short)0, (int)a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t wdesc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.Wt), (short)0, (int)w_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t cdesc = __builtin_amdgcn_make_buffer_rsrc(p.C, (short)0, (int)c_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t bdesc = __builtin_amdgcn_make_buffer_rsrc(
      p.bias ? const_cast<float*>(p.bias) : reinterpret_cast<float*>(p.C), (short)0, p.bias ? p.N * 4 : 0, 0x00020000);
  const uint32_t sa = (uint32_t)(32 * p.lda * 2), sw = (uint32_t)(64 * p.ldw * 2), rs = (uint32_t)(16 * p.ldc * 2);
  const uint32_t nk = (uint32_t)(p.K / BK), ntl_u = (uint32_t)ntl;
  const uint32_t tbl = lds0 + 128u * 1024u + (uint32_t)wave * 1024u;
  // (the table entries were written by this wave's lane 0: LDS operations of one wave execute in order - the statement's first
  // instruction is the read of entry 0; the compiler's own lgkmcnt wait for its stores is harmless)
#define PM_WS_OPERANDS                                                                                                               \
  : [aob] "v"(aob), [bob0] "v"(bob0), [bob1] "v"(bob1), [lwa] "v"(lwa), [lww] "v"(lww), [lra0] "v"(lra0), [lra1] "v"(lra1),          \
    [lrw0] "v"(lrw0), [lrw1] "v"(lrw1), [cvo] "v"(cvo), [bvo] "v"(bvo), [adesc] "s"(adesc), [wdesc] "s"(wdesc), [cdesc] "s"(cdesc),  \
    [bdesc] "s"(bdesc), [sa] "s"(sa), [sw] "s"(sw), [rs] "s"(rs), [nk] "s"(nk), [ntl] "s"(ntl_u), [tbl] "s"(tbl)                     \
  : PM_WSTREAM_CLOBBERS
  if constexpr (std::is_same<T, bf16>::value) {
    if constexpr (GEGLU) {
      asm volatile(PM_WSTREAM_GEGLU_BF16 : PM_WS_OPERANDS);
    } else {
      asm volatile(PM_WSTREAM_LIN_BF16 : PM_WS_OPERANDS);
    }
  } else {
    if constexpr (GEGLU) {
      asm volatile(PM_WSTREAM_GEGLU_F16 : PM_WS_OPERANDS);
    } else {
      asm volatile(PM_WSTREAM_LIN_F16 : PM_WS_OPERANDS);
    }
  }
#undef PM_WS_OPERANDS
}

static int g_wstream = 1;  // PANDORA_GEMM_WSTREAM (diagnostics build): 0 = never, 1 = by the rule, 2 = wherever legal
static int wstream_mode() {
  static const bool init = [] {
    const char* e = diag_env("PANDORA_GEMM_WSTREAM");
    if (e) g_wstream = atoi(e);
    return true;
  }();
  (void)init;
  return g_wstream;
}

bool gemm_wide_stream_wanted(const GemmParams& p, int flags, int num_cus) {
  const int mode = wstream_mode();
  if (mode == 0) return false;
  if (flags & (PM_FLAG_A_F32 | PM_FLAG_A_LO | PM_FLAG_OUT_F32 | PM_FLAG_BIAS_IS_SCALE)) return false;
  if (p.splits != 1 || p.kwrap != 0 || p.R != nullptr || p.colstats != nullptr || p.out32) return false;
  if (p.act != PM_ACT_NONE && p.act != PM_ACT_GEGLU) return false;
  if ((p.M & 255) || (p.N & 255) || (p.K % BK) || p.K < 4 * BK) return false;
  if ((p.lda & 7) || (p.ldw & 7) || (p.ldc & (p.act == PM_ACT_GEGLU ? 3 : 7))) return false;
  const int64_t tiles = (int64_t)(p.M / 256) * (p.N / 256);
  const int64_t grid = tiles < num_cus ? tiles : num_cus;
  if ((tiles + grid - 1) / grid + 1 > WS_MAX_TILES) return false;
  if (mode == 2) return true;
  // by the rule (profiles/r06/wide_stream_probe.txt: 15-33 % ahead of the other kernels on the GEGLU / wide projections from 150
  // tiles up; behind them on few-tile grids - 50 tiles at K = 5120 lose 2x to split-K - and on 1.25-1.4 rounds): at least 140 tiles,
  // and either one round or >= 0.75 of whole rounds
  const double rounds = (double)tiles / (double)(((tiles + num_cus - 1) / num_cus) * num_cus);
  return tiles >= 140 && (tiles <= num_cus || rounds >= 0.75);
}

template <typename T> int launch_gemm_wide_stream(const GemmParams& p, int num_cus, hipStream_t stream) {
  GemmParams q = p;
  q.mtiles = p.M / 256;
  q.ntiles = p.N / 256;
  const int64_t nwork = (int64_t)q.mtiles * q.ntiles;
  const int grid = (int)(nwork < num_cus ? nwork : num_cus);
  const bool geglu = p.act == PM_ACT_GEGLU;
  static bool attr_set[16][2] = {};
  int dev = 0;
  (void)hipGetDevice(&dev);
  dev = (dev >= 0 && dev < 16) ? dev : 0;
  if (!attr_set[dev][geglu]) {
    const void* fn = geglu ? reinterpret_cast<const void*>(gemm_wide_stream_kernel<T, true>)
                           : reinterpret_cast<const void*>(gemm_wide_stream_kernel<T, false>);
    (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, WS_LDS);
    attr_set[dev][geglu] = true;
  }
  if (geglu)
    hipLaunchKernelGGL((gemm_wide_stream_kernel<T, true>), dim3(grid), dim3(256), WS_LDS, stream, q);
  else
    hipLaunchKernelGGL((gemm_wide_stream_kernel<T, false>), dim3(grid), dim3(256), WS_LDS, stream, q);
  return check_launch();
}
template int launch_gemm_wide_stream<f16>(const GemmParams&, int, hipStream_t);
template int launch_gemm_wide_stream<bf16>(const GemmParams&, int, hipStream_t);

}  // namespace pm

#ifdef PM_DIAG
extern "C" void pm_debug_gemm_wstream(int mode) {
  (void)pm::wstream_mode();
  pm::g_wstream = mode;
}
#endif
