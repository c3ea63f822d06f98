// LayerNorm + Linear in ONE kernel: an A-panel-stationary MFMA GEMM for the short-K projections of the
// transformer blocks (norm1/2/3 followed by to_q|k|v, to_q or the GEGLU ff.net[0]: attention.py:242-246,
// :86-99, :418-442) at the shallowest U-Net level (K = 320), where M is 4e4..1.5e5 rows.
//
//   C[M, N] = epi( LayerNorm(X)[M, K] . W[N, K]^T ),   X = the f32 residual stream.
//
// Why a second GEMM structure.  gemm_kernel streams 128x64 A and W tiles through LDS per K-step: at K = 320 a
// tile lives for 5 K-steps, every one of them a DMA round trip (~1 us under load with 64 KiB in flight per CU),
// and the A panel is fetched again for every one of the N/128 column tiles; LayerNorm ran as its own pass in
// front (read 4 B + write 2 B per element, then the GEMM reads the 2 B).  Here a 256-thread workgroup owns a
// PANEL of 128 rows for its whole sweep over N; two such workgroups share a CU (80 KiB of LDS each):
//   * prologue: the 4 waves read their rows of X once (f32, 16 lanes per row, all loads in flight together),
//     normalise them in registers (two-pass mean / variance, the arithmetic of layernorm_rows_kernel) and write
//     the 16-bit panel into LDS in the K-tile-major, XOR-swizzled image the MFMA fragment reads want - the
//     LayerNorm output never exists in HBM;
//   * one barrier; after it the waves never synchronise again: wave w walks the 32-column blocks w, w+4, ...
//     of the workgroup's column range.  Its W fragments come straight from global memory into registers
//     (W is L2-resident: <= 1.6 MB, re-read by every panel), two K-steps ahead (three register sets); its A
//     fragments come from the LDS panel (one ds_read_b128 per MFMA pair, issued one substep ahead);
//   * per-wave register epilogue (bias or per-column scale, GEGLU) with 16-byte (8-byte GEGLU) stores; the waves
//     drift apart, so one wave's epilogue runs beside its SIMD partner's MFMAs.
// Operand traffic per FLOP: W only, 4 KiB per 128x32x64 wave-step = half of the 128x128 tile kernel's A + W.
// Measured (tools/lngemm_bench.py, profiles/r02/lngemm_*.txt): 1.2-1.5x the pm_layernorm + pm_gemm pair at K = 320;
// K = 640 panels (96 or 64 rows) were built and measured at 0.7-1.1x and are not served (pm_ln_gemm_supported);
// lower panels at K = 320 (80 / 64 rows, three workgroups per CU) lose to the 128-row panel by 5-25 %: the W
// fragments' L2 traffic per FLOP, not occupancy, is what the sweep is sensitive to.
#include <stdlib.h>
#include "common.hpp"

// waves per workgroup: two 4-wave workgroups share a CU (80 KiB of LDS each), so one's LayerNorm prologue (HBM
// reads, no MFMA) runs beside the other's sweep; a single 8-wave workgroup left the CU idle for every prologue
constexpr int LN_NW = 4;

#ifndef LNG_DBG
#define LNG_DBG 0  // diagnosis builds only (tools/lngemm_variants.sh): 1 no W loads, 2 no A reads, 4 no stores in the sweep
#endif

namespace pm {

struct LnGemmParams {
  const float* X;
  int64_t ldx;
  const float* gamma;
  const float* beta;
  float eps;
  const void* W;
  int64_t ldw;
  const float* bias;
  void* C;
  int64_t ldc;
  int M, N, K;
  int act, bias_mul;
  int nsplit, nbw;  // workgroup (panel, s) owns the 32-column blocks [s*nbw, min(N/32, (s+1)*nbw))
};

__device__ __forceinline__ int xcd_run(int bid, int nwg) {
  // blocks are dealt round-robin over the 8 XCDs: give each XCD a contiguous run of work ids (bijective)
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + (bid >> 3);
}

// RB: 16-row blocks per panel; KT: K / 64 (compile-time: the K loop and the prologue's register image unroll)
template <typename T, int RB, int KT>
__global__ __launch_bounds__(LN_NW * 64, 2) void ln_gemm_kernel(const LnGemmParams p) {
  constexpr int BMP = RB * 16;       // panel rows
  constexpr int RPW = BMP / LN_NW;   // rows normalised by one wave
  constexpr int PASSES = RPW / 4;    // 4 rows (16 lanes each) per pass
  constexpr int TILE = BMP * 128;    // bytes of one 64-wide K-tile of the panel
  static_assert(RPW % 4 == 0, "rows per wave");
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fq = lane >> 4;

  const int wg = xcd_run(blockIdx.x, gridDim.x);
  const int pnl = wg / p.nsplit, ns = wg - pnl * p.nsplit;
  const int m0 = pnl * BMP;

  // ---------------- prologue: LayerNorm(X[m0 .. m0+BMP)) -> LDS panel ----------------
  // image: K-tile kt at kt*TILE, row r at r*128 inside it, logical 16-byte chunk c at slot c ^ (r & 7)
  {
    const int sub = lane & 15, rq = lane >> 4;
    float v[PASSES][KT][4];
#pragma unroll
    for (int ps = 0; ps < PASSES; ++ps) {
      int m = m0 + wave * RPW + ps * 4 + rq;
      if (m > p.M - 1) m = p.M - 1;
      const float* xp = p.X + (int64_t)m * p.ldx + sub * 4;
#pragma unroll
      for (int i = 0; i < KT; ++i) {
        const f32x4 t = *reinterpret_cast<const f32x4*>(xp + 64 * i);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[ps][i][e] = t[e];
      }
    }
    const float inv_c = 1.0f / (float)p.K;
#pragma unroll
    for (int ps = 0; ps < PASSES; ++ps) {
      __builtin_amdgcn_sched_barrier(0);
      float s = 0.f;
#pragma unroll
      for (int i = 0; i < KT; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) s += v[ps][i][e];
#pragma unroll
      for (int o = 8; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
      const float mean = s * inv_c;
      float ss = 0.f;
#pragma unroll
      for (int i = 0; i < KT; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float d = v[ps][i][e] - mean;
          ss = fmaf(d, d, ss);
        }
#pragma unroll
      for (int o = 8; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
      const float rstd = rsqrtf(ss * inv_c + p.eps);
      const int r = wave * RPW + ps * 4 + rq;
      char* dst = smem + r * 128 + (((sub >> 1) ^ (r & 7)) << 4) + (sub & 1) * 8;
#pragma unroll
      for (int i = 0; i < KT; ++i) {
        const f32x4 g = *reinterpret_cast<const f32x4*>(p.gamma + sub * 4 + 64 * i);
        const f32x4 b = *reinterpret_cast<const f32x4*>(p.beta + sub * 4 + 64 * i);
        Pack4<T> o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o.e[e] = from_f32<T>(fmaf((v[ps][i][e] - mean) * rstd, g[e], b[e]));
        *reinterpret_cast<u32x2*>(dst + i * TILE) = o.u;
      }
    }
  }
  __syncthreads();

  // ---------------- sweep over this workgroup's column blocks ----------------
  const bool geglu = p.act == PM_ACT_GEGLU;
  const int nb_all = p.N >> 5;
  const int b_lo = ns * p.nbw;
  const int b_hi = (b_lo + p.nbw < nb_all) ? b_lo + p.nbw : nb_all;
  int blk = b_lo + wave;
  if (blk >= b_hi) return;

  const char* const Wb = reinterpret_cast<const char*>(p.W);
  // lane (fr, fq) of MFMA block j holds W row n0 + wrow(j) (the column interleave of gemm.hip's epilogue_regs:
  // a lane ends up with 8 ADJACENT output columns; GEGLU keeps the packed [16 value | 16 gate] order), k chunk fq
  auto w_off = [&](int b, int j) -> uint32_t {
    const int n = b * 32 + (geglu ? j * 16 + fr : (fr >> 2) * 8 + j * 4 + (fr & 3));
    return (uint32_t)((int64_t)n * p.ldw * 2 + fq * 16);
  };
  // W fragments of one K-step: [ks][j]
  auto load_w = [&](Pack8<T> (&w)[2][2], uint32_t o0, uint32_t o1, int kt) {
    const char* b = Wb + kt * 128;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      w[ks][0].u = ld_global16(b + o0 + ks * 64);
      w[ks][1].u = ld_global16(b + o1 + ks * 64);
    }
  };
  const char* const a_base0 = smem + fr * 128 + ((fq ^ (fr & 7)) << 4);
  const char* const a_base1 = smem + fr * 128 + (((4 + fq) ^ (fr & 7)) << 4);

  uint32_t o0 = w_off(blk, 0), o1 = w_off(blk, 1);
  // W fragments: three register sets, K-step s of the wave's whole walk lives in set s % 3 and is requested two
  // steps ahead.  Two, because vmcnt retires in order: a load issued AFTER the epilogue's stores cannot be
  // consumed before those stores have completed (1-2 us under load), so the first two K-steps of the next block
  // are requested BEFORE the stores (dbg builds: one step ahead left the stores' latency on the critical path).
  Pack8<T> w0[2][2], w1[2][2], w2[2][2];
  load_w(w0, o0, o1, 0);
  load_w(w1, o0, o1, 1);
  // A fragments: ONE register set.  Pair i of a substep's MFMAs frees fa[i], the ds_read of the NEXT substep's
  // fragment i is issued right behind it and has the other 2*(RB-1) MFMAs to land (reads run one substep ahead
  // across K-steps, blocks and the epilogue: the panel is the same for every block).
  Pack8<T> fa[RB];
  auto read_a = [&](int i, const char* base, int kt) {
    fa[i].u = *reinterpret_cast<const u32x4*>(base + kt * TILE + i * 2048);
  };
#pragma unroll
  for (int i = 0; i < RB; ++i) read_a(i, a_base0, 0);

  // one column block whose K-step 0 lives in `wa` (then wb, wc, wa, ...); returns false after the wave's last block
  auto run_block = [&](Pack8<T> (&wa)[2][2], Pack8<T> (&wb)[2][2], Pack8<T> (&wc)[2][2]) -> bool {
    const int n0 = blk * 32;
    const int nxt = blk + LN_NW;
    const bool has_next = nxt < b_hi;
    const uint32_t no0 = has_next ? w_off(nxt, 0) : o0, no1 = has_next ? w_off(nxt, 1) : o1;
    // bias / scale of this lane's columns: requested now, used in the epilogue
    f32x4 bv0 = f32x4{0.f, 0.f, 0.f, 0.f}, bv1 = bv0;
    if (p.bias != nullptr) {
      const float* bp = p.bias + n0 + (geglu ? 4 * fq : 8 * fq);
      bv0 = *reinterpret_cast<const f32x4*>(bp);
      bv1 = *reinterpret_cast<const f32x4*>(bp + (geglu ? 16 : 4));
    }
    f32x4 acc[RB][2];
#pragma unroll
    for (int i = 0; i < RB; ++i) acc[i][0] = acc[i][1] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) {
      Pack8<T> (&cur)[2][2] = (kt % 3 == 0) ? wa : ((kt % 3 == 1) ? wb : wc);
      Pack8<T> (&pre)[2][2] = (kt % 3 == 0) ? wc : ((kt % 3 == 1) ? wa : wb);  // set of K-step kt + 2
      // Issue order pinned per substep (left alone, hipcc sinks every read to just before its first use and the
      // MFMAs wait out the LDS / L2 latency: 3x slower): the W loads first, then MFMA pair i followed by the
      // ds_read that refills fa[i].
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (!(LNG_DBG & 1)) {
        if (kt + 2 < KT)
          load_w(pre, o0, o1, kt + 2);
        else
          load_w(pre, no0, no1, kt + 2 - KT);  // (after the last block: a harmless re-read of this block)
      }
#pragma unroll
      for (int i = 0; i < RB; ++i) {
        acc[i][0] = mfma16(cur[0][0].v, fa[i].v, acc[i][0]);  // C^T: lane (fr, fq) holds row i*16+fr,
        acc[i][1] = mfma16(cur[0][1].v, fa[i].v, acc[i][1]);  // W rows 4*fq .. 4*fq+3 of block j
        if constexpr (!(LNG_DBG & 2)) read_a(i, a_base1, kt);
      }
      __builtin_amdgcn_sched_group_barrier(0x020, 4, 0);  // 4 VMEM reads
#pragma unroll
      for (int t = 0; t < RB; ++t) {
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);  // 2 MFMA
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);  // 1 DS read
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < RB; ++i) {
        acc[i][0] = mfma16(cur[1][0].v, fa[i].v, acc[i][0]);
        acc[i][1] = mfma16(cur[1][1].v, fa[i].v, acc[i][1]);
        if constexpr (!(LNG_DBG & 2)) read_a(i, a_base0, (kt + 1 < KT) ? kt + 1 : 0);  // next K-step / next block
      }
#pragma unroll
      for (int t = 0; t < RB; ++t) {
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    // ---- epilogue from registers
    T* const Cg = reinterpret_cast<T*>(p.C);
    if (geglu) {
      const int n = (n0 >> 1) + 4 * fq;
#pragma unroll
      for (int i = 0; i < RB; ++i) {
        const int m = m0 + i * 16 + fr;
        Pack4<T> ov;
#pragma unroll
        for (int r = 0; r < 4; ++r) ov.e[r] = from_f32<T>((acc[i][0][r] + bv0[r]) * gelu_erf_f(acc[i][1][r] + bv1[r]));
        if (m < p.M && (!(LNG_DBG & 4) || ov.u[0] == 0x12345678u)) *reinterpret_cast<u32x2*>(Cg + (int64_t)m * p.ldc + n) = ov.u;
      }
    } else {
      const int n = n0 + 8 * fq;
#pragma unroll
      for (int i = 0; i < RB; ++i) {
        const int m = m0 + i * 16 + fr;
        Pack8<T> ov;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          ov.e[r] = from_f32<T>(p.bias_mul ? acc[i][0][r] * bv0[r] : acc[i][0][r] + bv0[r]);
          ov.e[4 + r] = from_f32<T>(p.bias_mul ? acc[i][1][r] * bv1[r] : acc[i][1][r] + bv1[r]);
        }
        if (m < p.M && (!(LNG_DBG & 4) || ov.u[0] == 0x12345678u)) st_global16(Cg + (int64_t)m * p.ldc + n, ov.u);
      }
    }
    blk = nxt;
    o0 = no0;
    o1 = no1;
    return has_next;
  };

  // block b starts in register set (b * KT) % 3
  for (;;) {
    if (!run_block(w0, w1, w2)) break;
    if constexpr (KT % 3 == 1) {
      if (!run_block(w1, w2, w0)) break;
      if (!run_block(w2, w0, w1)) break;
    } else if constexpr (KT % 3 == 2) {
      if (!run_block(w2, w0, w1)) break;
      if (!run_block(w1, w2, w0)) break;
    }
  }
}

constexpr int LN_MAX_DEVICES = 64;

static int ln_num_cus() {
  static int cus[LN_MAX_DEVICES] = {0};
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev < 0 || dev >= LN_MAX_DEVICES) dev = 0;
  if (cus[dev] == 0) {
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 1) n = 256;
    cus[dev] = n;
  }
  return cus[dev];
}

// panel geometry for a K; 0 rows = shape not served
static int ln_gemm_rb(int64_t K) { return K == 320 ? 8 : 0; }  // 16-row blocks per panel (80 KiB: two workgroups per CU)

// Column split: the grid is (panels x nsplit) workgroups, two resident per CU; every workgroup pays the LayerNorm
// prologue (an HBM read of its 160 KiB of X: bandwidth, so it does not get cheaper with more splits), its slowest
// wave walks ceil(blocks / 4) column blocks.  Costs in cycles, fitted to tools/lngemm_bench.py sweeps over nsplit at
// both resolutions (profiles/r02/lngemm_v3_*.txt): a block costs more with plain stores (8 KiB) than GEGLU (4 KiB).
static int ln_choose_nsplit(int64_t M, int64_t N, int64_t K, int act, int* nbw_out) {
  const int rb = ln_gemm_rb(K);
  const int64_t bmp = rb * 16, nb = N / 32;
  const int64_t panels = (M + bmp - 1) / bmp;
  const double prologue = 20000.0;
  const double per_block = act == PM_ACT_GEGLU ? 10800.0 : 13000.0;
  const int64_t cus = ln_num_cus(), slots = 2 * cus;
  double best = 0.0;
  int best_s = 1, best_nbw = (int)nb;
  for (int s = 1; s <= 16 && s <= nb; ++s) {
    const int64_t nbw = (nb + s - 1) / s;
    if (nbw * (s - 1) >= nb) continue;  // an empty last range
    const double t_wg = prologue + (double)((nbw + LN_NW - 1) / LN_NW) * per_block;
    const int64_t wgs = panels * s, full = wgs / slots, rem = wgs - full * slots;
    // a last round of at most one workgroup per CU runs its waves alone on their SIMDs
    const double cost = (double)full * t_wg + (rem == 0 ? 0.0 : (rem <= cus ? 0.75 * t_wg : t_wg));
    if (best == 0.0 || cost < best * 0.98) {
      best = cost;
      best_s = s;
      best_nbw = (int)nbw;
    }
  }
  *nbw_out = best_nbw;
  return best_s;
}

template <typename T, int RB, int KT> static int launch_ln_gemm(const LnGemmParams& p, hipStream_t stream) {
  constexpr int lds = KT * RB * 16 * 128;
  static bool attr_set[LN_MAX_DEVICES] = {false};  // per device; idempotent
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev < 0 || dev >= LN_MAX_DEVICES) dev = 0;
  if (!attr_set[dev]) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(ln_gemm_kernel<T, RB, KT>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    attr_set[dev] = true;
  }
  const int panels = (p.M + RB * 16 - 1) / (RB * 16);
  hipLaunchKernelGGL((ln_gemm_kernel<T, RB, KT>), dim3(panels * p.nsplit), dim3(LN_NW * 64), lds, stream, p);
  return check_launch();
}

}  // namespace pm

using namespace pm;

extern "C" int pm_ln_gemm_supported(int64_t M, int64_t N, int64_t K, int act) {
  if (ln_gemm_rb(K) == 0 || M < 1 || N < 256 || (N & 31)) return 0;
  if (act != PM_ACT_NONE && act != PM_ACT_GEGLU) return 0;
  // a panel workgroup re-reads its N x K slice of W: worth it only with enough rows to amortise it over
  return M >= 4096 ? 1 : 0;
}

extern "C" int pm_ln_gemm(const float* X, int64_t ldx, const float* gamma, const float* beta, float eps,
                          const void* W, int64_t ldw, const float* bias, void* C, int64_t ldc, int64_t M,
                          int64_t N, int64_t K, int act, int flags, int dtype, void* stream) {
  if (!X || !gamma || !beta || !W || !C) return PM_E_NULL;
  if (ln_gemm_rb(K) == 0 || M < 1 || M > (1ll << 30) || N < 32 || (N & 31) || N > (1ll << 24)) return PM_E_SHAPE;
  if (act != PM_ACT_NONE && act != PM_ACT_GEGLU) return PM_E_SHAPE;
  if (flags & ~PM_FLAG_BIAS_IS_SCALE) return PM_E_SHAPE;
  if ((flags & PM_FLAG_BIAS_IS_SCALE) && (bias == nullptr || act == PM_ACT_GEGLU)) return PM_E_SHAPE;
  if ((ldx & 3) || ldx < K || (ldw & 7) || ldw < K || N * ldw * 2 >= (1ll << 32)) return PM_E_SHAPE;
  if (ldc & (act == PM_ACT_GEGLU ? 3 : 7)) return PM_E_SHAPE;
  LnGemmParams p{};
  p.X = X; p.ldx = ldx; p.gamma = gamma; p.beta = beta; p.eps = eps; p.W = W; p.ldw = ldw; p.bias = bias;
  p.C = C; p.ldc = ldc; p.M = (int)M; p.N = (int)N; p.K = (int)K; p.act = act;
  p.bias_mul = (flags & PM_FLAG_BIAS_IS_SCALE) ? 1 : 0;
  p.nsplit = ln_choose_nsplit(M, N, K, act, &p.nbw);
  const char* force = diag_env("PANDORA_LNGEMM_NSPLIT");  // tuning override (kernel choice only, never results)
  if (force && atoi(force) > 0 && atoi(force) <= N / 32) {
    p.nsplit = atoi(force);
    p.nbw = (int)((N / 32 + p.nsplit - 1) / p.nsplit);
    while ((int64_t)p.nbw * (p.nsplit - 1) >= N / 32) --p.nsplit;
  }
  PM_DISPATCH_DTYPE(dtype, T, return (launch_ln_gemm<T, 8, 5>(p, (hipStream_t)stream)));
}
