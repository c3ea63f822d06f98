// Shared by the GEMM-family kernel files (gemm.hip: 128x128 kernels; gemm256.hip: the 256x256 8-phase kernel):
// the call parameters, the XCD-aware block remap and the register-direct epilogue (bias / activation / residual /
// GroupNorm column sums / store, all from the accumulators with 16-byte accesses).
#pragma once
#include <stdlib.h>
#include <type_traits>
#include "common.hpp"

namespace pm {

enum { A_DENSE = 0, A_CONV3X3 = 1, A_CONVT3 = 2, A_CONV3X3_FAST = 3 };

struct GemmParams {
  const void* A;
  int64_t lda;  // dense: row stride; conv: elements per pixel
  const void* Wt;
  int64_t ldw;
  const float* bias;
  const void* R;
  int64_t ldr;
  void* C;
  int64_t ldc;
  int M, N, K;
  int act;
  int out32, res32;  // f32 output / f32 residual (independent)
  int a_lo;          // PM_FLAG_A_LO (f32 A only): stage a - round16(a), the part the plain pass rounds away
  int bias_mul;      // PM_FLAG_BIAS_IS_SCALE: `bias` multiplies the accumulator (per-column scale) instead of adding
  int kwrap;         // PM_FLAG_W_WRAP (dense): W has kwrap = K/2 columns, K-tile kt reads W columns (64 kt) mod kwrap (0: off)
  int ntiles, mtiles;
  int splits, ktps;  // split-K: number of K slices and K-tiles per slice
  float* ws;         // split-K partial slabs [splits][M][N] f32
  float* colstats;   // optional [ceil(M/64)][Nout][2]: column {sum, sum of squares} of every 64 output rows
  // r06 (PM_FLAG_STATS_I64): `colstats` is instead int64 GroupNorm totals [NI][32 groups][4] that the epilogue ADDS to (fixed-
  // point limbs, group_stats_add below); gs_ni = NI > 0 switches it on, gs_rows = rows per instance (a multiple of the block)
  int gs_ni, gs_rows;
  int natural;       // W rows staged in natural column order (see cperm): GEGLU, and the all-f32 epilogue flavours
  // conv3x3
  int Hin, Win, Hv, Wv, Cin, Ho, Wo, stride, ups, pad;  // pad: zero rows/cols before the image (1, or 0)
  // temporal conv: F frames of P pixels; clips of Fc frames (Fc divides F) are independent: zero padding at BOTH ends of each
  int F, P, Fc;
  const void* halo_lo;
  const void* halo_hi;
  const void* zero;
  int64_t a_bytes;  // conv modes: bytes of the input tensor reachable from A (bound of the buffer descriptor)
  int64_t tap_a[9], tap_w[9];  // fast 3x3 conv: per tap, byte shift of the A base and byte offset of the W K-tile (channel 0)
#ifdef PM_RING_PROF
  long long* prof;  // [grid][8 waves][4] cycle sums (tools/ring_prof.py)
#endif
};

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = BM * BK * 2;  // 16 KiB per operand tile
constexpr int STAGE_LD = 132;            // f32 staging row stride (floats)

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;

__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  // Blocks are dealt round-robin over the 8 XCDs; give each XCD a contiguous run of tiles so that
  // neighbouring tiles (same A row panel) share one L2.  Bijective for any nwg.
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + (bid >> 3);
}

// ---- GroupNorm totals by deterministic integer atomics (r06, VERDICT r05 #3a/b) -------------------------------------------------
// The fused statistics used to leave per-(row block, column) sums behind and every consumer GroupNorm paid a finalize launch
// (156-165 gn_finalize_colstats + 20-35 gn_finalize of ~1050 launches per forward).  A float atomic would make the totals depend
// on arrival order; these are exact instead: a partial sum p (f32) is split into two FIXED-POINT limbs, A = rint(p 2^12) and
// B = rint((p - A 2^-12) 2^44) (the remainder is exact in f32: |B| <= 2^31), and both are added with 64-bit INTEGER atomics -
// associative, so the totals are bit-identical for every arrival order, every launch and every rank, and carry p's 24 bits down
// to |p| = 2^-20 (absolute resolution 2^-44 below; range 2^50).  Totals layout: int64 [NI][groups][4][GS_STRIDE] = {sum A, sum B, sumsq A,
// sumsq B} (each in element 0 of its 64-byte sector); the consumer (gn_apply_kernel) rebuilds sum = A 2^-12 + B 2^-44 in f64.  The buffer is zeroed by the host (one
// memset per forward over the op table's arena, ops_hip.py).
// Every value sits in a 64-byte sector of its own (GS_STRIDE int64s apart): the memory-side atomic unit serialises adds per
// sector - with the 128 values of an instance packed into 1 KiB the (T,H,W) statistics of a level-0 conv (90 000 adds) cost ~9 us
// per producer, more than the finalize launch they replace (profiles/r06/stats_i64_ab.txt).
constexpr int GS_STRIDE = 8;
constexpr float GS_SCALE_A = 4096.0f;                 // 2^12
constexpr float GS_SCALE_B = 17592186044416.0f;       // 2^44
constexpr double GS_INV_A = 1.0 / 4096.0, GS_INV_B = 1.0 / 17592186044416.0;
// The MFMA kernels' epilogues carry the atomics only in the diagnostics build (PANDORA_STATS_I64=2, an A/B that lost: +1 ... +3 % per
// step, profiles/r06/stats_i64_ab.txt): compiled in, the path costs the 256x128 ring conv kernels 10 more spilled registers
// (13 -> 23) and 9 - 13 % of their time whether it is taken or not.  The split-K reduce pass and the statistics pass keep them.
constexpr bool GS_EPILOGUE = PM_DIAG_BUILD;
__device__ __forceinline__ void gs_atomic_add(long long* dst, float p) {
  const float a = rintf(p * GS_SCALE_A);
  const float r = fmaf(-a, 1.0f / GS_SCALE_A, p);  // exact: p minus its rounding to the 2^-12 grid
  const long long ia = (long long)a, ib = (long long)rintf(r * GS_SCALE_B);
  __hip_atomic_fetch_add(dst, ia, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_fetch_add(dst + GS_STRIDE, ib, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// One column per lane (s, q = its {sum, sum of squares}; columns in LANE ORDER, group id g non-decreasing over the lanes,
// g < 0 = no column): segmented sum over the lanes of a group in a fixed order, then the group's first lane adds the limbs.
__device__ __forceinline__ void gs_wave_add(long long* inst_base, float s, float q, int g, int lane) {
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const float ts = __shfl_down(s, off, 64), tq = __shfl_down(q, off, 64);
    const int tg = __shfl_down(g, off, 64);
    const bool same = (lane + off < 64) && tg == g;
    s += same ? ts : 0.f;
    q += same ? tq : 0.f;
  }
  const int gp = __shfl_up(g, 1, 64);
  if (g >= 0 && (lane == 0 || gp != g)) {
    gs_atomic_add(inst_base + (int64_t)g * 4 * GS_STRIDE, s);
    gs_atomic_add(inst_base + ((int64_t)g * 4 + 2) * GS_STRIDE, q);
  }
}
// Epilogue form: after the xor tree every lane of quad fq holds the 16 column sums cs / cq [2][8] of its quad (replicated over
// fr).  Lane (fr, fq) takes column index fr of them, the columns are permuted into lane order and reduced per group.
__device__ __forceinline__ void group_stats_add(const GemmParams& p, const float (&cs)[2][8], const float (&cq)[2][8], int ncol0,
                                                int nout, int row0, int fr, int fq, bool nat) {
  float s = cs[0][0], q = cq[0][0];
#pragma unroll
  for (int k = 1; k < 16; ++k) {
    s = (fr == k) ? cs[k >> 3][k & 7] : s;
    q = (fr == k) ? cq[k >> 3][k & 7] : q;
  }
  // this lane's column inside the wave's 64: run_col(jp, h, fq) + (e & 3) with jp = fr >> 3, e = fr & 7, h = e >> 2
  const int lane = fr + 16 * fq;
  // lane (fr, fq) holds column offset run_col(jp, h, fq) + (e & 3) with jp = fr >> 3, e = fr & 7, h = e >> 2 (a permutation of
  // 0..63); lane L wants offset L.  interleaved: offset = jp*32 + fq*8 + e -> holder of c: fq = (c >> 3) & 3, fr = (c >> 5)*8 + (c & 7)
  // natural: offset = jp*32 + h*16 + fq*4 + (e & 3) -> holder of c: fq = (c >> 2) & 3, fr = (c >> 5)*8 + ((c >> 4) & 1)*4 + (c & 3)
  const int c = lane;
  const int src = nat ? (((c >> 2) & 3) * 16 + (c >> 5) * 8 + ((c >> 4) & 1) * 4 + (c & 3)) : (((c >> 3) & 3) * 16 + (c >> 5) * 8 + (c & 7));
  s = __shfl(s, src, 64);
  q = __shfl(q, src, 64);
  const int n = ncol0 + c;
  const int cpg = nout >> 5;  // 32 groups
  const int g = (n < nout) ? n / cpg : -1;
  long long* base = reinterpret_cast<long long*>(p.colstats) + (int64_t)(row0 / p.gs_rows) * 32 * 4 * GS_STRIDE;
  gs_wave_add(base, s, q, g, lane);
}

// Register-direct epilogue shared by both GEMM kernels.  The MFMAs run with the operands swapped (W
// fragment as the MFMA "A" operand), so a wave's accumulator tile is C^T: lane (fr, fq) holds output row
// m = i*16 + fr and, for column block j, the 4 CONSECUTIVE columns 4*fq .. 4*fq+3.  With the loaders'
// column interleave (cperm below) a block pair (2jp, 2jp+1) gives 8 ADJACENT output columns per lane, so
// bias / activation / residual / statistics / store all happen in registers with 16-byte accesses: no LDS
// round trip, no barrier (in-kernel stamps: the LDS-staged epilogue cost ~8600 cycles per 128x128 tile,
// as much as 12 K-steps of the main loop).
//   LDS row pr of the W tile holds output column n0 + cperm(pr); GEGLU keeps the natural order (its
//   [16 value | 16 gate] weight packing IS the block pair).
//   All-f32 flavours (f32 output or split-K slabs, residual none or f32: GemmParams::natural, set by the host) keep the
//   natural order too: a lane's block j is then the 16-byte f32 run at column 16 j + 4 fq, and the four lanes fq = 0..3
//   of a row write (and read the residual as) 64 CONTIGUOUS bytes per instruction; with the interleave a lane owns 32
//   contiguous bytes that two instructions cover half each - every 64-byte request half used.
__device__ __forceinline__ int cperm(int pr, bool natural) {
  const int nn = pr & 15, jb = (pr >> 4) & 3;
  return natural ? pr : (pr & ~63) + (jb >> 1) * 32 + (nn >> 2) * 8 + (jb & 1) * 4 + (nn & 3);
}
// column offset (inside a wave's 64) of the 4-column run h (0 / 1) of block pair jp held by lane group fq
__device__ __forceinline__ int run_col(int jp, int h, int fq, bool natural) {
  return natural ? jp * 32 + h * 16 + fq * 4 : jp * 32 + fq * 8 + h * 4;
}
// bias of this lane's 16 output columns, bv[j][r] <-> column block j, column 4*fq + r in MFMA order
__device__ __forceinline__ void load_bias_regs(const GemmParams& p, float (&bv)[4][4], int n0, int wn, int fq) {
  const float* bias_p = (p.splits > 1) ? nullptr : p.bias;
  const bool geglu = p.act == PM_ACT_GEGLU || p.natural;
  // column of element (j, 0): the 4 elements r = 0..3 are consecutive columns -> one 16-byte load per j when the
  // run lies inside [0, N) (N % 4 == 0 on every shape of the path); clamped address + select otherwise
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int n = geglu ? n0 + wn * 64 + j * 16 + 4 * fq : n0 + wn * 64 + (j >> 1) * 32 + fq * 8 + (j & 1) * 4;
    if (bias_p != nullptr && (p.N & 3) == 0) {
      const bool ok = n + 4 <= p.N;
      const f32x4 t = *reinterpret_cast<const f32x4*>(bias_p + (ok ? n : 0));
#pragma unroll
      for (int r = 0; r < 4; ++r) bv[j][r] = ok ? t[r] : 0.f;
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r) bv[j][r] = (bias_p != nullptr && n + r < p.N) ? bias_p[n + r] : 0.f;
    }
  }
}
// Straight-line store pass of one epilogue flavour (OUT32: f32 output; RES: 0 none, 1 16-bit, 2 f32 residual).
// The generic epilogue below carries every flavour behind runtime branches, replicated 8x by the unrolled
// (row block, column pair) loops: ~6000 cycles per tile by the in-kernel stamps (branch-bound and far beyond
// the instruction cache).  Here the flavour is chosen ONCE per tile, addresses of dead lanes are clamped
// (so the residual loads of a half tile issue back to back, unconditionally) and only the stores are
// predicated.  Needs N % 8 == 0 and 16-byte-aligned rows (checked by the caller).
template <typename T, bool OUT32, int RES>
__device__ __forceinline__ void store_fast(const GemmParams& p, f32x4 (&acc)[4][4], void* cbase, int64_t ldc,
                                           int mrow0, int ncol0, int nout, int fr, int fq, bool stats,
                                           float (&cs)[2][8], float (&cq)[2][8]) {
  // f32 flavours in natural column order (GemmParams::natural, wave-uniform; N % 32 == 0 there, so the two runs of a pair
  // share one fate as the interleaved ones do): run 1 of pair jp sits 16 columns, not 4, behind run 0 (run_col)
  const bool nat = OUT32 && RES != 1 && p.natural != 0;
  const int hstep = nat ? 16 : 4;
  int ncl[2];
  bool nok[2];
#pragma unroll
  for (int jp = 0; jp < 2; ++jp) {
    const int n = ncol0 + run_col(jp, 0, fq, nat);
    nok[jp] = n < nout;
    ncl[jp] = nok[jp] ? n : 0;  // dead lanes: clamped addresses, predicated stores
  }
#pragma unroll
  for (int ih = 0; ih < 2; ++ih) {  // two row-block halves: bounds the residual registers in flight
    u32x4 r16[2][2];
    f32x4 r32[2][2][2];
    int64_t rowoff[2];
    bool mok[2];
#pragma unroll
    for (int ii = 0; ii < 2; ++ii) {
      const int m = mrow0 + (ih * 2 + ii) * 16 + fr;
      mok[ii] = m < p.M;
      rowoff[ii] = mok[ii] ? m : 0;
#pragma unroll
      for (int jp = 0; jp < 2; ++jp) {
        if constexpr (RES == 1) {
          r16[ii][jp] = ld_global16(reinterpret_cast<const T*>(p.R) + rowoff[ii] * p.ldr + ncl[jp]);
        } else if constexpr (RES == 2) {
          const float* rp = reinterpret_cast<const float*>(p.R) + rowoff[ii] * p.ldr + ncl[jp];
          r32[ii][jp][0] = *reinterpret_cast<const f32x4*>(rp);
          r32[ii][jp][1] = *reinterpret_cast<const f32x4*>(rp + hstep);
        }
      }
    }
#pragma unroll
    for (int ii = 0; ii < 2; ++ii) {
      const int i = ih * 2 + ii;
#pragma unroll
      for (int jp = 0; jp < 2; ++jp) {
        float v[8];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          v[r] = acc[i][2 * jp][r];
          v[4 + r] = acc[i][2 * jp + 1][r];
        }
        if constexpr (RES == 1) {
          Pack8<T> rv;
          rv.u = r16[ii][jp];
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] += to_f32(rv.e[e]);
        } else if constexpr (RES == 2) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            v[e] += r32[ii][jp][0][e];
            v[e + 4] += r32[ii][jp][1][e];
          }
        }
        const bool ok = mok[ii] && nok[jp];
        if (stats) {  // (wave-uniform)
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float x = ok ? v[e] : 0.f;
            cs[jp][e] += x;
            cq[jp][e] = fmaf(x, x, cq[jp][e]);
          }
        }
        if (ok) {
          if constexpr (OUT32) {
            float* cptr = reinterpret_cast<float*>(cbase) + rowoff[ii] * ldc + ncl[jp];
            *reinterpret_cast<f32x4*>(cptr) = f32x4{v[0], v[1], v[2], v[3]};
            *reinterpret_cast<f32x4*>(cptr + hstep) = f32x4{v[4], v[5], v[6], v[7]};
          } else {
            Pack8<T> ov;
#pragma unroll
            for (int e = 0; e < 8; ++e) ov.e[e] = from_f32<T>(v[e]);
            st_global16(reinterpret_cast<T*>(cbase) + rowoff[ii] * ldc + ncl[jp], ov.u);
          }
        }
      }
    }
  }
}

// Residual prefetch: when the epilogue is "+ bias, + residual, store" (no activation, unsplit, 8-column vectors),
// the accumulators START from the residual instead of zero: its loads are issued at tile start and land behind the
// K loop instead of stalling the epilogue (these GEMMs stream A, residual and output once: HBM-latency-bound), and
// the epilogue then skips the add.  Returns whether it did (the caller zero-fills otherwise).
// A wave's 64 x 64 piece that lies wholly outside the matrix (the second column half of the ragged last column tile of N = 320 /
// 960 on 128-wide tiles: a third / an eighth of all tiles there) has nothing to load or store: the callers skip its residual
// prefetch and its epilogue (r06: the generic epilogue walked its ~900 masked instructions - and its clamped residual loads -
// anyway, and a tile's epilogue ends with its slowest wave).
__device__ __forceinline__ bool piece_dead(const GemmParams& p, int m_p, int n_p) { return n_p >= p.N || m_p >= p.M; }

template <typename T, bool SKIP_DEAD = true>
__device__ __forceinline__ bool residual_into_acc(const GemmParams& p, f32x4 (&acc)[4][4], int m0, int n0, int wm,
                                                  int wn, int fr, int fq) {
  if (SKIP_DEAD && piece_dead(p, m0 + wm * 64, n0 + wn * 64)) return false;
  if (p.R == nullptr || p.splits > 1 || p.act != PM_ACT_NONE || p.bias_mul || (p.N & 7) || (p.ldc & 7) || (p.ldr & 7)) return false;
  const bool f32res = p.res32 != 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int m = m0 + wm * 64 + i * 16 + fr;
    if (m > p.M - 1) m = p.M - 1;
#pragma unroll
    for (int jp = 0; jp < 2; ++jp) {
      int n = n0 + wn * 64 + run_col(jp, 0, fq, p.natural != 0);
      if (p.natural) {  // (implies an f32 residual and N % 32 == 0: two 16-byte runs 16 columns apart, store_fast)
        int nb = n0 + wn * 64 + jp * 32;  // (dead column groups: clamped to the last one)
        if (nb > p.N - 32) nb = p.N - 32;
        const float* rp = reinterpret_cast<const float*>(p.R) + (int64_t)m * p.ldr + nb + fq * 4;
        acc[i][2 * jp] = *reinterpret_cast<const f32x4*>(rp);
        acc[i][2 * jp + 1] = *reinterpret_cast<const f32x4*>(rp + 16);
        continue;
      }
      if (n > p.N - 8) n = p.N - 8;
      if (f32res) {
        const float* rp = reinterpret_cast<const float*>(p.R) + (int64_t)m * p.ldr + n;
        acc[i][2 * jp] = *reinterpret_cast<const f32x4*>(rp);
        acc[i][2 * jp + 1] = *reinterpret_cast<const f32x4*>(rp + 4);
      } else {
        Pack8<T> rv;
        rv.u = ld_global16(reinterpret_cast<const T*>(p.R) + (int64_t)m * p.ldr + n);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          acc[i][2 * jp][r] = to_f32(rv.e[r]);
          acc[i][2 * jp + 1][r] = to_f32(rv.e[4 + r]);
        }
      }
    }
  }
  return true;
}

// rows m0 + wm*64 + i*16 + fr, columns n0 + wn*64 + ...; sblock = index of this wave's 64-row block in colstats
// ONLY_FAST: compile the fast flavours alone (the caller guarantees their preconditions: gemm256_wanted).  The generic
// flavour's exec-masked loads and branches are not only dead weight there: left un-waited on a skipped branch, its bias
// loads stay "pending" in hipcc's bookkeeping on the K loop's back edge and cost an s_waitcnt vmcnt(0) inside the loop.
template <typename T, bool FAST = true, bool ONLY_FAST = false>
__device__ __forceinline__ void epilogue_regs(const GemmParams& p, f32x4 (&acc)[4][4], const float (&bv)[4][4],
                                              int m0, int n0, int wm, int wn, int fr, int fq, int sblock,
                                              int split, bool res_done = false) {
  const bool partial = p.splits > 1;
  // The MFMAs ran with the operands swapped (W fragment as "A"), so the accumulator tile is C^T: lane
  // (fr, fq) holds row m = i*16 + fr and, for column block j, the 4 CONSECUTIVE columns 4*fq .. 4*fq+3.
  // With the loader's column interleave a block pair (2jp, 2jp+1) gives 8 adjacent columns per lane:
  // bias / activation / residual / statistics / store all happen in registers with 16-byte accesses -
  // no LDS round trip, no barrier (the LDS-staged epilogue cost ~8600 cycles per tile = 12 K-steps).
  const int act = partial ? PM_ACT_NONE : p.act;
  const bool geglu = (act == PM_ACT_GEGLU);
  const int nout = geglu ? (p.N >> 1) : p.N;
  T* __restrict__ Cg = reinterpret_cast<T*>(p.C);
  const T* __restrict__ Rg = (partial || p.res32 || res_done) ? nullptr : reinterpret_cast<const T*>(p.R);
  float* __restrict__ Cf = partial ? p.ws + (int64_t)split * p.M * p.N : reinterpret_cast<float*>(p.C);
  const float* __restrict__ Rf = (partial || !p.res32 || res_done) ? nullptr : reinterpret_cast<const float*>(p.R);  // (see ONLY_FAST)
  const bool out32 = partial || p.out32 != 0;
  const int64_t ldc = partial ? p.N : p.ldc;
  // ONLY_FAST callers (gemm256) also drop the statistics and the f32-residual flavours (their host check): less live state
  const bool want_stats = !ONLY_FAST && (p.colstats != nullptr) && !partial;
  if constexpr (ONLY_FAST) Rf = nullptr;
  // ---- fast flavours (every shape of the U-Net except N % 8 != 0, i.e. the 4-channel output conv) ----
  if (FAST && !geglu && (nout & 7) == 0 && (ldc & 7) == 0 && (p.R == nullptr || (p.ldr & 7) == 0)) {
    // bias and activation in place (one uniform branch per activation, not per element group)
    if (p.bias_mul && !partial) {  // (uniform) per-column scale, e.g. the softmax scale on the q third of a q|k|v projection
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[i][j][r] *= bv[j][r];
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[i][j][r] += bv[j][r];
    }
    if (act == PM_ACT_SILU) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[i][j][r] = silu_f(acc[i][j][r]);
    } else if (act == PM_ACT_GELU) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[i][j][r] = gelu_erf_f(acc[i][j][r]);
    }
    float cs[2][8], cq[2][8];
#pragma unroll
    for (int jp = 0; jp < 2; ++jp)
#pragma unroll
      for (int e = 0; e < 8; ++e) cs[jp][e] = cq[jp][e] = 0.f;
    const int mrow0 = m0 + wm * 64, ncol0 = n0 + wn * 64;
    void* cb = out32 ? static_cast<void*>(Cf) : static_cast<void*>(Cg);
    if (out32) {
      if (Rf != nullptr)
        store_fast<T, true, 2>(p, acc, cb, ldc, mrow0, ncol0, nout, fr, fq, want_stats, cs, cq);
      else
        store_fast<T, true, 0>(p, acc, cb, ldc, mrow0, ncol0, nout, fr, fq, want_stats, cs, cq);
    } else {
      if (Rf != nullptr)
        store_fast<T, false, 2>(p, acc, cb, ldc, mrow0, ncol0, nout, fr, fq, want_stats, cs, cq);
      else if (Rg != nullptr)
        store_fast<T, false, 1>(p, acc, cb, ldc, mrow0, ncol0, nout, fr, fq, want_stats, cs, cq);
      else
        store_fast<T, false, 0>(p, acc, cb, ldc, mrow0, ncol0, nout, fr, fq, want_stats, cs, cq);
    }
    if (want_stats) {  // column sums of this wave's 64 rows: in-lane over i, then a fixed xor tree over fr
#pragma unroll
      for (int jp = 0; jp < 2; ++jp)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
#pragma unroll
          for (int o = 8; o > 0; o >>= 1) {
            cs[jp][e] += __shfl_xor(cs[jp][e], o, 64);
            cq[jp][e] += __shfl_xor(cq[jp][e], o, 64);
          }
        }
      if (GS_EPILOGUE && p.gs_ni > 0) {  // (r06: group totals by integer atomics, no finalize launch; diagnostics build)
        if (sblock * 64 < p.M)
          group_stats_add(p, cs, cq, ncol0, nout, sblock * 64, fr, fq, out32 && Rg == nullptr && p.natural != 0);
      } else if (fr == 0 && sblock * 64 < p.M) {  // (a ragged last tile has no second block)
        const bool nat = out32 && Rg == nullptr && p.natural != 0;  // (as store_fast)
#pragma unroll
        for (int jp = 0; jp < 2; ++jp)
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const int n = ncol0 + run_col(jp, h, fq, nat);
            if (n < nout) {
              float* dst = p.colstats + ((int64_t)sblock * nout + n) * 2;
#pragma unroll
              for (int e = 4 * h; e < 4 * h + 4; e += 2)
                *reinterpret_cast<f32x4*>(dst + 2 * (e - 4 * h)) = f32x4{cs[jp][e], cq[jp][e], cs[jp][e + 1], cq[jp][e + 1]};
            }
          }
      }
    }
    return;
  }
  if (FAST && geglu && (nout & 3) == 0 && (ldc & 3) == 0) {
    // fast GEGLU flavour: value block 2jj, gate block 2jj+1 (weights packed [16 value | 16 gate]); a lane owns
    // 4 adjacent output columns per pair: straight-line bias, erf-GELU gate, product, 8-byte stores
    const int mrow0 = m0 + wm * 64;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int m = mrow0 + i * 16 + fr;
      const int64_t mrow = m < p.M ? m : 0;
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) {
        const int n = (n0 >> 1) + wn * 32 + jj * 16 + 4 * fq;
        Pack4<T> ov;
#pragma unroll
        for (int r = 0; r < 4; ++r)
          ov.e[r] = from_f32<T>((acc[i][2 * jj][r] + bv[2 * jj][r]) * gelu_erf_f(acc[i][2 * jj + 1][r] + bv[2 * jj + 1][r]));
        if (m < p.M && n < nout) *reinterpret_cast<u32x2*>(Cg + mrow * ldc + n) = ov.u;
      }
    }
    return;
  }
  if constexpr (ONLY_FAST) return;
  if (geglu) {
    // value block 2jj, gate block 2jj+1 (weights packed [16 value | 16 gate]): 4 output columns per lane
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int m = m0 + wm * 64 + i * 16 + fr;
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) {
        const int n = (n0 >> 1) + wn * 32 + jj * 16 + 4 * fq;
        if (m < p.M && n < nout) {
          Pack4<T> ov;
#pragma unroll
          for (int r = 0; r < 4; ++r)
            ov.e[r] = from_f32<T>((acc[i][2 * jj][r] + bv[2 * jj][r]) * gelu_erf_f(acc[i][2 * jj + 1][r] + bv[2 * jj + 1][r]));
          T* cptr = Cg + (int64_t)m * ldc + n;
          if (n + 4 <= nout && ((ldc & 3) == 0))
            *reinterpret_cast<u32x2*>(cptr) = ov.u;
          else
            for (int e = 0; e < 4 && n + e < nout; ++e) cptr[e] = ov.e[e];
        }
      }
    }
  } else {
    float cs[2][8], cq[2][8];
#pragma unroll
    for (int jp = 0; jp < 2; ++jp)
#pragma unroll
      for (int e = 0; e < 8; ++e) cs[jp][e] = cq[jp][e] = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int m = m0 + wm * 64 + i * 16 + fr;
#pragma unroll
      for (int jp = 0; jp < 2; ++jp) {
        const int n = n0 + wn * 64 + jp * 32 + fq * 8;
        if (m < p.M && n < nout) {
          float v[8];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            v[r] = (p.bias_mul && !partial) ? acc[i][2 * jp][r] * bv[2 * jp][r] : acc[i][2 * jp][r] + bv[2 * jp][r];
            v[4 + r] = (p.bias_mul && !partial) ? acc[i][2 * jp + 1][r] * bv[2 * jp + 1][r]
                                                : acc[i][2 * jp + 1][r] + bv[2 * jp + 1][r];
          }
          if (act == PM_ACT_SILU) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = silu_f(v[e]);
          } else if (act == PM_ACT_GELU) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = gelu_erf_f(v[e]);
          }
          const bool fullr = (n + 8 <= nout);
          if (Rf != nullptr) {
            const float* rptr = Rf + (int64_t)m * p.ldr + n;
            if (fullr && ((p.ldr & 3) == 0)) {
              const f32x4 r0 = *reinterpret_cast<const f32x4*>(rptr);
              const f32x4 r1 = *reinterpret_cast<const f32x4*>(rptr + 4);
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                v[e] += r0[e];
                v[e + 4] += r1[e];
              }
            } else {
              for (int e = 0; e < 8 && n + e < nout; ++e) v[e] += rptr[e];
            }
          } else if (Rg != nullptr) {
            const T* rptr = Rg + (int64_t)m * p.ldr + n;
            if (fullr && ((p.ldr & 7) == 0)) {
              Pack8<T> rv;
              rv.u = ld_global16(rptr);
#pragma unroll
              for (int e = 0; e < 8; ++e) v[e] += to_f32(rv.e[e]);
            } else {
              for (int e = 0; e < 8 && n + e < nout; ++e) v[e] += to_f32(rptr[e]);
            }
          }
          if (want_stats) {
#pragma unroll
            for (int e = 0; e < 8; ++e)
              if (n + e < nout) {
                cs[jp][e] += v[e];
                cq[jp][e] = fmaf(v[e], v[e], cq[jp][e]);
              }
          }
          if (out32) {
            float* cptr = Cf + (int64_t)m * ldc + n;
            if (fullr && ((ldc & 3) == 0)) {
              *reinterpret_cast<f32x4*>(cptr) = f32x4{v[0], v[1], v[2], v[3]};
              *reinterpret_cast<f32x4*>(cptr + 4) = f32x4{v[4], v[5], v[6], v[7]};
            } else {
              for (int e = 0; e < 8 && n + e < nout; ++e) cptr[e] = v[e];
            }
          } else {
            T* cptr = Cg + (int64_t)m * ldc + n;
            if (fullr && ((ldc & 7) == 0)) {
              Pack8<T> ov;
#pragma unroll
              for (int e = 0; e < 8; ++e) ov.e[e] = from_f32<T>(v[e]);
              st_global16(cptr, ov.u);
            } else {
              for (int e = 0; e < 8 && n + e < nout; ++e) cptr[e] = from_f32<T>(v[e]);
            }
          }
        }
      }
    }
    if (want_stats) {  // column sums of this wave's 64 rows: in-lane over i, then a fixed xor tree over fr
#pragma unroll
      for (int jp = 0; jp < 2; ++jp)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
#pragma unroll
          for (int o = 8; o > 0; o >>= 1) {
            cs[jp][e] += __shfl_xor(cs[jp][e], o, 64);
            cq[jp][e] += __shfl_xor(cq[jp][e], o, 64);
          }
        }
      if (GS_EPILOGUE && p.gs_ni > 0) {
        if (sblock * 64 < p.M) group_stats_add(p, cs, cq, n0 + wn * 64, nout, sblock * 64, fr, fq, false);
      } else if (fr == 0 && sblock * 64 < p.M) {  // (a ragged last tile has no second block)
#pragma unroll
        for (int jp = 0; jp < 2; ++jp) {
          const int n = n0 + wn * 64 + jp * 32 + fq * 8;
          float* dst = p.colstats + ((int64_t)sblock * nout + n) * 2;
          for (int e = 0; e < 8 && n + e < nout; ++e) {
            dst[2 * e] = cs[jp][e];
            dst[2 * e + 1] = cq[jp][e];
          }
        }
      }
    }
  }
}

// LEAN epilogue of the 16-bit projection flavours (r06).  In-kernel stamps of gemm_wide with every store masked off
// (profiles/r06/wide_kernel_stamps.txt): epilogue_regs above costs ~3700 cycles of pure instruction issue per 64 x 64 piece and wave
// - ~900 issue slots for 64 values per lane: every flavour lives behind runtime branches, every store behind its own exec-mask
// branch, every row address is a 64-bit multiply, and the kernel's spilled scalars come back through v_readlane chains.  The
// flavours that make up the wide projections (q|k|v with its per-column scale, GEGLU, plain / bias; 16-bit output, no residual, no
// statistics, unsplit) need none of that when the piece lies wholly inside the matrix: one base pointer per lane, a constant row
// step, straight-line arithmetic, unconditional 16-byte (GEGLU: 8-byte) stores.  Caller checks `epilogue_lean16_ok`.
__device__ __forceinline__ bool epilogue_lean16_ok(const GemmParams& p, int m_p, int n_p) {
  if (p.splits > 1 || p.out32 || p.R != nullptr || p.colstats != nullptr) return false;
  if (m_p + 64 > p.M || n_p + 64 > p.N) return false;
  if (p.act == PM_ACT_GEGLU) return (p.ldc & 3) == 0 && (p.N & 7) == 0;
  return p.act == PM_ACT_NONE && !p.natural && (p.ldc & 7) == 0;
}
template <typename T, bool NOSTORE = false>  // (NOSTORE: diagnostics, timing only)
__device__ __forceinline__ void epilogue_lean16(const GemmParams& p, f32x4 (&acc)[4][4], const float (&bv)[4][4], int m_p, int n_p,
                                                int fr, int fq) {
  const int64_t rstep = (int64_t)16 * p.ldc;
  if (p.act == PM_ACT_GEGLU) {
    // value block 2jj, gate block 2jj+1 (weights packed [16 value | 16 gate]): 4 adjacent output columns per lane and pair
    T* rp = reinterpret_cast<T*>(p.C) + (int64_t)(m_p + fr) * p.ldc + (n_p >> 1) + 4 * fq;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) {
        Pack4<T> ov;
#pragma unroll
        for (int r = 0; r < 4; ++r)
          ov.e[r] = from_f32<T>((acc[i][2 * jj][r] + bv[2 * jj][r]) * gelu_erf_f(acc[i][2 * jj + 1][r] + bv[2 * jj + 1][r]));
        *reinterpret_cast<u32x2*>(rp + jj * 16) = ov.u;
      }
      rp += rstep;
    }
    return;
  }
  // interleaved W rows (cperm): the block pair (2jp, 2jp+1) is 8 ADJACENT columns jp*32 + fq*8 .. +7
  T* rp = reinterpret_cast<T*>(p.C) + (int64_t)(m_p + fr) * p.ldc + n_p + 8 * fq;
  if (p.bias_mul) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[i][j][r] *= bv[j][r];
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[i][j][r] += bv[j][r];
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
#pragma unroll
    for (int jp = 0; jp < 2; ++jp) {
      Pack8<T> ov;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        ov.e[r] = from_f32<T>(acc[i][2 * jp][r]);
        ov.e[4 + r] = from_f32<T>(acc[i][2 * jp + 1][r]);
      }
      if constexpr (NOSTORE)
        asm volatile("" ::"v"(ov.u), "v"(rp));
      else
        st_global16(rp + jp * 32, ov.u);
    }
    rp += rstep;
  }
}

// LEAN epilogue of the all-f32 flavours (r06): f32 output (or a split-K slab) in natural column order, bias add, no activation, f32
// residual none or already inside the accumulators (residual_into_acc), GroupNorm column statistics optional - i.e. what every
// 3x3 / temporal conv and every out-projection onto the residual stream ends with.  Same idea as epilogue_lean16: the piece lies
// wholly inside the matrix, so one base pointer per lane, a constant row step, unconditional 16-byte accesses (lane (fr, fq), block
// j <-> the 4 floats at column 16 j + 4 fq: the four fq lanes of a row cover 64 contiguous bytes per instruction).
__device__ __forceinline__ bool epilogue_lean32_ok(const GemmParams& p, int m_p, int n_p, bool res_done) {
  if (!p.natural || m_p + 64 > p.M || n_p + 64 > p.N) return false;
  if (p.splits > 1) return (p.N & 3) == 0;
  if (!p.out32 || p.act != PM_ACT_NONE || p.bias_mul || (p.ldc & 3)) return false;
  return p.R == nullptr || res_done;  // (a residual not yet inside the accumulators stays with epilogue_regs: its 32 registers of
                                      //  loads in flight on top of the statistics push the callers over 256 VGPRs)
}
template <typename T>
__device__ __forceinline__ void epilogue_lean32(const GemmParams& p, f32x4 (&acc)[4][4], const float (&bv)[4][4], int m_p, int n_p,
                                                int fr, int fq, int sblock, int split) {
  const bool partial = p.splits > 1;
  const int64_t ldc = partial ? p.N : p.ldc;
  float* cp = (partial ? p.ws + (int64_t)split * p.M * p.N : reinterpret_cast<float*>(p.C)) + (int64_t)(m_p + fr) * ldc + n_p + 4 * fq;
  const int64_t cstep = 16 * ldc;
  const bool stats = !partial && p.colstats != nullptr;
  float cs[2][8], cq[2][8];
#pragma unroll
  for (int jp = 0; jp < 2; ++jp)
#pragma unroll
    for (int e = 0; e < 8; ++e) cs[jp][e] = cq[jp][e] = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      f32x4 v;
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = acc[i][j][r] + bv[j][r];
      if (stats) {  // (wave-uniform) cs[jp][4 h + r] <-> block 2 jp + h: run_col's natural layout
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          cs[j >> 1][4 * (j & 1) + r] += v[r];
          cq[j >> 1][4 * (j & 1) + r] = fmaf(v[r], v[r], cq[j >> 1][4 * (j & 1) + r]);
        }
      }
      *reinterpret_cast<f32x4*>(cp + j * 16) = v;
    }
    cp += cstep;
  }
  if (stats) {  // column sums of this wave's 64 rows: a fixed xor tree over fr (as epilogue_regs)
#pragma unroll
    for (int jp = 0; jp < 2; ++jp)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) {
          cs[jp][e] += __shfl_xor(cs[jp][e], o, 64);
          cq[jp][e] += __shfl_xor(cq[jp][e], o, 64);
        }
      }
    if (GS_EPILOGUE && p.gs_ni > 0) {
      group_stats_add(p, cs, cq, n_p, p.N, sblock * 64, fr, fq, true);
    } else if (fr == 0) {
#pragma unroll
      for (int jp = 0; jp < 2; ++jp)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          float* dst = p.colstats + ((int64_t)sblock * p.N + n_p + run_col(jp, h, fq, true)) * 2;
#pragma unroll
          for (int e = 4 * h; e < 4 * h + 4; e += 2)
            *reinterpret_cast<f32x4*>(dst + 2 * (e - 4 * h)) = f32x4{cs[jp][e], cq[jp][e], cs[jp][e + 1], cq[jp][e + 1]};
        }
    }
  }
}

// gemm256.hip: the 256x256 8-phase kernel for large dense shapes
bool gemm256_wanted(const GemmParams& p, int flags, int num_cus);
template <typename T> int launch_gemm256(const GemmParams& p, int num_cus, hipStream_t stream);

// gemm_wide.hip: 256x256 tile, four waves of 128x128, assembly main loop (r06)
bool gemm_wide_wanted(const GemmParams& p, int flags, int num_cus);
template <typename T> int launch_gemm_wide(const GemmParams& p, int num_cus, hipStream_t stream);

// gemm_wide_stream.hip: the same tile as ONE assembly statement per workgroup (continuous K stream, assembly epilogue; r06)
bool gemm_wide_stream_wanted(const GemmParams& p, int flags, int num_cus);
template <typename T> int launch_gemm_wide_stream(const GemmParams& p, int num_cus, hipStream_t stream);

}  // namespace pm
