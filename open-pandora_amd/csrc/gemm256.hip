// 256x256x64 dense GEMM for the large MFMA-bound shapes of the U-Net (the GEGLU ff1, q|k|v and ff2 projections at
// M >= a few thousand rows): 8 waves, ping-pong phases, one long DMA pipeline across tiles.
//
//   C[M, N] = epi(A[M, K] . W[N, K]^T), 16-bit operands, f32 accumulate on v_mfma_f32_16x16x32_{f16,bf16}
//
// Why another kernel: a 128x128 tile needs 64 B of operands per MFMA cycle at full rate - more than a CU ingests -
// and its K-step (512 matrix cycles per wave) is shorter than a DMA round trip, so the 2-stage kernel waits for memory
// every step and the ring kernel idles during every epilogue (profiles/r03/negative_result_ring2_*).  Here a workgroup
// owns a 256x256 tile: 32 B per MFMA cycle, and a K-tile is 1024 matrix cycles per wave - as long as the round trip.
//
// Structure (per workgroup = per CU, 512 threads, 128 KiB of LDS):
//   * wave (wr, wc) = (wave >> 2, wave & 3) owns output rows wr*128 .. +128, columns wc*64 .. +64: 128 accumulator
//     registers, swapped operands (C^T) and the column interleave of epilogue_regs, so the epilogue is the register-
//     direct one of the 128x128 kernels, called once per 64-row half;
//   * a K-tile is computed in four PHASES, one 64x32 quadrant (rh, ch) of the wave's tile each (16 MFMAs), in the
//     order (0,0) (0,1) (1,1) (1,0): the A fragments of a row half serve two phases, and BOTH column halves' B
//     fragments stay in registers for the whole K-tile (32 registers), so every half-tile is read in ONE phase only;
//   * LDS holds two K-tiles (double buffer) of four HALF-TILES each: A-half Y = the rows every wave needs for rh = Y
//     (wr*128 + Y*64 .. +64 for both wr), B-half X = the columns every wave needs for ch = X - so quadrant (rh, ch)
//     reads exactly A-half rh and B-half ch, and the half-tiles of the next K-tiles can be staged one per phase, each as
//     soon as its buffer's last reader is done:  phase 0 stages B1 of tile g+1, phase 1 A1 of g+1, phase 2 A0 of g+2,
//     phase 3 B0 of g+2 (g = the K-tile being computed).  Every half-tile is issued >= 5 phases before its first read:
//     one counted s_waitcnt vmcnt(6) per phase - the DMAs of the three youngest phases (48 KiB per CU) stay in flight
//     across the barriers - never 0 before the last K-tiles of the workgroup;
//   * PING-PONG: waves 4-7 run half a phase behind waves 0-3 (one extra barrier at the start).  A phase is
//     [fragment reads + 2 DMA issues | barrier | 16 MFMAs | barrier], so on every SIMD one wave issues its MFMA cluster
//     while its partner reads LDS and issues DMAs: the matrix pipe sees back-to-back clusters;
//   * the K-tile sequence g runs over ALL tiles of the workgroup (persistent walk, XCD-aware supertile order): the
//     pipeline never drains, the next tile's operands stream in during the epilogue (registers only, no barrier).
// Hazards, by half-phase h (a barrier interval; waves 0-3 load in h = 2p and compute in 2p+1, waves 4-7 one later):
// the reads of A0, B0 / B1 / A1 of a buffer complete by h0+2 / +4 / +6 (h0 = 8g), their re-staging is issued at
// h0+4, h0+6 / h0+8 / h0+10 or later; at the top of phase p every wave waits for its own DMAs of phases <= p-4, a
// half-tile is first read >= 5 phases after it was issued, so the OTHER wave group's share was waited for at its phase
// p-1 at the latest, with a barrier in between.  Once a staging slot finds nothing left to fetch (the last K-tiles of
// the workgroup) the counted wait would under-wait: from then on the waits are vmcnt(0).
#include "gemm_common.hpp"

// r04: since gemm_ringw_kernel exists this kernel only took the K < 512 whole-round launches (1 launch
// of an 80-ms forward at 576x1024, 0.2-0.4 ms per step: -0.1 / -0.7 % in its own same-box A/B) - not worth 340 lines and a
// dispatch rule in the shipped library.  It is RETIRED from it: compiled only into the diagnostics build (-DPM_DIAG,
// PANDORA_GEMM256=1 / 2 there), kept as the measured record of the 256x256 8-phase design.
#ifndef PM_DIAG
namespace pm {
bool gemm256_wanted(const GemmParams&, int, int) { return false; }
template <typename T> int launch_gemm256(const GemmParams&, int, hipStream_t) { return PM_E_SHAPE; }
template int launch_gemm256<f16>(const GemmParams&, int, hipStream_t);
template int launch_gemm256<bf16>(const GemmParams&, int, hipStream_t);
}  // namespace pm
#else

namespace pm {

constexpr int BM2 = 256, BN2 = 256;
constexpr int HALF_BYTES = 128 * 128;       // one half-tile: 128 rows x 64 k x 2 B
constexpr int DBUF_BYTES = 4 * HALF_BYTES;  // A0 A1 B0 B1
constexpr int LDS256 = 2 * DBUF_BYTES;      // 128 KiB

// LDS-DMA (16 bytes per lane, 1 KiB per wave: LDS destination = M0 + 16 * lane) issued through inline asm: the wave that
// stages operands here also READS LDS, and for a __builtin_amdgcn_global_load_lds beside its own ds_reads hipcc inserts
// s_waitcnt vmcnt(0) in the load segments (it cannot tell the DMA's LDS bytes from the fragments') - draining the DMA
// pipeline twice per K-tile.  An asm statement is outside the compiler's bookkeeping: every wait for these transfers is
// the counted one at the top of a phase.  (M0 is compiler-reserved: saved and restored inside the statement.)
__device__ __forceinline__ void glds16_asm(const void* sbase, uint32_t voff, uint32_t lds_dst) {
  // source = uniform 64-bit base (SGPR pair) + this lane's 32-bit byte offset: no 64-bit per-lane pointers to keep alive
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(voff), "s"(sbase), "s"(lds_dst)
      : "memory");
}

template <typename T>
__global__ __launch_bounds__(512, 1) void gemm256_kernel(const GemmParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int fr = lane & 15, fq = lane >> 4;
  const int G = gridDim.x;
  const int nk = p.K / BK;
  const int ntiles_total = p.mtiles * p.ntiles;
  const bool geglu = p.act == PM_ACT_GEGLU;
  const bool wnat = geglu || p.natural;  // (W rows in natural column order: gemm_common.hpp cperm)

  // ---- work walk (the ring kernel's): workgroups are dealt round-robin over the 8 XCDs (blockIdx % 8), each with its
  // own L2.  XCD x owns the CONTIGUOUS run [x*per, (x+1)*per) of the supertile-ordered work list and its G/8 workgroups
  // sweep it in rounds of G/8 consecutive ids, so tiles that run together AND the tiles of consecutive rounds share A / W
  // panels in that L2.  (Measured neutral against tile t = slot + t*G: the kernel is not bound by where its operands
  // come from, profiles/r03/gemm256_ab.txt.)
  const bool xcd_walk = (G & 7) == 0 && ntiles_total > G;
  const int per_xcd = (ntiles_total + 7) >> 3, gx = G >> 3;
  const int slot_id = xcd_remap(blockIdx.x, G);  // (fallback walk: round t -> work item t*G + slot_id)
  int my_tiles = 0;
  if (xcd_walk) {
    const int xcd = blockIdx.x & 7, first = blockIdx.x >> 3;
    int avail = ntiles_total - xcd * per_xcd;
    avail = avail < per_xcd ? avail : per_xcd;
    if (avail > first) my_tiles = (avail - first + gx - 1) / gx;
  } else if (slot_id < ntiles_total) {
    my_tiles = (ntiles_total - slot_id + G - 1) / G;
  }
  auto decode = [&](int t, int& m0, int& n0) {
    constexpr int GM = 8;
    const int w = xcd_walk ? (blockIdx.x & 7) * per_xcd + t * gx + (blockIdx.x >> 3) : slot_id + t * G;
    const int grp = w / (GM * p.ntiles);
    const int first_m = grp * GM;
    const int gm = (p.mtiles - first_m < GM) ? p.mtiles - first_m : GM;
    const int rin = w - grp * GM * p.ntiles;
    const int nt = rin / gm;
    m0 = (first_m + (rin - nt * gm)) * BM2;
    n0 = nt * BN2;
  };
  const int total_g = my_tiles * nk;  // K-tiles this workgroup walks, across all its tiles

  // ---- staging: four STREAMS (A0, A1, B0, B1), each walking the workgroup's K-tile sequence on its own cursor ----
  // A stream keeps (tile, K-tile, buffer parity, tile origin) as scalars; a lane's rows are origin + a per-lane
  // constant, so issuing a half-tile is a clamp, one 64-bit multiply-add and one DMA per row, and only a tile change
  // pays the decode.  (Recomputing tile / row / address per call - integer divisions and 64-bit multiplies in every
  // phase - made the load segment ~3x the 256 matrix cycles it has to hide behind.)
  const char* const Ab = reinterpret_cast<const char*>(p.A);
  const char* const Wb = reinterpret_cast<const char*>(p.Wt);
  const uint32_t lds_base = (uint32_t)reinterpret_cast<uintptr_t>((lds_void*)smem);  // LDS byte address of the arena
  const int srow = wave * 8 + (lane >> 3);     // LDS row (+ 64 j) this lane fills; its physical 16-byte slot is lane & 7
  const int lc = (lane & 7) ^ (lane >> 3);     // logical k-chunk fetched into it (source-side swizzle: row & 7 == lane >> 3)
  struct Stream {  // scalars only: (tile, K-tile, buffer parity, first row / column of the tile)
    int t, kt, par, origin;
  };
  bool tail = false;  // a staging slot has found nothing to fetch: counted waits would under-wait from here on
  // per-lane constants of the staged rows: A rows srow + 128 j (+ 64 Y for half Y); W columns cperm(...) (+ 32 X)
  const int cb0 = cperm((srow >> 5) * 64 + (srow & 31), wnat), cb1 = cperm((2 + (srow >> 5)) * 64 + (srow & 31), wnat);
  const int64_t lda2 = p.lda * 2, ldw2 = p.ldw * 2;
  const uint32_t lc16 = (uint32_t)lc * 16;
  auto stream_origin = [&](Stream& s, bool is_a) {
    int m0, n0;
    decode(s.t, m0, n0);
    s.origin = is_a ? m0 : n0;
  };
  auto stream_open = [&](Stream& s, bool is_a, int g0) {  // position the cursor at global K-tile g0
    s.t = g0 / nk;
    s.kt = g0 - s.t * nk;
    s.par = g0 & 1;
    s.origin = 0;
    if (s.t < my_tiles) stream_origin(s, is_a);
  };
  auto stream_issue = [&](Stream& s, bool is_a, int half) {
    if (s.t >= my_tiles) {  // (uniform) past the end: nothing to fetch
      tail = true;
      return;
    }
    if (p.a_lo == 77) return;  // (diagnosis, PANDORA_GEMM256_NODMA=1: the same phases without their DMAs; results are garbage)
    const char* base = (is_a ? Ab : Wb) + (int64_t)s.kt * (BK * 2);  // (uniform: SGPRs)
    const uint32_t dst = lds_base + s.par * DBUF_BYTES + ((is_a ? 0 : 2) + half) * HALF_BYTES + (wave * 8) * 128;
    int r0, r1;
    if (is_a) {
      r0 = s.origin + half * 64 + srow;
      r1 = r0 + 128;
      const int last = p.M - 1;
      r0 = r0 < last ? r0 : last;
      r1 = r1 < last ? r1 : last;
    } else {
      r0 = s.origin + half * 32 + cb0;
      r1 = s.origin + half * 32 + cb1;
      const int last = p.N - 1;
      r0 = r0 < last ? r0 : last;
      r1 = r1 < last ? r1 : last;
    }
    const uint32_t ld2 = (uint32_t)(is_a ? lda2 : ldw2);  // (the operand spans < 4 GiB: pm_gemm's fits_u32 check)
    glds16_asm(base, (uint32_t)r0 * ld2 + lc16, dst);
    glds16_asm(base, (uint32_t)r1 * ld2 + lc16, dst + 64 * 128);
    s.par ^= 1;
    if (++s.kt == nk) {
      s.kt = 0;
      if (++s.t < my_tiles) stream_origin(s, is_a);
    }
  };

  if (total_g == 0) return;
  // ---- prologue: what the steady-state schedule would have requested before phase 0 of K-tile 0 ----
  Stream sA0, sA1, sB0, sB1;
  stream_open(sA0, true, 0);
  stream_open(sB0, false, 0);
  stream_open(sB1, false, 0);
  stream_open(sA1, true, 0);
  stream_issue(sA0, true, 0);   // K-tile 0, all four half-tiles
  stream_issue(sB0, false, 0);
  stream_issue(sB1, false, 1);
  stream_issue(sA1, true, 1);
  stream_issue(sA0, true, 0);   // K-tile 1: the two the schedule would have fetched in phases 2 and 3 of "K-tile -1"
  stream_issue(sB0, false, 0);
  tail = false;  // (a one-K-tile workgroup sets it above; the waits below then only over-wait)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if (wr == 1) __builtin_amdgcn_s_barrier();  // ping-pong: waves 4-7 run one barrier interval behind

  // fragment addressing (bytes inside a half-tile)
  const int slot0 = ((fq) ^ (fr & 7)) << 4, slot1 = ((4 + fq) ^ (fr & 7)) << 4;
  const int a_row = (wr * 64 + fr) * 128, b_row = (wc * 32 + fr) * 128;

  // (two NAMED accumulator halves: an array indexed by the row half went to scratch memory - every phase reloaded it)
  f32x4 acc0[4][4], acc1[4][4];
  Pack8<T> af[4][2], bf0[2][2], bf1[2][2];
  int t_cur = 0, kt_cur = 0, m0 = 0, n0 = 0;
  auto zero = [](f32x4 (&a)[4][4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) a[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  };
  // (accumulators start at zero and the residual is added in the epilogue: started from residual LOADS - the
  // 128x128 kernels' residual_into_acc - hipcc put an s_waitcnt vmcnt(0) in front of the first MFMA of EVERY K-tile,
  // because the loads may be pending on the loop's back edge; that wait drained the whole DMA pipeline each time)
  auto begin_tile = [&]() {
    decode(t_cur, m0, n0);
    zero(acc0);
    zero(acc1);
  };
  begin_tile();

  for (int g = 0; g < total_g; ++g) {
    const char* const buf = smem + (g & 1) * DBUF_BYTES;
    // one phase: RH / CH = the quadrant; LOAD_A / LOAD_B = which fragments this phase (re)reads
#define PM_PHASE(RH, CH, LOAD_A, LOAD_B, STREAM, STAGE_IS_A, STAGE_HALF)                                            \
  {                                                                                                                 \
    if (tail)                                                                                                       \
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                              \
    else /* this wave's DMAs of phases <= p-4 have landed; the three youngest phases' stay in flight */             \
      asm volatile("s_waitcnt vmcnt(6)" ::: "memory");                                                              \
    if (LOAD_B) {                                                                                                   \
      const char* bs = buf + (2 + CH) * HALF_BYTES + b_row;                                                         \
      _Pragma("unroll") for (int jj = 0; jj < 2; ++jj) {                                                            \
        bf##CH[jj][0].u = *reinterpret_cast<const u32x4*>(bs + jj * 2048 + slot0);                                  \
        bf##CH[jj][1].u = *reinterpret_cast<const u32x4*>(bs + jj * 2048 + slot1);                                  \
      }                                                                                                             \
    }                                                                                                               \
    if (LOAD_A) {                                                                                                   \
      __builtin_amdgcn_sched_barrier(0);                                                                            \
      const char* as = buf + (RH)*HALF_BYTES + a_row;                                                               \
      _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                               \
        af[i][0].u = *reinterpret_cast<const u32x4*>(as + i * 2048 + slot0);                                        \
        af[i][1].u = *reinterpret_cast<const u32x4*>(as + i * 2048 + slot1);                                        \
      }                                                                                                             \
    }                                                                                                               \
    stream_issue(STREAM, STAGE_IS_A, STAGE_HALF);                                                                   \
    __builtin_amdgcn_s_barrier();                                                                                   \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                              \
    __builtin_amdgcn_sched_barrier(0);                                                                              \
    __builtin_amdgcn_s_setprio(1);                                                                                  \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                                \
    _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                                   \
    _Pragma("unroll") for (int jj = 0; jj < 2; ++jj)                                                                \
        acc##RH[i][2 * (CH) + jj] = mfma16(bf##CH[jj][ks].v, af[i][ks].v, acc##RH[i][2 * (CH) + jj]);               \
    __builtin_amdgcn_s_setprio(0);                                                                                  \
    __builtin_amdgcn_sched_barrier(0);                                                                              \
    __builtin_amdgcn_s_barrier();                                                                                   \
  }
    PM_PHASE(0, 0, true, true, sB1, false, 1)    // reads A0, B0; stages B1 of K-tile g+1
    PM_PHASE(0, 1, false, true, sA1, true, 1)    // reads B1;     stages A1 of g+1
    PM_PHASE(1, 1, true, false, sA0, true, 0)    // reads A1;     stages A0 of g+2
    PM_PHASE(1, 0, false, false, sB0, false, 0) //               stages B0 of g+2
#undef PM_PHASE
    if (++kt_cur == nk) {
      // ---------------- epilogue of this tile (registers only: the DMA pipeline keeps streaming the next tile) ----
      // Both wave groups run it AT THE SAME TIME: waves 0-3 first wait one barrier for waves 4-7 to finish their last
      // MFMA cluster, and waves 4-7 re-open the half-phase gap with one barrier behind the epilogue.  (Left staggered,
      // each group would sit at a barrier through the other group's epilogue: two epilogues of idle matrix pipes per tile.)
      if (wr == 0) __builtin_amdgcn_s_barrier();
      // (laundered lane coordinates: hipcc otherwise hoists the lane-invariant part of the epilogue's address arithmetic
      // out of the tile loop - dozens of registers that then live through the K loop, at the 256-register cap: spills, whose
      // reloads are loads the compiler waits for with vmcnt(0) inside the phases)
      int fr_l = fr, fq_l = fq;
      asm volatile("" : "+v"(fr_l), "+v"(fq_l));
      float bv[4][4];
      load_bias_regs(p, bv, n0, wc, fq_l);
      {
        const int m_eff = m0 + wr * 128;
        epilogue_regs<T>(p, acc0, bv, m_eff, n0, 0, wc, fr_l, fq_l, m_eff >> 6, 0, false);
        epilogue_regs<T>(p, acc1, bv, m_eff + 64, n0, 0, wc, fr_l, fq_l, (m_eff + 64) >> 6, 0, false);
      }
      // The epilogue's loads / stores are the only memory operations of this kernel that hipcc counts.  Left pending on
      // the loop's back edge they made it put an s_waitcnt vmcnt(0) into the load segments of phases 1 and 2 of EVERY
      // K-tile (a register it redefines there may still be the target of one) - draining the DMA pipeline each time.  One
      // wait the compiler can see, here, once per tile: its bookkeeping is empty on the back edge.  (vmcnt 0; exp / lgkm untouched)
      __builtin_amdgcn_s_waitcnt(0x0F70);
      kt_cur = 0;
      ++t_cur;
      if (t_cur < my_tiles) begin_tile();
      if (wr == 1) __builtin_amdgcn_s_barrier();
    }
  }
}

static int g_gemm256 = 1;  // PANDORA_GEMM256: 0 = never, 1 = by gemm256_prefer(), 2 = wherever legal
static int g_gemm256_nodma = 0;  // PANDORA_GEMM256_NODMA=1: diagnosis build of the same phases without DMAs (garbage results)
static void init256() {
  static const bool once = [] {
    const char* e = diag_env("PANDORA_GEMM256");
    if (e) g_gemm256 = atoi(e);
    const char* nd = diag_env("PANDORA_GEMM256_NODMA");
    if (nd) g_gemm256_nodma = atoi(nd);
    return true;
  }();
  (void)once;
}

// Legal: dense 16-bit operands, whole K-tiles, no split-K, no W wrap, a fast-flavour epilogue.
// Preferred (PANDORA_GEMM256=1, measured in profiles/r03/gemm256_ab.txt): the launch is a whole number of rounds of
// one 256x256 tile per CU - the persistent grid has no second workgroup per CU to fill a ragged last round, and a tile
// boundary costs ~10 us (the in-order vmcnt makes the first phases of the next tile wait for the epilogue's stores),
// so the kernel only pays where
//   rounds efficiency  tiles / (ceil(tiles / CUs) * CUs)   x   tile utilisation  M*N / (padded M*N)   >= 0.89:
// the GEGLU projections of levels 1 / 2 at 576x1024 (-6 / -4 %), the K = 512 / N = 4096 and 1536 projections (-14 / -9 %),
// 36864x2560x2560 (-10 %); the q|k|v projections (efficiency 0.84 / 0.70: +12 / +31 %) and everything with M <= 2304
// stay on the 128x128 kernels.
bool gemm256_wanted(const GemmParams& p, int flags, int num_cus) {
  init256();
  if (g_gemm256 == 0) return false;
  if (flags & (PM_FLAG_A_F32 | PM_FLAG_W_WRAP | PM_FLAG_A_LO)) return false;
  if (p.splits > 1 || p.K < 4 * BK || (p.K % BK)) return false;
  {  // the fast epilogue flavours only: 8-column vectors (4 for GEGLU), aligned rows
    const bool geglu = p.act == PM_ACT_GEGLU;
    const int nout = geglu ? (p.N >> 1) : p.N;
    if (geglu ? ((nout & 3) || (p.ldc & 3)) : ((nout & 7) || (p.ldc & 7) || (p.R != nullptr && (p.ldr & 7)))) return false;
    if ((p.N & 3) != 0) return false;  // (load_bias_regs: 16-byte bias vectors)
    if (p.colstats != nullptr || (p.R != nullptr && p.res32)) return false;  // (statistics / f32 residual: the 128x128 kernels)
  }
  if (g_gemm256 == 2) return true;
  const int64_t mt = (p.M + BM2 - 1) / BM2, nt = (p.N + BN2 - 1) / BN2;
  const double util = (double)p.M * p.N / ((double)mt * BM2 * nt * BN2);
  const int64_t tiles = mt * nt, rounds = (tiles + num_cus - 1) / num_cus;
  return util * (double)tiles / (double)(rounds * num_cus) >= 0.89;
}

template <typename T> int launch_gemm256(const GemmParams& p, int num_cus, hipStream_t stream) {
  GemmParams q = p;
  if (g_gemm256_nodma) q.a_lo = 77;
  q.mtiles = (p.M + BM2 - 1) / BM2;
  q.ntiles = (p.N + BN2 - 1) / BN2;
  const int64_t nwork = (int64_t)q.mtiles * q.ntiles;
  const int grid = (int)(nwork < num_cus ? nwork : num_cus);
  static bool attr_set[64] = {false};
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev >= 0 && dev < 64 && !attr_set[dev]) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm256_kernel<T>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              LDS256);
    attr_set[dev] = true;
  }
  hipLaunchKernelGGL((gemm256_kernel<T>), dim3(grid), dim3(512), LDS256, stream, q);
  return check_launch();
}
template int launch_gemm256<f16>(const GemmParams&, int, hipStream_t);
template int launch_gemm256<bf16>(const GemmParams&, int, hipStream_t);

}  // namespace pm
#endif  // PM_DIAG
