// MFMA GEMM core for gfx950 with three A-operand loaders (dense rows, 3x3 im2col on channels-last
// frames, 3-tap temporal im2col) and a fused epilogue (bias, SiLU / GEGLU, residual).
//
//   C[M, N] = epi(A[M, K] . W[N, K]^T),  f32 accumulate on v_mfma_f32_16x16x32_{f16,bf16}.
//
// Tiling: 128x128 output tile per 256-thread workgroup (4 waves as 2x2, 64x64 per wave = 4x4 MFMA
// blocks), BK = 64, two LDS buffers, one barrier per K-tile.  16-bit A and W tiles go global -> LDS by
// DMA (global_load_lds_dwordx4: no VGPR round trip, no ds_write; the im2col gather and the padding
// are per-lane SOURCE addresses); an f32 A operand (residual stream) is staged through registers and
// rounded on the way.  LDS rows are 128 B with the 16-byte chunk index XOR-ed by (row & 7) - applied
// on the source side - so the ds_read_b128 fragment reads spread over the banks.  The epilogue works from
// registers (operands swapped in the MFMA, W rows interleaved by the loader: epilogue_regs), 16 bytes per lane.
#include "gemm_common.hpp"

namespace pm {

// 128x128 tile, 4 waves, 2 LDS stages, 2 workgroups per CU (any shape, split-K, f32 A).
template <typename T, int AMODE, bool A32>
__global__ __launch_bounds__(256, 2) void gemm_kernel(const GemmParams p) {
  typedef typename std::conditional<A32, float, T>::type TA;  // storage type of the A operand
  constexpr int ES = sizeof(TA);
  // Persistent walk (16-bit operands): a workgroup takes work items w, w + G, ...; the next item's first K-tile is
  // issued in the last K-step of this item and its barrier is deferred to BEHIND the epilogue (the epilogue is
  // LDS-free), so that DMA round trip and the next tile's residual / bias loads fly during the stores instead of
  // at the head of a fresh workgroup.  (The f32-operand instantiations stay one item per workgroup: the extra live
  // state pushed them past 256 VGPRs.)
  constexpr bool PERSIST = !A32;
  constexpr int NT = 256;                    // threads
  constexpr int WMW = 2;                     // waves along M (2 along N)
  constexpr int BMT = WMW * 64;              // tile rows
  constexpr int RSTEP = NT / 8;              // rows staged per loader pass
  constexpr int BP = BN / RSTEP;             // loader passes over the W tile (A tile: always 4)
  constexpr int A_BYTES = BMT * BK * 2, B_BYTES = BN * BK * 2;
  constexpr int STAGE_BYTES = A_BYTES + B_BYTES;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // stage s: A tile at s*STAGE_BYTES, W tile right behind it
  char* const As = smem;
  char* const Bs = smem + A_BYTES;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int fr = lane & 15, fq = lane >> 4;

  const int G = gridDim.x;
  const int nwork = p.mtiles * p.ntiles * p.splits;
  const int nk_all = (p.K + BK - 1) / BK;
  // work item -> (row tile, column tile, K slice).  Supertile order: workgroups that run together on one
  // XCD (a contiguous run of ids) cover GM row tiles x several column tiles, so both the A and the W
  // panels they stream stay in that XCD's 4 MiB L2 (row-major order re-fetched the whole W panel set
  // for every row tile: 9x the algorithmic bytes on the wide GEGLU GEMM, profiles/r01/pmc_traffic.md)
  auto decode = [&](int w, int& mt, int& nt, int& split) {
    constexpr int GM = 8;
    split = w % p.splits;
    const int wg = w / p.splits;
    const int grp = wg / (GM * p.ntiles);
    const int first_m = grp * GM;
    const int gm = (p.mtiles - first_m < GM) ? p.mtiles - first_m : GM;
    const int rin = wg - grp * GM * p.ntiles;
    nt = rin / gm;
    mt = first_m + (rin - nt * gm);
    if (AMODE == A_CONVT3) {
      // temporal conv: row tile (frame f, pixel block b) reads the same pixel block of frames f-1, f, f+1.
      // Enumerate the row tiles pixel-block-major (consecutive ids = consecutive frames of one block), so the
      // three readers of an input tile run together on one XCD instead of a whole frame of tiles apart
      const int bpf = p.P / BM;  // pixel blocks per frame (remap only when tiles do not straddle frames)
      if (bpf * BM == p.P && p.mtiles == bpf * p.F) mt = (mt % p.F) * bpf + (mt / p.F);
    }
  };

  const char* const Ab = reinterpret_cast<const char*>(p.A);
  const char* const Wb = reinterpret_cast<const char*>(p.Wt);
  const char* const zero = reinterpret_cast<const char*>(p.zero);

  // ---- loader state.  This thread stages rows lr + 32*j, filling the PHYSICAL 16-byte slot tid & 7
  // of each 128-byte LDS row.  The XOR swizzle is applied on the SOURCE side: the lane fetches logical
  // k-chunk lc = slot ^ (row & 7) (row & 7 == lr & 7 for all four rows), so that the LDS image is
  // lane-linear, which is what a direct global->LDS DMA (global_load_lds_dwordx4) writes.
  // Addresses are kept as (wave-uniform 64-bit base that walks K) + (constant 32-bit lane offset) so the
  // main loop spends SALU, not VALU, on them: the kernel is otherwise VALU-issue-bound next to the MFMAs.
  const int lr = tid >> 3;
  const int lc = (tid & 7) ^ (lr & 7);
  uint32_t b_off[BP], a_off[4];
  int a_y[4], a_x[4];
  uint32_t inv[4] = {0u, 0u, 0u, 0u};  // fast conv modes: per row, the taps that are zero padding (see load_tile)
  // (tap, channel) position of the K walk.  Fast modes: one tap per K-tile (Cin % 64 == 0), tracked as
  // wave-uniform scalars.  General conv (stem Cin = 8, nearest-x2 upsample): per-lane.
  int tap_s = 0, ch_s = 0;       // uniform: tap and first channel of the current K-tile
  int tap_l = 0, ch_l = lc * 8;  // per-lane (A_CONV3X3 only)
  // point the loader at work item w; returns its first K-tile
  auto loader_begin = [&](int w) -> int {
    int mt, nt, split;
    decode(w, mt, nt, split);
    const int m0 = mt * BMT, n0 = nt * BN;
    const int kt0 = split * p.ktps;
#pragma unroll
    for (int j = 0; j < BP; ++j) {
      int n = n0 + cperm(lr + RSTEP * j, p.act == PM_ACT_GEGLU || p.natural);  // LDS row -> output column (epilogue_regs)
      if (n > p.N - 1) n = p.N - 1;
      b_off[j] = (uint32_t)(((int64_t)n * p.ldw + lc * 8) * 2);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      int m = m0 + lr + RSTEP * j;
      if (m > p.M - 1) m = p.M - 1;
      if (AMODE == A_DENSE) {
        a_off[j] = (uint32_t)(((int64_t)m * p.lda + lc * 8) * ES);
        a_y[j] = a_x[j] = 0;
      } else if (AMODE == A_CONV3X3 || AMODE == A_CONV3X3_FAST) {
        const int hw = p.Ho * p.Wo;
        const int f = m / hw;
        const int rem = m - f * hw;
        const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
        a_y[j] = oy * p.stride + 1 - p.pad;  // tap (dy, dx) reads virtual pixel (a_y + dy - 1, a_x + dx - 1)
        a_x[j] = ox * p.stride + 1 - p.pad;
        if (AMODE == A_CONV3X3_FAST) {  // no upsample: offset of the centre tap, the tap shift is uniform
          a_off[j] = (uint32_t)((((int64_t)f * p.Hin + a_y[j]) * p.Win + a_x[j]) * p.lda * ES + lc * 8 * ES);
          uint32_t bad = 0;
#pragma unroll
          for (int t = 0; t < 9; ++t) {
            const int ty = t / 3, tx = t - ty * 3;
            const bool ok = (unsigned)(a_y[j] + ty - 1) < (unsigned)p.Hv && (unsigned)(a_x[j] + tx - 1) < (unsigned)p.Wv;
            bad |= ok ? 0u : (1u << t);
          }
          inv[j] = bad;
        } else
          a_off[j] = (uint32_t)((int64_t)f * p.Hin * p.Win * p.lda * ES);
      } else {
        const int f = m / p.P;
        const int pix = m - f * p.P;
        a_off[j] = (uint32_t)((((int64_t)f * p.P + pix) * p.lda + lc * 8) * ES);
        const int fc = f % p.Fc;  // frame inside its clip (Fc == F unless clips are batched along the rows)
        a_y[j] = fc;
        a_x[j] = (int)(uint32_t)(((int64_t)pix * p.lda + lc * 8) * ES);  // byte offset of this row inside a halo frame
        inv[j] = (fc == 0 ? 1u : 0u) | (fc == p.Fc - 1 ? 4u : 0u);  // taps reaching frame -1 / frame Fc of the clip
      }
    }
    if (AMODE == A_CONV3X3_FAST) {  // channel-chunk-major walk: K-tile s = (chunk s / 9, tap s % 9), see load_tile
      const int chunk = kt0 / 9;
      tap_s = kt0 - chunk * 9;
      ch_s = chunk * BK;
    } else if (AMODE != A_DENSE) {
      tap_s = (kt0 * BK) / p.Cin;
      ch_s = kt0 * BK - tap_s * p.Cin;
      const int k_l = kt0 * BK + lc * 8;
      tap_l = k_l / p.Cin;
      ch_l = k_l - tap_l * p.Cin;
    }
    return kt0;
  };

  u32x4 ra[4], ra_hi[4], rb[BP];  // register staging (A32 only); ra_hi: second half of an f32 chunk
  // issue the loads of K-tile kt (tiles are requested in increasing order) into LDS buffer `buf`
  auto load_tile = [&](int kt, int buf) {
    const int kb = kt * BK;
    const bool kin_l = kb + lc * 8 < p.K;  // only the general conv mode can have a K tail
    const char* wb = Wb + (int64_t)((AMODE == A_DENSE && p.kwrap && kb >= p.kwrap) ? kb - p.kwrap : kb) * 2;
    const char* ab = Ab;
    const char *hlo = nullptr, *hhi = nullptr;  // temporal mode: halo frames at this K-tile's channel (uniform)
    int dy = 0, dx = 0;
    if (AMODE == A_DENSE) {
      ab = Ab + (int64_t)kb * ES;
    } else if (AMODE == A_CONV3X3_FAST) {
      // The fast 3x3 mode walks K channel-chunk-major: all 9 taps of a 64-channel chunk back to back (the
      // packed weights stay tap-major, only the order of summation changes).  The 9 taps re-read the same
      // input rows; tap-major put a whole pass over all channels (more than an XCD's L2 holds, with 32-64
      // tiles in flight) between two reads of a line and the fabric saw 2-2.7x the algorithmic bytes.
      dy = tap_s / 3;
      dx = tap_s - dy * 3;
      wb = Wb + ((int64_t)tap_s * p.Cin + ch_s) * 2;
      ab = Ab + ((int64_t)((dy - 1) * p.Win + (dx - 1)) * p.lda + ch_s) * ES;
    } else if (AMODE == A_CONVT3) {
      ab = Ab + ((int64_t)(tap_s - 1) * p.P * p.lda + ch_s) * ES;
      hlo = p.halo_lo ? reinterpret_cast<const char*>(p.halo_lo) + (int64_t)ch_s * ES : nullptr;
      hhi = p.halo_hi ? reinterpret_cast<const char*>(p.halo_hi) + (int64_t)ch_s * ES : nullptr;
    } else {
      dy = tap_l / 3;
      dx = tap_l - dy * 3;
    }
    // Fast conv modes without halo frames, 16-bit operands: the A tile goes through a buffer descriptor whose
    // base carries the uniform tap shift; a padding lane gets offset ~0 = out of range, for which
    // `buffer_load ... lds` writes zeros (see gemm_ring_kernel): 2-3 VALU per DMA instead of ~8.
    const bool use_desc = !A32 && (AMODE == A_CONV3X3_FAST || AMODE == A_CONVT3) && p.halo_lo == nullptr && p.halo_hi == nullptr;
    if constexpr (!A32) {
      if (use_desc) {
        const uint32_t nrec = (uint32_t)((Ab + p.a_bytes) - ab);
        __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)ab, (short)0, (int)nrec, 0x00020000);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const uint32_t voff = a_off[j] | (uint32_t)(-(int)((inv[j] >> tap_s) & 1u));
          const int dst = buf * STAGE_BYTES + (RSTEP * j + 8 * wave) * 128;
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void*)(As + dst), 16, voff, 0, 0, 0);
        }
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (use_desc) break;
      const char* src;
      if (AMODE == A_DENSE) {
        src = ab + a_off[j];
      } else if (AMODE == A_CONV3X3_FAST) {
        const bool ok = (unsigned)(a_y[j] + dy - 1) < (unsigned)p.Hv && (unsigned)(a_x[j] + dx - 1) < (unsigned)p.Wv;
        src = ok ? ab + a_off[j] : zero;
      } else if (AMODE == A_CONVT3) {
        // (per-row halo offsets are precomputed: the 64-bit multiplies that stood here made the loader the
        // bottleneck of the temporal conv: 1 250-1 450 issue cycles per K-step against 550 for dense)
        const int sf = a_y[j] + tap_s - 1;
        src = ab + a_off[j];
        if (sf < 0) src = hlo ? hlo + (uint32_t)a_x[j] : zero;
        if (sf >= p.Fc) src = hhi ? hhi + (uint32_t)a_x[j] : zero;
      } else {
        int iy = a_y[j] + dy - 1, ix = a_x[j] + dx - 1;
        const bool ok = kin_l && (unsigned)iy < (unsigned)p.Hv && (unsigned)ix < (unsigned)p.Wv;
        if (p.ups) {
          iy >>= 1;
          ix >>= 1;
        }
        src = ok ? Ab + a_off[j] + (((int64_t)iy * p.Win + ix) * p.lda + ch_l) * ES : zero;
      }
      if constexpr (A32) {  // f32 stream operand: through registers (needs the f32 -> 16-bit rounding)
        ra[j] = ld_global16(src);
        ra_hi[j] = ld_global16(src + 16);
      } else {  // 16-bit operands: DMA straight into the tile, 1 KiB per wave-instruction
        const int dst = buf * STAGE_BYTES + (RSTEP * j + 8 * wave) * 128;
        __builtin_amdgcn_global_load_lds((glb_void*)src, (lds_void*)(As + dst), 16, 0, 0);
      }
    }
#pragma unroll
    for (int j = 0; j < BP; ++j) {
      const char* wsrc = (AMODE == A_CONV3X3 && !kin_l) ? zero : wb + b_off[j];
      if constexpr (A32) {
        rb[j] = ld_global16(wsrc);
      } else {
        const int dst = buf * STAGE_BYTES + (RSTEP * j + 8 * wave) * 128;
        __builtin_amdgcn_global_load_lds((glb_void*)wsrc, (lds_void*)(Bs + dst), 16, 0, 0);
      }
    }
    if (AMODE == A_CONV3X3_FAST) {  // next tap of this channel chunk, then the next chunk
      if (++tap_s == 9) {
        tap_s = 0;
        ch_s += BK;
      }
    } else if (AMODE != A_DENSE) {  // advance the (tap, channel) walk to the next K-tile
      ch_s += BK;
      if (ch_s >= p.Cin) {
        ch_s -= p.Cin;
        ++tap_s;
      }
      if (AMODE == A_CONV3X3) {
        ch_l += BK;
        while (ch_l >= p.Cin) {
          ch_l -= p.Cin;
          ++tap_l;
        }
      }
    }
  };
  auto store_tile = [&](int buf) {
    if constexpr (!A32) return;  // DMA path: the tile is already in flight to LDS
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int off = buf * STAGE_BYTES + (lr + RSTEP * j) * 128 + ((tid & 7) << 4);
      union { u32x4 u; float f[4]; } lo, hi;
      lo.u = ra[j];
      hi.u = ra_hi[j];
      Pack8<T> cv;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        cv.e[e] = from_f32<T>(lo.f[e]);
        cv.e[e + 4] = from_f32<T>(hi.f[e]);
      }
      if (p.a_lo) {  // (uniform) second pass of a split-operand call: what the first pass rounded away
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          cv.e[e] = from_f32<T>(lo.f[e] - to_f32(cv.e[e]));
          cv.e[e + 4] = from_f32<T>(hi.f[e] - to_f32(cv.e[e + 4]));
        }
      }
      *reinterpret_cast<u32x4*>(As + off) = cv.u;
      *reinterpret_cast<u32x4*>(Bs + off) = rb[j];
    }
  };

  // fragment read offsets: row & 7 == fr & 7 for every MFMA block of this wave, so the swizzled slot
  // depends on the k-step only; the block index becomes an immediate offset of the ds_read_b128
  const int a_frag = (wm * 64 + fr) * 128, b_frag = (wn * 64 + fr) * 128;
  const int slot0 = ((fq) ^ (fr & 7)) << 4, slot1 = ((4 + fq) ^ (fr & 7)) << 4;

  int w = xcd_remap(blockIdx.x, G);  // (G <= nwork: every workgroup has a first item)
  int buf = 0;
  load_tile(loader_begin(w), 0);
  // The residual (accumulator start values) and the bias of the FIRST tile are requested right behind its first
  // K-tile's DMA and before the barrier that waits for it: their round trips overlap.  They used to be issued after
  // that barrier and waited for on their own - two memory latencies in a row at the head of every workgroup, as
  // long as the whole K loop of a K = 320 GEMM.
  f32x4 acc[4][4];
  float bv[4][4];
  bool res_done;
  {
    int mt, nt, split;
    decode(w, mt, nt, split);
    res_done = residual_into_acc<T>(p, acc, mt * BMT, nt * BN, wm, wn, fr, fq);
    load_bias_regs(p, bv, nt * BN, wn, fq);
  }
  store_tile(0);
  __syncthreads();  // (hipcc drains vmcnt before the barrier, so the DMA'd tile is visible)

  for (bool first_item = true;; first_item = false) {
    int mt, nt, split;
    decode(w, mt, nt, split);
    const int m0 = mt * BMT, n0 = nt * BN;
    const int kt0 = split * p.ktps;
    const int kt1 = (kt0 + p.ktps < nk_all) ? kt0 + p.ktps : nk_all;
    const bool has_next = PERSIST && (w + G < nwork);

    if (!first_item) {
      res_done = residual_into_acc<T>(p, acc, m0, n0, wm, wn, fr, fq);
      load_bias_regs(p, bv, n0, wn, fq);  // requested now, needed in the epilogue
    }
    if (!res_done) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

    for (int kt = kt0; kt < kt1; ++kt) {
      const bool more = (kt + 1 < kt1) || has_next;
      if (kt + 1 < kt1)
        load_tile(kt + 1, buf ^ 1);
      else if (has_next)  // first K-tile of the next work item: in flight during this item's epilogue
        load_tile(loader_begin(w + G), buf ^ 1);
      const char* as = As + buf * STAGE_BYTES + a_frag;
      const char* bs = Bs + buf * STAGE_BYTES + b_frag;
      // all 16 fragment reads of the K-tile are issued first (64 VGPRs), so the MFMAs of k-step 0 start
      // as soon as ITS operands land and the reads of k-step 1 are hidden behind them; written per
      // k-step the compiler serialises read -> lgkmcnt(0) -> MFMA four times per tile
      Pack8<T> a[2][4], b[2][4];
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const int slot = ks ? slot1 : slot0;
#pragma unroll
        for (int i = 0; i < 4; ++i) a[ks][i].u = *reinterpret_cast<const u32x4*>(as + i * 2048 + slot);
#pragma unroll
        for (int j = 0; j < 4; ++j) b[ks][j].u = *reinterpret_cast<const u32x4*>(bs + j * 2048 + slot);
      }
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[i][j] = mfma16(b[ks][j].v, a[ks][i].v, acc[i][j]);  // C^T: see epilogue_regs
      }
      if (more) store_tile(buf ^ 1);
      if (kt + 1 < kt1 || !has_next) __syncthreads();  // (last step before a prefetched item: barrier after the epilogue)
      buf ^= 1;
    }
    // ---------------- epilogue: registers -> global (no LDS staging, no barrier) ----------------
    // split-K slices store raw f32 partials into their slab; bias/act/residual run in the reduce pass
    // (r06: the 16-bit projection flavours - q|k|v with its column scale, plain / bias - on the lean epilogue where the wave's
    // 64 x 64 piece lies inside the matrix: gemm_common.hpp epilogue_lean16)
    if (piece_dead(p, m0 + wm * 64, n0 + wn * 64)) {
      // (nothing of this wave's piece is inside the matrix)
    } else if (AMODE == A_DENSE && !res_done && epilogue_lean16_ok(p, m0 + wm * 64, n0 + wn * 64))
      epilogue_lean16<T>(p, acc, bv, m0 + wm * 64, n0 + wn * 64, fr, fq);
    else if (epilogue_lean32_ok(p, m0 + wm * 64, n0 + wn * 64, res_done))  // (the all-f32 flavours of every A mode)
      epilogue_lean32<T>(p, acc, bv, m0 + wm * 64, n0 + wn * 64, fr, fq, mt * WMW + wm, split);
    else
      epilogue_regs<T>(p, acc, bv, m0, n0, wm, wn, fr, fq, mt * WMW + wm, split, res_done);
    if (!has_next) break;
    w += G;
    __syncthreads();  // the prefetched first K-tile of item w has landed (vmcnt drained here, behind the stores)
  }
}

// ---------------------------------------------------------------------------------------------
// Ring-pipelined, wave-specialised persistent variant (16-bit operands, whole K-tiles: dense, fast 3x3
// and temporal modes).
//
// The 2-stage kernel above issues its DMAs from the waves that also issue the MFMAs: a wave's
// instructions issue in order, so whenever the CU's load path is backed up the MFMAs wait behind the
// stalled global_load_lds, and with one workgroup per CU (every grid of <= 256 tiles: the M = 2560 / 640
// levels) a K-step costs a full round trip (~1.3 us measured: 25 GB/s per CU against the ~68 GB/s a CU
// can ingest).  Here one 8-wave workgroup owns the CU:
//   * waves 4-7 only load: 4 LDS stages of 32 KiB form a ring with the DMA of three K-tiles (96 KiB) in
//     flight across the barrier (counted s_waitcnt vmcnt, never 0 in steady state);
//   * waves 0-3 only compute (ds_read_b128 + MFMA, 64x64 per wave) and run the epilogue;
//   * one raw s_barrier per K-step hands stage k to the consumers and stage k-1 back to the loaders;
//   * the workgroup is persistent: it walks its share of the (tile, split) list and the loader cursor
//     runs three K-steps ahead ACROSS tile boundaries, so the next tile streams in during the epilogue;
//   * the epilogue is wave-private and LDS-free (epilogue_regs): each consumer wave stores its own 64x64
//     quadrant straight from the accumulators while the loaders keep streaming.
// LDS: the 128 KiB ring only (the epilogue works from registers: epilogue_regs).
constexpr int RING_STAGES = 4;
constexpr int RING_LDS = RING_STAGES * 2 * TILE_BYTES;

template <typename T, int AMODE>
__global__ __launch_bounds__(512, 1) void gemm_ring_kernel(const GemmParams p) {
  static_assert(AMODE != A_CONV3X3, "K tails / nearest-x2 stay on gemm_kernel");
  constexpr int STAGE_BYTES = 2 * TILE_BYTES;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const As = smem;
  char* const Bs = smem + TILE_BYTES;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int G = gridDim.x;
  const int nwork = p.mtiles * p.ntiles * p.splits;
  // Work walk.  Workgroups are dealt round-robin over the 8 XCDs (blockIdx % 8), each with its own L2.  When
  // the grid is a whole number of workgroups per XCD, XCD x owns the CONTIGUOUS id range [x*per, (x+1)*per) and
  // its G/8 workgroups sweep it in rounds of G/8 consecutive ids: tiles that run together, and the tiles of
  // consecutive rounds, share A / W panels in that L2 (the round-major walk i*G + slot re-read the panels from
  // the fabric: FETCH_SIZE 2.2x the algorithmic bytes on the L0 conv, profiles/r01/pmc_traffic.md).
  const bool xcd_walk = (G & 7) == 0 && nwork > G;
  const int per_xcd = (nwork + 7) >> 3, gx = G >> 3;
  const int slot_id = xcd_remap(blockIdx.x, G);  // (fallback: round i -> work item i * G + slot_id)
  auto work_id = [&](int round) -> int {  // -1: this workgroup has no work in that round (nor later)
    if (!xcd_walk) {
      const int w = round * G + slot_id;
      return w < nwork ? w : -1;
    }
    const int xcd = blockIdx.x & 7, local = round * gx + (blockIdx.x >> 3);
    const int w = xcd * per_xcd + local;
    return (local < per_xcd && w < nwork) ? w : -1;
  };
  const int nk_all = p.K / BK;

  // work item -> (row tile, column tile, K slice), same supertile order as gemm_kernel
  auto decode = [&](int w, int& mt, int& nt, int& split) {
    constexpr int GM = 8;
    split = w % p.splits;
    const int wg = w / p.splits;
    const int grp = wg / (GM * p.ntiles);
    const int first_m = grp * GM;
    const int gm = (p.mtiles - first_m < GM) ? p.mtiles - first_m : GM;
    const int rin = wg - grp * GM * p.ntiles;
    nt = rin / gm;
    mt = first_m + (rin - nt * gm);
    if (AMODE == A_CONVT3) {
      // temporal conv: row tile (frame f, pixel block b) reads the same pixel block of frames f-1, f, f+1.
      // Enumerate the row tiles pixel-block-major (consecutive ids = consecutive frames of one block), so the
      // three readers of an input tile run together on one XCD instead of a whole frame of tiles apart
      const int bpf = p.P / BM;  // pixel blocks per frame (remap only when tiles do not straddle frames)
      if (bpf * BM == p.P && p.mtiles == bpf * p.F) mt = (mt % p.F) * bpf + (mt / p.F);
    }
  };

  if (wave >= 4) {
    // ================= loader waves (see gemm_kernel for the source-side swizzle / scalarised K walk) =====
    const int lw = __builtin_amdgcn_readfirstlane(wave - 4);  // (scalar: the DMA's LDS base goes to M0 by SALU)
    const int r8 = lane >> 3;             // row inside an 8-row DMA piece; tile row = 32*j + 8*lw + r8
    const int lc = (lane & 7) ^ r8;       // logical 16-byte k-chunk this lane fetches ((row & 7) == r8)
    const char* const Ab = reinterpret_cast<const char*>(p.A);
    const char* const Wb = reinterpret_cast<const char*>(p.Wt);
    const char* const zero = reinterpret_cast<const char*>(p.zero);
    uint32_t b_off[4], a_off[4];
    int a_y[4], a_x[4];
    uint32_t inv[4] = {0u, 0u, 0u, 0u};  // conv modes: per row, the taps that are zero padding
    int tap_s = 0, ch_s = 0;
    int64_t nxt_a = 0, nxt_w = 0;  // fast conv: table entries of the K-step about to be issued
    int l_round = 0, l_kt = 0, l_kt1 = 0;
    auto loader_begin = [&]() -> bool {
      const int w = work_id(l_round);
      if (w < 0) return false;
      int mt, nt, split;
      decode(w, mt, nt, split);
      const int m0 = mt * BM, n0 = nt * BN;
      l_kt = split * p.ktps;
      l_kt1 = (l_kt + p.ktps < nk_all) ? l_kt + p.ktps : nk_all;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int pr = 32 * j + 8 * lw + r8;  // LDS row of the W tile -> output column (see epilogue_regs)
        int n = n0 + cperm(pr, p.act == PM_ACT_GEGLU || p.natural);
        if (n > p.N - 1) n = p.N - 1;
        b_off[j] = (uint32_t)(((int64_t)n * p.ldw + lc * 8) * 2);
        int m = m0 + 32 * j + 8 * lw + r8;
        if (m > p.M - 1) m = p.M - 1;
        if (AMODE == A_DENSE) {
          a_off[j] = (uint32_t)(((int64_t)m * p.lda + lc * 8) * 2);
          a_y[j] = a_x[j] = 0;
        } else if (AMODE == A_CONV3X3_FAST) {
          const int hw = p.Ho * p.Wo;
          const int f = m / hw;
          const int rem = m - f * hw;
          const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
          a_y[j] = oy * p.stride + 1 - p.pad;
          a_x[j] = ox * p.stride + 1 - p.pad;
          a_off[j] = (uint32_t)((((int64_t)f * p.Hin + a_y[j]) * p.Win + a_x[j]) * p.lda * 2 + lc * 16);
          uint32_t bad = 0;  // bit t: tap t of this row is zero padding
#pragma unroll
          for (int t = 0; t < 9; ++t) {
            const int ty = t / 3, tx = t - ty * 3;
            const bool ok = (unsigned)(a_y[j] + ty - 1) < (unsigned)p.Hv && (unsigned)(a_x[j] + tx - 1) < (unsigned)p.Wv;
            bad |= ok ? 0u : (1u << t);
          }
          inv[j] = bad;
        } else {
          const int f = m / p.P;
          const int pix = m - f * p.P;
          a_off[j] = (uint32_t)((((int64_t)f * p.P + pix) * p.lda + lc * 8) * 2);
          const int fc = f % p.Fc;  // frame inside its clip (Fc == F unless clips are batched along the rows)
          a_y[j] = fc;
          a_x[j] = (int)(uint32_t)(((int64_t)pix * p.lda + lc * 8) * 2);  // byte offset inside a halo frame
          inv[j] = (fc == 0 ? 1u : 0u) | (fc == p.Fc - 1 ? 4u : 0u);  // taps reaching frame -1 / frame Fc of the clip
        }
      }
      if (AMODE == A_CONV3X3_FAST) {  // channel-chunk-major walk (see gemm_kernel's load_tile)
        const int chunk = l_kt / 9;
        tap_s = l_kt - chunk * 9;
        ch_s = chunk * BK;
        nxt_a = p.tap_a[tap_s];
        nxt_w = p.tap_w[tap_s];
      } else if (AMODE != A_DENSE) {  // temporal: chunk-major too (frames f-1, f, f+1 per 64-channel chunk), same table
        const int chunk = l_kt / 3;
        tap_s = l_kt - chunk * 3;
        ch_s = chunk * BK;
        nxt_a = p.tap_a[tap_s];
        nxt_w = p.tap_w[tap_s];
      }
      return true;
    };
    auto load_tile = [&](int kt, int buf) {
      const char* wb = Wb + (int64_t)((AMODE == A_DENSE && p.kwrap && kt * BK >= p.kwrap) ? kt * BK - p.kwrap : kt * BK) * 2;
      const char* ab = Ab;
      const char *hlo = nullptr, *hhi = nullptr;
      int dy = 0, dx = 0;
      if (AMODE == A_DENSE) {
        ab = Ab + (int64_t)kt * (BK * 2);
      } else if (AMODE == A_CONV3X3_FAST) {
        // per-tap A shift / W offset come from a host-filled table in the kernel arguments (scalar loads by a
        // uniform index, fetched one K-step ahead at the end of the previous call): the lone loader wave spent
        // ~150 cycles per K-step on the division by 3 and the 64-bit multiply chains that stood here
        wb = Wb + nxt_w + (int64_t)ch_s * 2;
        ab = Ab + nxt_a + (int64_t)ch_s * 2;
      } else {
        wb = Wb + nxt_w + (int64_t)ch_s * 2;
        ab = Ab + nxt_a + (int64_t)ch_s * 2;
        hlo = p.halo_lo ? reinterpret_cast<const char*>(p.halo_lo) + (int64_t)ch_s * 2 : nullptr;
        hhi = p.halo_hi ? reinterpret_cast<const char*>(p.halo_hi) + (int64_t)ch_s * 2 : nullptr;
      }
      // Conv modes without halo frames: the A tile goes through a buffer descriptor whose base carries the
      // (uniform) tap shift and channel offset; a lane's offset is its constant centre-tap offset, and a
      // padding lane gets offset ~0: out of range, for which `buffer_load ... lds` writes ZEROS into LDS
      // (tools/probes/buffer_lds_oob.hip).  2 VALU per DMA instead of the ~8 of a per-lane 64-bit pointer
      // select against a zero page (in-kernel stamps: loader issue 950 -> 560 cycles per K-step, which
      // had made the LOADER the bottleneck of the convs).
      const bool use_desc = (AMODE != A_DENSE) && p.halo_lo == nullptr && p.halo_hi == nullptr;
      if (use_desc) {
        const uint32_t nrec = (uint32_t)((Ab + p.a_bytes) - ab);
        __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)ab, (short)0, (int)nrec, 0x00020000);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const uint32_t voff = a_off[j] | (uint32_t)(-(int)((inv[j] >> tap_s) & 1u));
          const int dst = buf * STAGE_BYTES + (32 * j + 8 * lw) * 128;
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void*)(As + dst), 16, voff, 0, 0, 0);
          __builtin_amdgcn_global_load_lds((glb_void*)(wb + b_off[j]), (lds_void*)(Bs + dst), 16, 0, 0);
        }
      } else {
  #pragma unroll
        for (int j = 0; j < 4; ++j) {
          const char* src;
          if (AMODE == A_DENSE) {
            src = ab + a_off[j];
          } else if (AMODE == A_CONV3X3_FAST) {
            dy = tap_s / 3;
            dx = tap_s - dy * 3;
            const bool ok = (unsigned)(a_y[j] + dy - 1) < (unsigned)p.Hv && (unsigned)(a_x[j] + dx - 1) < (unsigned)p.Wv;
            src = ok ? ab + a_off[j] : zero;
          } else {
            // (per-row halo offsets are precomputed: the 64-bit multiplies that stood here made the loader the
            // bottleneck of the temporal conv: 1 250-1 450 issue cycles per K-step against 550 for dense)
            const int sf = a_y[j] + tap_s - 1;
            src = ab + a_off[j];
            if (sf < 0) src = hlo ? hlo + (uint32_t)a_x[j] : zero;
            if (sf >= p.Fc) src = hhi ? hhi + (uint32_t)a_x[j] : zero;
          }
          const int dst = buf * STAGE_BYTES + (32 * j + 8 * lw) * 128;
          __builtin_amdgcn_global_load_lds((glb_void*)src, (lds_void*)(As + dst), 16, 0, 0);
          __builtin_amdgcn_global_load_lds((glb_void*)(wb + b_off[j]), (lds_void*)(Bs + dst), 16, 0, 0);
        }
      }
      if (AMODE == A_CONV3X3_FAST) {
        if (++tap_s == 9) {
          tap_s = 0;
          ch_s += BK;
        }
        nxt_a = p.tap_a[tap_s];
        nxt_w = p.tap_w[tap_s];
      } else if (AMODE != A_DENSE) {
        if (++tap_s == 3) {
          tap_s = 0;
          ch_s += BK;
        }
        nxt_a = p.tap_a[tap_s];
        nxt_w = p.tap_w[tap_s];
      }
    };
    bool l_valid = loader_begin();
    int l_stage = 0, inflight = 0;
    auto issue = [&]() {
      if (!l_valid) return;
      load_tile(l_kt, l_stage);
      l_stage = (l_stage + 1) & (RING_STAGES - 1);
      ++inflight;
      if (++l_kt == l_kt1) {
        ++l_round;
        l_valid = loader_begin();
      }
    };
    issue();
    issue();
    issue();
#ifdef PM_RING_PROF
    long long t_wait = 0, t_bar = 0, t_issue = 0, t_n = 0;
#endif
    // one iteration per K-step q of the consumers.  Barrier q publishes stages q AND q+1 (the consumers
    // pre-read the first fragments of step q+1 during step q), so everything but the youngest K-tile (8
    // DMAs per loader wave) must have landed; it also proves the consumers are done with stage q-1, which
    // the issue right behind it overwrites with step q+3.
    while (inflight > 0) {
#ifdef PM_RING_PROF
      const long long c0 = clock64();
#endif
      if (inflight >= 3)
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef PM_RING_PROF
      const long long c1 = clock64();
#endif
      __builtin_amdgcn_s_barrier();
#ifdef PM_RING_PROF
      const long long c2 = clock64();
#endif
      issue();
      --inflight;
#ifdef PM_RING_PROF
      const long long c3 = clock64();
      t_wait += c1 - c0; t_bar += c2 - c1; t_issue += c3 - c2; ++t_n;
#endif
    }
#ifdef PM_RING_PROF
    if (p.prof && lane == 0) {
      long long* d = p.prof + ((long long)blockIdx.x * 8 + wave) * 4;
      d[0] = t_wait; d[1] = t_bar; d[2] = t_issue; d[3] = t_n;
    }
#endif
    return;
  }

  // ================= consumer waves =================
  const int wm = wave >> 1, wn = wave & 1;
  const int fr = lane & 15, fq = lane >> 4;
  const int a_frag = (wm * 64 + fr) * 128, b_frag = (wn * 64 + fr) * 128;
  const int slot0 = ((fq) ^ (fr & 7)) << 4, slot1 = ((4 + fq) ^ (fr & 7)) << 4;
  const bool partial = p.splits > 1;
  int c_stage = 0;
#ifdef PM_RING_PROF
  long long t_cbar = 0, t_comp = 0, t_epi = 0, t_cn = 0;
#endif

  // Fragment registers are double-buffered by k-substep: set 0 (k 0..31 of a stage) is read while the
  // MFMAs of the previous substep run - across the barrier for the first substep of the next stage - so
  // no MFMA ever waits for LDS latency (one consumer wave per SIMD: nobody else would hide it).
  Pack8<T> fa0[4], fb0[4], fa1[4], fb1[4];
  auto read_frags = [&](Pack8<T>* fa, Pack8<T>* fb, int stg, int slot) {
    const char* as = As + stg * STAGE_BYTES + a_frag + slot;
    const char* bs = Bs + stg * STAGE_BYTES + b_frag + slot;
#pragma unroll
    for (int i = 0; i < 4; ++i) fa[i].u = *reinterpret_cast<const u32x4*>(as + i * 2048);
#pragma unroll
    for (int j = 0; j < 4; ++j) fb[j].u = *reinterpret_cast<const u32x4*>(bs + j * 2048);
  };
  __builtin_amdgcn_s_barrier();  // barrier 0: stages 0 and 1 are in LDS
  read_frags(fa0, fb0, 0, slot0);

  for (int c_round = 0;; ++c_round) {
    const int w = work_id(c_round);
    if (w < 0) break;
    const bool last_tile = (work_id(c_round + 1) < 0);
    int mt, nt, split;
    decode(w, mt, nt, split);
    const int m0 = mt * BM, n0 = nt * BN;
    const int kt0 = split * p.ktps;
    const int kt1 = (kt0 + p.ktps < nk_all) ? kt0 + p.ktps : nk_all;
    float bv[4][4];  // requested now, needed in the epilogue
    load_bias_regs(p, bv, n0, wn, fq);

    f32x4 acc[4][4];
    const bool res_done = residual_into_acc<T>(p, acc, m0, n0, wm, wn, fr, fq);
    if (!res_done) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

    for (int kt = kt0; kt < kt1; ++kt) {
#ifdef PM_RING_PROF
      const long long c0 = clock64();
#endif
      const bool more = (kt + 1 < kt1) || !last_tile;  // does a next step exist (possibly the next tile's)?
      const int nstage = (c_stage + 1) & (RING_STAGES - 1);
      // Issue pattern pinned per substep: one ds_read_b128 behind each of the first 8 MFMAs (an MFMA holds
      // the SIMD's issue for 8 of its 16 cycles: the read rides in the gap), then 8 bare MFMAs that cover
      // the last read's latency.  Left alone, hipcc sinks each read group down to its first use (the MFMAs
      // then wait out the LDS latency); read groups issued en bloc between the MFMA groups cost ~100
      // issue cycles each (in-kernel stamps: 726 cycles per K-step against 512 of MFMA).
      __builtin_amdgcn_sched_barrier(0);
      read_frags(fa1, fb1, c_stage, slot1);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = mfma16(fb0[j].v, fa0[i].v, acc[i][j]);
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // 1 MFMA
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);  // 1 DS read
      }
      __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
      __builtin_amdgcn_sched_barrier(0);
      // next step's first fragments: that stage was published by the previous step's barrier.  Read
      // unconditionally (after the very last step the values are simply unused): a branch here would make
      // hipcc merge the LDS wait counts of both paths and stall the MFMAs on these reads
      read_frags(fa0, fb0, nstage, slot0);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = mfma16(fb1[j].v, fa1[i].v, acc[i][j]);
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
      __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
      __builtin_amdgcn_sched_barrier(0);
#ifdef PM_RING_PROF
      const long long c1 = clock64();
#endif
      // all reads of stage c_stage have returned (their MFMAs are issued): hand it back, get stage + 2
      if (more) __builtin_amdgcn_s_barrier();
      c_stage = nstage;
#ifdef PM_RING_PROF
      const long long c2 = clock64();
      t_comp += c1 - c0; t_cbar += c2 - c1; ++t_cn;
#endif
    }
#ifdef PM_RING_PROF
    const long long e0 = clock64();
#endif

    // ---------------- epilogue of this wave's 64x64 quadrant (the loaders keep streaming the next tile) ----
    if (piece_dead(p, m0 + wm * 64, n0 + wn * 64)) {
    } else if (AMODE == A_DENSE && !res_done && epilogue_lean16_ok(p, m0 + wm * 64, n0 + wn * 64))
      epilogue_lean16<T>(p, acc, bv, m0 + wm * 64, n0 + wn * 64, fr, fq);  // (r06: see gemm_kernel)
    else if (epilogue_lean32_ok(p, m0 + wm * 64, n0 + wn * 64, res_done))
      epilogue_lean32<T>(p, acc, bv, m0 + wm * 64, n0 + wn * 64, fr, fq, mt * 2 + wm, split);
    else
      epilogue_regs<T>(p, acc, bv, m0, n0, wm, wn, fr, fq, mt * 2 + wm, split, res_done);
#ifdef PM_RING_PROF
    t_epi += clock64() - e0;
#endif
  }
#ifdef PM_RING_PROF
  if (p.prof && lane == 0) {
    long long* d = p.prof + ((long long)blockIdx.x * 8 + wave) * 4;
    d[0] = t_cbar; d[1] = t_comp; d[2] = t_epi; d[3] = t_cn;
  }
#endif
}

// Entry `i` (wave-uniform, 0..8) of a per-tap table in the kernel arguments, by a select chain over constant offsets: indexed
// dynamically, hipcc demoted the WHOLE by-value GemmParams of gemm_ringw_kernel's 3x3 instantiation to scratch memory (every
// p.field became a scratch load - 1403 of them, the loader's DMA issue path included: 3-4x slower than the 128x128 ring).
__device__ __forceinline__ int64_t tap_entry(const int64_t (&t)[9], int i) {
  int64_t r = t[0];
#pragma unroll
  for (int k = 1; k < 9; ++k) r = (i == k) ? t[k] : r;
  return r;
}

// gemm_ringw: the ring kernel on a 256(M) x 128(N) tile (PANDORA_GEMM_RINGW).  Why: in-kernel stamps of the ring kernel on long
// K loops (tools/ring_prof.py, 8192^3): the four loader waves spend 645 clocks per K-step ISSUING its 32 DMAs and the consumers
// wait for them (563 clocks of compute, 207 at the barrier) - and eight loader waves change nothing
// (profiles/r03/negative_result_ring8_*): the CU's vector-memory path takes ~50 B/clk, a 128x128x64 K-step needs 64 B per
// MFMA clock.  A 256x128 tile needs 48: four consumer waves of 128x64 (two 64-row accumulator halves, 64 MFMAs and 24
// ds_read_b128 per K-step), the same four loaders (8 + 4 pieces each), three 48-KiB stages; barrier q publishes stage q.
template <typename T, int AMODE>
__global__ __launch_bounds__(512, 1) void gemm_ringw_kernel(const GemmParams p) {
  static_assert(AMODE != A_CONV3X3, "K tails / nearest-x2 stay on gemm_kernel");
  constexpr int BMW = 256;                        // tile rows
  constexpr int STAGE_BYTES = 3 * TILE_BYTES;     // A tile 256 x 64 (32 KiB) | W tile 128 x 64 (16 KiB)
  constexpr int NSTAGE = 3;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const As = smem;
  char* const Bs = smem + 2 * TILE_BYTES;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int G = gridDim.x;
  const int nwork = p.mtiles * p.ntiles * p.splits;
  // Work walk.  Workgroups are dealt round-robin over the 8 XCDs (blockIdx % 8), each with its own L2.  When
  // the grid is a whole number of workgroups per XCD, XCD x owns the CONTIGUOUS id range [x*per, (x+1)*per) and
  // its G/8 workgroups sweep it in rounds of G/8 consecutive ids: tiles that run together, and the tiles of
  // consecutive rounds, share A / W panels in that L2 (the round-major walk i*G + slot re-read the panels from
  // the fabric: FETCH_SIZE 2.2x the algorithmic bytes on the L0 conv, profiles/r01/pmc_traffic.md).
  const bool xcd_walk = (G & 7) == 0 && nwork > G;
  const int per_xcd = (nwork + 7) >> 3, gx = G >> 3;
  const int slot_id = xcd_remap(blockIdx.x, G);  // (fallback: round i -> work item i * G + slot_id)
  auto work_id = [&](int round) -> int {  // -1: this workgroup has no work in that round (nor later)
    if (!xcd_walk) {
      const int w = round * G + slot_id;
      return w < nwork ? w : -1;
    }
    const int xcd = blockIdx.x & 7, local = round * gx + (blockIdx.x >> 3);
    const int w = xcd * per_xcd + local;
    return (local < per_xcd && w < nwork) ? w : -1;
  };
  const int nk_all = p.K / BK;

  // work item -> (row tile, column tile, K slice), same supertile order as gemm_kernel
  auto decode = [&](int w, int& mt, int& nt, int& split) {
    constexpr int GM = 8;
    split = w % p.splits;
    const int wg = w / p.splits;
    const int grp = wg / (GM * p.ntiles);
    const int first_m = grp * GM;
    const int gm = (p.mtiles - first_m < GM) ? p.mtiles - first_m : GM;
    const int rin = wg - grp * GM * p.ntiles;
    nt = rin / gm;
    mt = first_m + (rin - nt * gm);
    if (AMODE == A_CONVT3) {
      // temporal conv: row tile (frame f, pixel block b) reads the same pixel block of frames f-1, f, f+1.
      // Enumerate the row tiles pixel-block-major (consecutive ids = consecutive frames of one block), so the
      // three readers of an input tile run together on one XCD instead of a whole frame of tiles apart
      const int bpf = p.P / BMW;  // pixel blocks per frame (remap only when tiles do not straddle frames)
      if (bpf * BMW == p.P && p.mtiles == bpf * p.F) mt = (mt % p.F) * bpf + (mt / p.F);
    }
  };

  if (wave >= 4) {
    // ================= loader waves (see gemm_kernel for the source-side swizzle / scalarised K walk) =====
    const int lw = __builtin_amdgcn_readfirstlane(wave - 4);  // (scalar: the DMA's LDS base goes to M0 by SALU)
    const int r8 = lane >> 3;             // row inside an 8-row DMA piece; tile row = 32*j + 8*lw + r8
    const int lc = (lane & 7) ^ r8;       // logical 16-byte k-chunk this lane fetches ((row & 7) == r8)
    const char* const Ab = reinterpret_cast<const char*>(p.A);
    const char* const Wb = reinterpret_cast<const char*>(p.Wt);
    const char* const zero = reinterpret_cast<const char*>(p.zero);
    uint32_t b_off[4], a_off[8];
    int a_y[8], a_x[8];
    uint32_t inv[8] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};  // conv modes: per row, the taps that are zero padding
    int tap_s = 0, ch_s = 0;
    int64_t nxt_a = 0, nxt_w = 0;  // fast conv: table entries of the K-step about to be issued
    int l_round = 0, l_kt = 0, l_kt1 = 0;
    // (always_inline: left to the inliner, the 3x3 instantiation kept loader_begin - 8 rows x 9 taps - as a real function:
    // its by-reference captures, the kernel's GemmParams among them, then live in scratch memory)
    auto loader_begin = [&]() __attribute__((always_inline)) -> bool {
      const int w = work_id(l_round);
      if (w < 0) return false;
      int mt, nt, split;
      decode(w, mt, nt, split);
      const int m0 = mt * BMW, n0 = nt * BN;
      l_kt = split * p.ktps;
      l_kt1 = (l_kt + p.ktps < nk_all) ? l_kt + p.ktps : nk_all;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int pr = 32 * j + 8 * lw + r8;  // LDS row of the W tile -> output column (see epilogue_regs)
        int n = n0 + cperm(pr, p.act == PM_ACT_GEGLU || p.natural);
        if (n > p.N - 1) n = p.N - 1;
        b_off[j] = (uint32_t)(((int64_t)n * p.ldw + lc * 8) * 2);
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        int m = m0 + 32 * j + 8 * lw + r8;
        if (m > p.M - 1) m = p.M - 1;
        if (AMODE == A_DENSE) {
          a_off[j] = (uint32_t)(((int64_t)m * p.lda + lc * 8) * 2);
          a_y[j] = a_x[j] = 0;
        } else if (AMODE == A_CONV3X3_FAST) {
          const int hw = p.Ho * p.Wo;
          const int f = m / hw;
          const int rem = m - f * hw;
          const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
          a_y[j] = oy * p.stride + 1 - p.pad;
          a_x[j] = ox * p.stride + 1 - p.pad;
          a_off[j] = (uint32_t)((((int64_t)f * p.Hin + a_y[j]) * p.Win + a_x[j]) * p.lda * 2 + lc * 16);
          uint32_t bad = 0;  // bit t: tap t of this row is zero padding
#pragma unroll
          for (int t = 0; t < 9; ++t) {
            const int ty = t / 3, tx = t - ty * 3;
            const bool ok = (unsigned)(a_y[j] + ty - 1) < (unsigned)p.Hv && (unsigned)(a_x[j] + tx - 1) < (unsigned)p.Wv;
            bad |= ok ? 0u : (1u << t);
          }
          inv[j] = bad;
        } else {
          const int f = m / p.P;
          const int pix = m - f * p.P;
          a_off[j] = (uint32_t)((((int64_t)f * p.P + pix) * p.lda + lc * 8) * 2);
          const int fc = f % p.Fc;  // frame inside its clip (Fc == F unless clips are batched along the rows)
          a_y[j] = fc;
          a_x[j] = (int)(uint32_t)(((int64_t)pix * p.lda + lc * 8) * 2);  // byte offset inside a halo frame
          inv[j] = (fc == 0 ? 1u : 0u) | (fc == p.Fc - 1 ? 4u : 0u);  // taps reaching frame -1 / frame Fc of the clip
        }
      }
      if (AMODE == A_CONV3X3_FAST) {  // channel-chunk-major walk (see gemm_kernel's load_tile)
        const int chunk = l_kt / 9;
        tap_s = l_kt - chunk * 9;
        ch_s = chunk * BK;
        nxt_a = tap_entry(p.tap_a, tap_s);
        nxt_w = tap_entry(p.tap_w, tap_s);
      } else if (AMODE != A_DENSE) {  // temporal: chunk-major too (frames f-1, f, f+1 per 64-channel chunk), same table
        const int chunk = l_kt / 3;
        tap_s = l_kt - chunk * 3;
        ch_s = chunk * BK;
        nxt_a = tap_entry(p.tap_a, tap_s);
        nxt_w = tap_entry(p.tap_w, tap_s);
      }
      return true;
    };
    auto load_tile = [&](int kt, int buf) __attribute__((always_inline)) {
      const char* wb = Wb + (int64_t)((AMODE == A_DENSE && p.kwrap && kt * BK >= p.kwrap) ? kt * BK - p.kwrap : kt * BK) * 2;
      const char* ab = Ab;
      const char *hlo = nullptr, *hhi = nullptr;
      int dy = 0, dx = 0;
      if (AMODE == A_DENSE) {
        ab = Ab + (int64_t)kt * (BK * 2);
      } else if (AMODE == A_CONV3X3_FAST) {
        // per-tap A shift / W offset come from a host-filled table in the kernel arguments (scalar loads by a
        // uniform index, fetched one K-step ahead at the end of the previous call): the lone loader wave spent
        // ~150 cycles per K-step on the division by 3 and the 64-bit multiply chains that stood here
        wb = Wb + nxt_w + (int64_t)ch_s * 2;
        ab = Ab + nxt_a + (int64_t)ch_s * 2;
      } else {
        wb = Wb + nxt_w + (int64_t)ch_s * 2;
        ab = Ab + nxt_a + (int64_t)ch_s * 2;
        hlo = p.halo_lo ? reinterpret_cast<const char*>(p.halo_lo) + (int64_t)ch_s * 2 : nullptr;
        hhi = p.halo_hi ? reinterpret_cast<const char*>(p.halo_hi) + (int64_t)ch_s * 2 : nullptr;
      }
      // Conv modes without halo frames: the A tile goes through a buffer descriptor whose base carries the
      // (uniform) tap shift and channel offset; a lane's offset is its constant centre-tap offset, and a
      // padding lane gets offset ~0: out of range, for which `buffer_load ... lds` writes ZEROS into LDS
      // (tools/probes/buffer_lds_oob.hip).  2 VALU per DMA instead of the ~8 of a per-lane 64-bit pointer
      // select against a zero page (in-kernel stamps: loader issue 950 -> 560 cycles per K-step, which
      // had made the LOADER the bottleneck of the convs).
      const bool use_desc = (AMODE != A_DENSE) && p.halo_lo == nullptr && p.halo_hi == nullptr;
      if (use_desc) {
        const uint32_t nrec = (uint32_t)((Ab + p.a_bytes) - ab);
        __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)ab, (short)0, (int)nrec, 0x00020000);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const uint32_t voff = a_off[j] | (uint32_t)(-(int)((inv[j] >> tap_s) & 1u));
          const int dst = buf * STAGE_BYTES + (32 * j + 8 * lw) * 128;
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void*)(As + dst), 16, voff, 0, 0, 0);
          if (j < 4) __builtin_amdgcn_global_load_lds((glb_void*)(wb + b_off[j]), (lds_void*)(Bs + dst), 16, 0, 0);
        }
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const char* src;
          if (AMODE == A_DENSE) {
            src = ab + a_off[j];
          } else if (AMODE == A_CONV3X3_FAST) {
            dy = tap_s / 3;
            dx = tap_s - dy * 3;
            const bool ok = (unsigned)(a_y[j] + dy - 1) < (unsigned)p.Hv && (unsigned)(a_x[j] + dx - 1) < (unsigned)p.Wv;
            src = ok ? ab + a_off[j] : zero;
          } else {
            const int sf = a_y[j] + tap_s - 1;
            src = ab + a_off[j];
            if (sf < 0) src = hlo ? hlo + (uint32_t)a_x[j] : zero;
            if (sf >= p.Fc) src = hhi ? hhi + (uint32_t)a_x[j] : zero;
          }
          const int dst = buf * STAGE_BYTES + (32 * j + 8 * lw) * 128;
          __builtin_amdgcn_global_load_lds((glb_void*)src, (lds_void*)(As + dst), 16, 0, 0);
          if (j < 4) __builtin_amdgcn_global_load_lds((glb_void*)(wb + b_off[j]), (lds_void*)(Bs + dst), 16, 0, 0);
        }
      }
      if (AMODE == A_CONV3X3_FAST) {
        if (++tap_s == 9) {
          tap_s = 0;
          ch_s += BK;
        }
        nxt_a = tap_entry(p.tap_a, tap_s);
        nxt_w = tap_entry(p.tap_w, tap_s);
      } else if (AMODE != A_DENSE) {
        if (++tap_s == 3) {
          tap_s = 0;
          ch_s += BK;
        }
        nxt_a = tap_entry(p.tap_a, tap_s);
        nxt_w = tap_entry(p.tap_w, tap_s);
      }
    };
    bool l_valid = loader_begin();
    int l_stage = 0, inflight = 0;
    auto issue = [&]() __attribute__((always_inline)) {
      if (!l_valid) return;
      load_tile(l_kt, l_stage);
      l_stage = (l_stage + 1 == NSTAGE) ? 0 : l_stage + 1;
      ++inflight;
      if (++l_kt == l_kt1) {
        ++l_round;
        l_valid = loader_begin();
      }
    };
    issue();
    issue();
#ifdef PM_RING_PROF
    long long t_wait = 0, t_bar = 0, t_issue = 0, t_n = 0;
#endif
    // one iteration per K-step q of the consumers.  Barrier q publishes stage q (no pre-read across the barrier here: with
    // three 48-KiB stages the stage after next is the one the consumers have just left), so everything but the youngest
    // K-tile (12 DMAs per loader wave) must have landed; it also proves the consumers are done with stage q-1, which the
    // issue right behind it overwrites with step q+2.
    while (inflight > 0) {
#ifdef PM_RING_PROF
      const long long c0 = clock64();
#endif
      if (inflight >= 2)
        asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
      else
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef PM_RING_PROF
      const long long c1 = clock64();
#endif
      __builtin_amdgcn_s_barrier();
#ifdef PM_RING_PROF
      const long long c2 = clock64();
#endif
      issue();
      --inflight;
#ifdef PM_RING_PROF
      const long long c3 = clock64();
      t_wait += c1 - c0; t_bar += c2 - c1; t_issue += c3 - c2; ++t_n;
#endif
    }
#ifdef PM_RING_PROF
    if (p.prof && lane == 0) {
      long long* d = p.prof + ((long long)blockIdx.x * 8 + wave) * 4;
      d[0] = t_wait; d[1] = t_bar; d[2] = t_issue; d[3] = t_n;
    }
#endif
    return;
  }

  // ================= consumer waves: wave (wm, wn) owns rows wm*128 .. +128 (two 64-row halves), columns wn*64 .. +64 ==========
  const int wm = wave >> 1, wn = wave & 1;
  const int fr = lane & 15, fq = lane >> 4;
  const int a_frag = (wm * 128 + fr) * 128, b_frag = (wn * 64 + fr) * 128;
  const int slot0 = ((fq) ^ (fr & 7)) << 4, slot1 = ((4 + fq) ^ (fr & 7)) << 4;
  int c_stage = 0;
#ifdef PM_RING_PROF
  long long t_cbar = 0, t_comp = 0, t_epi = 0, t_cn = 0;
#endif

  // fragments of one k-substep: 8 A blocks (two row halves) + 4 W blocks; double-buffered by substep INSIDE a K-step
  Pack8<T> fa0[8], fb0[4], fa1[8], fb1[4];
  auto read_frags = [&](Pack8<T>* fa, Pack8<T>* fb, int stg, int slot) {
    const char* as = As + stg * STAGE_BYTES + a_frag + slot;
    const char* bs = Bs + stg * STAGE_BYTES + b_frag + slot;
    // in the order the MFMAs need them (row block i of both halves against all four column blocks, i = 0 first): the first
    // MFMA waits for three reads, not nine
    fb[0].u = *reinterpret_cast<const u32x4*>(bs);
    fa[0].u = *reinterpret_cast<const u32x4*>(as);
    fa[4].u = *reinterpret_cast<const u32x4*>(as + 4 * 2048);
#pragma unroll
    for (int j = 1; j < 4; ++j) fb[j].u = *reinterpret_cast<const u32x4*>(bs + j * 2048);
#pragma unroll
    for (int i = 1; i < 4; ++i) {
      fa[i].u = *reinterpret_cast<const u32x4*>(as + i * 2048);
      fa[4 + i].u = *reinterpret_cast<const u32x4*>(as + (4 + i) * 2048);
    }
  };

  for (int c_round = 0;; ++c_round) {
    const int w = work_id(c_round);
    if (w < 0) break;
    int mt, nt, split;
    decode(w, mt, nt, split);
    const int m0 = mt * BMW, n0 = nt * BN;
    const int kt0 = split * p.ktps;
    const int kt1 = (kt0 + p.ktps < nk_all) ? kt0 + p.ktps : nk_all;
    float bv[4][4];
    // dense: requested now, needed in the epilogue.  Conv modes (long K loops, a loader with far more scalar state): requested
    // in front of the epilogue - 16 registers less through the K loop is what keeps their loaders' state out of scratch
    if constexpr (AMODE == A_DENSE) load_bias_regs(p, bv, n0, wn, fq);

    f32x4 acc0[4][4], acc1[4][4];  // rows m0 + wm*128 + {0, 64} ..
    // (dead pieces, gemm_common.hpp piece_dead: skipped in the conv modes only - with the two extra branches the DENSE instance of
    //  this kernel goes from 21 to 74 spilled registers under hipcc and loses 1 % of a step; the conv instances go from 17 to 0)
    constexpr bool SKIP_DEAD = AMODE != A_DENSE;
    const bool res_done = residual_into_acc<T, SKIP_DEAD>(p, acc0, m0 + wm * 128, n0, 0, wn, fr, fq);
    if (res_done) {
      (void)residual_into_acc<T, SKIP_DEAD>(p, acc1, m0 + wm * 128 + 64, n0, 0, wn, fr, fq);
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc0[i][j] = acc1[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

    for (int kt = kt0; kt < kt1; ++kt) {
#ifdef PM_RING_PROF
      const long long c0 = clock64();
#endif
      __builtin_amdgcn_s_barrier();  // stage c_stage is in LDS (and the loaders may overwrite the stage we left)
#ifdef PM_RING_PROF
      const long long c1 = clock64();
#endif
      __builtin_amdgcn_sched_barrier(0);
      read_frags(fa0, fb0, c_stage, slot0);  // (exposed: three stages leave no landed stage to pre-read across the barrier)
      __builtin_amdgcn_sched_barrier(0);
      // second k-substep's 12 reads ride behind the first 12 MFMAs of the first (an MFMA holds the SIMD's issue for 8 of
      // its 16 cycles: the read goes in the gap), the other 20 cover the last read's latency - the ring kernel's pattern
      read_frags(fa1, fb1, c_stage, slot1);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          acc0[i][j] = mfma16(fb0[j].v, fa0[i].v, acc0[i][j]);
          acc1[i][j] = mfma16(fb0[j].v, fa0[4 + i].v, acc1[i][j]);
        }
#pragma unroll
      for (int t = 0; t < 12; ++t) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // 1 MFMA
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);  // 1 DS read
      }
      __builtin_amdgcn_sched_group_barrier(0x008, 20, 0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          acc0[i][j] = mfma16(fb1[j].v, fa1[i].v, acc0[i][j]);
          acc1[i][j] = mfma16(fb1[j].v, fa1[4 + i].v, acc1[i][j]);
        }
      __builtin_amdgcn_sched_barrier(0);
      c_stage = (c_stage + 1 == NSTAGE) ? 0 : c_stage + 1;
#ifdef PM_RING_PROF
      const long long c2 = clock64();
      t_cbar += c1 - c0; t_comp += c2 - c1; ++t_cn;
#endif
    }
#ifdef PM_RING_PROF
    const long long e0 = clock64();
#endif

    // ---------------- epilogue of this wave's 128x64 block, one 64-row half at a time (the loaders keep streaming) ----
    {
      if constexpr (AMODE != A_DENSE) load_bias_regs(p, bv, n0, wn, fq);
      const int m_eff = m0 + wm * 128;
      if (SKIP_DEAD && piece_dead(p, m_eff, n0 + wn * 64)) {
      } else if (AMODE == A_DENSE && !res_done && epilogue_lean16_ok(p, m_eff, n0 + wn * 64) && m_eff + 128 <= p.M) {
        epilogue_lean16<T>(p, acc0, bv, m_eff, n0 + wn * 64, fr, fq);  // (r06: see gemm_kernel)
        epilogue_lean16<T>(p, acc1, bv, m_eff + 64, n0 + wn * 64, fr, fq);
      } else if (epilogue_lean32_ok(p, m_eff, n0 + wn * 64, res_done) && m_eff + 128 <= p.M) {
        epilogue_lean32<T>(p, acc0, bv, m_eff, n0 + wn * 64, fr, fq, m_eff >> 6, split);
        epilogue_lean32<T>(p, acc1, bv, m_eff + 64, n0 + wn * 64, fr, fq, (m_eff + 64) >> 6, split);
      } else {
        epilogue_regs<T>(p, acc0, bv, m_eff, n0, 0, wn, fr, fq, m_eff >> 6, split, res_done);
        epilogue_regs<T>(p, acc1, bv, m_eff + 64, n0, 0, wn, fr, fq, (m_eff + 64) >> 6, split, res_done);
      }
    }
#ifdef PM_RING_PROF
    t_epi += clock64() - e0;
#endif
  }
#ifdef PM_RING_PROF
  if (p.prof && lane == 0) {
    long long* d = p.prof + ((long long)blockIdx.x * 8 + wave) * 4;
    d[0] = t_cbar; d[1] = t_comp; d[2] = t_epi; d[3] = t_cn;
  }
#endif
}

// split-K second pass: out = epi(sum_s ws[s]) ; one thread per 4 consecutive columns
template <typename T>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const GemmParams p) {
  const int n4 = (p.N + 3) >> 2;
  const int64_t total = (int64_t)p.M * n4;
  const int64_t slab = (int64_t)p.M * p.N;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t m = i / n4;
    const int n = (int)(i - m * n4) * 4;
    const float* src = p.ws + m * p.N + n;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    const bool vec = (n + 4 <= p.N) && ((p.N & 3) == 0);
    for (int s = 0; s < p.splits; ++s) {
      if (vec) {
        const f32x4 t = *reinterpret_cast<const f32x4*>(src + s * slab);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] += t[e];
      } else {
        for (int e = 0; e < 4 && n + e < p.N; ++e) v[e] += src[s * slab + e];
      }
    }
    for (int e = 0; e < 4 && n + e < p.N; ++e) {
      float x = p.bias_mul ? v[e] * p.bias[n + e] : v[e] + (p.bias ? p.bias[n + e] : 0.f);
      if (p.act == PM_ACT_SILU) x = silu_f(x);
      if (p.act == PM_ACT_GELU) x = gelu_erf_f(x);
      if (p.R) x += p.res32 ? reinterpret_cast<const float*>(p.R)[m * p.ldr + n + e]
                            : to_f32(reinterpret_cast<const T*>(p.R)[m * p.ldr + n + e]);
      if (p.out32)
        reinterpret_cast<float*>(p.C)[m * p.ldc + n + e] = x;
      else
        reinterpret_cast<T*>(p.C)[m * p.ldc + n + e] = from_f32<T>(x);
    }
  }
}

// split-K second pass that also emits the fused GroupNorm column sums: grid (ceil(M/16), ceil(N/256)); the 256
// threads of a block are 64 column quads x 4 row lanes: a thread owns 4 consecutive columns of rows rl, rl+4,
// rl+8, rl+12 of a 16-row block (all its slab loads are independent and issued up front), sums the slabs in
// slab order, applies the epilogue, stores, and accumulates {sum, sum of squares} of exactly the values stored;
// the 4 row lanes are combined through LDS in lane order -> colstats [ceil(M/16)][N][2] (deterministic).
// (N % 4 == 0; the deep levels whose output has too few tiles to run unsplit are exactly the ones whose
// GroupNorms would otherwise need a separate statistics pass + finalize: 60 of the 166 per forward.)
template <typename T>
__global__ __launch_bounds__(256) void splitk_reduce_stats_kernel(const GemmParams p) {
  __shared__ float red[4][64][8];
  const int cq4 = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int n = (blockIdx.y * 64 + cq4) * 4;
  const int r0 = blockIdx.x * 16;
  const int64_t slab = (int64_t)p.M * p.N;
  const bool ncol = n < p.N;
  float cs[4] = {0.f, 0.f, 0.f, 0.f}, cq[4] = {0.f, 0.f, 0.f, 0.f};
  if (ncol) {
    f32x4 bias4 = f32x4{0.f, 0.f, 0.f, 0.f};
    if (p.bias) bias4 = *reinterpret_cast<const f32x4*>(p.bias + n);
    f32x4 v[4];
    bool live[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int m = r0 + rl + 4 * u;
      live[u] = m < p.M;
      v[u] = *reinterpret_cast<const f32x4*>(p.ws + (int64_t)(live[u] ? m : 0) * p.N + n);
    }
    for (int sidx = 1; sidx < p.splits; ++sidx) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int m = r0 + rl + 4 * u;
        const f32x4 t = *reinterpret_cast<const f32x4*>(p.ws + sidx * slab + (int64_t)(live[u] ? m : 0) * p.N + n);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[u][e] += t[e];
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (!live[u]) continue;
      const int m = r0 + rl + 4 * u;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float x = p.bias_mul ? v[u][e] * bias4[e] : v[u][e] + bias4[e];
        if (p.act == PM_ACT_SILU) x = silu_f(x);
        if (p.act == PM_ACT_GELU) x = gelu_erf_f(x);
        v[u][e] = x;
      }
      if (p.R) {
        if (p.res32) {
          const f32x4 rr = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(p.R) + (int64_t)m * p.ldr + n);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[u][e] += rr[e];
        } else {
          Pack4<T> rr;
          rr.u = *reinterpret_cast<const u32x2*>(reinterpret_cast<const T*>(p.R) + (int64_t)m * p.ldr + n);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[u][e] += to_f32(rr.e[e]);
        }
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        cs[e] += v[u][e];
        cq[e] = fmaf(v[u][e], v[u][e], cq[e]);
      }
      if (p.out32) {
        *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.C) + (int64_t)m * p.ldc + n) = v[u];
      } else {
        Pack4<T> ov;
#pragma unroll
        for (int e = 0; e < 4; ++e) ov.e[e] = from_f32<T>(v[u][e]);
        *reinterpret_cast<u32x2*>(reinterpret_cast<T*>(p.C) + (int64_t)m * p.ldc + n) = ov.u;
      }
    }
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    red[rl][cq4][2 * e] = cs[e];
    red[rl][cq4][2 * e + 1] = cq[e];
  }
  __syncthreads();
  if (rl == 0 && p.colstats != nullptr && p.gs_ni > 0) {
    // r06: GroupNorm totals by integer atomics (gemm_common.hpp).  This wave's 64 lanes hold 4 consecutive columns each, in
    // lane order; a 4-run can straddle a group boundary (10 or 30 channels per group): the part in the run's FIRST group goes
    // through the segmented wave sum, the rest (at most one lane per boundary) is added directly
    float o[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = ((red[0][cq4][e] + red[1][cq4][e]) + red[2][cq4][e]) + red[3][cq4][e];
    const int cpg = p.N >> 5;
    const int g0 = ncol ? n / cpg : -1;
    float s0 = 0.f, q0 = 0.f, s1 = 0.f, q1 = 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const bool first = (n + e) / cpg == g0;
      s0 += first ? o[2 * e] : 0.f;
      q0 += first ? o[2 * e + 1] : 0.f;
      s1 += first ? 0.f : o[2 * e];
      q1 += first ? 0.f : o[2 * e + 1];
    }
    long long* base = reinterpret_cast<long long*>(p.colstats) + (int64_t)(r0 / p.gs_rows) * 32 * 4 * GS_STRIDE;
    gs_wave_add(base, s0, q0, g0, cq4);
    if (ncol && (n + 3) / cpg != g0 && n + 3 < p.N) {
      gs_atomic_add(base + (int64_t)(g0 + 1) * 4 * GS_STRIDE, s1);
      gs_atomic_add(base + ((int64_t)(g0 + 1) * 4 + 2) * GS_STRIDE, q1);
    }
  } else if (rl == 0 && ncol && p.colstats != nullptr) {
    float o[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = ((red[0][cq4][e] + red[1][cq4][e]) + red[2][cq4][e]) + red[3][cq4][e];
    float* dst = p.colstats + ((int64_t)blockIdx.x * p.N + n) * 2;
    *reinterpret_cast<f32x4*>(dst) = f32x4{o[0], o[1], o[2], o[3]};
    *reinterpret_cast<f32x4*>(dst + 4) = f32x4{o[4], o[5], o[6], o[7]};
  }
}

// Process-level state of this file, all of it write-once: three environment overrides read by init_once() before
// their first use (from EVERY entry point that depends on them, so a sizing query and the launch it sizes agree),
// and per-DEVICE caches (CU count, "dynamic LDS limit raised" per kernel instantiation - the attribute is per
// device, a process may drive several GPUs).
constexpr int MAX_DEVICES = 64;
static int g_ring = 1;  // PANDORA_GEMM_RING: 0 = never, 1 = by prefer_ring(), 2 = always
static int g_ringw = 1;  // PANDORA_GEMM_RINGW: 0 = never, 1 = by prefer_ringw(), 2 (with PANDORA_GEMM_RING=2) = every unsplit 16-bit call
static int g_ring_max_work = 0;  // PANDORA_GEMM_RING_MAX_WORK > 0: never use the ring kernel above that many work items
static int g_num_cus[MAX_DEVICES] = {0};
static int g_natural = 1;          // PANDORA_GEMM_NATURAL (diagnostics build): 0 = interleaved W rows for the f32 flavours too (A/B)
static int g_persist_per_cu = 0;    // PANDORA_GEMM_PERSIST: persistent 2-stage workgroups per CU (0 = one work item per workgroup; measured: 2/CU = no gain, 1 or 3/CU 5 % slower)
static int g_split_min_nk = 24;     // PANDORA_SPLITK_MIN_NK: shortest K loop (in 64-wide tiles) that is split
static int g_split_max_tiles = 384; // PANDORA_SPLITK_MAX_TILES: grids of at least this many tiles never split
static int g_split_model = 1;       // PANDORA_SPLITK_MODEL: 0 = the round-1 rule (aim at 512 work items), for A/B runs
static int g_split_force = 0;       // PANDORA_SPLITK_FORCE = s > 0: every splittable call takes (up to) s slices (sweeps that fit the model)

static int current_device() {
  int dev = 0;
  (void)hipGetDevice(&dev);
  return (dev >= 0 && dev < MAX_DEVICES) ? dev : 0;
}
#ifdef PM_RING_PROF
static long long* g_ring_prof = nullptr;
#endif

static void init_once() {
  static const bool init = [] {
    const char* m = diag_env("PANDORA_SPLITK_MIN_NK");
    if (m) g_split_min_nk = atoi(m);
    const char* mt = diag_env("PANDORA_SPLITK_MAX_TILES");
    if (mt) g_split_max_tiles = atoi(mt);
    const char* rwe = diag_env("PANDORA_GEMM_RINGW");
    if (rwe) g_ringw = atoi(rwe);
    const char* r = diag_env("PANDORA_GEMM_RING");
    if (r) g_ring = atoi(r);
    const char* rw = diag_env("PANDORA_GEMM_RING_MAX_WORK");
    if (rw) g_ring_max_work = atoi(rw);
    const char* sm = diag_env("PANDORA_SPLITK_MODEL");
    if (sm) g_split_model = atoi(sm);
    const char* sf = diag_env("PANDORA_SPLITK_FORCE");
    if (sf) g_split_force = atoi(sf);
    const char* ps = diag_env("PANDORA_GEMM_PERSIST");
    if (ps) g_persist_per_cu = atoi(ps);
    const char* nl = diag_env("PANDORA_GEMM_NATURAL");
    if (nl) g_natural = atoi(nl);
    return true;
  }();
  (void)init;
}

// Split-K plan of a call shape.  Few-tile grids (the deep U-Net levels) can cut their K loop over several
// workgroups, but the slabs cost a second pass: s partial [M, N] f32 slabs written by the main kernel, read by
// the reduce launch.  The old rule (aim at 512 work items) split 200-tile grids three ways - 600 items run in
// three rounds of one workgroup per CU, no shorter than the unsplit round, plus the reduce: the temporal convs at
// M = 2560 ran 64 us against 42 unsplit - and cut 50-tile grids 11 ways where 5 fill the chip once.  Now: the s
// that minimises  rounds(s) x (K-steps per slice x t_step + t_fixed) + reduce(s)  with the ring kernel's measured
// step (0.55 us per 64-wide K-step per workgroup, ~2 us of prologue + epilogue per round) and the slabs' HBM time.
static int choose_splits(int64_t M, int64_t N, int64_t K, int act, int* ktps, bool two_stage = false) {
  init_once();
  const int64_t tiles = ((M + BM - 1) / BM) * ((N + BN - 1) / BN);
  const int nk = (int)((K + BK - 1) / BK);
  *ktps = nk;
  if (act == PM_ACT_GEGLU || tiles >= g_split_max_tiles || nk < g_split_min_nk) return 1;
  if (g_split_force > 0) {
    int s = g_split_force;
    if (s > nk / 4) s = nk / 4;
    if (s < 2) return 1;
    *ktps = (nk + s - 1) / s;
    return (nk + *ktps - 1) / *ktps;
  }
  if (g_split_model == 0) {  // the round-1 rule, for A/B runs
    int64_t s = (512 + tiles - 1) / tiles;
    if (s > nk / 4) s = nk / 4;
    if (s > 32) s = 32;
    if (s < 2) return 1;
    *ktps = (int)((nk + s - 1) / s);
    return (nk + *ktps - 1) / *ktps;
  }
  // Work items run in rounds of one workgroup per CU on the ring kernel (0.55 us per K-step), of two per CU on the
  // 2-stage kernel, the only one the general 3x3 mode (nearest-x2 upsample) has (1.1 us per K-step per workgroup).
  // A slab is written and read once, mostly out of the 256 MiB Infinity Cache: ~9 TB/s.  Constants and the two
  // exceptions checked against a sweep of forced slice counts (PANDORA_SPLITK_FORCE = 1..10, both resolutions:
  // profiles/r02/splitk_forced_sweep.txt): the model's pick is within 3 % of the best forced count on every shape.
  const double slots = two_stage ? 512.0 : 256.0, t_step = two_stage ? 1.1 : 0.55, t_fixed = two_stage ? 3.0 : 2.0;
  const double slab_us = (double)M * (double)N * 8.0 / 9.0e6;
  int best_s = 1, best_ktps = nk;
  double best = 0.0;
  for (int s = 1; s <= 32 && s <= nk / 4; ++s) {
    const int kt = (nk + s - 1) / s, splits = (nk + kt - 1) / kt;
    if (s > 1 && splits != s) continue;  // (the same plan as a smaller s)
    const double rounds = (double)((tiles * splits + (int64_t)slots - 1) / (int64_t)slots);
    double cost = rounds * (kt * t_step + t_fixed) + (splits > 1 ? 2.5 + splits * slab_us : 0.0);
    if (splits == 1 && nk >= 256) cost *= 1.1;  // (very long unsplit loops measured ~10 % over the model: L2 window)
    if (s == 1 || cost < best * 0.97) {
      best = cost;
      best_s = splits;
      best_ktps = kt;
    }
  }
  *ktps = best_ktps;
  return best_s;
}

static void plan_split(GemmParams& p, void* workspace, size_t workspace_bytes, bool two_stage = false) {
  init_once();
  int ktps;
  // (a call with fused statistics keeps the plan pm_gemm_colstats_rows answers for: its colstats are sized by it)
  int s = choose_splits(p.M, p.N, p.K, p.act, &ktps, two_stage && p.colstats == nullptr);
  // fused statistics: from the main kernel's epilogue when the call runs unsplit (64-row blocks), from the
  // reduce pass when it splits (16-row blocks, pm_gemm_colstats_rows); the reduce variant needs 4-column vectors
  if (p.colstats != nullptr && s > 1 && ((p.N & 3) || p.act == PM_ACT_GEGLU)) {  // (pm_gemm_colstats_rows says 64 too)
    s = 1;
    ktps = (p.K + BK - 1) / BK;
  }
  if (s > 1 && (workspace == nullptr || workspace_bytes < (size_t)s * p.M * p.N * sizeof(float))) {
    s = 1;  // no (or too small a) workspace: run unsplit
    ktps = (p.K + BK - 1) / BK;
  }
  p.splits = s;
  p.ktps = ktps;
  p.ws = reinterpret_cast<float*>(workspace);
  // all-f32 epilogue (f32 output - its residual, if any, is f32 too - or split-K slabs) on the fast flavour's
  // preconditions: W rows staged in natural column order, 64 contiguous bytes per row and store / residual-load
  // instruction (gemm_common.hpp cperm)
  const int64_t ldc = s > 1 ? p.N : p.ldc;
  p.natural = (g_natural && p.act != PM_ACT_GEGLU && (p.out32 || s > 1) && (p.N & 31) == 0 && (ldc & 7) == 0 &&
               (p.R == nullptr || (p.ldr & 7) == 0)) ? 1 : 0;
}

template <typename T> static void launch_reduce(const GemmParams& p, hipStream_t stream) {
  // (the 16-row x 256-column block mapping is also the faster plain reduce whenever 4-column vectors apply)
  if (p.colstats != nullptr || ((p.N & 3) == 0 && (p.ldc & 3) == 0 && (p.R == nullptr || (p.ldr & 3) == 0))) {
    dim3 grid((unsigned)((p.M + 15) / 16), (unsigned)((p.N + 255) / 256));
    hipLaunchKernelGGL((splitk_reduce_stats_kernel<T>), grid, dim3(256), 0, stream, p);
    return;
  }
  const int64_t work = (int64_t)p.M * ((p.N + 3) / 4);
  int64_t nb = (work + 255) / 256;
  if (nb > 4096) nb = 4096;
  hipLaunchKernelGGL((splitk_reduce_kernel<T>), dim3((unsigned)nb), dim3(256), 0, stream, p);
}

static int num_cus() {
  const int dev = current_device();
  if (g_num_cus[dev] == 0) {
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 1) n = 256;
    g_num_cus[dev] = n;
  }
  return g_num_cus[dev];
}

template <typename T, int AMODE, bool A32> static int launch1(const GemmParams& p, hipStream_t stream) {
  constexpr int lds = 4 * TILE_BYTES;
  GemmParams q = p;
  q.mtiles = (p.M + BM - 1) / BM;
  const int64_t nwork = (int64_t)q.mtiles * p.ntiles * p.splits;
  // 16-bit operands: persistent walk, up to g_persist_wgs workgroups per CU's worth of slots; f32 A: one item each
  const int64_t cap = (int64_t)g_persist_per_cu * num_cus();
  const int grid = (int)((A32 || g_persist_per_cu <= 0 || nwork <= cap) ? nwork : cap);
  static bool attr_set[MAX_DEVICES] = {false};  // per device; idempotent (a benign race sets the same value twice)
  const int dev = current_device();
  if (!attr_set[dev]) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_kernel<T, AMODE, A32>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    attr_set[dev] = true;
  }
  hipLaunchKernelGGL((gemm_kernel<T, AMODE, A32>), dim3(grid), dim3(256), lds, stream, q);
  if (p.splits > 1) launch_reduce<T>(p, stream);
  return check_launch();
}

constexpr int RINGW_LDS = 3 * 3 * TILE_BYTES;  // three stages of (A 256 x 64 | W 128 x 64)
// 256-row tiles halve the tile count: only where that does not cost whole-round efficiency (measured per shape in the model:
// profiles/r03/ringw_ab.txt), unsplit, 16-bit operands, the dense / fast 3x3 / temporal loaders
static bool prefer_ringw(int amode, const GemmParams& p) {
  if (amode == A_CONV3X3) return false;  // (general 3x3 mode: K tails / nearest-x2 stay on gemm_kernel)
  if (p.splits != 1 || p.K < 8 * BK) return false;  // (K = 320 .. 448: the 256x256 kernel / the 128x128 kernels measure better)
  const int64_t cu = num_cus();
  const int64_t n256 = (int64_t)((p.M + 255) / 256) * p.ntiles, n128 = (int64_t)((p.M + BM - 1) / BM) * p.ntiles;
  const double e256 = (double)n256 / (double)(((n256 + cu - 1) / cu) * cu) * ((double)p.M / (double)(((p.M + 255) / 256) * 256));
  const double e128 = (double)n128 / (double)(((n128 + cu - 1) / cu) * cu) * ((double)p.M / (double)(((p.M + BM - 1) / BM) * BM));
  return n256 >= cu * 3 / 4 && e256 >= e128 - 0.06;
}
template <typename T, int AMODE> static int launch_ringw(const GemmParams& p, hipStream_t stream) {
  GemmParams q = p;
#ifdef PM_RING_PROF
  q.prof = g_ring_prof;
#endif
  q.mtiles = (p.M + 255) / 256;
  const int64_t nwork = (int64_t)q.mtiles * p.ntiles;
  const int grid = (int)(nwork < num_cus() ? nwork : num_cus());
  static bool attr_set[MAX_DEVICES] = {false};
  const int dev = current_device();
  if (!attr_set[dev]) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_ringw_kernel<T, AMODE>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, RINGW_LDS);
    attr_set[dev] = true;
  }
  hipLaunchKernelGGL((gemm_ringw_kernel<T, AMODE>), dim3(grid), dim3(512), RINGW_LDS, stream, q);
  return check_launch();
}

template <typename T, int AMODE> static int launch_ring(const GemmParams& p, hipStream_t stream) {
  GemmParams q = p;
#ifdef PM_RING_PROF
  q.prof = g_ring_prof;
#endif
  q.mtiles = (p.M + BM - 1) / BM;
  const int64_t nwork = (int64_t)q.mtiles * p.ntiles * p.splits;
  const int grid = (int)(nwork < num_cus() ? nwork : num_cus());
  static bool attr_set[MAX_DEVICES] = {false};
  const int dev = current_device();
  if (!attr_set[dev]) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_ring_kernel<T, AMODE>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, RING_LDS);
    attr_set[dev] = true;
  }
  hipLaunchKernelGGL((gemm_ring_kernel<T, AMODE>), dim3(grid), dim3(512), RING_LDS, stream, q);
  if (p.splits > 1) launch_reduce<T>(p, stream);
  return check_launch();
}

// Which kernel for a 16-bit-operand call (measured per shape in the model, tools/shape_profile.py with
// PANDORA_GEMM_RING=0 / 2).  The ring kernel has the faster steady state (~700 vs ~950 cycles per K-step per
// CU) and hides the load round trip, but owns the CU alone: it pays for every epilogue with idle MFMA time and
// quantises the grid in rounds of 1 workgroup per CU, where the 2-stage kernel runs rounds of 2.
static bool prefer_ring(int amode, const GemmParams& p) {
  if (p.act == PM_ACT_GEGLU) return false;  // (in the model the 2-stage kernel is 6-10% faster on the GEGLU GEMMs)
  const int64_t nwork = (int64_t)((p.M + BM - 1) / BM) * p.ntiles * p.splits;
  if (g_ring_max_work > 0 && nwork > g_ring_max_work) return false;
  const int64_t cu = num_cus();
  const double eff_ring = (double)nwork / (double)(((nwork + cu - 1) / cu) * cu);
  const double eff_two = (double)nwork / (double)(((nwork + 2 * cu - 1) / (2 * cu)) * 2 * cu);
  // (temporal convs: ties go to the ring kernel - 2-8 % faster at M = 10240 .. 147456, tools/shape_profile.py with
  // PANDORA_GEMM_RING=0/2, profiles/r02/kernel_choice_ab.txt)
  if (amode == A_CONVT3) return eff_ring + 0.05 >= eff_two || (nwork <= cu && nwork >= cu / 2);
  if (nwork <= cu) return true;  // one tile per CU: the 2-stage kernel would be round-trip-bound
  if (amode == A_CONV3X3_FAST) return eff_ring >= eff_two;  // long K loops: ties go to the faster steady state
  return (p.M <= 2560 && p.splits == 1) || eff_ring > eff_two + 0.05;
}

template <typename T, int AMODE> static int launch(const GemmParams& p, int flags, hipStream_t stream) {
  if (flags & PM_FLAG_A_F32) return launch1<T, AMODE, true>(p, stream);
  if constexpr (AMODE != A_CONV3X3) {
    if (p.splits == 1 && ((g_ringw == 2 && g_ring == 2) || (g_ringw == 1 && g_ring != 0 && prefer_ringw(AMODE, p))))
      return launch_ringw<T, AMODE>(p, stream);
    if (g_ring == 2 || (g_ring == 1 && prefer_ring(AMODE, p))) return launch_ring<T, AMODE>(p, stream);
  }
  return launch1<T, AMODE, false>(p, stream);
}

// the loaders address an operand as (64-bit uniform base) + (32-bit lane byte offset)
static bool fits_u32(int64_t elements, int flags) {
  return elements * ((flags & PM_FLAG_A_F32) ? 4 : 2) < (1ll << 32) - (1ll << 24);  // (margin: a conv's descriptor base sits up to a tap shift before A)
}

static int check_common(const void* A, const void* W, void* C, int64_t M, int64_t N, int64_t K,
                        int act) {
  if (!A || !W || !C) return PM_E_NULL;
  if (M < 1 || N < 1 || K < 8 || (K & 7)) return PM_E_SHAPE;
  if (M > (1ll << 30) || N > (1ll << 30) || K > (1ll << 30)) return PM_E_SHAPE;
  if (act == PM_ACT_GEGLU && (N % 32) != 0) return PM_E_SHAPE;
  if (act < PM_ACT_NONE || act > PM_ACT_GELU) return PM_E_SHAPE;
  return PM_OK;
}

}  // namespace pm

using namespace pm;

extern "C" int pm_gemm(const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias,
                       const void* residual, int64_t ldr, void* C, int64_t ldc, int64_t M,
                       int64_t N, int64_t K, int act, int flags, int dtype, void* workspace,
                       size_t workspace_bytes, float* colstats, void* stream) {
  int rc = check_common(A, W, C, M, N, K, act);
  if (rc) return rc;
  const int64_t Kw = (flags & PM_FLAG_W_WRAP) ? K / 2 : K;  // columns of W
  if ((flags & PM_FLAG_W_WRAP) && ((K % (2 * BK)) || (flags & PM_FLAG_A_F32))) return PM_E_SHAPE;
  if ((lda & ((flags & PM_FLAG_A_F32) ? 3 : 7)) || (ldw & 7) || lda < K || ldw < Kw) return PM_E_SHAPE;
  if (!fits_u32(M * lda, flags) || N * ldw * 2 >= (1ll << 32)) return PM_E_SHAPE;  // 32-bit lane offsets
  if ((flags & PM_FLAG_OUT_F32) && act == PM_ACT_GEGLU) return PM_E_SHAPE;
  if ((flags & PM_FLAG_BIAS_IS_SCALE) && (bias == nullptr || act == PM_ACT_GEGLU)) return PM_E_SHAPE;
  if ((flags & PM_FLAG_A_LO) && !(flags & PM_FLAG_A_F32)) return PM_E_SHAPE;
  GemmParams p{};
  p.A = A; p.lda = lda; p.Wt = W; p.ldw = ldw; p.bias = bias; p.R = residual; p.ldr = ldr;
  p.C = C; p.ldc = ldc; p.M = (int)M; p.N = (int)N; p.K = (int)K; p.act = act;
  p.out32 = (flags & PM_FLAG_OUT_F32) ? 1 : 0;
  p.res32 = (flags & (PM_FLAG_OUT_F32 | PM_FLAG_RES_F32)) ? 1 : 0;
  p.bias_mul = (flags & PM_FLAG_BIAS_IS_SCALE) ? 1 : 0;
  p.a_lo = (flags & PM_FLAG_A_LO) ? 1 : 0;
  p.kwrap = (flags & PM_FLAG_W_WRAP) ? (int)Kw : 0;
  p.colstats = colstats;
  if (colstats != nullptr && (flags & PM_FLAG_STATS_I64)) {  // r06: int64 group totals [NI][32][4] instead of column sums
    p.gs_ni = (flags >> 8) & 0xff;
    if (p.gs_ni < 1 || (p.M % p.gs_ni) || (p.N & 31)) return PM_E_SHAPE;
    p.gs_rows = p.M / p.gs_ni;
  }
  p.ntiles = (int)((N + BN - 1) / BN);
  p.zero = A;  // dense K tails never occur (K % 8 == 0 and whole chunks only); see kin below
  // A dense K tail (K % 64 != 0) reads chunk-wise: chunks with k >= K take `zero`; any 16 readable
  // bytes would poison the accumulator, so require a real zero source only when a tail exists.
  if (K % BK) return PM_E_SHAPE;  // all Linear layers on the path have K % 64 == 0
  plan_split(p, workspace, workspace_bytes);
  if (!GS_EPILOGUE && p.gs_ni > 0 && p.splits <= 1) return PM_E_SHAPE;  // (integer totals from an unsplit call: diagnostics build only)
  if (p.colstats != nullptr && p.splits > 1 && ((p.ldc & 3) || (p.R != nullptr && (p.ldr & 3)))) return PM_E_SHAPE;
  // the 16-bit projection flavours on whole 256 x 256 tiles: the assembly kernel with the K stream continuous across tiles
  if (gemm_wide_stream_wanted(p, flags, num_cus()))
    PM_DISPATCH_DTYPE(dtype, T, return (launch_gemm_wide_stream<T>(p, num_cus(), (hipStream_t)stream)));
  // wide / long-K shapes whose grid keeps whole rounds of 256 x 256 tiles: four waves of 128 x 128, assembly main loop (gemm_wide.hip)
  if (gemm_wide_wanted(p, flags, num_cus())) PM_DISPATCH_DTYPE(dtype, T, return (launch_gemm_wide<T>(p, num_cus(), (hipStream_t)stream)));
  // large MFMA-bound shapes: 256x256 tiles, 8-phase ping-pong (gemm256.hip) - where the 256x128 ring kernel is not preferred
  if (!(g_ringw == 1 && g_ring != 0 && !(flags & PM_FLAG_A_F32) && prefer_ringw(A_DENSE, p)) && !(g_ringw == 2 && g_ring == 2) &&
      gemm256_wanted(p, flags, num_cus()))
    PM_DISPATCH_DTYPE(dtype, T, return (launch_gemm256<T>(p, num_cus(), (hipStream_t)stream)));
  PM_DISPATCH_DTYPE(dtype, T, return (launch<T, A_DENSE>(p, flags, (hipStream_t)stream)));
}

extern "C" int pm_conv2d_3x3(const void* x, int64_t ldx, const void* Wp, const float* bias,
                             const void* residual, int64_t ldr, void* y, int64_t ldy, int64_t F,
                             int64_t H, int64_t W, int64_t Cin, int64_t Cout, int stride,
                             int upsample2x, int pad_lo, const void* zero_page, int flags, int dtype,
                             void* workspace, size_t workspace_bytes, float* colstats, void* stream) {
  if (!zero_page) return PM_E_NULL;
  if (stride != 1 && stride != 2) return PM_E_SHAPE;
  if (upsample2x && stride != 1) return PM_E_SHAPE;
  if (pad_lo != 0 && pad_lo != 1) return PM_E_SHAPE;
  if ((Cin & 7) || (ldx & ((flags & PM_FLAG_A_F32) ? 3 : 7)) || ldx < Cin) return PM_E_SHAPE;
  if (!fits_u32(F * H * W * ldx, flags)) return PM_E_SHAPE;  // 32-bit lane offsets: chunk the frames
  const int64_t Hv = upsample2x ? 2 * H : H, Wv = upsample2x ? 2 * W : W;
  // zero padding: pad_lo rows/cols before, 1 after (pad_lo = 0: the first-stage encoder's (0,1,0,1) pad)
  const int64_t Ho = (Hv + pad_lo - 2) / stride + 1, Wo = (Wv + pad_lo - 2) / stride + 1;
  const int64_t M = F * Ho * Wo, K = 9 * Cin;
  int rc = check_common(x, Wp, y, M, Cout, K, PM_ACT_NONE);
  if (rc) return rc;
  GemmParams p{};
  p.A = x; p.lda = ldx; p.Wt = Wp; p.ldw = K; p.bias = bias; p.R = residual; p.ldr = ldr;
  p.C = y; p.ldc = ldy; p.M = (int)M; p.N = (int)Cout; p.K = (int)K; p.act = PM_ACT_NONE;
  p.out32 = (flags & PM_FLAG_OUT_F32) ? 1 : 0;
  p.res32 = (flags & (PM_FLAG_OUT_F32 | PM_FLAG_RES_F32)) ? 1 : 0;
  p.colstats = colstats;
  if (colstats != nullptr && (flags & PM_FLAG_STATS_I64)) {  // r06: int64 group totals [NI][32][4] instead of column sums
    p.gs_ni = (flags >> 8) & 0xff;
    if (p.gs_ni < 1 || (p.M % p.gs_ni) || (p.N & 31)) return PM_E_SHAPE;
    p.gs_rows = p.M / p.gs_ni;
  }
  p.ntiles = (int)((Cout + BN - 1) / BN);
  p.Hin = (int)H; p.Win = (int)W; p.Hv = (int)Hv; p.Wv = (int)Wv; p.Cin = (int)Cin;
  p.Ho = (int)Ho; p.Wo = (int)Wo; p.stride = stride; p.ups = upsample2x ? 1 : 0; p.pad = pad_lo;
  p.zero = zero_page;
  p.a_bytes = ((F * H * W - 1) * ldx + Cin) * ((flags & PM_FLAG_A_F32) ? 4 : 2);
  for (int t = 0; t < 9; ++t) {
    p.tap_a[t] = ((int64_t)(t / 3 - 1) * W + (t % 3 - 1)) * ldx * 2;
    p.tap_w[t] = (int64_t)t * Cin * 2;
  }
  plan_split(p, workspace, workspace_bytes, /*two_stage=*/upsample2x || (Cin % BK) != 0);
  if (!GS_EPILOGUE && p.gs_ni > 0 && p.splits <= 1) return PM_E_SHAPE;  // (integer totals from an unsplit call: diagnostics build only)
  if (p.colstats != nullptr && p.splits > 1 && ((p.ldc & 3) || (p.R != nullptr && (p.ldr & 3)))) return PM_E_SHAPE;
  if (!upsample2x && (Cin % BK) == 0)
    PM_DISPATCH_DTYPE(dtype, T, return (launch<T, A_CONV3X3_FAST>(p, flags, (hipStream_t)stream)));
  PM_DISPATCH_DTYPE(dtype, T, return (launch<T, A_CONV3X3>(p, flags, (hipStream_t)stream)));
}

extern "C" int pm_conv_temporal_k3(const void* x, int64_t ldx, const void* halo_lo,
                                   const void* halo_hi, const void* Wp, const float* bias,
                                   const void* residual, int64_t ldr, void* y, int64_t ldy,
                                   int64_t F, int64_t P, int64_t Cin, int64_t Cout,
                                   const void* zero_page, int flags, int dtype, void* workspace,
                                   size_t workspace_bytes, float* colstats, void* stream) {
  return pm_conv_temporal_k3_clips(x, ldx, halo_lo, halo_hi, Wp, bias, residual, ldr, y, ldy, F, F, P, Cin, Cout, zero_page,
                                   flags, dtype, workspace, workspace_bytes, colstats, stream);
}

extern "C" int pm_conv_temporal_k3_clips(const void* x, int64_t ldx, const void* halo_lo,
                                         const void* halo_hi, const void* Wp, const float* bias,
                                         const void* residual, int64_t ldr, void* y, int64_t ldy,
                                         int64_t F, int64_t clip_frames, int64_t P, int64_t Cin, int64_t Cout,
                                         const void* zero_page, int flags, int dtype, void* workspace,
                                         size_t workspace_bytes, float* colstats, void* stream) {
  if (!zero_page) return PM_E_NULL;
  if (clip_frames < 1 || F % clip_frames) return PM_E_SHAPE;
  if (clip_frames != F && (halo_lo || halo_hi)) return PM_E_SHAPE;  // (halo frames belong to ONE clip's ends)
  if ((Cin & 7) || (ldx & ((flags & PM_FLAG_A_F32) ? 3 : 7)) || ldx < Cin) return PM_E_SHAPE;
  if (Cin % BK) return PM_E_SHAPE;  // one tap per K-tile
  if (!fits_u32(F * P * ldx, flags)) return PM_E_SHAPE;
  const int64_t M = F * P, K = 3 * Cin;
  int rc = check_common(x, Wp, y, M, Cout, K, PM_ACT_NONE);
  if (rc) return rc;
  GemmParams p{};
  p.A = x; p.lda = ldx; p.Wt = Wp; p.ldw = K; p.bias = bias; p.R = residual; p.ldr = ldr;
  p.C = y; p.ldc = ldy; p.M = (int)M; p.N = (int)Cout; p.K = (int)K; p.act = PM_ACT_NONE;
  p.out32 = (flags & PM_FLAG_OUT_F32) ? 1 : 0;
  p.res32 = (flags & (PM_FLAG_OUT_F32 | PM_FLAG_RES_F32)) ? 1 : 0;
  p.colstats = colstats;
  if (colstats != nullptr && (flags & PM_FLAG_STATS_I64)) {  // r06: int64 group totals [NI][32][4] instead of column sums
    p.gs_ni = (flags >> 8) & 0xff;
    if (p.gs_ni < 1 || (p.M % p.gs_ni) || (p.N & 31)) return PM_E_SHAPE;
    p.gs_rows = p.M / p.gs_ni;
  }
  p.ntiles = (int)((Cout + BN - 1) / BN);
  p.Cin = (int)Cin; p.F = (int)F; p.Fc = (int)clip_frames; p.P = (int)P; p.halo_lo = halo_lo; p.halo_hi = halo_hi;
  p.zero = zero_page;
  p.a_bytes = ((F * P - 1) * ldx + Cin) * ((flags & PM_FLAG_A_F32) ? 4 : 2);
  for (int t = 0; t < 3; ++t) {
    p.tap_a[t] = (int64_t)(t - 1) * P * ldx * 2;
    p.tap_w[t] = (int64_t)t * Cin * 2;
  }
  plan_split(p, workspace, workspace_bytes);
  if (!GS_EPILOGUE && p.gs_ni > 0 && p.splits <= 1) return PM_E_SHAPE;  // (integer totals from an unsplit call: diagnostics build only)
  if (p.colstats != nullptr && p.splits > 1 && ((p.ldc & 3) || (p.R != nullptr && (p.ldr & 3)))) return PM_E_SHAPE;
  PM_DISPATCH_DTYPE(dtype, T, return (launch<T, A_CONVT3>(p, flags, (hipStream_t)stream)));
}

#ifdef PM_RING_PROF
extern "C" void pm_debug_ring_prof(void* buf) { g_ring_prof = reinterpret_cast<long long*>(buf); }
#endif

extern "C" int pm_gemm_colstats_rows(int64_t M, int64_t N, int64_t K, int act, size_t workspace_bytes) {
  int ktps;
  const int s = choose_splits(M, N, K, act, &ktps);
  const bool split = s > 1 && workspace_bytes >= (size_t)s * M * N * sizeof(float) && (N & 3) == 0 && act != PM_ACT_GEGLU;
  return split ? 16 : 64;
}

extern "C" int pm_gemm_kernel_choice(int64_t M, int64_t N, int64_t K, int act, int flags, size_t workspace_bytes) {
  // mirrors pm_gemm's own decision (plan_split + launch<T, A_DENSE>): which kernel a dense call of this shape runs on
  if (M < 1 || N < 1 || K < BK || (K % BK)) return PM_E_SHAPE;
  if (flags & PM_FLAG_A_F32) return 0;
  init_once();
  GemmParams p{};
  p.M = (int)M; p.N = (int)N; p.K = (int)K; p.act = act;
  p.ntiles = (int)((N + BN - 1) / BN);
  static float dummy_ws;  // (only its non-NULLness matters to plan_split)
  plan_split(p, workspace_bytes ? &dummy_ws : nullptr, workspace_bytes);
  // (dense 16-bit call of contiguous operands, no residual / statistics / f32 output: what the caller of this query describes)
  p.lda = K; p.ldw = K; p.ldc = act == PM_ACT_GEGLU ? N / 2 : N;
  if (gemm_wide_stream_wanted(p, flags, num_cus())) return 5;  // gemm_wide_stream (r06)
  if (gemm_wide_wanted(p, flags, num_cus())) return 4;         // gemm_wide (r06)
  if (p.splits == 1 && ((g_ringw == 2 && g_ring == 2) || (g_ringw == 1 && g_ring != 0 && prefer_ringw(A_DENSE, p))))
    return 3;  // the 256x128 ring kernel (launch<T, A_DENSE>'s first choice)
  if (!(g_ringw == 2 && g_ring == 2) && gemm256_wanted(p, flags, num_cus())) return 2;
  return (g_ring == 2 || (g_ring == 1 && prefer_ring(A_DENSE, p))) ? 1 : 0;
}

extern "C" size_t pm_gemm_workspace_bytes(int64_t M, int64_t N, int64_t K, int act) {
  int ktps;
  const int s = choose_splits(M, N, K, act, &ktps);
  return s > 1 ? (size_t)s * M * N * sizeof(float) : 0;
}
