// Flash-style attention for head dim 64 on gfx950 (spatial self-attention and the text+image
// cross-attention of the 3-D U-Net).  No score matrix ever reaches HBM.
//
// Workgroup = 4 waves, 128 query rows (32 per wave); K/V tiles of 64 keys are staged
// global -> VGPR -> LDS (two buffers, one barrier per tile).
//   S^T = K . Q^T      v_mfma_f32_32x32x16 with K as the A operand ("swapped QK^T"): every lane then
//                      owns ONE query row (col = lane & 31) and 32 of the tile's 64 scores, so the
//                      row max / row sum are in-lane reductions plus one exchange with lane ^ 32.
//   O^T += V^T . P^T   the S^T accumulator is re-used in place as the B operand (no LDS round trip,
//                      cdna guide §3 "accumulator tile as the next MFMA's operand"); V^T fragments
//                      come from the row-major V tile through ds_read_b64_tr_b16.
// Two key/value segments (text, image) are normalised independently and summed (attention.py:128-142).
#include "attn_common.hpp"

namespace pm {

// QB = 32-row query blocks per wave.  QB = 1: 128 query rows per workgroup (short sequences, the
// two-segment cross-attention).  QB = 2: 256 rows per workgroup - every K / V^T fragment read from LDS
// feeds two MFMAs and the two blocks' MFMA and softmax chains are independent, so the scheduler can
// overlap one block's exp/max/convert VALU work with the other's matrix work (the kernel is
// VALU-issue-bound at head dim 64: ~13 VALU per 32x32x16 MFMA).
template <typename T, int QB, bool SEG2>
__global__ __launch_bounds__(256, 2) void attn_kernel(const AttnParams p) {
  __shared__ __attribute__((aligned(16))) char smem[4 * KV_TILE_BYTES];  // K[2], V[2]
  char* const Ks = smem;
  char* const Vs = smem + 2 * KV_TILE_BYTES;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ql = lane & 31, hh = lane >> 5;
  // 1-D grid, XCD-aware: blocks are dealt round-robin over the 8 XCDs, so give each XCD a contiguous
  // run of ids = all query tiles of the same (frame, head) -> its K/V (re-read by every query tile)
  // stays in that XCD's L2 instead of being fetched by all eight
  int wg = blockIdx.x;
  {
    const int nwg = gridDim.x, qn = nwg >> 3, rn = nwg & 7, xcd = wg & 7;
    wg = ((xcd < rn) ? xcd * (qn + 1) : rn * (qn + 1) + (xcd - rn) * qn) + (wg >> 3);
  }
  const int bh = wg / p.nqt;
  const int qt = wg - bh * p.nqt;
  const int b = bh / p.heads, head = bh - b * p.heads;

  // Q^T fragments (B operand): element j of k-step s = Q[qrow][16 s + 8 hh + j]
  int qrow[QB];
  bool q_valid[QB];
  Pack8<T> qf[QB][4];
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
    qrow[qb] = qt * (128 * QB) + wave * (32 * QB) + qb * 32 + ql;
    q_valid[qb] = qrow[qb] < p.Nq;
    if (!q_valid[qb]) qrow[qb] = p.Nq - 1;
    const T* qp = reinterpret_cast<const T*>(p.q) + (int64_t)b * p.q_bs + (int64_t)qrow[qb] * p.q_rs +
                  head * 64 + 8 * hh;
#pragma unroll
    for (int s = 0; s < 4; ++s) qf[qb][s].u = ld_global16(qp + 16 * s);
  }

  // staging coordinates
  const int lc = tid & 7, lr = tid >> 3;  // chunk, row (+32)

  // two-segment (text + image) calls sum the independently normalised segment outputs here
  f32x16 out[SEG2 ? QB : 1][2];
  if constexpr (SEG2) {
#pragma unroll
    for (int qb = 0; qb < QB; ++qb)
#pragma unroll
      for (int d = 0; d < 2; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) out[qb][d][r] = 0.f;
  }

  for (int seg = 0; seg < (SEG2 ? p.nseg : 1); ++seg) {
    const int Nk = p.Nk[seg];
    // wave-uniform bases (this block's batch element and head); lanes add 32-bit byte offsets, so the DMAs use
    // the "scalar base + 32-bit lane offset" address form (a per-lane 64-bit address costs the DMA about twice
    // the issue time: measured on the GEMM loaders, profiles/r01/ring_gemm_inkernel_stamps.txt)
    const char* const kbase = reinterpret_cast<const char*>(reinterpret_cast<const T*>(p.k[seg]) + (int64_t)b * p.k_bs[seg] + head * 64);
    const char* const vbase = reinterpret_cast<const char*>(reinterpret_cast<const T*>(p.v[seg]) + (int64_t)b * p.k_bs[seg] + head * 64);
    const uint32_t krs2 = (uint32_t)(p.k_rs[seg] * 2);  // row stride in bytes (Nk * row bytes < 4 GiB: checked by the launcher)
    const int nkt = (Nk + KV_TILE - 1) / KV_TILE;

    // K/V tiles go global -> LDS by DMA (1 KiB per wave-instruction, lane-linear image); the XOR
    // swizzles are applied on the source side: this lane fills physical slot tid & 7 of rows
    // lr + 32 j with logical chunk  slot ^ ((row >> 1) & 7)  (K: conflict-free ds_read_b128 for the
    // 32-row MFMA operand) resp.  slot ^ (((row >> 1) & 1) << 2)  (V: conflict-free transposed reads).
    const int kc = (tid & 7) ^ ((lr >> 1) & 7);
    const int vc = (tid & 7) ^ (((lr >> 1) & 1) << 2);
    auto load_kv = [&](int kt, int buf) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        int key = kt * KV_TILE + lr + 32 * j;
        if (key > Nk - 1) key = Nk - 1;
        const int dst = buf * KV_TILE_BYTES + (32 * j + 8 * wave) * 128;
        const uint32_t row = (uint32_t)key * krs2;
        __builtin_amdgcn_global_load_lds((glb_void*)(kbase + (row + (uint32_t)kc * 16)), (lds_void*)(Ks + dst), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((glb_void*)(vbase + (row + (uint32_t)vc * 16)), (lds_void*)(Vs + dst), 16, 0, 0);
      }
    };

    f32x16 oacc[QB][2];
    float m_run[QB], l_run[QB];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
      m_run[qb] = -INFINITY;
      l_run[qb] = 0.f;
#pragma unroll
      for (int d = 0; d < 2; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[qb][d][r] = 0.f;
    }

    load_kv(0, 0);
    __syncthreads();  // drains the DMA (vmcnt) before the barrier

    for (int kt = 0; kt < nkt; ++kt) {
      const int buf = kt & 1;
      if (kt + 1 < nkt) load_kv(kt + 1, buf ^ 1);
      const char* ks = Ks + buf * KV_TILE_BYTES;
      const char* vs = Vs + buf * KV_TILE_BYTES;

      // ---- S^T = K . Q^T (raw scores, f32); each K fragment serves every query block ----
      f32x16 sacc[QB][2];
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
        for (int qb = 0; qb < QB; ++qb)
#pragma unroll
          for (int r = 0; r < 16; ++r) sacc[qb][kb][r] = 0.f;
        const int row = kb * 32 + ql;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          Pack8<T> kf;
          const int chunk = 2 * s + hh;
          kf.u = *reinterpret_cast<const u32x4*>(ks + row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4));
#pragma unroll
          for (int qb = 0; qb < QB; ++qb) sacc[qb][kb] = mfma32(kf.v, qf[qb][s].v, sacc[qb][kb]);
        }
      }
      // sacc[qb][kb][r] <-> key = kt*64 + kb*32 + (r&3) + 8*(r>>2) + 4*hh, query = ql of block qb
      if (kt * KV_TILE + KV_TILE > Nk) {
#pragma unroll
        for (int qb = 0; qb < QB; ++qb)
#pragma unroll
          for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int key = kt * KV_TILE + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
              if (key >= Nk) sacc[qb][kb][r] = -INFINITY;
            }
      }
      // ---- online softmax (base-2 domain) ----
#pragma unroll
      for (int qb = 0; qb < QB; ++qb) {
        float mx = sacc[qb][0][0];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
          for (int r = 0; r < 16; ++r) mx = fmaxf(mx, sacc[qb][kb][r]);
        mx = fmaxf(mx, other_half(mx));
        const float m_new = fmaxf(m_run[qb], mx * p.scale_log2e);
        // rescale only when some row's running max actually grew (alpha == 1 exactly otherwise)
        if (__builtin_amdgcn_ballot_w64(m_new > m_run[qb]) != 0) {
          const float alpha = __builtin_amdgcn_exp2f(m_run[qb] - m_new);
          l_run[qb] *= alpha;
#pragma unroll
          for (int d = 0; d < 2; ++d)
#pragma unroll
            for (int r = 0; r < 16; ++r) oacc[qb][d][r] *= alpha;
          m_run[qb] = m_new;
        }
        float psum = 0.f;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const float pv = __builtin_amdgcn_exp2f(sacc[qb][kb][r] * p.scale_log2e - m_run[qb]);
            sacc[qb][kb][r] = pv;
            psum += pv;
          }
        l_run[qb] += psum;
      }

      // ---- O^T += V^T . P^T; each V^T fragment serves every query block ----
      const int trow = (lane & 15) >> 2;                 // row inside the 4x16 block
      const int tcol8 = ((lane >> 4) & 1) * 2 + ((lane & 3) >> 1);  // 16-byte chunk inside the d-block
      const int tsub = (lane & 1) * 8;                   // byte inside the chunk
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) {
        Pack8<T> pf[QB];
        const int kb = s4 >> 1, sp = s4 & 1;
#pragma unroll
        for (int qb = 0; qb < QB; ++qb)
#pragma unroll
          for (int j = 0; j < 8; ++j) pf[qb].e[j] = from_f32<T>(sacc[qb][kb][8 * sp + j]);
#pragma unroll
        for (int db = 0; db < 2; ++db) {
          const int r0 = 16 * s4 + 4 * hh + trow;
          const int r1 = r0 + 8;
          const int ch = db * 4 + tcol8;
          const int off0 = r0 * 128 + ((ch ^ (((r0 >> 1) & 1) << 2)) << 4) + tsub;
          const int off1 = r1 * 128 + ((ch ^ (((r1 >> 1) & 1) << 2)) << 4) + tsub;
          typename Vec<T>::v8 vf = tr_pair<T>(vs, off0, off1);
#pragma unroll
          for (int qb = 0; qb < QB; ++qb) oacc[qb][db] = mfma32(vf, pf[qb].v, oacc[qb][db]);
        }
      }
      __syncthreads();
    }
    // oacc[qb][db][r] <-> d = db*32 + (r&3) + 8*(r>>2) + 4*hh for query row ql of block qb
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
      const float l_tot = l_run[qb] + other_half(l_run[qb]);
      const float inv = p.w[seg] / l_tot;
      if constexpr (SEG2) {
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
          for (int r = 0; r < 16; ++r) out[qb][d][r] += oacc[qb][d][r] * inv;
        if (seg + 1 < p.nseg) continue;
      }
      if (q_valid[qb]) {
        T* op = reinterpret_cast<T*>(p.o) + (int64_t)b * p.o_bs + (int64_t)qrow[qb] * p.o_rs + head * 64;
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            Pack4<T> ov;
#pragma unroll
            for (int e = 0; e < 4; ++e)
              ov.e[e] = from_f32<T>(SEG2 ? out[qb][db][4 * g + e] : oacc[qb][db][4 * g + e] * inv);
            *reinterpret_cast<u32x2*>(op + db * 32 + 8 * g + 4 * hh) = ov.u;
          }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Single-segment attention (the spatial self-attention: 19 % of a 576x1024 forward's FLOPs), VALU diet.
//
// At head dim 64 a 32-query x 64-key tile is 16 MFMAs (512 matrix cycles) against 2048 scores of softmax, so
// the kernel lives or dies by the vector instructions per score.  attn_kernel spends ~6.5 (max, scale-and-
// subtract FMA, exp, sum, convert, rescale bookkeeping); this one ~3:
//   * Q arrives pre-multiplied by scale*log2(e) (the host folds it into the to_q weights; otherwise the
//     fragments are scaled once here), so scores are already in the base-2 domain: no per-score multiply;
//   * the running maximum is subtracted BY THE MFMA: the S^T chain's initial accumulator is a register block
//     holding -m (nm[]), so the chain ends in s - m and p = exp2(S') needs no subtraction either.  m is the
//     "stale" maximum: it is only raised when a tile's scores exceed it by more than STALE_THR (p stays
//     <= 2^STALE_THR: harmless for f32 accumulation and for the 16-bit P operand, whose relative precision
//     does not depend on magnitude), which after the first tiles almost never happens, so the O/l rescale
//     and the rewrite of nm[] leave the steady state: per score there remain 1/2 v_max3 (the check), v_exp,
//     v_add (row sum) and 1/2 v_cvt_pk;
//   * the first tile and a ragged last tile run a "careful" variant (exact maximum, key masking): none of
//     that is in the hot loop;
//   * 64 query rows per wave as two 32-row blocks whose chains are issued block after block
//     (S'(0) | max(0) | S'(1) || exp(0) | max(1) | PV(0) || exp(1) | PV(1)), so one block's VALU work sits
//     beside the other block's MFMAs inside one wave, on top of the overlap between the two waves of a SIMD.
constexpr float STALE_THR = 6.0f;
constexpr float STALE_SUM = 1024.0f;

//
// PROBE > 0 (diagnosis builds of the production configuration, reached through pm_debug_attn_variant 11 / 12 / 13;
// their OUTPUT IS NOT AN ATTENTION RESULT): the same per-tile instruction stream with parts taken away, to measure what
// this chip sustains for the mix at the clock it holds (profiles/r03/attention_ceiling.txt):
//   1  no global traffic in the steady state: the first two K/V tiles stay in LDS and are re-read for every tile
//      (same 16 MFMAs, same 2048-score softmax stream, same LDS fragment reads, same barriers);
//   2  as 1 without the softmax's vector instructions (P = the low halves of S', no max / exp / sum / convert);
//   3  as 2 without the LDS fragment reads (K and V fragments stay in registers): the bare MFMA stream.
template <typename T, int QB, bool PIPE, bool POST, int RING, int PROBE = 0>
__global__ __launch_bounds__(256, (QB == 1 && RING == 2) ? 3 : 2) void attn_self_kernel(const AttnParams p) {
  constexpr bool NOBAR = false;
  __shared__ __attribute__((aligned(16))) char smem[2 * RING * KV_TILE_BYTES];  // K[RING], V[RING]
  char* const Ks = smem;
  char* const Vs = smem + RING * KV_TILE_BYTES;

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ql = lane & 31, hh = lane >> 5;
  int wg = blockIdx.x;
  {  // XCD-aware: all query tiles of one (frame, head) run on one XCD (its K/V stays in that L2)
    const int nwg = gridDim.x, qn = nwg >> 3, rn = nwg & 7, xcd = wg & 7;
    wg = ((xcd < rn) ? xcd * (qn + 1) : rn * (qn + 1) + (xcd - rn) * qn) + (wg >> 3);
  }
  const int bh = wg / p.nqt;
  const int qt = wg - bh * p.nqt;
  const int b = bh / p.heads, head = bh - b * p.heads;

  unsigned long long t_clk0 = 0, t_real0 = 0;
  if constexpr (PM_DIAG_BUILD) {
    if (p.stamps) {
      t_clk0 = __builtin_amdgcn_s_memtime();
      t_real0 = __builtin_amdgcn_s_memrealtime();
    }
  }

  int qrow[QB];
  bool q_valid[QB];
  Pack8<T> qf[QB][4];
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
    qrow[qb] = qt * (128 * QB) + wave * (32 * QB) + qb * 32 + ql;
    q_valid[qb] = qrow[qb] < p.Nq;
    if (!q_valid[qb]) qrow[qb] = p.Nq - 1;
    const T* qp = reinterpret_cast<const T*>(p.q) + (int64_t)b * p.q_bs + (int64_t)qrow[qb] * p.q_rs +
                  head * 64 + 8 * hh;
#pragma unroll
    for (int s = 0; s < 4; ++s) qf[qb][s].u = ld_global16(qp + 16 * s);
  }
  if (p.prescaled == 0) {  // (uniform) general callers: one extra 16-bit rounding of q * scale * log2(e)
#pragma unroll
    for (int qb = 0; qb < QB; ++qb)
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int e = 0; e < 8; ++e) qf[qb][s].e[e] = from_f32<T>(to_f32(qf[qb][s].e[e]) * p.scale_log2e);
  }

  const int lr = tid >> 3;
  const int Nk = p.Nk[0];
  const char* const kbase = reinterpret_cast<const char*>(reinterpret_cast<const T*>(p.k[0]) + (int64_t)b * p.k_bs[0] + head * 64);
  const char* const vbase = reinterpret_cast<const char*>(reinterpret_cast<const T*>(p.v[0]) + (int64_t)b * p.k_bs[0] + head * 64);
  const uint32_t krs2 = (uint32_t)(p.k_rs[0] * 2);
  const int nkt = (Nk + KV_TILE - 1) / KV_TILE;
  const int kc = (tid & 7) ^ ((lr >> 1) & 7);           // source-side swizzles: see attn_kernel
  const int vc = (tid & 7) ^ (((lr >> 1) & 1) << 2);
  const int last_full = (Nk % KV_TILE == 0) ? nkt : nkt - 1;  // tiles [0, last_full) lie wholly inside the sequence
  // Those tiles are fetched with `buffer_load ... lds`: per-lane offset fixed for the whole kernel, the tile's position in
  // the SCALAR offset, the LDS stage in M0 - no vector instruction for addresses in the K loop (the global_load_lds form
  // recomputes clamped 64-bit per-lane addresses, ~20 VALU incl. two v_mul_lo_u32 per tile, on a loop bound by vector
  // issue).  (range = through the head slice of the last key row; no tile fetched this way reaches it)
  const uint32_t kv_range = (uint32_t)(Nk - 1) * krs2 + 128u;
  const __amdgpu_buffer_rsrc_t krsrc = __builtin_amdgcn_make_buffer_rsrc((void*)kbase, (short)0, (int)kv_range, 0x00020000);
  const __amdgpu_buffer_rsrc_t vrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)vbase, (short)0, (int)kv_range, 0x00020000);
  const uint32_t kvoff = (uint32_t)lr * krs2 + (uint32_t)kc * 16, vvoff = (uint32_t)lr * krs2 + (uint32_t)vc * 16;
  auto load_kv = [&](int kt, int buf) {
    if (kt < last_full) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const uint32_t soff = (uint32_t)(kt * KV_TILE + 32 * j) * krs2;
        const int dst = buf * KV_TILE_BYTES + (32 * j + 8 * wave) * 128;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(krsrc, (lds_void*)(Ks + dst), 16, kvoff, soff, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(vrsrc, (lds_void*)(Vs + dst), 16, vvoff, soff, 0, 0);
      }
      return;
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {  // ragged last tile: rows clamped to the last key (masked by the careful tile)
      int key = kt * KV_TILE + lr + 32 * j;
      if (key > Nk - 1) key = Nk - 1;
      const int dst = buf * KV_TILE_BYTES + (32 * j + 8 * wave) * 128;
      const uint32_t row = (uint32_t)key * krs2;
      __builtin_amdgcn_global_load_lds((glb_void*)(kbase + (row + (uint32_t)kc * 16)), (lds_void*)(Ks + dst), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((glb_void*)(vbase + (row + (uint32_t)vc * 16)), (lds_void*)(Vs + dst), 16, 0, 0);
    }
  };

  f32x16 oacc[QB][2], nm[QB];  // nm: every register = -m_run, the initial accumulator of the S^T chains
  float m_run[QB], l_run[QB];
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
    m_run[qb] = 0.f;
    l_run[qb] = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) nm[qb][r] = 0.f;
    asm volatile("" : "+v"(nm[qb]));
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
      for (int r = 0; r < 16; ++r) oacc[qb][d][r] = 0.f;
  }

  // transposed-read lane coordinates of the V^T fragments (see attn_kernel)
  const int trow = (lane & 15) >> 2;
  const int tcol8 = ((lane >> 4) & 1) * 2 + ((lane & 3) >> 1);
  const int tsub = (lane & 1) * 8;

  // raise the stale maximum of block qb by the row maximum of the pending tile (mx: this lane's half)
  auto raise = [&](int qb, f32x16 (&sacc)[2], float mx, bool first) {
    float rmx = fmaxf(mx, other_half(mx));
    if (!first) {
      rmx = fmaxf(rmx, 0.f);  // never lower m: rows that did not outgrow it keep alpha == 1 exactly
      const float alpha = __builtin_amdgcn_exp2f(-rmx);
      l_run[qb] *= alpha;
#pragma unroll
      for (int d = 0; d < 2; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[qb][d][r] *= alpha;
    }
    m_run[qb] += rmx;
#pragma unroll
    for (int r = 0; r < 16; ++r) nm[qb][r] = -m_run[qb];
    // opaque to the optimiser: knowing the 16 registers equal, hipcc re-broadcasts them from one scalar in
    // every tile (15 v_mov per block and tile - the very instructions this block exists to save)
    asm volatile("" : "+v"(nm[qb]));
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) sacc[kb][r] -= rmx;
  };

  // one K/V tile.  CAREFUL: exact maximum + key masking (first tile, ragged last tile).
  auto tile = [&](int kt, auto bufc, auto careful_c) {
    constexpr bool CAREFUL = decltype(careful_c)::value;
    const int buf = bufc;  // (an integral_constant in the fast loop: LDS offsets fold into the reads' immediates)
    const char* ks = Ks + buf * KV_TILE_BYTES;
    const char* vs = Vs + buf * KV_TILE_BYTES;
    Pack8<T> pf[QB][4];
    auto pv = [&](int qb) {  // O^T += V^T . P^T
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) {
#pragma unroll
        for (int db = 0; db < 2; ++db) {
          const int r0 = 16 * s4 + 4 * hh + trow;
          const int r1 = r0 + 8;
          const int ch = db * 4 + tcol8;
          const int off0 = r0 * 128 + ((ch ^ (((r0 >> 1) & 1) << 2)) << 4) + tsub;
          const int off1 = r1 * 128 + ((ch ^ (((r1 >> 1) & 1) << 2)) << 4) + tsub;
          typename Vec<T>::v8 vf;
          if constexpr (PROBE >= 3) {
            vf = qf[qb][(s4 + db) & 3].v;  // (any resident register operand)
          } else {
            vf = tr_pair<T>(vs, off0, off1);
          }
          oacc[qb][db] = mfma32(vf, pf[qb][s4].v, oacc[qb][db]);
        }
      }
    };
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
      // ---- S'^T = K . Q'^T - m : the chain starts from nm (K fragments are re-read per block: LDS reads are
      // cheap here, registers are not) ----
      f32x16 sacc[2];
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
        const int row = kb * 32 + ql;
        Pack8<T> kf[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const int chunk = 2 * s + hh;
          if constexpr (PROBE >= 3) {
            kf[s].u = qf[qb][(s + kb) & 3].u;
          } else {
            kf[s].u = *reinterpret_cast<const u32x4*>(ks + row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4));
          }
        }
        sacc[kb] = mfma32(kf[0].v, qf[qb][0].v, nm[qb]);
#pragma unroll
        for (int s = 1; s < 4; ++s) sacc[kb] = mfma32(kf[s].v, qf[qb][s].v, sacc[kb]);
      }
      if constexpr (PIPE) {
        if (qb > 0) pv(qb - 1);  // the previous block's P.V sits between this block's S' chain and its softmax
      }
      if constexpr (CAREFUL) {
        if (kt * KV_TILE + KV_TILE > Nk) {
#pragma unroll
          for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int key = kt * KV_TILE + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
              if (key >= Nk) sacc[kb][r] = -INFINITY;
            }
        }
      }
      // this lane's maximum over its 32 scores of the tile
      auto lane_max = [&]() -> float {
        float m0 = fmaxf(sacc[0][0], sacc[1][0]);
#pragma unroll
        for (int r = 1; r < 16; ++r) m0 = fmaxf(fmaxf(m0, sacc[0][r]), sacc[1][r]);
        return m0;
      };
      // p = 2^S', 16-bit P operand; returns this lane's sum of p
      auto softmax = [&]() -> float {
        float ps[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const float pe = __builtin_amdgcn_exp2f(sacc[kb][r]);
            ps[r & 3] += pe;
            pf[qb][kb * 2 + (r >> 3)].e[r & 7] = from_f32<T>(pe);
          }
        return (ps[0] + ps[1]) + (ps[2] + ps[3]);
      };
      float psum;
      if constexpr (PROBE >= 2 && !CAREFUL) {
        // no softmax: the P operand is made of the S' registers as they stand (keeps the S' -> P.V dependence)
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
          for (int h2 = 0; h2 < 2; ++h2) {
            union { float f[4]; u32x4 u; } t;
#pragma unroll
            for (int e = 0; e < 4; ++e) t.f[e] = sacc[kb][8 * h2 + e];
            pf[qb][kb * 2 + h2].u = t.u;
          }
        psum = sacc[1][15];
      } else if constexpr (CAREFUL) {
        raise(qb, sacc, lane_max(), kt == 0);
        psum = softmax();
      } else if constexpr (POST) {
        // no maximum at all in the steady state: exponentiate against the stale m and look at the row sum that
        // is formed anyway - a score more than 10 above m (or an overflow: inf) shows as a lane sum > 2^10;
        // only then the exact maximum is taken, m raised and the tile's P redone (nothing of it has been
        // accumulated yet).  Below that bound every p <= 2^10: safe for f32 sums and the 16-bit operand.
        psum = softmax();
        if (__builtin_amdgcn_ballot_w64(!(psum <= STALE_SUM)) != 0) {
          raise(qb, sacc, lane_max(), false);
          psum = softmax();
        }
      } else {
        const float m0 = lane_max();
        if (__builtin_amdgcn_ballot_w64(m0 > STALE_THR) != 0) raise(qb, sacc, m0, false);
        psum = softmax();
      }
      l_run[qb] += psum;
      if constexpr (!PIPE) {
        pv(qb);
        __builtin_amdgcn_sched_barrier(0);  // keep the blocks apart: interleaved, their live ranges spill
      }
    }
    if constexpr (PIPE) pv(QB - 1);
  };

  // K/V ring: RING stages of 64 keys, the DMAs of the next RING-1 tiles in flight ACROSS the per-tile barrier
  // (counted s_waitcnt vmcnt + raw s_barrier; a __syncthreads() would drain them: with one tile of look-ahead
  // every tile paid the L2 / Infinity-Cache round trip, MFMA pipes 46 % busy at 2.0 GHz with the VALU diet).
  // Tile t lives in stage t % RING.  arrive(t): this wave's share of tile t has landed (each tile is 4 DMAs
  // per wave, younger tiles may stay in flight), then the barrier publishes everybody's share and proves that
  // every wave is done with tile t-1, whose stage the load issued right behind it (tile t+RING-1) overwrites.
  auto arrive = [&](int t) {
    if constexpr (RING == 2) {  // classic double buffer: 32 KiB of LDS, three workgroups per CU at 32 rows per wave
      __syncthreads();          // (drains this wave's DMAs, publishes tile t, frees tile t-1's stage)
      if (t + 1 < nkt && (PROBE == 0 || t == 0)) load_kv(t + 1, (t + 1) & 1);
    } else {
      const int ahead = (nkt - 1 - t < RING - 2) ? nkt - 1 - t : RING - 2;  // younger tiles already issued
      if (ahead >= 2)
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else if (ahead == 1)
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if constexpr (!NOBAR) __builtin_amdgcn_s_barrier();
      if (t + RING - 1 < nkt) load_kv(t + RING - 1, (t + RING - 1) % RING);
    }
  };
  static_assert(RING == 4 || RING == 2, "the vmcnt ladder above is written for 3 tiles of look-ahead");
#pragma unroll
  for (int t = 0; t < RING - 1; ++t)
    if (t < nkt) load_kv(t, t);
  // three separate loops (careful first tile | fast tiles | careful ragged tile): with the two variants as the
  // arms of one loop body hipcc copied the 16-register accumulator tuples at every merge (~50 v_mov per tile)
  arrive(0);  // (tiles [1, last_full) need no masking)
  tile(0, std::integral_constant<int, 0>{}, std::true_type{});
  int kt = 1;
  for (; kt + 3 < last_full; kt += 4) {  // four tiles per trip: every tile's stage is a compile-time constant
    arrive(kt);
    tile(kt, std::integral_constant<int, 1>{}, std::false_type{});
    arrive(kt + 1);
    tile(kt + 1, std::integral_constant<int, 2 % RING>{}, std::false_type{});
    arrive(kt + 2);
    tile(kt + 2, std::integral_constant<int, 3 % RING>{}, std::false_type{});
    arrive(kt + 3);
    tile(kt + 3, std::integral_constant<int, 0 % RING>{}, std::false_type{});
  }
  for (; kt < last_full; ++kt) {
    arrive(kt);
    tile(kt, kt % RING, std::false_type{});
  }
  if (nkt > 1 && last_full < nkt) {
    arrive(nkt - 1);
    tile(nkt - 1, (nkt - 1) % RING, std::true_type{});
  }

#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
    const float l_tot = l_run[qb] + other_half(l_run[qb]);
    const float inv = 1.0f / l_tot;
    if (q_valid[qb]) {
      T* op = reinterpret_cast<T*>(p.o) + (int64_t)b * p.o_bs + (int64_t)qrow[qb] * p.o_rs + head * 64;
#pragma unroll
      for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          Pack4<T> ov;
#pragma unroll
          for (int e = 0; e < 4; ++e) ov.e[e] = from_f32<T>(oacc[qb][db][4 * g + e] * inv);
          *reinterpret_cast<u32x2*>(op + db * 32 + 8 * g + 4 * hh) = ov.u;
        }
    }
  }
  if constexpr (PM_DIAG_BUILD) {
    if (p.stamps && tid == 0) {  // in-kernel clock = d(s_memtime) / d(s_memrealtime) x 100 MHz (diagnostics build only)
      unsigned long long* st = p.stamps + 2 * (size_t)blockIdx.x;
      st[0] = __builtin_amdgcn_s_memtime() - t_clk0;
      st[1] = __builtin_amdgcn_s_memrealtime() - t_real0;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// fp8 (OCP e4m3) attention on the block-scaled MFMA, v_mfma_scale_f32_32x32x64_f8f6f4 with unit scales: K = 64 per
// instruction at twice the bf16 rate, so a 32-query x 64-key tile is 2 + 2 MFMAs of 64 cycles (256 matrix cycles)
// instead of 8 + 8 of 32 (512), and every operand is 32 contiguous bytes per lane:
//   S'^T = K . Q'^T - m   A = K tile rows [key][64 d] (lane: key = lane & 31, d = 32 (lane >> 5) .. + 31),
//                         B = Q'^T from registers, C = the -m block (stale maximum, as attn_self_kernel);
//   O^T += V^T . P^T      B = P^T: the S' accumulators ARE that operand once converted (v_cvt_pk_fp8_f32): a lane
//                         holds, in register order, keys 8 g + 4 h + e (g = 0..7 dword, e byte, h = lane >> 5);
//                         A = V^T tile rows [d][64 keys] whose keys are stored in exactly that order per lane half
//                         (byte 32 h + 4 g + e of a row = key 8 g + 4 h + e): pm::attn_fp8_pack_kernel writes Q' and
//                         K as fp8 rows and V transposed with that key permutation, once per call.
// (The pairing of A and B elements inside the instruction is by (lane half, element index): any k order the two
// operands share is legal - verified with exact integers by tools/probes/mfma_scale_fp8_layout.hip.)
// LDS: K and V^T tiles are 64 rows x 64 bytes; the 16-byte chunk index is XOR-ed with (row >> 2) & 3 on the DMA's
// source side, which makes the 2 x ds_read_b128 of a lane's 32 bytes conflict-free.
typedef __attribute__((ext_vector_type(8))) int i32x8;

struct AttnFp8Params {
  const unsigned char* q8;   // [B*heads][Nq_pad][64]
  const unsigned char* k8;   // [B*heads][Nk_pad][64]
  const unsigned char* v8t;  // [B*heads][64][Nk_pad], keys permuted per 64-key tile
  void* o;
  int64_t o_bs, o_rs;
  int Nq, Nk, Nq_pad, Nk_pad, heads, nqt;
};

constexpr int F8_TILE_BYTES = 64 * 64;  // one K or V^T tile
constexpr float F8_STALE_THR = 4.0f;    // p <= 2^4 (e4m3 tops out at 448)

__device__ __forceinline__ f32x16 mfma_f8(i32x8 a, i32x8 b, f32x16 c) {
  return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, 127, 0, 127);  // fp8 x fp8, scales 2^0
}

template <typename T, int QB>
__global__ __launch_bounds__(256, 2) void attn_fp8_kernel(const AttnFp8Params p) {
  constexpr int RING = 2;
  __shared__ __attribute__((aligned(16))) char smem[2 * RING * F8_TILE_BYTES];
  char* const Ks = smem;
  char* const Vs = smem + RING * F8_TILE_BYTES;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ql = lane & 31, hh = lane >> 5;
  int wg = blockIdx.x;
  {
    const int nwg = gridDim.x, qn = nwg >> 3, rn = nwg & 7, xcd = wg & 7;
    wg = ((xcd < rn) ? xcd * (qn + 1) : rn * (qn + 1) + (xcd - rn) * qn) + (wg >> 3);
  }
  const int bh = wg / p.nqt;
  const int qt = wg - bh * p.nqt;
  const int b = bh / p.heads, head = bh - b * p.heads;

  int qrow[QB];
  bool q_valid[QB];
  i32x8 qf[QB];
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
    qrow[qb] = qt * (128 * QB) + wave * (32 * QB) + qb * 32 + ql;
    q_valid[qb] = qrow[qb] < p.Nq;
    if (!q_valid[qb]) qrow[qb] = p.Nq - 1;
    const unsigned char* qp = p.q8 + ((int64_t)bh * p.Nq_pad + qrow[qb]) * 64 + 32 * hh;
    const u32x4 lo = ld_global16(qp), hi = ld_global16(qp + 16);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      qf[qb][e] = (int)lo[e];
      qf[qb][4 + e] = (int)hi[e];
    }
  }

  const int Nk = p.Nk;
  const int nkt = (Nk + KV_TILE - 1) / KV_TILE;
  const char* const kbase = reinterpret_cast<const char*>(p.k8) + (int64_t)bh * p.Nk_pad * 64;
  const char* const vbase = reinterpret_cast<const char*>(p.v8t) + (int64_t)bh * 64 * p.Nk_pad;
  // DMA: lane t of wave w fills physical 16-byte chunk (t & 3) of tile row 16 w + (t >> 2)
  const int lrow = 16 * wave + (lane >> 2);
  const int lchunk = (lane & 3) ^ ((lrow >> 2) & 3);
  const uint32_t koff = (uint32_t)(lrow * 64 + lchunk * 16);                      // + kt * 4096
  const uint32_t voff = (uint32_t)lrow * (uint32_t)p.Nk_pad + (uint32_t)lchunk * 16;  // + kt * 64
  auto load_kv = [&](int kt, int buf) {
    const int dst = buf * F8_TILE_BYTES + wave * 1024;
    __builtin_amdgcn_global_load_lds((glb_void*)(kbase + ((uint32_t)kt * 4096u + koff)), (lds_void*)(Ks + dst), 16, 0, 0);
    __builtin_amdgcn_global_load_lds((glb_void*)(vbase + ((uint32_t)kt * 64u + voff)), (lds_void*)(Vs + dst), 16, 0, 0);
  };

  f32x16 oacc[QB][2], nm[QB];
  float m_run[QB], l_run[QB];
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
    m_run[qb] = 0.f;
    l_run[qb] = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) nm[qb][r] = 0.f;
    asm volatile("" : "+v"(nm[qb]));
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
      for (int r = 0; r < 16; ++r) oacc[qb][d][r] = 0.f;
  }

  // a lane's 32 operand bytes of tile row `row`: logical chunks 2 hh, 2 hh + 1
  auto frag = [&](const char* tile, int row) -> i32x8 {
    const int sw = (row >> 2) & 3;
    const u32x4 lo = *reinterpret_cast<const u32x4*>(tile + row * 64 + (((2 * hh) ^ sw) << 4));
    const u32x4 hi = *reinterpret_cast<const u32x4*>(tile + row * 64 + (((2 * hh + 1) ^ sw) << 4));
    i32x8 f;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      f[e] = (int)lo[e];
      f[4 + e] = (int)hi[e];
    }
    return f;
  };

  auto raise = [&](int qb, f32x16 (&sacc)[2], float mx, bool first) {
    float rmx = fmaxf(mx, other_half(mx));
    if (!first) {
      rmx = fmaxf(rmx, 0.f);
      const float alpha = __builtin_amdgcn_exp2f(-rmx);
      l_run[qb] *= alpha;
#pragma unroll
      for (int d = 0; d < 2; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[qb][d][r] *= alpha;
    }
    m_run[qb] += rmx;
#pragma unroll
    for (int r = 0; r < 16; ++r) nm[qb][r] = -m_run[qb];
    asm volatile("" : "+v"(nm[qb]));
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) sacc[kb][r] -= rmx;
  };

  auto tile = [&](int kt, auto bufc, auto careful_c) {
    constexpr bool CAREFUL = decltype(careful_c)::value;
    const int buf = bufc;
    const char* ks = Ks + buf * F8_TILE_BYTES;
    const char* vs = Vs + buf * F8_TILE_BYTES;
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
      f32x16 sacc[2];
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) sacc[kb] = mfma_f8(frag(ks, kb * 32 + ql), qf[qb], nm[qb]);
      if constexpr (CAREFUL) {
        if (kt * KV_TILE + KV_TILE > Nk) {
#pragma unroll
          for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int key = kt * KV_TILE + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
              if (key >= Nk) sacc[kb][r] = -INFINITY;
            }
        }
      }
      float m0 = fmaxf(sacc[0][0], sacc[1][0]);
#pragma unroll
      for (int r = 1; r < 16; ++r) m0 = fmaxf(fmaxf(m0, sacc[0][r]), sacc[1][r]);
      if constexpr (CAREFUL) {
        raise(qb, sacc, m0, kt == 0);
      } else {
        if (__builtin_amdgcn_ballot_w64(m0 > F8_STALE_THR) != 0) raise(qb, sacc, m0, false);
      }
      // p = 2^S', row sum (f32, unrounded), P^T operand in e4m3: dword g = keys 8 g' + 4 hh + (0..3)
      float ps[4] = {0.f, 0.f, 0.f, 0.f};
      i32x8 pf;
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          float e[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            e[j] = __builtin_amdgcn_exp2f(sacc[kb][4 * g + j]);
            ps[j] += e[j];
          }
          int w = __builtin_amdgcn_cvt_pk_fp8_f32(e[0], e[1], 0, false);
          w = __builtin_amdgcn_cvt_pk_fp8_f32(e[2], e[3], w, true);
          pf[kb * 4 + g] = w;
        }
      l_run[qb] += (ps[0] + ps[1]) + (ps[2] + ps[3]);
#pragma unroll
      for (int db = 0; db < 2; ++db) oacc[qb][db] = mfma_f8(frag(vs, db * 32 + ql), pf, oacc[qb][db]);
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  auto arrive = [&](int t) {
    __syncthreads();
    if (t + 1 < nkt) load_kv(t + 1, (t + 1) & 1);
  };
  load_kv(0, 0);
  const int last_full = (Nk % KV_TILE == 0) ? nkt : nkt - 1;
  arrive(0);
  tile(0, std::integral_constant<int, 0>{}, std::true_type{});
  int kt = 1;
  for (; kt + 1 < last_full; kt += 2) {
    arrive(kt);
    tile(kt, std::integral_constant<int, 1>{}, std::false_type{});
    arrive(kt + 1);
    tile(kt + 1, std::integral_constant<int, 0>{}, std::false_type{});
  }
  for (; kt < last_full; ++kt) {
    arrive(kt);
    tile(kt, kt & 1, std::false_type{});
  }
  if (nkt > 1 && last_full < nkt) {
    arrive(nkt - 1);
    tile(nkt - 1, (nkt - 1) & 1, std::true_type{});
  }

#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
    const float l_tot = l_run[qb] + other_half(l_run[qb]);
    const float inv = 1.0f / l_tot;
    if (q_valid[qb]) {
      T* op = reinterpret_cast<T*>(p.o) + (int64_t)b * p.o_bs + (int64_t)qrow[qb] * p.o_rs + head * 64;
#pragma unroll
      for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          Pack4<T> ov;
#pragma unroll
          for (int e = 0; e < 4; ++e) ov.e[e] = from_f32<T>(oacc[qb][db][4 * g + e] * inv);
          *reinterpret_cast<u32x2*>(op + db * 32 + 8 * g + 4 * hh) = ov.u;
        }
    }
  }
}

// q (scaled), k -> fp8 rows; v -> fp8, transposed per (batch, head) with the per-tile key permutation the P^T operand
// dictates.  grid (q tiles + k tiles, B * heads), block 256: thread t handles row t >> 2, 16 channels 16 (t & 3).
struct AttnFp8PackParams {
  const void *q, *k, *v;
  int64_t q_bs, q_rs, k_bs, k_rs;
  unsigned char *q8, *k8, *v8t;
  int Nq, Nk, Nq_pad, Nk_pad, heads, ntq;
  float qscale;
};

__device__ __forceinline__ float clamp_e4m3(float x) { return fminf(fmaxf(x, -448.f), 448.f); }

template <typename T> __global__ __launch_bounds__(256) void attn_fp8_pack_kernel(const AttnFp8PackParams p) {
  __shared__ unsigned char vt[64 * 64];
  const int t = threadIdx.x, row = t >> 2, c16 = (t & 3) * 16;
  const int bh = blockIdx.y, b = bh / p.heads, head = bh - b * p.heads;
  const bool is_q = (int)blockIdx.x < p.ntq;
  const int tile = is_q ? blockIdx.x : blockIdx.x - p.ntq;
  const int n = tile * 64 + row;
  const int N = is_q ? p.Nq : p.Nk;
  auto load16 = [&](const void* base, int64_t bs, int64_t rs, float scale, float (&x)[16]) {
    if (n < N) {
      const T* src = reinterpret_cast<const T*>(base) + (int64_t)b * bs + (int64_t)n * rs + head * 64 + c16;
      Pack8<T> a, c;
      a.u = ld_global16(src);
      c.u = ld_global16(src + 8);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        x[e] = clamp_e4m3(to_f32(a.e[e]) * scale);
        x[8 + e] = clamp_e4m3(to_f32(c.e[e]) * scale);
      }
    } else {
#pragma unroll
      for (int e = 0; e < 16; ++e) x[e] = 0.f;
    }
  };
  auto pack16 = [&](const float (&x)[16]) -> u32x4 {
    u32x4 w;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      int v = __builtin_amdgcn_cvt_pk_fp8_f32(x[4 * g], x[4 * g + 1], 0, false);
      v = __builtin_amdgcn_cvt_pk_fp8_f32(x[4 * g + 2], x[4 * g + 3], v, true);
      w[g] = (unsigned)v;
    }
    return w;
  };
  float x[16];
  if (is_q) {
    load16(p.q, p.q_bs, p.q_rs, p.qscale, x);
    st_global16(p.q8 + ((int64_t)bh * p.Nq_pad + n) * 64 + c16, pack16(x));
    return;
  }
  load16(p.k, p.k_bs, p.k_rs, 1.f, x);
  st_global16(p.k8 + ((int64_t)bh * p.Nk_pad + n) * 64 + c16, pack16(x));
  load16(p.v, p.k_bs, p.k_rs, 1.f, x);
  const u32x4 vw = pack16(x);
  // transpose through LDS: byte of (d, key = row) goes to vt[d][pos(key)], pos = 32 h + 4 g + e for key = 8 g + 4 h + e
  const int pos = 32 * ((row >> 2) & 1) + 4 * (row >> 3) + (row & 3);
#pragma unroll
  for (int e = 0; e < 16; ++e) vt[(c16 + e) * 64 + pos] = (unsigned char)(vw[e >> 2] >> (8 * (e & 3)));
  __syncthreads();
  const u32x4 o = *reinterpret_cast<const u32x4*>(vt + row * 64 + c16);  // row = d, 16 permuted keys
  st_global16(p.v8t + ((int64_t)bh * 64 + row) * p.Nk_pad + tile * 64 + c16, o);
}

// ------------------------------------------------------------------------------------------------
// Temporal self-attention: at every (pixel, head) a (Fq x Fk <= 16 x 16) attention over the frame
// axis, head dim 64.  HBM-bound (0.1 % of the FLOPs) - the job is to touch q, k, v, o once with wide
// loads and keep the arithmetic off the VALU:
//   one wave per (pixel, head) problem at a time, several problems per wave;
//   S^T = K . Q^T   2 x v_mfma_f32_16x16x32: both operands are 16-byte row pieces loaded straight from
//                   global memory in fragment order (lane = (frame, d-chunk)), no LDS;
//   softmax over the 16 keys: 4 scores per lane + two cross-lane exchanges (lane ^ 16, lane ^ 32);
//   O^T = V^T . P^T 4 x v_mfma_f32_16x16x16 (K = 16 keys, no padding): the S^T accumulator IS the P^T
//                   operand (same lane map), V^T fragments come from a wave-private 2 KiB LDS image of
//                   the [16 frames][64] V tile (filled by DMA) through ds_read_b64_tr_b16.
struct TAttnParams {
  const void* q;
  const void* k;
  const void* v;
  void* o;
  int64_t ldq, ldk, ldo;
  int Fq, Fk, P, heads;
  float scale_log2e;
  int nprob;  // P * heads
};

__device__ __forceinline__ f32x4 mfma16k16(f16x4 a, f16x4 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x16f16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 mfma16k16(bf16x4 a, bf16x4 b, f32x4 c) {
  union { bf16x4 v; s16x4 s; } ua, ub;
  ua.v = a;
  ub.v = b;
  return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ua.s, ub.s, c, 0, 0, 0);
}

template <typename T>
__global__ __launch_bounds__(256) void tattn_kernel(const TAttnParams p) {
  __shared__ __attribute__((aligned(16))) char smem[4 * 2048];  // one V image per wave
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  char* const vimg = smem + wave * 2048;
  const int fr = lane & 15, g = lane >> 4;  // fragment row (frame / d) and k-group
  const int64_t qfs = (int64_t)p.P * p.ldq, kfs = (int64_t)p.P * p.ldk, ofs = (int64_t)p.P * p.ldo;
  const T* Qg = reinterpret_cast<const T*>(p.q);
  const T* Kg = reinterpret_cast<const T*>(p.k);
  const T* Vg = reinterpret_cast<const T*>(p.v);
  T* Og = reinterpret_cast<T*>(p.o);
  const int fq_row = fr < p.Fq ? fr : p.Fq - 1;  // clamp: rows >= Fq are computed and dropped
  const int fk_row = fr < p.Fk ? fr : p.Fk - 1;
  // DMA of the V tile: instruction j fills keys 8j..8j+7 (lane i -> key 8j + i/8, physical 16-byte
  // slot i%8); the slot is XOR-ed by ((key>>1)&3)<<1 on the SOURCE side (conflict-free tr reads)
  const int dkey = lane >> 3, dslot = lane & 7;
  // transposed-read addresses: lane i of a 16-lane group supplies row (key 4g + i/4), columns 4(i&3)..
  const int tkey = 4 * g + (fr >> 2);
  const int tsw = ((tkey >> 1) & 3) << 1;

  const int waves_total = gridDim.x * 4;
  for (int prob = blockIdx.x * 4 + wave; prob < p.nprob; prob += waves_total) {
    const int pix = prob / p.heads, head = prob - pix * p.heads;
    const T* qp = Qg + (int64_t)pix * p.ldq + head * 64;
    const T* kp = Kg + (int64_t)pix * p.ldk + head * 64;
    const T* vp = Vg + (int64_t)pix * p.ldk + head * 64;
    // V tile -> LDS (2 x 1 KiB)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      int key = 8 * j + dkey;
      const int lslot = dslot ^ (((key >> 1) & 3) << 1);
      if (key > p.Fk - 1) key = p.Fk - 1;
      __builtin_amdgcn_global_load_lds((glb_void*)(vp + key * kfs + lslot * 8), (lds_void*)(vimg + j * 1024), 16, 0, 0);
    }
    // S^T = K . Q^T
    f32x4 st = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      Pack8<T> kf, qf;
      kf.u = ld_global16(kp + fk_row * kfs + 32 * ks + 8 * g);
      qf.u = ld_global16(qp + fq_row * qfs + 32 * ks + 8 * g);
      st = mfma16(kf.v, qf.v, st);
    }
    // st[r] = score(key = 4g + r, query = fr); softmax over keys in base 2
    float sc[4], mx = -INFINITY;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      sc[r] = (4 * g + r < p.Fk) ? st[r] * p.scale_log2e : -INFINITY;
      mx = fmaxf(mx, sc[r]);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float l = 0.f;
    Pack4<T> pf;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float e = __builtin_amdgcn_exp2f(sc[r] - mx);
      l += e;
      pf.e[r] = from_f32<T>(e);
    }
    l += __shfl_xor(l, 16, 64);
    l += __shfl_xor(l, 32, 64);
    const float inv = __builtin_amdgcn_rcpf(l);
    // O^T = V^T . P^T
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the V image has landed (wave-private: no barrier)
    T* op = Og + (int64_t)pix * p.ldo + head * 64 + fr * ofs;
#pragma unroll
    for (int db = 0; db < 4; ++db) {
      const int chunk = (2 * db + ((fr & 3) >> 1)) ^ tsw;
      union { s16x4 s; typename Vec<T>::v4 v; } vf;
      vf.s = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(vimg + tkey * 128 + chunk * 16 + (fr & 1) * 8));
      f32x4 ot = {0.f, 0.f, 0.f, 0.f};
      ot = mfma16k16(vf.v, pf.v, ot);
      // ot[r] = O[query fr][d = 16 db + 4g + r]
      Pack4<T> ov;
#pragma unroll
      for (int r = 0; r < 4; ++r) ov.e[r] = from_f32<T>(ot[r] * inv);
      if (fr < p.Fq) *reinterpret_cast<u32x2*>(op + 16 * db + 4 * g) = ov.u;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Small generic attention: any head dim D <= 128 (D % 8 == 0), Nk <= 1024, f32 arithmetic on the vector ALU.
// For the OpenCLIP ViT-H/14 image tower of the conditioning tail (condition.py:300-382: 257 tokens, 16 heads of
// 80 channels, 32 layers, once per generate call - 5 GFLOP in all): the MFMA kernels above are specialised for head
// dim 64.  One wave per query row: lanes own keys (scores: q from registers, K rows from L2), wave-wide max / sum,
// probabilities through a 4 KiB LDS row, then lanes own output channels (V rows read coalesced).
struct AttnSmallParams {
  const void *q, *k, *v;
  void* o;
  int64_t q_bs, q_rs, k_bs, k_rs, o_bs, o_rs;
  int Nq, Nk, heads, D;
  float scale;
};

template <typename T> __global__ __launch_bounds__(256) void attn_small_kernel(const AttnSmallParams p) {
  __shared__ float prob[4][1024];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int bh = blockIdx.y, b = bh / p.heads, head = bh - b * p.heads;
  const int qi = blockIdx.x * 4 + wave;
  if (qi >= p.Nq) return;  // (wave-uniform; no block-wide barrier below)
  const int D = p.D, nv = D >> 3;
  const T* qp = reinterpret_cast<const T*>(p.q) + (int64_t)b * p.q_bs + (int64_t)qi * p.q_rs + head * D;
  const T* kb = reinterpret_cast<const T*>(p.k) + (int64_t)b * p.k_bs + head * D;
  const T* vb = reinterpret_cast<const T*>(p.v) + (int64_t)b * p.k_bs + head * D;
  float qv[128];
#pragma unroll
  for (int c = 0; c < 16; ++c) {
    if (c < nv) {
      Pack8<T> t;
      t.u = ld_global16(qp + 8 * c);
#pragma unroll
      for (int e = 0; e < 8; ++e) qv[8 * c + e] = to_f32(t.e[e]) * p.scale;
    }
  }
  float sc[16];
  float mx = -INFINITY;
#pragma unroll
  for (int t = 0; t < 16; ++t) {
    const int j = lane + 64 * t;
    sc[t] = -INFINITY;
    if (j < p.Nk) {
      const T* kr = kb + (int64_t)j * p.k_rs;
      float a = 0.f;
#pragma unroll
      for (int c = 0; c < 16; ++c) {
        if (c < nv) {
          Pack8<T> kk;
          kk.u = ld_global16(kr + 8 * c);
#pragma unroll
          for (int e = 0; e < 8; ++e) a = fmaf(qv[8 * c + e], to_f32(kk.e[e]), a);
        }
      }
      sc[t] = a;
      mx = fmaxf(mx, a);
    }
  }
  mx = wave_max(mx);
  float l = 0.f;
#pragma unroll
  for (int t = 0; t < 16; ++t) {
    const int j = lane + 64 * t;
    if (j < p.Nk) {
      const float e = __expf(sc[t] - mx);
      l += e;
      prob[wave][j] = e;
    }
  }
  l = wave_sum(l);
  __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): this wave's LDS writes are done (wave-private row: no barrier)
  const float inv = 1.0f / l;
  // output channels d = lane and lane + 64
  float o0 = 0.f, o1 = 0.f;
  const bool h0 = lane < D, h1 = lane + 64 < D;
  for (int j = 0; j < p.Nk; ++j) {
    const float pj = prob[wave][j];
    const T* vr = vb + (int64_t)j * p.k_rs;
    if (h0) o0 = fmaf(pj, to_f32(vr[lane]), o0);
    if (h1) o1 = fmaf(pj, to_f32(vr[lane + 64]), o1);
  }
  T* op = reinterpret_cast<T*>(p.o) + (int64_t)b * p.o_bs + (int64_t)qi * p.o_rs + head * D;
  if (h0) op[lane] = from_f32<T>(o0 * inv);
  if (h1) op[lane + 64] = from_f32<T>(o1 * inv);
}

}  // namespace pm

using namespace pm;

// Kernel selection override for A/B measurements (tools/attn_bench.py) - ONLY in the -DPM_DIAG build
// (libpandora_mi355x_diag.so, include/pandora_mi355x_diag.h): the shipped library always runs the production kernels,
// holds no mutable global and instantiates none of the older variants / ceiling probes.
#ifdef PM_DIAG
static int g_attn_variant = -1;
static int attn_variant() {
  if (g_attn_variant < 0) {
    const char* e = diag_env("PANDORA_ATTN_VARIANT");
    g_attn_variant = e ? atoi(e) : 0;
  }
  return g_attn_variant;
}
extern "C" void pm_debug_attn_variant(int v) { g_attn_variant = v < 0 ? 0 : v; }
static unsigned long long* g_attn_stamps = nullptr;
extern "C" void pm_debug_attn_stamps(void* buf) { g_attn_stamps = reinterpret_cast<unsigned long long*>(buf); }
#else
static constexpr int attn_variant() { return 0; }
#endif

extern "C" int pm_attention(const void* q, int64_t q_bs, int64_t q_rs, const void* k1,
                            const void* v1, int64_t k1_bs, int64_t k1_rs, int64_t Nk1,
                            const void* k2, const void* v2, int64_t k2_bs, int64_t k2_rs,
                            int64_t Nk2, float w2, void* o, int64_t o_bs, int64_t o_rs, int64_t B,
                            int64_t heads, int64_t Nq, float scale, int dtype, void* stream) {
  if (!q || !k1 || !v1 || !o) return PM_E_NULL;
  if (k2 && !v2) return PM_E_NULL;
  if (B < 1 || heads < 1 || Nq < 1 || Nk1 < 1 || (k2 && Nk2 < 1)) return PM_E_SHAPE;
  if ((q_bs | q_rs | k1_bs | k1_rs | o_bs | o_rs) & 7) return PM_E_SHAPE;
  if (k2 && ((k2_bs | k2_rs) & 7)) return PM_E_SHAPE;
  if (Nk1 * k1_rs * 2 >= (1ll << 32) || (k2 && Nk2 * k2_rs * 2 >= (1ll << 32))) return PM_E_SHAPE;  // 32-bit lane byte offsets
  if (B * heads * ((Nq + 127) / 128) > (1ll << 30)) return PM_E_SHAPE;
  AttnParams p{};
  p.q = q; p.o = o; p.q_bs = q_bs; p.q_rs = q_rs; p.o_bs = o_bs; p.o_rs = o_rs;
  p.k[0] = k1; p.v[0] = v1; p.k_bs[0] = k1_bs; p.k_rs[0] = k1_rs; p.Nk[0] = (int)Nk1; p.w[0] = 1.f;
  p.k[1] = k2; p.v[1] = v2; p.k_bs[1] = k2_bs; p.k_rs[1] = k2_rs; p.Nk[1] = (int)Nk2; p.w[1] = w2;
  p.nseg = k2 ? 2 : 1;
  p.Nq = (int)Nq; p.heads = (int)heads;
  p.scale_log2e = scale * 1.4426950408889634f;
  p.prescaled = fabsf(p.scale_log2e - 1.0f) < 1e-6f ? 1 : 0;
  if (k2) {
    p.nqt = (int)((Nq + 127) / 128);
    dim3 grid((unsigned)(p.nqt * B * heads));
    PM_DISPATCH_DTYPE(dtype, T,
                      hipLaunchKernelGGL((attn_kernel<T, 1, true>), grid, dim3(256), 0, (hipStream_t)stream, p);
                      return check_launch());
  }
  // single segment: 32 query rows per wave (128 per workgroup), double-buffered K/V, three workgroups per CU.
  // Measured on the U-Net's shapes (tools/attn_bench.py, interleaved A/B in one process, random data): this is
  // the fastest at every N (1020-1035 TF/s at N = 9216).  The 64-rows-per-wave forms (variants 3 / 5: half the
  // LDS fragment reads per MFMA, a 4-stage K/V ring, 2.0 GHz instead of 1.75) lose 1-3 % there and more on
  // short sequences; kept for A/B runs.
  const int variant = attn_variant();
#ifdef PM_DIAG
  p.stamps = g_attn_stamps;
#endif
  const bool qb2 = variant == 3 || variant == 5;
  const int rows = qb2 ? 256 : 128;
  p.nqt = (int)((Nq + rows - 1) / rows);
  dim3 grid((unsigned)(p.nqt * B * heads));
#define PM_ATTN_LAUNCH(QB_, PIPE_, POST_, RING_)                                                                   \
  PM_DISPATCH_DTYPE(dtype, T,                                                                                      \
                    hipLaunchKernelGGL((attn_self_kernel<T, QB_, PIPE_, POST_, RING_>), grid, dim3(256), 0,         \
                                       (hipStream_t)stream, p);                                                    \
                    return check_launch())
#ifdef PM_DIAG
  if (variant == 16) return launch_attn_self16(p, dtype, grid, (hipStream_t)stream);  // force the 16x16x32-MFMA form (attn16.hip)
  if (variant == 17) return launch_attn_self16(p, dtype, grid, (hipStream_t)stream, 1);  // ... with per-lane K/V addresses
  if (variant == 18) return launch_attn_self16(p, dtype, grid, (hipStream_t)stream, 2);  // ... with the row sums on the vector pipe
  if (variant == 19) return launch_attn_self16(p, dtype, grid, (hipStream_t)stream, 3);  // ... with hipcc's own PV issue order
  if (variant == 9)  // (kept for A/B runs: the round-1 kernel)
    PM_DISPATCH_DTYPE(dtype, T,
                      hipLaunchKernelGGL((attn_kernel<T, 1, false>), grid, dim3(256), 0, (hipStream_t)stream, p);
                      return check_launch());
  if (variant == 3) PM_ATTN_LAUNCH(2, true, false, 4);
  if (variant == 5) PM_ATTN_LAUNCH(2, true, true, 4);
  if (variant >= 11 && variant <= 13) {  // ceiling probes (diagnosis: the output is not an attention result)
#define PM_ATTN_PROBE(PROBE_)                                                                                      \
  PM_DISPATCH_DTYPE(dtype, T,                                                                                      \
                    hipLaunchKernelGGL((attn_self_kernel<T, 1, false, false, 2, PROBE_>), grid, dim3(256), 0,       \
                                       (hipStream_t)stream, p);                                                    \
                    return check_launch())
    if (variant == 11) PM_ATTN_PROBE(1);
    if (variant == 12) PM_ATTN_PROBE(2);
    PM_ATTN_PROBE(3);
#undef PM_ATTN_PROBE
  }
  if (variant == 1) PM_ATTN_LAUNCH(1, false, false, 2);  // force the 32x32x16 form
#endif
  // long sequences: the 16x16x32-MFMA form (csrc/attn16.hip: +1-2 % by wall at N = 9216, +-1 % around 2304-2560 - the chip holds a ~11 % higher clock
  // under that shape, profiles/r04/attention_shapes.txt); shorter ones: the 32x32x16 form (fewer, longer MFMAs per tile)
  if (Nq >= 4096 && Nk1 >= 4096) return launch_attn_self16(p, dtype, grid, (hipStream_t)stream);
  PM_ATTN_LAUNCH(1, false, false, 2);
#undef PM_ATTN_LAUNCH
}

extern "C" size_t pm_attention_fp8_workspace_bytes(int64_t B, int64_t heads, int64_t Nq, int64_t Nk) {
  const int64_t nq = (Nq + 63) / 64 * 64, nk = (Nk + 63) / 64 * 64;
  return (size_t)(B * heads * 64 * (nq + 2 * nk));
}

extern "C" int pm_attention_fp8(const void* q, int64_t q_bs, int64_t q_rs, const void* k, const void* v,
                                int64_t k_bs, int64_t k_rs, int64_t Nk, void* o, int64_t o_bs, int64_t o_rs,
                                int64_t B, int64_t heads, int64_t Nq, float scale, int dtype, void* workspace,
                                size_t workspace_bytes, void* stream) {
  if (!q || !k || !v || !o || !workspace) return PM_E_NULL;
  if (B < 1 || heads < 1 || Nq < 1 || Nk < 1) return PM_E_SHAPE;
  if ((q_bs | q_rs | k_bs | k_rs | o_bs | o_rs) & 7) return PM_E_SHAPE;
  if (workspace_bytes < pm_attention_fp8_workspace_bytes(B, heads, Nq, Nk)) return PM_E_WORKSPACE;
  const int64_t nq = (Nq + 63) / 64 * 64, nk = (Nk + 63) / 64 * 64;
  if (nk * 64 >= (1ll << 31) || B * heads > 65535) return PM_E_SHAPE;
  unsigned char* ws = reinterpret_cast<unsigned char*>(workspace);
  AttnFp8PackParams pp{};
  pp.q = q; pp.k = k; pp.v = v; pp.q_bs = q_bs; pp.q_rs = q_rs; pp.k_bs = k_bs; pp.k_rs = k_rs;
  pp.q8 = ws; pp.k8 = ws + B * heads * nq * 64; pp.v8t = pp.k8 + B * heads * nk * 64;
  pp.Nq = (int)Nq; pp.Nk = (int)Nk; pp.Nq_pad = (int)nq; pp.Nk_pad = (int)nk; pp.heads = (int)heads;
  pp.ntq = (int)(nq / 64);
  pp.qscale = scale * 1.4426950408889634f;  // scores in the base-2 domain
  dim3 pgrid((unsigned)(nq / 64 + nk / 64), (unsigned)(B * heads));
  AttnFp8Params p{};
  p.q8 = pp.q8; p.k8 = pp.k8; p.v8t = pp.v8t; p.o = o; p.o_bs = o_bs; p.o_rs = o_rs;
  p.Nq = (int)Nq; p.Nk = (int)Nk; p.Nq_pad = (int)nq; p.Nk_pad = (int)nk; p.heads = (int)heads;
#ifdef PM_DIAG
  const bool qb2 = attn_variant() == 2;  // (32 rows per wave measured faster at every N: tools/attn_bench.py 101 / 102)
#else
  constexpr bool qb2 = false;
#endif
  const int rows = qb2 ? 256 : 128;
  p.nqt = (int)((Nq + rows - 1) / rows);
  dim3 grid((unsigned)(p.nqt * B * heads));
  PM_DISPATCH_DTYPE(dtype, T,
                    hipLaunchKernelGGL((attn_fp8_pack_kernel<T>), pgrid, dim3(256), 0, (hipStream_t)stream, pp);
                    if constexpr (PM_DIAG_BUILD) {
                      if (qb2) {
                        hipLaunchKernelGGL((attn_fp8_kernel<T, PM_DIAG_BUILD ? 2 : 1>), grid, dim3(256), 0, (hipStream_t)stream, p);
                        return check_launch();
                      }
                    }
                    hipLaunchKernelGGL((attn_fp8_kernel<T, 1>), grid, dim3(256), 0, (hipStream_t)stream, p);
                    return check_launch());
}

extern "C" int pm_attention_generic(const void* q, int64_t q_bs, int64_t q_rs, const void* k, const void* v,
                                    int64_t k_bs, int64_t k_rs, int64_t Nk, void* o, int64_t o_bs, int64_t o_rs,
                                    int64_t B, int64_t heads, int64_t Nq, int64_t D, float scale, int dtype,
                                    void* stream) {
  if (!q || !k || !v || !o) return PM_E_NULL;
  if (B < 1 || heads < 1 || Nq < 1 || Nk < 1 || Nk > 1024 || D < 8 || D > 128 || (D & 7)) return PM_E_SHAPE;
  if ((q_bs | q_rs | k_bs | k_rs | o_bs | o_rs) & 7) return PM_E_SHAPE;
  if (B * heads > 65535) return PM_E_SHAPE;
  AttnSmallParams p{};
  p.q = q; p.k = k; p.v = v; p.o = o; p.q_bs = q_bs; p.q_rs = q_rs; p.k_bs = k_bs; p.k_rs = k_rs;
  p.o_bs = o_bs; p.o_rs = o_rs; p.Nq = (int)Nq; p.Nk = (int)Nk; p.heads = (int)heads; p.D = (int)D; p.scale = scale;
  dim3 grid((unsigned)((Nq + 3) / 4), (unsigned)(B * heads));
  PM_DISPATCH_DTYPE(dtype, T,
                    hipLaunchKernelGGL((attn_small_kernel<T>), grid, dim3(256), 0, (hipStream_t)stream, p);
                    return check_launch());
}

extern "C" int pm_attention_temporal(const void* q, int64_t ldq, const void* k, const void* v,
                                     int64_t ldk, void* o, int64_t ldo, int64_t Fq, int64_t Fk,
                                     int64_t P, int64_t heads, float scale, int dtype,
                                     void* stream) {
  if (!q || !k || !v || !o) return PM_E_NULL;
  if (Fq < 1 || Fk < 1 || Fk > 16 || P < 1 || heads < 1) return PM_E_SHAPE;
  if ((ldq | ldk | ldo) & 7) return PM_E_SHAPE;
  TAttnParams p{};
  p.q = q; p.k = k; p.v = v; p.o = o; p.ldq = ldq; p.ldk = ldk; p.ldo = ldo;
  p.Fq = (int)Fq; p.Fk = (int)Fk; p.P = (int)P; p.heads = (int)heads;
  p.scale_log2e = scale * 1.4426950408889634f;
  if (Fq > 16 || P * heads > (1ll << 30)) return PM_E_SHAPE;
  p.nprob = (int)(P * heads);
  int64_t nb = (p.nprob + 3) / 4;  // 4 waves per block; cap the grid and loop (8 waves/SIMD resident)
  if (nb > 256 * 8) nb = 256 * 8;
  dim3 grid((unsigned)nb);
  PM_DISPATCH_DTYPE(dtype, T,
                    hipLaunchKernelGGL((tattn_kernel<T>), grid, dim3(256), 0, (hipStream_t)stream, p);
                    return check_launch());
}
