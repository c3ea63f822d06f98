// Spatial self-attention, head dim 64, on v_mfma_f32_16x16x32 (the other bf16 MFMA shape at the SAME
// per-wave output tile as attn_self_kernel<T, 1, ...> of attn.hip - 32 query rows x 64 keys per tile and wave - so that
// the two shapes can be ranked by wall time and by the clock the chip holds under each, MI355X_MICROARCH.md "DVFS
// give-back" item 7 / cdna_hip_programming.md rule 28).  Reference call site: CrossAttention.forward, attention.py:81-144.
// Measured (profiles/r04/attention_shapes.txt, interleaved, random data, in-kernel clock stamps): this shape holds
// 1.87-1.90 GHz where the 32x32x16 form holds 1.69 (+11 %) and spends ~9 % more cycles (an MFMA holds the SIMD's vector
// issue for 8 of its 16 cycles instead of 8 of 32, and the softmax's vector work is the same): +2 % by wall at N = 9216,
// +-1 % at 2304 - 2560, slower below ~600 tokens -> pm_attention runs it for sequences of >= 4096 tokens.
//
// Same algorithm as attn_self_kernel (flash-style, base-2 scores, the running maximum subtracted BY the MFMA - the S
// chain starts from a register block holding -m - and a STALE maximum raised only when a tile outgrows it by 2^6), other
// operand geometry:
//   S^T[kb][qb] = K[kb] . Q^T[qb] - m     16 keys x 16 queries per instruction, two chained over d = 0..31 | 32..63;
//                                         a lane (i = lane & 15, g = lane >> 4) owns query 16 qb + i and keys
//                                         16 kb + 4 g + r, r = 0..3: TWO query rows per lane (one per qb), 16 scores each;
//   O^T[db][qb] += V^T[db][kh] . P^T[kh][qb]   16 d x 16 queries, contraction over 32 keys: the B operand is made of the
//                                         S accumulators in place - element j of lane (i, g) is key 32 kh + 4 g + j for
//                                         j < 4 (block kb = 2 kh) and 32 kh + 16 + 4 g + (j - 4) for j >= 4 (block 2 kh + 1)
//                                         - and the V^T fragment uses the same key order: two ds_read_b64_tr_b16 blocks
//                                         16 keys apart (the MFMA pairs A and B elements by (lane group, element): any
//                                         contraction order both operands share is legal).
// LDS images: K rows of 128 B with the 16-byte chunk index XOR-ed with (row >> 1) & 7 (as attn.hip: the ds_read_b128 of
// the 16x16x32 A operand - row = lane & 15, chunk = 4 dh + g - is conflict-free on it); V rows of 128 B with the chunk
// index XOR-ed with ((row >> 1) & 3) << 1: the transposed 4-key x 16-d block reads of a 32-lane half (two blocks stacked
// in one aligned 8-row group) then touch every bank once (tools/diag/bank_sim.py).
#include "attn_common.hpp"

namespace pm {

constexpr float STALE_THR16 = 6.0f;

__device__ __forceinline__ float xor16(float v) { return __shfl_xor(v, 16, 64); }
// (This file is compiled with -fno-slp-vectorize, build.py PER_FILE_FLAGS: plain -O3 packs the row-sum chains of the two query
// blocks into v_pk_add_f32, which costs ~13 cycles each beside MFMAs instead of 4 - MI355X_MICROARCH.md, price of one filler.
// An inline-asm v_add_f32 is NOT the way to keep them single: hipcc inserts no wait state between a transcendental result
// (v_exp_f32) and an asm statement that reads it - gfx950's trans-forwarding hazard - and the sums come out wrong.)

// (120 VGPRs, 32 KiB of LDS: four workgroups = 4 waves per SIMD are resident per CU)
// DESC: K/V tiles that lie wholly inside the sequence are fetched with `buffer_load ... lds` - per-lane offset fixed for
// the whole kernel, the tile's position in the SCALAR offset, the LDS stage in M0 - so that the K loop issues no vector
// instruction for addresses (the global_load_lds form recomputes clamped 64-bit per-lane addresses: ~23 VALU incl. two
// v_mul_lo_u32 of the ~127 a tile costs).  A ragged last tile keeps the clamped form.  Measured at N = 9216 (A/B in one
// process, diagnostics variant 17 = per-lane addresses): -2.5 % cycles per workgroup at the same clock, +1.6 % by wall;
// the same change in attn_self_kernel (attn.hip): -5.5 % cycles, of which the clock gives most back (1.79 -> 1.69 GHz).
// MSUM: the softmax denominators come out of the matrix pipe - one more MFMA per (32-key half, query block) whose A operand
// is a constant all-ones fragment: every row of its 16 x 16 result block is sum_k P^T[k][q], accumulated in f32 over the
// SAME 16-bit P the numerator uses (a row's weights then sum to one exactly), already replicated over the lanes that need
// it.  4 MFMAs per tile replace 32 v_add_f32 and the two cross-lane sums at the end.  Measured (N = 9216, A/B in one process,
// diagnostics variant 18 = sums on the vector pipe): -0.6 % cycles, +2.0 % by wall (1088 -> 1111 TF/s); error against an
// f64 softmax attention unchanged (2.92e-3 bf16 / 3.63e-4 f16 either way, tools/diag/attn_msum_accuracy.py).  That a
// quarter of the loop's vector instructions buys 0.6 % of its cycles - and four more MFMAs cost nothing - says the loop
// is bound by neither pipe's issue rate but by the dependent chain of a tile (LDS read -> MFMA -> max -> exp -> MFMA)
// at four waves per SIMD.
// PVPIPE: the PV phase's issue order pinned with sched_group_barrier (see the end of tile()): -0.9 % cycles, +0.7...1.1 %
// by wall at N = 9216 / 4608 against hipcc's own order (diagnostics variant 19).  With the denominators on the matrix
// pipe the loop's issue costs sum to ~745 cycles per wave and tile (32 v_exp x 8 + 16 v_cvt_pk x 4.5 + 21 max x 4 +
// 11 address / compare x 4 + 36 MFMA issue holds x 8, MI355X_MICROARCH.md); the kernel runs 836: what is left to a perfect
// schedule at this instruction mix is ~11 %, i.e. ~0.49 of the dense peak at the clock the chip holds.
template <typename T, bool DESC, bool MSUM, bool PVPIPE>
__global__ __launch_bounds__(256, 4) void attn_self16_kernel(const AttnParams p) {
  __shared__ __attribute__((aligned(16))) char smem[4 * KV_TILE_BYTES];  // K[2], V[2]
  char* const Ks = smem;
  char* const Vs = smem + 2 * KV_TILE_BYTES;

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, g = lane >> 4;
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

  // Q^T fragments (B operand of the S chains): lane (i, g) holds q[16 qb + i][32 dh + 8 g .. + 7]
  int qrow[2];
  bool q_valid[2];
  Pack8<T> qf[2][2];
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
    qrow[qb] = qt * 128 + wave * 32 + qb * 16 + li;
    q_valid[qb] = qrow[qb] < p.Nq;
    if (!q_valid[qb]) qrow[qb] = p.Nq - 1;
    const T* qp = reinterpret_cast<const T*>(p.q) + (int64_t)b * p.q_bs + (int64_t)qrow[qb] * p.q_rs + head * 64 + 8 * g;
#pragma unroll
    for (int dh = 0; dh < 2; ++dh) qf[qb][dh].u = ld_global16(qp + 32 * dh);
  }
  if (p.prescaled == 0) {  // (uniform) general callers: one extra 16-bit rounding of q * scale * log2(e)
#pragma unroll
    for (int qb = 0; qb < 2; ++qb)
#pragma unroll
      for (int dh = 0; dh < 2; ++dh)
#pragma unroll
        for (int e = 0; e < 8; ++e) qf[qb][dh].e[e] = from_f32<T>(to_f32(qf[qb][dh].e[e]) * p.scale_log2e);
  }

  const int lr = tid >> 3;
  const int Nk = p.Nk[0];
  const char* const kbase = reinterpret_cast<const char*>(reinterpret_cast<const T*>(p.k[0]) + (int64_t)b * p.k_bs[0] + head * 64);
  const char* const vbase = reinterpret_cast<const char*>(reinterpret_cast<const T*>(p.v[0]) + (int64_t)b * p.k_bs[0] + head * 64);
  const uint32_t krs2 = (uint32_t)(p.k_rs[0] * 2);
  const int nkt = (Nk + KV_TILE - 1) / KV_TILE;
  const int kc = (tid & 7) ^ ((lr >> 1) & 7);          // source-side swizzles (the DMA writes LDS linearly)
  const int vc = (tid & 7) ^ (((lr >> 1) & 3) << 1);
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
  // (range = through the head slice of the last key row; a tile this form is used for never reaches it)
  const uint32_t kv_range = (uint32_t)(Nk - 1) * krs2 + 128u;
  const __amdgpu_buffer_rsrc_t krsrc = __builtin_amdgcn_make_buffer_rsrc((void*)kbase, (short)0, (int)kv_range, 0x00020000);
  const __amdgpu_buffer_rsrc_t vrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)vbase, (short)0, (int)kv_range, 0x00020000);
  const uint32_t kvoff = (uint32_t)lr * krs2 + (uint32_t)kc * 16, vvoff = (uint32_t)lr * krs2 + (uint32_t)vc * 16;
  auto load_kv_full = [&](int kt, int buf) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const uint32_t soff = (uint32_t)(kt * KV_TILE + 32 * j) * krs2;
      const int dst = buf * KV_TILE_BYTES + (32 * j + 8 * wave) * 128;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(krsrc, (lds_void*)(Ks + dst), 16, kvoff, soff, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(vrsrc, (lds_void*)(Vs + dst), 16, vvoff, soff, 0, 0);
    }
  };

  f32x4 oacc[4][2], nm[2];  // nm: every register = -m_run, the initial accumulator of the S^T chains
  float m_run[2], l_run[2];
  f32x4 lacc[2];  // MSUM: every register = the running denominator of the lane's query row in block qb
  Pack8<T> ones;
#pragma unroll
  for (int e = 0; e < 8; ++e) ones.e[e] = from_f32<T>(1.0f);
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
    m_run[qb] = 0.f;
    l_run[qb] = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) lacc[qb][r] = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) nm[qb][r] = 0.f;
    asm volatile("" : "+v"(nm[qb]));
#pragma unroll
    for (int db = 0; db < 4; ++db)
#pragma unroll
      for (int r = 0; r < 4; ++r) oacc[db][qb][r] = 0.f;
  }

  // lane coordinates of the reads
  const int krow_off = li * 128;                       // K row 16 kb + i
  const int kx = (li >> 1) & 7;                        // its chunk XOR (16 kb does not change (row >> 1) & 7)
  const int tq = li >> 2, tp = li & 3;                 // transposed read: lane 4 q + p supplies row q, columns 4 p .. 4 p + 3
  const int vrow0 = 4 * g + tq;                        // key row inside a 16-key block
  const int vx = ((vrow0 >> 1) & 3) << 1;              // V chunk XOR (+16 / +32 / +48 rows leave (row >> 1) & 3 alone)
  const int vsub = 8 * (tp & 1), vch = tp >> 1;

  // raise the stale maximum of block qb by the row maximum of the pending tile (mx: this lane's 16 scores of the row)
  auto raise = [&](int qb, f32x4 (&sacc)[4][2], float mx, bool first) {
    float rmx = fmaxf(mx, xor16(mx));
    rmx = fmaxf(rmx, other_half(rmx));
    if (!first) {
      rmx = fmaxf(rmx, 0.f);  // never lower m: rows that did not outgrow it keep alpha == 1 exactly
      const float alpha = __builtin_amdgcn_exp2f(-rmx);
      l_run[qb] *= alpha;
#pragma unroll
      for (int r = 0; r < 4; ++r) lacc[qb][r] *= alpha;
#pragma unroll
      for (int db = 0; db < 4; ++db)
#pragma unroll
        for (int r = 0; r < 4; ++r) oacc[db][qb][r] *= alpha;
    }
    m_run[qb] += rmx;
#pragma unroll
    for (int r = 0; r < 4; ++r) nm[qb][r] = -m_run[qb];
    asm volatile("" : "+v"(nm[qb]));  // (opaque: else hipcc re-broadcasts the block from one scalar in every tile)
#pragma unroll
    for (int kb = 0; kb < 4; ++kb)
#pragma unroll
      for (int r = 0; r < 4; ++r) sacc[kb][qb][r] -= rmx;
  };

  // one K/V tile.  CAREFUL: exact maximum + key masking (first tile, ragged last tile).
  auto tile = [&](int kt, auto bufc, auto careful_c) {
    constexpr bool CAREFUL = decltype(careful_c)::value;
    const int buf = bufc;
    const char* ks = Ks + buf * KV_TILE_BYTES;
    const char* vs = Vs + buf * KV_TILE_BYTES;
    // ---- S'^T = K . Q'^T - m ----
    f32x4 sacc[4][2];
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
      Pack8<T> kf[2];
#pragma unroll
      for (int dh = 0; dh < 2; ++dh)
        kf[dh].u = *reinterpret_cast<const u32x4*>(ks + kb * 2048 + krow_off + (((4 * dh + g) ^ kx) << 4));
#pragma unroll
      for (int qb = 0; qb < 2; ++qb) {
        sacc[kb][qb] = mfma16(kf[0].v, qf[qb][0].v, nm[qb]);
        sacc[kb][qb] = mfma16(kf[1].v, qf[qb][1].v, sacc[kb][qb]);
      }
    }
    if constexpr (CAREFUL) {
      if (kt * KV_TILE + KV_TILE > Nk) {
#pragma unroll
        for (int kb = 0; kb < 4; ++kb)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int key = kt * KV_TILE + 16 * kb + 4 * g + r;
            if (key >= Nk) {
              sacc[kb][0][r] = -INFINITY;
              sacc[kb][1][r] = -INFINITY;
            }
          }
      }
    }
    auto lane_max = [&](int qb) -> float {
      float m0 = fmaxf(sacc[0][qb][0], sacc[1][qb][0]);
#pragma unroll
      for (int kb = 0; kb < 4; kb += 2)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (kb || r) m0 = fmaxf(fmaxf(m0, sacc[kb][qb][r]), sacc[kb + 1][qb][r]);
      return m0;
    };
    if constexpr (CAREFUL) {
      raise(0, sacc, lane_max(0), kt == 0);
      raise(1, sacc, lane_max(1), kt == 0);
    } else {
      const float m0 = lane_max(0), m1 = lane_max(1);
      if (__builtin_amdgcn_ballot_w64(fmaxf(m0, m1) > STALE_THR16) != 0) {
        raise(0, sacc, m0, false);
        raise(1, sacc, m1, false);
      }
    }
    // ---- p = 2^S', 16-bit P^T operand (the accumulators in place), lane sums ----
    Pack8<T> pf[2][2];  // [kh][qb]
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
      float ps[4];
#pragma unroll
      for (int kb = 0; kb < 4; ++kb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float pe = __builtin_amdgcn_exp2f(sacc[kb][qb][r]);
          if constexpr (!MSUM) ps[r] = kb ? ps[r] + pe : pe;
          pf[kb >> 1][qb].e[4 * (kb & 1) + r] = from_f32<T>(pe);
        }
      if constexpr (!MSUM) l_run[qb] += (ps[0] + ps[1]) + (ps[2] + ps[3]);
    }
    if constexpr (PVPIPE) __builtin_amdgcn_sched_barrier(0);
    if constexpr (MSUM) {
#pragma unroll
      for (int kh = 0; kh < 2; ++kh)
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) lacc[qb] = mfma16(ones.v, pf[kh][qb].v, lacc[qb]);
    }
    // ---- O^T += V^T . P^T ----
#pragma unroll
    for (int db = 0; db < 4; ++db) {
#pragma unroll
      for (int kh = 0; kh < 2; ++kh) {
        const int off = (32 * kh + vrow0) * 128 + (((2 * db + vch) ^ vx) << 4) + vsub;
        const typename Vec<T>::v8 vf = tr_pair<T>(vs, off, off + 16 * 128);
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) oacc[db][qb] = mfma16(vf, pf[kh][qb].v, oacc[db][qb]);
      }
    }
    if constexpr (PVPIPE) {
      // issue order pinned: the first two V fragments' reads, the LDS-free denominator MFMAs under their latency, then
      // fragment f + 2's reads right behind fragment f's MFMAs (two register sets; left alone hipcc issues a row block's
      // four reads only after the previous block's four MFMAs and exposes every read's full latency)
      __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
      if constexpr (MSUM) __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
#pragma unroll
      for (int f = 0; f < 6; ++f) {
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
      }
      __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  // K/V double buffer: one barrier per tile (drains this wave's DMAs, publishes tile t, frees tile t-1's stage)
  const int last_full = (Nk % KV_TILE == 0) ? nkt : nkt - 1;  // tiles [1, last_full) need no masking
  auto arrive = [&](int t) {
    __syncthreads();
    if (t + 1 < nkt) {
      if (DESC && t + 1 < last_full) load_kv_full(t + 1, (t + 1) & 1);
      else load_kv(t + 1, (t + 1) & 1);
    }
  };
  load_kv(0, 0);
  arrive(0);
  tile(0, std::integral_constant<int, 0>{}, std::true_type{});
  int kt = 1;
  // (one tile per trip, stage = kt & 1 at run time: unrolled by two - stages as compile-time constants - hipcc hoists both
  // stages' fragment reads across the trip and spills 22 registers at 168; this form holds 120, no scratch)
  for (; kt < last_full; ++kt) {
    arrive(kt);
    tile(kt, kt & 1, std::false_type{});
  }
  if (nkt > 1 && last_full < nkt) {
    arrive(nkt - 1);
    tile(nkt - 1, (nkt - 1) & 1, std::true_type{});
  }

#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
    float l_tot;
    if constexpr (MSUM) {
      l_tot = lacc[qb][0];
    } else {
      l_tot = l_run[qb] + xor16(l_run[qb]);
      l_tot += other_half(l_tot);
    }
    const float inv = 1.0f / l_tot;
    if (q_valid[qb]) {
      T* op = reinterpret_cast<T*>(p.o) + (int64_t)b * p.o_bs + (int64_t)qrow[qb] * p.o_rs + head * 64 + 4 * g;
#pragma unroll
      for (int db = 0; db < 4; ++db) {
        Pack4<T> ov;
#pragma unroll
        for (int e = 0; e < 4; ++e) ov.e[e] = from_f32<T>(oacc[db][qb][e] * inv);
        *reinterpret_cast<u32x2*>(op + 16 * db) = ov.u;
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


}  // namespace pm

// launcher used by pm_attention (attn.hip)
namespace pm {
int launch_attn_self16(const AttnParams& p, int dtype, dim3 grid, hipStream_t stream, int form) {
  if constexpr (PM_DIAG_BUILD) {  // (A/B runs: 1 = per-lane addresses of the K/V fetch, 2 = row sums on the vector pipe)
    if (form == 1) {
      PM_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL((attn_self16_kernel<T, false, true, false>), grid, dim3(256), 0, stream, p);
                        return check_launch());
    }
    if (form == 2) {
      PM_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL((attn_self16_kernel<T, true, false, false>), grid, dim3(256), 0, stream, p);
                        return check_launch());
    }
  }
  if constexpr (PM_DIAG_BUILD) {
    if (form == 3) {  // hipcc's own issue order in the PV phase
      PM_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL((attn_self16_kernel<T, true, true, false>), grid, dim3(256), 0, stream, p);
                        return check_launch());
    }
  }
  PM_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL((attn_self16_kernel<T, true, true, true>), grid, dim3(256), 0, stream, p);
                    return check_launch());
}
}  // namespace pm
