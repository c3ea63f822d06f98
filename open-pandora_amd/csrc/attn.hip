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
#include "common.hpp"

namespace pm {

struct AttnParams {
  const void* q;
  const void* k[2];
  const void* v[2];
  void* o;
  int64_t q_bs, q_rs, o_bs, o_rs;
  int64_t k_bs[2], k_rs[2];
  int Nk[2];
  float w[2];
  int nseg;
  int Nq, heads;
  float scale_log2e;
};

constexpr int KV_TILE = 64;
constexpr int KV_TILE_BYTES = KV_TILE * 128;  // 64 keys x 64 dims x 2 B

typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

template <typename T> __device__ __forceinline__ typename Vec<T>::v8 tr_pair(const char* base, int off0, int off1) {
  // two transposed 4x16 block reads -> 8 keys of one d column (the 32x32x16 A-operand fragment)
  union {
    struct { s16x4 lo, hi; } s;
    typename Vec<T>::v8 v;
  } u;
  u.s.lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(base + off0));
  u.s.hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(base + off1));
  return u.v;
}

template <typename T>
__global__ __launch_bounds__(256, 2) void attn_kernel(const AttnParams p) {
  __shared__ __attribute__((aligned(16))) char smem[4 * KV_TILE_BYTES];  // K[2], V[2]
  char* const Ks = smem;
  char* const Vs = smem + 2 * KV_TILE_BYTES;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ql = lane & 31, hh = lane >> 5;
  const int bh = blockIdx.y;
  const int b = bh / p.heads, head = bh - b * p.heads;
  const int q0 = blockIdx.x * 128 + wave * 32;
  int qrow = q0 + ql;
  const bool q_valid = qrow < p.Nq;
  if (!q_valid) qrow = p.Nq - 1;

  // Q^T fragments (B operand): element j of k-step s = Q[qrow][16 s + 8 hh + j]
  const T* qp = reinterpret_cast<const T*>(p.q) + (int64_t)b * p.q_bs + (int64_t)qrow * p.q_rs +
                head * 64 + 8 * hh;
  Pack8<T> qf[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) qf[s].u = ld_global16(qp + 16 * s);

  // staging coordinates
  const int lc = tid & 7, lr = tid >> 3;  // chunk, row (+32)

  f32x16 out[2];
#pragma unroll
  for (int d = 0; d < 2; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) out[d][r] = 0.f;

  for (int seg = 0; seg < p.nseg; ++seg) {
    const int Nk = p.Nk[seg];
    const T* kg = reinterpret_cast<const T*>(p.k[seg]) + (int64_t)b * p.k_bs[seg] + head * 64 + lc * 8;
    const T* vg = reinterpret_cast<const T*>(p.v[seg]) + (int64_t)b * p.k_bs[seg] + head * 64 + lc * 8;
    const int64_t krs = p.k_rs[seg];
    const int nkt = (Nk + KV_TILE - 1) / KV_TILE;

    u32x4 rk[2], rv[2];
    auto load_kv = [&](int kt) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        int key = kt * KV_TILE + lr + 32 * j;
        if (key > Nk - 1) key = Nk - 1;
        rk[j] = ld_global16(kg + (int64_t)key * krs);
        rv[j] = ld_global16(vg + (int64_t)key * krs);
      }
    };
    auto store_kv = [&](int buf) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int row = lr + 32 * j;
        *reinterpret_cast<u32x4*>(Ks + buf * KV_TILE_BYTES + row * 128 + ((lc ^ (row & 7)) << 4)) = rk[j];
        *reinterpret_cast<u32x4*>(Vs + buf * KV_TILE_BYTES + row * 128 +
                                  ((lc ^ (((row >> 1) & 1) << 2)) << 4)) = rv[j];
      }
    };

    f32x16 oacc[2];
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
      for (int r = 0; r < 16; ++r) oacc[d][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;

    load_kv(0);
    store_kv(0);
    __syncthreads();

    for (int kt = 0; kt < nkt; ++kt) {
      const int buf = kt & 1;
      if (kt + 1 < nkt) load_kv(kt + 1);
      const char* ks = Ks + buf * KV_TILE_BYTES;
      const char* vs = Vs + buf * KV_TILE_BYTES;

      // ---- S^T = K . Q^T (raw scores, f32) ----
      f32x16 sacc[2];
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
        for (int r = 0; r < 16; ++r) sacc[kb][r] = 0.f;
        const int row = kb * 32 + ql;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          Pack8<T> kf;
          const int chunk = 2 * s + hh;
          kf.u = *reinterpret_cast<const u32x4*>(ks + row * 128 + ((chunk ^ (row & 7)) << 4));
          sacc[kb] = mfma32(kf.v, qf[s].v, sacc[kb]);
        }
      }
      // sacc[kb][r] <-> key = kt*64 + kb*32 + (r&3) + 8*(r>>2) + 4*hh, query = ql
      if (kt * KV_TILE + KV_TILE > Nk) {
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int key = kt * KV_TILE + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
            if (key >= Nk) sacc[kb][r] = -INFINITY;
          }
      }
      // ---- online softmax (base-2 domain) ----
      float mx = sacc[0][0];
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) mx = fmaxf(mx, sacc[kb][r]);
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      const float m_new = fmaxf(m_run, mx * p.scale_log2e);
      const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
      m_run = m_new;
      float psum = 0.f;
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float pv = __builtin_amdgcn_exp2f(sacc[kb][r] * p.scale_log2e - m_new);
          sacc[kb][r] = pv;
          psum += pv;
        }
      l_run = l_run * alpha + psum;
#pragma unroll
      for (int d = 0; d < 2; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[d][r] *= alpha;

      // ---- O^T += V^T . P^T ----
      const int trow = (lane & 15) >> 2;                 // row inside the 4x16 block
      const int tcol8 = ((lane >> 4) & 1) * 2 + ((lane & 3) >> 1);  // 16-byte chunk inside the d-block
      const int tsub = (lane & 1) * 8;                   // byte inside the chunk
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) {
        Pack8<T> pf;
        const int kb = s4 >> 1, sp = s4 & 1;
#pragma unroll
        for (int j = 0; j < 8; ++j) pf.e[j] = from_f32<T>(sacc[kb][8 * sp + j]);
#pragma unroll
        for (int db = 0; db < 2; ++db) {
          const int r0 = 16 * s4 + 4 * hh + trow;
          const int r1 = r0 + 8;
          const int ch = db * 4 + tcol8;
          const int off0 = r0 * 128 + ((ch ^ (((r0 >> 1) & 1) << 2)) << 4) + tsub;
          const int off1 = r1 * 128 + ((ch ^ (((r1 >> 1) & 1) << 2)) << 4) + tsub;
          typename Vec<T>::v8 vf = tr_pair<T>(vs, off0, off1);
          oacc[db] = mfma32(vf, pf.v, oacc[db]);
        }
      }
      if (kt + 1 < nkt) store_kv(buf ^ 1);
      __syncthreads();
    }
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = p.w[seg] / l_tot;
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
      for (int r = 0; r < 16; ++r) out[d][r] += oacc[d][r] * inv;
  }

  // out[db][r] <-> d = db*32 + (r&3) + 8*(r>>2) + 4*hh for query row ql
  if (q_valid) {
    T* op = reinterpret_cast<T*>(p.o) + (int64_t)b * p.o_bs + (int64_t)qrow * p.o_rs + head * 64;
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        Pack4<T> ov;
#pragma unroll
        for (int e = 0; e < 4; ++e) ov.e[e] = from_f32<T>(out[db][4 * g + e]);
        *reinterpret_cast<u32x2*>(op + db * 32 + 8 * g + 4 * hh) = ov.u;
      }
  }
}

// ------------------------------------------------------------------------------------------------
// Temporal self-attention: at every (pixel, head) a 16x16 (Fq x Fk) attention over the frame axis.
// HBM-bound (0.1 % of the FLOPs): one thread per (pixel, head, query frame), f32 math in registers;
// the Fq lanes of one (pixel, head) read the same K/V rows (served once from L1).
struct TAttnParams {
  const void* q;
  const void* k;
  const void* v;
  void* o;
  int64_t ldq, ldk, ldo;
  int Fq, Fk, P, heads;
  float scale;
};

template <typename T>
__global__ __launch_bounds__(256) void tattn_kernel(const TAttnParams p) {
  const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int fq = (int)(gid % p.Fq);
  const int64_t ph = gid / p.Fq;
  const int64_t total = (int64_t)p.P * p.heads;
  if (ph >= total) return;
  const int head = (int)(ph % p.heads);
  const int64_t pix = ph / p.heads;

  const T* qp = reinterpret_cast<const T*>(p.q) + ((int64_t)fq * p.P + pix) * p.ldq + head * 64;
  float qv[64];
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    Pack8<T> t;
    t.u = ld_global16(qp + 8 * c);
#pragma unroll
    for (int e = 0; e < 8; ++e) qv[8 * c + e] = to_f32(t.e[e]) * p.scale;
  }
  float sc[16];
  float mx = -INFINITY;
  const T* kp = reinterpret_cast<const T*>(p.k) + pix * p.ldk + head * 64;
  const T* vp = reinterpret_cast<const T*>(p.v) + pix * p.ldk + head * 64;
  const int64_t fstride = (int64_t)p.P * p.ldk;
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    if (j < p.Fk) {
      float s = 0.f;
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        Pack8<T> t;
        t.u = ld_global16(kp + j * fstride + 8 * c);
#pragma unroll
        for (int e = 0; e < 8; ++e) s = fmaf(qv[8 * c + e], to_f32(t.e[e]), s);
      }
      sc[j] = s;
      mx = fmaxf(mx, s);
    }
  }
  float l = 0.f;
#pragma unroll
  for (int j = 0; j < 16; ++j)
    if (j < p.Fk) {
      sc[j] = __expf(sc[j] - mx);
      l += sc[j];
    }
  const float inv = 1.f / l;
  float ov[64];
#pragma unroll
  for (int e = 0; e < 64; ++e) ov[e] = 0.f;
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    if (j < p.Fk) {
      const float pj = sc[j] * inv;
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        Pack8<T> t;
        t.u = ld_global16(vp + j * fstride + 8 * c);
#pragma unroll
        for (int e = 0; e < 8; ++e) ov[8 * c + e] = fmaf(pj, to_f32(t.e[e]), ov[8 * c + e]);
      }
    }
  }
  T* op = reinterpret_cast<T*>(p.o) + ((int64_t)fq * p.P + pix) * p.ldo + head * 64;
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    Pack8<T> t;
#pragma unroll
    for (int e = 0; e < 8; ++e) t.e[e] = from_f32<T>(ov[8 * c + e]);
    st_global16(op + 8 * c, t.u);
  }
}

}  // namespace pm

using namespace pm;

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
  if (B * heads > 65535) return PM_E_SHAPE;
  AttnParams p{};
  p.q = q; p.o = o; p.q_bs = q_bs; p.q_rs = q_rs; p.o_bs = o_bs; p.o_rs = o_rs;
  p.k[0] = k1; p.v[0] = v1; p.k_bs[0] = k1_bs; p.k_rs[0] = k1_rs; p.Nk[0] = (int)Nk1; p.w[0] = 1.f;
  p.k[1] = k2; p.v[1] = v2; p.k_bs[1] = k2_bs; p.k_rs[1] = k2_rs; p.Nk[1] = (int)Nk2; p.w[1] = w2;
  p.nseg = k2 ? 2 : 1;
  p.Nq = (int)Nq; p.heads = (int)heads;
  p.scale_log2e = scale * 1.4426950408889634f;
  dim3 grid((unsigned)((Nq + 127) / 128), (unsigned)(B * heads));
  PM_DISPATCH_DTYPE(dtype, T,
                    hipLaunchKernelGGL((attn_kernel<T>), grid, dim3(256), 0, (hipStream_t)stream, p);
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
  p.Fq = (int)Fq; p.Fk = (int)Fk; p.P = (int)P; p.heads = (int)heads; p.scale = scale;
  const int64_t threads = P * heads * Fq;
  dim3 grid((unsigned)((threads + 255) / 256));
  PM_DISPATCH_DTYPE(dtype, T,
                    hipLaunchKernelGGL((tattn_kernel<T>), grid, dim3(256), 0, (hipStream_t)stream, p);
                    return check_launch());
}
