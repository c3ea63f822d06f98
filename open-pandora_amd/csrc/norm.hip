// HBM-bound normalisation kernels on channels-last activations: GroupNorm(32) with optional fused
// SiLU (two passes: deterministic partial statistics, then apply) and LayerNorm.  All loads/stores
// are 16 bytes per lane along the contiguous channel axis; statistics are f32.
#include <type_traits>
#include "gemm_common.hpp"  // (gs_atomic_add / GS_INV_*: the int64 GroupNorm totals, r06)

namespace pm {

// rows per statistics chunk: enough chunks to fill 256 CUs, few enough for a cheap second level
__host__ __device__ inline int gn_rows_per_chunk(int64_t P) {
  return P <= 1024 ? 16 : (P <= 16384 ? 32 : (P <= 65536 ? 64 : 256));
}

// eight consecutive channels as f32, from a 16-bit or f32 row
template <typename TI> __device__ __forceinline__ void load8(const TI* p, float (&v)[8]) {
  if constexpr (std::is_same<TI, float>::value) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(p);
    const f32x4 b = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      v[e] = a[e];
      v[e + 4] = b[e];
    }
  } else {
    Pack8<TI> t;
    t.u = ld_global16(p);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = to_f32(t.e[e]);
  }
}

__host__ __device__ inline int gn_threads(int CV) {
  // threads per block = CV * k (every thread keeps a fixed 8-channel column), <= 1024
  int k = 256 / CV;
  if (k < 1) k = 1;
  return CV * k;
}

// grid (nchunks, NI); block gn_threads(C/8); LDS [k][2][C] floats
template <typename TI>
__global__ void gn_stats_kernel(const TI* __restrict__ x, int64_t ldx, float* __restrict__ partials,
                                int P, int C, int groups, int nchunks, long long* __restrict__ gtot) {
  extern __shared__ __attribute__((aligned(16))) float sh[];
  const int CV = C >> 3;
  const int k = blockDim.x / CV;
  const int cv = threadIdx.x % CV, rlane = threadIdx.x / CV;
  const int inst = blockIdx.y, chunk = blockIdx.x;
  const int rpc = gn_rows_per_chunk(P);
  const int r0 = chunk * rpc;
  int r1 = r0 + rpc;
  if (r1 > P) r1 = P;
  const TI* xp = x + ((int64_t)inst * P) * ldx + cv * 8;
  float s[8], ss[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) s[e] = ss[e] = 0.f;
  // four independent row loads in flight per thread (one per iteration left the pass latency-bound);
  // accumulated in row order, so the sums are those of the one-row-at-a-time walk
  for (int r = r0 + rlane; r < r1; r += 4 * k) {
    float v[4][8];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int ru = r + u * k;
      load8<TI>(xp + (int64_t)(ru < r1 ? ru : r) * ldx, v[u]);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (r + u * k < r1) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          s[e] += v[u][e];
          ss[e] = fmaf(v[u][e], v[u][e], ss[e]);
        }
      }
    }
  }
  // per-thread partials -> LDS [rlane][2][C]; then fixed-order reductions (deterministic)
  float* mine = sh + (int64_t)rlane * 2 * C;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    mine[cv * 8 + e] = s[e];
    mine[C + cv * 8 + e] = ss[e];
  }
  __syncthreads();
  for (int c = threadIdx.x; c < 2 * C; c += blockDim.x) {
    float a = sh[c];
    for (int j = 1; j < k; ++j) a += sh[j * 2 * C + c];
    sh[c] = a;
  }
  __syncthreads();
  const int cpg = C / groups;
  if ((int)threadIdx.x < 2 * groups) {
    const int g = threadIdx.x % groups, which = threadIdx.x / groups;
    const float* src = sh + which * C + g * cpg;
    float a = 0.f;
    for (int c = 0; c < cpg; ++c) a += src[c];
    if (gtot != nullptr)  // r06: straight into the int64 totals [NI][groups][4] (no finalize launch)
      gs_atomic_add(gtot + (((int64_t)inst * groups + g) * 4 + which * 2) * GS_STRIDE, a);
    else
      partials[(((int64_t)inst * nchunks + chunk) * groups + g) * 2 + which] = a;
  }
}

// grid NI; block 256: sums the per-chunk partials of one instance in a fixed order (deterministic):
// 256 threads = (2*groups values) x (256/(2*groups) chunk lanes); each lane walks its chunks in order,
// then the lanes are added in order.  Output totals [NI][groups][2].
__global__ __launch_bounds__(1024) void gn_finalize_kernel(const float* __restrict__ partials,
                                                           float* __restrict__ totals, int nchunks,
                                                           int groups) {
  __shared__ float red[1024];
  const int inst = blockIdx.x;
  const int nv = 2 * groups;
  const int lanes = 1024 / nv;
  const int vidx = threadIdx.x % nv, ln = threadIdx.x / nv;
  float a = 0.f;
  if (ln < lanes) {
    const float* pp = partials + (int64_t)inst * nchunks * nv + vidx;
    for (int c = ln; c < nchunks; c += lanes) a += pp[(int64_t)c * nv];
  }
  red[threadIdx.x] = a;
  __syncthreads();
  if ((int)threadIdx.x < nv) {
    float t = 0.f;
    for (int j = 0; j < lanes; ++j) t += red[j * nv + threadIdx.x];
    totals[(int64_t)inst * nv + threadIdx.x] = t;
  }
}

// grid (groups, NI); block 256: GroupNorm totals from the column statistics a GEMM-family epilogue left
// behind ([mtiles][C][2] per 128-row tile): instance i owns row tiles [i*tpi, (i+1)*tpi), group g the
// channels [g*cpg, (g+1)*cpg).  Fixed summation order (per-thread strided, then thread 0..255).
__global__ __launch_bounds__(256) void gn_finalize_colstats_kernel(const float* __restrict__ colstats,
                                                                   float* __restrict__ totals, int tpi,
                                                                   int C, int groups) {
  __shared__ float red[8];
  const int g = blockIdx.x, inst = blockIdx.y;
  const int cpg = C / groups;
  const int n = tpi * cpg;
  // entry i = (row block t, channel c) = (i / cpg, i % cpg), walked with stride 256 by increments (no division
  // in the loop) and four independent 8-byte loads in flight; per-thread order of addition is fixed
  float a = 0.f, b = 0.f;
  const int dq = 256 / cpg, dr = 256 - dq * cpg;
  int t = (int)threadIdx.x / cpg, c = (int)threadIdx.x - t * cpg;
  const float* base = colstats + ((int64_t)inst * tpi * C + g * cpg) * 2;
  for (int i = threadIdx.x; i < n; i += 1024) {
    float2 v[4];
    bool ok[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      ok[u] = (i + 256 * u) < n;
      v[u] = ok[u] ? *reinterpret_cast<const float2*>(base + ((int64_t)t * C + c) * 2) : make_float2(0.f, 0.f);
      t += dq;
      c += dr;
      if (c >= cpg) {
        c -= cpg;
        ++t;
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      a += v[u].x;
      b += v[u].y;
    }
  }
  // fixed reduction tree (deterministic): xor-shuffle inside each wave, then the 4 wave sums in order
  a = wave_sum(a);
  b = wave_sum(b);
  if ((threadIdx.x & 63) == 0) {
    red[threadIdx.x >> 6] = a;
    red[4 + (threadIdx.x >> 6)] = b;
  }
  __syncthreads();
  if (threadIdx.x < 2) {
    const float* r = red + 4 * threadIdx.x;
    totals[((int64_t)inst * groups + g) * 2 + threadIdx.x] = ((r[0] + r[1]) + r[2]) + r[3];
  }
}

// grid (ceil(P / rows_per_block), NI); block gn_threads(C/8) = (C/8) * k threads: a thread keeps one
// 8-channel column (its scale/shift live in registers: no LDS, no barrier, no index division) and walks
// rows r0 + rlane, + k, ... with four independent row loads in flight.
// nsum > 0 (r06): `totals` are int64 fixed-point limbs [NI * nsum][groups][4] (gemm_common.hpp) and instance i uses the SUM of
// entries i*nsum .. i*nsum + nsum-1 (per-frame sums of a clip add up to its (T,H,W) sums: exact in integers); nsum < 0: f32
// {sum, sumsq} entries [NI * -nsum][groups][2] summed the same way (in entry order)
template <typename TI, typename T>
__global__ void gn_apply_kernel(const TI* __restrict__ x, int64_t ldx, const float* __restrict__ totals,
                                const float* __restrict__ gamma, const float* __restrict__ beta,
                                T* __restrict__ y, int64_t ldy, int P, int C, int groups, float inv_count,
                                float eps, int silu, int rows_per_block, int lo_off, int nsum) {
  // lo_off != 0 (PM_OUT_HILO, the parity configuration): the row is written as [hi | lo], hi = round16(v) at its
  // column, lo = round16(v - hi) lo_off columns further: a consumer GEMM / conv over 2C channels with the weights
  // repeated sees the normalised activation at ~2x the mantissa
  const int CV = C >> 3;
  const int k = blockDim.x / CV;
  const int cv = threadIdx.x % CV, rlane = threadIdx.x / CV;
  const int inst = blockIdx.y;
  const int r0 = blockIdx.x * rows_per_block;
  int r1 = r0 + rows_per_block;
  if (r1 > P) r1 = P;
  const TI* xp = x + (int64_t)inst * P * ldx + cv * 8;
  T* yp = y + (int64_t)inst * P * ldy + cv * 8;
  float sc[8], sh[8];
  bool have = false;
  // integer totals (PM_TOTALS_I64): `nsum` partial entries of 4 limbs per group, a 64-byte sector each.  One thread per (group,
  // limb) sums its entries - independent loads, one round trip - and one thread per group turns the limbs into {mean, rstd} in
  // LDS (every thread walking all 4 nsum limbs of its groups itself: 64 dependent-looking loads in front of the first store,
  // +7 us per launch at nsum = 16)
  __shared__ long long s_limb[512];  // (groups <= 128: pm_groupnorm_apply)
  __shared__ float s_mr[256];
  if (nsum > 0) {
    const long long* ti = reinterpret_cast<const long long*>(totals);
    for (int idx = threadIdx.x; idx < groups * 4; idx += blockDim.x) {
      const long long* lp = ti + ((int64_t)inst * nsum * groups * 4 + idx) * GS_STRIDE;
      long long acc = 0;
      const int64_t fstep = (int64_t)groups * 4 * GS_STRIDE;
      int f = 0;
      for (; f + 8 <= nsum; f += 8) {  // eight requests in flight, then the adds (a rolled loop waits for every entry in turn:
        long long v[8];                //  n serial round trips, +6 us per launch at n = 16)
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = lp[(f + u) * fstep];
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += v[u];
      }
      for (; f < nsum; ++f) acc += lp[f * fstep];
      s_limb[idx] = acc;
    }
    __syncthreads();
    if ((int)threadIdx.x < groups) {
      const int g = threadIdx.x;
      const double sum = (double)s_limb[4 * g] * GS_INV_A + (double)s_limb[4 * g + 1] * GS_INV_B;
      const double sq = (double)s_limb[4 * g + 2] * GS_INV_A + (double)s_limb[4 * g + 3] * GS_INV_B;
      const double md = sum * (double)inv_count;
      double var = sq * (double)inv_count - md * md;
      if (var < 0.0) var = 0.0;
      s_mr[2 * g] = (float)md;
      s_mr[2 * g + 1] = rsqrtf((float)var + eps);
    }
    __syncthreads();
  } else if (nsum < 0) {  // f32 {sum, sumsq} entries, summed in entry order (a fixed order: the same bits every launch)
    float* s_f = reinterpret_cast<float*>(s_limb);
    for (int idx = threadIdx.x; idx < groups * 2; idx += blockDim.x) {
      const float* lp = totals + (int64_t)inst * (-nsum) * groups * 2 + idx;
      float acc = 0.f;
      const int64_t fstep = (int64_t)groups * 2;
      int f = 0;
      for (; f + 8 <= -nsum; f += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = lp[(f + u) * fstep];
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += v[u];
      }
      for (; f < -nsum; ++f) acc += lp[f * fstep];
      s_f[idx] = acc;
    }
    __syncthreads();
    if ((int)threadIdx.x < groups) {
      const int g = threadIdx.x;
      const float mean = s_f[2 * g] * inv_count;
      float var = s_f[2 * g + 1] * inv_count - mean * mean;
      if (var < 0.f) var = 0.f;
      s_mr[2 * g] = mean;
      s_mr[2 * g + 1] = rsqrtf(var + eps);
    }
    __syncthreads();
  }
  for (int r = r0 + rlane; r < r1; r += 4 * k) {
    float t[4][8];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int ru = r + u * k;
      load8<TI>(xp + (int64_t)(ru < r1 ? ru : r) * ldx, t[u]);
    }
    if (!have) {  // (behind the first row loads in program order: its loads overlap theirs)
      have = true;
      const int cpg = C / groups;
      const f32x4 g0 = *reinterpret_cast<const f32x4*>(gamma + cv * 8);
      const f32x4 g1 = *reinterpret_cast<const f32x4*>(gamma + cv * 8 + 4);
      const f32x4 b0 = *reinterpret_cast<const f32x4*>(beta + cv * 8);
      const f32x4 b1 = *reinterpret_cast<const f32x4*>(beta + cv * 8 + 4);
      int gprev = -1;
      float mean = 0.f, rstd = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int g = (cv * 8 + e) / cpg;
        if (g != gprev) {  // (a new group: at most two per 8-channel column once a group has >= 8 channels)
          gprev = g;
          if (nsum != 0) {
            mean = s_mr[2 * g];
            rstd = s_mr[2 * g + 1];
          } else {
            const float a = totals[((int64_t)inst * groups + g) * 2];
            const float bq = totals[((int64_t)inst * groups + g) * 2 + 1];
            mean = a * inv_count;
            float var = bq * inv_count - mean * mean;
            if (var < 0.f) var = 0.f;
            rstd = rsqrtf(var + eps);
          }
        }
        sc[e] = rstd * (e < 4 ? g0[e & 3] : g1[e & 3]);
        sh[e] = (e < 4 ? b0[e & 3] : b1[e & 3]) - mean * sc[e];
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int ru = r + u * k;
      if (ru < r1) {
        Pack8<T> o, lo;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          float v = fmaf(t[u][e], sc[e], sh[e]);
          if (silu) v = silu_f(v);
          o.e[e] = from_f32<T>(v);
          lo.e[e] = from_f32<T>(v - to_f32(o.e[e]));
        }
        st_global16(yp + (int64_t)ru * ldy, o.u);
        if (lo_off) st_global16(yp + (int64_t)ru * ldy + lo_off, lo.u);
      }
    }
  }
}

// LayerNorm: LPR lanes per row (16 / 32 / 64), 64/LPR rows per wave, 4 waves per block.  A lane holds up to
// 8 vectors of 4 channels (C <= 32 * LPR), all loaded before the first reduction: with one row per wave
// (the fallback below) a C = 320 row keeps 40 lanes busy with two loads each and the pass is latency-bound.
template <typename TI> __device__ __forceinline__ void load4(const TI* p, float (&v)[4]) {
  if constexpr (std::is_same<TI, float>::value) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(p);
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = a[e];
  } else {
    Pack4<TI> t;
    t.u = *reinterpret_cast<const u32x2*>(p);
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = to_f32(t.e[e]);
  }
}

template <typename TI, typename T, int LPR>
__global__ __launch_bounds__(256) void layernorm_rows_kernel(const TI* __restrict__ x, int64_t ldx,
                                                             const float* __restrict__ gamma,
                                                             const float* __restrict__ beta,
                                                             T* __restrict__ y, int64_t ldy, int M, int C,
                                                             float eps, int lo_off) {
  constexpr int RPW = 64 / LPR;
  const int lane = threadIdx.x & 63;
  const int sub = lane % LPR;
  const int row = (blockIdx.x * 4 + (threadIdx.x >> 6)) * RPW + lane / LPR;
  const bool live = row < M;
  const int C4 = C >> 2;
  const TI* xp = x + (int64_t)(live ? row : 0) * ldx;
  float v[8][4];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int c4 = sub + LPR * i;
    if (c4 < C4) {
      load4<TI>(xp + c4 * 4, v[i]);
#pragma unroll
      for (int e = 0; e < 4; ++e) s += v[i][e];
    }
  }
#pragma unroll
  for (int o = LPR / 2; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  const float mean = s / (float)C;
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    if (sub + LPR * i < C4) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float d = v[i][e] - mean;
        ss = fmaf(d, d, ss);
      }
    }
  }
#pragma unroll
  for (int o = LPR / 2; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
  const float rstd = rsqrtf(ss / (float)C + eps);
  if (!live) return;
  T* yp = y + (int64_t)row * ldy;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int c4 = sub + LPR * i;
    if (c4 < C4) {
      const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + c4 * 4);
      const f32x4 b = *reinterpret_cast<const f32x4*>(beta + c4 * 4);
      Pack4<T> o, lo;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float r = fmaf((v[i][e] - mean) * rstd, g[e], b[e]);
        o.e[e] = from_f32<T>(r);
        lo.e[e] = from_f32<T>(r - to_f32(o.e[e]));
      }
      *reinterpret_cast<u32x2*>(yp + c4 * 4) = o.u;
      if (lo_off) *reinterpret_cast<u32x2*>(yp + lo_off + c4 * 4) = lo.u;  // (PM_OUT_HILO: see gn_apply_kernel)
    }
  }
}

// one wave per row, 4 rows per block; up to 8 vectors (64 channels... 4096) per lane
template <typename TI, typename T>
__global__ __launch_bounds__(256) void layernorm_kernel(const TI* __restrict__ x, int64_t ldx,
                                                        const float* __restrict__ gamma,
                                                        const float* __restrict__ beta,
                                                        T* __restrict__ y, int64_t ldy, int M, int C,
                                                        float eps, int lo_off) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const int CV = C >> 3;
  const TI* xp = x + (int64_t)row * ldx;
  float v[8][8];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int cv = lane + 64 * i;
    if (cv < CV) {
      load8<TI>(xp + cv * 8, v[i]);
#pragma unroll
      for (int e = 0; e < 8; ++e) s += v[i][e];
    }
  }
  const float mean = wave_sum(s) / (float)C;
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int cv = lane + 64 * i;
    if (cv < CV) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float d = v[i][e] - mean;
        ss = fmaf(d, d, ss);
      }
    }
  }
  const float rstd = rsqrtf(wave_sum(ss) / (float)C + eps);
  T* yp = y + (int64_t)row * ldy;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int cv = lane + 64 * i;
    if (cv < CV) {
      Pack8<T> o, lo;
      const f32x4 g0 = *reinterpret_cast<const f32x4*>(gamma + cv * 8);
      const f32x4 g1 = *reinterpret_cast<const f32x4*>(gamma + cv * 8 + 4);
      const f32x4 b0 = *reinterpret_cast<const f32x4*>(beta + cv * 8);
      const f32x4 b1 = *reinterpret_cast<const f32x4*>(beta + cv * 8 + 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float r0 = fmaf((v[i][e] - mean) * rstd, g0[e], b0[e]);
        const float r1 = fmaf((v[i][e + 4] - mean) * rstd, g1[e], b1[e]);
        o.e[e] = from_f32<T>(r0);
        o.e[e + 4] = from_f32<T>(r1);
        lo.e[e] = from_f32<T>(r0 - to_f32(o.e[e]));
        lo.e[e + 4] = from_f32<T>(r1 - to_f32(o.e[e + 4]));
      }
      st_global16(yp + cv * 8, o.u);
      if (lo_off) st_global16(yp + lo_off + cv * 8, lo.u);
    }
  }
}

}  // namespace pm

using namespace pm;

extern "C" int64_t pm_groupnorm_nchunks(int64_t P, int64_t C) {
  (void)C;
  const int rpc = gn_rows_per_chunk(P);
  return (P + rpc - 1) / rpc;
}

static int gn_check(int64_t NI, int64_t P, int64_t C, int groups, int64_t ldx, int in_dtype) {
  if (NI < 1 || P < 1 || C < 8 || (C & 7) || groups < 1 || groups > 128 || (C % groups)) return PM_E_SHAPE;
  const int64_t amask = (in_dtype == PM_F32) ? 3 : 7;
  if ((C >> 3) > 1024 || (ldx & amask) || ldx < C || NI > 65535) return PM_E_SHAPE;
  return PM_OK;
}

// in_dtype in {F16, BF16, F32} x out_dtype in {F16, BF16}
#define PM_DISPATCH_IN_OUT(in_dtype, out_dtype, TI, TO, ...)                                   \
  do {                                                                                          \
    if ((out_dtype) == PM_F16) {                                                                \
      typedef pm::f16 TO;                                                                       \
      if ((in_dtype) == PM_F16) { typedef pm::f16 TI; __VA_ARGS__; }                            \
      else if ((in_dtype) == PM_F32) { typedef float TI; __VA_ARGS__; }                         \
      else return PM_E_DTYPE;                                                                   \
    } else if ((out_dtype) == PM_BF16) {                                                        \
      typedef pm::bf16 TO;                                                                      \
      if ((in_dtype) == PM_BF16) { typedef pm::bf16 TI; __VA_ARGS__; }                          \
      else if ((in_dtype) == PM_F32) { typedef float TI; __VA_ARGS__; }                         \
      else return PM_E_DTYPE;                                                                   \
    } else {                                                                                    \
      return PM_E_DTYPE;                                                                        \
    }                                                                                           \
  } while (0)

template <typename TI>
static int launch_stats(const void* x, int64_t ldx, float* partials, float* totals, int64_t NI, int64_t P,
                        int64_t C, int groups, hipStream_t stream, bool i64) {
  const int nchunks = (int)pm_groupnorm_nchunks(P, C);
  const int threads = gn_threads((int)(C >> 3));
  if (threads < 2 * groups) return PM_E_SHAPE;
  const int k = threads / (int)(C >> 3);
  dim3 grid(nchunks, (unsigned)NI);
  const size_t shmem = (size_t)k * 2 * C * sizeof(float);
  hipLaunchKernelGGL((gn_stats_kernel<TI>), grid, dim3(threads), shmem, stream, (const TI*)x, ldx,
                     partials, (int)P, (int)C, groups, nchunks, i64 ? reinterpret_cast<long long*>(totals) : nullptr);
  if (!i64)
    hipLaunchKernelGGL(gn_finalize_kernel, dim3((unsigned)NI), dim3(1024), 0, stream, partials, totals, nchunks,
                       groups);
  return check_launch();
}

extern "C" int pm_groupnorm_stats(const void* x, int64_t ldx, float* partials, float* totals, int64_t NI,
                                  int64_t P, int64_t C, int groups, int in_dtype, void* stream) {
  const bool i64 = (in_dtype & PM_TOTALS_I64) != 0;  // r06: `totals` = int64 [NI][groups][4], ADDED to (zeroed by the caller)
  in_dtype &= ~PM_TOTALS_I64;
  if (!x || (!partials && !i64) || !totals) return PM_E_NULL;
  int rc = gn_check(NI, P, C, groups, ldx, in_dtype);
  if (rc) return rc;
  hipStream_t st = (hipStream_t)stream;
  if (in_dtype == PM_F32) return launch_stats<float>(x, ldx, partials, totals, NI, P, C, groups, st, i64);
  PM_DISPATCH_DTYPE(in_dtype, T, return launch_stats<T>(x, ldx, partials, totals, NI, P, C, groups, st, i64));
}

extern "C" int pm_groupnorm_finalize_colstats(const float* colstats, float* totals, int64_t mtiles,
                                              int64_t C, int64_t NI, int groups, void* stream) {
  if (!colstats || !totals) return PM_E_NULL;
  if (NI < 1 || mtiles < NI || (mtiles % NI) || C < 1 || groups < 1 || (C % groups) || NI > 65535) return PM_E_SHAPE;
  hipLaunchKernelGGL(gn_finalize_colstats_kernel, dim3((unsigned)groups, (unsigned)NI), dim3(256), 0,
                     (hipStream_t)stream, colstats, totals, (int)(mtiles / NI), (int)C, groups);
  return check_launch();
}

extern "C" int pm_groupnorm_apply(const void* x, int64_t ldx, const float* totals,
                                  const float* gamma, const float* beta, void* y, int64_t ldy,
                                  int64_t NI, int64_t P, int64_t C, int groups, double count,
                                  float eps, int silu, int in_dtype, int out_dtype, void* stream) {
  if (!x || !totals || !gamma || !beta || !y) return PM_E_NULL;
  int rc = gn_check(NI, P, C, groups, ldx, in_dtype);
  if (rc) return rc;
  const int lo_off = (out_dtype & PM_OUT_HILO) ? (int)C : 0;
  // r06: PM_TOTALS_I64 in out_dtype: totals are int64 limbs; bits 16..23 = nsum (entries summed per instance, >= 1)
  // without PM_TOTALS_I64: bits 16..23 = n >= 2: f32 totals [NI * n][groups][2], instance i uses the sum of its n entries (0 / 1: one)
  int nsum = (out_dtype >> 16) & 0xff;
  if ((out_dtype & PM_TOTALS_I64) && nsum < 1) return PM_E_SHAPE;
  if (!(out_dtype & PM_TOTALS_I64)) nsum = nsum >= 2 ? -nsum : 0;
  out_dtype &= 0xff;
  if ((ldy & 7) || ldy < C + lo_off || count <= 0) return PM_E_SHAPE;
  const int threads = gn_threads((int)(C >> 3));
  const int k = threads / (int)(C >> 3);
  int64_t rpb = 4 * k;  // one batch of four rows per thread; more only to keep the grid under ~8192 blocks
  while (((P + rpb - 1) / rpb) * NI > 8192) rpb *= 2;
  // summed totals: every block adds up the n entries of its instance itself (n x groups x 4 limbs, a 64-byte sector each: 128 KB
  // of L2 reads per block at n = 16) - four batches per block keep that beside the rows' traffic instead of above it
  if ((nsum >= 4 || nsum <= -4) && ((P + 4 * rpb - 1) / (4 * rpb)) * NI >= 256) rpb *= 4;
  dim3 grid((unsigned)((P + rpb - 1) / rpb), (unsigned)NI);
  PM_DISPATCH_IN_OUT(in_dtype, out_dtype, TI, TO,
                     hipLaunchKernelGGL((gn_apply_kernel<TI, TO>), grid, dim3(threads), 0,
                                        (hipStream_t)stream, (const TI*)x, ldx, totals,
                                        gamma, beta, (TO*)y, ldy, (int)P, (int)C, groups,
                                        (float)(1.0 / count), eps, silu, (int)rpb, lo_off, nsum);
                     return check_launch());
}

extern "C" int pm_layernorm(const void* x, int64_t ldx, const float* gamma, const float* beta,
                            void* y, int64_t ldy, int64_t M, int64_t C, float eps, int in_dtype,
                            int out_dtype, void* stream) {
  if (!x || !gamma || !beta || !y) return PM_E_NULL;
  const int64_t amask = (in_dtype == PM_F32) ? 3 : 7;
  const int lo_off = (out_dtype & PM_OUT_HILO) ? (int)C : 0;
  out_dtype &= ~PM_OUT_HILO;
  if (M < 1 || C < 8 || (C & 7) || C > 4096 || (ldx & amask) || (ldy & 7) || ldx < C || ldy < C + lo_off)
    return PM_E_SHAPE;
  if (C <= 2048) {  // several rows per wave
    const int lpr = C <= 512 ? 16 : (C <= 1024 ? 32 : 64);
    const int rpb = 4 * (64 / lpr);
    dim3 g2((unsigned)((M + rpb - 1) / rpb));
#define PM_LN_ROWS(LPR)                                                                                   \
  PM_DISPATCH_IN_OUT(in_dtype, out_dtype, TI, TO,                                                         \
                     hipLaunchKernelGGL((layernorm_rows_kernel<TI, TO, LPR>), g2, dim3(256), 0,           \
                                        (hipStream_t)stream, (const TI*)x, ldx, gamma, beta, (TO*)y, ldy, \
                                        (int)M, (int)C, eps, lo_off);                                     \
                     return check_launch())
    if (lpr == 16) PM_LN_ROWS(16);
    if (lpr == 32) PM_LN_ROWS(32);
    PM_LN_ROWS(64);
#undef PM_LN_ROWS
  }
  dim3 grid((unsigned)((M + 3) / 4));
  PM_DISPATCH_IN_OUT(in_dtype, out_dtype, TI, TO,
                     hipLaunchKernelGGL((layernorm_kernel<TI, TO>), grid, dim3(256), 0,
                                        (hipStream_t)stream, (const TI*)x, ldx, gamma, beta, (TO*)y, ldy,
                                        (int)M, (int)C, eps, lo_off);
                     return check_launch());
}
