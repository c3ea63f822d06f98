// HBM-bound normalisation kernels on channels-last activations: GroupNorm(32) with optional fused
// SiLU (two passes: deterministic partial statistics, then apply) and LayerNorm.  All loads/stores
// are 16 bytes per lane along the contiguous channel axis; statistics are f32.
#include "common.hpp"

namespace pm {

constexpr int GN_ROWS_PER_CHUNK = 512;

__host__ __device__ inline int gn_threads(int CV) {
  // threads per block = CV * k (every thread keeps a fixed 8-channel column), <= 1024
  int k = 256 / CV;
  if (k < 1) k = 1;
  return CV * k;
}

// grid (nchunks, NI); block gn_threads(C/8)
template <typename T>
__global__ void gn_stats_kernel(const T* __restrict__ x, int64_t ldx, float* __restrict__ partials,
                                int P, int C, int groups, int nchunks) {
  extern __shared__ __attribute__((aligned(16))) float sh[];  // [2][C]
  const int CV = C >> 3;
  const int k = blockDim.x / CV;
  const int cv = threadIdx.x % CV, rlane = threadIdx.x / CV;
  const int inst = blockIdx.y, chunk = blockIdx.x;
  const int r0 = chunk * GN_ROWS_PER_CHUNK;
  int r1 = r0 + GN_ROWS_PER_CHUNK;
  if (r1 > P) r1 = P;
  const T* xp = x + ((int64_t)inst * P) * ldx + cv * 8;
  float s[8], ss[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) s[e] = ss[e] = 0.f;
  for (int r = r0 + rlane; r < r1; r += k) {
    Pack8<T> t;
    t.u = ld_global16(xp + (int64_t)r * ldx);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float v = to_f32(t.e[e]);
      s[e] += v;
      ss[e] = fmaf(v, v, ss[e]);
    }
  }
  // reduce the k row-lanes per channel through LDS (fixed order => deterministic)
  for (int i = threadIdx.x; i < 2 * C; i += blockDim.x) sh[i] = 0.f;
  __syncthreads();
  for (int turn = 0; turn < k; ++turn) {
    if (rlane == turn) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        sh[cv * 8 + e] += s[e];
        sh[C + cv * 8 + e] += ss[e];
      }
    }
    __syncthreads();
  }
  const int cpg = C / groups;
  if (threadIdx.x < groups) {
    float a = 0.f, b = 0.f;
    for (int c = 0; c < cpg; ++c) {
      a += sh[threadIdx.x * cpg + c];
      b += sh[C + threadIdx.x * cpg + c];
    }
    float* out = partials + (((int64_t)inst * nchunks + chunk) * groups + threadIdx.x) * 2;
    out[0] = a;
    out[1] = b;
  }
}

// grid (nblocks, NI); block 256.  sh: scale[C], shift[C]
template <typename T>
__global__ __launch_bounds__(256) void gn_apply_kernel(const T* __restrict__ x, int64_t ldx,
                                                       const float* __restrict__ partials,
                                                       int nchunks, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta,
                                                       T* __restrict__ y, int64_t ldy, int P, int C,
                                                       int groups, float inv_count, float eps,
                                                       int silu) {
  extern __shared__ __attribute__((aligned(16))) float sh[];  // [2][C] + [2][groups]
  float* scale = sh;
  float* shift = sh + C;
  float* gstat = sh + 2 * C;  // mean, rstd per group
  const int inst = blockIdx.y;
  if ((int)threadIdx.x < groups) {
    float a = 0.f, b = 0.f;
    const float* pp = partials + ((int64_t)inst * nchunks * groups + threadIdx.x) * 2;
    for (int c = 0; c < nchunks; ++c) {
      a += pp[(int64_t)c * groups * 2];
      b += pp[(int64_t)c * groups * 2 + 1];
    }
    const float mean = a * inv_count;
    float var = b * inv_count - mean * mean;
    if (var < 0.f) var = 0.f;
    gstat[threadIdx.x * 2] = mean;
    gstat[threadIdx.x * 2 + 1] = rsqrtf(var + eps);
  }
  __syncthreads();
  const int cpg = C / groups;
  for (int c = threadIdx.x; c < C; c += 256) {
    const int g = c / cpg;
    const float sc = gstat[g * 2 + 1] * gamma[c];
    scale[c] = sc;
    shift[c] = beta[c] - gstat[g * 2] * sc;
  }
  __syncthreads();
  const int CV = C >> 3;
  const int64_t total = (int64_t)P * CV;
  const T* xp = x + (int64_t)inst * P * ldx;
  T* yp = y + (int64_t)inst * P * ldy;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t r = i / CV;
    const int cv = (int)(i - r * CV);
    Pack8<T> t, o;
    t.u = ld_global16(xp + r * ldx + cv * 8);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float v = fmaf(to_f32(t.e[e]), scale[cv * 8 + e], shift[cv * 8 + e]);
      if (silu) v = silu_f(v);
      o.e[e] = from_f32<T>(v);
    }
    st_global16(yp + r * ldy + cv * 8, o.u);
  }
}

// one wave per row, 4 rows per block; up to 8 vectors (64 channels... 4096) per lane
template <typename T>
__global__ __launch_bounds__(256) void layernorm_kernel(const T* __restrict__ x, int64_t ldx,
                                                        const float* __restrict__ gamma,
                                                        const float* __restrict__ beta,
                                                        T* __restrict__ y, int64_t ldy, int M, int C,
                                                        float eps) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const int CV = C >> 3;
  const T* xp = x + (int64_t)row * ldx;
  float v[8][8];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int cv = lane + 64 * i;
    if (cv < CV) {
      Pack8<T> t;
      t.u = ld_global16(xp + cv * 8);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        v[i][e] = to_f32(t.e[e]);
        s += v[i][e];
      }
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
      Pack8<T> o;
      const f32x4 g0 = *reinterpret_cast<const f32x4*>(gamma + cv * 8);
      const f32x4 g1 = *reinterpret_cast<const f32x4*>(gamma + cv * 8 + 4);
      const f32x4 b0 = *reinterpret_cast<const f32x4*>(beta + cv * 8);
      const f32x4 b1 = *reinterpret_cast<const f32x4*>(beta + cv * 8 + 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        o.e[e] = from_f32<T>(fmaf((v[i][e] - mean) * rstd, g0[e], b0[e]));
        o.e[e + 4] = from_f32<T>(fmaf((v[i][e + 4] - mean) * rstd, g1[e], b1[e]));
      }
      st_global16(yp + cv * 8, o.u);
    }
  }
}

}  // namespace pm

using namespace pm;

extern "C" int64_t pm_groupnorm_nchunks(int64_t P, int64_t C) {
  (void)C;
  return (P + GN_ROWS_PER_CHUNK - 1) / GN_ROWS_PER_CHUNK;
}

static int gn_check(int64_t NI, int64_t P, int64_t C, int groups, int64_t ldx) {
  if (NI < 1 || P < 1 || C < 8 || (C & 7) || groups < 1 || groups > 256 || (C % groups)) return PM_E_SHAPE;
  if ((C >> 3) > 1024 || (ldx & 7) || ldx < C || NI > 65535) return PM_E_SHAPE;
  return PM_OK;
}

extern "C" int pm_groupnorm_stats(const void* x, int64_t ldx, float* partials, int64_t NI, int64_t P,
                                  int64_t C, int groups, int dtype, void* stream) {
  if (!x || !partials) return PM_E_NULL;
  int rc = gn_check(NI, P, C, groups, ldx);
  if (rc) return rc;
  const int nchunks = (int)pm_groupnorm_nchunks(P, C);
  const int threads = gn_threads((int)(C >> 3));
  if (threads < groups) return PM_E_SHAPE;
  dim3 grid(nchunks, (unsigned)NI);
  const size_t shmem = 2 * C * sizeof(float);
  PM_DISPATCH_DTYPE(dtype, T,
                    hipLaunchKernelGGL((gn_stats_kernel<T>), grid, dim3(threads), shmem,
                                       (hipStream_t)stream, (const T*)x, ldx, partials, (int)P,
                                       (int)C, groups, nchunks);
                    return check_launch());
}

extern "C" int pm_groupnorm_apply(const void* x, int64_t ldx, const float* partials, int64_t nchunks,
                                  const float* gamma, const float* beta, void* y, int64_t ldy,
                                  int64_t NI, int64_t P, int64_t C, int groups, double count,
                                  float eps, int silu, int dtype, void* stream) {
  if (!x || !partials || !gamma || !beta || !y) return PM_E_NULL;
  int rc = gn_check(NI, P, C, groups, ldx);
  if (rc) return rc;
  if ((ldy & 7) || ldy < C || nchunks < 1 || count <= 0) return PM_E_SHAPE;
  const int64_t vecs = P * (C >> 3);
  int64_t nb = (vecs + 256 * 4 - 1) / (256 * 4);  // ~4 vectors per thread
  const int64_t cap = (2048 + NI - 1) / NI;
  if (nb > cap) nb = cap;
  if (nb < 1) nb = 1;
  dim3 grid((unsigned)nb, (unsigned)NI);
  const size_t shmem = (2 * C + 2 * groups) * sizeof(float);
  PM_DISPATCH_DTYPE(dtype, T,
                    hipLaunchKernelGGL((gn_apply_kernel<T>), grid, dim3(256), shmem,
                                       (hipStream_t)stream, (const T*)x, ldx, partials, (int)nchunks,
                                       gamma, beta, (T*)y, ldy, (int)P, (int)C, groups,
                                       (float)(1.0 / count), eps, silu);
                    return check_launch());
}

extern "C" int pm_layernorm(const void* x, int64_t ldx, const float* gamma, const float* beta,
                            void* y, int64_t ldy, int64_t M, int64_t C, float eps, int dtype,
                            void* stream) {
  if (!x || !gamma || !beta || !y) return PM_E_NULL;
  if (M < 1 || C < 8 || (C & 7) || C > 4096 || (ldx & 7) || (ldy & 7) || ldx < C || ldy < C)
    return PM_E_SHAPE;
  dim3 grid((unsigned)((M + 3) / 4));
  PM_DISPATCH_DTYPE(dtype, T,
                    hipLaunchKernelGGL((layernorm_kernel<T>), grid, dim3(256), 0, (hipStream_t)stream,
                                       (const T*)x, ldx, gamma, beta, (T*)y, ldy, (int)M, (int)C, eps);
                    return check_launch());
}
