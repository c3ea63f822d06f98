// Small kernels on the edges of the path: f32 GEMV for the embedding MLPs, the fused DDIM update,
// and the boundary layout conversions (NCTHW f32 latent <-> channels-last 16-bit activations).
#include "common.hpp"

namespace pm {

// one wave per output row; K % 8 == 0
template <typename T>
__global__ __launch_bounds__(256) void gemv_kernel(const T* __restrict__ W, int64_t ldw,
                                                   const float* __restrict__ x,
                                                   const float* __restrict__ bias,
                                                   float* __restrict__ y, int N, int K, int silu_in,
                                                   int act) {
  const int lane = threadIdx.x & 63;
  const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (n >= N) return;
  const T* wp = W + (int64_t)n * ldw;
  float acc = 0.f;
  for (int k = lane * 8; k < K; k += 64 * 8) {
    Pack8<T> w;
    w.u = ld_global16(wp + k);
    const f32x4 x0 = *reinterpret_cast<const f32x4*>(x + k);
    const f32x4 x1 = *reinterpret_cast<const f32x4*>(x + k + 4);
    float xv[8] = {x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]};
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float xe = silu_in ? silu_f(xv[e]) : xv[e];
      acc = fmaf(to_f32(w.e[e]), xe, acc);
    }
  }
  acc = wave_sum(acc);
  if (lane == 0) {
    float v = acc + (bias ? bias[n] : 0.f);
    if (act == PM_ACT_SILU) v = silu_f(v);
    y[n] = v;
  }
}

template <typename T>
__global__ __launch_bounds__(256) void ddim_kernel(const float* __restrict__ x,
                                                   const T* __restrict__ e_c,
                                                   const T* __restrict__ e_u,
                                                   const float* __restrict__ noise,
                                                   float* __restrict__ x_prev,
                                                   float* __restrict__ pred_x0, int64_t n, float cfg,
                                                   float sqrt_ac, float sqrt_1mac, float rescale,
                                                   float sqrt_a_prev, float dir_coef, float sigma) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float ec = to_f32(e_c[i]);
    float v = ec;
    if (e_u != nullptr) {
      const float eu = to_f32(e_u[i]);
      v = eu + cfg * (ec - eu);
    }
    const float xi = x[i];
    const float eps = sqrt_ac * v + sqrt_1mac * xi;
    float x0 = sqrt_ac * xi - sqrt_1mac * v;
    x0 *= rescale;
    float xp = sqrt_a_prev * x0 + dir_coef * eps;
    if (noise != nullptr) xp += sigma * noise[i];
    x_prev[i] = xp;
    if (pred_x0 != nullptr) pred_x0[i] = x0;
  }
}

// y[f, p, c] = c < C1 ? x[c, f, p] : cond[c - C1, f, p]      (C1 + C2 is small: 8)
template <typename T>
__global__ __launch_bounds__(256) void pack_input_kernel(const float* __restrict__ x,
                                                         const float* __restrict__ cond,
                                                         T* __restrict__ y, int C1, int C2,
                                                         int64_t FP) {
  const int C = C1 + C2;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < FP; i += (int64_t)gridDim.x * 256) {
    for (int c = 0; c < C; ++c) {
      const float v = c < C1 ? x[(int64_t)c * FP + i] : cond[(int64_t)(c - C1) * FP + i];
      y[i * C + c] = from_f32<T>(v);
    }
  }
}

template <typename T>
__global__ __launch_bounds__(256) void unpack_output_kernel(const T* __restrict__ y,
                                                            T* __restrict__ out, int C,
                                                            int64_t FP) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < FP; i += (int64_t)gridDim.x * 256)
    for (int c = 0; c < C; ++c) out[(int64_t)c * FP + i] = y[i * C + c];
}

// y[m, :] = softmax(scale * x[m, :]) for f32 scores x [M, N] -> 16-bit probabilities; one block per row
template <typename T>
__global__ __launch_bounds__(256) void softmax_rows_kernel(const float* __restrict__ x, int64_t ldx,
                                                           T* __restrict__ y, int64_t ldy, int N,
                                                           float scale_log2e) {
  __shared__ float red[8];
  const float* xp = x + (int64_t)blockIdx.x * ldx;
  T* yp = y + (int64_t)blockIdx.x * ldy;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float mx = -INFINITY;
  for (int i = threadIdx.x * 4; i < N; i += 1024) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(xp + i);
    mx = fmaxf(fmaxf(mx, fmaxf(v[0], v[1])), fmaxf(v[2], v[3]));
  }
  mx = wave_max(mx);
  if (lane == 0) red[wave] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])) * scale_log2e;
  float sum = 0.f;
  for (int i = threadIdx.x * 4; i < N; i += 1024) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(xp + i);
#pragma unroll
    for (int e = 0; e < 4; ++e) sum += __builtin_amdgcn_exp2f(v[e] * scale_log2e - mx);
  }
  sum = wave_sum(sum);
  if (lane == 0) red[4 + wave] = sum;
  __syncthreads();
  const float inv = 1.0f / (red[4] + red[5] + red[6] + red[7]);
  for (int i = threadIdx.x * 4; i < N; i += 1024) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(xp + i);
    Pack4<T> o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o.e[e] = from_f32<T>(__builtin_amdgcn_exp2f(v[e] * scale_log2e - mx) * inv);
    *reinterpret_cast<u32x2*>(yp + i) = o.u;
  }
}

// y[f, p, c'] = sum_c W[c', c] * x[c, f, p] * inv_scale + b[c']  for c' < C, zero for C <= c' < Cpad
template <typename T>
__global__ __launch_bounds__(256) void latent_affine_kernel(const float* __restrict__ x,
                                                            const float* __restrict__ W,
                                                            const float* __restrict__ b,
                                                            T* __restrict__ y, int C, int Cpad,
                                                            int64_t FP, float inv_scale) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < FP; i += (int64_t)gridDim.x * 256) {
    for (int co = 0; co < Cpad; ++co) {
      float acc = 0.f;
      if (co < C) {
        acc = b[co];
        for (int c = 0; c < C; ++c) acc = fmaf(W[co * C + c], x[(int64_t)c * FP + i] * inv_scale, acc);
      }
      y[i * Cpad + co] = from_f32<T>(acc);
    }
  }
}

}  // namespace pm

using namespace pm;

extern "C" int pm_softmax_rows(const float* x, int64_t ldx, void* y, int64_t ldy, int64_t M, int64_t N,
                               float scale, int dtype, void* stream) {
  if (!x || !y) return PM_E_NULL;
  if (M < 1 || N < 4 || (N & 3) || (ldx & 3) || (ldy & 3) || ldx < N || ldy < N) return PM_E_SHAPE;
  PM_DISPATCH_DTYPE(dtype, T,
                    hipLaunchKernelGGL((softmax_rows_kernel<T>), dim3((unsigned)M), dim3(256), 0,
                                       (hipStream_t)stream, x, ldx, (T*)y, ldy, (int)N,
                                       scale * 1.4426950408889634f);
                    return check_launch());
}

extern "C" int pm_latent_affine(const float* x, const float* W, const float* b, void* y, int64_t C,
                                int64_t Cpad, int64_t F, int64_t P, float inv_scale, int dtype,
                                void* stream) {
  if (!x || !W || !b || !y) return PM_E_NULL;
  if (C < 1 || Cpad < C || F < 1 || P < 1) return PM_E_SHAPE;
  const int64_t FP = F * P;
  int64_t nb = (FP + 255) / 256;
  if (nb > 4096) nb = 4096;
  PM_DISPATCH_DTYPE(dtype, T,
                    hipLaunchKernelGGL((latent_affine_kernel<T>), dim3((unsigned)nb), dim3(256), 0,
                                       (hipStream_t)stream, x, W, b, (T*)y, (int)C, (int)Cpad, FP,
                                       inv_scale);
                    return check_launch());
}

// f32 rows -> 16-bit rows: y[:, :K] = round16(x) and, with_lo, y[:, K:2K] = round16(x - round16(x)) (the part the
// first rounding drops).  One thread per 8 elements: 2 x 16-byte loads, 1-2 x 16-byte stores.
template <typename T>
__global__ __launch_bounds__(256) void split16_kernel(const float* __restrict__ x, int64_t ldx, T* __restrict__ y,
                                                      int64_t ldy, int64_t M, int K8, int with_lo) {
  const int64_t total = M * K8;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t m = i / K8;
    const int c = (int)(i - m * K8) * 8;
    const float* xp = x + m * ldx + c;
    const pm::f32x4 a = *reinterpret_cast<const pm::f32x4*>(xp);
    const pm::f32x4 b = *reinterpret_cast<const pm::f32x4*>(xp + 4);
    pm::Pack8<T> hi, lo;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      hi.e[e] = pm::from_f32<T>(a[e]);
      hi.e[e + 4] = pm::from_f32<T>(b[e]);
    }
    T* yp = y + m * ldy + c;
    pm::st_global16(yp, hi.u);
    if (with_lo) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        lo.e[e] = pm::from_f32<T>(a[e] - pm::to_f32(hi.e[e]));
        lo.e[e + 4] = pm::from_f32<T>(b[e] - pm::to_f32(hi.e[e + 4]));
      }
      pm::st_global16(yp + (int64_t)K8 * 8, lo.u);
    }
  }
}

extern "C" int pm_split16(const float* x, int64_t ldx, void* y, int64_t ldy, int64_t M, int64_t K, int with_lo,
                          int dtype, void* stream) {
  if (!x || !y) return PM_E_NULL;
  if (M < 1 || K < 8 || (K & 7) || (ldx & 3) || ldx < K || (ldy & 7) || ldy < (with_lo ? 2 * K : K)) return PM_E_SHAPE;
  const int64_t total = M * (K / 8);
  int64_t nb = (total + 255) / 256;
  if (nb > 8192) nb = 8192;
  PM_DISPATCH_DTYPE(dtype, T,
                    hipLaunchKernelGGL((split16_kernel<T>), dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, x,
                                       ldx, reinterpret_cast<T*>(y), ldy, M, (int)(K / 8), with_lo));
  return pm::check_launch();
}

// The same conversion written through a nearest-neighbour x2 upsample: output pixel (f, oy, ox) of a 2H x 2W frame takes
// input pixel (f, oy >> 1, ox >> 1).  One thread per 8 elements of an OUTPUT row (each input chunk is read four times: L2).
template <typename T>
__global__ __launch_bounds__(256) void split16_up2_kernel(const float* __restrict__ x, int64_t ldx, T* __restrict__ y,
                                                          int64_t ldy, int64_t Mo, int K8, int with_lo, int H, int W) {
  const int64_t total = Mo * K8;
  const int Wo = 2 * W, HWo = 4 * H * W;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t mo = i / K8;
    const int c = (int)(i - mo * K8) * 8;
    const int64_t f = mo / HWo;
    const int rem = (int)(mo - f * HWo);
    const int oy = rem / Wo, ox = rem - oy * Wo;
    const int64_t mi = (f * H + (oy >> 1)) * W + (ox >> 1);
    const float* xp = x + mi * ldx + c;
    const pm::f32x4 a = *reinterpret_cast<const pm::f32x4*>(xp);
    const pm::f32x4 b = *reinterpret_cast<const pm::f32x4*>(xp + 4);
    pm::Pack8<T> hi, lo;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      hi.e[e] = pm::from_f32<T>(a[e]);
      hi.e[e + 4] = pm::from_f32<T>(b[e]);
    }
    T* yp = y + mo * ldy + c;
    pm::st_global16(yp, hi.u);
    if (with_lo) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        lo.e[e] = pm::from_f32<T>(a[e] - pm::to_f32(hi.e[e]));
        lo.e[e + 4] = pm::from_f32<T>(b[e] - pm::to_f32(hi.e[e + 4]));
      }
      pm::st_global16(yp + (int64_t)K8 * 8, lo.u);
    }
  }
}

extern "C" int pm_split16_upsample2x(const float* x, int64_t ldx, void* y, int64_t ldy, int64_t F, int64_t H, int64_t W,
                                     int64_t K, int with_lo, int dtype, void* stream) {
  if (!x || !y) return PM_E_NULL;
  if (F < 1 || H < 1 || W < 1 || H > 16384 || W > 16384 || K < 8 || (K & 7) || (ldx & 3) || ldx < K || (ldy & 7) ||
      ldy < (with_lo ? 2 * K : K))
    return PM_E_SHAPE;
  const int64_t Mo = F * 4 * H * W;
  const int64_t total = Mo * (K / 8);
  int64_t nb = (total + 255) / 256;
  if (nb > 16384) nb = 16384;
  PM_DISPATCH_DTYPE(dtype, T,
                    hipLaunchKernelGGL((split16_up2_kernel<T>), dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, x,
                                       ldx, reinterpret_cast<T*>(y), ldy, Mo, (int)(K / 8), with_lo, (int)H, (int)W));
  return pm::check_launch();
}

extern "C" const char* pm_strerror(int code) {
  switch (code) {
    case PM_OK: return "ok";
    case PM_E_DTYPE: return "unsupported dtype (want PM_F16 or PM_BF16)";
    case PM_E_SHAPE: return "shape or alignment precondition violated";
    case PM_E_NULL: return "required pointer is NULL";
    case PM_E_LAUNCH: return "kernel launch failed";
    case PM_E_WORKSPACE: return "workspace too small";
    default: return "unknown error";
  }
}

extern "C" int pm_abi_version(void) { return 1; }

extern "C" int pm_gemv_f32(const void* W, int64_t ldw, const float* x, const float* bias, float* y,
                           int64_t N, int64_t K, int silu_in, int act, int dtype, void* stream) {
  if (!W || !x || !y) return PM_E_NULL;
  if (N < 1 || K < 8 || (K & 7) || (ldw & 7) || ldw < K) return PM_E_SHAPE;
  if (act != PM_ACT_NONE && act != PM_ACT_SILU) return PM_E_SHAPE;
  dim3 grid((unsigned)((N + 3) / 4));
  PM_DISPATCH_DTYPE(dtype, T,
                    hipLaunchKernelGGL((gemv_kernel<T>), grid, dim3(256), 0, (hipStream_t)stream,
                                       (const T*)W, ldw, x, bias, y, (int)N, (int)K, silu_in, act);
                    return check_launch());
}

extern "C" int pm_ddim_update(const float* x, const void* e_c, const void* e_u, const float* noise,
                              float* x_prev, float* pred_x0, int64_t n, float cfg, float sqrt_ac,
                              float sqrt_1mac, float rescale, float sqrt_a_prev, float dir_coef,
                              float sigma, int dtype, void* stream) {
  if (!x || !e_c || !x_prev) return PM_E_NULL;
  if (n < 1) return PM_E_SHAPE;
  if (sigma != 0.f && !noise) return PM_E_NULL;
  int64_t nb = (n + 255) / 256;
  if (nb > 2048) nb = 2048;
  if (dtype == PM_F32) {
    hipLaunchKernelGGL((ddim_kernel<float>), dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, x,
                       (const float*)e_c, (const float*)e_u, noise, x_prev, pred_x0, n, cfg, sqrt_ac,
                       sqrt_1mac, rescale, sqrt_a_prev, dir_coef, sigma);
    return check_launch();
  }
  PM_DISPATCH_DTYPE(dtype, T,
                    hipLaunchKernelGGL((ddim_kernel<T>), dim3((unsigned)nb), dim3(256), 0,
                                       (hipStream_t)stream, x, (const T*)e_c, (const T*)e_u, noise,
                                       x_prev, pred_x0, n, cfg, sqrt_ac, sqrt_1mac, rescale,
                                       sqrt_a_prev, dir_coef, sigma);
                    return check_launch());
}

namespace pm {
// y[i, j] = cos(t_i f_j), y[i, half + j] = sin(t_i f_j): the product in f32 as the reference forms it, cosf / sinf with full
// range reduction (t f reaches 1000 rad)
__global__ __launch_bounds__(256) void timestep_embedding_kernel(const void* __restrict__ t, int is_i64,
                                                                 const float* __restrict__ freqs, float* __restrict__ y,
                                                                 int n, int half) {
  for (int idx = blockIdx.x * 256 + threadIdx.x; idx < n * half; idx += gridDim.x * 256) {
    const int i = idx / half, j = idx - i * half;
    const float tv = is_i64 ? (float)reinterpret_cast<const long long*>(t)[i] : reinterpret_cast<const float*>(t)[i];
    const float a = tv * freqs[j];
    y[(int64_t)i * 2 * half + j] = cosf(a);
    y[(int64_t)i * 2 * half + half + j] = sinf(a);
  }
}
}  // namespace pm

extern "C" int pm_timestep_embedding(const void* t, int t_is_i64, const float* freqs, float* y, int64_t n, int64_t half,
                                     void* stream) {
  if (!t || !freqs || !y) return PM_E_NULL;
  if (n < 1 || half < 1 || n * half > (1 << 24)) return PM_E_SHAPE;
  int64_t nb = (n * half + 255) / 256;
  if (nb > 1024) nb = 1024;
  hipLaunchKernelGGL(pm::timestep_embedding_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, t, t_is_i64, freqs, y,
                     (int)n, (int)half);
  return check_launch();
}

extern "C" int pm_pack_input(const float* x, const float* cond, void* y, int64_t C1, int64_t C2,
                             int64_t F, int64_t P, int dtype, void* stream) {
  if (!x || !y || (C2 > 0 && !cond)) return PM_E_NULL;
  if (C1 < 1 || C2 < 0 || F < 1 || P < 1) return PM_E_SHAPE;
  const int64_t FP = F * P;
  int64_t nb = (FP + 255) / 256;
  if (nb > 2048) nb = 2048;
  PM_DISPATCH_DTYPE(dtype, T,
                    hipLaunchKernelGGL((pack_input_kernel<T>), dim3((unsigned)nb), dim3(256), 0,
                                       (hipStream_t)stream, x, cond, (T*)y, (int)C1, (int)C2, FP);
                    return check_launch());
}

extern "C" int pm_unpack_output(const void* y, void* out, int64_t C, int64_t F, int64_t P,
                                int dtype, void* stream) {
  if (!y || !out) return PM_E_NULL;
  if (C < 1 || F < 1 || P < 1) return PM_E_SHAPE;
  const int64_t FP = F * P;
  int64_t nb = (FP + 255) / 256;
  if (nb > 2048) nb = 2048;
  if (dtype == PM_F32) {
    hipLaunchKernelGGL((unpack_output_kernel<float>), dim3((unsigned)nb), dim3(256), 0,
                       (hipStream_t)stream, (const float*)y, (float*)out, (int)C, FP);
    return check_launch();
  }
  PM_DISPATCH_DTYPE(dtype, T,
                    hipLaunchKernelGGL((unpack_output_kernel<T>), dim3((unsigned)nb), dim3(256), 0,
                                       (hipStream_t)stream, (const T*)y, (T*)out, (int)C, FP);
                    return check_launch());
}

#ifdef PM_DIAG
// diagnostics: the epilogues' erf / GELU approximants, element by element (mode 0: erf_fast_f, 1: gelu_erf_f) - the op-level
// test of common.hpp's fit against an f64 erf (tests/test_ops_gpu.py::test_erf_approximant)
namespace pm {
__global__ void debug_erf_kernel(const float* x, float* y, long n, int mode) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    y[i] = mode ? gelu_erf_f(x[i]) : erf_fast_f(x[i]);
}
}  // namespace pm
extern "C" int pm_debug_erf(const float* x, float* y, int64_t n, int mode, void* stream) {
  if (!x || !y) return PM_E_NULL;
  if (n < 1) return PM_E_SHAPE;
  hipLaunchKernelGGL(pm::debug_erf_kernel, dim3(1024), dim3(256), 0, (hipStream_t)stream, x, y, (long)n, mode);
  return pm::check_launch();
}
#endif
