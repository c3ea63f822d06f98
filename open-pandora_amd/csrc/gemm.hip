// MFMA GEMM core for gfx950 with three A-operand loaders (dense rows, 3x3 im2col on channels-last
// frames, 3-tap temporal im2col) and a fused epilogue (bias, SiLU / GEGLU, residual).
//
//   C[M, N] = epi(A[M, K] . W[N, K]^T),  f32 accumulate on v_mfma_f32_16x16x32_{f16,bf16}.
//
// Tiling: 128x128 output tile per 256-thread workgroup (4 waves as 2x2, 64x64 per wave = 4x4 MFMA
// blocks), BK = 64.  A and W tiles are staged global -> VGPR -> LDS (issue-early / write-late, one
// barrier per K-tile, two LDS buffers); LDS rows are 128 B with the 16-byte chunk index XOR-ed by
// (row & 7) so the ds_read_b128 fragment reads spread over the banks.  The epilogue goes through LDS
// (f32) so that residual loads and output stores are full 16-byte row segments.
#include <type_traits>
#include "common.hpp"

namespace pm {

enum { A_DENSE = 0, A_CONV3X3 = 1, A_CONVT3 = 2 };

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
  int out32;
  int ntiles;
  // conv3x3
  int Hin, Win, Hv, Wv, Cin, Ho, Wo, stride, ups;
  // temporal conv
  int F, P;
  const void* halo_lo;
  const void* halo_hi;
  const void* zero;
};

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = BM * BK * 2;  // 16 KiB per operand tile
constexpr int STAGE_LD = 132;            // f32 staging row stride (floats)

__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  // Blocks are dealt round-robin over the 8 XCDs; give each XCD a contiguous run of tiles so that
  // neighbouring tiles (same A row panel) share one L2.  Bijective for any nwg.
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + (bid >> 3);
}

template <typename T, int AMODE, bool A32>
__global__ __launch_bounds__(256, 2) void gemm_kernel(const GemmParams p) {
  typedef typename std::conditional<A32, float, T>::type TA;  // storage type of the A operand
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const As = smem;                    // [2][TILE_BYTES]
  char* const Bs = smem + 2 * TILE_BYTES;   // [2][TILE_BYTES]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int fr = lane & 15, fq = lane >> 4;

  const int wg = xcd_remap(blockIdx.x, gridDim.x);
  const int mt = wg / p.ntiles, nt = wg - mt * p.ntiles;
  const int m0 = mt * BM, n0 = nt * BN;

  const TA* __restrict__ Ag = reinterpret_cast<const TA*>(p.A);
  const T* __restrict__ Wg = reinterpret_cast<const T*>(p.Wt);
  const TA* zero = reinterpret_cast<const TA*>(p.zero);
  const T* zero_w = reinterpret_cast<const T*>(p.zero);

  // ---- loader state: this thread stages chunk `lc` (8 elements) of rows lr + 32*j ----
  const int lc = tid & 7;
  const int lr = tid >> 3;
  const TA* a_base[4];
  const T* b_base[4];
  int a_y[4], a_x[4];  // conv3x3: iy0, ix0 ; convt3: frame, unused
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    int m = m0 + lr + 32 * j;
    if (m > p.M - 1) m = p.M - 1;
    int n = n0 + lr + 32 * j;
    if (n > p.N - 1) n = p.N - 1;
    b_base[j] = Wg + (int64_t)n * p.ldw;
    if (AMODE == A_DENSE) {
      a_base[j] = Ag + (int64_t)m * p.lda;
      a_y[j] = a_x[j] = 0;
    } else if (AMODE == A_CONV3X3) {
      const int hw = p.Ho * p.Wo;
      const int f = m / hw;
      const int rem = m - f * hw;
      const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
      a_base[j] = Ag + (int64_t)f * p.Hin * p.Win * p.lda;
      a_y[j] = oy * p.stride - 1;
      a_x[j] = ox * p.stride - 1;
    } else {
      const int f = m / p.P;
      const int pix = m - f * p.P;
      a_base[j] = Ag + (int64_t)pix * p.lda;  // + frame * P * lda added per tap
      a_y[j] = f;
      a_x[j] = pix;
    }
  }
  // running (tap, channel) of this thread's chunk for the conv loaders
  int tap = 0, ch = lc * 8;
  if (AMODE != A_DENSE) {
    tap = ch / p.Cin;
    ch -= tap * p.Cin;
  }

  u32x4 ra[4], ra_hi[4], rb[4];  // ra_hi: second half of an f32 A chunk (A32 only)
  auto load_tile = [&](int kt) {
    const int k = kt * BK + lc * 8;
    const bool kin = k < p.K;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const TA* src;
      if (AMODE == A_DENSE) {
        src = kin ? a_base[j] + k : zero;
      } else if (AMODE == A_CONV3X3) {
        const int dy = tap / 3, dx = tap - dy * 3;
        int iy = a_y[j] + dy, ix = a_x[j] + dx;
        const bool ok = kin && iy >= 0 && iy < p.Hv && ix >= 0 && ix < p.Wv;
        if (p.ups) {
          iy >>= 1;
          ix >>= 1;
        }
        src = ok ? a_base[j] + ((int64_t)iy * p.Win + ix) * p.lda + ch : zero;
      } else {
        const int sf = a_y[j] + tap - 1;
        if (!kin) {
          src = zero;
        } else if (sf < 0) {
          src = p.halo_lo ? reinterpret_cast<const TA*>(p.halo_lo) + (int64_t)a_x[j] * p.lda + ch
                          : zero;
        } else if (sf >= p.F) {
          src = p.halo_hi ? reinterpret_cast<const TA*>(p.halo_hi) + (int64_t)a_x[j] * p.lda + ch
                          : zero;
        } else {
          src = a_base[j] + (int64_t)sf * p.P * p.lda + ch;
        }
      }
      ra[j] = ld_global16(src);
      if (A32) ra_hi[j] = ld_global16(src + 4);
      rb[j] = ld_global16(kin ? b_base[j] + k : zero_w);
    }
    if (AMODE != A_DENSE) {  // advance to the next K-tile
      ch += BK;
      while (ch >= p.Cin) {
        ch -= p.Cin;
        ++tap;
      }
    }
  };
  auto store_tile = [&](int buf) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int row = lr + 32 * j;
      const int off = row * 128 + ((lc ^ (row & 7)) << 4);
      if (A32) {  // f32 residual-stream operand: round to the MFMA input type while staging
        union { u32x4 u; float f[4]; } lo, hi;
        lo.u = ra[j];
        hi.u = ra_hi[j];
        Pack8<T> cv;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          cv.e[e] = from_f32<T>(lo.f[e]);
          cv.e[e + 4] = from_f32<T>(hi.f[e]);
        }
        *reinterpret_cast<u32x4*>(As + buf * TILE_BYTES + off) = cv.u;
      } else {
        *reinterpret_cast<u32x4*>(As + buf * TILE_BYTES + off) = ra[j];
      }
      *reinterpret_cast<u32x4*>(Bs + buf * TILE_BYTES + off) = rb[j];
    }
  };

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk = (p.K + BK - 1) / BK;
  load_tile(0);
  store_tile(0);
  __syncthreads();

  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nk) load_tile(kt + 1);
    const char* as = As + buf * TILE_BYTES;
    const char* bs = Bs + buf * TILE_BYTES;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      Pack8<T> a[4], b[4];
      const int chunk = ks * 4 + fq;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = wm * 64 + i * 16 + fr;
        a[i].u = *reinterpret_cast<const u32x4*>(as + row * 128 + ((chunk ^ (row & 7)) << 4));
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int row = wn * 64 + j * 16 + fr;
        b[j].u = *reinterpret_cast<const u32x4*>(bs + row * 128 + ((chunk ^ (row & 7)) << 4));
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = mfma16(a[i].v, b[j].v, acc[i][j]);
    }
    if (kt + 1 < nk) store_tile(buf ^ 1);
    __syncthreads();
  }

  // ---------------- epilogue ----------------
  // acc[i][j][r] <-> m = wm*64 + i*16 + 4*fq + r,  n = wn*64 + j*16 + fr
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int n = n0 + wn * 64 + j * 16 + fr;
    const float bv = (p.bias != nullptr && n < p.N) ? p.bias[n] : 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[i][j][r] += bv;
  }
  const bool geglu = (p.act == PM_ACT_GEGLU);
  if (p.act == PM_ACT_SILU) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[i][j][r] = silu_f(acc[i][j][r]);
  } else if (geglu) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int jj = 0; jj < 2; ++jj)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          acc[i][jj][r] = acc[i][2 * jj][r] * gelu_erf_f(acc[i][2 * jj + 1][r]);
  }
  const int tw = geglu ? 64 : 128;          // tile width in output columns
  const int wcols = tw >> 1;                // columns per wave
  const int nblk = geglu ? 2 : 4;           // 16-column blocks per wave
  const int nout = geglu ? (p.N >> 1) : p.N;
  const int nbase = geglu ? (n0 >> 1) : n0;
  float* stage = reinterpret_cast<float*>(smem);  // [64][STAGE_LD]
  T* __restrict__ Cg = reinterpret_cast<T*>(p.C);
  const T* __restrict__ Rg = reinterpret_cast<const T*>(p.R);
  float* __restrict__ Cf = reinterpret_cast<float*>(p.C);          // PM_FLAG_OUT_F32: residual stream
  const float* __restrict__ Rf = reinterpret_cast<const float*>(p.R);
  const bool out32 = p.out32 != 0;
  const int cpr = tw >> 3;         // 8-column chunks per row
  const int rpp = 256 / cpr;       // rows per pass
  const int scol = tid % cpr, srow = tid / cpr;

  for (int half = 0; half < 2; ++half) {
    if (wm == half) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (j < nblk)  // j is a compile-time constant after unrolling (no scratch indexing)
              stage[(i * 16 + 4 * fq + r) * STAGE_LD + wn * wcols + j * 16 + fr] = acc[i][j][r];
    }
    __syncthreads();
    for (int row = srow; row < 64; row += rpp) {
      const int m = m0 + half * 64 + row;
      const int n = nbase + scol * 8;
      if (m < p.M && n < nout) {
        const f32x4 v0 = *reinterpret_cast<const f32x4*>(stage + row * STAGE_LD + scol * 8);
        const f32x4 v1 = *reinterpret_cast<const f32x4*>(stage + row * STAGE_LD + scol * 8 + 4);
        float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
        if (out32) {
          float* cptr = Cf + (int64_t)m * p.ldc + n;
          const bool full = (n + 8 <= nout) && ((p.ldc & 3) == 0);
          if (Rf != nullptr) {
            const float* rptr = Rf + (int64_t)m * p.ldr + n;
            if (full && ((p.ldr & 3) == 0)) {
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
          }
          if (full) {
            *reinterpret_cast<f32x4*>(cptr) = f32x4{v[0], v[1], v[2], v[3]};
            *reinterpret_cast<f32x4*>(cptr + 4) = f32x4{v[4], v[5], v[6], v[7]};
          } else {
            for (int e = 0; e < 8 && n + e < nout; ++e) cptr[e] = v[e];
          }
        } else {
          T* cptr = Cg + (int64_t)m * p.ldc + n;
          const bool full = (n + 8 <= nout) && ((p.ldc & 7) == 0);
          if (Rg != nullptr) {
            const T* rptr = Rg + (int64_t)m * p.ldr + n;
            if (full && ((p.ldr & 7) == 0)) {
              Pack8<T> rv;
              rv.u = ld_global16(rptr);
#pragma unroll
              for (int e = 0; e < 8; ++e) v[e] += to_f32(rv.e[e]);
            } else {
              for (int e = 0; e < 8 && n + e < nout; ++e) v[e] += to_f32(rptr[e]);
            }
          }
          if (full) {
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
    __syncthreads();
  }
}

template <typename T, int AMODE, bool A32> static int launch1(const GemmParams& p, hipStream_t stream) {
  const int mtiles = (p.M + BM - 1) / BM;
  const int grid = mtiles * p.ntiles;
  static bool attr_set = false;  // idempotent; a benign race sets the same value twice
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_kernel<T, AMODE, A32>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 4 * TILE_BYTES);
    attr_set = true;
  }
  hipLaunchKernelGGL((gemm_kernel<T, AMODE, A32>), dim3(grid), dim3(256), 4 * TILE_BYTES, stream, p);
  return check_launch();
}

template <typename T, int AMODE> static int launch(const GemmParams& p, int flags, hipStream_t stream) {
  return (flags & PM_FLAG_A_F32) ? launch1<T, AMODE, true>(p, stream) : launch1<T, AMODE, false>(p, stream);
}

static int check_common(const void* A, const void* W, void* C, int64_t M, int64_t N, int64_t K,
                        int act) {
  if (!A || !W || !C) return PM_E_NULL;
  if (M < 1 || N < 1 || K < 8 || (K & 7)) return PM_E_SHAPE;
  if (M > (1ll << 30) || N > (1ll << 30) || K > (1ll << 30)) return PM_E_SHAPE;
  if (act == PM_ACT_GEGLU && (N % 32) != 0) return PM_E_SHAPE;
  if (act < PM_ACT_NONE || act > PM_ACT_GEGLU) return PM_E_SHAPE;
  return PM_OK;
}

}  // namespace pm

using namespace pm;

extern "C" int pm_gemm(const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias,
                       const void* residual, int64_t ldr, void* C, int64_t ldc, int64_t M,
                       int64_t N, int64_t K, int act, int flags, int dtype, void* stream) {
  int rc = check_common(A, W, C, M, N, K, act);
  if (rc) return rc;
  if ((lda & ((flags & PM_FLAG_A_F32) ? 3 : 7)) || (ldw & 7) || lda < K || ldw < K) return PM_E_SHAPE;
  if ((flags & PM_FLAG_OUT_F32) && act == PM_ACT_GEGLU) return PM_E_SHAPE;
  GemmParams p{};
  p.A = A; p.lda = lda; p.Wt = W; p.ldw = ldw; p.bias = bias; p.R = residual; p.ldr = ldr;
  p.C = C; p.ldc = ldc; p.M = (int)M; p.N = (int)N; p.K = (int)K; p.act = act;
  p.out32 = (flags & PM_FLAG_OUT_F32) ? 1 : 0;
  p.ntiles = (int)((N + BN - 1) / BN);
  p.zero = A;  // dense K tails never occur (K % 8 == 0 and whole chunks only); see kin below
  // A dense K tail (K % 64 != 0) reads chunk-wise: chunks with k >= K take `zero`; any 16 readable
  // bytes would poison the accumulator, so require a real zero source only when a tail exists.
  if (K % BK) return PM_E_SHAPE;  // all Linear layers on the path have K % 64 == 0
  PM_DISPATCH_DTYPE(dtype, T, return (launch<T, A_DENSE>(p, flags, (hipStream_t)stream)));
}

extern "C" int pm_conv2d_3x3(const void* x, int64_t ldx, const void* Wp, const float* bias,
                             const void* residual, int64_t ldr, void* y, int64_t ldy, int64_t F,
                             int64_t H, int64_t W, int64_t Cin, int64_t Cout, int stride,
                             int upsample2x, const void* zero_page, int flags, int dtype, void* stream) {
  if (!zero_page) return PM_E_NULL;
  if (stride != 1 && stride != 2) return PM_E_SHAPE;
  if (upsample2x && stride != 1) return PM_E_SHAPE;
  if ((Cin & 7) || (ldx & ((flags & PM_FLAG_A_F32) ? 3 : 7)) || ldx < Cin) return PM_E_SHAPE;
  const int64_t Hv = upsample2x ? 2 * H : H, Wv = upsample2x ? 2 * W : W;
  const int64_t Ho = (Hv + stride - 1) / stride, Wo = (Wv + stride - 1) / stride;
  const int64_t M = F * Ho * Wo, K = 9 * Cin;
  int rc = check_common(x, Wp, y, M, Cout, K, PM_ACT_NONE);
  if (rc) return rc;
  GemmParams p{};
  p.A = x; p.lda = ldx; p.Wt = Wp; p.ldw = K; p.bias = bias; p.R = residual; p.ldr = ldr;
  p.C = y; p.ldc = ldy; p.M = (int)M; p.N = (int)Cout; p.K = (int)K; p.act = PM_ACT_NONE;
  p.out32 = (flags & PM_FLAG_OUT_F32) ? 1 : 0;
  p.ntiles = (int)((Cout + BN - 1) / BN);
  p.Hin = (int)H; p.Win = (int)W; p.Hv = (int)Hv; p.Wv = (int)Wv; p.Cin = (int)Cin;
  p.Ho = (int)Ho; p.Wo = (int)Wo; p.stride = stride; p.ups = upsample2x ? 1 : 0;
  p.zero = zero_page;
  PM_DISPATCH_DTYPE(dtype, T, return (launch<T, A_CONV3X3>(p, flags, (hipStream_t)stream)));
}

extern "C" int pm_conv_temporal_k3(const void* x, int64_t ldx, const void* halo_lo,
                                   const void* halo_hi, const void* Wp, const float* bias,
                                   const void* residual, int64_t ldr, void* y, int64_t ldy,
                                   int64_t F, int64_t P, int64_t Cin, int64_t Cout,
                                   const void* zero_page, int flags, int dtype, void* stream) {
  if (!zero_page) return PM_E_NULL;
  if ((Cin & 7) || (ldx & ((flags & PM_FLAG_A_F32) ? 3 : 7)) || ldx < Cin) return PM_E_SHAPE;
  const int64_t M = F * P, K = 3 * Cin;
  int rc = check_common(x, Wp, y, M, Cout, K, PM_ACT_NONE);
  if (rc) return rc;
  GemmParams p{};
  p.A = x; p.lda = ldx; p.Wt = Wp; p.ldw = K; p.bias = bias; p.R = residual; p.ldr = ldr;
  p.C = y; p.ldc = ldy; p.M = (int)M; p.N = (int)Cout; p.K = (int)K; p.act = PM_ACT_NONE;
  p.out32 = (flags & PM_FLAG_OUT_F32) ? 1 : 0;
  p.ntiles = (int)((Cout + BN - 1) / BN);
  p.Cin = (int)Cin; p.F = (int)F; p.P = (int)P; p.halo_lo = halo_lo; p.halo_hi = halo_hi;
  p.zero = zero_page;
  PM_DISPATCH_DTYPE(dtype, T, return (launch<T, A_CONVT3>(p, flags, (hipStream_t)stream)));
}
