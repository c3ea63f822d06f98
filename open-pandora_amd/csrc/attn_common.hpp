// Shared definitions of the head-dim-64 attention kernels (attn.hip: 32x32x16 MFMA forms; attn16.hip: the 16x16x32 form).
#pragma once
#include <stdlib.h>
#include <math.h>
#include <type_traits>
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
  int Nq, heads, nqt;
  float scale_log2e;
  int prescaled;  // attn_self_kernel: q already carries scale * log2(e) (scale_log2e == 1)
  // diagnostics build only (pm_debug_attn_stamps, include/pandora_mi355x_diag.h; nullptr otherwise): per workgroup
  // {d s_memtime, d s_memrealtime} over the kernel body - the clock the chip holds under this instruction stream
  unsigned long long* stamps;
};

int launch_attn_self16(const AttnParams& p, int dtype, dim3 grid, hipStream_t stream, int form = 0);  // attn16.hip

constexpr int KV_TILE = 64;
constexpr int KV_TILE_BYTES = KV_TILE * 128;  // 64 keys x 64 dims x 2 B

typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;

// value held by lane ^ 32 (v_permlane32_swap: no LDS round trip)
__device__ __forceinline__ float other_half(float v) {
  const unsigned u = __float_as_uint(v);
  auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  return __uint_as_float((threadIdx.x & 32) ? r[0] : r[1]);
}

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

}  // namespace pm
