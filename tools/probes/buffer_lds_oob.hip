// Probe: does an out-of-range lane of `buffer_load_dwordx4 ... lds` write ZEROS into LDS (or skip the write)?
// hipcc --offload-arch=gfx950 -O2 tools/probes/buffer_lds_oob.hip -o /tmp/oob && /tmp/oob
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef __attribute__((address_space(3))) void lds_void;
__global__ void k(const char* x, unsigned nbytes, float* out, int shift) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* sf = reinterpret_cast<float*>(smem);
  for (int i = threadIdx.x; i < 64 * 4; i += 64) sf[i] = -7.0f;  // stale pattern
  __syncthreads();
  __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(x + shift), (short)0, (int)nbytes, 0x00020000);
  unsigned voff = threadIdx.x * 16;
  if (threadIdx.x & 1) voff = 0xfffffff0u;       // odd lanes: far out of range
  if (threadIdx.x == 62) voff = nbytes - 8;       // straddles the end
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void*)smem, 16, voff, 0, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int e = 0; e < 4; ++e) out[threadIdx.x * 4 + e] = sf[threadIdx.x * 4 + e];
}
int main() {
  const int n = 64 * 4;
  std::vector<float> h(n);
  for (int i = 0; i < n; ++i) h[i] = 1.0f + i;
  float *dx, *dout;
  hipMalloc(&dx, n * 4 + 256); hipMalloc(&dout, n * 4);
  hipMemcpy(dx, h.data(), n * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 64 * 16, 0, (const char*)dx, (unsigned)(n * 4), dout, 0);
  std::vector<float> o(n);
  hipMemcpy(o.data(), dout, n * 4, hipMemcpyDeviceToHost);
  for (int l : {0, 1, 2, 3, 61, 62, 63}) printf("lane %2d: %g %g %g %g\n", l, o[l * 4], o[l * 4 + 1], o[l * 4 + 2], o[l * 4 + 3]);
  return 0;
}
