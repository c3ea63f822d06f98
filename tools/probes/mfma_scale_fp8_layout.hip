// Operand layout probe for v_mfma_scale_f32_32x32x64_f8f6f4 with fp8 (e4m3) A and B and unit scales (E8M0 127):
// which k does element `idx` (= 4 * register + byte, 0..31) of lane l hold?  Exact small-integer data; prints the
// hypothesis that reproduces C = A . B.   build: hipcc --offload-arch=gfx950 -O2 mfma_scale_fp8_layout.hip -o probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
typedef __attribute__((ext_vector_type(8))) int v8i;
typedef __attribute__((ext_vector_type(16))) float v16f;
__global__ void k(const int* a, const int* b, float* c) {
  v8i A, B; v16f C = {0};
  for (int i = 0; i < 8; ++i) { A[i] = a[threadIdx.x * 8 + i]; B[i] = b[threadIdx.x * 8 + i]; }
  C = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A, B, C, 0, 0, 0, 127, 0, 127);
  for (int i = 0; i < 16; ++i) c[threadIdx.x * 16 + i] = C[i];
}
static unsigned char enc(int v) {  // e4m3 of a small integer in [-3, 3]
  static const unsigned char t[4] = {0x00, 0x38, 0x40, 0x44};
  return v < 0 ? (unsigned char)(t[-v] | 0x80) : t[v];
}
static int kmap(int hyp, int h, int idx) {
  switch (hyp) {
    case 0: return 32 * h + idx;                          // 32 consecutive k per lane half
    case 1: return 16 * h + (idx & 15) + 32 * (idx >> 4); // two K=32 halves, 16 per lane half each
    case 2: return 8 * h + (idx & 7) + 16 * (idx >> 3);   // four K=16 quarters
    default: return 4 * h + (idx & 3) + 8 * (idx >> 2);
  }
}
int main() {
  int A[32][64], B[64][32];
  srand(7);
  for (int i = 0; i < 32; ++i) for (int q = 0; q < 64; ++q) A[i][q] = rand() % 7 - 3;
  for (int q = 0; q < 64; ++q) for (int j = 0; j < 32; ++j) B[q][j] = (rand() % 7 - 3) * ((q * 5 + j) % 3 != 0);
  float want[32][32];
  for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) { int s = 0; for (int q = 0; q < 64; ++q) s += A[i][q] * B[q][j]; want[i][j] = (float)s; }
  int *da, *db; float* dc;
  hipMalloc(&da, 64 * 32); hipMalloc(&db, 64 * 32); hipMalloc(&dc, 64 * 16 * 4);
  for (int hyp = 0; hyp < 4; ++hyp) {
    unsigned char ha[64][32], hb[64][32];
    for (int l = 0; l < 64; ++l) for (int idx = 0; idx < 32; ++idx) {
      const int kk = kmap(hyp, l >> 5, idx);
      ha[l][idx] = enc(A[l & 31][kk]);
      hb[l][idx] = enc(B[kk][l & 31]);
    }
    hipMemcpy(da, ha, sizeof(ha), hipMemcpyHostToDevice); hipMemcpy(db, hb, sizeof(hb), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, da, db, dc);
    float hc[64][16];
    hipMemcpy(hc, dc, sizeof(hc), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l) for (int r = 0; r < 16; ++r) {
      const int i = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), j = l & 31;
      bad += hc[l][r] != want[i][j];
    }
    printf("hypothesis %d: %d / 1024 mismatches%s\n", hyp, bad, bad ? "" : "   <== operand layout");
  }
  return 0;
}
