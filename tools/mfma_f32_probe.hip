// Dev probe: operand / result layout of v_mfma_f32_32x32x2_f32 as csrc/pointconv.hip assumes it.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(16))) float f32x16;
__global__ void k(float* out) {
  const int l = threadIdx.x, i = l & 31, kk = l >> 5;
  // A[i][k] = (i + 1) * (k ? 1000 : 1), B[k][j] = (j + 1) * (k ? 7 : 3)  ->  D[i][j] = (i+1)(j+1)(3 + 7000)
  const float a = (float)(i + 1) * (kk ? 1000.f : 1.f), b = (float)(i + 1) * (kk ? 7.f : 3.f);
  f32x16 c = {0};
  c = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
  for (int r = 0; r < 16; ++r) out[l * 16 + r] = c[r];
}
int main() {
  float* d; hipMalloc(&d, 64 * 16 * 4);
  k<<<1, 64>>>(d);
  float h[64 * 16]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int l = 0; l < 64; ++l)
    for (int r = 0; r < 16; ++r) {
      const int i = 8 * (r / 4) + 4 * (l / 32) + r % 4, j = l % 32;
      const float want = (float)(i + 1) * (float)(j + 1) * 7003.f;
      if (h[l * 16 + r] != want) { if (bad < 5) printf("lane %d reg %d: got %g want %g\n", l, r, h[l * 16 + r], want); ++bad; }
    }
  printf("mfma_f32_32x32x2 layout %s (%d mismatches)\n", bad ? "DIFFERS" : "as assumed", bad);
  return 0;
}
