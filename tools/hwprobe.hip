// Hardware-semantics probe for gfx950 (dev tool, not product code).
// Prints (1) MFMA 32x32x16 bf16 / 16x16x32 layout check vs. the model used by the conv kernels,
// (2) raw lane->source mapping of ds_read_tr16_b64, (3) global_load_lds placement.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#include <cmath>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__device__ __host__ inline uint16_t f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); return (uint16_t)(u >> 16); }

// A: [32][16] row-major floats (small ints), B: [16][32]; C out [32][32]
__global__ void mfma32(const float* A, const float* B, float* C) {
  int l = threadIdx.x, h = l >> 5, i = l & 31;
  bf16x8 a, b;
  for (int j = 0; j < 8; ++j) {
    a[j] = (__bf16)A[i * 16 + 8 * h + j];       // A[m=i][k=8h+j]
    b[j] = (__bf16)B[(8 * h + j) * 32 + i];     // B[k=8h+j][n=i]
  }
  f32x16 c = {0};
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  for (int r = 0; r < 16; ++r) {
    int row = (r & 3) + 8 * (r >> 2) + 4 * h;
    C[row * 32 + i] = c[r];
  }
}
// 16x16x32: A [16][32], B [32][16], C [16][16]
__global__ void mfma16(const float* A, const float* B, float* C) {
  int l = threadIdx.x, q = l >> 4, i = l & 15;
  bf16x8 a, b;
  for (int j = 0; j < 8; ++j) {
    a[j] = (__bf16)A[i * 32 + 8 * q + j];       // A[m=i][k=8q+j]
    b[j] = (__bf16)B[(8 * q + j) * 16 + i];     // B[k=8q+j][n=i]
  }
  f32x4 c = {0};
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) C[(4 * q + r) * 16 + i] = c[r];
}
// fp32 mfma 32x32x2: A [32][2], B [2][32]
__global__ void mfma32f(const float* A, const float* B, float* C) {
  int l = threadIdx.x, h = l >> 5, i = l & 31;
  f32x16 c = {0};
  c = __builtin_amdgcn_mfma_f32_32x32x2f32(A[i * 2 + h], B[h * 32 + i], c, 0, 0, 0);
  for (int r = 0; r < 16; ++r) C[((r & 3) + 8 * (r >> 2) + 4 * h) * 32 + i] = c[r];
}

// tr16 probe: lds[i] = i (u16). mode 0: lane address = lane*8 bytes. mode 1: model addresses for a
// row-major [P][C] tile (C = 64 shorts/row): block for 16-lane group g: rows 8*(g>>1)+(i>>2), cols 16*(g&1)+4*(i&3)
__global__ void trprobe(int mode, int* out) {
  __shared__ __attribute__((aligned(16))) short lds[4096];
  int l = threadIdx.x;
  for (int i = l; i < 4096; i += 64) lds[i] = (short)i;
  __syncthreads();
  int off;
  if (mode == 0) off = l * 4;  // in shorts
  else { int g = l >> 4, i = l & 15; off = (8 * (g >> 1) + (i >> 2)) * 64 + 16 * (g & 1) + 4 * (i & 3); }
  s16x4 r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(lds + off));
  for (int j = 0; j < 4; ++j) out[l * 4 + j] = r[j];
}

// global_load_lds probe: each lane loads 16B from src + perm(lane)*16 ; LDS dest base uniform
__global__ void gldsprobe(const int* src, int* out) {
  __shared__ __attribute__((aligned(16))) int lds[512];
  int l = threadIdx.x;
  for (int i = l; i < 512; i += 64) lds[i] = -1;
  __syncthreads();
  const int* p = src + ((l * 7) & 63) * 4;  // permuted source chunk
  __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)p,
                                   (void __attribute__((address_space(3)))*)(lds + 64), 16, 0, 0);
  __builtin_amdgcn_s_waitcnt(0);
  __syncthreads();
  for (int i = l; i < 512; i += 64) out[i] = lds[i];
}

int main() {
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  printf("device %s arch %s CUs %d LDS/blk %zu clock %d kHz memclk %d buswidth %d L2 %d\n", prop.name, prop.gcnArchName,
         prop.multiProcessorCount, prop.sharedMemPerBlock, prop.clockRate, prop.memoryClockRate, prop.memoryBusWidth, prop.l2CacheSize);
  srand(1);
  auto run_mm = [&](int M, int N, int K, void (*kern)(const float*, const float*, float*), const char* name) {
    std::vector<float> A(M * K), B(K * N), C(M * N), R(M * N, 0.f);
    for (auto& v : A) v = (float)(rand() % 7 - 3);
    for (auto& v : B) v = (float)(rand() % 5 - 2);
    for (int m = 0; m < M; ++m) for (int n = 0; n < N; ++n) { float s = 0; for (int k = 0; k < K; ++k) s += A[m * K + k] * B[k * N + n]; R[m * N + n] = s; }
    float *dA, *dB, *dC; CK(hipMalloc(&dA, A.size() * 4)); CK(hipMalloc(&dB, B.size() * 4)); CK(hipMalloc(&dC, C.size() * 4));
    CK(hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(kern, dim3(1), dim3(64), 0, 0, dA, dB, dC); CK(hipDeviceSynchronize());
    CK(hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost));
    int bad = 0; for (size_t i = 0; i < C.size(); ++i) if (C[i] != R[i]) ++bad;
    printf("%s layout model: %s (%d mismatches)\n", name, bad ? "FAIL" : "PASS", bad);
  };
  run_mm(32, 32, 16, mfma32, "mfma_f32_32x32x16_bf16");
  run_mm(16, 16, 32, mfma16, "mfma_f32_16x16x32_bf16");
  run_mm(32, 32, 2, mfma32f, "mfma_f32_32x32x2f32");
  int* dout; CK(hipMalloc(&dout, 4096 * 4)); std::vector<int> out(4096);
  for (int mode = 0; mode < 2; ++mode) {
    hipLaunchKernelGGL(trprobe, dim3(1), dim3(64), 0, 0, mode, dout); CK(hipDeviceSynchronize());
    CK(hipMemcpy(out.data(), dout, 256 * 4, hipMemcpyDeviceToHost));
    printf("tr16 mode %d:\n", mode);
    for (int l = 0; l < 64; ++l) printf("  lane %2d: %4d %4d %4d %4d\n", l, out[l * 4], out[l * 4 + 1], out[l * 4 + 2], out[l * 4 + 3]);
    if (mode == 1) {  // model check: lane (h=l>>5, m=l&31) read t=0 should hold X[p=8h+j][c=m], j<4 => value p*64+c
      int bad = 0;
      for (int l = 0; l < 64; ++l) for (int j = 0; j < 4; ++j) { int h = l >> 5, m = l & 31; if (out[l * 4 + j] != (8 * h + j) * 64 + m) ++bad; }
      printf("tr16 A-fragment model: %s (%d)\n", bad ? "FAIL" : "PASS", bad);
    }
  }
  std::vector<int> src(256); for (int i = 0; i < 256; ++i) src[i] = i; int* dsrc; CK(hipMalloc(&dsrc, 1024));
  CK(hipMemcpy(dsrc, src.data(), 1024, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(gldsprobe, dim3(1), dim3(64), 0, 0, dsrc, dout); CK(hipDeviceSynchronize());
  CK(hipMemcpy(out.data(), dout, 512 * 4, hipMemcpyDeviceToHost));
  int bad = 0;
  for (int l = 0; l < 64; ++l) for (int j = 0; j < 4; ++j) if (out[64 + l * 4 + j] != ((l * 7) & 63) * 4 + j) ++bad;
  for (int i = 0; i < 64; ++i) if (out[i] != -1) ++bad;
  printf("global_load_lds lane-linear model: %s (%d)\n", bad ? "FAIL" : "PASS", bad);
  return 0;
}
