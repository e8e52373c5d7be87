// Dev probe: throughput of fire-and-forget LDS float atomics (ds_add_f32) in the access pattern the pair-list
// convolution kernel uses (32 consecutive dwords per half-wave, two different rows per instruction).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, const int* rows, int iters) {
  extern __shared__ float acc[];
  for (int i = threadIdx.x; i < 256 * 128; i += blockDim.x) acc[i] = 0.f;
  __syncthreads();
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  float v2 = 0.f;
  for (int it = 0; it < iters; ++it) {
    const int r = (it * 37 + (lane >> 5) * 101) & 255;
    const float v = (float)it + v2;
    if (MODE == 3 || MODE == 4) {
      float* base = &acc[w * 32 + (lane & 31)];
      float t[16];
#pragma unroll
      for (int q = 0; q < 16; ++q) t[q] = base[((r + 7 * q) & 255) * 128];
      if (MODE == 3) {
#pragma unroll
        for (int q = 0; q < 16; ++q) base[((r + 7 * q) & 255) * 128] = t[q] + v;
      } else {
#pragma unroll
        for (int q = 0; q < 16; ++q) v2 += t[q];
      }
      continue;
    }
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      float* p = &acc[((r + 7 * q) & 255) * 128 + w * 32 + (lane & 31)];
      if (MODE == 0) __builtin_amdgcn_ds_faddf((__attribute__((address_space(3))) float*)p, v, 0, 0, false);
      if (MODE == 1) *p += v;
      if (MODE == 2) *p = v;
    }
  }
  __syncthreads();
  if (blockIdx.x == 0)
    for (int i = threadIdx.x; i < 256 * 128; i += blockDim.x) out[i] = acc[i];
}
template <int MODE> void run(const char* name, float* out, int* rows) {
  const int iters = 20000;
  hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  k<MODE><<<256, 256, 128 * 1024>>>(out, rows, 100);
  hipEventRecord(a);
  k<MODE><<<256, 256, 128 * 1024>>>(out, rows, iters);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  // per CU: 4 waves * iters * 16 wave-instructions
  const double instr = 4.0 * iters * 16;
  printf("%s: %.3f ms, %.2f ns per wave-instr per CU (%.1f clk @2.4GHz), LDS update rate %.1f B/clk/CU\n", name, ms,
         ms * 1e6 / instr, ms * 1e6 / instr * 2.4, 256.0 / (ms * 1e6 / instr * 2.4));
}
int main() {
  float* out; int* rows;
  hipMalloc(&out, 256 * 128 * 4); hipMalloc(&rows, 2048 * 4);
  std::vector<int> h(2048);
  for (int i = 0; i < 2048; ++i) h[i] = (i * 2654435761u >> 8) & 255;
  hipMemcpy(rows, h.data(), 2048 * 4, hipMemcpyHostToDevice);
  run<0>("ds_add_f32", out, rows);
  run<1>("read+add+write", out, rows);
  run<2>("ds_write_b32", out, rows);
  run<3>("batched 16 reads then 16 add+writes", out, rows);
  run<4>("16 reads only", out, rows);
  return 0;
}
