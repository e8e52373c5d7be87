// Dev probe: per-CU ingest rate of L2-resident data (a 442 KB weight image re-read by every workgroup), through
// LDS-DMA (global_load_lds_dwordx4) vs plain 16-B global loads to registers.
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../warpconvnet_amd/csrc/wcn_common.h"
using namespace wcn;
typedef __attribute__((ext_vector_type(4))) float f4;
template <int MODE>
__global__ __launch_bounds__(256) void k(const char* __restrict__ w, int slabs, int iters, float* out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  f4 acc = {0, 0, 0, 0};
  const uint32_t base = lds_addr_of(smem);
  for (int it = 0; it < iters; ++it) {
    const char* src = w + (size_t)((it * 7 + blockIdx.x) % slabs) * 16384;
    if (MODE == 0) {
#pragma unroll
      for (int u = 0; u < 4; ++u)
        glds16(src + (u * 4 + wave) * 1024 + lane * 16, __builtin_amdgcn_readfirstlane(base + ((it & 1) * 16384) + (u * 4 + wave) * 1024));
      if ((it & 3) == 3) wait_vmcnt<0>();
    } else {
      f4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const f4*>(src + (u * 4 + wave) * 1024 + lane * 16);
#pragma unroll
      for (int u = 0; u < 4; ++u) acc += v[u];
    }
  }
  wait_vmcnt<0>();
  __syncthreads();
  if (MODE == 0) acc.x = reinterpret_cast<float*>(smem)[threadIdx.x];
  out[blockIdx.x * 256 + threadIdx.x] = acc.x + acc.y + acc.z + acc.w;
}
template <int MODE> void run(const char* name, const char* w, float* out, int wgs_per_cu, int lds) {
  const int iters = 2000, grid = 256 * wgs_per_cu;
  hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  k<MODE><<<grid, 256, lds>>>(w, 27, 10, out);
  hipEventRecord(a);
  k<MODE><<<grid, 256, lds>>>(w, 27, iters, out);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  const double bytes_per_cu = (double)wgs_per_cu * iters * 16384;
  printf("%-28s %d WG/CU: %.3f ms  %.1f GB/s per CU (%.1f B/clk @2.1GHz)  %.1f TB/s chip\n", name, wgs_per_cu, ms,
         bytes_per_cu / ms / 1e6, bytes_per_cu / ms / 1e6 / 2.1, bytes_per_cu * 256 / ms / 1e9);
}
int main() {
  char* w; float* out;
  hipMalloc(&w, 27 * 16384); hipMalloc(&out, 1 << 22);
  hipMemset(w, 0, 27 * 16384);
  for (int n : {1, 2, 4}) {
    run<0>("LDS-DMA, L2-resident slab", w, out, n, 32768);
    run<1>("global_load to registers", w, out, n, 32768);
  }
  return 0;
}
