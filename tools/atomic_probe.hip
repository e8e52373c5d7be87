// Dev probe: throughput of returning atomicAdd on ~35 k counters hit by 1 M threads in random order (the bin_count
// pattern): device-scope atomics on one shared array vs workgroup-scope atomics on a per-XCD copy selected by XCC_ID.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
__device__ __forceinline__ int xcc_id() {
  int v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  return v & 7;
}
template <int MODE>
__global__ __launch_bounds__(256) void k(const int* __restrict__ target, int n, int ncounters, int* __restrict__ cnt,
                                         int* __restrict__ pos) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int t = target[i];
  int p;
  if (MODE == 0) p = atomicAdd(&cnt[t], 1);
  if (MODE == 1) p = __hip_atomic_fetch_add(&cnt[xcc_id() * ncounters + t], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  if (MODE == 2) p = __hip_atomic_fetch_add(&cnt[xcc_id() * ncounters + t], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (MODE == 3) p = __hip_atomic_fetch_add(&cnt[t], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);  // (incoherent: rate only)
  pos[i] = p;
}
int main() {
  const int n = 1000000, nc = 2200 * 16;
  std::vector<int> h(n);
  srand(1);
  for (auto& v : h) v = (rand() % 2200) * 16 + (rand() % 14);
  int *t, *cnt, *pos;
  hipMalloc(&t, n * 4); hipMalloc(&cnt, 8 * nc * 4); hipMalloc(&pos, n * 4);
  hipMemcpy(t, h.data(), n * 4, hipMemcpyHostToDevice);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int mode = 0; mode < 4; ++mode) {
    float best = 1e9f;
    for (int rep = 0; rep < 6; ++rep) {
      hipMemset(cnt, 0, 8 * nc * 4);
      hipDeviceSynchronize();
      hipEventRecord(a);
      if (mode == 0) hipLaunchKernelGGL(k<0>, dim3((n + 255) / 256), dim3(256), 0, 0, t, n, nc, cnt, pos);
      if (mode == 1) hipLaunchKernelGGL(k<1>, dim3((n + 255) / 256), dim3(256), 0, 0, t, n, nc, cnt, pos);
      if (mode == 2) hipLaunchKernelGGL(k<2>, dim3((n + 255) / 256), dim3(256), 0, 0, t, n, nc, cnt, pos);
      if (mode == 3) hipLaunchKernelGGL(k<3>, dim3((n + 255) / 256), dim3(256), 0, 0, t, n, nc, cnt, pos);
      hipEventRecord(b); hipEventSynchronize(b);
      float ms; hipEventElapsedTime(&ms, a, b);
      if (ms < best) best = ms;
    }
    // correctness of the sharded modes: per counter, the sum over the 8 copies must equal the number of hits
    std::vector<int> c(8 * nc);
    hipMemcpy(c.data(), cnt, 8 * nc * 4, hipMemcpyDeviceToHost);
    long total = 0;
    for (int v : c) total += v;
    printf("mode %d: %.1f us, sum of counters %ld (expect %d)\n", mode, best * 1e3f, total, n);
  }
  return 0;
}
