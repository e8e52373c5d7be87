// Dev probe (round 2): cost of the primitives the cell-table kernel-map builder is made of, 1 M voxels / 15.6 k blocks of
// 8^3 cells (32 MB table): plain 4-B stores, non-returning / returning atomicMin to random cells, an empty-kernel
// launch train, and dword loads whose 64 lanes touch 64 / 12 / 2 distinct cache lines (texture-addresser cost model).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <int MODE>
__global__ __launch_bounds__(256) void scatter_k(const unsigned* __restrict__ cell, int n, unsigned* __restrict__ table,
                                                 unsigned* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const unsigned c = cell[i];
  if (MODE == 0) table[c] = (unsigned)i;
  if (MODE == 1) __hip_atomic_fetch_min(&table[c], (unsigned)i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (MODE == 2) out[i] = __hip_atomic_fetch_min(&table[c], (unsigned)i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (MODE == 3) out[i] = table[c];  // random 4-B read
  if (MODE == 4) out[i] = __hip_atomic_fetch_add(&table[c], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__global__ void empty_k(int* p) {
  if (p && threadIdx.x == 12345) *p = 0;
}

// every wave issues `iters` dword loads; the 64 lanes of one instruction touch `lines` distinct 128-B lines
__global__ __launch_bounds__(256) void lines_k(const unsigned* __restrict__ src, int words, int lines, int iters,
                                               unsigned* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const unsigned wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  unsigned acc = 0;
  unsigned base = (wave * 2654435761u) % (unsigned)(words - 64 * 32 * 2);
  for (int it = 0; it < iters; ++it) {
    // lane -> line (lane % lines), word inside the line (lane / lines)
    const unsigned idx = (base + (unsigned)(lane % lines) * 32u * 2u + (unsigned)(lane / lines)) % (unsigned)words;
    acc += src[idx];
    base = (base * 1664525u + 1013904223u) % (unsigned)(words - 64 * 32 * 2);
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

int main() {
  const int n = 1000000, nblk = 15625, cells = nblk * 512;
  std::vector<unsigned> h(n);
  srand(1);
  for (auto& v : h) v = (unsigned)((rand() % nblk) * 512 + (rand() % 512));
  unsigned *cell, *table, *out;
  hipMalloc(&cell, n * 4); hipMalloc(&table, (size_t)cells * 4); hipMalloc(&out, (size_t)4 << 20);
  hipMemcpy(cell, h.data(), n * 4, hipMemcpyHostToDevice);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  const char* names[] = {"plain store", "atomicMin (no return)", "atomicMin (returning)", "random 4-B read", "atomicAdd (returning)"};
  for (int mode = 0; mode < 5; ++mode) {
    float best = 1e9f;
    for (int rep = 0; rep < 6; ++rep) {
      hipMemset(table, 0xFF, (size_t)cells * 4);
      hipDeviceSynchronize();
      hipEventRecord(a);
      const dim3 g((n + 255) / 256), t(256);
      if (mode == 0) hipLaunchKernelGGL(scatter_k<0>, g, t, 0, 0, cell, n, table, out);
      if (mode == 1) hipLaunchKernelGGL(scatter_k<1>, g, t, 0, 0, cell, n, table, out);
      if (mode == 2) hipLaunchKernelGGL(scatter_k<2>, g, t, 0, 0, cell, n, table, out);
      if (mode == 3) hipLaunchKernelGGL(scatter_k<3>, g, t, 0, 0, cell, n, table, out);
      if (mode == 4) hipLaunchKernelGGL(scatter_k<4>, g, t, 0, 0, cell, n, table, out);
      hipEventRecord(b); hipEventSynchronize(b);
      float ms; hipEventElapsedTime(&ms, a, b);
      if (ms < best) best = ms;
    }
    printf("%-26s 1M into 32 MB: %.1f us\n", names[mode], best * 1e3f);
  }
  {  // memset 32 MB
    float best = 1e9f;
    for (int rep = 0; rep < 6; ++rep) {
      hipDeviceSynchronize();
      hipEventRecord(a);
      hipMemsetAsync(table, 0xFF, (size_t)cells * 4, 0);
      hipEventRecord(b); hipEventSynchronize(b);
      float ms; hipEventElapsedTime(&ms, a, b);
      if (ms < best) best = ms;
    }
    printf("memset 32 MB: %.1f us\n", best * 1e3f);
  }
  for (int train : {1, 10, 40}) {
    float best = 1e9f;
    for (int rep = 0; rep < 6; ++rep) {
      hipDeviceSynchronize();
      hipEventRecord(a);
      for (int i = 0; i < train; ++i) hipLaunchKernelGGL(empty_k, dim3(1), dim3(64), 0, 0, (int*)nullptr);
      hipEventRecord(b); hipEventSynchronize(b);
      float ms; hipEventElapsedTime(&ms, a, b);
      if (ms < best) best = ms;
    }
    printf("empty kernel x%d: %.2f us per launch\n", train, best * 1e3f / train);
  }
  for (int train : {10}) {  // wide empty kernel (1M threads)
    float best = 1e9f;
    for (int rep = 0; rep < 6; ++rep) {
      hipDeviceSynchronize();
      hipEventRecord(a);
      for (int i = 0; i < train; ++i) hipLaunchKernelGGL(empty_k, dim3(3907), dim3(256), 0, 0, (int*)nullptr);
      hipEventRecord(b); hipEventSynchronize(b);
      float ms; hipEventElapsedTime(&ms, a, b);
      if (ms < best) best = ms;
    }
    printf("empty 1M-thread kernel x%d: %.2f us per launch\n", train, best * 1e3f / train);
  }
  // line-count model: 4 MB source (L2 resident), 2048 workgroups x 4 waves x 64 loads
  const int words = 1 << 20;
  for (int lines : {64, 32, 16, 12, 8, 4, 2, 1}) {
    float best = 1e9f;
    const int wgs = 2048, iters = 64;
    for (int rep = 0; rep < 5; ++rep) {
      hipDeviceSynchronize();
      hipEventRecord(a);
      hipLaunchKernelGGL(lines_k, dim3(wgs), dim3(256), 0, 0, (const unsigned*)table, words, lines, iters, out);
      hipEventRecord(b); hipEventSynchronize(b);
      float ms; hipEventElapsedTime(&ms, a, b);
      if (ms < best) best = ms;
    }
    const double instrs = (double)wgs * 4 * iters;
    printf("dword load, %2d lines per wave instruction: %.1f us, %.1f clk per instruction per CU (2.4 GHz, 256 CUs)\n", lines,
           best * 1e3f, best * 1e-3 * 2.4e9 * 256 / instrs);
  }
  return 0;
}
