// Dev probe: achievable random-row gather bandwidth on MI355X for 128-B rows out of a 128 MB array, comparing the
// MFMA-fragment-shaped access (lane = (half,row), 4 x 16 B per lane) with line-coalesced access (8 lanes per row).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#include <random>
typedef __attribute__((ext_vector_type(4))) float f4;
// MODE 0: fragment-shaped: lane (h = l>>5, n = l&31) reads 4 x 16 B = its half (64 B) of row idx[n]
// MODE 1: line-coalesced: one instruction reads 8 rows, 8 lanes x 16 B each
template <int MODE, int UNROLL>
__global__ __launch_bounds__(256) void gather(const char* __restrict__ in, const int* __restrict__ idx, long npairs,
                                              float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const long nwaves = (long)gridDim.x * 4;
  f4 acc = {0, 0, 0, 0};
  // each wave-iteration consumes 32*UNROLL pairs
  for (long base = wave * 32 * UNROLL; base + 32 * UNROLL <= npairs; base += nwaves * 32 * UNROLL) {
    f4 v[UNROLL][4];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      if (MODE == 0) {
        const int r = idx[base + u * 32 + (lane & 31)];
        const f4* p = reinterpret_cast<const f4*>(in + (long)r * 128 + (lane >> 5) * 64);
#pragma unroll
        for (int s = 0; s < 4; ++s) v[u][s] = p[s];
      } else if (MODE == 2) {  // 16x16x32 operand order: lane (g, n) takes piece g of each half of row n (4 lanes, not adjacent)
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const int r = idx[base + u * 32 + (s >> 1) * 16 + (lane & 15)];
          v[u][s] = *reinterpret_cast<const f4*>(in + (long)r * 128 + (s & 1) * 64 + (lane >> 4) * 16);
        }
      } else if (MODE == 3) {  // row-shaped 64 B: 4 ADJACENT lanes cover half a row, 16 rows per instruction
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const int r = idx[base + u * 32 + (s >> 1) * 16 + (lane >> 2)];
          v[u][s] = *reinterpret_cast<const f4*>(in + (long)r * 128 + (s & 1) * 64 + (lane & 3) * 16);
        }
      } else {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const int r = idx[base + u * 32 + s * 8 + (lane >> 3)];
          v[u][s] = *reinterpret_cast<const f4*>(in + (long)r * 128 + (lane & 7) * 16);
        }
      }
    }
#pragma unroll
    for (int u = 0; u < UNROLL; ++u)
#pragma unroll
      for (int s = 0; s < 4; ++s) acc += v[u][s];
  }
  out[(long)blockIdx.x * 256 + threadIdx.x] = acc.x + acc.y + acc.z + acc.w;
}
template <int MODE, int UNROLL> void run(const char* name, const char* in, const int* idx, long npairs, float* out, int grid) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  gather<MODE, UNROLL><<<grid, 256>>>(in, idx, npairs, out);
  hipEventRecord(a);
  for (int i = 0; i < 5; ++i) gather<MODE, UNROLL><<<grid, 256>>>(in, idx, npairs, out);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b); ms /= 5;
  printf("%-44s grid %5d: %7.1f us  %6.2f TB/s\n", name, grid, ms * 1e3, npairs * 128.0 / ms / 1e9);
}
int main() {
  const long nrows = getenv("NROWS") ? atol(getenv("NROWS")) : 1000000, npairs = 4250000 / 256 * 256;
  char* in; int* idx; float* out;
  hipMalloc(&in, nrows * 128); hipMalloc(&idx, npairs * 4); hipMalloc(&out, 1 << 24);
  hipMemset(in, 0, nrows * 128);
  std::vector<int> h(npairs);
  std::mt19937 rng(1);
  for (long i = 0; i < npairs; ++i) h[i] = rng() % nrows;
  hipMemcpy(idx, h.data(), npairs * 4, hipMemcpyHostToDevice);
  for (int grid : {1024, 2048, 4096}) {
    run<0, 2>("fragment-shaped, 2x32 rows in flight/wave", in, idx, npairs, out, grid);
    run<0, 4>("fragment-shaped, 4x32 rows in flight/wave", in, idx, npairs, out, grid);
    run<1, 2>("line-coalesced,  2x32 rows in flight/wave", in, idx, npairs, out, grid);
    run<1, 4>("line-coalesced,  4x32 rows in flight/wave", in, idx, npairs, out, grid);
  }
  // sorted indices (sequential-ish) for reference
  std::sort(h.begin(), h.end());
  hipMemcpy(idx, h.data(), npairs * 4, hipMemcpyHostToDevice);
  run<0, 4>("fragment-shaped, sorted idx", in, idx, npairs, out, 2048);
  run<1, 4>("line-coalesced,  sorted idx", in, idx, npairs, out, 2048);
  run<2, 4>("16x16x32 operand order, sorted idx", in, idx, npairs, out, 2048);
  run<3, 4>("row-shaped 64 B (4 adjacent lanes), sorted", in, idx, npairs, out, 2048);
  // everything in a 128 KB window: L1 / L2 hits only -> the address pipeline itself
  for (long i = 0; i < npairs; ++i) h[i] = (int)(rng() % 1024);
  hipMemcpy(idx, h.data(), npairs * 4, hipMemcpyHostToDevice);
  run<0, 4>("fragment-shaped, 128 KB window", in, idx, npairs, out, 2048);
  run<1, 4>("line-coalesced,  128 KB window", in, idx, npairs, out, 2048);
  run<2, 4>("16x16x32 operand order, 128 KB window", in, idx, npairs, out, 2048);
  run<3, 4>("row-shaped 64 B, 128 KB window", in, idx, npairs, out, 2048);
  // window sweep: where the rate falls from the L2 figure to the fabric / Infinity Cache figure
  for (long win : {1024L, 8192L, 32768L, 131072L, 524288L}) {  // rows of 128 B: 128 KB, 1 MB, 4 MB, 16 MB, 64 MB
    for (long i = 0; i < npairs; ++i) h[i] = (int)(rng() % win);
    hipMemcpy(idx, h.data(), npairs * 4, hipMemcpyHostToDevice);
    char name[64];
    snprintf(name, sizeof name, "fragment-shaped, %ld KB window", win / 8);
    run<0, 4>(name, in, idx, npairs, out, 2048);
    snprintf(name, sizeof name, "row-shaped 64 B, %ld KB window", win / 8);
    run<3, 4>(name, in, idx, npairs, out, 2048);
    snprintf(name, sizeof name, "line-coalesced 128 B, %ld KB window", win / 8);
    run<1, 4>(name, in, idx, npairs, out, 2048);
  }
  return 0;
}
