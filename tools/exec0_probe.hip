// Dev probe (gfx950): does a VMEM / LDS-DMA instruction issued with EXEC = 0 take a vmcnt slot (in order)?
// Each workgroup: one real LDS-DMA from a cold line, then NZ instructions with EXEC = 0, then s_waitcnt vmcnt(NZ), then
// the LDS data is checked.  If EXEC = 0 instructions were NOT counted, vmcnt(NZ) would not wait for the real load and the
// check would read stale LDS (zeros).  Prints the number of stale reads for NZ = 0 (control: vmcnt(0)), 1, 3, 7.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>

template <int NZ>
__global__ void probe(const uint32_t* __restrict__ src, size_t nwords, uint32_t* __restrict__ bad, uint32_t seed) {
  __shared__ __attribute__((aligned(16))) uint32_t buf[64 * 4 + 64 * 4];
  const int lane = threadIdx.x;
  for (int i = lane; i < 512; i += 64) buf[i] = 0u;
  __syncthreads();
  // a pseudo-random 1-KiB block of the (cold) source per workgroup and trial
  uint64_t h = (uint64_t)(blockIdx.x + 1) * 0x9E3779B97F4A7C15ull + seed * 0xBF58476D1CE4E5B9ull;
  h ^= h >> 29;
  const size_t blk = (size_t)(h % (nwords / 256)) * 256;
  const uint32_t* g = src + blk + lane * 4;
  const uint32_t lds = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) char*)buf;
  uint32_t keep;
  unsigned long long ex;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %3\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %2, off\n\t"
      "s_mov_b64 %1, exec\n\t"
      "s_mov_b64 exec, 0\n\t"
      "s_mov_b32 m0, %4\n\t"
      "s_nop 0\n\t"
      ".rept %5\n\t"
      "global_load_lds_dwordx4 %2, off\n\t"
      ".endr\n\t"
      "s_mov_b64 exec, %1\n\t"
      "s_mov_b32 m0, %0\n\t"
      "s_waitcnt vmcnt(%5)\n\t"
      : "=&s"(keep), "=&s"(ex)
      : "v"(g), "s"(lds), "s"(lds + 1024u), "n"(NZ)
      : "memory");
  uint32_t nb = 0;
  for (int j = 0; j < 4; ++j) {
    const uint32_t got = buf[lane * 4 + j];
    const uint32_t want = (uint32_t)((blk + lane * 4 + j) * 2654435761u) | 1u;
    if (got != want) ++nb;
  }
  if (nb) atomicAdd(bad, nb);
}

__global__ void fill(uint32_t* p, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) p[i] = (uint32_t)(i * 2654435761u) | 1u;
}

int main() {
  const size_t nwords = (size_t)1 << 29;  // 2 GiB: far beyond the Infinity Cache
  uint32_t *src, *bad;
  hipMalloc(&src, nwords * 4);
  hipMalloc(&bad, 4 * 4);
  fill<<<4096, 256>>>(src, nwords);
  hipMemset(bad, 0, 16);
  hipDeviceSynchronize();
  for (int t = 0; t < 20; ++t) {
    probe<0><<<4096, 64>>>(src, nwords, bad + 0, t);
    probe<1><<<4096, 64>>>(src, nwords, bad + 1, 100 + t);
    probe<3><<<4096, 64>>>(src, nwords, bad + 2, 200 + t);
    probe<7><<<4096, 64>>>(src, nwords, bad + 3, 300 + t);
  }
  uint32_t hb[4];
  hipMemcpy(hb, bad, 16, hipMemcpyDeviceToHost);
  printf("exec0_probe: stale words with vmcnt(NZ) after 1 real + NZ exec=0 LDS-DMA: NZ=0: %u  NZ=1: %u  NZ=3: %u  NZ=7: %u\n", hb[0], hb[1],
         hb[2], hb[3]);
  printf("  (all zero => EXEC=0 VMEM instructions occupy vmcnt slots in order; NZ>0 nonzero => they do not)\n");
  return 0;
}
