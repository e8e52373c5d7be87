// wcn_common.h - shared device helpers for the gfx950 sparse-conv kernels.
//
// Key packing / hash follow the reference's published table format so that range errors and
// wrap-around behaviour are identical (reference: warpconvnet/csrc/include/cuhash/hash_functions.cuh:29-84).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/wcn.h"

namespace wcn {

constexpr int kBatchBits = 9;
constexpr int kCoordBits = 18;
constexpr uint32_t kBatchMask = (1u << kBatchBits) - 1;  // 0x1FF
constexpr uint32_t kCoordMask = (1u << kCoordBits) - 1;  // 0x3FFFF
constexpr int kCoordMax = (1 << (kCoordBits - 1)) - 1;   // 131071
constexpr int kCoordMin = -(1 << (kCoordBits - 1));      // -131072
constexpr int kBatchMax = (1 << kBatchBits) - 1;         // 511
constexpr uint64_t kValidBit = 1ull << 63;

// 16-byte hash slot: one dwordx4 load returns key and value together.
struct __attribute__((aligned(16))) Slot {
  uint64_t key;    // 0 = empty
  int32_t value;   // row index (smallest among duplicates); -1 while empty
  int32_t pad;
};

__host__ __device__ __forceinline__ uint64_t pack_key(int b, int x, int y, int z) {
  return kValidBit | ((uint64_t)((uint32_t)b & kBatchMask) << 54) | ((uint64_t)((uint32_t)x & kCoordMask) << 36) |
         ((uint64_t)((uint32_t)y & kCoordMask) << 18) | (uint64_t)((uint32_t)z & kCoordMask);
}

// Splitmix64 finaliser, masked to the (power-of-two) capacity.
__host__ __device__ __forceinline__ uint32_t hash_slot(uint64_t key, uint32_t capacity_mask) {
  key ^= key >> 30;
  key *= 0xBF58476D1CE4E5B9ull;
  key ^= key >> 27;
  key *= 0x94D049BB133111EBull;
  key ^= key >> 31;
  return (uint32_t)key & capacity_mask;
}

__host__ __device__ __forceinline__ bool coord_in_range(int b, int x, int y, int z) {
  return b >= 0 && b <= kBatchMax && x >= kCoordMin && x <= kCoordMax && y >= kCoordMin && y <= kCoordMax &&
         z >= kCoordMin && z <= kCoordMax;
}

// Linear-probe lookup; returns the stored row index or -1.
__device__ __forceinline__ int slot_lookup(const Slot* __restrict__ slots, uint32_t capacity_mask, uint64_t key) {
  uint32_t s = hash_slot(key, capacity_mask);
  for (uint32_t attempts = 0; attempts <= capacity_mask; ++attempts) {
    const uint4 v = *reinterpret_cast<const uint4*>(slots + s);
    const uint64_t k = ((uint64_t)v.y << 32) | v.x;
    if (k == 0ull) return -1;
    if (k == key) return (int)v.z;
    s = (s + 1) & capacity_mask;
  }
  return -1;
}

// LDS-DMA issued through inline asm: hipcc neither counts these in its own s_waitcnt bookkeeping nor orders
// later LDS reads behind them (with the builtin it drains vmcnt(0) before the first ds_read of every step, which
// serialises the ring).  Completion is tracked by the counted s_waitcnt vmcnt(N) below.  M0 (LDS destination base)
// is written and restored inside the statement; `lds_addr` must be wave-uniform.
__device__ __forceinline__ uint32_t lds_addr_of(const void* p) {
  return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) char*)p;
}
__device__ __forceinline__ void glds16(const void* gsrc, uint32_t lds_addr) {
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(gsrc), "s"(lds_addr)
               : "memory");
}
__device__ __forceinline__ void glds4(const void* gsrc, uint32_t lds_addr) {
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(gsrc), "s"(lds_addr)
               : "memory");
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

inline int launch_status() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? WCN_SUCCESS : WCN_ERROR_KERNEL_EXECUTION;
}

inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

// Per-device one-time setup (kernel attributes are per device context): bit d of `done` = device d is set up.  The only
// process-wide state of the library, idempotent: a lost race or a device index above 63 just repeats the setup call.
template <typename F>
inline int once_per_device(unsigned long long& done, F&& setup) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return WCN_ERROR_KERNEL_INITIALIZATION;
  if (dev >= 0 && dev < 64 && ((__atomic_load_n(&done, __ATOMIC_RELAXED) >> dev) & 1ull)) return WCN_SUCCESS;
  if (!setup()) return WCN_ERROR_KERNEL_INITIALIZATION;
  if (dev >= 0 && dev < 64) __atomic_fetch_or(&done, 1ull << dev, __ATOMIC_RELAXED);
  return WCN_SUCCESS;
}

// Epilogue of the gather GEMM, applied in fp32 before the result is rounded to the storage dtype:
//   y = act((acc + bias) * scale + shift + residual)      every term optional (null / 0 = absent)
// BatchNorm in inference mode is scale = gamma / sqrt(var + eps), shift = beta - mean * scale.
struct ConvEpilogue {
  const float* bias = nullptr;     // [cout]
  const float* scale = nullptr;    // [cout], used together with shift
  const float* shift = nullptr;    // [cout]
  const void* residual = nullptr;  // [n_out][cout] in the storage dtype
  int relu = 0;
};

}  // namespace wcn
