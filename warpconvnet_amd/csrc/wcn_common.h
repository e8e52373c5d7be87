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

inline int launch_status() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? WCN_SUCCESS : WCN_ERROR_KERNEL_EXECUTION;
}

inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

}  // namespace wcn
