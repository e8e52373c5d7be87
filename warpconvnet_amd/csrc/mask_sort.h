// mask_sort.h - internal interface of the mask radix sort (mask_sort.hip), shared with the kernel-map tally pass
// (kmap_bucket.hip), which counts and scans the first digit while it reads the masks anyway.
#pragma once

#include "wcn_common.h"

namespace wcn {

constexpr int kRsBits = 9;
constexpr int kRsBins = 1 << kRsBits;
constexpr int kRsTile = 2048;
constexpr int kRsThreads = 256;

// descending order = ascending order of the inverted key
__device__ __forceinline__ uint32_t rs_digit(uint32_t key, int shift) { return ((~key) >> shift) & (kRsBins - 1); }

struct SortPlan {
  int nblk;           // tiles of kRsTile keys
  int passes;
  uint32_t* kbuf[2];  // ping-pong keys
  int32_t* vtmp;      // ping-pong values (the other buffer is `perm`)
  int32_t* counts;    // [kRsBins][nblk] per-(digit, tile) counts, scanned in place
  int32_t* totals;    // [kRsBins] digit totals
  size_t bytes;
};

SortPlan sort_plan(void* workspace, int64_t n, int num_bits);
// `first_counted`: counts / totals already hold the scanned histogram of the first digit
void sort_run(const SortPlan& plan, const uint32_t* mask, int mask_words, int64_t n, int32_t* perm, bool first_counted,
              hipStream_t stream);

}  // namespace wcn
