// mask_sort.h - stable LSD radix argsort of the neighbour masks (descending), hand-written for wave64: the kernel BODIES
// (shared with kmap_bucket.hip, which counts / scans the first digit inside its tally pass and co-schedules the pair
// scatter with the later passes) and the launch plan.
//
// perm = rows ordered by DESCENDING mask word 0, ties in ascending row order.  Rows with the same neighbourhood
// pattern become adjacent, so a wavefront of the gather-GEMM can skip absent offsets.
// Reference counterpart: mask_argsort_uint32 (warpconvnet/csrc/mask_data_kernels.cu:187-220, CUB radix sort).
//
// 9-bit digits (512 bins) => 3 passes for the 27-bit masks of a 3x3x3 kernel (4 for 32 bits).  Per pass:
//   hist     block-local LDS histogram of its 2048-key tile              -> counts[digit][block]
//   scan     one workgroup per 64 digits: exclusive scan over blocks + digit totals (digit bases are scanned in the
//            scatter prologue)                                           -> global base of every (digit, block)
//   scatter  wave w owns a contiguous quarter of the tile; per-wave digit counts give each wave its base, then keys
//            are ranked 64 at a time: lanes with equal digits find each other with 9 ballots ("match-any"),
//            rank = popcount(peers & lower lanes); the running base lives in LDS.  Stable by construction.
// Descending order = ascending order of the inverted key.
//
// TILE ORDER (round 5).  The gather GEMMs cut `perm` into tiles of 64 - 128 rows and run one step per offset that ANY row of
// the tile has, so what a tile costs is the size of the UNION of its rows' masks.  Plain descending-mask order groups rows by
// their high offsets and leaves the low ones random inside a tile: 8.8 steps per 128-row tile on the uniform 1 M scene for
// 4.2 offsets per row.  For odd kernel volumes K = 2c + 1 <= 31 the builder therefore sorts by tile_key(mask) instead - same
// key width, same passes: offsets are paired with their mirror image (k, K-1-k); the high half of the key says which PAIRS a
// row touches, the low half which member, and the whole is ranked in reflected-Gray order so that neighbouring keys differ in
// few bits.  Rows whose neighbours lie in the same few directions become adjacent: 7.2 steps per tile on the uniform scene
// (-18 %; 64-row tiles 7.8 -> 6.8), 9.08 -> 9.11 on the surface scene, whose rows already share whole masks
// (tools/sim_tile_order.py).  Results do not depend on the order (a row accumulates its offsets in ascending k either way).
// wcn_mask_argsort keeps the reference's exact descending-mask semantics (mask_data_kernels.cu:187-220).
#pragma once

#include "wcn_common.h"

namespace wcn {

constexpr int kRsBits = 9;                      // digit width of the exact sorts
constexpr int kRsBins = 1 << kRsBits;
// Tile order of kernel volumes above 18 offsets (the 3x3x3 kernel: 27 key bits): TWO passes of 10-bit digits over the top 20
// bits of tile_key instead of three 9-bit passes over all of it.  The Gray-ranked key degrades gently when its low bits are cut
// (offsets per 128-row tile, uniform / surface 1 M scenes: 27 bits 7.19 / 9.11, 20 bits 7.39 / 9.42, 18 bits 7.47 / 9.98 -
// tools/sim_tile_order.py) and the GEMMs' time moves by a quarter of the step count, while a pass of the sort (histogram + scan +
// scatter launches) is 23 us per million rows.
constexpr int kRsBitsWide = 10;
constexpr int kRsMaxBins = 1 << kRsBitsWide;
constexpr int kRsTile = 2048;
constexpr int kRsThreads = 256;
constexpr int kRsWaves = kRsThreads / 64;
constexpr int kRsPerWave = kRsTile / kRsWaves;  // 512 keys, 8 batches of 64
constexpr size_t rs_scatter_lds(int bits) { return (size_t)(kRsWaves * (1 << bits) + (1 << bits) + kRsWaves) * 4; }

__device__ __forceinline__ uint32_t rs_digit(uint32_t key, int shift, int bits) { return ((~key) >> shift) & ((1u << bits) - 1u); }

// centre index c of a kernel volume the tile order applies to (K = 2c + 1 <= 31, one mask word), else 0 = plain mask order
__host__ __device__ __forceinline__ int tile_key_centre(int num_offsets, int mask_words) {
#ifdef WCN_TILE_KEY_OFF  // A/B build: the reference's descending-mask order everywhere
  return 0;
#endif
  return (mask_words == 1 && num_offsets >= 3 && num_offsets <= 31 && (num_offsets & 1)) ? num_offsets / 2 : 0;
}
__device__ __forceinline__ uint32_t tile_key(uint32_t m, int kc) {
  if (kc <= 0) return m;
  const uint32_t half = (1u << kc) - 1u;
  const uint32_t lo = m & half;                                       // offsets 0 .. c-1
  const uint32_t rh = __brev((m >> (kc + 1)) & half) >> (32 - kc);    // offsets K-1 .. c+1: bit i = offset K-1-i, the mirror of i
  uint32_t k = ((lo | rh) << kc) | lo;                                // touched pairs | which member (01 / 11 tie: same pair set)
  k ^= k >> 1; k ^= k >> 2; k ^= k >> 4; k ^= k >> 8; k ^= k >> 16;   // rank of k in reflected-Gray order
  return (k << 1) | ((m >> kc) & 1u);                                 // the centre offset last (set in every submanifold row)
}

enum RsRole { kRsHist = 0, kRsScan = 1, kRsScatter = 2 };

// A pass reads either the mask tensor itself (first pass: keys at `kin[i * stride]`, row id = i) or the (key, row) PAIRS the
// previous pass wrote, and writes pairs again - ONE 8-B scattered store per element, the scatter kernels are bound by
// the number of scattered stores (tools/cell_probe.hip: 16 us per million) - or, in the last pass, the row ids alone.
struct RsArgs {
  const uint32_t* kin;  // first pass only
  int kc;               // first pass: keys are tile_key(kin[..], kc) (0 = the mask word itself)
  int64_t stride;
  const uint2* pin;     // later passes: (key, row) pairs
  int64_t n;
  int shift;            // first key bit of this pass's digit
  int bits;             // digit width (kRsBits or kRsBitsWide)
  int nblk;
  int32_t* counts;      // [nblk][1 << bits]: a tile's digit counts are one contiguous row (coalesced by its writer and its reader)
  int32_t* totals;      // [1 << bits]
  uint2* pout;          // all but the last pass
  int32_t* vout;        // last pass: the permutation
};

struct RsLaunch {
  int role;
  int blocks;
  RsArgs a;
};

struct SortPlan {
  int nblk;           // tiles of kRsTile keys
  int passes;
  int bits;           // digit width
  int shift0;         // low key bits no pass looks at (wide plans sort the top passes * bits bits of the key)
  uint2* pbuf[2];     // ping-pong (key, row) pairs
  int32_t* counts;    // [nblk][1 << bits] per-(tile, digit) counts, scanned in place along the tiles
  int32_t* totals;    // [1 << bits] digit totals
  size_t bytes;
};

// `tile_order`: the plan of a tile_key sort (wide digits over the top bits for keys above 18 bits); else the exact sort
SortPlan sort_plan(void* workspace, int64_t n, int num_bits, bool tile_order);
// every launch of the sort, in order (at most 3 per pass); `first_counted`: counts / totals already hold the scanned
// histogram of the first digit, so the first pass is its scatter alone
int sort_launches(const SortPlan& plan, const uint32_t* mask, int mask_words, int64_t n, int32_t* perm, bool first_counted,
                  int kc, RsLaunch out[12]);
void sort_run_range(const RsLaunch* launches, int begin, int end, hipStream_t stream);

// ---- kernel bodies: `blk` = tile / digit index, `smem` >= 4 B << bits (hist), 16 B (scan), rs_scatter_lds(bits) (scatter) ----
__device__ __forceinline__ void rs_hist_body(const RsArgs& a, int blk, char* smem) {
  int* s_hist = reinterpret_cast<int*>(smem);
  const int bins = 1 << a.bits;
  for (int i = threadIdx.x; i < bins; i += kRsThreads) s_hist[i] = 0;
  __syncthreads();
  const int64_t base = (int64_t)blk * kRsTile;
  // all of a thread's keys are requested before the first one is used: predicated loads inside the loop are waited for
  // one at a time (8 memory round trips instead of 1)
  constexpr int kPer = kRsTile / kRsThreads;
  uint32_t k[kPer];
#pragma unroll
  for (int j = 0; j < kPer; ++j) {
    const int64_t idx = base + threadIdx.x + j * kRsThreads;
    const int64_t at = idx < a.n ? idx : a.n - 1;
    k[j] = a.pin ? a.pin[at].x : tile_key(a.kin[at * a.stride], a.kc);
  }
#pragma unroll
  for (int j = 0; j < kPer; ++j)
    if (base + threadIdx.x + j * kRsThreads < a.n) atomicAdd(&s_hist[rs_digit(k[j], a.shift, a.bits)], 1);
  __syncthreads();
  for (int i = threadIdx.x; i < bins; i += kRsThreads) a.counts[(int64_t)blk * bins + i] = s_hist[i];
}

// Exclusive scan over the TILES of every digit's counts, in place, and the digit totals.  counts is [nblk][bins]: workgroup g owns
// the 16 digits [16 g, 16 g + 16) - a 64-B piece of every row - and its 16 row groups a sixteenth of the rows each (the
// [digit][tile] layout of rounds 2-4 made the writers of the counts store 512 - 1 024 words to as many different lines per tile).
// Up to 512 tiles (1 M rows) a thread keeps its <= 32 counts in registers: one round trip of loads, the group prefix through LDS,
// stores.  `smem` >= 256 ints.
constexpr int kRsScanCols = 16;
constexpr int kRsScanTrip = 32;
__device__ __forceinline__ void rs_scan_body(int32_t* __restrict__ counts, int32_t* __restrict__ totals, int nblk, int bins, int g,
                                             char* smem) {
  int* s_part = reinterpret_cast<int*>(smem);  // [16 row groups][16 digits]
  const int tid = threadIdx.x, dl = tid & (kRsScanCols - 1), rgp = tid / kRsScanCols;
  constexpr int kGroups = kRsThreads / kRsScanCols;
  const int d = g * kRsScanCols + dl;
  const bool live = d < bins;
  const int chunk = (nblk + kGroups - 1) / kGroups;
  const int b0 = rgp * chunk, b1 = (b0 + chunk < nblk) ? (b0 + chunk) : nblk;
  int32_t* col = counts + (live ? d : 0);
  int sum = 0;
  int v[kRsScanTrip];
  const bool in_regs = chunk <= kRsScanTrip;
  if (in_regs) {
#pragma unroll
    for (int j = 0; j < kRsScanTrip; ++j) v[j] = (live && b0 + j < b1) ? col[(int64_t)(b0 + j) * bins] : 0;
#pragma unroll
    for (int j = 0; j < kRsScanTrip; ++j) sum += v[j];
  } else if (live) {
    for (int b = b0; b < b1; ++b) sum += col[(int64_t)b * bins];
  }
  s_part[rgp * kRsScanCols + dl] = sum;
  __syncthreads();
  int run = 0, total = 0;
#pragma unroll
  for (int r = 0; r < kGroups; ++r) {
    const int q = s_part[r * kRsScanCols + dl];
    if (r < rgp) run += q;
    total += q;
  }
  if (!live) return;
  if (in_regs) {
#pragma unroll
    for (int j = 0; j < kRsScanTrip; ++j)
      if (b0 + j < b1) {
        col[(int64_t)(b0 + j) * bins] = run;
        run += v[j];
      }
  } else {
    for (int b = b0; b < b1; ++b) {
      const int c = col[(int64_t)b * bins];
      col[(int64_t)b * bins] = run;
      run += c;
    }
  }
  if (rgp == 0) totals[d] = total;
}
__device__ __forceinline__ void rs_scan_body(const RsArgs& a, int g, char* smem) {
  rs_scan_body(a.counts, a.totals, a.nblk, 1 << a.bits, g, smem);
}

template <int BITS>
__device__ __forceinline__ void rs_scatter_body(const RsArgs& a, int blk, char* smem) {
  constexpr int BINS = 1 << BITS;
  constexpr int kPerT = BINS / kRsThreads;  // digits per thread (2 or 4)
  int* s_base = reinterpret_cast<int*>(smem);  // [kRsWaves][BINS] per-wave digit counts, then running output positions
  int* s_dstart = s_base + kRsWaves * BINS;  // first output position of every digit (scan of the digit totals)
  int* s_wsum = s_dstart + BINS;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t wave_begin = (int64_t)blk * kRsTile + (int64_t)wave * kRsPerWave;
  // the wave's 8 x 64 keys (and carried values) are requested up front and kept in registers: the ranking loop below is a
  // serial chain through LDS, and a global load inside it costs one memory round trip per 64 keys
  constexpr int kBatches = kRsPerWave / 64;
  uint32_t kreg[kBatches];
  int32_t vreg[kBatches];
#pragma unroll
  for (int j = 0; j < kBatches; ++j) {
    const int64_t idx = wave_begin + j * 64 + lane;
    const int64_t at = idx < a.n ? idx : a.n - 1;
    if (a.pin) {
      const uint2 kv = a.pin[at];
      kreg[j] = kv.x;
      vreg[j] = (int32_t)kv.y;
    } else {
      kreg[j] = tile_key(a.kin[at * a.stride], a.kc);
      vreg[j] = (int32_t)at;
    }
  }
  // ... and so are the digit totals and this block's (digit, block) bases: everything the kernel reads from global memory
  // is in flight at once
  int tt[kPerT];
#pragma unroll
  for (int j = 0; j < kPerT; ++j) tt[j] = a.totals[kPerT * tid + j];
  int blk_base[kPerT];
#pragma unroll
  for (int j = 0; j < kPerT; ++j) blk_base[j] = a.counts[(int64_t)blk * BINS + tid + j * kRsThreads];
  {  // exclusive scan of the digit totals: kPerT consecutive digits per thread, wave scan, 4 wave partials
    int mine = 0;
#pragma unroll
    for (int j = 0; j < kPerT; ++j) mine += tt[j];
    int incl = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int t = __shfl_up(incl, d);
      if (lane >= d) incl += t;
    }
    if (lane == 63) s_wsum[wave] = incl;
    __syncthreads();
    int base = incl - mine;
    for (int w = 0; w < wave; ++w) base += s_wsum[w];
#pragma unroll
    for (int j = 0; j < kPerT; ++j) {
      s_dstart[kPerT * tid + j] = base;
      base += tt[j];
    }
  }
  for (int i = tid; i < kRsWaves * BINS; i += kRsThreads) s_base[i] = 0;
  __syncthreads();
  // phase 1: digit counts of this wave's sub-tile
#pragma unroll
  for (int j = 0; j < kBatches; ++j)
    if (wave_begin + j * 64 + lane < a.n) atomicAdd(&s_base[wave * BINS + rs_digit(kreg[j], a.shift, BITS)], 1);
  __syncthreads();
  // phase 2: counts -> starting positions (global base of (digit, block) + waves before this one)
#pragma unroll
  for (int j = 0; j < kPerT; ++j) {
    const int d = tid + j * kRsThreads;
    int run = s_dstart[d] + blk_base[j];
#pragma unroll
    for (int w = 0; w < kRsWaves; ++w) {
      const int c = s_base[w * BINS + d];
      s_base[w * BINS + d] = run;
      run += c;
    }
  }
  __syncthreads();
  // phase 3: rank 64 keys at a time, in order
  volatile int* my_base = s_base + wave * BINS;
  const unsigned long long lt = (1ull << lane) - 1ull;
#pragma unroll
  for (int j = 0; j < kBatches; ++j) {
    const int64_t idx = wave_begin + j * 64 + lane;
    const bool live = idx < a.n;
    const uint32_t key = kreg[j];
    const uint32_t d = live ? rs_digit(key, a.shift, BITS) : 0u;
    unsigned long long peers = __ballot(live);
#pragma unroll
    for (int b = 0; b < BITS; ++b) {
      const bool bit = (d >> b) & 1u;
      const unsigned long long ball = __ballot(bit);
      peers &= bit ? ball : ~ball;
    }
    if (live) {
      const int rank = __popcll(peers & lt);
      const int pos = my_base[d] + rank;
      if (a.pout) a.pout[pos] = make_uint2(key, (uint32_t)vreg[j]);
      else a.vout[pos] = vreg[j];
      // the last peer advances the running base after every peer has read it (same wave, LDS ops are in order)
      if ((peers >> lane) == 1ull) my_base[d] = pos + 1;
    }
  }
}

template <int ROLE>
__device__ __forceinline__ void rs_body(const RsArgs& a, int blk, char* smem) {
  if (ROLE == kRsHist) rs_hist_body(a, blk, smem);
  else if (ROLE == kRsScan) rs_scan_body(a, blk, smem);
  else if (a.bits == kRsBitsWide) rs_scatter_body<kRsBitsWide>(a, blk, smem);
  else rs_scatter_body<kRsBits>(a, blk, smem);
}

}  // namespace wcn
