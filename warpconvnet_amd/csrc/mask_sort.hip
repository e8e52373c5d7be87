// mask_sort.hip - stable LSD radix argsort of the neighbour masks (descending), hand-written for wave64.
//
// perm = rows ordered by DESCENDING mask word 0, ties in ascending row order.  Rows with the same neighbourhood
// pattern become adjacent, so a wavefront of the gather-GEMM can skip absent offsets.
// Reference counterpart: mask_argsort_uint32 (warpconvnet/csrc/mask_data_kernels.cu:187-220, CUB radix sort).
//
// 9-bit digits (512 bins) => 3 passes for the 27-bit masks of a 3x3x3 kernel (4 for 32 bits).  Per pass:
//   hist     block-local LDS histogram of its 8192-key tile              -> counts[digit][block]
//   scan     one workgroup per digit: exclusive scan over blocks + digit total (digit bases are scanned in the
//            scatter prologue)                                           -> global base of every (digit, block)
//   scatter  wave w owns a contiguous quarter of the tile; per-wave digit counts give each wave its base, then keys
//            are ranked 64 at a time: lanes with equal digits find each other with 9 ballots ("match-any"),
//            rank = popcount(peers & lower lanes); the running base lives in LDS.  Stable by construction.
// Descending order = ascending order of the inverted key.
#include "mask_sort.h"

namespace wcn {

constexpr int kRsWaves = kRsThreads / 64;
constexpr int kRsPerWave = kRsTile / kRsWaves;  // 512 keys, 8 batches of 64

// first pass reads the mask tensor directly (stride mw); later passes read the ping-pong key buffer
__global__ __launch_bounds__(kRsThreads) void rs_hist_kernel(const uint32_t* __restrict__ keys, int64_t key_stride,
                                                             int64_t n, int shift, int nblk,
                                                             int32_t* __restrict__ counts) {
  __shared__ int s_hist[kRsBins];
  for (int i = threadIdx.x; i < kRsBins; i += kRsThreads) s_hist[i] = 0;
  __syncthreads();
  const int64_t base = (int64_t)blockIdx.x * kRsTile;
  // all of a thread's keys are requested before the first one is used: predicated loads inside the loop are waited for
  // one at a time (8 memory round trips instead of 1)
  constexpr int kPer = kRsTile / kRsThreads;
  uint32_t k[kPer];
#pragma unroll
  for (int j = 0; j < kPer; ++j) {
    const int64_t idx = base + threadIdx.x + j * kRsThreads;
    k[j] = keys[(idx < n ? idx : n - 1) * key_stride];
  }
#pragma unroll
  for (int j = 0; j < kPer; ++j)
    if (base + threadIdx.x + j * kRsThreads < n) atomicAdd(&s_hist[rs_digit(k[j], shift)], 1);
  __syncthreads();
  for (int i = threadIdx.x; i < kRsBins; i += kRsThreads) counts[(int64_t)i * nblk + blockIdx.x] = s_hist[i];
}

// one workgroup per digit: exclusive scan of counts[d][0..nblk) in place, totals[d] = row sum
__global__ __launch_bounds__(256) void rs_scan_kernel(int32_t* __restrict__ counts, int nblk,
                                                      int32_t* __restrict__ totals) {
  __shared__ int s_part[4];
  int32_t* row = counts + (int64_t)blockIdx.x * nblk;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int chunk = (nblk + 255) / 256;
  const int b0 = tid * chunk;
  const int b1 = (b0 + chunk < nblk) ? (b0 + chunk) : nblk;
  int sum = 0;
  for (int b = b0; b < b1; ++b) sum += row[b];
  int incl = sum;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int t = __shfl_up(incl, d);
    if (lane >= d) incl += t;
  }
  if (lane == 63) s_part[wave] = incl;
  __syncthreads();
  int run = incl - sum;
  for (int w = 0; w < wave; ++w) run += s_part[w];
  for (int b = b0; b < b1; ++b) {
    const int v = row[b];
    row[b] = run;
    run += v;
  }
  if (tid == 255) totals[blockIdx.x] = run;
}

__global__ __launch_bounds__(kRsThreads) void rs_scatter_kernel(const uint32_t* __restrict__ keys_in, int64_t key_stride,
                                                                const int32_t* __restrict__ vals_in, int64_t n,
                                                                int shift, int nblk, const int32_t* __restrict__ counts,
                                                                const int32_t* __restrict__ totals,
                                                                uint32_t* __restrict__ keys_out,
                                                                int32_t* __restrict__ vals_out) {
  __shared__ int s_base[kRsWaves][kRsBins];  // per-wave digit counts, then running output positions
  __shared__ int s_dstart[kRsBins];          // first output position of every digit (scan of the digit totals)
  __shared__ int s_wsum[kRsWaves];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t wave_begin = (int64_t)blockIdx.x * kRsTile + (int64_t)wave * kRsPerWave;
  // the wave's 8 x 64 keys (and carried values) are requested up front and kept in registers: the ranking loop below is a
  // serial chain through LDS, and a global load inside it costs one memory round trip per 64 keys
  constexpr int kBatches = kRsPerWave / 64;
  uint32_t kreg[kBatches];
  int32_t vreg[kBatches];
#pragma unroll
  for (int j = 0; j < kBatches; ++j) {
    const int64_t idx = wave_begin + j * 64 + lane;
    const int64_t at = idx < n ? idx : n - 1;
    kreg[j] = keys_in[at * key_stride];
    vreg[j] = vals_in ? vals_in[at] : (int32_t)at;
  }
  // ... and so are the digit totals and this block's (digit, block) bases: everything the kernel reads from global memory
  // is in flight at once (three dependent round trips were most of its 15 us)
  const int t0 = totals[2 * tid], t1 = totals[2 * tid + 1];
  int blk_base[kRsBins / kRsThreads];
#pragma unroll
  for (int j = 0; j < kRsBins / kRsThreads; ++j) blk_base[j] = counts[(int64_t)(tid + j * kRsThreads) * nblk + blockIdx.x];
  {  // exclusive scan of the 512 digit totals: 2 per thread, wave scan, 4 wave partials
    int incl = t0 + t1;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int t = __shfl_up(incl, d);
      if (lane >= d) incl += t;
    }
    if (lane == 63) s_wsum[wave] = incl;
    __syncthreads();
    int base = incl - (t0 + t1);
    for (int w = 0; w < wave; ++w) base += s_wsum[w];
    s_dstart[2 * tid] = base;
    s_dstart[2 * tid + 1] = base + t0;
  }
  for (int i = tid; i < kRsWaves * kRsBins; i += kRsThreads) (&s_base[0][0])[i] = 0;
  __syncthreads();
  // phase 1: digit counts of this wave's sub-tile
#pragma unroll
  for (int j = 0; j < kBatches; ++j)
    if (wave_begin + j * 64 + lane < n) atomicAdd(&s_base[wave][rs_digit(kreg[j], shift)], 1);
  __syncthreads();
  // phase 2: counts -> starting positions (global base of (digit, block) + waves before this one)
#pragma unroll
  for (int j = 0; j < kRsBins / kRsThreads; ++j) {
    const int d = tid + j * kRsThreads;
    int run = s_dstart[d] + blk_base[j];
#pragma unroll
    for (int w = 0; w < kRsWaves; ++w) {
      const int c = s_base[w][d];
      s_base[w][d] = run;
      run += c;
    }
  }
  __syncthreads();
  // phase 3: rank 64 keys at a time, in order
  volatile int* my_base = &s_base[wave][0];
  const unsigned long long lt = (1ull << lane) - 1ull;
#pragma unroll
  for (int j = 0; j < kBatches; ++j) {
    const int64_t idx = wave_begin + j * 64 + lane;
    const bool live = idx < n;
    const uint32_t key = kreg[j];
    const uint32_t d = live ? rs_digit(key, shift) : 0u;
    unsigned long long peers = __ballot(live);
#pragma unroll
    for (int b = 0; b < kRsBits; ++b) {
      const bool bit = (d >> b) & 1u;
      const unsigned long long ball = __ballot(bit);
      peers &= bit ? ball : ~ball;
    }
    if (live) {
      const int rank = __popcll(peers & lt);
      const int pos = my_base[d] + rank;
      keys_out[pos] = key;
      vals_out[pos] = vreg[j];
      // the last peer advances the running base after every peer has read it (same wave, LDS ops are in order)
      if ((peers >> lane) == 1ull) my_base[d] = pos + 1;
    }
  }
}


static size_t al256(size_t v) { return (v + 255) & ~(size_t)255; }

SortPlan sort_plan(void* workspace, int64_t n, int num_bits) {
  SortPlan p;
  p.nblk = (int)ceil_div(n > 0 ? n : 1, kRsTile);
  char* ws = (char*)workspace;
  const size_t seg = al256((size_t)(n > 0 ? n : 1) * 4);
  p.kbuf[0] = (uint32_t*)ws;
  p.kbuf[1] = (uint32_t*)(ws ? ws + seg : nullptr);
  p.vtmp = (int32_t*)(ws ? ws + 2 * seg : nullptr);
  p.counts = (int32_t*)(ws ? ws + 3 * seg : nullptr);
  p.totals = p.counts ? p.counts + (int64_t)kRsBins * p.nblk : nullptr;
  // only the low `num_bits` bits of word 0 can be set (num_bits = min(K, 32)): 3 passes for a 3x3x3 kernel
  if (num_bits > 32) num_bits = 32;
  if (num_bits < 1) num_bits = 1;
  p.passes = (num_bits + kRsBits - 1) / kRsBits;
  p.bytes = 3 * seg + al256((size_t)kRsBins * (p.nblk + 1) * 4) + 256;
  return p;
}

void sort_run(const SortPlan& p, const uint32_t* mask, int mask_words, int64_t n, int32_t* perm, bool first_counted,
              hipStream_t s) {
  const uint32_t* kin = mask;
  int64_t stride = mask_words;
  const int32_t* vin = nullptr;
  for (int pass = 0; pass < p.passes; ++pass) {
    const int shift = pass * kRsBits;
    uint32_t* kout = p.kbuf[pass & 1];
    int32_t* vout = ((p.passes - 1 - pass) & 1) ? p.vtmp : perm;  // last pass writes perm
    if (!(pass == 0 && first_counted)) {  // the tally pass of the kernel-map build counts and scans the first digit itself
      hipLaunchKernelGGL(rs_hist_kernel, dim3(p.nblk), dim3(kRsThreads), 0, s, kin, stride, n, shift, p.nblk, p.counts);
      hipLaunchKernelGGL(rs_scan_kernel, dim3(kRsBins), dim3(256), 0, s, p.counts, p.nblk, p.totals);
    }
    hipLaunchKernelGGL(rs_scatter_kernel, dim3(p.nblk), dim3(kRsThreads), 0, s, kin, stride, vin, n, shift, p.nblk,
                       (const int32_t*)p.counts, (const int32_t*)p.totals, kout, vout);
    kin = kout;
    stride = 1;
    vin = vout;
  }
}

}  // namespace wcn

using namespace wcn;

extern "C" {

size_t wcn_mask_argsort_workspace(int64_t n) { return sort_plan(nullptr, n, 32).bytes; }

int wcn_mask_argsort(const uint32_t* mask, int32_t mask_words, int32_t num_bits, int64_t n, int32_t* perm,
                     void* workspace, size_t workspace_bytes, wcn_stream_t stream) {
  if (n < 0 || mask_words < 1 || num_bits < 1) return WCN_ERROR_INVALID_PARAMETERS;
  if (n == 0) return WCN_SUCCESS;
  if (n >= (1ll << 31) || !mask || !perm || !workspace || workspace_bytes < wcn_mask_argsort_workspace(n))
    return WCN_ERROR_INVALID_PARAMETERS;
  sort_run(sort_plan(workspace, n, num_bits), mask, mask_words, n, perm, false, (hipStream_t)stream);
  return launch_status();
}

}  // extern "C"
