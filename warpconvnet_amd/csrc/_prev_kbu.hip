// kmap_bucket.hip - per-offset bucketing of a neighbour table: the CSR pair lists (in_maps / out_maps / offsets) the
// weight-gradient kernel and the reference API consume, in DETERMINISTIC order (every bucket ordered by output row,
// no atomics on cursors).
//
//   tally    one pass over the masks: per-(offset, 256-row tile) pair counts from wave ballots, the first digit's
//            histogram of the mask sort (it reads the same words), and - on the binned path - repair of the rows of
//            duplicate coordinates (rows no block enumerated copy their winner's table row)
//   scan     one workgroup per offset (exclusive scan over tiles) and per sort digit, ONE launch; the workgroup that
//            finishes last turns the totals into offsets[K+1] and mirrors them + the status word to pinned host memory
//   scatter  one workgroup per tile: neighbour rows through LDS with whole-row loads, pairs ranked by ballot +
//            popcount, STAGED in LDS bucket by bucket and written out as contiguous runs (a 64-row wave storing its ~10
//            pairs per bucket directly is one partial-line write request per bucket, wave and array)
//
// Reference behaviour replaced: warpconvnet/csrc/cuhash_kernel_map.cu:508-599 (postprocess_count / postprocess_scatter
// with atomic cursors), mask_data_kernels.cu:23-124.
#include "/tmp/prevhdr/kmap_cells.h"
#include "mask_sort.h"

namespace wcn {

constexpr int kTileRows = 256;  // rows per bucket tile = 4 waves x 64
constexpr int kBkThreads = 256;
constexpr int kTallyThreads = (kRsTile / kTileRows) * 64;  // one wave per tile, one workgroup per sort tile
constexpr int kStageCap = 3072;  // pairs staged in LDS per (tile, mask word); denser tiles store directly

static_assert(kRsTile % kTileRows == 0, "a sort tile is a whole number of bucket tiles");
static_assert(kBkThreads == kRsThreads, "the scan launch runs rs_scan_body (mask_sort.h) in its digit workgroups");

// rows of duplicate coordinates (binned path): copy the table row of the row the cell keeps
__device__ __noinline__ void repair_row(int64_t row, const int4* __restrict__ coords, const CellTable& t, uint32_t cmask,
                                        int kp, int mw, int32_t* __restrict__ nbr, uint32_t* __restrict__ mask,
                                        int32_t* __restrict__ status) {
  const int4 c = coords[row];
  const int s = block_find(t.slots, cmask, pack_key(c.x, c.y >> kBlkShift, c.z >> kBlkShift, c.w >> kBlkShift));
  int id = s >= 0 ? t.slots[s].id : -1;
  if (id >= 0) id &= ~kIdLateBit;
  int w = -1;
  if (id >= 0) {
    const int cell = ((c.y & (kBlk - 1)) * kBlk + (c.z & (kBlk - 1))) * kBlk + (c.w & (kBlk - 1));
    w = t.cells[(int64_t)id * kCells + cell];
  }
  // kp: ints per table row (the dense pitch, or kCompactPitch | kCompactFlag: a compact row is copied like a dense one)
  const bool compact = kp < 0;
  kp &= 0x7FFFFFFF;
  if (w >= 0 && w != row && !(mask[(int64_t)w * mw + (mw - 1)] & kMaskUnwritten)) {
    atomicOr(status, (int)(WCN_FLAG_DUPLICATE_COORD | (w > row ? WCN_FLAG_NEED_STRICT : 0)));
    for (int k = 0; k < kp; ++k) nbr[row * kp + k] = nbr[(int64_t)w * kp + k];
    for (int q = 0; q < mw; ++q) mask[row * mw + q] = mask[(int64_t)w * mw + q];
  } else {  // block table overflow (flagged by the builder): defined, empty content
    for (int k = 0; k < kp; ++k) nbr[row * kp + k] = compact ? 0 : -1;  // (compact: word 0 = mask = no offsets)
    for (int q = 0; q < mw; ++q) mask[row * mw + q] = 0u;
  }
}

// counts[k][tile] = rows of the 256-row tile that have offset k (k-major for the scan);  dcounts[digit][block] = first
// digit histogram of the mask sort over the block's 2048 rows (HIST).
template <bool HIST, bool REPAIR>
__global__ __launch_bounds__(kTallyThreads) void kmap_tally_kernel(uint32_t* __restrict__ mask, int32_t* __restrict__ nbr,
                                                                int64_t m, int K, int kp, int mw, int64_t ntile,
                                                                int32_t* __restrict__ counts, int32_t* __restrict__ ticket,
                                                                int nblk_sort, int32_t* __restrict__ dcounts,
                                                                const int4* __restrict__ coords, CellTable t,
                                                                uint32_t cmask, int32_t* __restrict__ status, int kc,
                                                                int sort_shift, int sort_bits) {
  __shared__ int s_hist[kRsMaxBins];
  const int sort_bins = 1 << sort_bits;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (HIST) {
    for (int i = tid; i < sort_bins; i += kTallyThreads) s_hist[i] = 0;
    __syncthreads();
  }
  if (blockIdx.x == 0 && tid == 0) *ticket = 0;  // consumed by the scan launch behind this one
  constexpr int kTilesPerBlock = kRsTile / kTileRows;  // 8 = waves per workgroup
  for (int tt = 0; tt < 1; ++tt) {
    const int64_t tile = (int64_t)blockIdx.x * kTilesPerBlock + wave;
    if (tile >= ntile) break;
    const int64_t row0 = tile * kTileRows;
    uint32_t last[kTileRows / 64];
#pragma unroll
    for (int sb = 0; sb < kTileRows / 64; ++sb) {  // the four 64-row groups of the tile: requested together
      const int64_t row = row0 + sb * 64 + lane;
      last[sb] = row < m ? mask[row * mw + (mw - 1)] : 0u;
    }
    if (REPAIR) {
#pragma unroll
      for (int sb = 0; sb < kTileRows / 64; ++sb) {
        const int64_t row = row0 + sb * 64 + lane;
        if (last[sb] & kMaskUnwritten) {
          repair_row(row, coords, t, cmask, kp, mw, nbr, mask, status);
          last[sb] = mask[row * mw + (mw - 1)];
        }
      }
    }
    for (int w = 0; w < mw; ++w) {
      const int kend = (K - w * 32) < 32 ? (K - w * 32) : 32;
      int mine = 0;
#pragma unroll
      for (int sb = 0; sb < kTileRows / 64; ++sb) {
        const int64_t row = row0 + sb * 64 + lane;
        const uint32_t bits = (w == mw - 1) ? last[sb] : (row < m ? mask[row * mw + w] : 0u);
        if (HIST && w == 0 && row < m) atomicAdd(&s_hist[rs_digit(tile_key(bits, kc), sort_shift, sort_bits)], 1);
        for (int b = 0; b < kend; ++b) {
          const int c = __popcll(__ballot((bits >> b) & 1u));
          if (lane == b) mine += c;
        }
      }
      if (lane < kend) counts[(int64_t)(w * 32 + lane) * ntile + tile] = mine;
    }
  }
  if (HIST) {
    __syncthreads();
    if ((int)blockIdx.x < nblk_sort)
      for (int i = tid; i < sort_bins; i += kTallyThreads) dcounts[(int64_t)blockIdx.x * sort_bins + i] = s_hist[i];  // [tile][digit]
  }
}

// exclusive scan of one row of 32-bit counts by one 256-thread workgroup; returns the row total (all threads)
__device__ __forceinline__ int scan_row_256(int32_t* __restrict__ c, int64_t n, int* s_wave) {
  constexpr int kPer = 16;  // 4096 counts per trip: a 1 M-row map is one trip (one memory round trip)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int carry = 0;
  for (int64_t base = 0; base < n; base += kBkThreads * kPer) {
    const int64_t i0 = base + (int64_t)tid * kPer;
    int v[kPer];
    int sum = 0;
    // 16-B pieces per lane (rows are 16-B aligned, n is a multiple of 4): a lane-strided 4-B access costs one
    // texture-addresser slot per lane and element
#pragma unroll
    for (int j = 0; j < kPer; j += 4) {
      int4 q = make_int4(0, 0, 0, 0);
      if (i0 + j < n) q = *reinterpret_cast<const int4*>(c + i0 + j);
      v[j] = q.x; v[j + 1] = q.y; v[j + 2] = q.z; v[j + 3] = q.w;
      sum += q.x + q.y + q.z + q.w;
    }
    int incl = sum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int up = __shfl_up(incl, d);
      if (lane >= d) incl += up;
    }
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    int wave_base = 0, trip_total = 0;
#pragma unroll
    for (int w = 0; w < kBkThreads / 64; ++w) {
      const int q = s_wave[w];
      if (w < wave) wave_base += q;
      trip_total += q;
    }
    int run = carry + wave_base + incl - sum;
#pragma unroll
    for (int j = 0; j < kPer; j += 4) {
      int4 q;
      q.x = run; run += v[j];
      q.y = run; run += v[j + 1];
      q.z = run; run += v[j + 2];
      q.w = run; run += v[j + 3];
      if (i0 + j < n) *reinterpret_cast<int4*>(c + i0 + j) = q;
    }
    carry += trip_total;
    __syncthreads();  // s_wave is rewritten by the next trip
  }
  return carry;
}

// blocks [0, K): offset rows; blocks [K, K + bins / 16): 16 sort digits each (nblk_sort > 0).  `mirror` (may be null):
// device-accessible pinned HOST buffer [K+2] that receives the offsets and the status word in the same kernel - the host
// waits for an event behind this launch instead of queueing a separate D2H copy.
__global__ __launch_bounds__(kBkThreads) void kmap_scan_kernel(int32_t* __restrict__ counts, int64_t ntile, int K,
                                                               int32_t* __restrict__ totals, int32_t* __restrict__ ticket,
                                                               int32_t* __restrict__ offsets,
                                                               const int32_t* __restrict__ status,
                                                               int32_t* __restrict__ mirror, int32_t* __restrict__ dcounts,
                                                               int nblk_sort, int32_t* __restrict__ dtotals, int sort_bins) {
  __shared__ int s_wave[kBkThreads / 64];
  __shared__ int s_last;
  const int tid = threadIdx.x;
  if ((int)blockIdx.x >= K) {  // 16 sort digits: exclusive scan over the sort tiles, digit totals (mask_sort.h)
    __shared__ int s_cols[kBkThreads];
    rs_scan_body(dcounts, dtotals, nblk_sort, sort_bins, (int)blockIdx.x - K, reinterpret_cast<char*>(s_cols));
    return;
  }
  const int total = scan_row_256(counts + (int64_t)blockIdx.x * ntile, ntile, s_wave);
  if (tid == 0) {
    // device-scope store + fence + ticket: the workgroup that draws the last ticket sees every total
    __hip_atomic_store(&totals[blockIdx.x], total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __threadfence();
    s_last = (atomicAdd(ticket, 1) == K - 1);
  }
  __syncthreads();
  if (!s_last) return;
  __threadfence();
  // offsets = inclusive scan of the K totals (all requested at once: device-scope loads are uncached round trips)
  int carry = 0;
  for (int base = 0; base < K; base += kBkThreads) {
    const int k = base + tid;
    const int v = k < K ? __hip_atomic_load(&totals[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
    const int lane = tid & 63, wave = tid >> 6;
    int incl = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int up = __shfl_up(incl, d);
      if (lane >= d) incl += up;
    }
    __syncthreads();
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    int wave_base = 0, trip_total = 0;
#pragma unroll
    for (int w = 0; w < kBkThreads / 64; ++w) {
      const int q = s_wave[w];
      if (w < wave) wave_base += q;
      trip_total += q;
    }
    if (k < K) {
      const int o = carry + wave_base + incl;
      offsets[k + 1] = o;
      if (mirror) mirror[k + 1] = o;
    }
    carry += trip_total;
  }
  __syncthreads();
  if (tid == 0) {
    offsets[0] = 0;
    if (mirror) {
      mirror[0] = 0;
      mirror[K + 1] = status ? __hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
      // READY word, written last and behind a system-scope fence: a host that spins on it (instead of sleeping in an event
      // wait, 20-50 us of wake-up latency on this platform) sees complete offsets and flags once it reads non-zero
      __threadfence_system();
      __hip_atomic_store(&mirror[K + 2], 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

// One workgroup per 256-row tile.  Rank of a pair inside its bucket = offsets[k] + scanned count of the tile + pairs of
// lower waves + popcount(row bitmap of the offset & lower rows): no atomics, order = output row.
//
// A wave reads its 64 neighbour rows as whole 16-B pieces (8 adjacent lanes = one 128-B row): lane l then HOLDS entries
// (row l/8 + 8j, offsets 4(l%8) .. +3) for j = 0..7.  Instead of transposing them through an LDS tile (8 KB per wave: two
// workgroups per CU) every holder ranks its own entries against the per-offset row bitmaps of the wave (27 ballots of the
// mask bits, 256 B of LDS), so the only large LDS buffer is the staging area of the output.
struct KsArgs {
  const int32_t* nbr;
  const uint32_t* mask;
  int64_t m;
  int K, kp, mw;
  int64_t ntile;
  const int32_t* counts;
  const int32_t* offsets;
  int32_t* in_maps;
  int32_t* out_maps;
  int64_t pair_capacity;
  int32_t* status;
  int compact;  // nbr holds COMPACT rows (kmap_cells.h); one mask word
};
constexpr size_t kKsLds = (size_t)2 * kStageCap * 4 + 32 * 8 + (size_t)(kBkThreads / 64) * 32 * 12 + 36 * 4 + kStageCap;

__device__ __forceinline__ void kmap_scatter_body(const KsArgs& q, int64_t tile_id, char* smem) {
  const int32_t* __restrict__ nbr = q.nbr;
  const uint32_t* __restrict__ mask = q.mask;
  const int64_t m = q.m, ntile = q.ntile, pair_capacity = q.pair_capacity;
  const int K = q.K, kp = q.kp, mw = q.mw;
  const int32_t* __restrict__ counts = q.counts;
  const int32_t* __restrict__ offsets = q.offsets;
  int32_t* __restrict__ in_maps = q.in_maps;
  int32_t* __restrict__ out_maps = q.out_maps;
  int32_t* s_in = reinterpret_cast<int32_t*>(smem);
  int32_t* s_out = s_in + kStageCap;
  int64_t* s_gbase = reinterpret_cast<int64_t*>(s_out + kStageCap);  // [32] first global position of the tile's pairs of every offset
  unsigned long long(*s_ball)[32] = reinterpret_cast<unsigned long long(*)[32]>(s_gbase + 32);  // rows of the wave that have the offset
  int(*s_cnt)[32] = reinterpret_cast<int(*)[32]>(s_ball + kBkThreads / 64);  // pairs per (wave, offset), then exclusive over the waves
  int* s_seg = reinterpret_cast<int*>(s_cnt + kBkThreads / 64);               // [33] first staged position of every offset
  unsigned char* s_bk = reinterpret_cast<unsigned char*>(s_seg + 36);         // [kStageCap] offset (inside the word) of a staged pair
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t row0 = tile_id * kTileRows + wave * 64;
  const int64_t row = row0 + lane;
  bool overflow = false;
  for (int w = 0; w < mw; ++w) {
    const int kend = (K - w * 32) < 32 ? (K - w * 32) : 32;            // offsets in this mask word
    const int cols4 = ((kp - w * 32) < 32 ? (kp - w * 32) : 32) >> 2;  // 16-B chunks per row in this word
    // cols4 (<= 8) 16-B pieces per lane, all requested up front
    int4 piece[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int e = lane + 64 * j;
      const int r = e / cols4, c = e - r * cols4;
      piece[j] = make_int4(-1, -1, -1, -1);
      if (j < cols4 && row0 + r < m) piece[j] = *reinterpret_cast<const int4*>(nbr + (row0 + r) * kp + w * 32 + c * 4);
    }
    const uint32_t bits = row < m ? mask[row * mw + w] : 0u;
    // row bitmaps and pair counts of the word's offsets (lane b keeps offset w*32+b)
    unsigned long long mine = 0ull;
    for (int b = 0; b < kend; ++b) {
      const unsigned long long ball = __ballot((bits >> b) & 1u);
      if (lane == b) mine = ball;
    }
    if (lane < 32) {
      s_ball[wave][lane] = mine;
      s_cnt[wave][lane] = __popcll(mine);
    }
    __syncthreads();
    if (wave == 0) {
      int tot = 0;
      if (lane < 32) {
#pragma unroll
        for (int q = 0; q < kBkThreads / 64; ++q) {
          const int c = s_cnt[q][lane];
          s_cnt[q][lane] = tot;
          tot += c;
        }
      }
      int incl = tot;
#pragma unroll
      for (int d = 1; d < 32; d <<= 1) {
        const int up = __shfl_up(incl, d);
        if (lane >= d) incl += up;
      }
      if (lane < 32) s_seg[lane] = incl - tot;
      if (lane == 31) s_seg[32] = incl;
      if (lane < kend) s_gbase[lane] = (int64_t)offsets[w * 32 + lane] + counts[(int64_t)(w * 32 + lane) * ntile + tile_id];
    }
    __syncthreads();
    const int total = s_seg[32];
    const bool staged = total <= kStageCap;
    if ((64 % cols4) == 0) {
      // 1 / 2 / 4 / 8 pieces per row: a lane holds the SAME four table columns 4c .. 4c+3 in every piece (rows r0 + j * rstep), so
      // everything that depends on the offset alone - the wave's pair count below it, its row bitmap, the staged and the global
      // base - is read once per word instead of once per value (3 LDS reads of 4 per value were these)
      const int c = lane % cols4, r0 = lane / cols4, rstep = 64 / cols4;
      int cnt_q[4], seg_q[4];
      unsigned long long ball_q[4];
      int64_t gb_q[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int b = (c * 4 + t) & 31;
        cnt_q[t] = s_cnt[wave][b];
        ball_q[t] = s_ball[wave][b];
        seg_q[t] = s_seg[b];
        gb_q[t] = s_gbase[b];
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if (j >= cols4) break;
        const int r = r0 + j * rstep;
        const unsigned long long below = (1ull << r) - 1ull;
        const int vals[4] = {piece[j].x, piece[j].y, piece[j].z, piece[j].w};
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const int b = c * 4 + t;
          if (vals[t] < 0 || b >= kend) continue;
          const int local = cnt_q[t] + __popcll(ball_q[t] & below);
          if (staged) {
            const int at = seg_q[t] + local;
            s_in[at] = vals[t];
            s_out[at] = (int32_t)(row0 + r);
            s_bk[at] = (unsigned char)b;
          } else {
            const int64_t pos = gb_q[t] + local;
            if (pos < pair_capacity) {
              in_maps[pos] = vals[t];
              out_maps[pos] = (int32_t)(row0 + r);
            } else {
              overflow = true;
            }
          }
        }
      }
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if (j >= cols4) break;
        const int e = lane + 64 * j;
        const int r = e / cols4, c = e - r * cols4;
        const int vals[4] = {piece[j].x, piece[j].y, piece[j].z, piece[j].w};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int b = c * 4 + q;  // offset inside the word
          if (vals[q] < 0 || b >= kend) continue;
          const int local = s_cnt[wave][b] + __popcll(s_ball[wave][b] & ((1ull << r) - 1ull));
          if (staged) {
            s_in[s_seg[b] + local] = vals[q];
            s_out[s_seg[b] + local] = (int32_t)(row0 + r);
            s_bk[s_seg[b] + local] = (unsigned char)b;
          } else {
            const int64_t pos = s_gbase[b] + local;
            if (pos < pair_capacity) {
              in_maps[pos] = vals[q];
              out_maps[pos] = (int32_t)(row0 + r);
            } else {
              overflow = true;
            }
          }
        }
      }
    }
    __syncthreads();
    if (staged) {
      for (int e = tid; e < total; e += kBkThreads) {
        const int b = s_bk[e];  // (offset of staged entry e, written with it: no search over the segment starts)
        const int64_t pos = s_gbase[b] + (e - s_seg[b]);
        if (pos < pair_capacity) {
          in_maps[pos] = s_in[e];
          out_maps[pos] = s_out[e];
        } else {
          overflow = true;
        }
      }
    }
    __syncthreads();  // the next word rewrites the bitmaps and the staging area
  }
  if (overflow) atomicOr(q.status, (int)WCN_FLAG_PAIR_OVERFLOW);
}

// The same for COMPACT rows (kmap_cells.h; one mask word).  Word 4c + t of a row is the neighbour of its (4c + t - 1)-th SET
// offset.  The dense body spends 4 LDS reads and 3 LDS writes per pair (counts, row bitmap, segment start, global base; row id,
// output row, bucket) - at 4.2 M pairs that is what the kernel's time was (round 6: ~1.7 M wave-level LDS instructions, 41 us for
// a kernel that moves 100 MB).  Here a pair costs ONE 16-B read ({row bitmap of the wave, staged position of the wave's first
// pair} per (wave, offset)) and ONE 8-B write ({input row, output row inside the tile << 8 | offset}).
struct __attribute__((aligned(16))) KcSlot {
  unsigned long long ball;  // rows of the wave that have the offset
  int base;                 // staged position of the wave's first pair of the offset
  int pad;
};
constexpr size_t kKcLds = (size_t)kStageCap * 8 + (size_t)(kBkThreads / 64) * 32 * sizeof(KcSlot) + 32 * 8 + 36 * 4 + kBkThreads * 4;

__global__ __launch_bounds__(kBkThreads) void kmap_scatter_compact_kernel(KsArgs q) {
  extern __shared__ __attribute__((aligned(16))) char s_kc[];
  uint2* s_pair = reinterpret_cast<uint2*>(s_kc);                                   // [kStageCap] staged pairs
  KcSlot(*s_wb)[32] = reinterpret_cast<KcSlot(*)[32]>(s_pair + kStageCap);          // [waves][32]
  int64_t* s_delta = reinterpret_cast<int64_t*>(s_wb + kBkThreads / 64);            // [32] global position minus staged position
  int* s_seg = reinterpret_cast<int*>(s_delta + 32);                                // [33] first staged position of every offset
  uint32_t* s_rowmask = reinterpret_cast<uint32_t*>(s_seg + 36);                    // [kBkThreads]
  const int32_t* __restrict__ nbr = q.nbr;
  const int64_t m = q.m, pair_capacity = q.pair_capacity;
  const int K = q.K;
  int32_t* __restrict__ in_maps = q.in_maps;
  int32_t* __restrict__ out_maps = q.out_maps;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t tile_row0 = (int64_t)blockIdx.x * kTileRows;
  const int64_t row0 = tile_row0 + wave * 64;
  const int64_t row = row0 + lane;
  // 4 pieces per 64-B row: lane l holds piece l % 4 of rows l / 4 + 16 j - the wave's 4 KB of rows are one contiguous stream
  int4 piece[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int r = (lane >> 2) + 16 * j;
    piece[j] = make_int4(0, 0, 0, 0);
    if (row0 + r < m) piece[j] = *reinterpret_cast<const int4*>(nbr + (row0 + r) * kCompactPitch + (lane & 3) * 4);
  }
  uint32_t bits = row < m ? q.mask[row] : 0u;
  // a row that did not fit its compact row (> 15 neighbours; the build is flagged ROW_OVERFLOW and redone) stages NOTHING and
  // must not be counted either: a counted pair that is never staged leaves a stale slot in the staging area
  if (__popc(bits) > kCompactIds) bits = 0u;
  s_rowmask[tid] = bits;
  unsigned long long mine = 0ull;
  for (int b = 0; b < K; ++b) {
    const unsigned long long ball = __ballot((bits >> b) & 1u);
    if (lane == b) mine = ball;
  }
  if (lane < 32) {
    s_wb[wave][lane].ball = mine;
    s_wb[wave][lane].base = __popcll(mine);
  }
  __syncthreads();
  if (wave == 0) {
    int tot = 0, cw[kBkThreads / 64];
    if (lane < 32) {
#pragma unroll
      for (int w2 = 0; w2 < kBkThreads / 64; ++w2) {
        cw[w2] = tot;
        tot += s_wb[w2][lane].base;
      }
    }
    int incl = tot;
#pragma unroll
    for (int d = 1; d < 32; d <<= 1) {
      const int up = __shfl_up(incl, d);
      if (lane >= d) incl += up;
    }
    if (lane < 32) {
      const int seg = incl - tot;
      s_seg[lane] = seg;
#pragma unroll
      for (int w2 = 0; w2 < kBkThreads / 64; ++w2) s_wb[w2][lane].base = seg + cw[w2];
      int64_t g = 0;
      if (lane < K) g = (int64_t)q.offsets[lane] + q.counts[(int64_t)lane * q.ntile + blockIdx.x];
      s_delta[lane] = g - seg;
    }
    if (lane == 31) s_seg[32] = incl;
  }
  __syncthreads();
  const int total = s_seg[32];
  const bool staged = total <= kStageCap;
  bool overflow = false;
  const int c = lane & 3;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int r = (lane >> 2) + 16 * j;
    uint32_t rem = s_rowmask[wave * 64 + r];
    for (int sk = 4 * c - 1; sk > 0; --sk) rem &= rem - 1u;  // strip the offsets of the words in front of this piece
    const unsigned long long below = (1ull << r) - 1ull;
    const int vals[4] = {piece[j].x, piece[j].y, piece[j].z, piece[j].w};
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      if (c == 0 && t == 0) continue;  // word 0: the mask
      if (rem == 0u) break;
      const int b = __builtin_ctz(rem);
      rem &= rem - 1u;
      const uint4 slot = *reinterpret_cast<const uint4*>(&s_wb[wave][b]);
      const unsigned long long ball = ((unsigned long long)slot.y << 32) | slot.x;
      const int at = (int)slot.z + __popcll(ball & below);
      if (staged) {
        s_pair[at] = make_uint2((uint32_t)vals[t], (uint32_t)(((wave * 64 + r) << 8) | b));
      } else {
        const int64_t pos = s_delta[b] + at;
        if (pos < pair_capacity) {
          in_maps[pos] = vals[t];
          out_maps[pos] = (int32_t)(row0 + r);
        } else {
          overflow = true;
        }
      }
    }
  }
  __syncthreads();
  if (staged) {
    for (int e = tid; e < total; e += kBkThreads) {
      const uint2 pr = s_pair[e];
      const int64_t pos = s_delta[pr.y & 31u] + e;
      if (pos < pair_capacity) {
        in_maps[pos] = (int32_t)pr.x;
        out_maps[pos] = (int32_t)(tile_row0 + (pr.y >> 8));
      } else {
        overflow = true;
      }
    }
  }
  if (overflow) atomicOr(q.status, (int)WCN_FLAG_PAIR_OVERFLOW);
}

__global__ __launch_bounds__(kBkThreads) void kmap_scatter_kernel(KsArgs q) {
  extern __shared__ char s_ks[];
  kmap_scatter_body(q, blockIdx.x, s_ks);
}

static inline bool valid_k(int32_t k) { return k >= 1 && k <= 4096; }

static void launch_scatter(const KsArgs& q, hipStream_t s) {
  const dim3 grid((unsigned)ceil_div(q.m, kTileRows)), block(kBkThreads);
  if (q.compact) hipLaunchKernelGGL(kmap_scatter_compact_kernel, grid, block, kKcLds, s, q);
  else hipLaunchKernelGGL(kmap_scatter_kernel, grid, block, kKsLds, s, q);
}

// compact rows -> the dense [m, kp] table (-1 = absent): one thread per 16-B piece of a dense row
__global__ __launch_bounds__(256) void kmap_densify_kernel(const int32_t* __restrict__ nbrc, int64_t m, int kp,
                                                           int32_t* __restrict__ nbr) {
  const int cols4 = kp >> 2;
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= m * cols4) return;
  const int64_t row = e / cols4;
  const int c = (int)(e - row * cols4);
  const int32_t* src = nbrc + row * kCompactPitch;
  uint32_t bits = (uint32_t)src[0];
  if (__popc(bits) > kCompactIds) bits = 0u;
  int at = 1 + __popc(bits & ((1u << (4 * c)) - 1u));
  int v[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const bool has = (bits >> (4 * c + t)) & 1u;
    v[t] = has ? src[at] : -1;
    at += has ? 1 : 0;
  }
  *reinterpret_cast<int4*>(nbr + row * kp + 4 * c) = make_int4(v[0], v[1], v[2], v[3]);
}

}  // namespace wcn

using namespace wcn;

extern "C" {

// rounded up to a multiple of 4 so that every offset's row of the counts array is 16-B aligned (vector access in the scan)
int64_t wcn_kmap_num_blocks(int64_t m) { return (ceil_div(m > 0 ? m : 0, kTileRows) + 3) & ~(int64_t)3; }

size_t wcn_kmap_counts_bytes(int64_t m, int32_t num_offsets) {
  // K rows of counts + K totals + the ticket word (+ slack)
  return ((size_t)num_offsets * (size_t)(wcn_kmap_num_blocks(m) + 1) + 64) * 4;
}

static void launch_tally(uint32_t* mask, int32_t* nbr, int64_t m, int K, int32_t* counts, bool hist, int nblk_sort,
                         int32_t* dcounts, const int32_t* coords, const CellTable* cells, int32_t* status, int kc, int sort_shift,
                         int sort_bits, hipStream_t s, bool compact = false) {
  const int kp = compact ? (kCompactPitch | kCompactFlag) : wcn_kmap_row_pitch(K), mw = wcn_kmap_mask_words(K);
  const int64_t ntile = wcn_kmap_num_blocks(m);
  int32_t* ticket = counts + (int64_t)K * (ntile + 1);
  const dim3 grid((unsigned)ceil_div(m, kRsTile)), block(kTallyThreads);
  CellTable none{};
  const CellTable& t = cells ? *cells : none;
  const uint32_t cmask = cells ? (uint32_t)(cells->capacity - 1) : 0u;
#define WCN_TALLY(H, R)                                                                                                \
  hipLaunchKernelGGL((kmap_tally_kernel<H, R>), grid, block, 0, s, mask, nbr, m, K, kp, mw, ntile, counts, ticket,       \
                     nblk_sort, dcounts, (const int4*)coords, t, cmask, status, kc, sort_shift, sort_bits)
  if (hist && cells) WCN_TALLY(true, true);
  else if (hist) WCN_TALLY(true, false);
  else if (cells) WCN_TALLY(false, true);
  else WCN_TALLY(false, false);
#undef WCN_TALLY
}

static void launch_scan(int32_t* counts, int64_t ntile, int K, int32_t* offsets, const int32_t* status, int32_t* mirror,
                        int nblk_sort, int32_t* dcounts, int32_t* dtotals, int sort_bins, hipStream_t s) {
  int32_t* totals = counts + (int64_t)K * ntile;
  int32_t* ticket = totals + K;
  hipLaunchKernelGGL(kmap_scan_kernel, dim3((unsigned)(K + (nblk_sort > 0 ? (sort_bins + kRsScanCols - 1) / kRsScanCols : 0))), dim3(kBkThreads), 0,
                     s, counts, ntile, K, totals, ticket, offsets, status, mirror, dcounts, nblk_sort, dtotals, sort_bins);
}

int wcn_kmap_count(const uint32_t* mask, int64_t m, int32_t num_offsets, int32_t* counts, wcn_stream_t stream) {
  if (m < 0 || !valid_k(num_offsets)) return WCN_ERROR_INVALID_PARAMETERS;
  if (m == 0) return WCN_SUCCESS;
  if (!mask || !counts) return WCN_ERROR_INVALID_PARAMETERS;
  launch_tally(const_cast<uint32_t*>(mask), nullptr, m, num_offsets, counts, false, 0, nullptr, nullptr, nullptr, nullptr, 0, 0,
               kRsBits, (hipStream_t)stream);
  return launch_status();
}

// `counts` as written by wcn_kmap_count for the same map (the count launch also arms the ticket word behind the totals)
static int scan_impl(int32_t* counts, int64_t num_blocks, int32_t K, int32_t* offsets, const int32_t* status,
                     int32_t* mirror, wcn_stream_t stream) {
  if (num_blocks < 0 || (num_blocks & 3) || !valid_k(K) || !offsets || !counts) return WCN_ERROR_INVALID_PARAMETERS;
  launch_scan(counts, num_blocks, K, offsets, status, mirror, 0, nullptr, nullptr, 0, (hipStream_t)stream);
  return launch_status();
}

int wcn_kmap_scan(int32_t* counts, int64_t num_blocks, int32_t num_offsets, int32_t* offsets, wcn_stream_t stream) {
  return scan_impl(counts, num_blocks, num_offsets, offsets, nullptr, nullptr, stream);
}

int wcn_kmap_scan_to_host(int32_t* counts, int64_t num_blocks, int32_t num_offsets, int32_t* offsets, const int32_t* status,
                          int32_t* host_mirror, wcn_stream_t stream) {
  if (!host_mirror) return WCN_ERROR_INVALID_PARAMETERS;
  return scan_impl(counts, num_blocks, num_offsets, offsets, status, host_mirror, stream);
}

size_t wcn_kmap_tally_sort_workspace(int64_t m) { return wcn_mask_argsort_workspace(m); }

static KsArgs ks_args(const int32_t* nbr, const uint32_t* mask, int64_t m, int32_t K, const int32_t* counts,
                      const int32_t* offsets, int32_t* in_maps, int32_t* out_maps, int64_t pair_capacity, int32_t* status,
                      int compact);

int wcn_kmap_tally_sort(uint32_t* mask, int32_t* nbr, int64_t m, int32_t num_offsets, int32_t* counts, int32_t* offsets,
                        int32_t* status, int32_t* host_mirror, int32_t* perm, void* sort_workspace,
                        size_t sort_workspace_bytes, const int32_t* coords, void* binned_workspace, int64_t binned_n,
                        int64_t max_blocks, int32_t compact, int32_t* in_maps, int32_t* out_maps, int64_t pair_capacity,
                        wcn_stream_t stream) {
  if (m < 0 || !valid_k(num_offsets) || !counts || !offsets || !status) return WCN_ERROR_INVALID_PARAMETERS;
  if (pair_capacity < 0 || ((in_maps == nullptr) != (out_maps == nullptr))) return WCN_ERROR_INVALID_PARAMETERS;
  if (compact && !wcn_kmap_compact_supported(num_offsets)) return WCN_ERROR_INVALID_PARAMETERS;
  hipStream_t s = (hipStream_t)stream;
  if (m == 0) {  // no rows: offsets are all zero, nothing to sort
    if (hipMemsetAsync(offsets, 0, (size_t)(num_offsets + 1) * 4, s) != hipSuccess) return WCN_ERROR_KERNEL_EXECUTION;
    if (host_mirror) {  // (the READY word stays clear: the caller of an empty build waits for its event)
      if (hipMemsetAsync(host_mirror, 0, (size_t)(num_offsets + 1) * 4, s) != hipSuccess ||
          hipMemcpyAsync(host_mirror + num_offsets + 1, status, 4, hipMemcpyDeviceToHost, s) != hipSuccess)
        return WCN_ERROR_KERNEL_EXECUTION;
    }
    return WCN_SUCCESS;
  }
  if (m >= (1ll << 31) || !mask || !nbr || !perm || !sort_workspace ||
      sort_workspace_bytes < wcn_kmap_tally_sort_workspace(m))
    return WCN_ERROR_INVALID_PARAMETERS;
  if (binned_workspace && (!coords || binned_n != m || max_blocks < 1)) return WCN_ERROR_INVALID_PARAMETERS;
  // rows ordered for the gather GEMMs' tiles: tile_key (mask_sort.h) for odd kernel volumes up to 31, else descending mask
  const int kc = tile_key_centre(num_offsets, wcn_kmap_mask_words(num_offsets));
  const SortPlan plan = sort_plan(sort_workspace, m, num_offsets < 32 ? num_offsets : 32, kc > 0);
  CellTable cells{};
  if (binned_workspace) cells = carve_cells(binned_workspace, binned_n, max_blocks);
  launch_tally(mask, nbr, m, num_offsets, counts, true, plan.nblk, plan.counts, coords, binned_workspace ? &cells : nullptr,
               status, kc, plan.shift0, plan.bits, s, compact != 0);
  launch_scan(counts, wcn_kmap_num_blocks(m), num_offsets, offsets, status, host_mirror, plan.nblk, plan.counts,
              plan.totals, 1 << plan.bits, s);
  RsLaunch l[12];
  const int count = sort_launches(plan, mask, wcn_kmap_mask_words(num_offsets), m, perm, true, kc, l);
  sort_run_range(l, 0, count, s);
  if (in_maps && pair_capacity > 0) {
    // the pair lists right behind the sort, in the same call (one C call per build instead of two).  Round 6 also measured the
    // scatter INSIDE the sort's launches (workgroups [0, nsort) of every sort launch in the sort role, the rest pair-scatter
    // tiles): 82.8 us for the four launches against 77.7 us for sort + scatter behind each other - a scatter workgroup lives
    // ~16 us (2.5 rounds of 3 907 workgroups), so every launch that carries some lasts that long, and both kinds of
    // workgroup are bound by the rate of scattered store requests, which does not overlap (OPTIMISATION_LOG appendix G)
    const KsArgs q = ks_args(nbr, mask, m, num_offsets, counts, offsets, in_maps, out_maps, pair_capacity, status, compact ? 1 : 0);
    launch_scatter(q, s);
  }
  return launch_status();
}

static KsArgs ks_args(const int32_t* nbr, const uint32_t* mask, int64_t m, int32_t K, const int32_t* counts,
                      const int32_t* offsets, int32_t* in_maps, int32_t* out_maps, int64_t pair_capacity, int32_t* status,
                      int compact) {
  KsArgs q;
  q.compact = compact;
  q.nbr = nbr; q.mask = mask; q.m = m; q.K = K; q.kp = wcn_kmap_row_pitch(K); q.mw = wcn_kmap_mask_words(K);
  q.ntile = wcn_kmap_num_blocks(m); q.counts = counts; q.offsets = offsets; q.in_maps = in_maps; q.out_maps = out_maps;
  q.pair_capacity = pair_capacity; q.status = status;
  return q;
}

int wcn_kmap_densify(const int32_t* nbr_compact, int64_t m, int32_t num_offsets, int32_t* nbr, wcn_stream_t stream) {
  if (m < 0 || !wcn_kmap_compact_supported(num_offsets)) return WCN_ERROR_INVALID_PARAMETERS;
  if (m == 0) return WCN_SUCCESS;
  if (!nbr_compact || !nbr) return WCN_ERROR_INVALID_PARAMETERS;
  const int kp = wcn_kmap_row_pitch(num_offsets);
  hipLaunchKernelGGL(kmap_densify_kernel, dim3((unsigned)ceil_div(m * (kp >> 2), 256)), dim3(256), 0, (hipStream_t)stream,
                     nbr_compact, m, kp, nbr);
  return launch_status();
}

int wcn_kmap_scatter(const int32_t* nbr, const uint32_t* mask, int64_t m, int32_t num_offsets, const int32_t* counts,
                     const int32_t* offsets, int32_t* in_maps, int32_t* out_maps, int64_t pair_capacity, int32_t* status,
                     int32_t compact, wcn_stream_t stream) {
  if (m < 0 || !valid_k(num_offsets) || pair_capacity < 0 || !status) return WCN_ERROR_INVALID_PARAMETERS;
  if (compact && !wcn_kmap_compact_supported(num_offsets)) return WCN_ERROR_INVALID_PARAMETERS;
  if (m == 0) return WCN_SUCCESS;
  if (!nbr || !mask || !counts || !offsets || (pair_capacity > 0 && (!in_maps || !out_maps)))
    return WCN_ERROR_INVALID_PARAMETERS;
  const KsArgs q = ks_args(nbr, mask, m, num_offsets, counts, offsets, in_maps, out_maps, pair_capacity, status, compact ? 1 : 0);
  launch_scatter(q, (hipStream_t)stream);
  return launch_status();
}

}  // extern "C"
