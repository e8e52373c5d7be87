// kmap_stride.hip - strided layers answered from the block-hashed CELL TABLE of the fine coordinate set (kmap_cells.h),
// which the submanifold layers of the same resolution level have already built (kmap_binned.hip) - no global hash table:
//
//   wcn_cells_stride_count / _emit   coordinate down-sampling: out = distinct floor(in / stride), rows in order of their FIRST
//                                    occurrence in the input (the contract of coords/ops/stride.py).  A coarse cell is
//                                    stride_x * stride_y * stride_z cells of ONE 8^3 block (strides 1, 2, 4, 8), so "is row i the
//                                    first of its coarse cell" = "is i the smallest row id among those cells": one block
//                                    lookup + a few adjacent cell reads per voxel, no insert, no probe sequence; survivors are
//                                    compacted in row order (ballots + one scan).  With kernel_size == stride the kernel map of
//                                    the strided convolution is those very cells: emitted in the same pass.
//   wcn_kmap_probe_cells             any other map between two coordinate sets (kernel 3, stride 2; transposed layers that
//                                    find no cached forward map): one lane per (output row, offset) as in wcn_kmap_probe, but
//                                    the probe is a block-table lookup (2 MB per million voxels: L2-resident) + one 4-B cell
//                                    read instead of a probe sequence in a 32 MB slot table, and nothing is inserted.
// Both require the table's owner to have been validated (no TABLE_FULL; duplicate coordinates resolved to the smallest row -
// strict build or a plain build the tally pass accepted), which the Python side guarantees.
//
// Replaces, for these layers: wcn_hash_insert + wcn_hash_search + the torch op chain of unique_first_indices_with_offsets
// (utils/unique.py) and wcn_hash_insert + wcn_kmap_probe.  Reference: warpconvnet/geometry/coords/ops/stride.py:18-56,
// csrc/cuhash_kernel_map.cu:93-134; coarse-to-fine idea geometry/coords/search/hierarchical_search.py:25-66.
#include "/tmp/prevhdr/kmap_cells.h"

namespace wcn {

constexpr int kStTile = 256;  // rows per compaction tile (4 waves)

struct StrideGeom {
  int lx, ly, lz;  // log2 of the stride per axis (0..3)
};

__device__ __forceinline__ int block_id_of(const BSlot* __restrict__ slots, uint32_t cmask, const int4& c) {
  const int s = block_find(slots, cmask, pack_key(c.x, c.y >> kBlkShift, c.z >> kBlkShift, c.w >> kBlkShift));
  if (s < 0) return -1;
  const int id = slots[s].id;
  return id < 0 ? -1 : (id & ~kIdLateBit);
}

// smallest row id among the cells of the coarse cell that holds fine cell (x, y, z) of block `id`
__device__ __forceinline__ int coarse_min_row(const int32_t* __restrict__ cells, int id, int x, int y, int z,
                                              const StrideGeom& g) {
  const int x0 = (x >> g.lx) << g.lx, y0 = (y >> g.ly) << g.ly, z0 = (z >> g.lz) << g.lz;
  const int32_t* sub = cells + (int64_t)id * kCells;
  int mn = 0x7FFFFFFF;
  for (int dx = 0; dx < (1 << g.lx); ++dx)
    for (int dy = 0; dy < (1 << g.ly); ++dy)
      for (int dz = 0; dz < (1 << g.lz); ++dz) {
        const int v = sub[((x0 + dx) * kBlk + (y0 + dy)) * kBlk + (z0 + dz)];
        if (v >= 0 && v < mn) mn = v;
      }
  return mn;
}

// flags[w] = ballot over rows 64 w .. 64 w + 63 of "first row of its coarse cell"; counts[t] = survivors of tile t
__global__ __launch_bounds__(kStTile) void stride_first_kernel(const BSlot* __restrict__ slots, uint32_t cmask, CellTable t,
                                                               const int4* __restrict__ coords, int64_t n, StrideGeom g,
                                                               unsigned long long* __restrict__ flags,
                                                               int32_t* __restrict__ counts) {
  __shared__ int s_c[kStTile / 64];
  const int64_t i = (int64_t)blockIdx.x * kStTile + threadIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  bool first = false;
  if (i < n) {
    const int4 c = coords[i];
    const int id = block_id_of(slots, cmask, c);
    if (id >= 0) first = coarse_min_row(t.cells, id, c.y & (kBlk - 1), c.z & (kBlk - 1), c.w & (kBlk - 1), g) == (int)i;
  }
  const unsigned long long ball = __ballot(first);
  if (lane == 0) {
    if ((int64_t)blockIdx.x * kStTile + wave * 64 < n) flags[(int64_t)blockIdx.x * (kStTile / 64) + wave] = ball;
    s_c[wave] = __popcll(ball);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    int tot = 0;
    for (int w = 0; w < kStTile / 64; ++w) tot += s_c[w];
    counts[blockIdx.x] = tot;
  }
}

// one workgroup: exclusive scan of the tile counts in place (counts[ntile] = total), then the survivors in front of every
// batch boundary: out_offsets[b] = survivors among the rows of batch indices < b (rows are batch-sorted: the boundary is
// found by bisection on the batch column, so the host uploads nothing)
__global__ __launch_bounds__(256) void stride_scan_kernel(int32_t* __restrict__ counts, int64_t ntile,
                                                          const unsigned long long* __restrict__ flags, int64_t n,
                                                          const int4* __restrict__ coords, int num_batches,
                                                          int32_t* __restrict__ out_offsets) {
  __shared__ int s_w[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int carry = 0;
  for (int64_t base = 0; base < ntile; base += 256 * 8) {
    int v[8], sum = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int64_t e = base + (int64_t)tid * 8 + j;
      v[j] = e < ntile ? counts[e] : 0;
      sum += v[j];
    }
    int incl = sum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int up = __shfl_up(incl, d);
      if (lane >= d) incl += up;
    }
    if (lane == 63) s_w[wave] = incl;
    __syncthreads();
    int wbase = 0, trip = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      if (w < wave) wbase += s_w[w];
      trip += s_w[w];
    }
    int run = carry + wbase + incl - sum;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int64_t e = base + (int64_t)tid * 8 + j;
      if (e < ntile) counts[e] = run;
      run += v[j];
    }
    carry += trip;
    __syncthreads();
  }
  if (tid == 0) counts[ntile] = carry;
  __syncthreads();
  __threadfence_block();
  for (int b = tid; b <= num_batches; b += 256) {
    int64_t lo = 0, hi = n;  // first row whose batch index is >= b
    while (lo < hi) {
      const int64_t mid = (lo + hi) >> 1;
      if (coords[mid].x < b) lo = mid + 1; else hi = mid;
    }
    const int64_t r = b >= num_batches ? n : lo;
    int v;
    if (r >= n) {
      v = carry;
    } else {
      const int64_t tile = r / kStTile, w0 = tile * (kStTile / 64), w1 = r >> 6;
      v = counts[tile];
      for (int64_t w = w0; w < w1; ++w) v += __popcll(flags[w]);
      v += __popcll(flags[w1] & ((1ull << (r & 63)) - 1ull));
    }
    out_offsets[b] = v;
  }
}

// survivors -> out_coords[pos] = (b, x >> lx, y >> ly, z >> lz); with nbr != null also the map of a convolution whose kernel
// is the stride window (kernel_size == stride, dilation 1): nbr[pos][k] = row in cell (x0 + i, y0 + j, z0 + l), k = (i*sy + j)*sz + l
__global__ __launch_bounds__(kStTile) void stride_emit_kernel(const BSlot* __restrict__ slots, uint32_t cmask, CellTable t,
                                                              const int4* __restrict__ coords, int64_t n, StrideGeom g,
                                                              const unsigned long long* __restrict__ flags,
                                                              const int32_t* __restrict__ counts, int4* __restrict__ out_coords,
                                                              int32_t* __restrict__ first_rows, int K, int kp,
                                                              int32_t* __restrict__ nbr, uint32_t* __restrict__ mask) {
  const int64_t i = (int64_t)blockIdx.x * kStTile + threadIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if ((int64_t)blockIdx.x * kStTile + wave * 64 >= n) return;
  const unsigned long long* fw = flags + (int64_t)blockIdx.x * (kStTile / 64);
  const unsigned long long mine = fw[wave];
  if (!((mine >> lane) & 1ull)) return;
  int pos = counts[blockIdx.x] + __popcll(mine & ((1ull << lane) - 1ull));
  for (int w = 0; w < wave; ++w) pos += __popcll(fw[w]);
  const int4 c = coords[i];
  out_coords[pos] = make_int4(c.x, c.y >> g.lx, c.z >> g.ly, c.w >> g.lz);
  if (first_rows) first_rows[pos] = (int32_t)i;
  if (!nbr) return;
  const int id = block_id_of(slots, cmask, c);
  const int x0 = ((c.y & (kBlk - 1)) >> g.lx) << g.lx, y0 = ((c.z & (kBlk - 1)) >> g.ly) << g.ly,
            z0 = ((c.w & (kBlk - 1)) >> g.lz) << g.lz;
  const int32_t* sub = t.cells + (int64_t)id * kCells;
  uint32_t bits = 0u;
  int k = 0;
  for (int dx = 0; dx < (1 << g.lx); ++dx)
    for (int dy = 0; dy < (1 << g.ly); ++dy)
      for (int dz = 0; dz < (1 << g.lz); ++dz, ++k) {
        const int v = sub[((x0 + dx) * kBlk + (y0 + dy)) * kBlk + (z0 + dz)];
        nbr[(int64_t)pos * kp + k] = v;
        if (v >= 0 && k < 32) bits |= 1u << k;
      }
  for (; k < kp; ++k) nbr[(int64_t)pos * kp + k] = -1;
  mask[pos] = bits;
}

struct CpGeom {
  int kx, ky, kz, cx, cy, cz, sx, sy, sz, dx, dy, dz;
};

// one lane per (output row, offset): in = out * stride + offset, looked up in the cell table
template <int LPR>
__global__ __launch_bounds__(256) void cells_probe_kernel(const BSlot* __restrict__ slots, uint32_t cmask, CellTable t,
                                                          const int4* __restrict__ query, int64_t m, CpGeom g, int K,
                                                          int kp, int mw, int32_t* __restrict__ nbr,
                                                          uint32_t* __restrict__ mask) {
  constexpr int kRowsPerIter = 64 / LPR;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int sub = lane % LPR, rsel = lane / LPR;
  const int64_t wave_row0 = (int64_t)blockIdx.x * 256 + wave * 64;
  const int num_chunks = (kp + LPR - 1) / LPR;
  for (int kc = 0; kc < num_chunks; ++kc) {
    const int k = kc * LPR + sub;
    const bool k_real = k < K, k_store = k < kp;
    const int l = k % g.kz, j = (k / g.kz) % g.ky, i = k / (g.kz * g.ky);
    const int ox = (i - g.cx) * g.dx, oy = (j - g.cy) * g.dy, oz = (l - g.cz) * g.dz;
#pragma unroll 4
    for (int it = 0; it < 64 / kRowsPerIter; ++it) {
      const int64_t row = wave_row0 + it * kRowsPerIter + rsel;
      int found = -1;
      if (row < m && k_real) {
        const int4 q = query[row];
        // the 18-bit wrap of the packed key, as the hash path applies it
        const int x = ((q.y * g.sx + ox) << (32 - kCoordBits)) >> (32 - kCoordBits);
        const int y = ((q.z * g.sy + oy) << (32 - kCoordBits)) >> (32 - kCoordBits);
        const int z = ((q.w * g.sz + oz) << (32 - kCoordBits)) >> (32 - kCoordBits);
        const int id = block_id_of(slots, cmask, make_int4(q.x, x, y, z));
        if (id >= 0)
          found = t.cells[(int64_t)id * kCells + (((x & (kBlk - 1)) * kBlk + (y & (kBlk - 1))) * kBlk + (z & (kBlk - 1)))];
      }
      if (row < m && k_store) nbr[row * kp + k] = found;
      const unsigned long long ball = __ballot(found >= 0);
      if (row < m && sub == 0) {
        const unsigned long long bits = (LPR == 64) ? ball : ((ball >> (rsel * LPR)) & ((1ull << (LPR & 63)) - 1ull));
        const int w0 = (kc * LPR) >> 5;
        if (w0 < mw) mask[row * mw + w0] = (uint32_t)bits;
        if (LPR == 64 && w0 + 1 < mw) mask[row * mw + w0 + 1] = (uint32_t)(bits >> 32);
      }
    }
  }
}

static inline int log2_stride(int s) { return s == 1 ? 0 : s == 2 ? 1 : s == 4 ? 2 : s == 8 ? 3 : -1; }

}  // namespace wcn

using namespace wcn;

extern "C" {

int wcn_cells_stride_supported(const int32_t stride[3]) {
  if (!stride) return 0;
  for (int d = 0; d < 3; ++d)
    if (log2_stride(stride[d]) < 0) return 0;
  return 1;
}

int64_t wcn_cells_stride_tiles(int64_t n) { return ceil_div(n > 0 ? n : 0, kStTile); }

int wcn_cells_stride_count(const void* cells_workspace, int64_t n, int64_t max_blocks, const int32_t* coords,
                           const int32_t stride[3], uint64_t* flags, int32_t* counts, int32_t num_batches,
                           int32_t* out_offsets, wcn_stream_t stream) {
  if (n < 0 || max_blocks < 1 || num_batches < 0 || !wcn_cells_stride_supported(stride)) return WCN_ERROR_INVALID_PARAMETERS;
  if (!cells_workspace || !flags || !counts || !out_offsets || (n > 0 && !coords) || n >= (1ll << 31))
    return WCN_ERROR_INVALID_PARAMETERS;
  hipStream_t s = (hipStream_t)stream;
  const CellTable t = carve_cells(const_cast<void*>(cells_workspace), n, max_blocks);
  const StrideGeom g{log2_stride(stride[0]), log2_stride(stride[1]), log2_stride(stride[2])};
  const int64_t ntile = wcn_cells_stride_tiles(n);
  if (ntile > 0)
    hipLaunchKernelGGL(stride_first_kernel, dim3((unsigned)ntile), dim3(kStTile), 0, s, (const BSlot*)t.slots,
                       (uint32_t)(t.capacity - 1), t, (const int4*)coords, n, g, (unsigned long long*)flags, counts);
  hipLaunchKernelGGL(stride_scan_kernel, dim3(1), dim3(256), 0, s, counts, ntile, (const unsigned long long*)flags, n,
                     (const int4*)coords, (int)num_batches, out_offsets);
  return launch_status();
}

int wcn_cells_stride_emit(const void* cells_workspace, int64_t n, int64_t max_blocks, const int32_t* coords,
                          const int32_t stride[3], const uint64_t* flags, const int32_t* counts, int32_t* out_coords,
                          int32_t* first_rows, int32_t* nbr, uint32_t* mask, wcn_stream_t stream) {
  if (n < 0 || max_blocks < 1 || !wcn_cells_stride_supported(stride)) return WCN_ERROR_INVALID_PARAMETERS;
  if (n == 0) return WCN_SUCCESS;
  if (!cells_workspace || !flags || !counts || !coords || !out_coords || n >= (1ll << 31) || ((nbr == nullptr) != (mask == nullptr)))
    return WCN_ERROR_INVALID_PARAMETERS;
  const CellTable t = carve_cells(const_cast<void*>(cells_workspace), n, max_blocks);
  const StrideGeom g{log2_stride(stride[0]), log2_stride(stride[1]), log2_stride(stride[2])};
  const int K = stride[0] * stride[1] * stride[2];
  if (nbr && K > 32) return WCN_ERROR_PROBLEM_NOT_SUPPORTED;  // (windows above 32 cells: the probe entry point)
  hipLaunchKernelGGL(stride_emit_kernel, dim3((unsigned)wcn_cells_stride_tiles(n)), dim3(kStTile), 0, (hipStream_t)stream,
                     (const BSlot*)t.slots, (uint32_t)(t.capacity - 1), t, (const int4*)coords, n, g,
                     (const unsigned long long*)flags, counts, (int4*)out_coords, first_rows, K, (int)wcn_kmap_row_pitch(K), nbr,
                     mask);
  return launch_status();
}

int wcn_kmap_probe_cells(const void* cells_workspace, int64_t n_in, int64_t max_blocks, const int32_t* query, int64_t m,
                         const int32_t ksize[3], const int32_t stride[3], const int32_t dilation[3], int32_t* nbr,
                         uint32_t* mask, wcn_stream_t stream) {
  if (n_in < 0 || m < 0 || max_blocks < 1 || !ksize || !stride || !dilation) return WCN_ERROR_INVALID_PARAMETERS;
  for (int d = 0; d < 3; ++d)
    if (ksize[d] < 1 || stride[d] < 1 || dilation[d] < 1) return WCN_ERROR_INVALID_PARAMETERS;
  const int64_t K64 = (int64_t)ksize[0] * ksize[1] * ksize[2];
  if (K64 > 4096) return WCN_ERROR_PROBLEM_NOT_SUPPORTED;
  if (m == 0) return WCN_SUCCESS;
  if (!cells_workspace || !query || !nbr || !mask) return WCN_ERROR_INVALID_PARAMETERS;
  const int K = (int)K64, kp = wcn_kmap_row_pitch(K), mw = wcn_kmap_mask_words(K);
  const CellTable t = carve_cells(const_cast<void*>(cells_workspace), n_in, max_blocks);
  CpGeom g;
  g.kx = ksize[0]; g.ky = ksize[1]; g.kz = ksize[2];
  g.cx = (g.kx & 1) ? g.kx / 2 : 0; g.cy = (g.ky & 1) ? g.ky / 2 : 0; g.cz = (g.kz & 1) ? g.kz / 2 : 0;
  g.sx = stride[0]; g.sy = stride[1]; g.sz = stride[2];
  g.dx = dilation[0]; g.dy = dilation[1]; g.dz = dilation[2];
  const dim3 grid((unsigned)ceil_div(m, 256)), block(256);
  hipStream_t s = (hipStream_t)stream;
  const BSlot* slots = (const BSlot*)t.slots;
  const uint32_t cmask = (uint32_t)(t.capacity - 1);
  int lpr = 8;
  while (lpr < kp && lpr < 64) lpr <<= 1;
  switch (lpr) {
    case 8: hipLaunchKernelGGL(cells_probe_kernel<8>, grid, block, 0, s, slots, cmask, t, (const int4*)query, m, g, K, kp, mw, nbr, mask); break;
    case 16: hipLaunchKernelGGL(cells_probe_kernel<16>, grid, block, 0, s, slots, cmask, t, (const int4*)query, m, g, K, kp, mw, nbr, mask); break;
    case 32: hipLaunchKernelGGL(cells_probe_kernel<32>, grid, block, 0, s, slots, cmask, t, (const int4*)query, m, g, K, kp, mw, nbr, mask); break;
    default: hipLaunchKernelGGL(cells_probe_kernel<64>, grid, block, 0, s, slots, cmask, t, (const int4*)query, m, g, K, kp, mw, nbr, mask); break;
  }
  return launch_status();
}

}  // extern "C"
