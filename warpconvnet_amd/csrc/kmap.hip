// kmap.hip - kernel-map construction for gfx950 (wave64).
//
//   hash table  : 16-B slots, Splitmix64, linear probing, duplicates keep the smallest row index
//   probe       : one LANE per (output row, kernel offset); a 32-lane group covers one row so the
//                 neighbour row is written as one contiguous 128-B line and the row's neighbour mask
//                 is a single wave ballot
//   bucketing   : kmap_bucket.hip
//
// Reference behaviour being replaced (semantics only, nothing copied):
//   warpconvnet/csrc/cuhash_hash_table.cu:19-262, cuhash_kernel_map.cu:68-134, 508-599,
//   mask_data_kernels.cu:23-124, 187-220.
#include "wcn_common.h"

namespace wcn {

constexpr int kBlockRows = 256;  // rows per probe workgroup
constexpr int kThreads = 256;

// ------------------------------------------------------------------------------------------------
// hash table
// ------------------------------------------------------------------------------------------------
__global__ void hash_prepare_kernel(uint4* __restrict__ slots, int64_t capacity) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < capacity) slots[i] = make_uint4(0u, 0u, 0xFFFFFFFFu, 0u);
}

__global__ void hash_insert_kernel(Slot* __restrict__ slots, uint32_t capacity_mask, const int4* __restrict__ coords,
                                   int64_t n, int32_t* __restrict__ status) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int4 c = coords[i];
  if (!coord_in_range(c.x, c.y, c.z, c.w)) {
    atomicOr(status, (int)WCN_FLAG_COORD_RANGE);
    return;
  }
  const uint64_t key = pack_key(c.x, c.y, c.z, c.w);
  uint32_t s = hash_slot(key, capacity_mask);
  for (uint32_t attempts = 0; attempts <= capacity_mask; ++attempts) {
    unsigned long long* kp = reinterpret_cast<unsigned long long*>(&slots[s].key);
    const unsigned long long prev = atomicCAS(kp, 0ull, (unsigned long long)key);
    if (prev == 0ull || prev == key) {
      // unsigned min: the empty marker 0xFFFFFFFF loses against every row index
      atomicMin(reinterpret_cast<unsigned int*>(&slots[s].value), (unsigned int)i);
      if (prev == key) atomicOr(status, (int)WCN_FLAG_DUPLICATE_COORD);
      return;
    }
    s = (s + 1) & capacity_mask;
  }
  atomicOr(status, (int)WCN_FLAG_TABLE_FULL);
}

__global__ void hash_search_kernel(const Slot* __restrict__ slots, uint32_t capacity_mask,
                                   const int4* __restrict__ queries, int64_t m, int32_t* __restrict__ results) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m) return;
  const int4 q = queries[i];
  results[i] = slot_lookup(slots, capacity_mask, pack_key(q.x, q.y, q.z, q.w));
}

// ------------------------------------------------------------------------------------------------
// probe: nbr[m][kp], mask[m][mw], block_counts[blk][K]
// ------------------------------------------------------------------------------------------------
struct ProbeGeom {
  int kx, ky, kz;  // kernel size
  int cx, cy, cz;  // centre
  int sx, sy, sz;  // stride (query = out * stride)
  int dx, dy, dz;  // dilation
};

template <int LPR>  // lanes per row: 8, 16, 32 or 64
__global__ __launch_bounds__(kThreads) void kmap_probe_kernel(const Slot* __restrict__ slots, uint32_t capacity_mask,
                                                              const int4* __restrict__ query, int64_t m, ProbeGeom g,
                                                              int K, int kp, int mw, int32_t* __restrict__ nbr,
                                                              uint32_t* __restrict__ mask) {
  const int tid = threadIdx.x;
  constexpr int kRowsPerIter = 64 / LPR;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int sub = lane % LPR;
  const int rsel = lane / LPR;
  const int64_t wave_row0 = (int64_t)blockIdx.x * kBlockRows + wave * 64;
  const int num_chunks = (kp + LPR - 1) / LPR;

  for (int kc = 0; kc < num_chunks; ++kc) {
    const int k = kc * LPR + sub;
    const bool k_real = k < K;
    const bool k_store = k < kp;
    // k = (i*ky + j)*kz + l  ->  offset (i-cx, j-cy, l-cz) * dilation
    const int l = k % g.kz;
    const int j = (k / g.kz) % g.ky;
    const int i = k / (g.kz * g.ky);
    const int ox = (i - g.cx) * g.dx, oy = (j - g.cy) * g.dy, oz = (l - g.cz) * g.dz;
#pragma unroll 4
    for (int it = 0; it < 64 / kRowsPerIter; ++it) {
      const int64_t row = wave_row0 + it * kRowsPerIter + rsel;
      int found = -1;
      if (row < m && k_real) {
        const int4 q = query[row];
        const uint64_t key = pack_key(q.x, q.y * g.sx + ox, q.z * g.sy + oy, q.w * g.sz + oz);
        found = slot_lookup(slots, capacity_mask, key);
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

// [n, d] int32 coordinates + batch offsets -> [n, d+1] with the batch index in column 0 (one launch instead of
// repeat_interleave + fill + cat; reference: warpconvnet/geometry/coords/ops/batch_index.py:90-148)
__global__ __launch_bounds__(256) void batch_indexed_coords_kernel(const int32_t* __restrict__ coords, int64_t n, int d,
                                                                   const int32_t* __restrict__ offsets, int num_batches,
                                                                   int32_t* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int b = 0;
  if (num_batches > 1) {  // last batch whose offset is <= i
    int lo = 0, hi = num_batches;
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if ((int64_t)offsets[mid] <= i) lo = mid; else hi = mid;
    }
    b = lo;
  }
  if (d == 3) {
    const int32_t* c = coords + i * 3;
    *reinterpret_cast<int4*>(out + i * 4) = make_int4(b, c[0], c[1], c[2]);
  } else {
    out[i * (d + 1)] = b;
    for (int j = 0; j < d; ++j) out[i * (d + 1) + 1 + j] = coords[i * d + j];
  }
}

// Z-order codes.  spread3: bit i of a 21-bit value -> bit 3i (the usual shift-and-mask ladder).
__device__ __forceinline__ uint64_t spread3(uint64_t v) {
  v &= 0x1FFFFFull;
  v = (v | (v << 32)) & 0x001F00000000FFFFull;
  v = (v | (v << 16)) & 0x001F0000FF0000FFull;
  v = (v | (v << 8)) & 0x100F00F00F00F00Full;
  v = (v | (v << 4)) & 0x10C30C30C30C30C3ull;
  v = (v | (v << 2)) & 0x1249249249249249ull;
  return v;
}

// num_dims 3: [n,3] (x,y,z) -> 63-bit interleave, x in the lowest bit of every triple (single-batch "20-bit" path of the
// reference, csrc/morton_code.cu:49-58).  num_dims 4: [n,4] (b,x,y,z) -> (b << 48) | 48-bit interleave of the low 16
// bits per axis (batched "16-bit" path, morton_code.cu:27-45).  `origin` (device, may be NULL) is subtracted first -
// the reference normalises by the per-column minimum (serialization.py:211-212); `axis[3]` = which spatial column feeds
// the x / y / z slot (the MORTON_xyz permutations, serialization.py:44-51, 215-225).
__global__ __launch_bounds__(256) void morton_code_kernel(const int32_t* __restrict__ coords, int64_t n, int d,
                                                          const int32_t* __restrict__ origin, int a0, int a1, int a2,
                                                          int64_t* __restrict__ codes) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int32_t* c = coords + i * d;
  const int sp = d - 3;  // first spatial column
  int64_t v[3];
  const int ax[3] = {a0, a1, a2};
#pragma unroll
  for (int j = 0; j < 3; ++j) v[j] = (int64_t)c[sp + ax[j]] - (origin ? (int64_t)origin[sp + ax[j]] : 0);
  uint64_t m = spread3((uint64_t)v[0]) | (spread3((uint64_t)v[1]) << 1) | (spread3((uint64_t)v[2]) << 2);
  if (d == 4) {
    const int64_t b = (int64_t)c[0] - (origin ? (int64_t)origin[0] : 0);
    m = (m & 0x0000FFFFFFFFFFFFull) | ((uint64_t)b << 48);
  }
  codes[i] = (int64_t)m;
}

// nbr [m][kp] -> pair_table [K][m]
__global__ __launch_bounds__(kThreads) void kmap_transpose_kernel(const int32_t* __restrict__ nbr, int64_t m, int K,
                                                                  int kp, int tr, int32_t* __restrict__ pair_table) {
  extern __shared__ int s_tile[];  // [tr][kp+1], tr rows per workgroup (64, fewer for large kernel volumes)
  const int64_t row0 = (int64_t)blockIdx.x * tr;
  const int pitch = kp + 1;
  for (int e = threadIdx.x; e < tr * kp; e += kThreads) {
    const int r = e / kp, k = e % kp;
    s_tile[r * pitch + k] = (row0 + r < m) ? nbr[(row0 + r) * kp + k] : -1;
  }
  __syncthreads();
  for (int e = threadIdx.x; e < tr * K; e += kThreads) {
    const int k = e / tr, r = e % tr;
    if (row0 + r < m) pair_table[(int64_t)k * m + row0 + r] = s_tile[r * pitch + k];
  }
}

__global__ void fill_i32_kernel(int32_t* __restrict__ p, int64_t n, int32_t v) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}

__device__ __forceinline__ int find_bucket(const int32_t* __restrict__ offsets, int K, int64_t p) {
  int lo = 0, hi = K;  // largest k with offsets[k] <= p
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (offsets[mid] <= p) lo = mid; else hi = mid;
  }
  return lo;
}

// For every pair p of bucket k: tbl[row_of(p)][k] = other_of(p); mask bit k.  `by_in` selects which map
// indexes the table (reverse table: by input row; from_csr: by output row).
__global__ void kmap_pairs_to_table_kernel(const int32_t* __restrict__ in_maps, const int32_t* __restrict__ out_maps,
                                           const int32_t* __restrict__ offsets, int K, int kp, int mw, int64_t max_pairs,
                                           int by_in, int32_t* __restrict__ tbl, uint32_t* __restrict__ mask) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= max_pairs || p >= offsets[K]) return;
  const int k = find_bucket(offsets, K, p);
  const int i = in_maps[p], o = out_maps[p];
  const int64_t row = by_in ? i : o;
  tbl[row * kp + k] = by_in ? o : i;
  atomicOr(&mask[row * mw + (k >> 5)], 1u << (k & 31));
}

static inline int lanes_per_row(int kp) {
  int l = 8;
  while (l < kp && l < 64) l <<= 1;
  return l;
}

}  // namespace wcn

using namespace wcn;

extern "C" {

// 2: wcn_pack_weight[_f32] take the size of the destination buffer
// 3 (additions only): identity map in wcn_conv_gather_gemm (nbr = mask = NULL, one offset) + wcn_conv_identity_supported,
//    outputs wider than 128 channels on the channel-split kernels, wcn_bn_apply_residual / wcn_bn_backward_*_masked
// 4 (additions only): wcn_mask_tile_order (the binned builder's row order as an entry point), wcn_pack_weight_f32_pair,
//    mask = NULL with a binned 32-column table in wcn_conv_gather_gemm / wcn_conv_bn_backward (mask in column 31) +
//    wcn_conv_mask_in_table_supported, wcn_dense_rows[_supported] (narrow 1 x 1 x 1 layers), wcn_bn_train_backward_ld /
//    wcn_conv_bn_backward_ld (row pitch for the incoming gradient; the entries without _ld are unchanged)
// 5 (signatures changed): COMPACT neighbour rows - `compact` argument of wcn_kmap_build_binned / wcn_kmap_tally_sort /
//    wcn_kmap_scatter, wcn_kmap_compact_supported, wcn_kmap_densify, WCN_FLAG_ROW_OVERFLOW; mask = NULL in the gather GEMMs now means
//    a compact table (wcn_conv_compact_table_supported replaces wcn_conv_mask_in_table_supported; dense rows no longer carry a mask)
int wcn_abi_version(void) { return 5; }

const char* wcn_status_string(int status) {
  switch (status) {
    case WCN_SUCCESS: return "Success";
    case WCN_ERROR_PROBLEM_NOT_SUPPORTED: return "Problem size not supported";
    case WCN_ERROR_KERNEL_INITIALIZATION: return "Kernel initialization failed";
    case WCN_ERROR_KERNEL_EXECUTION: return "Kernel execution failed";
    case WCN_ERROR_UNSUPPORTED_CONFIG: return "Unsupported precision/configuration";
    case WCN_ERROR_INVALID_PARAMETERS: return "Invalid parameters";
    case WCN_ERROR_MIXED_INPUT_UNSUPPORTED: return "Mixed input precision unsupported";
    default: return "Unknown error";
  }
}

static inline bool is_pow2(int64_t v) { return v > 0 && (v & (v - 1)) == 0; }

int wcn_hash_prepare(void* slots, int64_t capacity, wcn_stream_t stream) {
  if (!slots || !is_pow2(capacity) || capacity > (1ll << 31)) return WCN_ERROR_INVALID_PARAMETERS;
  hipLaunchKernelGGL(hash_prepare_kernel, dim3((unsigned)ceil_div(capacity, 256)), dim3(256), 0, (hipStream_t)stream,
                     (uint4*)slots, capacity);
  return launch_status();
}

int wcn_hash_insert(void* slots, int64_t capacity, const int32_t* coords, int64_t n, int32_t* status,
                    wcn_stream_t stream) {
  if (!slots || !is_pow2(capacity) || capacity > (1ll << 31) || n < 0 || !status) return WCN_ERROR_INVALID_PARAMETERS;
  if (n == 0) return WCN_SUCCESS;
  if (!coords) return WCN_ERROR_INVALID_PARAMETERS;
  hipLaunchKernelGGL(hash_insert_kernel, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, (hipStream_t)stream,
                     (Slot*)slots, (uint32_t)(capacity - 1), (const int4*)coords, n, status);
  return launch_status();
}

int wcn_hash_search(const void* slots, int64_t capacity, const int32_t* queries, int64_t m, int32_t* results,
                    wcn_stream_t stream) {
  if (!slots || !is_pow2(capacity) || m < 0) return WCN_ERROR_INVALID_PARAMETERS;
  if (m == 0) return WCN_SUCCESS;
  if (!queries || !results) return WCN_ERROR_INVALID_PARAMETERS;
  hipLaunchKernelGGL(hash_search_kernel, dim3((unsigned)ceil_div(m, 256)), dim3(256), 0, (hipStream_t)stream,
                     (const Slot*)slots, (uint32_t)(capacity - 1), (const int4*)queries, m, results);
  return launch_status();
}

int32_t wcn_kmap_row_pitch(int32_t num_offsets) { return (num_offsets + 7) & ~7; }
int32_t wcn_kmap_mask_words(int32_t num_offsets) { return (num_offsets + 31) / 32; }
int wcn_kmap_probe(const void* slots, int64_t capacity, const int32_t* query, int64_t m, const int32_t ksize[3],
                   const int32_t stride[3], const int32_t dilation[3], int32_t* nbr, uint32_t* mask,
                   wcn_stream_t stream) {
  if (!slots || !is_pow2(capacity) || m < 0 || !ksize || !stride || !dilation) return WCN_ERROR_INVALID_PARAMETERS;
  for (int d = 0; d < 3; ++d)
    if (ksize[d] < 1 || stride[d] < 1 || dilation[d] < 1) return WCN_ERROR_INVALID_PARAMETERS;
  const int64_t K64 = (int64_t)ksize[0] * ksize[1] * ksize[2];
  if (K64 > 4096) return WCN_ERROR_PROBLEM_NOT_SUPPORTED;
  if (m == 0) return WCN_SUCCESS;
  if (!query || !nbr || !mask) return WCN_ERROR_INVALID_PARAMETERS;
  const int K = (int)K64, kp = wcn_kmap_row_pitch(K), mw = wcn_kmap_mask_words(K);
  ProbeGeom g;
  g.kx = ksize[0]; g.ky = ksize[1]; g.kz = ksize[2];
  g.cx = (g.kx & 1) ? g.kx / 2 : 0; g.cy = (g.ky & 1) ? g.ky / 2 : 0; g.cz = (g.kz & 1) ? g.kz / 2 : 0;
  g.sx = stride[0]; g.sy = stride[1]; g.sz = stride[2];
  g.dx = dilation[0]; g.dy = dilation[1]; g.dz = dilation[2];
  const dim3 grid((unsigned)ceil_div(m, kBlockRows)), block(kThreads);
  const size_t shm = 0;
  const uint32_t cmask = (uint32_t)(capacity - 1);
  hipStream_t s = (hipStream_t)stream;
  switch (lanes_per_row(kp)) {
    case 8:
      hipLaunchKernelGGL(kmap_probe_kernel<8>, grid, block, shm, s, (const Slot*)slots, cmask, (const int4*)query, m, g, K,
                         kp, mw, nbr, mask);
      break;
    case 16:
      hipLaunchKernelGGL(kmap_probe_kernel<16>, grid, block, shm, s, (const Slot*)slots, cmask, (const int4*)query, m, g,
                         K, kp, mw, nbr, mask);
      break;
    case 32:
      hipLaunchKernelGGL(kmap_probe_kernel<32>, grid, block, shm, s, (const Slot*)slots, cmask, (const int4*)query, m, g,
                         K, kp, mw, nbr, mask);
      break;
    default:
      hipLaunchKernelGGL(kmap_probe_kernel<64>, grid, block, shm, s, (const Slot*)slots, cmask, (const int4*)query, m, g,
                         K, kp, mw, nbr, mask);
      break;
  }
  return launch_status();
}

int wcn_batch_indexed_coords(const int32_t* coords, int64_t n, int32_t num_dims, const int32_t* offsets,
                             int32_t num_batches, int32_t* out, wcn_stream_t stream) {
  if (n < 0 || num_dims < 1 || num_dims > 8 || num_batches < 1) return WCN_ERROR_INVALID_PARAMETERS;
  if (n == 0) return WCN_SUCCESS;
  if (!coords || !out || (num_batches > 1 && !offsets)) return WCN_ERROR_INVALID_PARAMETERS;
  hipLaunchKernelGGL(batch_indexed_coords_kernel, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, (hipStream_t)stream, coords,
                     n, (int)num_dims, offsets, (int)num_batches, out);
  return launch_status();
}

int wcn_morton_code(const int32_t* coords, int64_t n, int32_t num_dims, const int32_t* origin, const int32_t axis[3],
                    int64_t* codes, wcn_stream_t stream) {
  if (n < 0 || (num_dims != 3 && num_dims != 4) || !axis) return WCN_ERROR_INVALID_PARAMETERS;
  int seen = 0;
  for (int j = 0; j < 3; ++j) {
    if (axis[j] < 0 || axis[j] > 2) return WCN_ERROR_INVALID_PARAMETERS;
    seen |= 1 << axis[j];
  }
  if (seen != 7) return WCN_ERROR_INVALID_PARAMETERS;
  if (n == 0) return WCN_SUCCESS;
  if (!coords || !codes) return WCN_ERROR_INVALID_PARAMETERS;
  hipLaunchKernelGGL(morton_code_kernel, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, (hipStream_t)stream, coords, n,
                     (int)num_dims, origin, (int)axis[0], (int)axis[1], (int)axis[2], codes);
  return launch_status();
}

int wcn_kmap_transpose(const int32_t* nbr, int64_t m, int32_t num_offsets, int32_t* pair_table, wcn_stream_t stream) {
  if (m < 0 || num_offsets < 1 || num_offsets > 4096) return WCN_ERROR_INVALID_PARAMETERS;
  if (m == 0) return WCN_SUCCESS;
  if (!nbr || !pair_table) return WCN_ERROR_INVALID_PARAMETERS;
  const int kp = wcn_kmap_row_pitch(num_offsets);
  int tr = 64;  // rows per workgroup: the LDS tile stays within 64 KB
  while (tr > 1 && (size_t)tr * (kp + 1) * sizeof(int) > 64 * 1024) tr >>= 1;
  hipLaunchKernelGGL(kmap_transpose_kernel, dim3((unsigned)ceil_div(m, tr)), dim3(kThreads),
                     (size_t)tr * (kp + 1) * sizeof(int), (hipStream_t)stream, nbr, m, (int)num_offsets, kp, tr, pair_table);
  return launch_status();
}

static int pairs_to_table(const int32_t* in_maps, const int32_t* out_maps, const int32_t* offsets, int32_t K,
                          int64_t max_pairs, int64_t rows, int by_in, int32_t* tbl, uint32_t* mask, hipStream_t s) {
  if (K < 1 || K > 4096 || rows < 0 || max_pairs < 0 || !offsets) return WCN_ERROR_INVALID_PARAMETERS;
  if (rows == 0) return WCN_SUCCESS;
  if (!tbl || !mask) return WCN_ERROR_INVALID_PARAMETERS;
  const int kp = wcn_kmap_row_pitch(K), mw = wcn_kmap_mask_words(K);
  hipLaunchKernelGGL(fill_i32_kernel, dim3((unsigned)ceil_div(rows * kp, 256)), dim3(256), 0, s, tbl, rows * kp, -1);
  hipLaunchKernelGGL(fill_i32_kernel, dim3((unsigned)ceil_div(rows * mw, 256)), dim3(256), 0, s, (int32_t*)mask,
                     rows * mw, 0);
  if (max_pairs > 0) {
    if (!in_maps || !out_maps) return WCN_ERROR_INVALID_PARAMETERS;
    hipLaunchKernelGGL(kmap_pairs_to_table_kernel, dim3((unsigned)ceil_div(max_pairs, 256)), dim3(256), 0, s, in_maps,
                       out_maps, offsets, (int)K, kp, mw, max_pairs, by_in, tbl, mask);
  }
  return launch_status();
}

int wcn_kmap_reverse(const int32_t* in_maps, const int32_t* out_maps, const int32_t* offsets, int32_t num_offsets,
                     int64_t max_pairs, int64_t n_in, int32_t* rev_nbr, uint32_t* rev_mask, wcn_stream_t stream) {
  return pairs_to_table(in_maps, out_maps, offsets, num_offsets, max_pairs, n_in, 1, rev_nbr, rev_mask,
                        (hipStream_t)stream);
}

int wcn_kmap_from_csr(const int32_t* in_maps, const int32_t* out_maps, const int32_t* offsets, int32_t num_offsets,
                      int64_t max_pairs, int64_t n_out, int32_t* nbr, uint32_t* mask, wcn_stream_t stream) {
  return pairs_to_table(in_maps, out_maps, offsets, num_offsets, max_pairs, n_out, 0, nbr, mask, (hipStream_t)stream);
}

}  // extern "C"
