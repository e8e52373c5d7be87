/*
 * kmap_oracle.c - CPU restatement of the reference's kernel-map algorithm.  TEST INFRASTRUCTURE ONLY:
 * imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg; never by the product.
 *
 * PARITY PINNING: the reference's own hash-table / kernel-map code is CUDA and cannot run in the
 * authoring container, so this file follows the cited reference sources line by line in meaning and is
 * pinned by (a) the reference's Python `kernel_offsets_from_size` tables captured in
 * tests/golden/ (offset enumeration and centring), (b) an independent brute-force dictionary builder
 * (oracle/brute.py, the method of the reference's tests/coords/test_kernel_map_invariants.py:205-230) and
 * (c) the invariants the reference's tests assert (tests/test_oracle_kmap.py).
 *
 * Each function cites the reference file:line it restates (paths relative to the reference tree).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define BATCH_MASK 0x1FFu
#define COORD_MASK 0x3FFFFu
#define COORD_MAX 131071
#define COORD_MIN (-131072)
#define BATCH_MAX 511

/* warpconvnet/csrc/include/cuhash/hash_functions.cuh:40-44  pack_key_4d */
static uint64_t pack_key(int b, int x, int y, int z) {
  return (1ull << 63) | ((uint64_t)((uint32_t)b & BATCH_MASK) << 54) | ((uint64_t)((uint32_t)x & COORD_MASK) << 36) |
         ((uint64_t)((uint32_t)y & COORD_MASK) << 18) | (uint64_t)((uint32_t)z & COORD_MASK);
}

/* hash_functions.cuh:75-84  Splitmix64Hash::hash */
static uint32_t splitmix(uint64_t key, uint32_t capacity_mask) {
  key ^= key >> 30;
  key *= 0xBF58476D1CE4E5B9ull;
  key ^= key >> 27;
  key *= 0x94D049BB133111EBull;
  key ^= key >> 31;
  return (uint32_t)key & capacity_mask;
}

/* warpconvnet/geometry/coords/search/_packed_base.py:19-30  _next_power_of_2 */
int64_t oracle_next_pow2(int64_t n) {
  int64_t p = 1;
  while (p < n) p <<= 1;
  return p;
}

/* packed_hashmap.py:66-82 range validation -> returns 2 if any coordinate is out of range, else 0 */
int oracle_check_range(const int32_t* coords, int64_t n) {
  for (int64_t i = 0; i < n; ++i) {
    const int32_t* c = coords + 4 * i;
    if (c[0] < 0 || c[0] > BATCH_MAX) return 2;
    for (int d = 1; d < 4; ++d)
      if (c[d] < COORD_MIN || c[d] > COORD_MAX) return 2;
  }
  return 0;
}

/* cuhash_hash_table.cu:19-25 (prepare: keys = 0, values = -1) + hash_table.cuh:36-63 (linear-probe insert,
 * first inserter of a key keeps the slot; serial row order => the smallest row index wins).
 * Returns 0, or 1 when the table is full (status flag of hash_table.cuh:60-62). */
int oracle_hash_build(const int32_t* coords, int64_t n, uint64_t* keys, int32_t* values, int64_t capacity) {
  const uint32_t cmask = (uint32_t)(capacity - 1);
  for (int64_t i = 0; i < capacity; ++i) { keys[i] = 0ull; values[i] = -1; }
  for (int64_t i = 0; i < n; ++i) {
    const int32_t* c = coords + 4 * i;
    const uint64_t key = pack_key(c[0], c[1], c[2], c[3]);
    uint32_t slot = splitmix(key, cmask);
    int64_t attempts = 0;
    int placed = 0;
    while (attempts <= (int64_t)cmask) {
      if (keys[slot] == 0ull) { keys[slot] = key; values[slot] = (int32_t)i; placed = 1; break; }
      if (keys[slot] == key) { placed = 1; break; } /* dedup */
      slot = (slot + 1) & cmask;
      ++attempts;
    }
    if (!placed) return 1;
  }
  return 0;
}

/* hash_table.cuh:94-108  packed_search */
static int32_t hash_search_key(const uint64_t* keys, const int32_t* values, uint32_t cmask, uint64_t q) {
  uint32_t slot = splitmix(q, cmask);
  int64_t attempts = 0;
  while (attempts <= (int64_t)cmask) {
    const uint64_t k = keys[slot];
    if (k == 0ull) return -1;
    if (k == q) return values[slot];
    slot = (slot + 1) & cmask;
    ++attempts;
  }
  return -1;
}

/* cuhash_hash_table.cu:222-262  packed_search kernel */
void oracle_hash_search(const uint64_t* keys, const int32_t* values, int64_t capacity, const int32_t* queries, int64_t m,
                        int32_t* results) {
  const uint32_t cmask = (uint32_t)(capacity - 1);
  for (int64_t i = 0; i < m; ++i) {
    const int32_t* q = queries + 4 * i;
    results[i] = hash_search_key(keys, values, cmask, pack_key(q[0], q[1], q[2], q[3]));
  }
}

/* cuhash_kernel_map.cu:93-134 packed_kernel_map_size_kernel + kernel_map.cuh:34-54 (k -> (ii,jj,kk), centre =
 * size/2 for odd sizes else 0) + torch_discrete.py:24-56 (dilation) + :352-361 (query = out * stride).
 * found is the reference layout [K, M]: found[k*M + j]. */
void oracle_kernel_map_found(const uint64_t* keys, const int32_t* values, int64_t capacity, const int32_t* out_coords,
                             int64_t m, const int32_t ksize[3], const int32_t stride[3], const int32_t dilation[3],
                             int32_t* found) {
  const uint32_t cmask = (uint32_t)(capacity - 1);
  const int kx = ksize[0], ky = ksize[1], kz = ksize[2];
  const int cx = (kx % 2) ? kx / 2 : 0, cy = (ky % 2) ? ky / 2 : 0, cz = (kz % 2) ? kz / 2 : 0;
  const int K = kx * ky * kz;
  for (int k = 0; k < K; ++k) {
    const int kk = k % kz, jj = (k / kz) % ky, ii = k / (kz * ky);
    const int ox = (ii - cx) * dilation[0], oy = (jj - cy) * dilation[1], oz = (kk - cz) * dilation[2];
    for (int64_t j = 0; j < m; ++j) {
      const int32_t* q = out_coords + 4 * j;
      const uint64_t key = pack_key(q[0], q[1] * stride[0] + ox, q[2] * stride[1] + oy, q[3] * stride[2] + oz);
      found[(int64_t)k * m + j] = hash_search_key(keys, values, cmask, key);
    }
  }
}

/* cuhash_kernel_map.cu:508-544 postprocess_count + torch.cumsum (torch_discrete.py:268-272):
 * offsets[K+1]; cuhash_kernel_map.cu:546-599 postprocess_scatter: in_maps/out_maps per bucket.  The
 * reference's order inside a bucket is racy; the canonical order used for parity is ascending output row.
 * Pass in_maps = NULL to only count.  Returns the total number of pairs. */
int64_t oracle_compact(const int32_t* found, int64_t m, int K, int32_t* offsets, int32_t* in_maps, int32_t* out_maps) {
  int64_t total = 0;
  offsets[0] = 0;
  for (int k = 0; k < K; ++k) {
    for (int64_t j = 0; j < m; ++j) {
      const int32_t v = found[(int64_t)k * m + j];
      if (v >= 0) {
        if (in_maps) { in_maps[total] = v; out_maps[total] = (int32_t)j; }
        ++total;
      }
    }
    offsets[k + 1] = (int32_t)total;
  }
  return total;
}

/* mask_data_kernels.cu:23-44 build_pair_mask: bit (k%32) of mask[i*mw + k/32] <=> found[k][i] >= 0 */
void oracle_pair_mask(const int32_t* found, int64_t m, int K, int mw, uint32_t* mask) {
  memset(mask, 0, (size_t)m * mw * sizeof(uint32_t));
  for (int k = 0; k < K; ++k)
    for (int64_t i = 0; i < m; ++i)
      if (found[(int64_t)k * m + i] >= 0) mask[i * mw + (k >> 5)] |= 1u << (k & 31);
}

/* mask_data_kernels.cu:101-124 build_reverse_mask_data: rev[k][in] = out (reference layout [K, N_in]) + mask */
void oracle_reverse(const int32_t* found, int64_t m, int64_t n_in, int K, int mw, int32_t* rev, uint32_t* rev_mask) {
  for (int64_t i = 0; i < (int64_t)K * n_in; ++i) rev[i] = -1;
  memset(rev_mask, 0, (size_t)n_in * mw * sizeof(uint32_t));
  for (int k = 0; k < K; ++k)
    for (int64_t j = 0; j < m; ++j) {
      const int32_t v = found[(int64_t)k * m + j];
      if (v >= 0) {
        rev[(int64_t)k * n_in + v] = (int32_t)j;
        rev_mask[(int64_t)v * mw + (k >> 5)] |= 1u << (k & 31);
      }
    }
}

/* coords/ops/stride.py:38-56: floor-divide and de-duplicate.  Output rows = first occurrences in input order
 * (the reference's order is implementation-defined: torch.unique of winner indices + unstable argsort by batch).
 * Returns the number of unique rows; out_coords must hold n rows; first_index[n] receives the source row. */
int64_t oracle_stride_coords(const int32_t* coords, int64_t n, const int32_t stride[3], int32_t* out_coords,
                             int32_t* first_index) {
  int64_t cap = oracle_next_pow2(2 * n > 16 ? 2 * n : 16);
  uint64_t* keys = (uint64_t*)calloc((size_t)cap, sizeof(uint64_t));
  const uint32_t cmask = (uint32_t)(cap - 1);
  int64_t cnt = 0;
  for (int64_t i = 0; i < n; ++i) {
    int32_t c[4];
    c[0] = coords[4 * i];
    for (int d = 0; d < 3; ++d) {
      const int32_t v = coords[4 * i + 1 + d], s = stride[d];
      int32_t q = v / s;
      if ((v % s != 0) && ((v < 0) != (s < 0))) --q; /* floor division */
      c[1 + d] = q;
    }
    const uint64_t key = pack_key(c[0], c[1], c[2], c[3]);
    uint32_t slot = splitmix(key, cmask);
    int dup = 0;
    while (keys[slot] != 0ull) {
      if (keys[slot] == key) { dup = 1; break; }
      slot = (slot + 1) & cmask;
    }
    if (!dup) {
      keys[slot] = key;
      memcpy(out_coords + 4 * cnt, c, sizeof(c));
      first_index[cnt] = (int32_t)i;
      ++cnt;
    }
  }
  free(keys);
  return cnt;
}
