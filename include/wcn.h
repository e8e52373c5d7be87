/*
 * wcn.h - C-ABI of the MI355X (gfx950) sparse-3D-convolution engine.
 *
 * This is the drop-in boundary for the SparseConv3d hot path: every entry point replaces one
 * binding of the reference's pybind11 module `warpconvnet._C` (the file:line cited at each
 * declaration is relative to the reference tree).  Conventions:
 *
 *   - plain C: raw device pointers, sizes, and a `hipStream_t` passed as `void*`; no torch types;
 *   - the CALLER allocates every buffer (outputs, workspaces); the library never allocates or frees
 *     device memory; its only process-wide state is a per-device once-flag (kernel attributes), so the
 *     entry points are re-entrant.  A single call is a fixed sequence of launches on the stream and may be
 *     stream-captured; a kernel-map BUILD as the host drives it is not a capturable unit - the host reads
 *     the status word / pair count (pinned mirror written by wcn_kmap_tally_sort) and may rebuild with a
 *     larger block table or the strict insert, exactly as the reference syncs at torch_discrete.py:272;
 *   - every launch goes to the stream argument and is asynchronous; nothing here synchronises;
 *   - return value: 0 on success, negative `wcn_status` otherwise (same numbering as the reference's
 *     `GemmStatus`, warpconvnet/csrc/include/gemm_error_codes.h:7-15);
 *   - data-dependent failures (hash table full, coordinate out of packed range, pair-buffer
 *     overflow) are reported through a device status word the caller reads back when it chooses
 *     (reference: warpconvnet/geometry/coords/search/_packed_base.py:113-120).
 *
 * All index tensors are int32.  Feature tensors are row-major [N, C].
 */
#ifndef WCN_H_
#define WCN_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* wcn_stream_t; /* hipStream_t */

/* Status codes: numbering of reference GemmStatus (include/gemm_error_codes.h:7-15). */
enum wcn_status {
  WCN_SUCCESS = 0,
  WCN_ERROR_PROBLEM_NOT_SUPPORTED = -1,
  WCN_ERROR_KERNEL_INITIALIZATION = -2,
  WCN_ERROR_KERNEL_EXECUTION = -3,
  WCN_ERROR_UNSUPPORTED_CONFIG = -4,
  WCN_ERROR_INVALID_PARAMETERS = -5,
  WCN_ERROR_MIXED_INPUT_UNSUPPORTED = -6
};

enum wcn_dtype { WCN_F32 = 0, WCN_F16 = 1, WCN_BF16 = 2 };

/* Bits of the device status word written by the kernel-map kernels. */
enum wcn_kmap_flag {
  WCN_FLAG_TABLE_FULL = 1,      /* hash insert ran out of slots   (reference hash_table.cuh:60-62)   */
  WCN_FLAG_COORD_RANGE = 2,     /* batch not in [0,511] or coord not in [-131072,131071]
                                   (reference packed_hashmap.py:66-82 raises ValueError)            */
  WCN_FLAG_PAIR_OVERFLOW = 4,   /* in_maps/out_maps capacity smaller than the number of pairs       */
  WCN_FLAG_DUPLICATE_COORD = 8, /* informational: the inserted coordinates are not all distinct (the smallest row wins,
                                   so the centre neighbour of a later duplicate is not the row itself)          */
  WCN_FLAG_NEED_STRICT = 16,    /* binned builder, strict = 0: a duplicate coordinate kept a row that is not the smallest
                                   one - the tables are NOT valid, rebuild with strict = 1                         */
  WCN_FLAG_ROW_OVERFLOW = 32    /* binned builder, compact = 1: a row has more than 15 neighbours and does not fit a compact
                                   row - the tables are NOT valid, rebuild with compact = 0                        */
};

/* ---- misc ------------------------------------------------------------------------------------ */
int wcn_abi_version(void);
/* reference: _C.gemm.gemm_status_to_string (bindings/gemm_bindings.cpp) */
const char* wcn_status_string(int status);

/* ---- packed 64-bit coordinate hash table -------------------------------------------------------
 * Key = 1<<63 | batch(9b)<<54 | x(18b)<<36 | y(18b)<<18 | z(18b)   (reference hash_functions.cuh:40-44)
 * Slot = 16 bytes {uint64 key, int32 value, int32 pad}: one 16-B load per probe returns key AND
 * value (the reference keeps two arrays -> two random loads per hit).  Empty key = 0, value = row
 * index of the inserted coordinate; duplicates keep the SMALLEST row index (deterministic; the
 * reference keeps whichever CAS wins, hash_table.cuh:36-63).  Splitmix64 & (capacity-1), linear
 * probing (hash_functions.cuh:75-84, hash_table.cuh:94-108).  capacity must be a power of two.
 */
/* reference: _C.cuhash.packed_prepare (bindings/cuhash_bindings.cpp:244-262, cuhash_hash_table.cu:19-25) */
int wcn_hash_prepare(void* slots, int64_t capacity, wcn_stream_t stream);
/* reference: _C.cuhash.packed_insert (cuhash_hash_table.cu:179-220); coords int32 [n,4]; status int32[1] (OR of wcn_kmap_flag) */
int wcn_hash_insert(void* slots, int64_t capacity, const int32_t* coords, int64_t n, int32_t* status,
                    wcn_stream_t stream);
/* reference: _C.cuhash.packed_search (cuhash_hash_table.cu:222-262); results int32 [m], -1 on miss */
int wcn_hash_search(const void* slots, int64_t capacity, const int32_t* queries, int64_t m,
                    int32_t* results, wcn_stream_t stream);

/* ---- kernel map ----------------------------------------------------------------------------------
 * Neighbour table layout is ROW-major: nbr[m*kp + k] = input row matched by output row m at kernel
 * offset k, or -1 (kp = wcn_kmap_row_pitch(K), rows are 32-B aligned so a tile of rows is one
 * contiguous slab).  The reference's `found_in_coord_index` is the transpose [K, M]
 * (cuhash_kernel_map.cu:93-134); `wcn_kmap_transpose` converts for consumers that want that layout.
 * Kernel offset enumeration: k = (i*ky + j)*kz + l -> offset ((i-cx)*dx, (j-cy)*dy, (l-cz)*dz) with
 * c = (size-1)/2 for odd sizes and 0 for even sizes (torch_discrete.py:24-56, kernel_map.cuh:34-54).
 * Query coordinate = out_coord * stride (torch_discrete.py:352-361).
 */
int32_t wcn_kmap_row_pitch(int32_t num_offsets);
int32_t wcn_kmap_mask_words(int32_t num_offsets);
/* number of 256-row bucket tiles for m query rows, rounded up to a multiple of 4 (rows of the counts array stay 16-B
 * aligned); the counts buffer holds wcn_kmap_counts_bytes(m, K) bytes: K rows of tile counts, K totals, a ticket word */
int64_t wcn_kmap_num_blocks(int64_t m);
size_t wcn_kmap_counts_bytes(int64_t m, int32_t num_offsets);

/* reference: _C.cuhash.packed_kernel_map_size + packed_kernel_map_offset (cuhash_kernel_map.cu:68-134)
 * fused with build_pair_mask (mask_data_kernels.cu:23-44).
 *   query   int32 [m,4]      mask   uint32 [m, mask_words]  (bit k <=> nbr[m][k] >= 0)
 *   nbr     int32 [m,kp]
 */
int wcn_kmap_probe(const void* slots, int64_t capacity, const int32_t* query, int64_t m,
                   const int32_t ksize[3], const int32_t stride[3], const int32_t dilation[3],
                   int32_t* nbr, uint32_t* mask, wcn_stream_t stream);
/* [n, num_dims] int32 coordinates -> [n, num_dims + 1] with the batch index of the row in column 0; `offsets`
 * [num_batches + 1] int32 on the device (may be NULL for one batch).  One launch; replaces the batch-index kernel +
 * torch.cat of warpconvnet/geometry/coords/ops/batch_index.py:90-148. */
int wcn_batch_indexed_coords(const int32_t* coords, int64_t n, int32_t num_dims, const int32_t* offsets,
                             int32_t num_batches, int32_t* out, wcn_stream_t stream);

/* Z-order (Morton) codes of integer coordinates: int64 codes[n].  num_dims 3: rows (x,y,z), 21 bits per axis
 * interleaved with x in the lowest bit of every triple; num_dims 4: rows (b,x,y,z), (b << 48) | interleave of the low 16
 * bits per axis.  `origin` [num_dims] int32 on the DEVICE (may be NULL) is subtracted first (the reference normalises by
 * the per-column minimum); `axis[3]` (host) names the spatial column that feeds the x / y / z slot (MORTON_XYZ = {0,1,2},
 * MORTON_ZYX = {2,1,0}, ...).  Replaces _C.coords.morton_code_20bit / morton_code_16bit
 * (warpconvnet/csrc/morton_code.cu:27-103, call site geometry/coords/ops/serialization.py:196-245). */
int wcn_morton_code(const int32_t* coords, int64_t n, int32_t num_dims, const int32_t* origin, const int32_t axis[3],
                    int64_t* codes, wcn_stream_t stream);

/* LDS-binned neighbour search for SUBMANIFOLD maps (query coords == input coords, stride 1): replaces
 * wcn_hash_insert + wcn_kmap_probe.  The hash table is BLOCK-level: voxels are binned into 8^3 blocks, every occupied
 * block owns a dense 512-cell sub-grid of row ids (one plain 4-B store per voxel), and one wavefront per block stages
 * its sub-grid and the halo of the 26 neighbouring sub-grids in an LDS grid and answers all K probes from LDS.  Same
 * outputs as the hash path (bit-exact), incl. the range flags in *status (the status word is CLEARED by this call).
 *   max_blocks  capacity of the block table (occupied 8^3 blocks); more blocks -> WCN_FLAG_TABLE_FULL, tables invalid,
 *               the caller retries with max_blocks = n (always enough) - workspace = wcn_kmap_binned_workspace(n, max_blocks)
 *   strict      0: cells are written with plain stores; rows of duplicate coordinates are repaired by
 *               wcn_kmap_tally_sort, which raises WCN_FLAG_NEED_STRICT when a cell did not keep the smallest row
 *               (the caller then rebuilds with strict = 1: atomicMin, ~2.5x the store cost)
 *   mask        rows that no block has enumerated yet (duplicates) carry 0x80000000 in their LAST mask word until
 *               wcn_kmap_tally_sort repairs them, which is why K % 32 == 0 is not supported here
 * Returns WCN_ERROR_PROBLEM_NOT_SUPPORTED when the kernel halo (max |offset| per axis, dilation included) exceeds 8 cells or K % 32 == 0
 * (wcn_kmap_binned_supported == 0): the caller then uses the hash path.
 *   compact     0: nbr is the dense [n, kp] table above.  1 (wcn_kmap_compact_supported: one mask word and 17 <= K <= 31, e.g. the
 *               3 x 3 x 3 kernel): nbr is [n, 16] COMPACT rows - word 0 = the row's mask, words 1 .. popcount(mask) = the neighbour
 *               rows of its SET offsets in ascending k, the rest unspecified - 64 B per row instead of 128 for 4 - 9 neighbours
 *               (round 6: half the bytes for the builder's stores, the pair scatter and both gather GEMMs).  A row with more
 *               than 15 neighbours raises WCN_FLAG_ROW_OVERFLOW (tables invalid, rebuild with compact = 0).  Consumers of compact
 *               tables: wcn_kmap_tally_sort / wcn_kmap_scatter (compact = 1), the gather GEMMs where
 *               wcn_conv_compact_table_supported (`mask` = NULL), wcn_kmap_densify for everything else.
 * reference being replaced: cuhash_hash_table.cu:179-220 + cuhash_kernel_map.cu:93-134. */
size_t wcn_kmap_binned_workspace(int64_t n, int64_t max_blocks);
int wcn_kmap_binned_supported(const int32_t ksize[3], const int32_t dilation[3]);
int wcn_kmap_compact_supported(int32_t num_offsets);
int wcn_kmap_build_binned(const int32_t* coords, int64_t n, const int32_t ksize[3], const int32_t dilation[3],
                          int64_t max_blocks, int32_t strict, int32_t compact, void* workspace, size_t workspace_bytes,
                          int32_t* nbr, uint32_t* mask, int32_t* status, wcn_stream_t stream);
/* compact rows [m, 16] -> the dense table nbr [m, kp] (-1 = absent): for consumers without a compact path. */
int wcn_kmap_densify(const int32_t* nbr_compact, int64_t m, int32_t num_offsets, int32_t* nbr, wcn_stream_t stream);
/* Everything between the neighbour table and the ONE host read of a build:
 *   tally  per-(offset, tile) pair counts from the masks, the first digit histogram of the mask sort, and - when
 *          `binned_workspace` (the workspace of the wcn_kmap_build_binned call that produced nbr / mask, with its n and
 *          max_blocks) is given - repair of the rows of duplicate coordinates
 *   scan   offsets int32 [K+1] on the device and, if `host_mirror` (K + 3 int32 of pinned, device-accessible HOST
 *          memory) is given, offsets ++ [*status] ++ [ready] there in the same launch: `ready` (cleared by the caller)
 *          becomes 1 last, behind a system-scope fence, so the host may spin on it instead of waiting for an event
 *   sort   perm = rows by descending mask (wcn_mask_argsort), the first digit already counted
 *   pairs  (in_maps / out_maps given, pair_capacity > 0: round 6) the pair lists of wcn_kmap_scatter queued behind the sort by
 *          the same call, at a capacity the caller guesses before the pair count is known.  A capacity below the pair count:
 *          nothing past it is written, the caller compares it with offsets[K] (in the mirror by then) and runs
 *          wcn_kmap_scatter at the exact length.
 *   compact  nbr holds COMPACT rows (wcn_kmap_build_binned)
 * counts: wcn_kmap_counts_bytes(m, K) bytes, consumed by wcn_kmap_scatter.  sort_workspace:
 * wcn_kmap_tally_sort_workspace(m) bytes.  reference: postprocess_count + torch.cumsum + mask_argsort
 * (cuhash_kernel_map.cu:508-544, torch_discrete.py:268-272, mask_data_kernels.cu:187-220). */
size_t wcn_kmap_tally_sort_workspace(int64_t m);
int wcn_kmap_tally_sort(uint32_t* mask, int32_t* nbr, int64_t m, int32_t num_offsets, int32_t* counts, int32_t* offsets,
                        int32_t* status, int32_t* host_mirror, int32_t* perm, void* sort_workspace,
                        size_t sort_workspace_bytes, const int32_t* coords, void* binned_workspace, int64_t binned_n,
                        int64_t max_blocks, int32_t compact, int32_t* in_maps, int32_t* out_maps, int64_t pair_capacity,
                        wcn_stream_t stream);
/* counts[k][b] = pairs of offset k in the 256-row tile b (k-major, from the masks); counts = wcn_kmap_counts_bytes bytes.
 * reference: _C.cuhash.postprocess_count (cuhash_kernel_map.cu:508-544). */
int wcn_kmap_count(const uint32_t* mask, int64_t m, int32_t num_offsets, int32_t* counts, wcn_stream_t stream);
/* exclusive scan of counts over tiles (in place, one workgroup per offset) and over offsets ->
 * offsets int32 [K+1].  `counts` as left by wcn_kmap_count (the tail receives the bucket totals).
 * reference: host torch.cumsum in torch_discrete.py:268-272. */
int wcn_kmap_scan(int32_t* counts, int64_t num_blocks, int32_t num_offsets, int32_t* offsets,
                  wcn_stream_t stream);
/* the same, and in the same launch offsets[0..K] ++ [*status] ++ [ready = 1] are also written to `host_mirror` - K + 3 int32 of pinned
 * (device-accessible) HOST memory: the one host read of a build (torch_discrete.py:268-272 does `.item()` per value)
 * becomes an event wait behind this kernel, no copy command. */
int wcn_kmap_scan_to_host(int32_t* counts, int64_t num_blocks, int32_t num_offsets, int32_t* offsets, const int32_t* status,
                          int32_t* host_mirror, wcn_stream_t stream);
/* deterministic compaction (pairs of one offset ordered by output row).
 * reference: _C.cuhash.postprocess_scatter (cuhash_kernel_map.cu:546-599, order there is racy).
 * pair_capacity = length of in_maps/out_maps; sets WCN_FLAG_PAIR_OVERFLOW in *status if too small. */
int wcn_kmap_scatter(const int32_t* nbr, const uint32_t* mask, int64_t m, int32_t num_offsets,
                     const int32_t* counts, const int32_t* offsets, int32_t* in_maps, int32_t* out_maps, int64_t pair_capacity,
                     int32_t* status, int32_t compact, wcn_stream_t stream);
/* ---- strided layers from the cell table of the FINE set (no global hash table) ------------------------------------------
 * `cells_workspace` = the workspace of a wcn_kmap_build_binned call on the fine coordinates (with its n and max_blocks)
 * whose status word came back without TABLE_FULL / NEED_STRICT.
 * Down-sampling (reference coords/ops/stride.py:18-56: floor-divide, hash de-duplicate, unique): strides 1 / 2 / 4 / 8 per
 * axis, so that a coarse cell lies inside one 8^3 block (wcn_cells_stride_supported).
 *   _count  flags = one bit per fine row: "first (smallest) row of its coarse cell" (uint64 per 64 rows); counts
 *           [wcn_cells_stride_tiles(n) + 1] int32 = survivors in front of every 256-row tile, the last entry their total;
 *           out_offsets[b] = survivors with a batch index < b, b = 0 .. num_batches (the batch offsets of the output; the
 *           rows must be sorted by batch index, as everywhere)
 *   _emit   out_coords [M, 4] = (b, x >> log2 sx, ...) of the survivors in input order; first_rows [M] (may be NULL) their
 *           fine rows; nbr [M, kp] / mask [M] (both or neither): the kernel map of a convolution with kernel_size == stride,
 *           dilation 1 - nbr[m][(i*sy + j)*sz + l] = fine row in cell (x0 + i, y0 + j, z0 + l) of the coarse cell, or -1
 * wcn_kmap_probe_cells: wcn_kmap_probe answered from the cell table - same table / mask layout, same 18-bit wrap, the
 * smallest row of a duplicated coordinate - for any kernel size, stride and dilation; replaces wcn_hash_insert +
 * wcn_kmap_probe when the input set already has a cell table.  reference: cuhash_kernel_map.cu:93-134;
 * coarse-to-fine search geometry/coords/search/hierarchical_search.py:25-66. */
/* the cell table alone (prepare + inserts + finish of wcn_kmap_build_binned, strict insert: the smallest row of a duplicated
 * coordinate wins by construction) for a coordinate set that has no submanifold build yet - e.g. a network whose first
 * spatial layer is strided.  `scratch`: n uint32.  status: WCN_FLAG_TABLE_FULL (retry with a larger max_blocks) /
 * WCN_FLAG_COORD_RANGE.  workspace: wcn_kmap_binned_workspace(n, max_blocks). */
int wcn_kmap_cells_build(const int32_t* coords, int64_t n, int64_t max_blocks, void* workspace, size_t workspace_bytes,
                         uint32_t* scratch, int32_t* status, wcn_stream_t stream);
int wcn_cells_stride_supported(const int32_t stride[3]);
int64_t wcn_cells_stride_tiles(int64_t n);
int wcn_cells_stride_count(const void* cells_workspace, int64_t n, int64_t max_blocks, const int32_t* coords,
                           const int32_t stride[3], uint64_t* flags, int32_t* counts, int32_t num_batches,
                           int32_t* out_offsets, wcn_stream_t stream);
int wcn_cells_stride_emit(const void* cells_workspace, int64_t n, int64_t max_blocks, const int32_t* coords,
                          const int32_t stride[3], const uint64_t* flags, const int32_t* counts, int32_t* out_coords,
                          int32_t* first_rows, int32_t* nbr, uint32_t* mask, wcn_stream_t stream);
int wcn_kmap_probe_cells(const void* cells_workspace, int64_t n_in, int64_t max_blocks, const int32_t* query, int64_t m,
                         const int32_t ksize[3], const int32_t stride[3], const int32_t dilation[3], int32_t* nbr,
                         uint32_t* mask, wcn_stream_t stream);
/* nbr [m,kp] -> pair_table [K,m]   (the reference layout, cuhash_kernel_map.cu:133) */
int wcn_kmap_transpose(const int32_t* nbr, int64_t m, int32_t num_offsets, int32_t* pair_table,
                       wcn_stream_t stream);
/* reverse table for dgrad of strided / transposed maps: rev_nbr[in][k] = out, rev_mask bit k.
 * The call itself pre-fills rev_nbr with -1 and rev_mask with 0.  `max_pairs` bounds the launch
 * (length of in_maps/out_maps); the true pair count is read from offsets[K] on the device.
 * reference: _C.gemm.build_reverse_mask_data_cuda (mask_data_kernels.cu:101-124). */
int wcn_kmap_reverse(const int32_t* in_maps, const int32_t* out_maps, const int32_t* offsets,
                     int32_t num_offsets, int64_t max_pairs, int64_t n_in, int32_t* rev_nbr,
                     uint32_t* rev_mask, wcn_stream_t stream);
/* reference: _C.gemm.csr_to_pair_table_cuda + build_pair_mask_cuda (mask_data_kernels.cu:23-82):
 * rebuild nbr/mask from a CSR map (used for maps that were swapped for transposed conv). */
int wcn_kmap_from_csr(const int32_t* in_maps, const int32_t* out_maps, const int32_t* offsets,
                      int32_t num_offsets, int64_t max_pairs, int64_t n_out, int32_t* nbr, uint32_t* mask,
                      wcn_stream_t stream);
/* permutation of rows sorted by descending mask word 0 (rows with equal neighbourhood pattern become
 * adjacent so a wavefront can skip absent offsets).  reference: _C.gemm.mask_argsort_cuda
 * (mask_data_kernels.cu:187-220, CUB radix sort).  Hand-written stable LSD radix sort (ties keep ascending
 * row order); `num_bits` = number of significant low bits of word 0 (min(K, 32)) bounds the passes.
 * workspace: wcn_mask_argsort_workspace(n) bytes. */
size_t wcn_mask_argsort_workspace(int64_t n);
int wcn_mask_argsort(const uint32_t* mask, int32_t mask_words, int32_t num_bits, int64_t n, int32_t* perm,
                     void* workspace, size_t workspace_bytes, wcn_stream_t stream);
/* The row order the gather GEMMs cut into tiles (and the `perm` wcn_kmap_tally_sort writes): a tile runs one step per offset ANY of
 * its rows has, so rows are grouped to keep the UNION of a tile's masks small.  Odd kernel volumes K = 2c + 1 <= 31: stable sort
 * by a key of the same width as the mask - mirror-image offsets (k, K-1-k) paired, "which pairs" in the high half, "which member"
 * in the low half, ranked in reflected-Gray order (csrc/mask_sort.h) - 8.8 -> 7.2 steps per 128-row tile on the uniform 1 M scene,
 * unchanged on surface-like scenes.  Any other volume: wcn_mask_argsort.  The GEMM results do not depend on the order.
 * Role of the reference's mask_argsort inside its mask-GEMM path (mask_gemm.py:127-254).  workspace: wcn_mask_argsort_workspace. */
int wcn_mask_tile_order(const uint32_t* mask, int32_t mask_words, int32_t num_offsets, int64_t n, int32_t* perm,
                        void* workspace, size_t workspace_bytes, wcn_stream_t stream);

/* ---- sparse convolution GEMMs ------------------------------------------------------------------
 * Forward  (AB gather-scatter):  y[m]  = sum_k x[nbr[m][k]] . w[k]            w: [K, Cin, Cout]
 * Dgrad    (ABt gather-scatter): dx[n] = sum_k dy[rnbr[n][k]] . w[k]^T
 * Wgrad    (AtB gather-gather):  dw[k] = sum_p x[in_maps[p]]^T . dy[out_maps[p]],  p in bucket k
 * reference semantics: nn/functional/sparse_conv/detail/explicit.py:22-101; fused production kernels
 * _C.mask_gemm.fwd/.dgrad/.wgrad (bindings/mask_gemm_bindings.cu:2071-2123).
 *
 * `algo`: 1 = hip_ref (any channel count / dtype), 2 = hip_mfma (bf16/f16; forward/dgrad: Cin%16==0,
 * Cout in {32,64,96,128,192,256}, K<=32; wgrad: Cin%32==0, Cout%32==0; WCN_ERROR_UNSUPPORTED_CONFIG
 * otherwise).  0 (auto) is resolved by the caller with wcn_mfma_*_supported because the two algorithms
 * take different weight images.  Accumulation is always fp32.
 */
enum wcn_algo { WCN_ALGO_AUTO = 0, WCN_ALGO_REF = 1, WCN_ALGO_MFMA = 2 };

/* 1 if the MFMA kernels cover the shape (so the caller can resolve "auto" without a trial launch). */
int wcn_mfma_gather_supported(int32_t cin, int32_t cout, int32_t num_offsets, int32_t dtype);
int wcn_mfma_wgrad_supported(int32_t cin, int32_t cout, int32_t dtype);
/* 1 x 1 x 1 convolutions (reference shortcut helper.py:206-213: feats @ weight[0]): wcn_conv_gather_gemm with num_offsets = 1
 * and nbr = mask = perm = NULL treats every row as its own only neighbour - the dense [N, cin] x [cin, cout] product streamed
 * through the gather kernel at its HBM rate (the vendor GEMM runs these skinny shapes at 0.15-0.33 of it,
 * profiles/r04_unet_1M_roofline.md).  Shapes: cin % 32 == 0, cin >= 64, cout in {64, 96, 128}, f16 / bf16. */
int wcn_conv_identity_supported(int32_t cin, int32_t cout, int32_t dtype);
/* The NARROW 1 x 1 x 1 layers (same reference line, helper.py:206-213): y[n, cout] = x[n, cin] * W (+ bias) for any
 * cin <= 128, cout <= 96 - the 3 -> 32 stem and the 96 -> 20 head of a MinkUNet and their input gradients - as one streaming
 * kernel (csrc/dense_rows.hip; the vendor GEMM serves a 20- or 3-column operand at 0.09-0.15 of the HBM rate).  `w` is the
 * layer's matrix as stored, row-major: [cin, cout], or [cout, cin] with w_transposed = 1 (the input gradient dy * W^T reads the
 * forward weight in place); its elements are fp32 (w_is_f32 = 1, master weights) or `dtype`, converted in the kernel - no
 * packed image.  The rows `x` are `dtype`, or fp32 with x_is_f32 = 1 (rounded to `dtype` on the fly, as a cast in front of the
 * product would).  `y`: [n, cout] in `dtype` (f16 / bf16), fp32 accumulation, optional fp32 bias[cout]. */
int wcn_dense_rows_supported(int32_t cin, int32_t cout, int32_t dtype);
int wcn_dense_rows(const void* x, int32_t x_is_f32, const void* w, int32_t w_is_f32, int32_t w_transposed, const float* bias,
                   void* y, int64_t n, int32_t cin, int32_t cout, int32_t dtype, wcn_stream_t stream);

/* Packed weight image consumed by the MFMA kernels (fragment order, zero padding).  `transpose`=1
 * packs w[k]^T (dgrad); `flip`=1 additionally reverses k (dgrad of a submanifold map reuses the
 * forward table: rnbr[n][k] == nbr[n][K-1-k]).  Returns bytes needed / fills `packed`.  `cin` / `cout` are the kernel-side
 * roles (reduce over cin, produce cout); the image holds num_offsets * round_up(cin, 64) * cout elements (the channel-split
 * kernels reduce in 64-channel chunks and zero-pad a trailing 32-channel chunk) - size `packed` with wcn_packed_weight_bytes;
 * `packed_bytes` = the size of the caller's buffer: WCN_ERROR_INVALID_PARAMETERS (nothing launched) if it is smaller than that. */
size_t wcn_packed_weight_bytes(int32_t num_offsets, int32_t cin, int32_t cout, int32_t dtype, int32_t transpose);
int wcn_pack_weight(const void* w, int32_t num_offsets, int32_t cin, int32_t cout, int32_t dtype,
                    int32_t transpose, int32_t flip, void* packed, size_t packed_bytes, wcn_stream_t stream);
/* the same packed image (`dtype` = WCN_F16 / WCN_BF16) straight from fp32 master weights, rounded to nearest even like the
 * framework's cast: one launch instead of cast + pack per convolution and direction. */
int wcn_pack_weight_f32(const float* w, int32_t num_offsets, int32_t cin, int32_t cout, int32_t dtype, int32_t transpose,
                        int32_t flip, void* packed, size_t packed_bytes, wcn_stream_t stream);
/* Both images a training step needs - forward (w [K, cin, cout] as is) and dgrad (kernel-side roles exchanged: transposed, and
 * k-flipped when `flip_dgrad`, i.e. for a submanifold map whose dgrad reads the forward table) - of an fp32 master weight in ONE
 * launch: an optimizer step invalidates both at once.  Shapes: wcn_pack_weight_pair_supported (both directions on the
 * channel-split kernels); sizes: wcn_packed_weight_bytes(K, cin, cout, dtype, 0) and (K, cout, cin, dtype, 1). */
int wcn_pack_weight_pair_supported(int32_t num_offsets, int32_t cin, int32_t cout, int32_t dtype);
int wcn_pack_weight_f32_pair(const float* w, int32_t num_offsets, int32_t cin, int32_t cout, int32_t dtype, int32_t flip_dgrad,
                             void* packed_fwd, size_t packed_fwd_bytes, void* packed_dgrad, size_t packed_dgrad_bytes,
                             wcn_stream_t stream);

/* y = gather-GEMM over a neighbour table.  Serves forward (x, packed w) and dgrad (dy, packed w^T).
 *   in   [n_in, cin]   out [n_out, cout]   nbr [n_out, kp]   mask [n_out, mw]   perm [n_out] or NULL
 *   bias fp32 [cout] or NULL: fused epilogue out += bias (reference adds it afterwards, helper.py:339-342)
 *   w:   for WCN_ALGO_REF the plain [K, cin, cout] tensor (or [K, cout', cin'] with w_transposed=1),
 *        for WCN_ALGO_MFMA the image made by wcn_pack_weight.
 * reference: _C.mask_gemm.fwd / .dgrad (mask_gemm_bindings.cu:2074-2101). */
/* Compact tables (round 6; round 5 carried the mask in column 31 of a dense row instead).  Where wcn_conv_compact_table_supported
 * (channel-split kernel shapes, wcn_kmap_compact_supported(K)), wcn_conv_gather_gemm / _fused / _f32out / wcn_conv_bn_backward accept
 * `mask` = NULL with `nbr` = the COMPACT table of wcn_kmap_build_binned(compact = 1): the rows are expanded into the kernel's index
 * slab - half the table bytes, and no gather of mask[perm[i]] (one 128-B line per row: 128 MB of fabric requests per launch at 1 M
 * rows for 4 MB of masks).  Dense tables (every other builder) always come with their mask. */
int wcn_conv_compact_table_supported(int32_t cin, int32_t cout, int32_t num_offsets, int32_t dtype);
int wcn_conv_gather_gemm(const void* in, const void* w, void* out, const int32_t* nbr,
                         const uint32_t* mask, const int32_t* perm, const float* bias, int64_t n_in,
                         int64_t n_out, int32_t cin, int32_t cout, int32_t num_offsets, int32_t dtype,
                         int32_t algo, int32_t w_transposed, int32_t k_flip, wcn_stream_t stream);

/* Channel groups in ONE launch (weight [K, G, Cin/G, Cout/G]; reference: group index on grid.z of the fused kernel,
 * MaskGemm_forward_64x64x32_1s_flat.h:117-123; sparse_conv.py:147-157).  Per-group widths cin_g x cout_g must be a shape
 * of the 32x32x16 kernels (wcn_mfma_grouped_supported); feature rows hold all groups side by side ([N, G*cin_g] in,
 * [N, G*cout_g] out), `bias` (optional) has G*cout_g entries.  wcn_pack_weight_grouped writes the G fragment-ordered
 * images back to back in one launch from the forward weight ([K, G, cin_g, cout_g], fp32 or the storage dtype);
 * transpose = 1, flip = 1 with swapped widths gives the dgrad images of a submanifold map, as for wcn_pack_weight. */
int wcn_mfma_grouped_supported(int32_t cin_g, int32_t cout_g, int32_t num_offsets, int32_t dtype);
int wcn_pack_weight_grouped(const void* w, int32_t w_is_f32, int32_t num_offsets, int32_t groups, int32_t cin_g,
                            int32_t cout_g, int32_t dtype, int32_t transpose, int32_t flip, void* packed, wcn_stream_t stream);
int wcn_conv_gather_gemm_grouped(const void* in, const void* w_packed, void* out, const int32_t* nbr, const uint32_t* mask,
                                 const int32_t* perm, const float* bias, int64_t n_in, int64_t n_out, int32_t cin_g,
                                 int32_t cout_g, int32_t groups, int32_t num_offsets, int32_t dtype, wcn_stream_t stream);
/* As wcn_conv_gather_gemm with algo = WCN_ALGO_MFMA, but the result is written as fp32 straight from the fp32
 * accumulators (in / w_packed are WCN_F16 or WCN_BF16).  Used for fp32 feature tensors: operands are cast to fp16 with
 * an exact power-of-two rescale by the caller, the product is scaled back in fp32 - the reference's production
 * treatment of fp32 inputs (warpconvnet/nn/functional/sparse_conv/detail/mask_gemm.py:72-103, 696-745). */
int wcn_conv_gather_gemm_f32out(const void* in, const void* w_packed, float* out, const int32_t* nbr,
                                const uint32_t* mask, const int32_t* perm, const float* bias, int64_t n_in,
                                int64_t n_out, int32_t cin, int32_t cout, int32_t num_offsets, int32_t dtype,
                                wcn_stream_t stream);

/* Gather GEMM with the elementwise chain of a ConvBlock folded into the store (the reference runs
 * conv -> BatchNorm1d -> ReLU (+ residual add) as separate kernels, models/mink_unet.py:31-53, 161-175):
 *   out[m][co] = act( (sum_k in[nbr[m][k]] . W[k] + bias[co]) * scale[co] + shift[co] + residual[m][co] )
 * evaluated in fp32 before the single rounding to the storage dtype.  bias / scale+shift / residual may be NULL
 * (scale and shift only together); BatchNorm in inference mode is scale = gamma / sqrt(var + eps),
 * shift = beta - mean * scale; relu != 0 applies max(., 0).  `residual` [n_out, cout] has the storage dtype and must not
 * alias `out`.  MFMA path only (f16 / bf16, shapes of wcn_mfma_gather_supported, `w_packed` from wcn_pack_weight). */
int wcn_conv_gather_gemm_fused(const void* in, const void* w_packed, void* out, const int32_t* nbr, const uint32_t* mask,
                               const int32_t* perm, const float* bias, const float* scale, const float* shift,
                               const void* residual, int32_t relu, int64_t n_in, int64_t n_out, int32_t cin, int32_t cout,
                               int32_t num_offsets, int32_t dtype, wcn_stream_t stream);

/* out[c] = sum_r in[r][c] in fp32 (bias gradient; reference: autograd of `out + bias`, helper.py:339-342).
 * Deterministic two-pass reduction; workspace: wcn_colsum_workspace(channels) bytes. */
size_t wcn_colsum_workspace(int32_t channels);
int wcn_colsum(const void* in, int64_t n, int32_t channels, int32_t dtype, float* out, void* workspace,
               size_t workspace_bytes, wcn_stream_t stream);

/* dw [K, cin, cout] fp32 (overwritten).  workspace: wcn_conv_wgrad_workspace(...) bytes.
 * reference: _C.mask_gemm.wgrad (mask_gemm_bindings.cu:2103-2116), fp32 output. */
size_t wcn_conv_wgrad_workspace(int32_t num_offsets, int32_t cin, int32_t cout, int32_t algo);
int wcn_conv_wgrad(const void* x, const void* dy, float* dw, const int32_t* in_maps,
                   const int32_t* out_maps, const int32_t* offsets, int64_t n_in, int64_t n_out,
                   int32_t cin, int32_t cout, int32_t num_offsets, int32_t dtype, int32_t algo,
                   void* workspace, size_t workspace_bytes, wcn_stream_t stream);

/* Weight gradient + bias gradient in one pass.  As wcn_conv_wgrad with algo = WCN_ALGO_MFMA, and additionally
 * bias_grad[co] = sum over rows r of dy[r][co] (fp32 [cout], overwritten): while the kernel streams bucket
 * `self_offset` - the offset whose pairs are (r, r) for every row r, i.e. the centre of an odd submanifold kernel
 * built from distinct coordinates (WCN_FLAG_DUPLICATE_COORD clear) - it also multiplies a ones-row fragment with the
 * dy fragments it holds, so the column sums come out of the matrix cores with no extra read of dy.  Deterministic.
 * Replaces the autograd reduce of `out + bias` (warpconvnet/nn/functional/sparse_conv/helper.py:339-342).
 * WCN_ERROR_UNSUPPORTED_CONFIG unless wcn_mfma_wgrad_bias_supported (then use wcn_conv_wgrad + wcn_colsum).
 * workspace: wcn_conv_wgrad_workspace(num_offsets, cin, cout, WCN_ALGO_MFMA) bytes. */
int wcn_mfma_wgrad_bias_supported(int32_t cin, int32_t cout, int32_t dtype);
int wcn_conv_wgrad_bias(const void* x, const void* dy, float* dw, const int32_t* in_maps, const int32_t* out_maps,
                        const int32_t* offsets, int64_t n_in, int64_t n_out, int32_t cin, int32_t cout,
                        int32_t num_offsets, int32_t dtype, int32_t self_offset, float* bias_grad, void* workspace,
                        size_t workspace_bytes, wcn_stream_t stream);

/* ---- depthwise sparse convolution (weight [K, C], same dtype as the features) ----------------------------------------
 * reference semantics: warpconvnet/nn/functional/sparse_conv_depth.py:227-306 (explicit depthwise forward/backward);
 * replaces _C.fma.implicit_fma / _C.fma.implicit_reduction (call sites sparse_conv_depth.py:309-421).
 *
 * wcn_dwconv_gather: out[r][c] = bias[c] + sum_k in[nbr[r][kt]][c] * w[kw][c], kt = k_flip ? K-1-kw : kw, fp32
 *   accumulate, fixed order, no atomics.  Forward: the forward table, k_flip = 0.  Dgrad: in = grad_output and either
 *   the forward table of a submanifold map with k_flip = 1 or a reverse table (wcn_kmap_reverse) with k_flip = 0.
 * wcn_dwconv_wgrad: dw[k][c] = sum over the pairs p of bucket k of x[in_p][c] * dy[out_p][c]  (fp32 [K, C],
 *   overwritten, deterministic).  workspace: wcn_dwconv_wgrad_workspace(num_offsets, channels) bytes. */
int wcn_dwconv_gather(const void* in, const void* w, void* out, const int32_t* nbr, const float* bias, int64_t n_in,
                      int64_t n_out, int32_t channels, int32_t num_offsets, int32_t dtype, int32_t k_flip,
                      wcn_stream_t stream);
size_t wcn_dwconv_wgrad_workspace(int32_t num_offsets, int32_t channels);
int wcn_dwconv_wgrad(const void* x, const void* dy, float* dw, const int32_t* in_maps, const int32_t* out_maps,
                     const int32_t* offsets, int64_t n_in, int64_t n_out, int32_t channels, int32_t num_offsets,
                     int32_t dtype, void* workspace, size_t workspace_bytes, wcn_stream_t stream);

/* ---- point-cloud front end (PointConv path) ----------------------------------------------------------------------------
 * wcn_knn_grid: exact k nearest reference points of every query (k <= 64), ascending by distance.  The caller bins the
 *   reference points of ONE batch element into a uniform grid: cell = floor((p - origin) / cell_size) clamped to dims,
 *   cell id = (z * dims[1] + y) * dims[0] + x; `ref_sorted` [N,3] fp32 holds the points sorted by cell id, `ref_ids` [N]
 *   their original row ids, `cell_start` [cells + 1] the CSR over cells.  out_index [M,k] int64 (-1 if fewer than k
 *   points exist), out_dist2 [M,k] fp32 squared distances (may be NULL).  Equals the brute-force answer (up to ties);
 *   replaces the chunked cdist + topk of warpconvnet/geometry/coords/search/knn.py:11-26, 108-142.
 * wcn_segment_reduce: out[m][c] = op over rows [row_splits[m], row_splits[m+1]) of in[.][c], op: 0 sum, 1 mean, 2 max,
 *   3 min (empty segment -> 0); arg_rows [M,C] int64 (max / min only, may be NULL) = row of the first extremum, for the
 *   backward pass.  Role of torch_scatter.segment_csr in warpconvnet/ops/reductions.py:36-75. */
int wcn_knn_grid(const float* ref_sorted, const int32_t* ref_ids, const int32_t* cell_start, const float origin[3],
                 float cell_size, const int32_t dims[3], const float* query, int64_t num_query, int32_t k,
                 int64_t* out_index, float* out_dist2, wcn_stream_t stream);
/* Radius search over the same cell layout (cell_size >= radius, so the 27 cells around a query hold every point within
 * the radius; test `dist^2 <= radius^2`).  Two passes like the reference (warpconvnet/csrc/radius_search_kernels.cu:30-134,
 * call site geometry/coords/search/radius.py:16-124): wcn_radius_grid_count writes counts [M] int32; the caller scans them
 * into splits [M+1] int64 (device) and sizes the outputs; wcn_radius_grid_write fills out_index [total] int32 (original row
 * ids) and out_dist [total] fp32 distances (may be NULL).  Rows are in cell-walk order, deterministic. */
int wcn_radius_grid_count(const float* ref_sorted, const int32_t* ref_ids, const int32_t* cell_start, const float origin[3],
                          float cell_size, const int32_t dims[3], const float* query, int64_t num_query, float radius,
                          int32_t* counts, wcn_stream_t stream);
int wcn_radius_grid_write(const float* ref_sorted, const int32_t* ref_ids, const int32_t* cell_start, const float origin[3],
                          float cell_size, const int32_t dims[3], const float* query, int64_t num_query, float radius,
                          const int64_t* splits, int32_t* out_index, float* out_dist, wcn_stream_t stream);
int wcn_segment_reduce(const void* in, const int64_t* row_splits, int64_t num_segments, int32_t channels, int32_t dtype,
                       int32_t op, void* out, int64_t* arg_rows, wcn_stream_t stream);

/* ---- PointConv edge pipeline in one pass ----------------------------------------------------------------------------
 * Replaces the op sequence of warpconvnet/nn/modules/point_conv.py:231-273 (features[neighbors] / repeat_interleave / cat
 * -> edge_transform_mlp -> row_reduction) for the default edge MLP of warpconvnet/nn/modules/mlp.py:124-177
 * (Linear - LayerNorm - ReLU - Linear - LayerNorm + identity or Linear shortcut): for query q and its k neighbours j (uniform k, a
 * power of two <= 32, `nbr` [n_query * k] int32 rows of in_feats):
 *     x = [in_feats[j] | q_feats[q] | in_xyz[j] - q_xyz[q] (nrel = 3, else 0)],   out[q] = sum or mean over j of
 *     LN2(W2 ReLU(LN1(W1 x + b1)) + b2) + (x  or  Ws x + bs)     fp32, matrix cores (v_mfma_f32_32x32x2_f32).
 * No [n_query * k, C] tensor is written to HBM in either direction; the backward recomputes the chain per 32-edge tile.
 * wcn_pointconv_supported: cin + cq + nrel == cout unless linear_shortcut, and the widths fit an instantiated tile
 *   (edge-in <= 64, hidden <= 128, cout <= 64; smaller widths run zero-padded).
 * wcn_pointconv_pack: torch-layout parameters (w1 [hidden][ein], w2 [cout][hidden], biases / LayerNorm weights may be
 *   NULL = 0; ws [cout][ein] / bs = the Linear shortcut or NULL) -> operand images, wcn_pointconv_packed_floats floats.
 * wcn_pointconv_edge_backward: d_in [n_in][cin] must be ZERO-FILLED by the caller (neighbour rows are accumulated with
 *   hardware fp32 atomics: the one non-deterministic sum of the path, like index_add in the reference's autograd);
 *   d_q [n_query][cq] is written; d_params (wcn_pointconv_grad_floats floats) = dW1 [hidden][ein] | db1 | dLN1.weight |
 *   dLN1.bias | dW2 [cout][hidden] | db2 | dLN2.weight | dLN2.bias (| dWs [cout][ein] | dbs with a Linear shortcut),
 *   reduced over the workgroups in fixed order. */
int wcn_pointconv_supported(int32_t cin, int32_t cq, int32_t nrel, int32_t hidden, int32_t cout, int32_t k,
                            int32_t linear_shortcut);
int64_t wcn_pointconv_packed_floats(int32_t ein, int32_t hidden, int32_t cout);
int64_t wcn_pointconv_grad_floats(int32_t ein, int32_t hidden, int32_t cout, int32_t linear_shortcut);
size_t wcn_pointconv_backward_workspace(int64_t n_query, int32_t k, int32_t ein, int32_t hidden, int32_t cout,
                                        int32_t linear_shortcut);
int wcn_pointconv_pack(const float* w1, const float* b1, const float* g1, const float* be1, const float* w2,
                       const float* b2, const float* g2, const float* be2, const float* ws, const float* bs, int32_t ein,
                       int32_t hidden, int32_t cout, float* packed, wcn_stream_t stream);
int wcn_pointconv_edge_forward(const float* in_feats, const float* q_feats, const float* in_xyz, const float* q_xyz,
                               const int32_t* nbr, int64_t n_query, int32_t k, int32_t cin, int32_t cq, int32_t nrel,
                               const float* packed, int32_t hidden, int32_t cout, float eps1, float eps2, int32_t mean,
                               int32_t linear_shortcut, float* out, wcn_stream_t stream);
int wcn_pointconv_edge_backward(const float* in_feats, const float* q_feats, const float* in_xyz, const float* q_xyz,
                                const int32_t* nbr, int64_t n_query, int32_t k, int32_t cin, int32_t cq, int32_t nrel,
                                const float* packed, int32_t hidden, int32_t cout, float eps1, float eps2, int32_t mean,
                                int32_t linear_shortcut, const float* grad_out, float* d_in, float* d_q, float* d_params,
                                void* workspace, size_t workspace_bytes, wcn_stream_t stream);
/* Bitwise-reproducible variant (uniform lists): per-EDGE input gradients d_edge [n_query * k][cin] by plain stores instead of
 * the fp32 atomics on d_in; the caller sums the rows of each input point in a fixed order.  Used when the framework asks
 * for deterministic algorithms; reference: autograd of `features[neighbors]` (point_conv.py:243-246) is an index_add. */
int wcn_pointconv_edge_backward_peredge(const float* in_feats, const float* q_feats, const float* in_xyz, const float* q_xyz,
                                        const int32_t* nbr, int64_t n_query, int32_t k, int32_t cin, int32_t cq, int32_t nrel,
                                        const float* packed, int32_t hidden, int32_t cout, float eps1, float eps2, int32_t mean,
                                        int32_t linear_shortcut, const float* grad_out, float* d_edge, float* d_q,
                                        float* d_params, void* workspace, size_t workspace_bytes, wcn_stream_t stream);

/* The same pipeline over RAGGED neighbour lists (radius search, warpconvnet/geometry/coords/search/radius.py): `nbr`
 * [n_edges] = the lists of the queries behind each other, `edge_q` [n_edges] the query of every edge (non-decreasing, i.e.
 * repeat_interleave(arange(n_query), list lengths)), `q_scale` [n_query] the reduction scale per query (1 / length for mean,
 * NULL = sum).  `out` (forward) and `d_q` (backward) must be ZERO-FILLED by the caller: a list may straddle two 32-edge
 * tiles, so list segments are added to their rows (hardware fp32 atomics).  Workspace: wcn_pointconv_backward_workspace
 * (n_edges, 1, ...). */
int wcn_pointconv_edge_forward_ragged(const float* in_feats, const float* q_feats, const float* in_xyz, const float* q_xyz,
                                      const int32_t* nbr, const int32_t* edge_q, const float* q_scale, int64_t n_edges,
                                      int64_t n_query, int32_t cin, int32_t cq, int32_t nrel, const float* packed,
                                      int32_t hidden, int32_t cout, float eps1, float eps2, int32_t linear_shortcut,
                                      float* out, wcn_stream_t stream);
int wcn_pointconv_edge_backward_ragged(const float* in_feats, const float* q_feats, const float* in_xyz, const float* q_xyz,
                                       const int32_t* nbr, const int32_t* edge_q, const float* q_scale, int64_t n_edges,
                                       int64_t n_query, int32_t cin, int32_t cq, int32_t nrel, const float* packed,
                                       int32_t hidden, int32_t cout, float eps1, float eps2, int32_t linear_shortcut,
                                       const float* grad_out, float* d_in, float* d_q, float* d_params, void* workspace,
                                       size_t workspace_bytes, wcn_stream_t stream);

/* Sparse pooling over a kernel map (REDUCE_AND_STRIDE, SparsePool / SparseMaxPool / SparseUnpool): out[m][c] =
 * reduce over the present neighbours k of in[tbl[m][k]][c]; `tbl` is the row-major neighbour table [n_out][row pitch of
 * num_offsets] (-1 = absent) that wcn_kmap_probe / wcn_kmap_from_csr / wcn_kmap_reverse produce.  op: 0 sum, 1 mean,
 * 2 max, 3 min; rows without a neighbour give 0.  `arg` (may be NULL) int32 [n_out][channels] = winning input row of
 * max / min (first extremum in offset order, -1 if none); `count` (may be NULL) int32 [n_out] = neighbours per row.
 * One pass, no intermediate tensors; replaces to_csr + index gather + torch_scatter.segment_csr
 * (warpconvnet/nn/functional/sparse_pool.py:84-110). */
int wcn_pool_gather(const void* in, const int32_t* tbl, int64_t n_in, int64_t n_out, int32_t channels, int32_t num_offsets,
                    int32_t dtype, int32_t op, void* out, int32_t* arg, int32_t* count, wcn_stream_t stream);
/* Gradient of max / min pooling, output-stationary and deterministic: dx[n][c] = sum over the present k of
 * dy[r][c] where r = tbl[n][k] and arg[r][c] == n.  `tbl` is the REVERSE table [n_in][pitch] (pooled rows per input
 * row), `arg` the forward pass's [n_out][channels]. */
int wcn_pool_select(const void* dy, const int32_t* arg, const int32_t* tbl, int64_t n_in, int64_t n_out, int32_t channels,
                    int32_t num_offsets, int32_t dtype, void* dx, wcn_stream_t stream);

/* ---- BatchNorm over sparse feature tensors [n, channels] (f32 / f16 / bf16 storage, fp32 statistics) --------------------
 * The elementwise chain behind every sparse convolution (reference models/mink_unet.py:31-53: SparseConv3d ->
 * nn.BatchNorm1d -> ReLU, the framework's stock BatchNorm kernels).  All passes stream the tensor once; the two
 * reductions are fixed-order two-level sums (deterministic).  `workspace` >= wcn_bn_workspace(channels) bytes.
 *   wcn_bn_stats            mean[c], var[c] (biased, /n) in one pass (sums around the pivot x[0][c]).
 *   wcn_bn_apply            y = x * scale[c] + shift[c]; relu != 0: max(., 0).  (training: scale = gamma * rstd,
 *                           shift = beta - mean * scale; inference: the same from the running statistics.)
 *   wcn_bn_backward_reduce  sum_dy[c] = sum_r g, sum_dy_xhat[c] = sum_r g * (x - mean) * rstd, where g = dy, or 0 where the
 *                           fused ReLU of the forward pass stored a zero: `relu_scale` / `relu_shift` (both or neither) are the
 *                           scale / shift the forward applied, the mask is recomputed from x with the same fused
 *                           multiply-add and rounding - the forward output is not read again.
 *   wcn_bn_backward_apply   dx = gamma * rstd * (g - sum_dy / n - xhat * sum_dy_xhat / n); gamma may be NULL (= 1). */
size_t wcn_bn_workspace(int32_t channels);
/*   wcn_bn_stats_fold       wcn_bn_stats plus, in the same launches, everything a training step derives from the
 *                           statistics: rstd = 1/sqrt(var + eps), scale = gamma * rstd, shift = beta - mean * scale (gamma /
 *                           beta may be NULL) and the in-place update of the fp32 running statistics (NULL: none) with
 *                           `momentum` and the unbiased variance - the ~10 tiny framework kernels of a BatchNorm step;
 *                           `num_batches_tracked` (device int64 scalar, NULL: none) is incremented by one.
 *   wcn_bn_fold             inference: mean / rstd / scale / shift from the running statistics, one launch. */
int wcn_bn_stats_fold(const void* x, int64_t n, int32_t channels, int32_t dtype, const float* gamma, const float* beta,
                      float* running_mean, float* running_var, float momentum, float eps, float* mean, float* var,
                      float* rstd, float* scale, float* shift, int64_t* num_batches_tracked, void* workspace,
                      size_t workspace_bytes, wcn_stream_t stream);
int wcn_bn_fold(const float* running_mean, const float* running_var, const float* gamma, const float* beta, float eps,
                int32_t channels, float* mean, float* rstd, float* scale, float* shift, wcn_stream_t stream);
int wcn_bn_stats(const void* x, int64_t n, int32_t channels, int32_t dtype, float* mean, float* var, void* workspace,
                 size_t workspace_bytes, wcn_stream_t stream);
int wcn_bn_apply(const void* x, int64_t n, int32_t channels, int32_t dtype, const float* scale, const float* shift,
                 int32_t relu, void* y, wcn_stream_t stream);
int wcn_bn_backward_reduce(const void* dy, const void* x, const float* relu_scale, const float* relu_shift, int64_t n,
                           int32_t channels, int32_t dtype, const float* mean, const float* rstd, float* sum_dy,
                           float* sum_dy_xhat, void* workspace, size_t workspace_bytes, wcn_stream_t stream);
int wcn_bn_backward_apply(const void* dy, const void* x, const float* relu_scale, const float* relu_shift, int64_t n,
                          int32_t channels, int32_t dtype, const float* mean, const float* rstd, const float* gamma,
                          const float* sum_dy, const float* sum_dy_xhat, void* dx, wcn_stream_t stream);
/* The tail of a residual block, z = ReLU(BN(x) + r) - the reference runs it as three passes over the feature tensor
 * (`models/mink_unet.py:160-172`: conv2's BatchNorm, `out += identity`, `relu(out)`).  Here it rides on the two BatchNorm passes:
 *   wcn_bn_apply_residual          y = [ReLU](round(x * scale + shift) + residual), every intermediate rounded to the storage
 *                                  type as the three modules would (bit-identical to them).
 *   wcn_bn_backward_reduce_masked  as wcn_bn_backward_reduce with g = dy where the stored output z is positive, else 0.
 *   wcn_bn_backward_apply_masked   as wcn_bn_backward_apply with that mask; `dres` (may be NULL) also receives g itself -
 *                                  the gradient of the residual branch. */
int wcn_bn_apply_residual(const void* x, const void* residual, int64_t n, int32_t channels, int32_t dtype, const float* scale,
                          const float* shift, int32_t relu, void* y, wcn_stream_t stream);
int wcn_bn_backward_reduce_masked(const void* dy, const void* x, const void* z, int64_t n, int32_t channels, int32_t dtype,
                                  const float* mean, const float* rstd, float* sum_dy, float* sum_dy_xhat, void* workspace,
                                  size_t workspace_bytes, wcn_stream_t stream);
int wcn_bn_backward_apply_masked(const void* dy, const void* x, const void* z, int64_t n, int32_t channels, int32_t dtype,
                                 const float* mean, const float* rstd, const float* gamma, const float* sum_dy,
                                 const float* sum_dy_xhat, void* dx, void* dres, wcn_stream_t stream);

/* Layer entries: the BatchNorm of a training step in ONE call per direction (the host side of a network pays per call, not per
 * launch: tools/host_attrib.py).  `stats` [5][channels] fp32 = mean | rstd | scale | shift | biased variance, written by the
 * forward and read by the backward.  wcn_bn_train_forward = wcn_bn_stats_fold + wcn_bn_apply[_residual] (`residual` may be NULL);
 * wcn_bn_train_backward = wcn_bn_backward_reduce[_masked] + wcn_bn_backward_apply[_masked]: `z` (stored output of a residual
 * tail, its sign is the ReLU mask) or NULL (mask recomputed from x when `relu`), `sums` [2][channels] = sum_dy | sum_dy_xhat,
 * `dx` NULL = sums only, `dres` (masked gradient, residual tails) may be NULL, `training` 0 = statistics were constants. */
int wcn_bn_train_forward(const void* x, const void* residual, int64_t n, int32_t channels, int32_t dtype, const float* gamma,
                         const float* beta, float* running_mean, float* running_var, float momentum, float eps,
                         int64_t* num_batches_tracked, int32_t relu, float* stats, void* y, void* workspace,
                         size_t workspace_bytes, wcn_stream_t stream);
int wcn_bn_train_backward(const void* dy, const void* x, const void* z, int32_t relu, int64_t n, int32_t channels, int32_t dtype,
                          const float* stats, const float* gamma, int32_t training, float* sums, void* dx, void* dres,
                          void* workspace, size_t workspace_bytes, wcn_stream_t stream);
/* The same with a row pitch for `dy` (elements; >= channels, 0 = channels): the gradient a channel concatenation hands to one of
 * its inputs is a column slice of a wider row-major tensor (reference models/mink_unet.py:392-404: `cat` of the up-sampled and
 * the skip tensor) - read in place instead of through a contiguous copy.  16-B pieces need dy and dy_ld * sizeof(T) 16-B
 * aligned; anything else takes the element path. */
int wcn_bn_train_backward_ld(const void* dy, int64_t dy_ld, const void* x, const void* z, int32_t relu, int64_t n, int32_t channels,
                             int32_t dtype, const float* stats, const float* gamma, int32_t training, float* sums, void* dx,
                             void* dres, void* workspace, size_t workspace_bytes, wcn_stream_t stream);

/* Layer entry: backward of SparseConv3d -> BatchNorm (-> ReLU | residual tail) (reference models/mink_unet.py:31-53, 160-172 run
 * as three autograd nodes) in one call: wcn_bn_train_backward into `dy_conv` ([n_out][cout], caller's buffer), then
 * wcn_conv_gather_gemm on the reverse tables with the transposed packed weight (`dx` NULL: skipped), then wcn_conv_wgrad (`dw`
 * NULL: skipped).  16-bit dtypes, MFMA shapes (wcn_mfma_gather_supported in both directions, wcn_mfma_wgrad_supported). */
int wcn_conv_bn_backward(const void* grad_out, const void* x, const void* y, const void* z, int32_t relu, const float* stats,
                         const float* gamma, int32_t training, float* sums, void* dy_conv, void* dres, const void* w_packed_dgrad,
                         const int32_t* rev_nbr, const uint32_t* rev_mask, const int32_t* rev_perm, int32_t flip, void* dx,
                         const int32_t* in_maps, const int32_t* out_maps, const int32_t* offsets, float* dw, void* wgrad_workspace,
                         size_t wgrad_workspace_bytes, int64_t n_in, int64_t n_out, int32_t cin, int32_t cout, int32_t num_offsets,
                         int32_t dtype, void* bn_workspace, size_t bn_workspace_bytes, wcn_stream_t stream);
/* ... with a row pitch for `grad_out` (see wcn_bn_train_backward_ld). */
int wcn_conv_bn_backward_ld(const void* grad_out, int64_t grad_out_ld, const void* x, const void* y, const void* z, int32_t relu,
                            const float* stats, const float* gamma, int32_t training, float* sums, void* dy_conv, void* dres,
                            const void* w_packed_dgrad, const int32_t* rev_nbr, const uint32_t* rev_mask, const int32_t* rev_perm,
                            int32_t flip, void* dx, const int32_t* in_maps, const int32_t* out_maps, const int32_t* offsets,
                            float* dw, void* wgrad_workspace, size_t wgrad_workspace_bytes, int64_t n_in, int64_t n_out,
                            int32_t cin, int32_t cout, int32_t num_offsets, int32_t dtype, void* bn_workspace,
                            size_t bn_workspace_bytes, wcn_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* WCN_H_ */
