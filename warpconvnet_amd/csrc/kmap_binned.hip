// kmap_binned.hip - LDS-binned neighbour search for submanifold kernel maps (output coords == input coords).
//
// A global hash probe per (voxel, offset) is 27 random 16-B reads per voxel (27 M for a 1 M-voxel scene): HBM /
// fabric latency bound.  Here the hash table is BLOCK-level only: voxels are binned into 8x8x8 blocks, every occupied
// block owns a dense 512-cell sub-grid of row ids in HBM (the "cell table"), and one WAVEFRONT per block stages its
// sub-grid plus the halo taken from the 26 neighbouring sub-grids into an LDS grid and answers all K probes of the
// block's voxels from LDS.  Global traffic = one plain 4-B store per voxel + streaming sub-grid reads + full-line
// neighbour-row writes.  (Round 1 binned with a counting sort - one RETURNING atomic per voxel, measured 42 us per
// million on MI355X for any scope or address spread, tools/cell_probe.hip, against 16 us for plain stores - and paid
// four more passes for positions, sizes, scan and scatter.)
//
//   cell_prepare   clears the block table and the counters, builds the halo gather list for this kernel geometry
//   cell_insert<0> every 16th voxel: find-or-create its block (CAS on first touch only); the creating WAVE hands out the
//                  dense block id and clears the block's sub-grid
//   cell_insert<1> every voxel: block lookup (plain cached reads) and cells[id][cell] = row.  Voxels of the few blocks
//                  the sample missed are created here and marked deferred (their sub-grid is being cleared by another wave)
//   cell_finish    the deferred voxels (normally < 2 %), and the ids of the 27 neighbour blocks of every block
//   cell_neighbors one wave per block: LDS grid from the block's own 2 KB + the halo list, occupied cells enumerated
//                  by a wave prefix sum, then one LANE per (voxel, offset): the neighbour row is written as one
//                  contiguous line and the mask is a wave ballot.  One barrier, no atomics; the next block's loads are
//                  in flight under the probe loop of the current one.
//
// Duplicate coordinates: the hash path keeps the smallest row.  A plain store keeps an arbitrary one, and only the
// kept row is enumerated; the other rows keep the "unwritten" mark the insert pass puts into their mask, and the tally
// pass (kmap.hip) copies the winner's table row to them.  If a kept row is not the smallest, the tally pass raises
// WCN_FLAG_NEED_STRICT and the host rebuilds with strict = 1 (atomicMin instead of the plain store).
//
// Semantics equal wcn_hash_insert + wcn_kmap_probe with stride 1 (incl. 18-bit coordinate wrap of the packed
// key: neighbour blocks are looked up with wrapped block coordinates, positions are block-relative).
// Reference behaviour replaced: warpconvnet/csrc/cuhash_hash_table.cu:179-220, cuhash_kernel_map.cu:93-134.
#include "kmap_cells.h"

namespace wcn {

constexpr int kInsertSample = 8;  // cell_insert<0>: 1 voxel in 8 goes first (see the kernel; 16 left 1.6 % of the blocks of a 64-voxel-per-block
// scene to the second pass, whose creation path is the expensive one: uniform 1 M scene 58.5 -> 53.4 us for the three insert launches, surface
// scene 92 -> 76 us; 4 costs the sampled pass more than it saves: 60 / 70 us)
constexpr int kInsertThreads = 512;
constexpr int kRowBufPitch = 66;  // cell_neighbors<COMPACT>: LDS row buffer [16 words + 1 spare][66]
constexpr int kRowBufInts = (kCompactPitch + 1) * kRowBufPitch;
#ifndef WCN_NB_WPE
#define WCN_NB_WPE 3
#endif
constexpr int kNbWavesPerSimd = WCN_NB_WPE;  // cell_neighbors<COMPACT>: register budget 512 / this
constexpr int kNbThreads = 256;    // cell_neighbors: up to 4 independent waves per workgroup (fewer when the LDS grid is large)

__device__ __forceinline__ int wrap_blk(int v) {  // wrap to the signed range of the block coordinate field
  const int bits = kBlkCoordBits;
  v &= (1 << bits) - 1;
  return (v ^ (1 << (bits - 1))) - (1 << (bits - 1));
}

// Clears the block table, the counters and the caller's status word, and writes the halo gather list of this kernel
// geometry: entry = dir (5 bits) << 27 | cell inside the neighbour's sub-grid (9 bits) << 16 | LDS grid index (16 bits),
// ordered by direction and then by neighbour cell, so adjacent lanes read adjacent cells.
__global__ void cell_prepare_kernel(uint4* __restrict__ slots, int64_t capacity, int32_t* __restrict__ ctr,
                                    uint32_t* __restrict__ halo, CellGeom g, int32_t* __restrict__ status) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < capacity) slots[i] = make_uint4(0u, 0u, 0xFFFFFFFFu, 0u);
  if (i < 64) ctr[i] = 0;
  if (i == 0) *status = 0;
  if (i < g.halo_cells) {
    const int h3[3] = {g.hx, g.hy, g.hz};
    int t = (int)i, dir = 0;
    int sx = 0, sy = 0, sz = 0;
    for (; dir < 27; ++dir) {
      if (dir == 13) continue;
      const int dd[3] = {dir / 9 - 1, (dir / 3) % 3 - 1, dir % 3 - 1};
      sx = dd[0] ? h3[0] : kBlk; sy = dd[1] ? h3[1] : kBlk; sz = dd[2] ? h3[2] : kBlk;
      const int cells = sx * sy * sz;
      if (t < cells) break;
      t -= cells;
    }
    const int ddx = dir / 9 - 1, ddy = (dir / 3) % 3 - 1, ddz = dir % 3 - 1;
    const int tz = t % sz, ty = (t / sz) % sy, tx = t / (sz * sy);
    // position inside the neighbour block: its low cells for a +1 neighbour, its high cells for a -1 neighbour
    const int lx = ddx < 0 ? kBlk - g.hx + tx : tx;
    const int ly = ddy < 0 ? kBlk - g.hy + ty : ty;
    const int lz = ddz < 0 ? kBlk - g.hz + tz : tz;
    const int X = lx + kBlk * ddx + g.hx, Y = ly + kBlk * ddy + g.hy, Z = lz + kBlk * ddz + g.hz;
    halo[i] = ((uint32_t)dir << 27) | ((uint32_t)((lx * kBlk + ly) * kBlk + lz) << 16) | (uint32_t)(X * g.px + Y * g.py + Z);
  }
}

__device__ __forceinline__ void store_cell(int32_t* cell, int row, bool strict) {
  if (strict) __hip_atomic_fetch_min(reinterpret_cast<uint32_t*>(cell), (uint32_t)row, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else *cell = row;
}

__device__ __forceinline__ int cell_of(const int4& c) {
  return ((c.y & (kBlk - 1)) * kBlk + (c.z & (kBlk - 1))) * kBlk + (c.w & (kBlk - 1));
}

// PHASE 0: every kInsertSample-th voxel creates its block.  PHASE 1: every voxel stores its cell; blocks the sample
// missed are created here and their voxels marked "deferred" in the mask (cell_finish stores those cells).
//
// Why two launches: with ONE launch half a million resident threads meet an empty table at the same instant and all
// of them CAS the same few thousand block keys (same-address atomics serialise); after the sampled pass nearly every
// block exists and the rest of the voxels only read.  Why ids / cleared sub-grids are consumed by LATER launches only:
// the L2 caches of the 8 XCDs are not coherent with each other inside a kernel, so "publish an id, then let another
// wave store into the sub-grid" would need a device-scope release/acquire (an L2 write-back) per block.
// Same-address atomics serialise at ~11 ns each on MI355X: deferred voxels are marked in place (a "deferred list" cursor
// was ~10 k atomics = 110 us of this kernel's first version) and block ids are handed out once per WORKGROUP.
template <int PHASE>
__global__ __launch_bounds__(kInsertThreads) void cell_insert_kernel(BSlot* __restrict__ slots, uint32_t cmask,
                                                          const int4* __restrict__ coords, int64_t n, CellTable t,
                                                          int32_t* __restrict__ status, int kp, int mw,
                                                          int32_t* __restrict__ nbr, uint32_t* __restrict__ mask,
                                                          int strict) {
  // kp: ints per table row - the dense pitch, or kCompactPitch with the top bit set for COMPACT rows (kmap_cells.h)
  const bool compact = kp < 0;
  kp &= 0x7FFFFFFF;
  const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t i = PHASE == 0 ? tid * kInsertSample : tid;
  const int lane = threadIdx.x & 63;
  bool live = i < n;
  int4 c = make_int4(0, 0, 0, 0);
  if (live) {
    c = coords[i];
    if (!coord_in_range(c.x, c.y, c.z, c.w)) {
      live = false;
      if (PHASE == 1) {
        atomicOr(status, (int)WCN_FLAG_COORD_RANGE);
        // the voxel is in no block, so cell_neighbors never visits it: give its table row defined ("no neighbour")
        // content - consumers may already be queued behind this build when the host sees the flag
        for (int k = 0; k < kp; ++k) nbr[i * kp + k] = compact ? 0 : -1;  // (compact row: word 0 = mask = no offsets)
        for (int w = 0; w < mw; ++w) mask[i * mw + w] = 0u;
      }
    }
  }
  int found = -1;  // slot of the voxel's block
  int id = -1;     // dense block id when it may be used by this launch
  bool created = false;
  uint64_t key = 0;
  if (live) {
    key = pack_key(c.x, c.y >> kBlkShift, c.z >> kBlkShift, c.w >> kBlkShift);
    uint32_t s = hash_slot(key, cmask);
    int idv = -1;
    for (uint32_t a = 0; a <= cmask; ++a) {
      unsigned long long* kptr = reinterpret_cast<unsigned long long*>(&slots[s].key);
      // optimistic cached read first, key and id in ONE 16-B load: a key, once written, never changes, so a matching
      // value is final (and so is the id of a block of an EARLIER launch); an empty - possibly stale - key falls through
      // to the coherent read below
      const uint4 v = *reinterpret_cast<const uint4*>(slots + s);
      unsigned long long cur = ((unsigned long long)v.y << 32) | v.x;
      if (cur == key) { found = (int)s; idv = (int)v.z; break; }
      if (cur == 0ull) {
        // sampled pass: most keys really are absent, go straight to the CAS (it returns the current value either way);
        // second pass: an empty-looking slot is almost always a stale line, a coherent load is cheaper than an atomic
        if (PHASE == 1) cur = __hip_atomic_load(kptr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (cur == 0ull) {
          cur = atomicCAS(kptr, 0ull, (unsigned long long)key);
          if (cur == 0ull) {  // this thread created the block (its dense id is handed out below)
            created = true;
            found = (int)s;
            break;
          }
        }
        if (cur == key) { found = (int)s; break; }
      }
      s = (s + 1) & cmask;
    }
    if (found < 0) atomicOr(status, (int)WCN_FLAG_TABLE_FULL);
    if (PHASE == 1 && idv >= 0 && !(idv & kIdLateBit)) id = idv;
  }
  // dense block ids: ONE counter update per workgroup (sampled pass: nearly every wave creates blocks) or per wave
  // (second pass: a few hundred creations in all, and no barrier for the million threads that only look up)
  __shared__ int s_made[kInsertThreads / 64 + 1];
  const int wave = threadIdx.x >> 6;
  const unsigned long long makers = __ballot(created);
  if (PHASE == 0) {
    if (lane == 0) s_made[wave] = __popcll(makers);
    __syncthreads();
    if (threadIdx.x == 0) {
      int tot = 0;
      for (int w = 0; w < kInsertThreads / 64; ++w) {
        const int c = s_made[w];
        s_made[w] = tot;
        tot += c;
      }
      s_made[kInsertThreads / 64] = tot > 0 ? atomicAdd(&t.ctr[0], tot) : 0;
    }
    __syncthreads();
  }
  if (makers != 0ull) {
    int base;
    if (PHASE == 0) {
      base = s_made[kInsertThreads / 64] + s_made[wave];
    } else {
      const int leader = __ffsll((long long)makers) - 1;
      base = 0;
      if (lane == leader) base = atomicAdd(&t.ctr[0], __popcll(makers));
      base = __shfl(base, leader);
    }
    int my_id = base + __popcll(makers & ((1ull << lane) - 1ull));
    if (!created) my_id = -1;
    if (my_id >= t.max_blocks) {
      atomicOr(status, (int)WCN_FLAG_TABLE_FULL);
      my_id = -1;
    }
    if (my_id >= 0) {
      t.blk_key[my_id] = key;
      slots[found].id = my_id | (PHASE == 1 ? kIdLateBit : 0);  // read by LATER launches only
    }
    // the wave clears the sub-grids of the blocks it created: 2 KB each, two 16-B stores per lane
    unsigned long long todo = makers;
    while (todo != 0ull) {
      const int src = __ffsll((long long)todo) - 1;
      todo &= todo - 1ull;
      const int bid = __shfl(my_id, src);
      if (bid >= 0) {
        int4* gcells = reinterpret_cast<int4*>(t.cells + (int64_t)bid * kCells);
        gcells[lane] = make_int4(-1, -1, -1, -1);
        gcells[lane + 64] = make_int4(-1, -1, -1, -1);
      }
    }
  }
  if (PHASE == 0) return;
  // the mark is cleared by cell_neighbors for the row each cell keeps
  if (live) mask[i * mw + (mw - 1)] = (id >= 0 || found < 0) ? kMaskUnwritten : kMaskDeferred;
  if (id >= 0) store_cell(t.cells + (int64_t)id * kCells + cell_of(c), (int)i, strict != 0);
}

// (a) cells of the voxels the second insert pass deferred; (b) ids of the 27 neighbour blocks of every block - all
// blocks exist by now, and a kernel of its own runs the two dependent lookups of ALL (block, direction) pairs at once
// instead of one wave at a time in front of its LDS staging.
__global__ __launch_bounds__(256) void cell_finish_kernel(const BSlot* __restrict__ slots, uint32_t cmask,
                                                          const int4* __restrict__ coords, int64_t n, CellTable t,
                                                          CellGeom g, int mw, const uint32_t* __restrict__ mask,
                                                          int strict) {
  const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t nthreads = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = tid; i < n; i += nthreads) {
    if (mask[i * mw + (mw - 1)] != kMaskDeferred) continue;
    const int4 c = coords[i];
    const int s = block_find(slots, cmask, pack_key(c.x, c.y >> kBlkShift, c.z >> kBlkShift, c.w >> kBlkShift));
    const int id = s >= 0 ? slots[s].id : -1;
    if (id < 0) continue;  // block table overflow: flagged by the insert pass
    store_cell(t.cells + (int64_t)(id & ~kIdLateBit) * kCells + cell_of(c), (int)i, strict != 0);
  }
  // neighbour table: one wave per block, lane = direction
  const int lane = threadIdx.x & 63;
  int nblocks = t.ctr[0];
  if (nblocks > t.max_blocks) nblocks = (int)t.max_blocks;
  for (int64_t seq = tid >> 6; seq < nblocks; seq += nthreads >> 6) {
    const int id = (int)seq;
    int nid = -1;
    if (lane < 27) {
      const int ddx = lane / 9 - 1, ddy = (lane / 3) % 3 - 1, ddz = lane % 3 - 1;
      const bool needed = (ddx == 0 || g.hx > 0) && (ddy == 0 || g.hy > 0) && (ddz == 0 || g.hz > 0);
      if (lane == 13) nid = id;
      else if (needed) {
        const uint64_t key = t.blk_key[id];
        const int b = (int)((key >> 54) & kBatchMask);
        const int bx = wrap_blk((int)((key >> 36) & kCoordMask));
        const int by = wrap_blk((int)((key >> 18) & kCoordMask));
        const int bz = wrap_blk((int)(key & kCoordMask));
        const int s = block_find(slots, cmask, pack_key(b, wrap_blk(bx + ddx), wrap_blk(by + ddy), wrap_blk(bz + ddz)));
        if (s >= 0) {
          nid = slots[s].id;
          if (nid >= 0) nid &= ~kIdLateBit;
        }
      }
    }
    if (lane < 32) t.nbtab[(int64_t)id * 32 + lane] = nid;
  }
}

#ifdef WCN_PROF
__device__ unsigned long long g_bprof[4096 * 12];
// per wave: time accumulated up to stamp i from the previous stamp, over all the wave's blocks (slot 0: the wait at the loop top)
#define BSTAMP(i) do { const unsigned long long now_ = wall_clock64(); prof_acc[i] += now_ - prof_prev; prof_prev = now_; } while (0)
#else
#define BSTAMP(i)
#endif

// One wave per block.  LDS: the halo gather list (shared by the workgroup; not for COMPACT with up to 8 rounds), then per wave
// grid[g.cells] row ids (-1 = empty), own[512] u16 grid indices of the block's occupied cells and (COMPACT) the 64 rows being
// assembled.
// FAST: every byte offset into nbr / mask fits 31 bits and every row index 24 bits (n < 2^24, n * kp * 4 < 2^31): the
// addresses of the stores are one full-rate 24-bit multiply-add instead of two quarter-rate 64-bit ones.
// Dense rows (COMPACT = false): LPR lanes per voxel, one lane per (voxel, offset), masks by ballot, every lane stores its answer.
// COMPACT (one mask word, 17 <= K <= 31): the voxel's row is 16 ints - its mask, then the neighbour rows of its SET offsets in
// ascending k (kmap_cells.h: kCompactPitch) - instead of 32 columns of which 4 - 9 hold a neighbour on the scenes of the bench:
// half the bytes for every later reader of the table (pair scatter, both gather GEMMs).  One lane per VOXEL probes the K offsets
// and appends to its row in LDS; 64 rows leave as 16-B pieces, 4 lanes a row (see the probe section).  A voxel with more than
// kCompactIds neighbours raises WCN_FLAG_ROW_OVERFLOW: the host rebuilds with dense rows.
// Register budget: amdgpu_waves_per_eu(3) = 168 VGPRs (the compact path uses ~150); the host sizes the compact grid to exactly
// the resident workgroups (3 waves a SIMD measured best: 58.3 us against 61.8 at 4 and 62.4 at 2).
template <int LPR, bool FAST, bool COMPACT>
__global__ __launch_bounds__(kNbThreads) __attribute__((amdgpu_waves_per_eu(kNbWavesPerSimd))) void cell_neighbors_kernel(CellTable t, const uint32_t* __restrict__ halo,
                                                                    CellGeom g, int K, int kp, int mw,
                                                                    int32_t* __restrict__ nbr,
                                                                    uint32_t* __restrict__ mask,
                                                                    int32_t* __restrict__ status) {
  extern __shared__ int s_mem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t* s_halo = reinterpret_cast<uint32_t*>(s_mem);
  const int rounds = (g.halo_cells + 63) >> 6;  // 64-lane gather rounds
  // The first 8 rounds of the halo list (all of a 3x3x3 kernel's) live in registers - the list is the same for every block -
  // and only a larger halo (or the dense-row instantiations, which keep their registers for 6 waves a SIMD) reads it from LDS.
  const bool halo_regs = COMPACT && rounds <= 8;
  const int halo_pad = halo_regs ? 0 : rounds * 64;
  // ints per wave: grid, null cell, own list (+ padding entries), COMPACT: the 64 rows being assembled
  const int per_wave = g.cells + 4 + kCells / 2 + 8 + (COMPACT ? kRowBufInts : 0);
  int* s_grid = s_mem + halo_pad + wave * per_wave;
  // byte offsets (into s_grid) of the block's occupied cells, padded to whole probe trips with the NULL cell: a cell
  // behind the grid that holds -1, so the probe loop needs no bounds checks (row -1 = nothing stored)
  unsigned short* s_own = reinterpret_cast<unsigned short*>(s_grid + g.cells + 4);
  const int null_cell = g.cells;
  for (int h = threadIdx.x; h < halo_pad; h += blockDim.x) s_halo[h] = h < g.halo_cells ? halo[h] : 0xFFFFFFFFu;
  if (lane == 0) s_grid[null_cell] = -1;
  __syncthreads();  // the only workgroup barrier: the waves are independent from here on

  int nblocks = t.ctr[0];
  if (nblocks > t.max_blocks) nblocks = (int)t.max_blocks;
  const int gwave = blockIdx.x * (blockDim.x / 64) + wave;
  const int nwaves = gridDim.x * (blockDim.x / 64);
  constexpr int kVoxPerIter = 64 / LPR;
  const int sub = lane % LPR, vsel = lane / LPR;
  const int num_chunks = (kp + LPR - 1) / LPR;
  // Software pipeline over the wave's blocks (stride nwaves).  While block i is probed out of LDS, the halo cells of
  // block i+1 (gathered with the neighbour ids that arrived during block i-1) and the neighbour ids + own 2 KB of block
  // i+2 are in flight: a block costs two dependent memory round trips, and a wave owns only two or three blocks.
  uint32_t he[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    const int idx = u * 64 + lane;
    uint32_t e = COMPACT && idx < g.halo_cells ? halo[idx] : 0xFFFFFFFFu;
    // no entry: neighbour slot 31 (never a block) and a dump cell behind the NULL cell, so the rounds need no validity checks
    if (e == 0xFFFFFFFFu) e = (31u << 27) | (uint32_t)(g.cells + 1);
    he[u] = e;
  }
  auto load_head = [&](int seq, int& nb, int4& v0, int4& v1) {
    nb = -1;
    v0 = make_int4(-1, -1, -1, -1);
    v1 = v0;
    if (seq < nblocks) {
      if (lane < 27) nb = t.nbtab[(int64_t)seq * 32 + lane];
      const int4* own = reinterpret_cast<const int4*>(t.cells + (int64_t)seq * kCells) + lane * 2;
      v0 = own[0];
      v1 = own[1];
    }
  };
  int seq = gwave;
  int nb_c, nb_n;
  int4 c0, c1, n0, n1;
  load_head(seq, nb_c, c0, c1);
  load_head(seq + nwaves, nb_n, n0, n1);
  int hv[8];  // halo values of the current block (3x3x3: all of them; larger halos reload per 8 rounds below)
  auto gather = [&](int nb, int r0) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      hv[u] = -1;
      if (r0 + u < rounds) {
        const uint32_t e = s_halo[(r0 + u) * 64 + lane];
        const int nid = __shfl(nb, (int)((e >> 27) & 31u));
        if (e != 0xFFFFFFFFu && nid >= 0) hv[u] = t.cells[(int64_t)nid * kCells + ((e >> 16) & (kCells - 1))];
      }
    }
  };
  // rounds 0..7 out of the registers, branch-free: all the neighbour ids first (one LDS round trip for the 8 shuffles, where
  // the loop above pays two dependent ones PER ROUND - in-kernel stamps: 2.4 us a block went into issuing these 8 loads), an
  // absent neighbour block reads block 0 instead and is turned into "empty" when the value is scattered.
  int nidv[8];
  auto gather8 = [&](int nb) {
#pragma unroll
    for (int u = 0; u < 8; ++u) nidv[u] = __shfl(nb, (int)(he[u] >> 27));
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (u < rounds) {
        const int safe = nidv[u] < 0 ? 0 : nidv[u];
        hv[u] = t.cells[(int64_t)safe * kCells + ((he[u] >> 16) & (kCells - 1))];
      }
  };
  auto scatter8 = [&]() {
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (u < rounds) s_grid[he[u] & 0xFFFFu] = nidv[u] >= 0 ? hv[u] : -1;
  };
  auto scatter_halo = [&](int r0) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (r0 + u < rounds) {
        const uint32_t e = s_halo[(r0 + u) * 64 + lane];
        if (e != 0xFFFFFFFFu) s_grid[e & 0xFFFFu] = hv[u];
      }
  };
  // COMPACT: 64 assembled rows ([word][lane], kRowBufPitch) -> the table, whole 64-B segments, 4 lanes per row; plus their
  // masks.  Deferred for the last 64 voxels of a block: the wave waits for ALL its outstanding memory operations before it
  // stages the next block (vmcnt counts stores too), so rows stored at the end of the probe exposed a full write round trip
  // per block; stored after the next block's staging they - and the gathers issued behind them - have the whole probe of
  // that block to complete.
  int* s_rows = s_grid + g.cells + 4 + kCells / 2 + 8;
  int pend_row = -1;
  auto flush_rows = [&](int row) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (row >= 0) mask[row] = (uint32_t)s_rows[lane];  // the dense mask array (tally, sort); also clears the "unwritten" mark
    const int piece = lane & 3;
    int rid[4];
    int4 v[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) rid[q] = __shfl(row, (lane >> 2) + 16 * q);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int r = (lane >> 2) + 16 * q;
      v[q].x = s_rows[(piece * 4 + 0) * kRowBufPitch + r];
      v[q].y = s_rows[(piece * 4 + 1) * kRowBufPitch + r];
      v[q].z = s_rows[(piece * 4 + 2) * kRowBufPitch + r];
      v[q].w = s_rows[(piece * 4 + 3) * kRowBufPitch + r];
    }
    // a 16-B piece goes out only if the row reaches it (words 4p .. 4p+3 hold ids 4p-1 .. 4p+2): 5 - 6 neighbours a row on the
    // scenes of the bench, so half of the pieces stay home
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if (rid[q] >= 0 && __popc((uint32_t)s_rows[(lane >> 2) + 16 * q]) >= 4 * piece) {
        if (FAST)
          *reinterpret_cast<int4*>(reinterpret_cast<char*>(nbr) + (__umul24((uint32_t)rid[q], kCompactPitch * 4u) + (uint32_t)piece * 16u)) = v[q];
        else
          *reinterpret_cast<int4*>(nbr + (int64_t)rid[q] * kCompactPitch + piece * 4) = v[q];
      }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");  // the rows are reused by the next 64 voxels
    __builtin_amdgcn_wave_barrier();
  };
#ifdef WCN_PROF
  if (lane == 0 && gwave < 4096) g_bprof[gwave * 12 + 9] = wall_clock64();
#endif
  if (seq < nblocks) {
    if (halo_regs) gather8(nb_c);
    else gather(nb_c, 0);
  }
#ifdef WCN_PROF
  unsigned long long prof_prev = wall_clock64(), prof_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, prof_blocks = 0;
  const unsigned long long prof_loop0 = prof_prev;
#endif
  for (; seq < nblocks; seq += nwaves) {
    BSTAMP(0);
#ifdef WCN_PROF
    ++prof_blocks;
#endif
    // ---- halo cells into the grid (absent neighbour: empty) ----
    if (halo_regs) scatter8();
    else scatter_halo(0);
    for (int r0 = 8; r0 < rounds; r0 += 8) {  // kernels with a halo above 1: the rest of the list, not pipelined
      gather(nb_c, r0);
      scatter_halo(r0);
    }
    BSTAMP(1);
    // ---- own cells into the grid; occupied ones enumerated with a wave prefix sum ----
    const int vals[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
    const int cell0 = ((lane >> 3) + g.hx) * g.px + ((lane & 7) + g.hy) * g.py + g.hz;
    int mine = 0;
#pragma unroll
    for (int z = 0; z < 8; ++z) {
      s_grid[cell0 + z] = vals[z];
      mine += vals[z] >= 0;
    }
    int incl = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int up = __shfl_up(incl, d);
      if (lane >= d) incl += up;
    }
    const int own_cnt = __shfl(incl, 63);
    int at = incl - mine;
#pragma unroll
    for (int z = 0; z < 8; ++z)
      if (vals[z] >= 0) s_own[at++] = (unsigned short)((cell0 + z) * 4);
    if (lane < 2 * kVoxPerIter) s_own[own_cnt + lane] = (unsigned short)(null_cell * 4);
    BSTAMP(4);
    if (COMPACT) {  // the previous block's last rows
      flush_rows(pend_row);
      pend_row = -1;
    }
    BSTAMP(5);
    // ---- advance the pipeline: gather for block i+1, head loads for block i+2 ----
    nb_c = nb_n; c0 = n0; c1 = n1;
    if (seq + nwaves < nblocks) {
      if (halo_regs) gather8(nb_c);
      else gather(nb_c, 0);
    }
    BSTAMP(6);
    load_head(seq + 2 * nwaves, nb_n, n0, n1);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    BSTAMP(2);
    const char* grid_bytes = reinterpret_cast<const char*>(s_grid);
    if (COMPACT) {
      // ---- compact rows: one lane per VOXEL, the K probes in a wave-uniform loop ----
      // A lane appends the neighbours it finds to its row in LDS ([word][lane], word pitch 66: the appends of one k are
      // conflict-free, the transposed read below is 2-way); then the wave writes the 64 rows out as whole 64-B segments,
      // 4 lanes per row.  Per block: K LDS probes + 4 stores per 64 voxels, where one lane per (voxel, offset) took
      // 16 trips of ~100 instructions (in-kernel stamps: 6.45 of the 8.3 us a block took were that loop).
      // lane k holds the grid byte offset of kernel offset k
      int delta_lane = 0;
      if (lane < K) {
        const int l = lane % g.kz, j = (lane / g.kz) % g.ky, i = lane / (g.kz * g.ky);
        delta_lane = ((i - g.cx) * g.dx * g.px + (j - g.cy) * g.dy * g.py + (l - g.cz) * g.dz) * 4;
      }
      bool over = false;
      char* rows_bytes = reinterpret_cast<char*>(s_rows);
      const int origin4 = (g.hx * g.px + g.hy * g.py + g.hz) * 4;  // an interior cell: every probe from it stays in the grid
      const uint32_t wp0 = (uint32_t)(kRowBufPitch + lane) * 4u, wp_max = (uint32_t)(kCompactPitch * kRowBufPitch + lane) * 4u;
      for (int e0 = 0; e0 < own_cnt; e0 += 64) {
        const bool live = e0 + lane < own_cnt;
        // (a lane past the block's last voxel probes from the origin cell into its own column of the rows and stores nothing)
        const int cell = live ? (int)s_own[e0 + lane] : origin4;
        const int row = live ? *reinterpret_cast<const int*>(grid_bytes + cell) : -1;
        // Branch-free append: the answer is ALWAYS written at the lane's write pointer, the pointer advances only past a
        // neighbour (the word behind the last one holds a don't-care: a spare 17th word takes it when the row is full).  The
        // pointer stops there: a row with more neighbours than fit is caught by the popcount below.  The mask is collected bit-reversed (one shift-or a probe).
        uint32_t wp = wp0, rev = 0;
        auto answer = [&](int f) {
          *reinterpret_cast<int*>(rows_bytes + wp) = f;
          const uint32_t h = f >= 0 ? 1u : 0u;
          rev = (rev << 1) | h;
          wp = min(wp + h * (uint32_t)(kRowBufPitch * 4), wp_max);
        };
        int k = 0;
        for (; k + 4 <= K; k += 4) {  // four probes in flight
          int f[4];
#pragma unroll
          for (int u = 0; u < 4; ++u)
            f[u] = *reinterpret_cast<const int*>(grid_bytes + cell + __builtin_amdgcn_readlane(delta_lane, k + u));
#pragma unroll
          for (int u = 0; u < 4; ++u) answer(f[u]);
        }
        for (; k < K; ++k) answer(*reinterpret_cast<const int*>(grid_bytes + cell + __builtin_amdgcn_readlane(delta_lane, k)));
        const uint32_t m = __brev(rev) >> (32 - K);
        over = over || (live && __popc(m) > kCompactIds);
        s_rows[lane] = (int)m;
        // every 64 rows but the block's last go out now; the last wait in LDS until the next block is staged (see flush_rows)
        if (e0 + 64 < own_cnt) flush_rows(row);
        else pend_row = row;
      }
      if (__any(over) && lane == 0) atomicOr(status, (int)WCN_FLAG_ROW_OVERFLOW);
    } else
    // ---- answer the K probes of the block's voxels: one lane per (voxel, offset) ----
    for (int kc = 0; kc < num_chunks; ++kc) {
      const int k = kc * LPR + sub;
      const bool k_real = k < K, k_store = k < kp;
      const int l = k % g.kz, j = (k / g.kz) % g.ky, i = k / (g.kz * g.ky);
      const int ox = (i - g.cx) * g.dx, oy = (j - g.cy) * g.dy, oz = (l - g.cz) * g.dz;
      // grid byte offset of this lane's kernel offset; lanes beyond K read the voxel's own cell and ignore it
      const int delta4 = k_real ? (ox * g.px + oy * g.py + oz) * 4 : 0;
      const int w0 = (kc * LPR) >> 5;
      // two voxel groups per trip: two independent LDS chains in flight
      for (int e0 = 0; e0 < own_cnt; e0 += 2 * kVoxPerIter) {
        const int cell_a = s_own[e0 + vsel], cell_b = s_own[e0 + kVoxPerIter + vsel];
        const int row_a = *reinterpret_cast<const int*>(grid_bytes + cell_a);
        const int row_b = *reinterpret_cast<const int*>(grid_bytes + cell_b);
        int found_a = *reinterpret_cast<const int*>(grid_bytes + cell_a + delta4);
        int found_b = *reinterpret_cast<const int*>(grid_bytes + cell_b + delta4);
        if (!k_real) { found_a = -1; found_b = -1; }
        const unsigned long long ball_a = __ballot(found_a >= 0), ball_b = __ballot(found_b >= 0);
        uint32_t bits_a, bits_b, hi_a = 0, hi_b = 0;
        if (LPR == 64) {
          bits_a = (uint32_t)ball_a; hi_a = (uint32_t)(ball_a >> 32);
          bits_b = (uint32_t)ball_b; hi_b = (uint32_t)(ball_b >> 32);
        } else if (LPR == 32) {
          bits_a = vsel ? (uint32_t)(ball_a >> 32) : (uint32_t)ball_a;
          bits_b = vsel ? (uint32_t)(ball_b >> 32) : (uint32_t)ball_b;
        } else {
          bits_a = (uint32_t)(ball_a >> (vsel * LPR)) & ((1u << LPR) - 1u);
          bits_b = (uint32_t)(ball_b >> (vsel * LPR)) & ((1u << LPR) - 1u);
        }
        if (FAST) {
          char* nbr_b = reinterpret_cast<char*>(nbr);
          char* mask_b = reinterpret_cast<char*>(mask);
          const uint32_t kp4 = (uint32_t)kp * 4u, mw4 = (uint32_t)mw * 4u, k4 = (uint32_t)k * 4u, w4 = (uint32_t)w0 * 4u;
          if (row_a >= 0 && k_store) *reinterpret_cast<int*>(nbr_b + (__umul24((uint32_t)row_a, kp4) + k4)) = found_a;
          if (row_b >= 0 && k_store) *reinterpret_cast<int*>(nbr_b + (__umul24((uint32_t)row_b, kp4) + k4)) = found_b;
          if (sub == 0 && w0 < mw) {  // also clears the "unwritten" mark
            if (row_a >= 0) *reinterpret_cast<uint32_t*>(mask_b + (__umul24((uint32_t)row_a, mw4) + w4)) = bits_a;
            if (row_b >= 0) *reinterpret_cast<uint32_t*>(mask_b + (__umul24((uint32_t)row_b, mw4) + w4)) = bits_b;
            if (LPR == 64 && w0 + 1 < mw) {
              if (row_a >= 0) *reinterpret_cast<uint32_t*>(mask_b + (__umul24((uint32_t)row_a, mw4) + w4 + 4u)) = hi_a;
              if (row_b >= 0) *reinterpret_cast<uint32_t*>(mask_b + (__umul24((uint32_t)row_b, mw4) + w4 + 4u)) = hi_b;
            }
          }
        } else {
          if (row_a >= 0 && k_store) nbr[(int64_t)row_a * kp + k] = found_a;
          if (row_b >= 0 && k_store) nbr[(int64_t)row_b * kp + k] = found_b;
          if (sub == 0 && w0 < mw) {
            if (row_a >= 0) mask[(int64_t)row_a * mw + w0] = bits_a;
            if (row_b >= 0) mask[(int64_t)row_b * mw + w0] = bits_b;
            if (LPR == 64 && w0 + 1 < mw) {
              if (row_a >= 0) mask[(int64_t)row_a * mw + w0 + 1] = hi_a;
              if (row_b >= 0) mask[(int64_t)row_b * mw + w0 + 1] = hi_b;
            }
          }
        }
      }
    }
    BSTAMP(3);

    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");  // the grid is rewritten for the next block
    __builtin_amdgcn_wave_barrier();
  }
  if (COMPACT) flush_rows(pend_row);
#ifdef WCN_PROF
  if (lane == 0 && gwave < 4096) {
    for (int i = 0; i < 8; ++i) g_bprof[gwave * 12 + i] = prof_acc[i];
    g_bprof[gwave * 12 + 8] = prof_blocks;
    g_bprof[gwave * 12 + 10] = prof_loop0;
    g_bprof[gwave * 12 + 11] = wall_clock64();
  }
#endif
}

// prepare -> sampled insert -> insert -> finish: the cell table of `coords` (and the halo list / neighbour-block ids of geometry g)
static void launch_cell_table(const CellTable& t, const CellGeom& g, const int4* coords, int64_t n, int kp, int mw, int32_t* nbr,
                              uint32_t* mask, int32_t* status, int strict, hipStream_t s) {
  const int64_t capacity = t.capacity;
  const uint32_t cmask = (uint32_t)(capacity - 1);
  const int64_t max_blocks = t.max_blocks;
  const int64_t prep = capacity > g.halo_cells ? capacity : g.halo_cells;
  hipLaunchKernelGGL(cell_prepare_kernel, dim3((unsigned)ceil_div(prep, 256)), dim3(256), 0, s, (uint4*)t.slots, capacity,
                     t.ctr, t.halo, g, status);
  const int64_t n_first = ceil_div(n, kInsertSample);
  hipLaunchKernelGGL(cell_insert_kernel<0>, dim3((unsigned)ceil_div(n_first, kInsertThreads)), dim3(kInsertThreads), 0, s, t.slots, cmask,
                     coords, n, t, status, kp, mw, nbr, mask, strict);
  hipLaunchKernelGGL(cell_insert_kernel<1>, dim3((unsigned)ceil_div(n, kInsertThreads)), dim3(kInsertThreads), 0, s, t.slots, cmask,
                     coords, n, t, status, kp, mw, nbr, mask, strict);
  {
    const int64_t work = n > 64 * max_blocks ? n : 64 * max_blocks;  // voxels vs one wave per block
    int64_t wgs = ceil_div(work, 256);
    if (wgs > 8192) wgs = 8192;
    hipLaunchKernelGGL(cell_finish_kernel, dim3((unsigned)wgs), dim3(256), 0, s, (const BSlot*)t.slots, cmask,
                       coords, n, t, g, mw, (const uint32_t*)mask, strict);
  }
}

static inline int lanes_per_row_b(int kp) {
  int l = 8;
  while (l < kp && l < 64) l <<= 1;
  return l;
}

}  // namespace wcn

using namespace wcn;

extern "C" {

size_t wcn_kmap_binned_workspace(int64_t n, int64_t max_blocks) {
  if (n < 0) n = 0;
  if (max_blocks < 1) max_blocks = 1;
  return carve_cells(nullptr, n, max_blocks).bytes;
}

int wcn_kmap_binned_supported(const int32_t ksize[3], const int32_t dilation[3]) {
  if (!ksize || !dilation) return 0;
  for (int d = 0; d < 3; ++d) {
    if (ksize[d] < 1 || dilation[d] < 1) return 0;
    const int c = (ksize[d] & 1) ? ksize[d] / 2 : 0;
    const int lo = c * dilation[d], hi = (ksize[d] - 1 - c) * dilation[d];
    if ((lo > hi ? lo : hi) > kMaxHalo) return 0;
  }
  const int64_t K = (int64_t)ksize[0] * ksize[1] * ksize[2];
  // the top bit of the last mask word marks rows no block has written yet: it must not be a real offset
  return (K <= 4096 && (K & 31) != 0) ? 1 : 0;
}

int wcn_kmap_compact_supported(int32_t num_offsets) {
  // one mask word and 32 lanes per voxel in cell_neighbors (row pitches 24 and 32)
  return (num_offsets >= 17 && num_offsets <= 31) ? 1 : 0;
}

int wcn_kmap_build_binned(const int32_t* coords, int64_t n, const int32_t ksize[3], const int32_t dilation[3],
                          int64_t max_blocks, int32_t strict, int32_t compact, void* workspace, size_t workspace_bytes,
                          int32_t* nbr, uint32_t* mask, int32_t* status, wcn_stream_t stream) {
  if (n < 0 || !status || max_blocks < 1 || max_blocks > (1ll << 29)) return WCN_ERROR_INVALID_PARAMETERS;
  if (!wcn_kmap_binned_supported(ksize, dilation)) return WCN_ERROR_PROBLEM_NOT_SUPPORTED;
  if (compact && (!ksize || !wcn_kmap_compact_supported(ksize[0] * ksize[1] * ksize[2]))) return WCN_ERROR_PROBLEM_NOT_SUPPORTED;
  if (n == 0) return WCN_SUCCESS;
  if (n >= (1ll << 31) || !coords || !nbr || !mask || !workspace ||
      workspace_bytes < wcn_kmap_binned_workspace(n, max_blocks))
    return WCN_ERROR_INVALID_PARAMETERS;
  hipStream_t s = (hipStream_t)stream;
  const CellTable t = carve_cells(workspace, n, max_blocks);
  const int K = ksize[0] * ksize[1] * ksize[2];
  const int kp = wcn_kmap_row_pitch(K), mw = wcn_kmap_mask_words(K);
  const CellGeom g = make_cell_geom(ksize, dilation);
  const int64_t capacity = t.capacity;
  const uint32_t cmask = (uint32_t)(capacity - 1);

  launch_cell_table(t, g, (const int4*)coords, n, compact ? (kCompactPitch | kCompactFlag) : kp, mw, nbr, mask, status, (int)strict, s);
  const int halo_rounds = (g.halo_cells + 63) >> 6;
  const int halo_pad = compact && halo_rounds <= 8 ? 0 : halo_rounds * 64;  // (compact rows: a list of up to 8 rounds lives in registers)
  // waves per workgroup: 4, fewer when halo list + one LDS grid per wave would not fit (halo 6..8: 20^3..24^3 cells)
  int nb_waves = kNbThreads / 64;
  auto shm_for = [&](int waves) {
    return ((size_t)halo_pad + (size_t)waves * (g.cells + 4 + kCells / 2 + 8 + (compact ? kRowBufInts : 0))) * 4;
  };
  while (nb_waves > 1 && shm_for(nb_waves) > 156 * 1024) nb_waves >>= 1;
  const size_t shm = shm_for(nb_waves);
  if (shm > 160 * 1024) return WCN_ERROR_PROBLEM_NOT_SUPPORTED;
  // resident waves only (the loop strides over the blocks): LDS allows 160 KB / shm workgroups per CU
  int per_cu = (int)((160 * 1024) / (shm + 512));
  if (per_cu > 8) per_cu = 8;
  if (compact && per_cu * nb_waves > 4 * kNbWavesPerSimd) per_cu = 4 * kNbWavesPerSimd / nb_waves;  // (amdgpu_waves_per_eu)
  if (per_cu < 1) per_cu = 1;
  int64_t want = ceil_div(max_blocks < n ? max_blocks : n, nb_waves);  // never more waves than blocks
  if (want > 256 * per_cu) want = 256 * per_cu;
  const dim3 grid((unsigned)want), block(nb_waves * 64);
  if (shm > 64 * 1024) {  // above the default dynamic-LDS limit: raise it once per device for every instance
    static unsigned long long attr_done = 0ull;
    const int rc = once_per_device(attr_done, [] {
      bool ok = true;
      for (const void* f : {reinterpret_cast<const void*>(cell_neighbors_kernel<8, true, false>), reinterpret_cast<const void*>(cell_neighbors_kernel<8, false, false>),
                            reinterpret_cast<const void*>(cell_neighbors_kernel<16, true, false>), reinterpret_cast<const void*>(cell_neighbors_kernel<16, false, false>),
                            reinterpret_cast<const void*>(cell_neighbors_kernel<32, true, false>), reinterpret_cast<const void*>(cell_neighbors_kernel<32, false, false>),
                            reinterpret_cast<const void*>(cell_neighbors_kernel<32, true, true>), reinterpret_cast<const void*>(cell_neighbors_kernel<32, false, true>),
                            reinterpret_cast<const void*>(cell_neighbors_kernel<64, true, false>), reinterpret_cast<const void*>(cell_neighbors_kernel<64, false, false>)})
        ok = ok && hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess;
      return ok;
    });
    if (rc != WCN_SUCCESS) return rc;
  }
  const bool fast = n < (1ll << 24) && n * kp * 4 < (1ll << 31);
#define WCN_CELL_NB(L, C)                                                                                                 \
  do {                                                                                                                    \
    if (fast)                                                                                                             \
      hipLaunchKernelGGL((cell_neighbors_kernel<L, true, C>), grid, block, shm, s, t, (const uint32_t*)t.halo, g, K, kp,  \
                         mw, nbr, mask, status);                                                                          \
    else                                                                                                                  \
      hipLaunchKernelGGL((cell_neighbors_kernel<L, false, C>), grid, block, shm, s, t, (const uint32_t*)t.halo, g, K, kp, \
                         mw, nbr, mask, status);                                                                          \
  } while (0)
  if (compact) {
    WCN_CELL_NB(32, true);  // (wcn_kmap_compact_supported: 32 lanes per voxel, one mask word)
  } else {
    switch (lanes_per_row_b(kp)) {
      case 8: WCN_CELL_NB(8, false); break;
      case 16: WCN_CELL_NB(16, false); break;
      case 32: WCN_CELL_NB(32, false); break;
      default: WCN_CELL_NB(64, false); break;
    }
  }
#undef WCN_CELL_NB
  return launch_status();
}

int wcn_kmap_cells_build(const int32_t* coords, int64_t n, int64_t max_blocks, void* workspace, size_t workspace_bytes,
                         uint32_t* scratch, int32_t* status, wcn_stream_t stream) {
  if (n < 0 || !status || max_blocks < 1 || max_blocks > (1ll << 29)) return WCN_ERROR_INVALID_PARAMETERS;
  if (n == 0) return WCN_SUCCESS;
  if (n >= (1ll << 31) || !coords || !scratch || !workspace || workspace_bytes < wcn_kmap_binned_workspace(n, max_blocks))
    return WCN_ERROR_INVALID_PARAMETERS;
  const CellTable t = carve_cells(workspace, n, max_blocks);
  const int32_t one[3] = {1, 1, 1};
  const CellGeom g = make_cell_geom(one, one);  // no halo: the table alone
  // strict = 1: a duplicated coordinate keeps its smallest row by construction (nothing checks afterwards)
  launch_cell_table(t, g, (const int4*)coords, n, 0, 1, nullptr, scratch, status, 1, (hipStream_t)stream);
  return launch_status();
}

}  // extern "C"

#ifdef WCN_PROF
extern "C" int wcn_debug_read_bprof(void* dst, size_t bytes) {
  return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(wcn::g_bprof), bytes);
}
#endif
