// kmap_binned.hip - LDS-binned neighbour search for submanifold kernel maps (output coords == input coords).
//
// A global hash probe per (voxel, offset) is 27 random 16-B reads per voxel (27 M for a 1 M-voxel scene): HBM /
// fabric latency bound.  Here voxels are first BINNED into 16x16x16 blocks (a small block-level hash table
// assigns dense block ids; counting sort by block), then one workgroup per block stages the block and its
// one-cell... H-cell halo (taken from the 26 neighbouring bins) into a dense LDS grid and answers all
// K probes of its voxels from LDS.  Global traffic becomes streaming bin reads + full-line row writes.
//
//   pass 1a bin_insert   voxel -> block slot (CAS on the block key only on first touch), dense block ids
//   pass 1b bin_count    position inside the bin.  Voxels within H cells of a block face - the only ones a NEIGHBOUR
//                        block can need - go first, grouped by the first face they are near (6 groups); interior
//                        voxels use 8 row-keyed sub-counters (same-address atomics serialise)
//   pass 2  bin_assign   bin size per block, exclusive prefix over the block's groups
//   pass 3  bin_scan     exclusive scan of bin sizes
//   pass 4  bin_scatter  voxels -> binned array {x, y, z, row}
//   pass 5  bin_neighbors  per block: LDS grid of (16+2H)^3 row ids (atomicMin => duplicates keep the smallest
//                          row, same rule as the hash path), then one LANE per (voxel, offset): the neighbour row
//                          is written as one contiguous line and the mask is a wave ballot.
//
// Semantics equal wcn_hash_insert + wcn_kmap_probe with stride 1 (incl. 18-bit coordinate wrap of the packed
// key: neighbour blocks are looked up with wrapped block coordinates, positions are block-relative).
// Reference behaviour replaced: warpconvnet/csrc/cuhash_hash_table.cu:179-220, cuhash_kernel_map.cu:93-134.
#include "wcn_common.h"

namespace wcn {

constexpr int kBlkShift = 4;
constexpr int kBlk = 1 << kBlkShift;  // 16 cells per axis
constexpr int kMaxHalo = 4;
constexpr int kBinThreads = 256;
constexpr int kOwnChunk = 512;  // own-bin entries staged in LDS per pass
constexpr int kBlkCoordBits = kCoordBits - kBlkShift;  // 14-bit signed block coordinates

struct BinGeom {
  int kx, ky, kz, cx, cy, cz, dx, dy, dz;
  int hx, hy, hz;  // halo per axis (max |offset|)
  int gx, gy, gz;  // LDS grid extent per axis
};

__device__ __forceinline__ int wrap_blk(int v) {  // wrap to the signed range of the block coordinate field
  const int bits = kBlkCoordBits;
  v &= (1 << bits) - 1;
  return (v ^ (1 << (bits - 1))) - (1 << (bits - 1));
}

__device__ __forceinline__ uint64_t block_key(int b, int bx, int by, int bz) { return pack_key(b, bx, by, bz); }

// slot of an existing block key, or -1
__device__ __forceinline__ int block_find(const Slot* __restrict__ slots, uint32_t cmask, uint64_t key) {
  uint32_t s = hash_slot(key, cmask);
  for (uint32_t a = 0; a <= cmask; ++a) {
    const uint64_t k = slots[s].key;
    if (k == 0ull) return -1;
    if (k == key) return (int)s;
    s = (s + 1) & cmask;
  }
  return -1;
}

// Slot use on this path: key = block key, value / pad unused (zeroed).  Bin sizes live in cnt[id][2 * kSub]:
// groups 0..5 = BOUNDARY voxels by first near face (x-, x+, y-, y+, z-, z+; 6, 7 unused), groups kSub.. = INTERIOR
// voxels spread over kSub sub-counters chosen by the row index.  Same-address atomics serialise (~280 ns each,
// ~455 voxels per 16^3 block on the uniform scene: ~25 per face group, ~38 per interior sub-counter).
constexpr int kSub = 8;
constexpr int kInsertSample = 16;  // bin_insert: 1 voxel in 16 goes first (see the kernel)

// clears the block table and - in the same launch - the block counter and the caller's status word (every extra
// memset / fill is a ~5 us launch on this pipeline of ~20 short kernels)
__global__ void bin_prepare_kernel(uint4* __restrict__ slots, int64_t capacity, int32_t* __restrict__ nblk,
                                   int32_t* __restrict__ status) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < capacity) slots[i] = make_uint4(0u, 0u, 0u, 0u);
  if (i < 64) nblk[i] = 0;
  if (i == 0) *status = 0;
}

// pass 1a: create the block entries (CAS only on first touch; everybody else just reads) and remember the slot
__global__ void bin_insert_kernel(Slot* __restrict__ slots, uint32_t cmask, const int4* __restrict__ coords, int64_t n,
                                  int32_t* __restrict__ vox_slot, int32_t* __restrict__ blk_slot,
                                  int32_t* __restrict__ slot_id, int32_t* __restrict__ nblk, int32_t* __restrict__ cnt,
                                  int32_t* __restrict__ status, int kp, int mw, int32_t* __restrict__ nbr,
                                  uint32_t* __restrict__ mask, int phase) {
  // Two launches: phase 0 inserts every kInsertSample-th voxel, phase 1 the rest.  With ONE launch half a million
  // resident threads meet an empty table at the same instant and all of them CAS the ~2 200 block keys (~240 same-address
  // atomics per key, serialised: 36 us); after the sampled pass nearly every block exists, and the rest of the voxels
  // only read (a block that the sample missed is simply created in phase 1).
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t i = phase == 0 ? t * kInsertSample : t + t / (kInsertSample - 1) + 1;
  if (i >= n) return;
  const int4 c = coords[i];
  if (!coord_in_range(c.x, c.y, c.z, c.w)) {
    atomicOr(status, (int)WCN_FLAG_COORD_RANGE);
    vox_slot[i] = -1;
    // the voxel is in no bin, so bin_neighbors never visits it: give its table row defined ("no neighbour") content -
    // consumers may already be queued behind this build when the host sees the flag
    for (int k = 0; k < kp; ++k) nbr[i * kp + k] = -1;
    for (int w = 0; w < mw; ++w) mask[i * mw + w] = 0u;
    return;
  }
  const uint64_t key = block_key(c.x, c.y >> kBlkShift, c.z >> kBlkShift, c.w >> kBlkShift);
  uint32_t s = hash_slot(key, cmask);
  int found = -1;
  bool created = false;
  for (uint32_t a = 0; a <= cmask; ++a) {
    unsigned long long* kp = reinterpret_cast<unsigned long long*>(&slots[s].key);
    // optimistic cached read first: a key, once written, never changes, so a matching value is final; an empty or
    // stale value falls through to the coherent read below
    // (wavefront-scope relaxed load = an ordinary cached load the compiler must still perform; a `volatile` access is
    // emitted as a system-coherent `sc0 sc1` load that misses every cache - 36 us instead of 10 for this kernel)
    unsigned long long cur = __hip_atomic_load(kp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    if (cur == key) { found = (int)s; break; }
    cur = __hip_atomic_load(kp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (cur == 0ull) {
      cur = atomicCAS(kp, 0ull, (unsigned long long)key);
      if (cur == 0ull) {  // this thread created the block (its dense id is handed out below, once per wave)
        created = true;
        found = (int)s;
        break;
      }
    }
    if (cur == key) { found = (int)s; break; }
    s = (s + 1) & cmask;
  }
  // dense block ids: ONE counter update per wave for all the blocks its lanes created (every creator doing its own
  // atomicAdd on the same address serialises ~2 200 round trips on the uniform scene - most of this kernel's time)
  const unsigned long long makers = __ballot(created);
  if (makers != 0ull) {
    const int lane = threadIdx.x & 63;
    const int leader = __ffsll((long long)makers) - 1;
    int base = 0;
    if (lane == leader) base = atomicAdd(nblk, __popcll(makers));
    base = __shfl(base, leader);
    if (created) {
      const int id = base + __popcll(makers & ((1ull << lane) - 1ull));
      blk_slot[id] = found;
      slot_id[found] = id;  // read by the NEXT kernels only
      // the block's position counters, used by bin_count (next kernel): cleared here by the one thread that created
      // the block instead of a worst-case 32 MB memset (the block count is only known on the device)
#pragma unroll
      for (int q = 0; q < (2 * kSub) / 4; ++q)
        reinterpret_cast<int4*>(cnt + (int64_t)id * (2 * kSub))[q] = make_int4(0, 0, 0, 0);
    }
  }
  if (found < 0) atomicOr(status, (int)WCN_FLAG_TABLE_FULL);
  vox_slot[i] = found;
}

// pass 1b: position of every voxel inside its (block, class, sub-counter) group.
// vox_pos = (class * kSub + sub) << 24 | position  (a 16^3 block holds at most 4096 distinct voxels, duplicates are
// bounded by n < 2^24 per sub-counter in practice; larger counts set the overflow flag)
__global__ void bin_count_kernel(const int4* __restrict__ coords, int64_t n, int hx, int hy, int hz,
                                 const int32_t* __restrict__ vox_slot, const int32_t* __restrict__ slot_id,
                                 int32_t* __restrict__ cnt, int32_t* __restrict__ vox_pos,
                                 int32_t* __restrict__ status) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int s = vox_slot[i];
  if (s < 0) return;
  const int4 c = coords[i];
  const int lx = c.y & (kBlk - 1), ly = c.z & (kBlk - 1), lz = c.w & (kBlk - 1);
  // boundary voxels are grouped by the FIRST face they are near (x-, x+, y-, y+, z-, z+): a neighbour block then reads
  // only the groups that can hold voxels near the face it shares with this block (see bin_neighbors) instead of the
  // whole boundary shell - 3x fewer halo candidates on the uniform scene.  Interior voxels keep kSub row-keyed
  // sub-counters (same-address atomics serialise).
  int face = -1;
  if (lx < hx) face = 0;
  else if (lx >= kBlk - hx) face = 1;
  else if (ly < hy) face = 2;
  else if (ly >= kBlk - hy) face = 3;
  else if (lz < hz) face = 4;
  else if (lz >= kBlk - hz) face = 5;
  const int group = face >= 0 ? face : kSub + (int)(i & (kSub - 1));
  const int pos = atomicAdd(&cnt[(int64_t)slot_id[s] * (2 * kSub) + group], 1);
  if (pos >= (1 << 24)) atomicOr(status, (int)WCN_FLAG_TABLE_FULL);
  vox_pos[i] = (group << 24) | (pos & 0xFFFFFF);
}

// pass 2: bin size per block, group counters -> exclusive prefix
__global__ void bin_assign_kernel(int32_t* __restrict__ cnt, const int32_t* __restrict__ nblk, int64_t max_blocks,
                                  int32_t* __restrict__ blk_cnt) {
  const int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (id >= max_blocks || id >= *nblk) return;
  int32_t* c = cnt + id * (2 * kSub);
  int tot = 0;
#pragma unroll
  for (int g = 0; g < 2 * kSub; ++g) {
    const int v = c[g];
    c[g] = tot;  // exclusive prefix over the groups: bin_scatter adds it to the position inside the group, bin_neighbors
    tot += v;    // reads the run of face groups a neighbour needs
  }
  blk_cnt[id] = tot;
}

// single workgroup: exclusive scan of blk_cnt[0..nblk) -> blk_off[0..nblk]
__global__ __launch_bounds__(1024) void bin_scan_kernel(const int32_t* __restrict__ blk_cnt,
                                                        const int32_t* __restrict__ nblk,
                                                        int32_t* __restrict__ blk_off) {
  __shared__ int s_part[16];
  const int n = *nblk;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int chunk = (n + 1023) / 1024;
  const int b0 = tid * chunk;
  const int b1 = (b0 + chunk < n) ? (b0 + chunk) : n;
  int sum = 0;
  for (int b = b0; b < b1; ++b) sum += blk_cnt[b];
  int incl = sum;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int t = __shfl_up(incl, d);
    if (lane >= d) incl += t;
  }
  if (lane == 63) s_part[wave] = incl;
  __syncthreads();
  int base = 0;
  for (int w = 0; w < wave; ++w) base += s_part[w];
  int run = base + incl - sum;
  for (int b = b0; b < b1; ++b) {
    blk_off[b] = run;
    run += blk_cnt[b];
  }
  if (tid == 1023) blk_off[n] = base + incl;
}

__global__ void bin_scatter_kernel(const int32_t* __restrict__ slot_id, const int32_t* __restrict__ cnt,
                                   const int4* __restrict__ coords, int64_t n, const int32_t* __restrict__ vox_slot,
                                   const int32_t* __restrict__ vox_pos, const int32_t* __restrict__ blk_off,
                                   int4* __restrict__ binned) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int s = vox_slot[i];
  if (s < 0) return;
  const int id = slot_id[s];
  const int4 c = coords[i];
  const int p = vox_pos[i];
  const int group = p >> 24;
  // groups are laid out in order: boundary sub-bins, then interior sub-bins; cnt holds their exclusive prefix
  const int local = (p & 0xFFFFFF) + cnt[(int64_t)id * (2 * kSub) + group];
  binned[blk_off[id] + local] = make_int4(c.y, c.z, c.w, (int)i);
}

#ifdef WCN_PROF
__device__ unsigned long long g_bprof[4096 * 8];
#define BSTAMP(i) do { if (threadIdx.x == 0 && blockIdx.x < 4096) g_bprof[blockIdx.x * 8 + (i)] = wall_clock64(); } while (0)
#else
#define BSTAMP(i)
#endif
template <int LPR>
__global__ __launch_bounds__(kBinThreads) void bin_neighbors_kernel(const Slot* __restrict__ slots, uint32_t cmask,
                                                                    const int32_t* __restrict__ slot_id,
                                                                    const int32_t* __restrict__ grp_pre,
                                                                    const int32_t* __restrict__ blk_slot,
                                                                    const int32_t* __restrict__ nblk,
                                                                    const int32_t* __restrict__ blk_off,
                                                                    const int4* __restrict__ binned, BinGeom g, int K,
                                                                    int kp, int mw, int32_t* __restrict__ nbr,
                                                                    uint32_t* __restrict__ mask,
                                                                    int32_t* __restrict__ status) {
  extern __shared__ unsigned int s_grid[];  // [gx*gy*gz] row ids, 0xFFFFFFFF = empty; then 27*3 ints of bin info
  const int cells = g.gx * g.gy * g.gz;
  int* s_nb_beg = reinterpret_cast<int*>(s_grid + cells);  // [27] first entry of neighbour bin
  int* s_nb_pre = s_nb_beg + 27;                            // [28] prefix of neighbour bin sizes
  int2* s_own = reinterpret_cast<int2*>(s_grid + ((cells + 64 + 3) & ~3));  // [kOwnChunk] staged own voxels: (grid cell, row)
  const int tid = threadIdx.x, lane = tid & 63;
  const int nblocks = *nblk;
  constexpr int kVoxPerIter = kBinThreads / LPR;
  const int sub = tid % LPR, vsel = tid / LPR;

  for (int id = blockIdx.x; id < nblocks; id += gridDim.x) {
    BSTAMP(0);
    const uint64_t key = slots[blk_slot[id]].key;
    const int b = (int)((key >> 54) & kBatchMask);
    const int bx = wrap_blk((int)((key >> 36) & kCoordMask));
    const int by = wrap_blk((int)((key >> 18) & kCoordMask));
    const int bz = wrap_blk((int)(key & kCoordMask));
    // ---- neighbour bins (27 lanes probe the block table) ----
    if (tid < 27) {
      const int ddx = tid / 9 - 1, ddy = (tid / 3) % 3 - 1, ddz = tid % 3 - 1;
      const bool needed = (ddx == 0 || g.hx > 0) && (ddy == 0 || g.hy > 0) && (ddz == 0 || g.hz > 0);
      int beg = 0, cnt = 0;
      if (needed) {
        const int s = block_find(slots, cmask, block_key(b, wrap_blk(bx + ddx), wrap_blk(by + ddy), wrap_blk(bz + ddz)));
        if (s >= 0) {
          const int nid = slot_id[s];
          beg = blk_off[nid];
          if (tid == 13) {
            cnt = blk_off[nid + 1] - beg;  // the own block is read whole
          } else {
            // a neighbour contributes voxels near the face it shares with this block.  Boundary voxels are stored first,
            // grouped by the first face they are near (bin_count), so the candidates are one contiguous run of groups:
            //   +x neighbour: its x- group; -x: x+; same x, +y: x-, x+, y-; same x, -y: x- .. y+; same x and y: .. z-/z+
            const int g0 = ddx > 0 ? 0 : (ddx < 0 ? 1 : 0);
            const int g1 = ddx > 0 ? 0 : (ddx < 0 ? 1 : (ddy > 0 ? 2 : (ddy < 0 ? 3 : (ddz > 0 ? 4 : 5))));
            const int32_t* pre = grp_pre + (int64_t)nid * (2 * kSub);  // exclusive prefix over the groups (bin_assign)
            const int p0 = pre[g0];
            beg += p0;
            cnt = pre[g1 + 1] - p0;
          }
        }
      }
      s_nb_beg[tid] = beg;
      s_nb_pre[tid + 1] = cnt;
    }
    for (int c = tid; c < cells; c += kBinThreads) s_grid[c] = 0xFFFFFFFFu;
    __syncthreads();
    BSTAMP(1);
    if (tid == 0) {
      s_nb_pre[0] = 0;
      for (int q = 0; q < 27; ++q) s_nb_pre[q + 1] += s_nb_pre[q];
    }
    __syncthreads();
    // ---- fill the grid from the 27 bins (4 independent loads in flight per thread) ----
    const int total = s_nb_pre[27];
    for (int e0 = tid; e0 < total; e0 += 4 * kBinThreads) {
      int4 v[4];
      int qq[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int e = e0 + u * kBinThreads;
        qq[u] = -1;
        if (e < total) {
          int q = 0;
#pragma unroll
          for (int t = 1; t < 27; ++t) q += (e >= s_nb_pre[t]);
          qq[u] = q;
          v[u] = binned[s_nb_beg[q] + (e - s_nb_pre[q])];
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int q = qq[u];
        if (q < 0) continue;
        const int ddx = q / 9 - 1, ddy = (q / 3) % 3 - 1, ddz = q % 3 - 1;
        const int lx = (v[u].x & (kBlk - 1)) + kBlk * ddx + g.hx;
        const int ly = (v[u].y & (kBlk - 1)) + kBlk * ddy + g.hy;
        const int lz = (v[u].z & (kBlk - 1)) + kBlk * ddz + g.hz;
        if (lx >= 0 && lx < g.gx && ly >= 0 && ly < g.gy && lz >= 0 && lz < g.gz)
          atomicMin(&s_grid[(lx * g.gy + ly) * g.gz + lz], (unsigned int)v[u].w);
      }
    }
    __syncthreads();
    BSTAMP(2);
    // ---- answer the K probes of the block's own voxels: one lane per (voxel, offset).  The block's own entries are
    //      staged through LDS in chunks (coalesced, upfront) so the probe loop itself has no global loads. ----
    const int own_beg = s_nb_beg[13], own_cnt = s_nb_pre[14] - s_nb_pre[13];
    const int num_chunks = (kp + LPR - 1) / LPR;
    for (int c0 = 0; c0 < own_cnt; c0 += kOwnChunk) {
      const int cn = (own_cnt - c0) < kOwnChunk ? (own_cnt - c0) : kOwnChunk;
      // staged as (grid cell of the voxel, row): the probe loop is instruction-issue bound, so everything that does
      // not depend on the offset is computed once here
      for (int e = tid; e < cn; e += kBinThreads) {
        const int4 v = binned[own_beg + c0 + e];
        const int cell = (((v.x & (kBlk - 1)) + g.hx) * g.gy + (v.y & (kBlk - 1)) + g.hy) * g.gz + (v.z & (kBlk - 1)) + g.hz;
        s_own[e] = make_int2(cell, v.w);
      }
      __syncthreads();
      for (int kc = 0; kc < num_chunks; ++kc) {
        const int k = kc * LPR + sub;
        const bool k_real = k < K, k_store = k < kp;
        const int l = k % g.kz, j = (k / g.kz) % g.ky, i = k / (g.kz * g.ky);
        const int ox = (i - g.cx) * g.dx, oy = (j - g.cy) * g.dy, oz = (l - g.cz) * g.dz;
        const int delta = (ox * g.gy + oy) * g.gz + oz;  // cell offset of this lane's kernel offset
        const bool centre = (ox | oy | oz) == 0;
        int32_t* nbr_k = nbr + k;
        for (int e0 = 0; e0 < cn; e0 += kVoxPerIter) {
          const int e = e0 + vsel;
          int found = -1;
          int row = -1;
          if (e < cn) {
            const int2 v = s_own[e];
            row = v.y;
            if (k_real) {
              found = (int)s_grid[v.x + delta];  // 0xFFFFFFFF -> -1
              if (centre && found != row) atomicOr(status, (int)WCN_FLAG_DUPLICATE_COORD);
            }
            if (k_store) nbr_k[(int64_t)row * kp] = found;
          }
          const unsigned long long ball = __ballot(found >= 0);
          if (row >= 0 && sub == 0) {
            const int lsel = lane / LPR;  // voxel slot inside this wavefront
            const unsigned long long bits = (LPR == 64) ? ball : ((ball >> (lsel * LPR)) & ((1ull << (LPR & 63)) - 1ull));
            const int w0 = (kc * LPR) >> 5;
            if (w0 < mw) mask[(int64_t)row * mw + w0] = (uint32_t)bits;
            if (LPR == 64 && w0 + 1 < mw) mask[(int64_t)row * mw + w0 + 1] = (uint32_t)(bits >> 32);
          }
        }
      }
      __syncthreads();  // s_own is refilled by the next chunk
    }
#ifdef WCN_PROF
    if (threadIdx.x == 0 && blockIdx.x < 4096) g_bprof[blockIdx.x * 8 + 4] = own_cnt;
#endif
    BSTAMP(3);
    __syncthreads();  // grid is reused by the next block
  }
}

static inline size_t align256b(size_t v) { return (v + 255) & ~(size_t)255; }

struct BinWorkspace {
  int32_t *vox_slot, *vox_pos, *blk_slot, *blk_cnt, *blk_off, *nblk, *slot_id, *cnt;
  int4* binned;
  size_t bytes;
};

static BinWorkspace carve(void* ws, int64_t n, int64_t capacity) {
  BinWorkspace w;
  char* p = (char*)ws;
  size_t off = 0;
  auto take = [&](size_t bytes) { char* q = p ? p + off : nullptr; off += align256b(bytes); return q; };
  w.nblk = (int32_t*)take(256);
  w.vox_slot = (int32_t*)take((size_t)n * 4);
  w.vox_pos = (int32_t*)take((size_t)n * 4);
  w.blk_slot = (int32_t*)take((size_t)n * 4);
  w.blk_cnt = (int32_t*)take((size_t)n * 4);
  w.cnt = (int32_t*)take((size_t)n * 2 * kSub * 4);
  w.blk_off = (int32_t*)take((size_t)(n + 1) * 4);
  w.binned = (int4*)take((size_t)n * 16);
  w.slot_id = (int32_t*)take((size_t)capacity * 4);
  w.bytes = off;
  return w;
}

static inline int lanes_per_row_b(int kp) {
  int l = 8;
  while (l < kp && l < 64) l <<= 1;
  return l;
}

}  // namespace wcn

using namespace wcn;

extern "C" {

static int64_t binned_capacity(int64_t n) {
  int64_t c = 16;
  while (c < 2 * n) c <<= 1;
  return c;
}

size_t wcn_kmap_binned_workspace(int64_t n) {
  if (n < 0) n = 0;
  return carve(nullptr, n, binned_capacity(n)).bytes;
}

int wcn_kmap_binned_supported(const int32_t ksize[3], const int32_t dilation[3]) {
  if (!ksize || !dilation) return 0;
  for (int d = 0; d < 3; ++d) {
    if (ksize[d] < 1 || dilation[d] < 1) return 0;
    const int c = (ksize[d] & 1) ? ksize[d] / 2 : 0;
    const int lo = c * dilation[d], hi = (ksize[d] - 1 - c) * dilation[d];
    if ((lo > hi ? lo : hi) > kMaxHalo) return 0;
  }
  return (int64_t)ksize[0] * ksize[1] * ksize[2] <= 4096 ? 1 : 0;
}

int wcn_kmap_build_binned(const int32_t* coords, int64_t n, const int32_t ksize[3], const int32_t dilation[3],
                          void* slots, int64_t capacity, void* workspace, size_t workspace_bytes, int32_t* nbr,
                          uint32_t* mask, int32_t* status, wcn_stream_t stream) {
  if (n < 0 || !status || !slots || capacity <= 0 || (capacity & (capacity - 1)) != 0 || capacity > (1ll << 31))
    return WCN_ERROR_INVALID_PARAMETERS;
  if (!wcn_kmap_binned_supported(ksize, dilation)) return WCN_ERROR_PROBLEM_NOT_SUPPORTED;
  if (n == 0) return WCN_SUCCESS;
  if (!coords || !nbr || !mask || !workspace || workspace_bytes < wcn_kmap_binned_workspace(n))
    return WCN_ERROR_INVALID_PARAMETERS;
  hipStream_t s = (hipStream_t)stream;
  if (capacity != binned_capacity(n)) return WCN_ERROR_INVALID_PARAMETERS;
  const BinWorkspace w = carve(workspace, n, capacity);
  const int K = ksize[0] * ksize[1] * ksize[2];
  const int kp = wcn_kmap_row_pitch(K), mw = wcn_kmap_mask_words(K);
  BinGeom g;
  g.kx = ksize[0]; g.ky = ksize[1]; g.kz = ksize[2];
  g.cx = (g.kx & 1) ? g.kx / 2 : 0; g.cy = (g.ky & 1) ? g.ky / 2 : 0; g.cz = (g.kz & 1) ? g.kz / 2 : 0;
  g.dx = dilation[0]; g.dy = dilation[1]; g.dz = dilation[2];
  auto halo = [](int ks, int c, int d) { const int lo = c * d, hi = (ks - 1 - c) * d; return lo > hi ? lo : hi; };
  g.hx = halo(g.kx, g.cx, g.dx); g.hy = halo(g.ky, g.cy, g.dy); g.hz = halo(g.kz, g.cz, g.dz);
  g.gx = kBlk + 2 * g.hx; g.gy = kBlk + 2 * g.hy; g.gz = kBlk + 2 * g.hz;

  hipLaunchKernelGGL(bin_prepare_kernel, dim3((unsigned)ceil_div(capacity, 256)), dim3(256), 0, s, (uint4*)slots, capacity,
                     w.nblk, status);
  const uint32_t cmask = (uint32_t)(capacity - 1);
  const unsigned gn = (unsigned)ceil_div(n, 256);
  const int64_t n_first = ceil_div(n, kInsertSample), n_rest = n - n_first;
  hipLaunchKernelGGL(bin_insert_kernel, dim3((unsigned)ceil_div(n_first, 256)), dim3(256), 0, s, (Slot*)slots, cmask,
                     (const int4*)coords, n, w.vox_slot, w.blk_slot, w.slot_id, w.nblk, w.cnt, status, kp, mw, nbr, mask, 0);
  if (n_rest > 0)
    hipLaunchKernelGGL(bin_insert_kernel, dim3((unsigned)ceil_div(n_rest, 256)), dim3(256), 0, s, (Slot*)slots, cmask,
                       (const int4*)coords, n, w.vox_slot, w.blk_slot, w.slot_id, w.nblk, w.cnt, status, kp, mw, nbr, mask,
                       1);
  hipLaunchKernelGGL(bin_count_kernel, dim3(gn), dim3(256), 0, s, (const int4*)coords, n, g.hx, g.hy, g.hz,
                     (const int32_t*)w.vox_slot, (const int32_t*)w.slot_id, w.cnt, w.vox_pos, status);
  hipLaunchKernelGGL(bin_assign_kernel, dim3(gn), dim3(256), 0, s, w.cnt, (const int32_t*)w.nblk, n,
                     w.blk_cnt);
  hipLaunchKernelGGL(bin_scan_kernel, dim3(1), dim3(1024), 0, s, (const int32_t*)w.blk_cnt, (const int32_t*)w.nblk,
                     w.blk_off);
  hipLaunchKernelGGL(bin_scatter_kernel, dim3(gn), dim3(256), 0, s, (const int32_t*)w.slot_id, (const int32_t*)w.cnt,
                     (const int4*)coords, n, (const int32_t*)w.vox_slot, (const int32_t*)w.vox_pos,
                     (const int32_t*)w.blk_off, w.binned);
  const size_t shm = (((size_t)g.gx * g.gy * g.gz + 64 + 3) & ~(size_t)3) * 4 + (size_t)kOwnChunk * 8;
  const int64_t want = n / 64 + 1;  // never more workgroups than could have work
  const dim3 grid((unsigned)(want < 4096 ? want : 4096)), block(kBinThreads);
#define WCN_BIN_NB(L)                                                                                                  \
  hipLaunchKernelGGL(bin_neighbors_kernel<L>, grid, block, shm, s, (const Slot*)slots, cmask,                            \
                     (const int32_t*)w.slot_id, (const int32_t*)w.cnt, (const int32_t*)w.blk_slot,                       \
                     (const int32_t*)w.nblk, (const int32_t*)w.blk_off, (const int4*)w.binned, g, K, kp, mw, nbr, mask,  \
                     status)
  switch (lanes_per_row_b(kp)) {
    case 8: WCN_BIN_NB(8); break;
    case 16: WCN_BIN_NB(16); break;
    case 32: WCN_BIN_NB(32); break;
    default: WCN_BIN_NB(64); break;
  }
#undef WCN_BIN_NB
  return launch_status();
}

}  // extern "C"

#ifdef WCN_PROF
extern "C" int wcn_debug_read_bprof(void* dst, size_t bytes) {
  return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(wcn::g_bprof), bytes);
}
#endif
