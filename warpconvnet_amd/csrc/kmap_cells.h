// kmap_cells.h - the block-hashed cell table shared by the binned builder (kmap_binned.hip) and the tally pass that
// repairs the rows of duplicate coordinates (kmap.hip).
#pragma once

#include "wcn_common.h"

namespace wcn {

constexpr int kBlkShift = 3;
constexpr int kBlk = 1 << kBlkShift;    // 8 cells per axis
constexpr int kCells = kBlk * kBlk * kBlk;  // 512 row ids = 2 KB per occupied block
constexpr int kMaxHalo = 8;  // one block width: the 26 adjacent blocks still cover every probe
constexpr int kBlkCoordBits = kCoordBits - kBlkShift;  // 15-bit signed block coordinates
constexpr uint32_t kMaskUnwritten = 0x80000000u;       // top bit of a row's LAST mask word: no block has written the row
constexpr uint32_t kMaskDeferred = 0x80000001u;        // ... and its cell is not stored yet (block created by the 2nd insert pass)

// COMPACT neighbour rows (binned builder, one mask word, 17 <= K <= 31): 16 ints = 64 B per output row instead of 32 columns:
//   word 0 = the row's mask (bit k <=> offset k has a neighbour), words 1 .. popcount(mask) = the neighbour rows of the SET
//   offsets in ascending k, the rest unspecified.  Rows with more than kCompactIds neighbours do not fit: the builder raises
//   WCN_FLAG_ROW_OVERFLOW and the host rebuilds with dense rows.
constexpr int kCompactPitch = 16;
constexpr int kCompactIds = kCompactPitch - 1;
constexpr int kCompactFlag = (int)0x80000000;  // OR-ed into a `row pitch` kernel argument: the table is compact

constexpr int kIdLateBit = 1 << 30;  // set in a slot's id when the block was created by the SECOND insert pass

// block table slot: key 0 = empty; id = dense block id, valid for LATER launches than the one that created the block
struct __attribute__((aligned(16))) BSlot {
  uint64_t key;
  int32_t id;
  int32_t pad;
};

struct CellTable {
  BSlot* slots;
  int64_t capacity;     // power of two >= 2 * max_blocks
  int32_t* ctr;         // [0] = number of blocks (dense ids 0 .. ctr[0]-1; may exceed max_blocks on overflow)
  uint32_t* halo;       // halo gather list of the current kernel geometry
  uint64_t* blk_key;    // [max_blocks] dense id -> block key
  int32_t* nbtab;       // [max_blocks][32] ids of the 27 neighbour blocks (-1 = absent), [13] = the block itself
  int32_t* cells;       // [max_blocks][512] row ids, -1 = empty
  int64_t max_blocks;
  size_t bytes;
};

struct CellGeom {
  int kx, ky, kz, cx, cy, cz, dx, dy, dz;
  int hx, hy, hz;   // halo per axis (max |offset|)
  int px, py;       // LDS grid pitches (x, y); z pitch 1
  int cells;        // LDS grid size in ints
  int halo_cells;   // entries of the halo gather list
};

static inline size_t align256c(size_t v) { return (v + 255) & ~(size_t)255; }

static inline int64_t cell_capacity(int64_t max_blocks) {
  int64_t c = 1024;
  while (c < 2 * max_blocks) c <<= 1;
  return c;
}

static inline CellTable carve_cells(void* ws, int64_t n, int64_t max_blocks) {
  CellTable t;
  char* p = (char*)ws;
  size_t off = 0;
  auto take = [&](size_t bytes) { char* q = p ? p + off : nullptr; off += align256c(bytes); return q; };
  (void)n;
  t.max_blocks = max_blocks;
  t.capacity = cell_capacity(max_blocks);
  t.ctr = (int32_t*)take(256);
  t.halo = (uint32_t*)take((size_t)16384 * 4);  // (8 + 2*8)^3 - 512 = 13312 entries at most
  t.slots = (BSlot*)take((size_t)t.capacity * sizeof(BSlot));
  t.blk_key = (uint64_t*)take((size_t)max_blocks * 8);
  t.nbtab = (int32_t*)take((size_t)max_blocks * 32 * 4);
  t.cells = (int32_t*)take((size_t)max_blocks * kCells * 4);
  t.bytes = off;
  return t;
}

static inline CellGeom make_cell_geom(const int32_t ksize[3], const int32_t dilation[3]) {
  CellGeom g;
  g.kx = ksize[0]; g.ky = ksize[1]; g.kz = ksize[2];
  g.cx = (g.kx & 1) ? g.kx / 2 : 0; g.cy = (g.ky & 1) ? g.ky / 2 : 0; g.cz = (g.kz & 1) ? g.kz / 2 : 0;
  g.dx = dilation[0]; g.dy = dilation[1]; g.dz = dilation[2];
  auto halo = [](int ks, int c, int d) { const int lo = c * d, hi = (ks - 1 - c) * d; return lo > hi ? lo : hi; };
  g.hx = halo(g.kx, g.cx, g.dx); g.hy = halo(g.ky, g.cy, g.dy); g.hz = halo(g.kz, g.cz, g.dz);
  const int gx = kBlk + 2 * g.hx, gy = kBlk + 2 * g.hy, gz = kBlk + 2 * g.hz;
  // odd y pitch: the 27 cells a voxel probes then fall into 27 different LDS banks for a 3x3x3 kernel (pitches 11, 110)
  g.py = gz | 1;
  g.px = gy * g.py;
  g.cells = gx * g.px;
  g.halo_cells = gx * gy * gz - kCells;
  return g;
}

// slot of an existing block key, or -1
__device__ __forceinline__ int block_find(const BSlot* __restrict__ slots, uint32_t cmask, uint64_t key) {
  uint32_t s = hash_slot(key, cmask);
  for (uint32_t a = 0; a <= cmask; ++a) {
    const uint64_t k = slots[s].key;
    if (k == 0ull) return -1;
    if (k == key) return (int)s;
    s = (s + 1) & cmask;
  }
  return -1;
}

}  // namespace wcn
