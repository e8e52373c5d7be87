// conv_mfma_cs.hip - fused gather -> LDS -> MFMA -> store kernel for the AB (forward) and ABt (dgrad) sparse-conv GEMMs,
// "channel-split" family (round 3): the gathered feature rows are STAGED THROUGH LDS with row-shaped requests and SHARED
// by the waves of a workgroup, every wave owns a 32-channel slice of the output and keeps ITS weight fragments in registers.
//
// Why (measured in rounds 1-2): the register-gather kernel (conv_mfma.hip) is bound by the texture
// addresser, not by HBM - a lane pulling 16-B fragments of its own row makes every wave instruction touch 32 different
// 128-B lines (each line is visited by four instructions), and every 128-row workgroup re-fetches the whole 16 KB weight
// slab of a step through LDS-DMA.  Here, per step (kernel offset k, 64 input channels) of a 128-row mask-sorted tile:
//   * rows: 16 LDS-DMA wave instructions per WORKGROUP, 8 adjacent lanes per 128-B row piece (exec-masked for absent
//     neighbours: no request), landing row-major in a ring stage; the 16-B pieces are XOR-swizzled on the source side
//     (the DMA destination is lane-linear) so that the B-fragment ds_read_b128 (row pitch 128 B) are bank-conflict free;
//   * weights: wave `cs` loads only W[k][64 ci][32 co of its slice] = 4 KB straight HBM/L2 -> VGPR (4 coalesced
//     dwordx4 loads, packed lane-linear by wcn_pack_weight), double buffered in registers - no LDS hop for weights;
//   * MFMA (v_mfma_f32_32x32x16, transposed: A = weight fragment, B = 32 rows): every wave multiplies its slice with
//     ALL row blocks of its row group that have the offset, so the four waves of a workgroup do equal work per step
//     (the register-gather kernel gives a wave the rows and lets it idle at the barrier when its rows lack the offset);
//   * rows that lack the offset read a 128-B zero row in LDS instead of their stage slot (one address select per
//     32-row block and step; no zero fills, no per-fragment selects).
// Workgroup = WR row groups x WC channel slices waves, WC = CO / 32 (shapes measured in DESIGN.md section 4.2a):
//   CO = 128: 4 waves, 128-row tile, a wave holds 4 row blocks x 32 channels;  CO = 96: 3 waves, 96-row tile;
//   CO = 64: 2 waves, 64-row tile.  cin % 64 == 32: the last chunk requests the four pieces that exist, the packed image
//   carries zero weights for the rest.  Three workgroups per CU (CO = 128: 157 VGPRs, 47.7 KB of LDS).
// What was measured on top of this structure and NOT kept (ring depth 3 with counted waits, persistent workgroups with a
// pipelined prologue, dense packing of a step's rows, 128-channel steps, 64-row tiles): DESIGN.md section 4.2a.
//
// Math and epilogue exactly as conv_mfma.hip (same C-ABI entry points pick the kernel by shape):
//   out[r] = act((sum_k in[nbr[r][k]] . W[k] + bias) * scale + shift + residual).
// Reference semantics: warpconvnet/nn/functional/sparse_conv/detail/explicit.py:22-57, 60-92; role of
// _C.mask_gemm.fwd/.dgrad (warpconvnet/csrc/bindings/mask_gemm_bindings.cu:2074-2101); index prefetch idea
// MaskGemm_forward_64x64x32_1s_flat.h:287-296.
#include <cstdlib>

#include "wcn_common.h"

namespace wcn {

typedef __attribute__((ext_vector_type(8))) __bf16 c_bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 c_f16x8;
typedef __attribute__((ext_vector_type(16))) float c_f32x16;

template <typename T> struct CFrag;
template <> struct CFrag<__bf16> {
  typedef c_bf16x8 type;
  static __device__ __forceinline__ c_f32x16 mfma(c_bf16x8 a, c_bf16x8 b, c_f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  }
};
template <> struct CFrag<_Float16> {
  typedef c_f16x8 type;
  static __device__ __forceinline__ c_f32x16 mfma(c_f16x8 a, c_f16x8 b, c_f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  }
};

constexpr int kCsSlabPitch = 28;   // staged table columns (ints): kernel volumes up to 28 (3^3 = 27)
constexpr int kCsMaxK = 28;
constexpr int kCsCIC = 64;         // input channels per step (one 128-B row piece)

// Workgroup shape: WC = CO / 32 channel slices x WR row groups waves; a wave holds RBW 32-row blocks of its 32 channels.
//   CO = 128: RBW 4, WR 1 -> 4 waves, 128-row tile (RBW 2: 64-row tile, half the LDS, twice the weight traffic per row)
//   CO =  64: RBW 2, WR 2 -> 4 waves, 128-row tile;  RBW 2, WR 1 -> 2 waves, 64-row tile (same weight traffic per row)
template <int CO, int RBW_, int WR_>
struct CsCfg {
#ifdef WCN_CS_PAIR  // ablation build (round 6): two (offset, chunk) steps per barrier, four ring stages - profiles/r06_gemm_limits.md
  static constexpr int D = 4;
#else
  static constexpr int D = 2;                    // ring depth (deeper rings need counted waits)
#endif
  static constexpr int WC = CO / 32;             // channel slices (waves across the output width)
  static constexpr int WR = WR_;                 // row groups
  static constexpr int RBW = RBW_;               // 32-row blocks per wave
  static constexpr int WAVES = WC * WR;
  static constexpr int NT = 64 * WAVES;          // threads
  static constexpr int TILE = 32 * RBW * WR;     // output rows per workgroup
  static constexpr int NBLK = TILE / 32;
  static constexpr int DMA_ROWS = TILE / WAVES;  // rows each wave requests per step
  static constexpr int DMA_INSTR = DMA_ROWS / 8;
  static constexpr int STAGE_BYTES = TILE * kCsCIC * 2;
  static_assert(CO == 32 || CO == 64 || CO == 96 || CO == 128, "channel-split kernel: CO in {32, 64, 96, 128}");
  static_assert(DMA_ROWS % 8 == 0 && TILE % 32 == 0 && NT >= TILE, "tile shape");
  static constexpr size_t OFF_NBR = (size_t)D * STAGE_BYTES;
  static constexpr size_t OFF_ROWS = OFF_NBR + (size_t)TILE * kCsSlabPitch * 4;
  static constexpr size_t OFF_MASK = OFF_ROWS + (size_t)TILE * 4;
  static constexpr size_t OFF_WMASK = OFF_MASK + (size_t)TILE * 4;
  static constexpr size_t OFF_ZERO = OFF_WMASK + 32;                   // (up to 8 row blocks)
  static constexpr size_t OFF_EPI = OFF_ZERO + 128;                     // bias / scale / shift, CO floats each
  static constexpr size_t LDS_BYTES = OFF_EPI + 3 * (size_t)CO * 4;
  // compact table rows (kmap_cells.h) stay compact in LDS - [TILE][16] ints instead of the [TILE][28] slab - and everything behind
  // the slab moves up: CO = 64 (dgrad of the headline layer) 24.4 -> 21.4 KB per workgroup = 7 instead of 6 workgroups per CU
  static constexpr size_t SLAB_SAVED = (size_t)TILE * (kCsSlabPitch - 16) * 4;
  static constexpr size_t LDS_BYTES_COMPACT = LDS_BYTES - SLAB_SAVED;
  static constexpr int OUT_PITCH = CO * 2 + 16;  // epilogue stage: +16 B keeps the b128 stage writes conflict-free
  static_assert((size_t)TILE * OUT_PITCH <= OFF_ROWS - SLAB_SAVED, "the epilogue stage reuses the ring and the index slab");
};

// ---- weight packing: [k][chunk][cs][s][lane][j], lane = (h << 5) | m ------------------------------------------------
//   ci = chunk*64 + 16*s + 8*h + j                      (the K index of the MFMA: natural channel order)
//   co = cs*32 + 16*((m >> 2) & 1) + 4*(m >> 3) + (m & 3)
// so that the C fragment of lane (h', n) holds output channels cs*32 + 16*h' + reg, reg = 0..15, of row n.
// Outputs wider than 128 channels: `cout / cob` images of `cob` channels behind each other (column block on grid.y of the
// main kernel), each the image of w[:, :, cb * cob : (cb + 1) * cob].
template <typename TS, typename TD>
__device__ __forceinline__ void pack_weight_cs_element(const TS* __restrict__ w, TD* __restrict__ packed, int64_t e, int K, int cin,
                                                       int cout, int cob, int transpose, int flip) {
  const int WC = cob / 32, nchunk = (cin + kCsCIC - 1) / kCsCIC;  // a last chunk of 32 channels is zero-padded to 64
  const int64_t image = (int64_t)K * nchunk * kCsCIC * cob;
  if (e >= image * (cout / cob)) return;
  const int cb = (int)(e / image);
  int64_t t = e - cb * image;
  const int j = (int)(t % 8); t /= 8;
  const int lane = (int)(t % 64); t /= 64;
  const int s = (int)(t % 4); t /= 4;
  const int cs = (int)(t % WC); t /= WC;
  const int chunk = (int)(t % nchunk); t /= nchunk;
  const int k = (int)t;
  const int h = lane >> 5, m = lane & 31;
  const int ci = chunk * kCsCIC + 16 * s + 8 * h + j;
  const int co = cb * cob + cs * 32 + 16 * ((m >> 2) & 1) + 4 * (m >> 3) + (m & 3);
  const int kw = flip ? (K - 1 - k) : k;
  // not transposed: w[kw][ci][co] ([K, cin, cout]); transposed: w is the forward weight [K, cout, cin]
  const int64_t src = transpose ? (((int64_t)kw * cout + co) * cin + ci) : (((int64_t)kw * cin + ci) * cout + co);
  packed[e] = ci < cin ? (TD)w[src] : (TD)0;
}

template <typename TS, typename TD>
__global__ void pack_weight_cs_kernel(const TS* __restrict__ w, TD* __restrict__ packed, int K, int cin, int cout, int cob,
                                      int transpose, int flip) {
  pack_weight_cs_element(w, packed, (int64_t)blockIdx.x * blockDim.x + threadIdx.x, K, cin, cout, cob, transpose, flip);
}

// Both images of a training step in ONE launch: blockIdx.y = 0 the forward image of w [K, cin, cout], 1 the dgrad image
// (kernel-side roles exchanged: reduce over cout, produce cin; transposed, k-flipped for a submanifold map).  An optimizer step
// invalidates both at once, so every layer of a network saves a launch per iteration.
template <typename TD>
__global__ void pack_weight_cs_pair_kernel(const float* __restrict__ w, TD* __restrict__ packed_fwd, TD* __restrict__ packed_dgrad,
                                           int K, int cin, int cout, int cob_fwd, int cob_dgrad, int flip_dgrad) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (blockIdx.y == 0) pack_weight_cs_element(w, packed_fwd, e, K, cin, cout, cob_fwd, 0, 0);
  else pack_weight_cs_element(w, packed_dgrad, e, K, cout, cin, cob_dgrad, 1, flip_dgrad);
}

#ifdef WCN_PROF
// dev build (`make prof`, tools/prof_phases.py): thread 0 of every 4th workgroup stamps its phases - per slot: prologue, step
// loop, epilogue clocks | 1 | steps | in-loop wait + barrier, request issue, LDS reads + MFMAs
__device__ unsigned long long g_csprof[2048 * 8];
// wall-clock start / end (100 MHz) and XCC_ID : HW_ID of every workgroup of the LAST launch (tools/prof_cs_timeline.py)
__device__ unsigned long long g_cstime[16384 * 4];
#define CS_CLK() __builtin_readcyclecounter()
#define CS_PROF(stmt) do { if (cs_prof) { stmt; } } while (0)
#else
#define CS_PROF(stmt)
#endif

// ---- main kernel --------------------------------------------------------------------------------------------------------
template <typename T, int CO, int RBW_, int WR_, int MINW>
__global__ __launch_bounds__(64 * (CO / 32) * WR_, MINW) void gather_gemm_cs_kernel(const T* __restrict__ in, const T* __restrict__ wp,
                                                                T* __restrict__ out, const int32_t* __restrict__ nbr,
                                                                const uint32_t* __restrict__ mask,
                                                                const int32_t* __restrict__ perm, const ConvEpilogue epi,
                                                                int64_t n_out, int cin, int K, int kp,
                                                                float* __restrict__ out32, int ldc) {
  // ldc: channels of an output (and residual) row; blockIdx.y: the CO-wide column block of it this workgroup produces
  typedef CsCfg<CO, RBW_, WR_> G;
  typedef typename CFrag<T>::type frag_t;
  constexpr int WC = G::WC, RBW = G::RBW, SP = kCsSlabPitch, TILE = G::TILE, NT = G::NT;
  constexpr int kCsStageBytes = G::STAGE_BYTES;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  // `mask` null with a table: `nbr` holds COMPACT rows (kmap_cells.h: 16 ints - the row's mask, then the neighbour rows of its set
  // offsets in ascending k; written by wcn_kmap_build_binned) - 64 B per row instead of a 128-B table row plus a 128-B line for the
  // 4 bytes of mask[perm[i]].  They stay compact in LDS: the row DMA derives its source row from (mask, k) on the fly.
  const bool compact = mask == nullptr && nbr != nullptr;
  const size_t shift = compact ? G::SLAB_SAVED : 0;                         // (launch_cs sizes the dynamic LDS accordingly)
  char* s_ring = smem;                                                     // [D][128 rows][128 B]
  int32_t* s_nbr = reinterpret_cast<int32_t*>(smem + G::OFF_NBR);         // [TILE][SP], or [TILE][16] compact rows
  int32_t* s_rows = reinterpret_cast<int32_t*>(smem + G::OFF_ROWS - shift);       // [TILE]
  uint32_t* s_mask = reinterpret_cast<uint32_t*>(smem + G::OFF_MASK - shift);     // [TILE]
  uint32_t* s_wmask = reinterpret_cast<uint32_t*>(smem + G::OFF_WMASK - shift);   // [4]: OR of the row masks per 32-row block
  char* s_zero = smem + G::OFF_ZERO - shift;                                       // 128 B of zeros
  float* s_epi = reinterpret_cast<float*>(smem + G::OFF_EPI - shift);             // [3][CO]: bias, scale, shift

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h = lane >> 5, n = lane & 31;
  const int cs = wave % WC, rg = wave / WC;  // channel slice, row group
  const int nchunk = (cin + kCsCIC - 1) / kCsCIC;
  const int last_pieces = (cin - (nchunk - 1) * kCsCIC) / 8;  // 16-B pieces of the last chunk that exist (8, or 4 when cin % 64 == 32)
  const int64_t row0 = (int64_t)blockIdx.x * TILE;
  const int col0 = blockIdx.y * CO;  // first output channel of this column block
  wp += (size_t)blockIdx.y * ((size_t)K * nchunk * kCsCIC * CO);
#ifdef WCN_PROF
  const bool cs_prof = tid == 0 && (blockIdx.x & 3) == 0 && (blockIdx.x >> 2) < 2048;
  unsigned long long pt0 = CS_CLK(), pt1 = 0, pt2 = 0, pa = 0, pw = 0, pi = 0, pc = 0, pn = 0;
  if (tid == 0 && blockIdx.y == 0 && blockIdx.x < 16384) {
    g_cstime[blockIdx.x * 4] = wall_clock64();
    g_cstime[blockIdx.x * 4 + 2] = ((unsigned long long)__builtin_amdgcn_s_getreg(0xF814) << 32) | (unsigned)__builtin_amdgcn_s_getreg(0xF804);
  }
#endif

  // ---- output row ids (through the mask-sorted permutation), masks, index slab ----
  if (tid < TILE) {
    const int64_t pr = row0 + tid;
    int32_t r = -1;
    if (pr < n_out) r = perm ? perm[pr] : (int32_t)pr;
    s_rows[tid] = r;
  }
  if (tid < 8) reinterpret_cast<int4*>(s_zero)[tid] = make_int4(0, 0, 0, 0);
  if (tid < 8) s_wmask[tid] = 0;
  if (last_pieces < 8) {
    // cin % 64 == 32: the upper half of the last chunk is never requested; it meets zero weights in the packed image, so it
    // only has to be FINITE - clear the ring once (uninitialised LDS may hold NaN patterns)
    for (int e = tid; e < (int)(G::OFF_NBR / 16); e += NT) reinterpret_cast<int4*>(smem)[e] = make_int4(0, 0, 0, 0);
  }
  // per-channel epilogue terms: requested first, used last (their latency is off the critical path)
  for (int c = tid; c < CO; c += NT) {
    s_epi[c] = epi.bias ? epi.bias[col0 + c] : 0.f;
    s_epi[CO + c] = epi.scale ? epi.scale[col0 + c] : 1.f;
    s_epi[2 * CO + c] = epi.scale ? epi.shift[col0 + c] : 0.f;
  }
  __syncthreads();
  if (compact) {
    constexpr int kIterC = (TILE * 4 + NT - 1) / NT;
    int32_t rr[kIterC];
    int4 vv[kIterC];
#pragma unroll
    for (int t = 0; t < kIterC; ++t) {
      const int e = tid + t * NT;
      rr[t] = (e < TILE * 4) ? s_rows[e >> 2] : -1;
    }
#pragma unroll
    for (int t = 0; t < kIterC; ++t) {
      const int e = tid + t * NT;
      vv[t] = make_int4(0, 0, 0, 0);  // (no row: mask 0)
      if (rr[t] >= 0) {  // read once: non-temporal
        typedef __attribute__((ext_vector_type(4))) int i32x4;
        const i32x4 q = __builtin_nontemporal_load(reinterpret_cast<const i32x4*>(nbr + (int64_t)rr[t] * 16) + (e & 3));
        vv[t] = make_int4(q.x, q.y, q.z, q.w);
      }
    }
#pragma unroll
    for (int t = 0; t < kIterC; ++t) {
      const int e = tid + t * NT;
      if (e >= TILE * 4) continue;
      if ((e & 3) == 0) {  // word 0 of the row: its mask
        // (a row that did not fit its compact row - more than 15 neighbours - belongs to a build that is flagged ROW_OVERFLOW and
        // redone with dense rows: here it has no neighbours)
        uint32_t m = (uint32_t)vv[t].x;
        if (__popc(m) > 15) m = 0u;
        vv[t].x = (int)m;
        s_mask[e >> 2] = m;
        if (m) atomicOr(&s_wmask[e >> 7], m);
      }
      reinterpret_cast<int4*>(s_nbr)[e] = vv[t];
    }
  } else {
    {
      uint32_t my_mask = 0;
      if (tid < TILE) {
        const int32_t r = s_rows[tid];
        if (r >= 0) my_mask = mask ? mask[r] : 1u;  // (no table: the identity map of a 1 x 1 x 1 kernel, see below)
        s_mask[tid] = my_mask;
        if (my_mask) atomicOr(&s_wmask[tid >> 5], my_mask);
      }
    }
    // all row ids first, then all table loads, then all LDS writes (one global round trip)
    constexpr int kVec = SP / 4;  // 16-B pieces per slab row
    constexpr int kIter = (TILE * kVec + NT - 1) / NT;
    int32_t rr[kIter];
    int4 vv[kIter];
#pragma unroll
    for (int t = 0; t < kIter; ++t) {
      const int e = tid + t * NT;
      rr[t] = (e < TILE * kVec) ? s_rows[e / kVec] : -1;
    }
#pragma unroll
    for (int t = 0; t < kIter; ++t) {
      const int e = tid + t * NT;
      const int c = e % kVec;
      vv[t] = make_int4(-1, -1, -1, -1);
      if (rr[t] >= 0 && (c * 4 < kp)) {  // read once: non-temporal
        typedef __attribute__((ext_vector_type(4))) int i32x4;
        if (nbr) {
          const i32x4 q = __builtin_nontemporal_load(reinterpret_cast<const i32x4*>(nbr + (int64_t)rr[t] * kp) + c);
          vv[t] = make_int4(q.x, q.y, q.z, q.w);
        } else if (c == 0) {
          vv[t].x = rr[t];  // nbr == null (K = 1): every row is its own only neighbour - a dense [N, cin] x [cin, cout] product
        }
      }
    }
#pragma unroll
    for (int t = 0; t < kIter; ++t) {
      const int e = tid + t * NT;
      if (e >= TILE * kVec) continue;
      reinterpret_cast<int4*>(s_nbr + (e / kVec) * SP)[e % kVec] = vv[t];
    }
  }
  __syncthreads();
  uint32_t rb_mask[RBW];
  uint32_t block_mask = 0u;
#pragma unroll
  for (int rb = 0; rb < RBW; ++rb) rb_mask[rb] = __builtin_amdgcn_readfirstlane(s_wmask[rg * RBW + rb]);  // SGPR
#pragma unroll
  for (int q = 0; q < G::NBLK; ++q) block_mask |= s_wmask[q];
  block_mask = __builtin_amdgcn_readfirstlane(block_mask);
  // the DMA instructions of this wave cover tile rows [DMA_ROWS * wave, DMA_ROWS * (wave + 1)): OR of their blocks' masks (skip empty instructions)
  uint32_t dma_mask = 0u;
#pragma unroll
  for (int q = 0; q <= (G::DMA_ROWS + 30) / 32; ++q)  // (rows that do not start on a block boundary reach one block further)
    if ((wave * G::DMA_ROWS) / 32 + q < G::NBLK && (wave * G::DMA_ROWS) / 32 + q <= (wave * G::DMA_ROWS + G::DMA_ROWS - 1) / 32)
      dma_mask |= s_wmask[(wave * G::DMA_ROWS) / 32 + q];
  dma_mask = __builtin_amdgcn_readfirstlane(dma_mask);
  // mask of the row this lane holds in the B fragment of row block rb
  uint32_t mrow[RBW];
#pragma unroll
  for (int rb = 0; rb < RBW; ++rb) mrow[rb] = s_mask[(rg * RBW + rb) * 32 + n];

  c_f32x16 acc[RBW];
#pragma unroll
  for (int rb = 0; rb < RBW; ++rb)
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[rb][q] = 0.f;

  if (block_mask != 0u) {
    // per-lane constants (LDS addresses as plain 32-bit integers: no generic-pointer null checks in the loop)
    typedef const __attribute__((address_space(3))) frag_t* lds_frag_p;
    typedef const __attribute__((address_space(3))) int32_t* lds_i32_p;
    const uint32_t lds0 = lds_addr_of(smem);
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);  // wave-uniform copy (SGPR)
    const int sw = (n >> 1) & 7;                      // swizzle of the row this lane reads
    uint32_t boff[4];                                 // byte offset of k-slice s inside a staged row
#pragma unroll
    for (int s = 0; s < 4; ++s) boff[s] = (uint32_t)(((2 * s + h) ^ sw) << 4);
    const uint32_t zaddr = lds0 + (uint32_t)(G::OFF_ZERO - shift);
    const uint32_t rowaddr = lds0 + (uint32_t)(rg * RBW * 32 + n) * 128u;  // + stage, + rb * 4096
    const char* wbase = reinterpret_cast<const char*>(wp) + (size_t)cs * 4096 + lane * 16;
    const size_t wstep = (size_t)WC * 4096;           // bytes of one (k, chunk) weight slab
    const uint32_t rowbytes = (uint32_t)cin * 2u;
    // DMA: lane covers row (lane >> 3) of an 8-row instruction, 16-B position (lane & 7); source piece = position ^ swizzle
    const uint32_t idxaddr = lds0 + (uint32_t)G::OFF_NBR + (uint32_t)((wave * G::DMA_ROWS + (lane >> 3)) * SP) * 4u;
    const uint32_t cptaddr = lds0 + (uint32_t)G::OFF_NBR + (uint32_t)((wave * G::DMA_ROWS + (lane >> 3)) * 16) * 4u;  // compact rows
    uint32_t mdma[G::DMA_INSTR];  // masks of the rows this lane requests (compact rows)
#pragma unroll
    for (int it = 0; it < G::DMA_INSTR; ++it) mdma[it] = compact ? s_mask[wave * G::DMA_ROWS + it * 8 + (lane >> 3)] : 0u;
    // (tile row >> 1) & 7 of the row a lane requests: (lane >> 4) + 4 * (instruction index + first instruction of the wave)
    const int first_odd = ((wave * G::DMA_ROWS) >> 3) & 1;
    const int piece_e = (lane & 7) ^ ((lane >> 4) + 4 * first_odd), piece_o = (lane & 7) ^ ((lane >> 4) + 4 * (1 - first_odd));
    const char* gsrc_e = reinterpret_cast<const char*>(in) + (piece_e << 4);
    const char* gsrc_o = reinterpret_cast<const char*>(in) + (piece_o << 4);

    auto issue_rows = [&](int buf, int k, int chunk) {
      if (!((dma_mask >> k) & 1u)) return;  // wave-uniform: none of this wave's 32 DMA rows has the offset
      const uint32_t dst = lds0 + (uint32_t)buf * kCsStageBytes + (uint32_t)wave_u * (G::DMA_ROWS * 128u);
      int32_t idx[G::DMA_INSTR];
      if (compact) {  // the neighbour at offset k is word 1 + (set offsets below k) of the row
        const uint32_t below = (1u << k) - 1u;
#pragma unroll
        for (int it = 0; it < G::DMA_INSTR; ++it) {
          const int32_t v = *(lds_i32_p)(uintptr_t)(cptaddr + (uint32_t)((it * 8 * 16 + 1 + __popc(mdma[it] & below)) * 4));
          idx[it] = ((mdma[it] >> k) & 1u) ? v : -1;
        }
      } else {
#pragma unroll
        for (int it = 0; it < G::DMA_INSTR; ++it) idx[it] = *(lds_i32_p)(uintptr_t)(idxaddr + (uint32_t)(k * 4 + it * 8 * SP * 4));
      }
      const int npieces = chunk + 1 < nchunk ? 8 : last_pieces;  // pieces of this chunk that exist in the row
#pragma unroll
      for (int it = 0; it < G::DMA_INSTR; ++it) {
        if (idx[it] >= 0 && ((it & 1) ? piece_o : piece_e) < npieces) {
          const char* src = ((it & 1) ? gsrc_o : gsrc_e) + (uint64_t)(uint32_t)idx[it] * rowbytes + (uint32_t)(chunk * 128);
          glds16(src, dst + it * 1024);
        }
      }
    };
    auto load_w = [&](frag_t (&w)[4], int k, int chunk) {
      const char* p = wbase + (size_t)(k * nchunk + chunk) * wstep;
#pragma unroll
      for (int s = 0; s < 4; ++s) w[s] = *reinterpret_cast<const frag_t*>(p + s * 1024);
    };
    // B fragments of row block rb (rows that lack the offset read the zero row); the fragments of the next active block
    // are requested before the MFMAs of the current one
    auto load_b = [&](frag_t (&b)[4], uint32_t sbase, int rb, int k) {
      const uint32_t rbase = ((mrow[rb] >> k) & 1u) ? (sbase + (uint32_t)rb * 4096u) : zaddr;
#pragma unroll
      for (int s = 0; s < 4; ++s) b[s] = *(lds_frag_p)(uintptr_t)(rbase + boff[s]);
    };
    auto compute = [&](const frag_t (&w)[4], int buf, int k) {
      const uint32_t sbase = rowaddr + (uint32_t)buf * kCsStageBytes;
      // ONE set of B fragments: the LDS reads of row block rb + 1 are issued behind the MFMAs of block rb and ride under their
      // execution.  (Round 6: the double-buffered set - reads of rb + 1 in front of the MFMAs of rb - cost 27 VGPRs and bought
      // nothing: forward 204.7 vs 203.6 us, dgrad 222.0 vs 229.0 us WITH the single set, same box; `-DWCN_CS_BDOUBLE`.)
#ifdef WCN_CS_BDOUBLE
      frag_t b[2][4];
      if ((rb_mask[0] >> k) & 1u) load_b(b[0], sbase, 0, k);
#pragma unroll
      for (int rb = 0; rb < RBW; ++rb) {
        if (rb + 1 < RBW && ((rb_mask[rb + 1 < RBW ? rb + 1 : rb] >> k) & 1u)) load_b(b[(rb + 1) & 1], sbase, rb + 1, k);
        if ((rb_mask[rb] >> k) & 1u) {  // wave-uniform: some row of this block has the offset
#pragma unroll
          for (int s = 0; s < 4; ++s) acc[rb] = CFrag<T>::mfma(w[s], b[rb & 1][s], acc[rb]);
        }
      }
#else
      frag_t b[4];
#pragma unroll
      for (int rb = 0; rb < RBW; ++rb) {
        if ((rb_mask[rb] >> k) & 1u) {  // wave-uniform: some row of this block has the offset
          load_b(b, sbase, rb, k);
#pragma unroll
          for (int s = 0; s < 4; ++s) acc[rb] = CFrag<T>::mfma(w[s], b[s], acc[rb]);
        }
      }
#endif
    };
    // step iterator over (set bits of block_mask ascending) x (channel chunks)
    uint32_t rem = block_mask;
    auto next_step = [&](int& k, int& chunk) -> bool {
      if (k >= 0 && chunk + 1 < nchunk) { ++chunk; return true; }
      if (rem == 0u) return false;
      k = __builtin_ctz(rem);
      rem &= rem - 1u;
      chunk = 0;
      return true;
    };
    auto sync_step = [&]() {
      __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0), gfx9 encoding; also resets hipcc's own load scoreboard
      __syncthreads();
    };

#ifdef WCN_CS_PAIR
    // TWO steps per barrier: the rows and weight fragments of steps s+2, s+3 are requested while s, s+1 multiply
    frag_t W0[4], W1[4], W2[4], W3[4];
    int ka = -1, ca = 0, kb = -1, cb = 0, kc = -1, cc = 0, kd = -1, cd = 0;
    bool hb, hc, hd;
    next_step(ka, ca);
    kb = ka; cb = ca;
    hb = next_step(kb, cb);
    issue_rows(0, ka, ca);
    load_w(W0, ka, ca);
    if (hb) { issue_rows(1, kb, cb); load_w(W1, kb, cb); }
    for (;;) {
      kc = hb ? kb : ka; cc = hb ? cb : ca;
      hc = hb && next_step(kc, cc);
      kd = kc; cd = cc;
      hd = hc && next_step(kd, cd);
      sync_step();
      if (hc) { issue_rows(2, kc, cc); load_w(W2, kc, cc); }
      if (hd) { issue_rows(3, kd, cd); load_w(W3, kd, cd); }
      compute(W0, 0, ka);
      if (hb) compute(W1, 1, kb);
      if (!hc) break;
      ka = hd ? kd : kc; ca = hd ? cd : cc;
      const bool ha = hd && next_step(ka, ca);
      kb = ka; cb = ca;
      hb = ha && next_step(kb, cb);
      sync_step();
      if (ha) { issue_rows(0, ka, ca); load_w(W0, ka, ca); }
      if (hb) { issue_rows(1, kb, cb); load_w(W1, kb, cb); }
      compute(W2, 2, kc);
      if (hd) compute(W3, 3, kd);
      if (!ha) break;
    }
#else
    frag_t Wa[4], Wb[4];
    int k0 = -1, c0 = 0, k1 = -1, c1 = 0;
    next_step(k0, c0);
    CS_PROF(pt1 = CS_CLK());
    issue_rows(0, k0, c0);
    load_w(Wa, k0, c0);
    for (;;) {
      k1 = k0; c1 = c0;
      const bool has1 = next_step(k1, c1);
      CS_PROF(pa = CS_CLK());
      sync_step();  // stage 0 has landed for every wave; every wave is done reading stage 1
      CS_PROF(pw += CS_CLK() - pa; pa = CS_CLK());
#ifdef WCN_CS_WFIRST  // ablation build (round 6): the L2-resident weight fragments requested in front of the row DMA
      if (has1) { load_w(Wb, k1, c1); issue_rows(1, k1, c1); }
#else
      if (has1) { issue_rows(1, k1, c1); load_w(Wb, k1, c1); }
#endif
      CS_PROF(pi += CS_CLK() - pa; pa = CS_CLK());
      compute(Wa, 0, k0);
      CS_PROF(pc += CS_CLK() - pa; ++pn);
      if (!has1) break;
      k0 = k1; c0 = c1;
      const bool has0 = next_step(k0, c0);
      CS_PROF(pa = CS_CLK());
      sync_step();
      CS_PROF(pw += CS_CLK() - pa; pa = CS_CLK());
#ifdef WCN_CS_WFIRST
      if (has0) { load_w(Wa, k0, c0); issue_rows(0, k0, c0); }
#else
      if (has0) { issue_rows(0, k0, c0); load_w(Wa, k0, c0); }
#endif
      CS_PROF(pi += CS_CLK() - pa; pa = CS_CLK());
      compute(Wb, 1, k1);
      CS_PROF(pc += CS_CLK() - pa; ++pn);
      if (!has0) break;
    }
#endif
    CS_PROF(pt2 = CS_CLK());
  }

  // ---- epilogue: lane (h, n) of wave (rg, cs) holds channels cs*32 + 16*h + q, q = 0..15, of row (rg, rb, n) ----
  const int cbase = cs * 32 + 16 * h;
  if (out32) {
    // fp32 output (the fp32-feature path: fp16 operands, fp32 accumulate, unrounded result)
#pragma unroll
    for (int rb = 0; rb < RBW; ++rb) {
      const int32_t r = s_rows[(rg * RBW + rb) * 32 + n];
      if (r < 0) continue;
      float* dst = out32 + (int64_t)r * ldc + col0 + cbase;
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const float4 bv = reinterpret_cast<const float4*>(s_epi + cbase)[v];
        reinterpret_cast<float4*>(dst)[v] = make_float4(acc[rb][4 * v + 0] + bv.x, acc[rb][4 * v + 1] + bv.y,
                                                        acc[rb][4 * v + 2] + bv.z, acc[rb][4 * v + 3] + bv.w);
      }
    }
    return;
  }
  __syncthreads();  // ring and index slab are dead: reuse them as the [TILE][OUT_PITCH] output stage
#pragma unroll
  for (int rb = 0; rb < RBW; ++rb) {
    frag_t lo, hi;
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const float4 bv = reinterpret_cast<const float4*>(s_epi + cbase)[v];
      const float4 sv = reinterpret_cast<const float4*>(s_epi + CO + cbase)[v];
      const float4 tv = reinterpret_cast<const float4*>(s_epi + 2 * CO + cbase)[v];
      float f[4] = {(acc[rb][4 * v + 0] + bv.x) * sv.x + tv.x, (acc[rb][4 * v + 1] + bv.y) * sv.y + tv.y,
                    (acc[rb][4 * v + 2] + bv.z) * sv.z + tv.z, (acc[rb][4 * v + 3] + bv.w) * sv.w + tv.w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (epi.relu && !epi.residual) f[j] = fmaxf(f[j], 0.f);  // (with a residual the activation follows the add below)
        const int q = 4 * v + j;
        if (q < 8) lo[q] = (T)f[j]; else hi[q - 8] = (T)f[j];
      }
    }
    frag_t* sp = reinterpret_cast<frag_t*>(smem + (size_t)((rg * RBW + rb) * 32 + n) * G::OUT_PITCH + cbase * 2);
    sp[0] = lo;
    sp[1] = hi;
  }
  __syncthreads();
  {
    constexpr int kPieces = CO / 8;  // 16-B pieces per output row
    constexpr int kLanesPerRow = kPieces <= 4 ? 4 : (kPieces <= 8 ? 8 : 16);  // lanes set aside per row (CO = 96: 12 of 16 work)
    constexpr int kRowsPerInstr = 64 / kLanesPerRow;
    constexpr int kStores = G::DMA_ROWS / kRowsPerInstr;  // every wave stores TILE / WAVES rows
    constexpr int kBatch = kStores < 4 ? kStores : 4;     // rows in flight per lane (residual loads)
    const int piece = lane % kLanesPerRow, rsub = lane / kLanesPerRow;
#pragma unroll
    for (int j0 = 0; j0 < kStores; j0 += kBatch) {
      int32_t orow[kBatch];
      frag_t ov[kBatch], rv[kBatch];
#pragma unroll
      for (int j = 0; j < kBatch; ++j)
        orow[j] = j0 + j < kStores ? s_rows[wave * G::DMA_ROWS + (j0 + j) * kRowsPerInstr + rsub] : -1;  // (kStores % kBatch != 0: 160-row tiles)
      if (epi.residual) {  // residual rows are read the way the output is written: whole rows, adjacent lanes, all in flight
#pragma unroll
        for (int j = 0; j < kBatch; ++j)
          if (orow[j] >= 0)
            rv[j] = __builtin_nontemporal_load(reinterpret_cast<const frag_t*>(
                reinterpret_cast<const T*>(epi.residual) + (int64_t)orow[j] * ldc + col0 + piece * 8));
      }
#pragma unroll
      for (int j = 0; j < kBatch; ++j)
        if (j0 + j < kStores)
          ov[j] = *reinterpret_cast<const frag_t*>(
              smem + (size_t)(wave * G::DMA_ROWS + (j0 + j) * kRowsPerInstr + rsub) * G::OUT_PITCH + piece * 16);
#pragma unroll
      for (int j = 0; j < kBatch; ++j) {
        if (orow[j] < 0 || piece >= kPieces) continue;
        frag_t o = ov[j];
        if (epi.residual) {
#pragma unroll
          for (int q = 0; q < 8; ++q) {
            float f = (float)o[q] + (float)rv[j][q];
            if (epi.relu) f = fmaxf(f, 0.f);
            o[q] = (T)f;
          }
        }
        // streamed once: non-temporal, so the output does not push the gathered input out of the caches
        __builtin_nontemporal_store(o, reinterpret_cast<frag_t*>(out + (int64_t)orow[j] * ldc + col0 + piece * 8));
      }
    }
  }
#ifdef WCN_PROF
  if (tid == 0 && blockIdx.y == 0 && blockIdx.x < 16384) g_cstime[blockIdx.x * 4 + 1] = wall_clock64();
  if (cs_prof) {
    unsigned long long* p = g_csprof + (size_t)(blockIdx.x >> 2) * 8;
    const unsigned long long pt3 = CS_CLK();
    p[0] += pt1 - pt0; p[1] += pt2 - pt1; p[2] += pt3 - pt2; p[3] += 1; p[4] += pn; p[5] += pw; p[6] += pi; p[7] += pc;
  }
#endif
}

template <typename T, int CO, int RBW, int WR, int MINW>
static int launch_cs(const void* in, const void* wp, void* out, const int32_t* nbr, const uint32_t* mask,
                     const int32_t* perm, const ConvEpilogue& epi, int64_t n_out, int cin, int cout, int K, float* out32,
                     hipStream_t s) {
  typedef CsCfg<CO, RBW, WR> G;
  static unsigned long long attr_done = 0ull;  // per device (wcn_common.h)
  const int rc = once_per_device(attr_done, [] {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(gather_gemm_cs_kernel<T, CO, RBW, WR, MINW>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::LDS_BYTES) == hipSuccess;
  });
  if (rc != WCN_SUCCESS) return rc;
  const int kp = wcn_kmap_row_pitch(K);
  const size_t lds = (mask == nullptr && nbr != nullptr) ? G::LDS_BYTES_COMPACT : G::LDS_BYTES;  // (compact rows: a smaller slab)
  hipLaunchKernelGGL((gather_gemm_cs_kernel<T, CO, RBW, WR, MINW>), dim3((unsigned)ceil_div(n_out, G::TILE), (unsigned)(cout / CO)),
                     dim3(G::NT), lds, s, (const T*)in, (const T*)wp, (T*)out, nbr, mask, perm, epi, n_out, cin, K, kp,
                     out32, cout);
  return launch_status();
}

// WARPCONVNET_AMD_GEMM_CS: 1 (default) = shapes below take this family, 0 = the register-gather kernels (conv_mfma.hip)
static int cs_mode() {
  static const int v = [] {
    const char* e = getenv("WARPCONVNET_AMD_GEMM_CS");
    return e ? atoi(e) : 1;
  }();
  return v;
}

// Width of the column blocks an output of `cout` channels is produced in (0: not this family's).  Up to 128 channels: one
// block; wider: the widest of 128 / 96 / 64 that divides it (256 = 2 x 128, 192 = 2 x 96, 320 = 5 x 64) - the rows are gathered
// once per block, which the coarse levels of a U-Net (a few thousand rows, 192 - 512 channels) repay with 2 - 4 x the workgroups.
static int cs_col_block(int cout) {
  if (cout <= 128) return (cout == 64 || cout == 96 || cout == 128) ? cout : 0;
  if (cout > 1024) return 0;
  return cout % 128 == 0 ? 128 : cout % 96 == 0 ? 96 : cout % 64 == 0 ? 64 : 0;
}

bool gather_gemm_cs_supported(int cin, int cout, int K, int dtype) {
  if (cs_mode() == 0) return false;
  if (dtype != WCN_F16 && dtype != WCN_BF16) return false;
  if (K < 1 || K > kCsMaxK) return false;
  if (cin < kCsCIC || cin % 32 != 0) return false;  // (a last chunk of 32 channels runs zero-padded)
  return cs_col_block(cout) != 0;
}

template <typename T>
static int dispatch_cs(const void* in, const void* wp, void* out, const int32_t* nbr, const uint32_t* mask,
                       const int32_t* perm, const ConvEpilogue& epi, int64_t n_out, int cin, int cout, int K, float* out32,
                       hipStream_t s) {
  // Measured on the 1 M-voxel scenes (uniform / surface, in-step us): CO = 128 with 128-row tiles 208 / 251 vs 64-row tiles
  // 238 / 272 (twice the weight traffic per row); CO = 64 with 2 waves x 64 rows 256 / 374 vs 4 waves x 128 rows 264 / 378
  // vs 2 waves x 128 rows 275 / 392.
  switch (cs_col_block(cout)) {
#ifdef WCN_CS_PAIR  // (four ring stages and four weight-fragment sets: one wave per SIMD less)
    case 64: return launch_cs<T, 64, 2, 1, 3>(in, wp, out, nbr, mask, perm, epi, n_out, cin, cout, K, out32, s);
    case 96: return launch_cs<T, 96, 3, 1, 2>(in, wp, out, nbr, mask, perm, epi, n_out, cin, cout, K, out32, s);
    case 128: return launch_cs<T, 128, 4, 1, 2>(in, wp, out, nbr, mask, perm, epi, n_out, cin, cout, K, out32, s);
#else
    // (CO = 64, round 6 again: 4 waves x 64 rows `<64, 1, 2>` 251.8 us, 2 waves x 96 rows `<64, 3, 1>` 255.8 us, against 221.8)
    case 64: return launch_cs<T, 64, 2, 1, 4>(in, wp, out, nbr, mask, perm, epi, n_out, cin, cout, K, out32, s);
    case 96: return launch_cs<T, 96, 3, 1, 3>(in, wp, out, nbr, mask, perm, epi, n_out, cin, cout, K, out32, s);  // 3 waves x 96 rows
    // CO = 128, round 6: 96-row tiles, FOUR workgroups a CU (122 VGPRs with the single B-fragment set, 32.6 KB) instead of 128-row
    // tiles and three (141 VGPRs, 42.3 KB): 197.3 vs 200.3 us in the trace, 0.207 vs 0.212 ms in the step, surface scene 728.6 vs
    // 730.1 M voxels/s (same box; `-DWCN_CS_RBW4` builds the old shape).  160-row tiles `<T, 128, 5, 1, 3>` - weight fragments per
    // row -20 %, 166 VGPRs, 54.2 KB - run 244.6 us: the third workgroup of a CU no longer fits.
#ifdef WCN_CS_RBW4
    case 128: return launch_cs<T, 128, 4, 1, 3>(in, wp, out, nbr, mask, perm, epi, n_out, cin, cout, K, out32, s);
#else
    case 128: return launch_cs<T, 128, 3, 1, 4>(in, wp, out, nbr, mask, perm, epi, n_out, cin, cout, K, out32, s);
#endif
#endif
    default: return WCN_ERROR_UNSUPPORTED_CONFIG;
  }
}

int conv_gather_gemm_cs(const void* in, const void* wp, void* out, const int32_t* nbr, const uint32_t* mask,
                        const int32_t* perm, const ConvEpilogue& epi, int64_t n_out, int cin, int cout, int K, int dtype,
                        float* out32, hipStream_t s) {
  if (!gather_gemm_cs_supported(cin, cout, K, dtype)) return WCN_ERROR_UNSUPPORTED_CONFIG;
  if (dtype == WCN_BF16) return dispatch_cs<__bf16>(in, wp, out, nbr, mask, perm, epi, n_out, cin, cout, K, out32, s);
  return dispatch_cs<_Float16>(in, wp, out, nbr, mask, perm, epi, n_out, cin, cout, K, out32, s);
}

int pack_weight_cs(const void* w, int w_is_f32, int K, int cin, int cout, int dtype, int transpose, int flip, void* packed,
                   hipStream_t s) {
  if (!gather_gemm_cs_supported(cin, cout, K, dtype)) return WCN_ERROR_UNSUPPORTED_CONFIG;
  const int64_t total = (int64_t)K * ((cin + kCsCIC - 1) / kCsCIC) * kCsCIC * cout;  // = wcn_packed_weight_elements
  const dim3 grid((unsigned)ceil_div(total, 256)), block(256);
  const int cob = cs_col_block(cout);
  if (w_is_f32) {
    if (dtype == WCN_BF16)
      hipLaunchKernelGGL((pack_weight_cs_kernel<float, __bf16>), grid, block, 0, s, (const float*)w, (__bf16*)packed, K, cin,
                         cout, cob, transpose, flip);
    else
      hipLaunchKernelGGL((pack_weight_cs_kernel<float, _Float16>), grid, block, 0, s, (const float*)w, (_Float16*)packed, K,
                         cin, cout, cob, transpose, flip);
  } else {
    hipLaunchKernelGGL((pack_weight_cs_kernel<uint16_t, uint16_t>), grid, block, 0, s, (const uint16_t*)w, (uint16_t*)packed,
                       K, cin, cout, cob, transpose, flip);
  }
  return launch_status();
}

// forward + dgrad images of an fp32 master weight in one launch; both directions must be this family's shapes
int pack_weight_cs_pair(const float* w, int K, int cin, int cout, int dtype, int flip_dgrad, void* packed_fwd, void* packed_dgrad,
                        hipStream_t s) {
  if (!gather_gemm_cs_supported(cin, cout, K, dtype) || !gather_gemm_cs_supported(cout, cin, K, dtype))
    return WCN_ERROR_UNSUPPORTED_CONFIG;
  const int64_t tot_f = (int64_t)K * ((cin + kCsCIC - 1) / kCsCIC) * kCsCIC * cout;
  const int64_t tot_d = (int64_t)K * ((cout + kCsCIC - 1) / kCsCIC) * kCsCIC * cin;
  const dim3 grid((unsigned)ceil_div(tot_f > tot_d ? tot_f : tot_d, 256), 2), block(256);
  if (dtype == WCN_BF16)
    hipLaunchKernelGGL((pack_weight_cs_pair_kernel<__bf16>), grid, block, 0, s, w, (__bf16*)packed_fwd, (__bf16*)packed_dgrad, K, cin,
                       cout, cs_col_block(cout), cs_col_block(cin), flip_dgrad);
  else
    hipLaunchKernelGGL((pack_weight_cs_pair_kernel<_Float16>), grid, block, 0, s, w, (_Float16*)packed_fwd, (_Float16*)packed_dgrad,
                       K, cin, cout, cs_col_block(cout), cs_col_block(cin), flip_dgrad);
  return launch_status();
}

}  // namespace wcn

#ifdef WCN_PROF
extern "C" int wcn_debug_read_prof_cs(void* dst, size_t bytes) {
  return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(wcn::g_csprof), bytes);
}
extern "C" int wcn_debug_read_time_cs(void* dst, size_t bytes) {
  return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(wcn::g_cstime), bytes);
}
extern "C" int wcn_debug_reset_prof_cs(void) {
  static unsigned long long zeros[2048 * 8];
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(wcn::g_csprof), zeros, sizeof(zeros));
}
#endif
