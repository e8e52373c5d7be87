// wgrad_mfma.hip - AtB gather-gather GEMM:  dw[k] = sum_{p in bucket k} x[in_maps[p]]^T . dy[out_maps[p]]   (fp32 out)
//
// Design (gfx950, wave64):
//   * the reduction runs over PAIRS, but gathered rows are contiguous along CHANNELS, so both MFMA
//     operands are "k-strided".  Rows are gathered HBM -> LDS as full 16-B pieces by LDS-DMA
//     (global_load_lds; the LDS image is the plain row-major [pair][channel] tile with a 16-B-chunk XOR
//     swizzle applied on the SOURCE side), and fragments are read back TRANSPOSED with
//     ds_read_b64_tr_b16 (gfx950 hardware transpose read): no register shuffles, no scalar LDS reads.
//   * work split: a fixed grid of G workgroups cuts the global pair list [0, L) into G equal ranges
//     (L is read from offsets[K] on the device - no host sync).  A range may span several offsets; the
//     workgroup flushes its fp32 partial tile to workspace slab (g + k) whenever the offset changes.
//     Slab ids are unique, so a second tiny kernel reduces the slabs of each offset in ascending g:
//     deterministic, no fp32 atomics.
//   * grid.y tiles the (Cin, Cout) plane in CIT x COT blocks; 4 waves = 2 x 2 sub-tiles.
//
// Reference semantics: warpconvnet/nn/functional/sparse_conv/detail/explicit.py:93-97; role of
// _C.mask_gemm.wgrad (warpconvnet/csrc/bindings/mask_gemm_bindings.cu:2103-2116, split-K atomics there).
#include "wcn_common.h"

namespace wcn {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

constexpr int kWgradGrid = 512;   // G: pair ranges (2 workgroups per CU) - the upper bound, see wgrad_ranges()
constexpr int kPairs = 64;        // pairs per pipeline step
constexpr int kZeroPage = 1024;   // bytes reserved at the start of the workspace (kept for layout compatibility)
__device__ uint4 g_wgrad_zero_page[64];  // 1 KiB of zeros: source rows of padded pairs (no per-call memset)

template <typename T> struct WFrag;
template <> struct WFrag<__bf16> {
  static __device__ __forceinline__ f32x16 mfma(s16x8 a, s16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  }
};
template <> struct WFrag<_Float16> {
  static __device__ __forceinline__ f32x16 mfma(s16x8 a, s16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  }
};

// 16-B chunk swizzle (an involution on the chunk index) that spreads the four rows touched by one
// 16-lane transpose-read group over different LDS bank groups.  CH = chunks per row.
template <int CH>
__device__ __forceinline__ int chunk_swizzle(int p) {
  if (CH >= 16) return (p & 3) << 2;
  if (CH == 8) return ((p >> 1) & 1) << 2;
  return 0;
}

// Fragment for MFMA rows/cols [c0, c0+32) and pairs [p0, p0+16): lane (h, m) receives tile[p0+8h+j][c0+m], j<8.
template <int CH>
__device__ __forceinline__ s16x8 read_frag_tr(const char* tile, int p0, int c0, int lane) {
  const int g = lane >> 4, i = lane & 15;
  const int col = c0 + 16 * (g & 1) + 4 * (i & 3);  // 4 consecutive channels, inside one 16-B chunk
  const int chunk = col >> 3, within = (col & 7) * 2;
  s16x8 out;
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int p = p0 + 8 * (g >> 1) + 4 * t + (i >> 2);
    const char* addr = tile + p * (CH * 16) + ((chunk ^ chunk_swizzle<CH>(p)) << 4) + within;
    const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)addr);
    out[4 * t + 0] = v[0]; out[4 * t + 1] = v[1]; out[4 * t + 2] = v[2]; out[4 * t + 3] = v[3];
  }
  return out;
}

// A-operand fragment of a 32-row block whose row 0 is all ones (rows 1..31 zero): MFMA(ones, dy-fragment) puts the
// column sums over the step's pairs into row 0 of the product - the bias gradient falls out of the matrix cores.
template <typename T> __device__ __forceinline__ s16x8 ones_row_frag(int lane) {
  const short one = sizeof(T) == 2 && __is_same(T, __bf16) ? (short)0x3F80 : (short)0x3C00;
  const short v = ((lane & 31) == 0) ? one : (short)0;
  s16x8 f;
#pragma unroll
  for (int q = 0; q < 8; ++q) f[q] = v;
  return f;
}

constexpr int kStages = 3;     // data ring: one stage computing, two in flight / landing
constexpr int kIdxSlots = 4;   // index ring (64 in + 64 out rows per slot)

template <typename T, int CIT, int COT>
struct Wgrad {
  static constexpr int CHX = CIT / 8;   // 16-B chunks per staged x row
  static constexpr int CHY = COT / 8;
  static constexpr int XT_BYTES = kPairs * CIT * 2;
  static constexpr int YT_BYTES = kPairs * COT * 2;
  static constexpr int STAGE_BYTES = XT_BYTES + YT_BYTES;
  static constexpr int X_UNITS = XT_BYTES / 1024, Y_UNITS = YT_BYTES / 1024;  // 1 KiB per wave-instruction
  static constexpr int DATA_OPS = (X_UNITS + Y_UNITS) / 4;                    // DMA instructions per wave per stage
  static constexpr int MBLK = CIT / 32, NBLK = COT / 32;  // 32x32 MFMA blocks of the tile
  static constexpr int BLOCKS = MBLK * NBLK;
  // wave layout: 2 x 2 grid of sub-tiles when both block counts are even (operand fragments are shared across the
  // wave's blocks: MB + NB transpose-reads feed MB * NB MFMAs); otherwise blocks go round robin (w, w+4, ...)
  static constexpr bool GRID = (MBLK % 2 == 0) && (NBLK % 2 == 0);
  static constexpr int MB = GRID ? MBLK / 2 : 1, NB = GRID ? NBLK / 2 : 1;
  static constexpr int PER_WAVE = GRID ? MB * NB : (BLOCKS + 3) / 4;
  static constexpr int IDX_BYTES = kIdxSlots * 2 * kPairs * 4;
  static constexpr size_t LDS_BYTES = (size_t)kStages * STAGE_BYTES + IDX_BYTES;
  static_assert(CIT % 32 == 0 && COT % 32 == 0, "tile must be a multiple of 32 channels");
  static_assert(X_UNITS % 4 == 0 && Y_UNITS % 4 == 0, "every wave must issue the same number of DMA instructions");
};

// Pipeline (per 64-pair step i of a segment; every wave issues the same instruction counts so the counted
// vmcnt below means the same thing in all waves):
//   A  index DMA for step i+3   (2 instructions: 16 in-rows + 16 out-rows per wave, 4 B per lane)
//   B  data DMA for step i+2    (DATA_OPS instructions; row addresses come from the index ring in LDS)
//   C  MFMA on step i           (transpose reads from stage i % 3)
//   D  s_waitcnt vmcnt(DATA_OPS)  -> everything but B of this step has landed (data i+1, indices i+3)
//      s_barrier                  -> ... and is visible to all waves; stage (i % 3) may be overwritten
// CS (bias gradient): while a workgroup streams bucket cs_k - the offset whose pairs are (r, r) for every row r - the
// waves that own ci-block row 0 also multiply a ones-row fragment with the dy fragments they already hold; row 0 of
// that product is sum_pairs dy[out][co].  Partial sums go to cs_slabs[g][cout], reduced in fixed order afterwards.
template <typename T, int CIT, int COT, bool CS>
__global__ __launch_bounds__(256, 2) void wgrad_mfma_kernel(const T* __restrict__ x, const T* __restrict__ dy,
                                                         const int32_t* __restrict__ in_maps,
                                                         const int32_t* __restrict__ out_maps,
                                                         const int32_t* __restrict__ offsets, int K, int cin, int cout,
                                                         const char* __restrict__ zero_page, float* __restrict__ slabs,
                                                         float* __restrict__ cs_slabs, int cs_k) {
  typedef Wgrad<T, CIT, COT> W;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* s_idx = smem + (size_t)kStages * W::STAGE_BYTES;  // [kIdxSlots][2][64] int32
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int G = gridDim.x, g = blockIdx.x;
  const int tiles_co = cout / COT;
  const int ci0 = (blockIdx.y / tiles_co) * CIT, co0 = (blockIdx.y % tiles_co) * COT;

  const int64_t L = offsets[K];
  int64_t Q = (L + G - 1) / G;
  Q = ((Q + kPairs - 1) / kPairs) * kPairs;
  const int64_t r_begin = (int64_t)g * Q;
  const int64_t r_end = (r_begin + Q < L) ? (r_begin + Q) : L;
  if (r_begin >= r_end) return;

  int k = 0;  // first bucket containing r_begin
  {
    int lo = 0, hi = K;
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if ((int64_t)offsets[mid] <= r_begin) lo = mid; else hi = mid;
    }
    k = lo;
  }

  static_assert(!CS || W::GRID, "the fused bias gradient needs the 2 x 2 wave layout");
  f32x16 acc[W::PER_WAVE];
  f32x16 acc1[CS ? W::NB : 1];  // ones-row products (CS only)
  auto zero_acc = [&]() {
#pragma unroll
    for (int j = 0; j < W::PER_WAVE; ++j)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[j][q] = 0.f;
    if (CS) {
#pragma unroll
      for (int b = 0; b < W::NB; ++b)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc1[b][q] = 0.f;
    }
  };
  const bool cs_tile = CS && (blockIdx.y / tiles_co) == 0;  // one ci tile is enough: the sums do not depend on ci
  const s16x8 ones = ones_row_frag<T>(lane);
  auto flush = [&](int kk) {
    float* slab = slabs + (int64_t)(g + kk) * cin * cout;
    const int h = lane >> 5, n = lane & 31;
#pragma unroll
    for (int j = 0; j < W::PER_WAVE; ++j) {
      const int blk = W::GRID ? ((wave >> 1) * W::MB + j / W::NB) * W::NBLK + (wave & 1) * W::NB + j % W::NB
                              : wave + 4 * j;
      if (blk < W::BLOCKS) {
        const int a = blk / W::NBLK, b = blk % W::NBLK;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const int ci = ci0 + a * 32 + (q & 3) + 8 * (q >> 2) + 4 * h;
          const int co = co0 + b * 32 + n;
          slab[(int64_t)ci * cout + co] = acc[j][q];
        }
      }
    }
  };

  // A: indices of step `st` -> index ring.  Wave w moves rows [16w, 16w+16) of in_maps and of out_maps.
  auto dma_idx = [&](int st, int64_t seg_begin, int64_t seg_end) {
    char* slot = s_idx + (size_t)(st % kIdxSlots) * (2 * kPairs * 4);
    int64_t p = seg_begin + (int64_t)st * kPairs + wave * 16 + (lane & 15);
    if (p >= seg_end) p = seg_end - 1;  // padded rows: any valid index (their data comes from the zero page)
    const uint32_t d_in = __builtin_amdgcn_readfirstlane(lds_addr_of(slot + wave * 64));
    const uint32_t d_out = __builtin_amdgcn_readfirstlane(lds_addr_of(slot + kPairs * 4 + wave * 64));
    if (lane < 16) {
      glds4(in_maps + p, d_in);
      glds4(out_maps + p, d_out);
    }
  };
  // B: gathered rows of step `st` -> data ring (row-major [pair][channel] tiles, chunk-swizzled on the source side)
  auto dma_data = [&](int st, int64_t seg_begin, int64_t seg_end) {
    char* xt = smem + (size_t)(st % kStages) * W::STAGE_BYTES;
    char* yt = xt + W::XT_BYTES;
    const int* idx_in = reinterpret_cast<const int*>(s_idx + (size_t)(st % kIdxSlots) * (2 * kPairs * 4));
    const int* idx_out = idx_in + kPairs;
    const int64_t p0 = seg_begin + (int64_t)st * kPairs;
#pragma unroll
    for (int it = 0; it < W::X_UNITS / 4; ++it) {
      const int u = it * 4 + wave;
      const int piece = u * 64 + lane;
      const int p = piece / W::CHX, qs = piece % W::CHX;
      const int q = qs ^ chunk_swizzle<W::CHX>(p);
      const char* src = zero_page;
      if (p0 + p < seg_end) src = reinterpret_cast<const char*>(x + (int64_t)idx_in[p] * cin + ci0) + q * 16;
      glds16(src, __builtin_amdgcn_readfirstlane(lds_addr_of(xt + u * 1024)));
    }
#pragma unroll
    for (int it = 0; it < W::Y_UNITS / 4; ++it) {
      const int u = it * 4 + wave;
      const int piece = u * 64 + lane;
      const int p = piece / W::CHY, qs = piece % W::CHY;
      const int q = qs ^ chunk_swizzle<W::CHY>(p);
      const char* src = zero_page;
      if (p0 + p < seg_end) src = reinterpret_cast<const char*>(dy + (int64_t)idx_out[p] * cout + co0) + q * 16;
      glds16(src, __builtin_amdgcn_readfirstlane(lds_addr_of(yt + u * 1024)));
    }
  };
  auto compute = [&](int st, bool do_cs) {
    const char* xt = smem + (size_t)(st % kStages) * W::STAGE_BYTES;
    const char* yt = xt + W::XT_BYTES;
#pragma unroll
    for (int ks = 0; ks < kPairs / 16; ++ks) {
      if constexpr (W::GRID) {
        s16x8 af[W::MB], bf[W::NB];
#pragma unroll
        for (int a = 0; a < W::MB; ++a) af[a] = read_frag_tr<W::CHX>(xt, ks * 16, ((wave >> 1) * W::MB + a) * 32, lane);
#pragma unroll
        for (int b = 0; b < W::NB; ++b) bf[b] = read_frag_tr<W::CHY>(yt, ks * 16, ((wave & 1) * W::NB + b) * 32, lane);
#pragma unroll
        for (int a = 0; a < W::MB; ++a)
#pragma unroll
          for (int b = 0; b < W::NB; ++b) acc[a * W::NB + b] = WFrag<T>::mfma(af[a], bf[b], acc[a * W::NB + b]);
        if (CS && do_cs && (wave >> 1) == 0) {
#pragma unroll
          for (int b = 0; b < W::NB; ++b) acc1[b] = WFrag<T>::mfma(ones, bf[b], acc1[b]);
        }
      } else {
#pragma unroll
        for (int j = 0; j < W::PER_WAVE; ++j) {
          const int blk = wave + 4 * j;
          if (blk < W::BLOCKS) {  // wave-uniform
            const s16x8 af = read_frag_tr<W::CHX>(xt, ks * 16, (blk / W::NBLK) * 32, lane);
            const s16x8 bf = read_frag_tr<W::CHY>(yt, ks * 16, (blk % W::NBLK) * 32, lane);
            acc[j] = WFrag<T>::mfma(af, bf, acc[j]);
          }
        }
      }
    }
  };
  auto barrier = [&]() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  };

  int64_t pos = r_begin;
  while (pos < r_end) {
    while ((int64_t)offsets[k + 1] <= pos) ++k;  // skip empty buckets
    const int64_t seg_end = ((int64_t)offsets[k + 1] < r_end) ? (int64_t)offsets[k + 1] : r_end;
    const int nsteps = (int)((seg_end - pos + kPairs - 1) / kPairs);
    const bool do_cs = cs_tile && k == cs_k;
    zero_acc();
    // ---- prologue: indices of steps 0..2, data of steps 0..1 ----
    dma_idx(0, pos, seg_end);
    if (nsteps > 1) dma_idx(1, pos, seg_end);
    if (nsteps > 2) dma_idx(2, pos, seg_end);
    wait_vmcnt<0>();
    barrier();
    dma_data(0, pos, seg_end);
    if (nsteps > 1) {
      dma_data(1, pos, seg_end);
      wait_vmcnt<W::DATA_OPS>();
    } else {
      wait_vmcnt<0>();
    }
    barrier();
    // ---- steady state ----
    for (int i = 0; i < nsteps; ++i) {
      if (i + 3 < nsteps) dma_idx(i + 3, pos, seg_end);
      const bool more = i + 2 < nsteps;
      if (more) dma_data(i + 2, pos, seg_end);
      compute(i, do_cs);
      if (more) {
        wait_vmcnt<W::DATA_OPS>();
      } else {
        wait_vmcnt<0>();
      }
      barrier();
    }
    flush(k);
    if (CS && do_cs && (wave >> 1) == 0 && lane < 32) {  // row 0 of the ones-row product: lanes 0..31, register 0
#pragma unroll
      for (int b = 0; b < W::NB; ++b)
        cs_slabs[(int64_t)g * cout + co0 + ((wave & 1) * W::NB + b) * 32 + lane] = acc1[b][0];
    }
    pos = seg_end;
  }
}

// dw[k][e] = sum over ranges g that intersect bucket k (ascending) of slab[g + k][e].
// 16 threads share four consecutive elements (ranges g mod 16), 16-B loads, partial sums combined through LDS in a fixed
// order => deterministic.  (ce is a multiple of 64: channel counts are multiples of 32.)
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ slabs,
                                                           const int32_t* __restrict__ offsets, int K, int64_t ce, int G,
                                                           float* __restrict__ dw, const float* __restrict__ cs_slabs,
                                                           int cs_k, int cout, float* __restrict__ bias_grad) {
  __shared__ float4 s_part[16][16];
  const int part = threadIdx.x >> 4, el = threadIdx.x & 15;
  const int64_t e = (int64_t)blockIdx.x * 64 + el * 4;
  const int k = blockIdx.y;
  const int64_t L = offsets[K];
  int64_t Q = (L + G - 1) / G;
  Q = ((Q + kPairs - 1) / kPairs) * kPairs;
  if (k == K) {
    // extra grid row (bias-gradient launches only): bias_grad[co] = sum over the ranges g that intersect bucket cs_k
    // (ascending) of cs_slabs[g][co]; one wavefront per channel - same launch instead of a ~5 us kernel of its own
    const int co = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (co >= cout) return;
    const int64_t b = offsets[cs_k], en = offsets[cs_k + 1];
    float s = 0.f;
    if (en > b && Q > 0) {
      const int g_lo = (int)(b / Q), g_hi = (int)((en - 1) / Q);
      for (int g = g_lo + lane; g <= g_hi; g += 64) s += cs_slabs[(int64_t)g * cout + co];
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) s += __shfl_xor(s, d);
    if (lane == 0) bias_grad[co] = s;
    return;
  }
  const int64_t b = offsets[k], en = offsets[k + 1];
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  if (e < ce && en > b && Q > 0) {
    const int g_lo = (int)(b / Q), g_hi = (int)((en - 1) / Q);
    for (int g = g_lo + part; g <= g_hi; g += 16) {
      const float4 v = *reinterpret_cast<const float4*>(slabs + (int64_t)(g + k) * ce + e);
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
  }
  s_part[part][el] = s;
  __syncthreads();
  if (part == 0 && e < ce) {
    float4 t = s_part[0][el];
#pragma unroll
    for (int p = 1; p < 16; ++p) {
      const float4 v = s_part[p][el];
      t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w;
    }
    *reinterpret_cast<float4*>(dw + (int64_t)k * ce + e) = t;
  }
}

static int wgrad_tile(int c) {  // largest supported tile dividing the channel count
  if (c % 128 == 0) return 128;
  if (c % 96 == 0) return 96;
  if (c % 64 == 0) return 64;
  if (c % 32 == 0) return 32;
  return 0;
}

bool mfma_wgrad_supported(int cin, int cout, int dtype) {
  if (dtype != WCN_F16 && dtype != WCN_BF16) return false;
  return wgrad_tile(cin) != 0 && wgrad_tile(cout) != 0;
}

bool mfma_wgrad_bias_supported(int cin, int cout, int dtype) {
  if (!mfma_wgrad_supported(cin, cout, dtype)) return false;
  return (wgrad_tile(cin) / 32) % 2 == 0 && (wgrad_tile(cout) / 32) % 2 == 0;  // 2 x 2 wave layout (Wgrad::GRID)
}

size_t wgrad_mfma_workspace(int K, int cin, int cout) {
  // zero page + weight-gradient slabs + bias-gradient partials
  return (size_t)kZeroPage + (size_t)(kWgradGrid + K) * cin * cout * sizeof(float) + (size_t)kWgradGrid * cout * sizeof(float);
}

// Pair ranges of a launch.  Every range flushes one fp32 [CIT, COT] tile per bucket it touches, and the reduce kernel reads them
// all back: G + K slabs of cin * cout * 4 B.  512 ranges cost nothing next to a million gathered rows, but at the coarse levels
// of a U-Net (a few thousand rows, 128 - 256 channels: 141 MB of slabs for ~40 k pairs) the slabs ARE the launch.  So: about
// 512 pairs per range, never fewer workgroups than two per CU (the (cin, cout) tiles on grid.y count), never more than 512.
// `pair_bound`: an upper bound of the pair count known on the host (K * rows), 0 = unknown.
static int wgrad_ranges(int64_t pair_bound, int tiles) {
#ifdef WCN_WGRAD_FIXED_G  // A/B build: 512 ranges whatever the size
  return kWgradGrid;
#endif
  if (pair_bound <= 0) return kWgradGrid;
  int64_t g = (pair_bound + 511) / 512;
  const int64_t g_min = (kWgradGrid + tiles - 1) / tiles;
  if (g < g_min) g = g_min;
  if (g > kWgradGrid) g = kWgradGrid;
  return (int)g;
}

template <typename T, int CIT, int COT>
static int launch_wgrad(const void* x, const void* dy, float* dw, const int32_t* in_maps, const int32_t* out_maps,
                        const int32_t* offsets, int cin, int cout, int K, void* workspace, int cs_k, float* bias_grad,
                        int64_t pair_bound, hipStream_t s) {
  typedef Wgrad<T, CIT, COT> W;
  static unsigned long long attr_done = 0ull;  // per device (wcn_common.h)
  const int rc = once_per_device(attr_done, [] {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_mfma_kernel<T, CIT, COT, false>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)W::LDS_BYTES) != hipSuccess)
      return false;
    if constexpr (W::GRID) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_mfma_kernel<T, CIT, COT, true>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)W::LDS_BYTES) != hipSuccess)
        return false;
    }
    return true;
  });
  if (rc != WCN_SUCCESS) return rc;
  // read-only page of zeros: a per-device constant (never written), its address looked up once per device
  static char* zero_pages[64] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return WCN_ERROR_KERNEL_INITIALIZATION;
  char* zero_page = (dev >= 0 && dev < 64) ? __atomic_load_n(&zero_pages[dev], __ATOMIC_RELAXED) : nullptr;
  if (!zero_page) {
    if (hipGetSymbolAddress((void**)&zero_page, HIP_SYMBOL(g_wgrad_zero_page)) != hipSuccess)
      return WCN_ERROR_KERNEL_INITIALIZATION;
    if (dev >= 0 && dev < 64) __atomic_store_n(&zero_pages[dev], zero_page, __ATOMIC_RELAXED);
  }
  float* slabs = (float*)((char*)workspace + kZeroPage);
  float* cs_slabs = slabs + (size_t)(kWgradGrid + K) * cin * cout;
  const int G = wgrad_ranges(pair_bound, (cin / CIT) * (cout / COT));
  const dim3 grid(G, (cin / CIT) * (cout / COT));
  if (bias_grad) {
    if constexpr (W::GRID) {
      hipLaunchKernelGGL((wgrad_mfma_kernel<T, CIT, COT, true>), grid, dim3(256), W::LDS_BYTES, s, (const T*)x,
                         (const T*)dy, in_maps, out_maps, offsets, K, cin, cout, (const char*)zero_page, slabs, cs_slabs,
                         cs_k);
    } else {
      return WCN_ERROR_UNSUPPORTED_CONFIG;
    }
  } else {
    hipLaunchKernelGGL((wgrad_mfma_kernel<T, CIT, COT, false>), grid, dim3(256), W::LDS_BYTES, s, (const T*)x,
                       (const T*)dy, in_maps, out_maps, offsets, K, cin, cout, (const char*)zero_page, slabs, nullptr, -1);
  }
  const int64_t ce = (int64_t)cin * cout;
  // (the row K of the grid, present with a bias gradient, reduces the column-sum partials: 4 channels per workgroup)
  const unsigned gx = (unsigned)ceil_div(ce, 64);
  if (bias_grad && (int64_t)gx * 4 < cout) return WCN_ERROR_UNSUPPORTED_CONFIG;
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(gx, bias_grad ? K + 1 : K), dim3(256), 0, s, (const float*)slabs, offsets, K, ce,
                     G, dw, (const float*)cs_slabs, cs_k, cout, bias_grad);
  return launch_status();
}

template <typename T, int CIT>
static int dispatch_wgrad_co(int cot, const void* x, const void* dy, float* dw, const int32_t* in_maps,
                             const int32_t* out_maps, const int32_t* offsets, int cin, int cout, int K, void* workspace,
                             int cs_k, float* bias_grad, int64_t pair_bound, hipStream_t s) {
  switch (cot) {
    case 32: return launch_wgrad<T, CIT, 32>(x, dy, dw, in_maps, out_maps, offsets, cin, cout, K, workspace, cs_k, bias_grad, pair_bound, s);
    case 64: return launch_wgrad<T, CIT, 64>(x, dy, dw, in_maps, out_maps, offsets, cin, cout, K, workspace, cs_k, bias_grad, pair_bound, s);
    case 96: return launch_wgrad<T, CIT, 96>(x, dy, dw, in_maps, out_maps, offsets, cin, cout, K, workspace, cs_k, bias_grad, pair_bound, s);
    default: return launch_wgrad<T, CIT, 128>(x, dy, dw, in_maps, out_maps, offsets, cin, cout, K, workspace, cs_k, bias_grad, pair_bound, s);
  }
}

template <typename T>
static int dispatch_wgrad(const void* x, const void* dy, float* dw, const int32_t* in_maps, const int32_t* out_maps,
                          const int32_t* offsets, int cin, int cout, int K, void* workspace, int cs_k, float* bias_grad,
                          int64_t pair_bound, hipStream_t s) {
  const int cot = wgrad_tile(cout);
  int cit = wgrad_tile(cin);
  // 128 input channels: two 64-wide tiles instead of one 128-wide.  dY is then streamed twice (+43 % algorithmic traffic at
  // 128 -> 96), but the 128 x 96 / 128 x 128 tiles spill (22 / 24 VGPRs at the 256-register limit) and run at 0.49 of the
  // roofline against 0.87 for 64 x 128: measured on MinkUNet-14 at 1 M voxels, 128 -> 96: 320 -> 272 us per call,
  // 128 -> 128: 50 -> 31 us.
  if (cit == 128) cit = 64;
  switch (cit) {
    case 32: return dispatch_wgrad_co<T, 32>(cot, x, dy, dw, in_maps, out_maps, offsets, cin, cout, K, workspace, cs_k, bias_grad, pair_bound, s);
    case 64: return dispatch_wgrad_co<T, 64>(cot, x, dy, dw, in_maps, out_maps, offsets, cin, cout, K, workspace, cs_k, bias_grad, pair_bound, s);
    case 96: return dispatch_wgrad_co<T, 96>(cot, x, dy, dw, in_maps, out_maps, offsets, cin, cout, K, workspace, cs_k, bias_grad, pair_bound, s);
    default: return dispatch_wgrad_co<T, 128>(cot, x, dy, dw, in_maps, out_maps, offsets, cin, cout, K, workspace, cs_k, bias_grad, pair_bound, s);
  }
}

int conv_wgrad_mfma(const void* x, const void* dy, float* dw, const int32_t* in_maps, const int32_t* out_maps,
                    const int32_t* offsets, int cin, int cout, int K, int dtype, void* workspace, size_t workspace_bytes,
                    int cs_k, float* bias_grad, int64_t pair_bound, hipStream_t s) {
  if (!mfma_wgrad_supported(cin, cout, dtype)) return WCN_ERROR_UNSUPPORTED_CONFIG;
  if (bias_grad && (!mfma_wgrad_bias_supported(cin, cout, dtype) || cs_k < 0 || cs_k >= K)) return WCN_ERROR_UNSUPPORTED_CONFIG;
  if (!workspace || workspace_bytes < wgrad_mfma_workspace(K, cin, cout)) return WCN_ERROR_INVALID_PARAMETERS;
  if (dtype == WCN_BF16)
    return dispatch_wgrad<__bf16>(x, dy, dw, in_maps, out_maps, offsets, cin, cout, K, workspace, cs_k, bias_grad, pair_bound, s);
  return dispatch_wgrad<_Float16>(x, dy, dw, in_maps, out_maps, offsets, cin, cout, K, workspace, cs_k, bias_grad, pair_bound, s);
}

}  // namespace wcn
