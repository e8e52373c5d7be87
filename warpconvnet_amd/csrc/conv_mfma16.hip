// conv_mfma16.hip - fused gather -> MFMA -> store kernel for the AB (forward) and ABt (dgrad) sparse-conv GEMMs on the
// 16x16x32 matrix-core shape (v_mfma_f32_16x16x32_{bf16,f16}, gfx950).
//
// Why this shape.  conv_mfma.hip issues v_mfma_f32_32x32x16 transposed (A = weight fragment, B = 32 gathered rows) and
// needs cout in {32, 64, 96, 128, 192, 256}.  With K = 32 per instruction the B operand of the 16x16x32 shape spreads one
// row over four lanes (k-groups of 8 channels) and the output tile of one instruction is 16 channels x 16 rows: any
// cout that is a multiple of 16 fits, the unit a wave can skip shrinks from 32 to 16 rows (fewer matrix-core cycles spent
// on rows that lack the offset), and the accumulator budget scales by rows per wave (64 / 32 / 16 rows for cout up to
// 128 / 256 / 512).  Same math, same operand bytes per flop; measured within a few per cent of the 32x32x16 kernels on the
// shapes both take (1 M voxels, bf16: 64->128 forward 212 vs 200 us, 128->64 dgrad 228 vs 231 us).
//
// Structure (unchanged ideas, see conv_mfma.hip): output-stationary 4-wave workgroup over a tile of the mask-sorted
// row permutation, fp32 accumulators in registers, rows straight HBM -> VGPR (exec-masked for absent neighbours) with
// the next step's rows in flight under the current step's MFMAs, weight slab of the step by LDS-DMA in fragment order
// (double buffered, one barrier per step), index slab staged once per tile, offsets absent from the tile / wave / 16-row
// group skipped, epilogue (bias, BatchNorm-inference affine, residual, ReLU) in fp32, rows transposed through a
// wave-private LDS stage and written as whole lines.
//
// Layout of one MFMA (lane l: g = l >> 4, n = l & 15):  A[i = n][k = 8g + j],  B[k = 8g + j][col n],  D[i = 4g + r][col n].
// Packed weight image (wcn_pack_weight): [offset][chunk][c][cb][lane][j] with
//     ci = chunk*CIC + 32*c + 8*g + j          co = (n >> 2)*(CO/4) + 4*cb + (n & 3)
// so the D fragments of lane (g, n) over cb = 0..CO/16-1 are the CO/4 CONTIGUOUS output channels [g*CO/4, (g+1)*CO/4) of
// output row n.  Channel counts: cin % 32 == 0, cout in {16, 32, 48, 64, 96, 128, 160, 192, 256, 384, 512} (rows per wave
// shrink as cout grows: 64 up to 128 channels, 32 up to 256, 16 above).
// Math: out[r] = sum_k in[nbr[r][k]] . Wp[k]  (fp32 accumulate), Wp = packed image of w (forward), of w^T with k reversed
// (dgrad of a submanifold map), or of w^T (dgrad with a reverse table).
// Reference semantics: warpconvnet/nn/functional/sparse_conv/detail/explicit.py:22-57, 60-92; role of
// _C.mask_gemm.fwd/.dgrad (warpconvnet/csrc/bindings/mask_gemm_bindings.cu:2074-2101).
#include <cstdlib>

#include "wcn_common.h"

namespace wcn {

typedef __attribute__((ext_vector_type(8))) __bf16 m_bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 m_f16x8;
typedef __attribute__((ext_vector_type(4))) float m_f32x4;

template <typename T> struct MFrag;
template <> struct MFrag<__bf16> {
  typedef m_bf16x8 type;
  static __device__ __forceinline__ m_f32x4 mfma(m_bf16x8 a, m_bf16x8 b, m_f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  }
};
template <> struct MFrag<_Float16> {
  typedef m_f16x8 type;
  static __device__ __forceinline__ m_f32x4 mfma(m_f16x8 a, m_f16x8 b, m_f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
  }
};

constexpr int kMWaves = 4;
constexpr int kMMaxKp = 32;    // table columns staged per pass (one mask word)
constexpr int kMMaxK = 1024;   // kernel volumes up to 32 mask words

// ---- weight packing ----------------------------------------------------------------------------------
template <typename TS, typename TD>
__global__ void pack_weight16_kernel(const TS* __restrict__ w, TD* __restrict__ packed, int K, int cin, int cout, int cic,
                                     int transpose, int flip) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t total = (int64_t)K * cin * cout;
  if (e >= total) return;
  const int NC = cic / 32, NCB = cout / 16, nchunk = cin / cic;
  int64_t t = e;
  const int j = (int)(t % 8); t /= 8;
  const int lane = (int)(t % 64); t /= 64;
  const int cb = (int)(t % NCB); t /= NCB;
  const int c = (int)(t % NC); t /= NC;
  const int chunk = (int)(t % nchunk); t /= nchunk;
  const int k = (int)t;
  const int g = lane >> 4, n = lane & 15;
  const int ci = chunk * cic + 32 * c + 8 * g + j;
  const int co = (n >> 2) * (cout / 4) + 4 * cb + (n & 3);
  const int kw = flip ? (K - 1 - k) : k;
  // not transposed: w[kw][ci][co] ([K, cin, cout]); transposed: w is the forward weight [K, cout, cin]
  const int64_t src = transpose ? (((int64_t)kw * cout + co) * cin + ci) : (((int64_t)kw * cin + ci) * cout + co);
  packed[e] = (TD)w[src];
}

// ---- main kernel -------------------------------------------------------------------------------------
template <typename T, int CIC, int CO, int RG>
struct GG16 {
  static constexpr int NC = CIC / 32;         // 32-channel sub-chunks (= MFMA K) per step
  static constexpr int NCB = CO / 16;         // 16-channel output blocks
  static constexpr int RPW = 16 * RG;         // rows per wave
  static constexpr int TILE = kMWaves * RPW;  // rows per workgroup
  static constexpr int SLAB_BYTES = CIC * CO * 2;
  static constexpr int DMA_UNITS = SLAB_BYTES / 1024;  // one wave-instruction of LDS-DMA moves 1 KiB
  static_assert(SLAB_BYTES % 1024 == 0, "weight slab must be a multiple of 1 KiB");
  static constexpr int PITCH = CO * 2 + 16;   // epilogue stage row pitch (bytes)
  static constexpr size_t STAGE_BYTES = (size_t)kMWaves * 16 * PITCH;
  static constexpr size_t OFF_NBR = 2 * (size_t)SLAB_BYTES > STAGE_BYTES ? 2 * (size_t)SLAB_BYTES : STAGE_BYTES;
  static constexpr size_t OFF_ROWS = OFF_NBR + (size_t)TILE * kMMaxKp * 4;
  static constexpr size_t LDS_BYTES = OFF_ROWS + (size_t)TILE * 4 + 128;
  typedef typename MFrag<T>::type frag_t;
};

template <typename T, int CIC, int CO, int RG, bool MULTI>
__global__ __launch_bounds__(256, 2) void gather_gemm16_kernel(const T* __restrict__ in, const T* __restrict__ wp,
                                                               T* __restrict__ out, const int32_t* __restrict__ nbr,
                                                               const uint32_t* __restrict__ mask,
                                                               const int32_t* __restrict__ perm, const ConvEpilogue epi,
                                                               int64_t n_out, int cin, int K, int kp, int mw,
                                                               float* __restrict__ out32, int ilv) {
  typedef GG16<T, CIC, CO, RG> G;
  typedef typename G::frag_t frag_t;
  constexpr int NC = G::NC, NCB = G::NCB, RPW = G::RPW, TILE = G::TILE;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* s_w = smem;                                                   // [2][SLAB_BYTES]; reused as the epilogue stage
  int32_t* s_nbr = reinterpret_cast<int32_t*>(smem + G::OFF_NBR);    // [TILE][kpw]
  int32_t* s_rows = reinterpret_cast<int32_t*>(smem + G::OFF_ROWS);  // [TILE]
  uint32_t* s_gmask = reinterpret_cast<uint32_t*>(s_rows + TILE);    // [kMWaves * RG] OR of the row masks per 16-row group

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, n = lane & 15;
  const int nchunk = cin / CIC;
  const int64_t row0 = (int64_t)blockIdx.x * TILE;
  // 16-row group rg of this wave within the tile.  ilv: the groups of a wave are spread over the (mask-sorted) tile, so
  // every wave sees the same mix of masks and the per-step barriers do not wait for the one wave whose rows are dense.
  auto grp = [&](int rg) { return ilv ? rg * kMWaves + wave : wave * RG + rg; };

  // ---- output row ids (through the mask-sorted permutation) ----
  if (tid < TILE) {
    const int64_t pr = row0 + tid;
    int32_t r = -1;
    if (pr < n_out) r = perm ? perm[pr] : (int32_t)pr;
    s_rows[tid] = r;
  }
  __syncthreads();

  m_f32x4 acc[RG][NCB];
#pragma unroll
  for (int rg = 0; rg < RG; ++rg)
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
      for (int q = 0; q < 4; ++q) acc[rg][cb][q] = 0.f;

  // Kernel volumes above 32 offsets: one pass per 32-bit mask word - the index slab holds the 32 table columns of the
  // current word, the accumulators persist across words.  (K <= 32: a single pass.)
  const int nwords = MULTI ? mw : 1;
  for (int word = 0; word < nwords; ++word) {
    const int kbase = word * 32;
    const int kpw = (kp - kbase) < kMMaxKp ? (kp - kbase) : kMMaxKp;  // table columns staged for this word
    uint32_t my_mask = 0;
    if (tid < TILE) {
      const int32_t r = s_rows[tid];
      if (r >= 0) my_mask = mask[(int64_t)r * mw + word];  // thread tid stages row tid (16-row group tid / 16)
    }
    {
      // index slab: all row ids first, then all table loads, then all LDS writes (one global round trip)
      const int vec_per_row = kpw >> 2;
      constexpr int kIter = (TILE * (kMMaxKp / 4) + 255) / 256;
      int32_t rr[kIter];
      int4 vv[kIter];
#pragma unroll
      for (int t = 0; t < kIter; ++t) {
        const int e = tid + t * 256;
        rr[t] = (e < TILE * vec_per_row) ? s_rows[e / vec_per_row] : -1;
      }
#pragma unroll
      for (int t = 0; t < kIter; ++t) {
        const int e = tid + t * 256;
        const int i = e / vec_per_row, c = e - i * vec_per_row;
        vv[t] = make_int4(-1, -1, -1, -1);
        if (rr[t] >= 0) {  // read once: non-temporal
          typedef __attribute__((ext_vector_type(4))) int i32x4;
          const i32x4 q = __builtin_nontemporal_load(reinterpret_cast<const i32x4*>(nbr + (int64_t)rr[t] * kp + kbase) + c);
          vv[t] = make_int4(q.x, q.y, q.z, q.w);
        }
      }
#pragma unroll
      for (int t = 0; t < kIter; ++t) {
        const int e = tid + t * 256;
        const int i = e / vec_per_row, c = e - i * vec_per_row;
        if (e < TILE * vec_per_row) reinterpret_cast<int4*>(s_nbr + i * kpw)[c] = vv[t];
      }
    }
    if (tid < kMWaves * RG) s_gmask[tid] = 0;
    __syncthreads();
    if (tid < TILE && my_mask) atomicOr(&s_gmask[tid >> 4], my_mask);
    __syncthreads();
    uint32_t rg_mask[RG];
    uint32_t wave_mask = 0u, block_mask = 0u;
#pragma unroll
    for (int rg = 0; rg < RG; ++rg) {
      rg_mask[rg] = __builtin_amdgcn_readfirstlane(s_gmask[grp(rg)]);  // wave-uniform: keep it in an SGPR
      wave_mask |= rg_mask[rg];
    }
#pragma unroll
    for (int q = 0; q < kMWaves * RG; ++q) block_mask |= s_gmask[q];
    block_mask = __builtin_amdgcn_readfirstlane(block_mask);

    if (block_mask != 0u) {
      auto dma_weights = [&](int buf, int k, int chunk) {
        const char* src = reinterpret_cast<const char*>(wp) + ((size_t)(kbase + k) * nchunk + chunk) * G::SLAB_BYTES;
        char* dst = s_w + (size_t)buf * G::SLAB_BYTES;
#pragma unroll
        for (int it = 0; it < (G::DMA_UNITS + kMWaves - 1) / kMWaves; ++it) {
          const int u = it * kMWaves + wave;  // wave-uniform 1-KiB unit
          if (u < G::DMA_UNITS)
            glds16(src + u * 1024 + lane * 16, __builtin_amdgcn_readfirstlane(lds_addr_of(dst + u * 1024)));
        }
      };
      // rows of step (k, chunk): lane (g, n) of row group rg pulls channels [32c + 8g, +8) of row n straight into the B
      // operand - 16 rows per instruction; absent neighbours issue no request.
      // (Measured alternative, round 2: ROW-SHAPED loads - lane l takes piece l & 3 of row l >> 2, four adjacent lanes
      // per 64 contiguous bytes, 3x the address-pipeline rate in tools/gather_probe.hip - plus a 16 x 4 lane transpose with
      // four ds_bpermute per fragment in front of the MFMAs: correct, but 64->128 forward 227 vs 212 us and 128->64 dgrad
      // 284 vs 228 us.  The kernel is not bound by the gather path - with every gather redirected into a 128 KB window
      // its time does not change - so the transposes are pure added work.)
      auto gather = [&](frag_t (&bf)[RG][NC], int k, int chunk) {
        if (!((wave_mask >> k) & 1u)) return;
#pragma unroll
        for (int rg = 0; rg < RG; ++rg) {
          if (RG > 1 && !((rg_mask[rg] >> k) & 1u)) continue;  // wave-uniform: no row of this group has offset k
          const int32_t idx = s_nbr[(grp(rg) * 16 + n) * kpw + k];
#ifdef WCN_ABL_LOCAL
          const T* p = in + (int64_t)(idx & 8191) * cin + chunk * CIC + 8 * g;
#else
          const T* p = in + (int64_t)idx * cin + chunk * CIC + 8 * g;
#endif
#pragma unroll
          for (int c = 0; c < NC; ++c) {
            frag_t v;
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = (T)0.f;
            if (idx >= 0) v = *reinterpret_cast<const frag_t*>(p + 32 * c);
            bf[rg][c] = v;
          }
        }
      };
      auto compute = [&](const frag_t (&bf)[RG][NC], int buf, int k) {
        if (!((wave_mask >> k) & 1u)) return;
        const frag_t* wl = reinterpret_cast<const frag_t*>(s_w + (size_t)buf * G::SLAB_BYTES) + lane;
        // weight fragments of output block cb+1 are read from LDS while the MFMAs of block cb run
#pragma unroll
        for (int c = 0; c < NC; ++c) {
          frag_t bt[RG];
#pragma unroll
          for (int rg = 0; rg < RG; ++rg) bt[rg] = bf[rg][c];
          frag_t a_cur = wl[(c * NCB) * 64], a_nxt = a_cur;
#pragma unroll
          for (int cb = 0; cb < NCB; ++cb) {
            if (cb + 1 < NCB) a_nxt = wl[(c * NCB + cb + 1) * 64];
#pragma unroll
            for (int rg = 0; rg < RG; ++rg) {
              if (RG > 1 && !((rg_mask[rg] >> k) & 1u)) continue;  // wave-uniform
              acc[rg][cb] = MFrag<T>::mfma(a_cur, bt[rg], acc[rg][cb]);
            }
            a_cur = a_nxt;
          }
        }
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
      // LDS-DMA completion is tracked by vmcnt; drained explicitly (builtin: also resets hipcc's own load scoreboard)
      auto sync_step = [&]() {
        __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0), gfx9 encoding
        __syncthreads();
      };
      frag_t B0[RG][NC], B1[RG][NC];
      int k0 = -1, c0 = 0, k1 = -1, c1 = 0;
      next_step(k0, c0);
      dma_weights(0, k0, c0);
      gather(B0, k0, c0);
      sync_step();
      bool more = true;
      while (more) {
        // even half-iteration: compute (k0,c0) from buffer 0 while fetching (k1,c1) into buffer 1
        k1 = k0; c1 = c0;
        const bool has1 = next_step(k1, c1);
        if (has1) { dma_weights(1, k1, c1); gather(B1, k1, c1); }
        compute(B0, 0, k0);
        sync_step();
        if (!has1) break;
        // odd half-iteration
        k0 = k1; c0 = c1;
        const bool has0 = next_step(k0, c0);
        if (has0) { dma_weights(0, k0, c0); gather(B0, k0, c0); }
        compute(B1, 1, k1);
        sync_step();
        more = has0;
      }
    }
    __syncthreads();  // the slab and the group masks are rewritten by the next pass
  }  // word

  // ---- epilogue: lane (g, n) holds out channels [g*CO/4, (g+1)*CO/4) of row (rg, n): channel g*CO/4 + 4*cb + q. ----
  constexpr int CQ = CO / 4;  // channels per lane and row
  if (out32) {
    // fp32 output (the fp32-feature path: fp16 operands, fp32 accumulate, unrounded result): straight from the accumulators
#pragma unroll
    for (int rg = 0; rg < RG; ++rg) {
      const int32_t r = s_rows[grp(rg) * 16 + n];
      if (r < 0) continue;
      float* dst = out32 + (int64_t)r * CO + g * CQ;
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb) {
        float4 o = make_float4(acc[rg][cb][0], acc[rg][cb][1], acc[rg][cb][2], acc[rg][cb][3]);
        if (epi.bias) {
          const float4 bv = reinterpret_cast<const float4*>(epi.bias + g * CQ)[cb];
          o.x += bv.x; o.y += bv.y; o.z += bv.z; o.w += bv.w;
        }
        reinterpret_cast<float4*>(dst)[cb] = o;
      }
    }
    return;
  }
  // Each wave transposes 16 rows at a time through its own LDS stage and writes whole rows with adjacent lanes
  // (full-line, non-temporal writes; the direct store would be one partial-line request per lane).
  constexpr int kPitch = G::PITCH;
  char* stage = smem + wave * 16 * kPitch;  // the weight slabs are dead (the last step ended with a barrier)
  constexpr int kLanesPerRow = CO / 8;       // 16-B pieces per output row
  constexpr int kRowsPerInstr = kLanesPerRow >= 64 ? 1 : 64 / kLanesPerRow;
  constexpr int kPiecesPerLane = kLanesPerRow > 64 ? kLanesPerRow / 64 : 1;
#pragma unroll
  for (int rg = 0; rg < RG; ++rg) {
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) {
      float4 o = make_float4(acc[rg][cb][0], acc[rg][cb][1], acc[rg][cb][2], acc[rg][cb][3]);
      if (epi.bias) {  // + bias[co] in fp32 before the rounding to the storage dtype
        const float4 bv = reinterpret_cast<const float4*>(epi.bias + g * CQ)[cb];
        o.x += bv.x; o.y += bv.y; o.z += bv.z; o.w += bv.w;
      }
      if (epi.scale) {  // per-channel affine (BatchNorm in inference mode)
        const float4 sv = reinterpret_cast<const float4*>(epi.scale + g * CQ)[cb];
        const float4 tv = reinterpret_cast<const float4*>(epi.shift + g * CQ)[cb];
        o.x = o.x * sv.x + tv.x; o.y = o.y * sv.y + tv.y; o.z = o.z * sv.z + tv.z; o.w = o.w * sv.w + tv.w;
      }
      if (epi.relu && !epi.residual) {  // (with a residual the activation follows the add below)
        o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f);
      }
      typedef __attribute__((ext_vector_type(4))) T t4;
      t4 v;
      v[0] = (T)o.x; v[1] = (T)o.y; v[2] = (T)o.z; v[3] = (T)o.w;
      *reinterpret_cast<t4*>(stage + n * kPitch + (g * CQ + 4 * cb) * 2) = v;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  // stage is wave-private: LDS ops of one wave execute in order
#pragma unroll
    for (int pp = 0; pp < kPiecesPerLane; ++pp) {
      const int piece = (lane % (kLanesPerRow < 64 ? kLanesPerRow : 64)) + pp * 64;
      const int rsub = kLanesPerRow < 64 ? lane / kLanesPerRow : 0;
#pragma unroll
      for (int r0 = 0; r0 < 16; r0 += kRowsPerInstr) {
        const int row = r0 + rsub;
        if (rsub < kRowsPerInstr && row < 16 && piece < kLanesPerRow) {
          const int32_t rr = s_rows[grp(rg) * 16 + row];
          if (rr >= 0) {
            frag_t o = *reinterpret_cast<const frag_t*>(stage + row * kPitch + piece * 16);
            if (epi.residual) {  // residual rows are read the way the output is written: whole rows, adjacent lanes
              const frag_t rv = __builtin_nontemporal_load(
                  reinterpret_cast<const frag_t*>(reinterpret_cast<const T*>(epi.residual) + (int64_t)rr * CO + piece * 8));
#pragma unroll
              for (int q = 0; q < 8; ++q) {
                float f = (float)o[q] + (float)rv[q];
                if (epi.relu) f = fmaxf(f, 0.f);
                o[q] = (T)f;
              }
            }
            // streamed once: non-temporal, so the output does not push the gathered input out of the caches
            __builtin_nontemporal_store(o, reinterpret_cast<frag_t*>(out + (int64_t)rr * CO + piece * 8));
          }
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");  // next row group overwrites the stage
  }
}

// ---- host side ---------------------------------------------------------------------------------------
// This family takes the channel shapes the other two gather-GEMM families do not cover (cout 16, 48, 160, 384, 512, ...).
// Measured on MI355X (1 M voxels, bf16, round 2) where both apply: 64->128 forward 212 vs 200 us, 128->64 dgrad 228 vs
// 231 us - a wash, so the older kernels keep their shapes.  (Row-shaped gathers with an LDS or ds_bpermute operand
// transpose in front of the MFMAs were measured too: 248 / 282 and 227 / 284 us - dropped.)
bool mfma32_shape(int cin, int cout);  // conv_mfma.hip

// reduction chunk per step: 64 channels when they divide cin and the two weight slabs stay within 64 KB, else 32
static int chunk16(int cin, int cout) { return (cin % 64 == 0 && 2 * 64 * cout * 2 <= 65536) ? 64 : 32; }

// Shapes this kernel family takes.  ONE pure function of the shape for the weight packer and the launcher.
bool mfma16_supported(int cin, int cout, int K, int dtype) {
  if (mfma32_shape(cin, cout)) return false;
  if (dtype != WCN_F16 && dtype != WCN_BF16) return false;
  if (K < 1 || K > kMMaxK) return false;
  if (cin < 32 || cin % 32 != 0) return false;
  switch (cout) {  // instantiated output widths (every one is a kernel per dtype, chunk size and mask-word mode)
    case 16: case 32: case 48: case 64: case 96: case 128: case 160: case 192: case 256: case 384: case 512: return true;
    default: return false;
  }
}

template <typename T, int CIC, int CO, int RG>
static int launch16(const void* in, const void* wp, void* out, const int32_t* nbr, const uint32_t* mask,
                    const int32_t* perm, const ConvEpilogue& epi, int64_t n_out, int cin, int K, float* out32,
                    hipStream_t s) {
  typedef GG16<T, CIC, CO, RG> G;
  const int kp = wcn_kmap_row_pitch(K), mw = wcn_kmap_mask_words(K);
  static unsigned long long attr_done = 0ull;  // per device (wcn_common.h)
  const int rc = once_per_device(attr_done, [] {
    bool ok = true;
    for (const void* f : {reinterpret_cast<const void*>(gather_gemm16_kernel<T, CIC, CO, RG, false>),
                          reinterpret_cast<const void*>(gather_gemm16_kernel<T, CIC, CO, RG, true>)})
      ok = ok && hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::LDS_BYTES) == hipSuccess;
    return ok;
  });
  if (rc != WCN_SUCCESS) return rc;
  const unsigned grid = (unsigned)ceil_div(n_out, G::TILE);
  if (mw == 1)
    hipLaunchKernelGGL((gather_gemm16_kernel<T, CIC, CO, RG, false>), dim3(grid), dim3(256), G::LDS_BYTES, s,
                       (const T*)in, (const T*)wp, (T*)out, nbr, mask, perm, epi, n_out, cin, K, kp, mw, out32, 1);
  else
    hipLaunchKernelGGL((gather_gemm16_kernel<T, CIC, CO, RG, true>), dim3(grid), dim3(256), G::LDS_BYTES, s,
                       (const T*)in, (const T*)wp, (T*)out, nbr, mask, perm, epi, n_out, cin, K, kp, mw, out32, 1);
  return launch_status();
}

template <typename T, int CIC>
static int dispatch16_co(int cout, const void* in, const void* wp, void* out, const int32_t* nbr, const uint32_t* mask,
                         const int32_t* perm, const ConvEpilogue& epi, int64_t n_out, int cin, int K, float* out32,
                         hipStream_t s) {
#define WCN_CASE16(CO, RG) \
  case CO: return launch16<T, CIC, CO, RG>(in, wp, out, nbr, mask, perm, epi, n_out, cin, K, out32, s)
  switch (cout) {
    // rows per wave = 16 * RG: the accumulators take RG * CO / 4 registers
    WCN_CASE16(16, 4); WCN_CASE16(32, 4); WCN_CASE16(48, 4); WCN_CASE16(64, 4); WCN_CASE16(96, 4); WCN_CASE16(128, 4);
    WCN_CASE16(160, 2); WCN_CASE16(192, 2); WCN_CASE16(256, 2);
    default: return WCN_ERROR_UNSUPPORTED_CONFIG;
  }
#undef WCN_CASE16
}

template <typename T>
static int dispatch16_wide(int cout, const void* in, const void* wp, void* out, const int32_t* nbr, const uint32_t* mask,
                           const int32_t* perm, const ConvEpilogue& epi, int64_t n_out, int cin, int K, float* out32,
                           hipStream_t s) {
#define WCN_CASE16W(CO) \
  case CO: return launch16<T, 32, CO, 1>(in, wp, out, nbr, mask, perm, epi, n_out, cin, K, out32, s)
  switch (cout) {
    WCN_CASE16W(384); WCN_CASE16W(512);
    default: return WCN_ERROR_UNSUPPORTED_CONFIG;
  }
#undef WCN_CASE16W
}

template <typename T>
static int dispatch16(int cin, int cout, const void* in, const void* wp, void* out, const int32_t* nbr, const uint32_t* mask,
                      const int32_t* perm, const ConvEpilogue& epi, int64_t n_out, int K, float* out32, hipStream_t s) {
  if (cout > 256) return dispatch16_wide<T>(cout, in, wp, out, nbr, mask, perm, epi, n_out, cin, K, out32, s);
  if (chunk16(cin, cout) == 64) return dispatch16_co<T, 64>(cout, in, wp, out, nbr, mask, perm, epi, n_out, cin, K, out32, s);
  return dispatch16_co<T, 32>(cout, in, wp, out, nbr, mask, perm, epi, n_out, cin, K, out32, s);
}

int conv_gather_gemm16(const void* in, const void* wp, void* out, const int32_t* nbr, const uint32_t* mask,
                       const int32_t* perm, const ConvEpilogue& epi, int64_t n_out, int cin, int cout, int K, int dtype,
                       float* out32, hipStream_t s) {
  if (!mfma16_supported(cin, cout, K, dtype)) return WCN_ERROR_UNSUPPORTED_CONFIG;
  if (dtype == WCN_BF16) return dispatch16<__bf16>(cin, cout, in, wp, out, nbr, mask, perm, epi, n_out, K, out32, s);
  return dispatch16<_Float16>(cin, cout, in, wp, out, nbr, mask, perm, epi, n_out, K, out32, s);
}

// packed image for this kernel family; `w` fp32 (w_is_f32) or already in the 16-bit storage dtype
int pack_weight16(const void* w, int w_is_f32, int K, int cin, int cout, int dtype, int transpose, int flip, void* packed,
                  hipStream_t s) {
  if (!mfma16_supported(cin, cout, K, dtype)) return WCN_ERROR_UNSUPPORTED_CONFIG;
  const int cic = cout > 256 ? 32 : chunk16(cin, cout);
  const int64_t total = (int64_t)K * cin * cout;
  const dim3 grid((unsigned)ceil_div(total, 256)), block(256);
  if (w_is_f32) {
    if (dtype == WCN_BF16)
      hipLaunchKernelGGL((pack_weight16_kernel<float, __bf16>), grid, block, 0, s, (const float*)w, (__bf16*)packed, K, cin,
                         cout, cic, transpose, flip);
    else
      hipLaunchKernelGGL((pack_weight16_kernel<float, _Float16>), grid, block, 0, s, (const float*)w, (_Float16*)packed, K,
                         cin, cout, cic, transpose, flip);
  } else {  // bf16 and f16 are both 2-byte moves
    hipLaunchKernelGGL((pack_weight16_kernel<uint16_t, uint16_t>), grid, block, 0, s, (const uint16_t*)w, (uint16_t*)packed,
                       K, cin, cout, cic, transpose, flip);
  }
  return launch_status();
}

}  // namespace wcn
