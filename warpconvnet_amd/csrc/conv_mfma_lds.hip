// conv_mfma_lds.hip - fused gather -> LDS -> MFMA -> store kernel for the AB (forward) and ABt (dgrad) sparse-conv GEMMs:
// the variant whose gathered feature rows are STAGED THROUGH LDS with coalesced requests.
//
// conv_mfma.hip lets every lane pull 16-B fragments of its own input row straight into the MFMA B registers: a wave
// instruction then touches 32 different rows, and each 128-B line is visited by four instructions - the texture
// addresser, not HBM, bounds that kernel (16.6 B/clk/CU measured against 57 B/clk/CU for row-shaped requests).  Here
//   * rows land ROW-MAJOR in LDS by LDS-DMA (global_load_lds_dwordx4): 4 adjacent lanes fetch the 64 B (one 32-channel
//     chunk) of one row, so a wave instruction is 16 row-shaped requests and no gather registers exist at all; the 16-B
//     pieces are XOR-swizzled on the SOURCE side (the DMA destination is fixed: lane i -> slot i) so that the B-fragment
//     reads (ds_read_b128, row stride 64 B) are bank-conflict free;
//   * rows that lack the current offset issue NO request (exec-masked DMA) and are zeroed at READ time with a per-lane
//     select on the row's mask bit - no zero page, no LDS clears;
//   * the reduction runs in 32-channel chunks: weight slab 32 x CO (8 KB at CO = 128) + row buffers 4 x 2 x 4 KB + the index
//     slab (256 rows x 28 columns) = 77 KB, TWO workgroups per CU (the 64-channel version needs 128 KB);
//   * everything else as in conv_mfma.hip: 256-row mask-sorted tile per workgroup, wave w owns rows [64w, 64w+64) and all
//     CO channels (transposed MFMA: A = packed weight fragment, B = 32 rows), fp32 accumulators, offsets absent from the
//     tile / wave / 32-row block skipped, index slab staged once, weights by LDS-DMA in fragment order, fused epilogue.
// Covers cin % 32 == 0, CO in {64, 128}, kernel volumes up to 28 offsets (one mask word); other shapes stay on
// conv_mfma.hip.  The packed weight image is the CIC = 32 layout of wcn_pack_weight (mfma_pack_chunk decides, one
// pure function of the shape for the packer and the launcher).
// Reference semantics: warpconvnet/nn/functional/sparse_conv/detail/explicit.py:22-57, 60-92; role of
// _C.mask_gemm.fwd/.dgrad (warpconvnet/csrc/bindings/mask_gemm_bindings.cu:2074-2101).
#include <cstdlib>

#include "wcn_common.h"

namespace wcn {

typedef __attribute__((ext_vector_type(8))) __bf16 l_bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 l_f16x8;
typedef __attribute__((ext_vector_type(16))) float l_f32x16;

template <typename T> struct LFrag;
template <> struct LFrag<__bf16> {
  typedef l_bf16x8 type;
  static __device__ __forceinline__ l_f32x16 mfma(l_bf16x8 a, l_bf16x8 b, l_f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  }
};
template <> struct LFrag<_Float16> {
  typedef l_f16x8 type;
  static __device__ __forceinline__ l_f32x16 mfma(l_f16x8 a, l_f16x8 b, l_f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  }
};

constexpr int kLWaves = 4;
constexpr int kLSlabPitch = 28;  // index slab columns (ints): kernel volumes up to 28, 16-B pieces
constexpr int kLMaxK = 28;

template <typename T, int CO>
struct GGLds {
  static constexpr int CIC = 32, NS = 2, NB = CO / 32, RB = 2, RPW = 64, TILE = kLWaves * RPW;
  static constexpr int W_BYTES = CIC * CO * 2;   // weight slab of one step
  static constexpr int X_BYTES = RPW * CIC * 2;  // one wave's rows of one step (4 KB)
  static constexpr int W_UNITS = W_BYTES / 1024;
  static constexpr size_t OFF_X = 2 * (size_t)W_BYTES;
  static constexpr size_t OFF_NBR = OFF_X + (size_t)kLWaves * 2 * X_BYTES;
  static constexpr size_t OFF_ROWS = OFF_NBR + (size_t)TILE * kLSlabPitch * 4;
  static constexpr size_t LDS_BYTES = OFF_ROWS + (size_t)TILE * 4 + 64;
  typedef typename LFrag<T>::type frag_t;
};

template <typename T, int CO>
__global__ __launch_bounds__(256, 2) void gather_gemm_lds_kernel(const T* __restrict__ in, const T* __restrict__ wp,
                                                                 T* __restrict__ out, const int32_t* __restrict__ nbr,
                                                                 const uint32_t* __restrict__ mask,
                                                                 const int32_t* __restrict__ perm, const ConvEpilogue epi,
                                                                 int64_t n_out, int cin, int K, int kp,
                                                                 float* __restrict__ out32) {
  typedef GGLds<T, CO> G;
  typedef typename G::frag_t frag_t;
  constexpr int NS = G::NS, NB = G::NB, RB = G::RB, RPW = G::RPW, TILE = G::TILE, SP = kLSlabPitch;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* s_w = smem;                                                    // [2][W_BYTES]
  char* s_x = smem + G::OFF_X + (size_t)(threadIdx.x >> 6) * 2 * G::X_BYTES;  // this wave's [2][X_BYTES]
  int32_t* s_nbr = reinterpret_cast<int32_t*>(smem + G::OFF_NBR);     // [TILE][SP]
  int32_t* s_rows = reinterpret_cast<int32_t*>(smem + G::OFF_ROWS);   // [TILE]
  uint32_t* s_wmask = reinterpret_cast<uint32_t*>(s_rows + TILE);     // [kLWaves * RB]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h = lane >> 5, n = lane & 31;
  const int nchunk = cin / G::CIC;
  const int64_t row0 = (int64_t)blockIdx.x * TILE;

  // ---- output row ids (through the mask-sorted permutation), their masks ----
  {
    const int64_t pr = row0 + tid;
    int32_t r = -1;
    if (pr < n_out) r = perm ? perm[pr] : (int32_t)pr;
    s_rows[tid] = r;
  }
  __syncthreads();
  uint32_t my_mask = 0;
  {
    const int32_t r = s_rows[tid];
    if (r >= 0) my_mask = mask[r];  // thread tid stages row tid, which belongs to wave tid / RPW
  }
  {
    // index slab: all row ids first, then all table loads, then all LDS writes (one global round trip)
    constexpr int kVec = SP / 4;  // 16-B pieces per row
    constexpr int kIter = (TILE * kVec + 255) / 256;
    int32_t rr[kIter];
    int4 vv[kIter];
#pragma unroll
    for (int t = 0; t < kIter; ++t) {
      const int e = tid + t * 256;
      rr[t] = (e < TILE * kVec) ? s_rows[e / kVec] : -1;
    }
#pragma unroll
    for (int t = 0; t < kIter; ++t) {
      const int e = tid + t * 256;
      const int c = e % kVec;
      vv[t] = make_int4(-1, -1, -1, -1);
      if (rr[t] >= 0 && c * 4 < kp) {  // read once: non-temporal
        typedef __attribute__((ext_vector_type(4))) int i32x4;
        const i32x4 q = __builtin_nontemporal_load(reinterpret_cast<const i32x4*>(nbr + (int64_t)rr[t] * kp) + c);
        vv[t] = make_int4(q.x, q.y, q.z, q.w);
      }
    }
#pragma unroll
    for (int t = 0; t < kIter; ++t) {
      const int e = tid + t * 256;
      if (e < TILE * kVec) reinterpret_cast<int4*>(s_nbr + (e / kVec) * SP)[e % kVec] = vv[t];
    }
  }
  if (tid < kLWaves * RB) s_wmask[tid] = 0;
  __syncthreads();
  if (my_mask) atomicOr(&s_wmask[tid / 32], my_mask);
  __syncthreads();
  uint32_t rb_mask[RB];
  uint32_t wave_mask = 0u, block_mask = 0u;
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) {
    rb_mask[rb] = __builtin_amdgcn_readfirstlane(s_wmask[wave * RB + rb]);  // wave-uniform: SGPR
    wave_mask |= rb_mask[rb];
  }
#pragma unroll
  for (int q = 0; q < kLWaves * RB; ++q) block_mask |= s_wmask[q];
  block_mask = __builtin_amdgcn_readfirstlane(block_mask);
  // mask of the row this lane holds in the B fragment of row block rb: row wave*64 + rb*32 + n
  uint32_t mrow[RB];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) mrow[rb] = (uint32_t)__shfl((int)my_mask, rb * 32 + n);

  l_f32x16 acc[NB][RB];
#pragma unroll
  for (int b = 0; b < NB; ++b)
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[b][rb][q] = 0.f;

  if (block_mask != 0u) {
    // everything step (k, chunk) needs from HBM: the weight slab (all waves together) and this wave's rows
    auto issue = [&](int buf, int k, int chunk) {
      const char* wsrc = reinterpret_cast<const char*>(wp) + ((size_t)k * nchunk + chunk) * G::W_BYTES;
      char* wdst = s_w + (size_t)buf * G::W_BYTES;
#pragma unroll
      for (int it = 0; it < (G::W_UNITS + kLWaves - 1) / kLWaves; ++it) {
        const int u = it * kLWaves + wave;  // wave-uniform 1-KiB unit
        if (u < G::W_UNITS)
          glds16(wsrc + u * 1024 + lane * 16, __builtin_amdgcn_readfirstlane(lds_addr_of(wdst + u * 1024)));
      }
      if (!((wave_mask >> k) & 1u)) return;
      char* xdst = s_x + (size_t)buf * G::X_BYTES;
#pragma unroll
      for (int it = 0; it < RPW / 16; ++it) {  // 16 rows x 64 B per instruction, 4 lanes per row
        if (!((rb_mask[it >> 1] >> k) & 1u)) continue;  // wave-uniform: no row of this 32-row block has the offset
        const int rl = it * 16 + (lane >> 2);
        const int32_t idx = s_nbr[(wave * RPW + rl) * SP + k];
        if (idx >= 0) {
          const int piece = (lane & 3) ^ ((rl >> 1) & 3);  // source-side swizzle: slot (lane & 3) receives this piece
          glds16(reinterpret_cast<const char*>(in + (int64_t)idx * cin + chunk * G::CIC) + piece * 16,
                 __builtin_amdgcn_readfirstlane(lds_addr_of(xdst + it * 1024)));
        }
      }
    };
    auto compute = [&](int buf, int k) {
      if (!((wave_mask >> k) & 1u)) return;
      const char* wl = s_w + (size_t)buf * G::W_BYTES + lane * 16;
      const char* xl = s_x + (size_t)buf * G::X_BYTES + n * 64;
      const int sw = (n >> 1) & 3;
      bool present[RB];
#pragma unroll
      for (int rb = 0; rb < RB; ++rb) present[rb] = (mrow[rb] >> k) & 1u;
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        frag_t a[NB];
#pragma unroll
        for (int b = 0; b < NB; ++b) a[b] = *reinterpret_cast<const frag_t*>(wl + (b * NS + s) * 1024);
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
          if (!((rb_mask[rb] >> k) & 1u)) continue;  // wave-uniform
          frag_t bf = *reinterpret_cast<const frag_t*>(xl + rb * 32 * 64 + (((2 * h + s) ^ sw) << 4));
          if (!present[rb]) {
#pragma unroll
            for (int q = 0; q < 8; ++q) bf[q] = (T)0.f;
          }
#pragma unroll
          for (int b = 0; b < NB; ++b) acc[b][rb] = LFrag<T>::mfma(a[b], bf, acc[b][rb]);
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
    int kc = -1, cc = 0, buf = 0;
    next_step(kc, cc);
    issue(0, kc, cc);
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0), gfx9 encoding; also resets hipcc's own load scoreboard
    __syncthreads();
    for (;;) {
      int kn = kc, cn = cc;
      const bool more = next_step(kn, cn);
      if (more) issue(buf ^ 1, kn, cn);
      compute(buf, kc);
      __builtin_amdgcn_s_waitcnt(0x0F70);
      __syncthreads();
      if (!more) break;
      kc = kn; cc = cn; buf ^= 1;
    }
  }

  // ---- epilogue: lane (h, n) holds out channels h*CO/2 + 16*b + q of row (rb, n).  Each wave transposes 32 rows at a
  // time through its own LDS stage and writes whole rows with adjacent lanes (full-line, non-temporal writes). ----
  if (out32) {
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
      const int32_t r = s_rows[wave * RPW + rb * 32 + n];
      if (r < 0) continue;
      float* dst = out32 + (int64_t)r * CO + h * (CO / 2);
#pragma unroll
      for (int b = 0; b < NB; ++b) {
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          float4 o = make_float4(acc[b][rb][4 * v + 0], acc[b][rb][4 * v + 1], acc[b][rb][4 * v + 2], acc[b][rb][4 * v + 3]);
          if (epi.bias) {
            const float4 bv = reinterpret_cast<const float4*>(epi.bias + h * (CO / 2) + 16 * b)[v];
            o.x += bv.x; o.y += bv.y; o.z += bv.z; o.w += bv.w;
          }
          reinterpret_cast<float4*>(dst + 16 * b)[v] = o;
        }
      }
    }
    return;
  }
  constexpr int kPitch = CO * 2 + 16;  // bytes; +16 keeps the b128 stage writes conflict-free
  constexpr int kStage = 32 * kPitch;  // one 32-row block per wave
  static_assert((size_t)kLWaves * kStage <= G::OFF_NBR, "the epilogue stage reuses the weight / row buffers");
  char* stage = smem + wave * kStage;  // (the last barrier of the main loop already separates the two uses)
  constexpr int kLanesPerRow = CO / 8;  // 16-B pieces per output row
  constexpr int kRowsPerInstr = 64 / kLanesPerRow;
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) {
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      frag_t lo, hi;
      if (epi.bias) {  // + bias[co] in fp32 before the rounding to the storage dtype
        const float4* bp = reinterpret_cast<const float4*>(epi.bias + h * (CO / 2) + 16 * b);
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          const float4 bv = bp[v];
          acc[b][rb][4 * v + 0] += bv.x; acc[b][rb][4 * v + 1] += bv.y;
          acc[b][rb][4 * v + 2] += bv.z; acc[b][rb][4 * v + 3] += bv.w;
        }
      }
      if (epi.scale) {  // per-channel affine (BatchNorm in inference mode)
        const float4* sp4 = reinterpret_cast<const float4*>(epi.scale + h * (CO / 2) + 16 * b);
        const float4* tp4 = reinterpret_cast<const float4*>(epi.shift + h * (CO / 2) + 16 * b);
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          const float4 sv = sp4[v], tv = tp4[v];
          acc[b][rb][4 * v + 0] = acc[b][rb][4 * v + 0] * sv.x + tv.x; acc[b][rb][4 * v + 1] = acc[b][rb][4 * v + 1] * sv.y + tv.y;
          acc[b][rb][4 * v + 2] = acc[b][rb][4 * v + 2] * sv.z + tv.z; acc[b][rb][4 * v + 3] = acc[b][rb][4 * v + 3] * sv.w + tv.w;
        }
      }
      if (epi.relu && !epi.residual) {  // (with a residual the activation follows the add below)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[b][rb][q] = fmaxf(acc[b][rb][q], 0.f);
      }
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        lo[q] = (T)acc[b][rb][q];
        hi[q] = (T)acc[b][rb][8 + q];
      }
      frag_t* sp = reinterpret_cast<frag_t*>(stage + n * kPitch + (h * (CO / 2) + 16 * b) * 2);
      sp[0] = lo;
      sp[1] = hi;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  // stage is wave-private: LDS ops of one wave execute in order
    const int piece = lane % kLanesPerRow, rsub = lane / kLanesPerRow;
#pragma unroll
    for (int r0 = 0; r0 < 32; r0 += kRowsPerInstr) {
      const int row = r0 + rsub;
      if (rsub < kRowsPerInstr && row < 32) {
        const int32_t rr = s_rows[wave * RPW + rb * 32 + row];
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
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");  // next block overwrites the stage
  }
}

// ---- host side ---------------------------------------------------------------------------------------
static bool lds_variant_enabled() {
  static const int v = [] {
    // opt-in: measured on MI355X (1 M voxels, bf16, round 2) this variant is correct but slower than the register-pipelined
    // kernels (64->128 forward 240 vs 203 us, 128->64 dgrad 329 vs 235 us): with 32-channel steps and two buffers per
    // wave the DMA of step i+1 has one 16-MFMA step to hide ~2 us of memory latency behind the per-step barrier
    const char* e = getenv("WARPCONVNET_AMD_GEMM_LDS");
    return e ? atoi(e) : 0;
  }();
  return v != 0;
}

// Shapes the LDS-staged kernel takes.  ONE pure function of the shape for the weight packer and the launcher.
bool gather_gemm_lds_supported(int cin, int cout, int K, int dtype) {
  if (!lds_variant_enabled()) return false;
  if (dtype != WCN_F16 && dtype != WCN_BF16) return false;
  return cin % 32 == 0 && cin >= 32 && (cout == 64 || cout == 128) && K >= 1 && K <= kLMaxK;
}

template <typename T, int CO>
static int launch_lds(const void* in, const void* wp, void* out, const int32_t* nbr, const uint32_t* mask,
                      const int32_t* perm, const ConvEpilogue& epi, int64_t n_out, int cin, int K, float* out32,
                      hipStream_t s) {
  typedef GGLds<T, CO> G;
  static unsigned long long attr_done = 0ull;  // per device (wcn_common.h)
  const int rc = once_per_device(attr_done, [] {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(gather_gemm_lds_kernel<T, CO>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::LDS_BYTES) == hipSuccess;
  });
  if (rc != WCN_SUCCESS) return rc;
  const unsigned grid = (unsigned)ceil_div(n_out, G::TILE);
  hipLaunchKernelGGL((gather_gemm_lds_kernel<T, CO>), dim3(grid), dim3(256), G::LDS_BYTES, s, (const T*)in, (const T*)wp,
                     (T*)out, nbr, mask, perm, epi, n_out, cin, K, wcn_kmap_row_pitch(K), out32);
  return launch_status();
}

int conv_gather_gemm_lds(const void* in, const void* wp, void* out, const int32_t* nbr, const uint32_t* mask,
                         const int32_t* perm, const ConvEpilogue& epi, int64_t n_out, int cin, int cout, int K, int dtype,
                         float* out32, hipStream_t s) {
  if (!gather_gemm_lds_supported(cin, cout, K, dtype)) return WCN_ERROR_UNSUPPORTED_CONFIG;
  if (dtype == WCN_BF16) {
    if (cout == 64) return launch_lds<__bf16, 64>(in, wp, out, nbr, mask, perm, epi, n_out, cin, K, out32, s);
    return launch_lds<__bf16, 128>(in, wp, out, nbr, mask, perm, epi, n_out, cin, K, out32, s);
  }
  if (cout == 64) return launch_lds<_Float16, 64>(in, wp, out, nbr, mask, perm, epi, n_out, cin, K, out32, s);
  return launch_lds<_Float16, 128>(in, wp, out, nbr, mask, perm, epi, n_out, cin, K, out32, s);
}

}  // namespace wcn
