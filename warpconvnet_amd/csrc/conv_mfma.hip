// conv_mfma.hip - fused gather -> MFMA -> store kernel for the AB (forward) and ABt (dgrad) sparse-conv GEMMs.
//
// Design (gfx950, wave64):
//   * output-stationary: a workgroup of 4 waves owns TILE = 4*32*RB output rows (taken through the
//     mask-sorted permutation) and all CO output channels; fp32 accumulators live in registers.
//   * the MFMA is issued "transposed": A operand = weight fragment (M = 32 output channels),
//     B operand = 32 gathered feature rows (N = rows).  With the row/column permutations folded into
//     the PACKED weight image (wcn_pack_weight) every lane (a) gathers CIC/2 CONTIGUOUS channels of one
//     input row straight from HBM into its B registers - no LDS hop, no transpose - and (b) ends up
//     holding CO/2 contiguous output channels of one output row, stored with 16-B writes.
//   * weights: the [CIC x CO] slab of the current (offset, channel-chunk) step is streamed into LDS by
//     LDS-DMA (global_load_lds, 16 B/lane) in exactly the order the A fragments are read back
//     (lane-linear => conflict-free ds_read_b128), double buffered, one barrier per step.  The DMA goes through
//     inline asm and the per-step drain through the s_waitcnt BUILTIN: with the DMA builtin hipcc drains vmcnt(0)
//     in front of the first LDS read of every step (i.e. in front of the MFMAs), and with an asm s_waitcnt its own
//     scoreboard still believes the gathered registers are in flight and waits again - either way gather latency
//     and math serialise.
//   * the tile's neighbour rows (TILE x 32 int32) are staged in LDS once ("index slab").
//   * offsets absent from every row of the workgroup are skipped (bitmask OR), and a wave skips the
//     gather + MFMA of an offset none of its own rows has.
//   * epilogue: + bias in fp32, round, transpose 32 rows at a time through a wave-private LDS stage and write whole
//     rows with adjacent lanes (full-line writes instead of 8 partial-line write requests per 128 B).
//
// Math: out[r] = sum_k in[nbr[r][k]] . Wp[k]  (fp32 accumulate), Wp = packed image of w (forward),
// or of w^T with k reversed (dgrad of a submanifold map), or of w^T (dgrad with a reverse table).
// Reference semantics: warpconvnet/nn/functional/sparse_conv/detail/explicit.py:22-57, 60-92; role of
// _C.mask_gemm.fwd/.dgrad (warpconvnet/csrc/bindings/mask_gemm_bindings.cu:2074-2101).
#include "wcn_common.h"

namespace wcn {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <typename T> struct Frag;
template <> struct Frag<__bf16> {
  typedef bf16x8 type;
  static __device__ __forceinline__ f32x16 mfma(bf16x8 a, bf16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  }
};
template <> struct Frag<_Float16> {
  typedef f16x8 type;
  static __device__ __forceinline__ f32x16 mfma(f16x8 a, f16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  }
};

constexpr int kWaves = 4;
constexpr int kMaxKp = 32;  // table columns staged per pass (one mask word)
constexpr int kMaxK = 1024;  // kernel volumes up to 32 mask words (5^3 = 125 and 7^3 = 343 included)

#ifdef WCN_PROF
// dev-only phase stamps (wall clock, 10 ns ticks): [wg][8] = {entry, after perm, after slab, loop end, end, steps}
__device__ unsigned long long g_prof[8192 * 8];
__device__ unsigned long long g_prof2[8192 * 4];  // per-WG sums (wave 0): issue, compute, vmcnt wait, barrier (clocks)
#define WCN_STAMP(i) do { if (threadIdx.x == 0 && blockIdx.x < 8192) g_prof[blockIdx.x * 8 + (i)] = wall_clock64(); } while (0)
#define WCN_STAMPV(i, v) do { if (threadIdx.x == 0 && blockIdx.x < 8192) g_prof[blockIdx.x * 8 + (i)] = (v); } while (0)
#else
#define WCN_STAMP(i)
#define WCN_STAMPV(i, v)
#endif


// ---- weight packing --------------------------------------------------------------------------------
// packed[k][chunk][b][s][lane][j], lane = (h<<5)|m:
//   ci = chunk*CIC + h*(CIC/2) + 8*s + j
//   co = ((m>>2)&1)*(CO/2) + 16*b + 4*(m>>3) + (m&3)
// so that the C fragment of lane (h', n) holds out channels h'*(CO/2) + 16*b + reg, reg = 0..15.
template <typename T>
__global__ void pack_weight_kernel(const T* __restrict__ w, T* __restrict__ packed, int K, int cin, int cout, int cic,
                                   int transpose, int flip) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t total = (int64_t)K * cin * cout;
  if (e >= total) return;
  const int NS = cic / 16, NB = cout / 32, nchunk = cin / cic;
  int64_t t = e;
  const int j = (int)(t % 8); t /= 8;
  const int lane = (int)(t % 64); t /= 64;
  const int s = (int)(t % NS); t /= NS;
  const int b = (int)(t % NB); t /= NB;
  const int chunk = (int)(t % nchunk); t /= nchunk;
  const int k = (int)t;
  const int h = lane >> 5, m = lane & 31;
  const int ci = chunk * cic + h * (cic / 2) + 8 * s + j;
  const int co = ((m >> 2) & 1) * (cout / 2) + 16 * b + 4 * (m >> 3) + (m & 3);
  const int kw = flip ? (K - 1 - k) : k;
  // not transposed: w[kw][ci][co] ([K, cin, cout]); transposed: w is the forward weight [K, cout, cin]
  const int64_t src = transpose ? (((int64_t)kw * cout + co) * cin + ci) : (((int64_t)kw * cin + ci) * cout + co);
  packed[e] = w[src];
}

// the same image straight from the fp32 master weights (round to nearest even, what `.to(bf16 / f16)` does): one launch
// instead of a cast kernel plus a pack kernel per convolution and direction
template <typename TD>
__global__ void pack_weight_cast_kernel(const float* __restrict__ w, TD* __restrict__ packed, int K, int cin, int cout,
                                        int cic, int transpose, int flip) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t total = (int64_t)K * cin * cout;
  if (e >= total) return;
  const int NS = cic / 16, NB = cout / 32, nchunk = cin / cic;
  int64_t t = e;
  const int j = (int)(t % 8); t /= 8;
  const int lane = (int)(t % 64); t /= 64;
  const int s = (int)(t % NS); t /= NS;
  const int b = (int)(t % NB); t /= NB;
  const int chunk = (int)(t % nchunk); t /= nchunk;
  const int k = (int)t;
  const int h = lane >> 5, m = lane & 31;
  const int ci = chunk * cic + h * (cic / 2) + 8 * s + j;
  const int co = ((m >> 2) & 1) * (cout / 2) + 16 * b + 4 * (m >> 3) + (m & 3);
  const int kw = flip ? (K - 1 - k) : k;
  const int64_t src = transpose ? (((int64_t)kw * cout + co) * cin + ci) : (((int64_t)kw * cin + ci) * cout + co);
  packed[e] = (TD)w[src];
}

// grouped weights [K, G, cin, cout] (forward layout; cin / cout per group) -> G packed images back to back, one launch
template <typename TS, typename TD>
__global__ void pack_weight_grouped_kernel(const TS* __restrict__ w, TD* __restrict__ packed, int K, int groups, int cin,
                                           int cout, int cic, int transpose, int flip) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t per_group = (int64_t)K * cin * cout;
  if (e >= per_group * groups) return;
  const int grp = (int)(e / per_group);
  const int NS = cic / 16, NB = cout / 32, nchunk = cin / cic;
  int64_t t = e - (int64_t)grp * per_group;
  const int j = (int)(t % 8); t /= 8;
  const int lane = (int)(t % 64); t /= 64;
  const int s = (int)(t % NS); t /= NS;
  const int b = (int)(t % NB); t /= NB;
  const int chunk = (int)(t % nchunk); t /= nchunk;
  const int k = (int)t;
  const int h = lane >> 5, m = lane & 31;
  const int ci = chunk * cic + h * (cic / 2) + 8 * s + j;
  const int co = ((m >> 2) & 1) * (cout / 2) + 16 * b + 4 * (m >> 3) + (m & 3);
  const int64_t kg = (int64_t)(flip ? (K - 1 - k) : k) * groups + grp;
  // not transposed: w[k][g][ci][co]; transposed: w is the forward weight [K, G, cout, cin] in kernel-side names
  const int64_t src = transpose ? ((kg * cout + co) * cin + ci) : ((kg * cin + ci) * cout + co);
  packed[e] = (TD)w[src];
}

// ---- main kernel -------------------------------------------------------------------------------------
template <typename T, int CIC, int CO, int RB>
struct GatherGemm {
  static constexpr int NS = CIC / 16;
  static constexpr int NB = CO / 32;
  static constexpr int ROWS_PER_WAVE = 32 * RB;
  static constexpr int TILE = kWaves * ROWS_PER_WAVE;
  static constexpr int SLAB_ELEMS = CIC * CO;
  static constexpr int SLAB_BYTES = SLAB_ELEMS * 2;
  static constexpr int DMA_UNITS = SLAB_BYTES / 1024;  // one wave-instruction of LDS-DMA moves 1 KiB
  static_assert(SLAB_BYTES % 1024 == 0, "weight slab must be a multiple of 1 KiB");
  static constexpr size_t LDS_BYTES = 2 * (size_t)SLAB_BYTES + (size_t)TILE * kMaxKp * 4 + (size_t)TILE * 4 + 64;
  typedef typename Frag<T>::type frag_t;
};

template <typename T, int CIC, int CO, int RB, bool MULTI>
__global__ __launch_bounds__(256, 2) void gather_gemm_mfma_kernel(const T* __restrict__ in, const T* __restrict__ wp,
                                                               T* __restrict__ out, const int32_t* __restrict__ nbr,
                                                               const uint32_t* __restrict__ mask,
                                                               const int32_t* __restrict__ perm,
                                                               ConvEpilogue epi, int64_t n_out, int cin,
                                                               int K, int kp, int mw, float* __restrict__ out32,
                                                               int in_stride, int out_stride) {
  // Channel groups (weight [K, G, Cin/G, Cout/G], reference MaskGemm_forward_64x64x32_1s_flat.h:117-123): ONE launch, the
  // group on grid.y.  `cin` / CO are the PER-GROUP widths, rows are in_stride / out_stride elements apart, group g reads
  // channels [g*cin, (g+1)*cin) and writes [g*CO, (g+1)*CO); the packed weight images of the groups follow each other.
  {
    const int grp = blockIdx.y;
    in += (int64_t)grp * cin;
    wp += (int64_t)grp * K * cin * CO;
    if (out) out += (int64_t)grp * CO;
    if (out32) out32 += (int64_t)grp * CO;
    if (epi.bias) epi.bias += grp * CO;
    if (epi.scale) { epi.scale += grp * CO; epi.shift += grp * CO; }
    if (epi.residual) epi.residual = reinterpret_cast<const T*>(epi.residual) + (int64_t)grp * CO;
  }
  typedef GatherGemm<T, CIC, CO, RB> G;
  typedef typename G::frag_t frag_t;
  constexpr int NS = G::NS, NB = G::NB, RPW = G::ROWS_PER_WAVE, TILE = G::TILE;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  T* s_w = reinterpret_cast<T*>(smem);                                        // [2][SLAB_ELEMS]
  int32_t* s_nbr = reinterpret_cast<int32_t*>(smem + 2 * G::SLAB_BYTES);      // [TILE][kp]
  int32_t* s_rows = s_nbr + TILE * kMaxKp;                                    // [TILE]
  uint32_t* s_wmask = reinterpret_cast<uint32_t*>(s_rows + TILE);             // [4]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h = lane >> 5, n = lane & 31;
  const int nchunk = cin / CIC;
  const int64_t row0 = (int64_t)blockIdx.x * TILE;

  WCN_STAMP(0);
  // ---- stage output row ids ----
  if (tid < TILE) {
    const int64_t pr = row0 + tid;
    int32_t r = -1;
    if (pr < n_out) r = perm ? perm[pr] : (int32_t)pr;
    s_rows[tid] = r;
  }
  __syncthreads();
  WCN_STAMP(1);

  f32x16 acc[NB][RB];
#pragma unroll
  for (int b = 0; b < NB; ++b)
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[b][rb][q] = 0.f;

  // Kernel volumes above 32 offsets: one pass per 32-bit mask word - the index slab holds the 32 table columns of the
  // current word, the accumulators persist across words.  (K <= 32: a single pass, as before.)
  const int nwords = MULTI ? mw : 1;  // compile-time 1 for K <= 32: the accumulators are then not live during staging
  for (int word = 0; word < nwords; ++word) {
  const int kbase = word * 32;
  const int kpw = (kp - kbase) < kMaxKp ? (kp - kbase) : kMaxKp;  // table columns staged for this word
  uint32_t my_mask = 0;
  if (tid < TILE) {
    const int32_t r = s_rows[tid];
    if (r >= 0) my_mask = mask[(int64_t)r * mw + word];  // thread tid stages row tid, which belongs to wave tid / RPW
  }
  {
    // all row ids first, then all table loads, then all LDS writes: written as one loop, every s_rows read is ordered
    // behind the previous s_nbr write (same LDS array) and the global round trips serialise
    const int vec_per_row = kpw >> 2;
    constexpr int kIter = TILE * (kMaxKp / 4) / 256;
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
  // OR-reduce masks: rows of wave w are [w*RPW, (w+1)*RPW)
  // (one word per 32-row block: a wave also skips the MFMAs of a row block that has no row with the offset)
  if (tid < kWaves * RB) s_wmask[tid] = 0;
  __syncthreads();
  if (tid < TILE && my_mask) atomicOr(&s_wmask[tid / 32], my_mask);
  __syncthreads();
  uint32_t rb_mask[RB];
  uint32_t wave_mask = 0u, block_mask = 0u;
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) {
    rb_mask[rb] = __builtin_amdgcn_readfirstlane(s_wmask[wave * RB + rb]);  // wave-uniform: keep it in an SGPR
    wave_mask |= rb_mask[rb];
  }
#pragma unroll
  for (int q = 0; q < kWaves * RB; ++q) block_mask |= s_wmask[q];
  block_mask = __builtin_amdgcn_readfirstlane(block_mask);
  WCN_STAMP(2);
  WCN_STAMPV(5, (unsigned long long)__builtin_popcount(block_mask));

  if (block_mask != 0u) {
    // ---- helpers ----
    auto dma_weights = [&](int buf, int k, int chunk) {
      const T* src = wp + ((int64_t)(kbase + k) * nchunk + chunk) * G::SLAB_ELEMS;
      char* dst = reinterpret_cast<char*>(s_w) + (size_t)buf * G::SLAB_BYTES;
#pragma unroll
      for (int it = 0; it < (G::DMA_UNITS + kWaves - 1) / kWaves; ++it) {
        const int u = it * kWaves + wave;  // wave-uniform 1-KiB unit
        if (u < G::DMA_UNITS)
          glds16(reinterpret_cast<const char*>(src) + u * 1024 + lane * 16,
                 __builtin_amdgcn_readfirstlane(lds_addr_of(dst + u * 1024)));
      }
    };
    // LDS-DMA completion is tracked by vmcnt; drain it explicitly before every barrier.
    auto sync_step = [&]() {
      __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0), gfx9 encoding; also resets hipcc's own load scoreboard
      __syncthreads();
    };
    auto gather = [&](frag_t (&bf)[RB][NS], int k, int chunk) {
      if (!((wave_mask >> k) & 1u)) return;
#pragma unroll
      for (int rb = 0; rb < RB; ++rb) {
        const int i = wave * RPW + rb * 32 + n;
        int32_t idx = s_nbr[i * kpw + k];
#ifdef WCN_ABL_LOCAL
        if (idx >= 0) idx &= 1023;  // dev ablation: all gathers hit a 128 KB window
#endif
        const T* p = in + (int64_t)idx * in_stride + chunk * CIC + h * (CIC / 2);
#pragma unroll
        for (int s = 0; s < NS; ++s) {
          frag_t v;
#pragma unroll
          for (int q = 0; q < 8; ++q) v[q] = (T)0.f;
          if (idx >= 0) v = *reinterpret_cast<const frag_t*>(p + s * 8);
          bf[rb][s] = v;
        }
      }
    };
    auto compute = [&](const frag_t (&bf)[RB][NS], int buf, int k) {
      if (!((wave_mask >> k) & 1u)) return;
      const frag_t* wl = reinterpret_cast<const frag_t*>(reinterpret_cast<const char*>(s_w) + (size_t)buf * G::SLAB_BYTES);
      // the NB weight fragments of channel slice s+1 are read from LDS while the NB*RB MFMAs of slice s run
      frag_t a_cur[NB], a_nxt[NB];
#pragma unroll
      for (int b = 0; b < NB; ++b) a_cur[b] = wl[(b * NS) * 64 + lane];
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        if (s + 1 < NS) {
#pragma unroll
          for (int b = 0; b < NB; ++b) a_nxt[b] = wl[(b * NS + s + 1) * 64 + lane];
        }
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
          if (RB > 1 && !((rb_mask[rb] >> k) & 1u)) continue;  // wave-uniform: no row of this block has offset k
#pragma unroll
          for (int b = 0; b < NB; ++b) acc[b][rb] = Frag<T>::mfma(a_cur[b], bf[rb][s], acc[b][rb]);
        }
#pragma unroll
        for (int b = 0; b < NB; ++b) a_cur[b] = a_nxt[b];
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

#ifdef WCN_PROF
    unsigned long long pq[4] = {0, 0, 0, 0};
#endif
    frag_t B0[RB][NS], B1[RB][NS];
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
#ifdef WCN_PROF
      const unsigned long long q0 = clock64();
#endif
      if (has1) { dma_weights(1, k1, c1); gather(B1, k1, c1); }
#ifdef WCN_PROF
      const unsigned long long q1 = clock64();
#endif
      compute(B0, 0, k0);
#ifdef WCN_PROF
      const unsigned long long q2 = clock64();
#endif
      __builtin_amdgcn_s_waitcnt(0x0F70);
#ifdef WCN_PROF
      const unsigned long long q3 = clock64();
#endif
      __syncthreads();
#ifdef WCN_PROF
      const unsigned long long q4 = clock64();
      pq[0] += q1 - q0; pq[1] += q2 - q1; pq[2] += q3 - q2; pq[3] += q4 - q3;
#endif
      if (!has1) break;
      // odd half-iteration
      k0 = k1; c0 = c1;
      const bool has0 = next_step(k0, c0);
      if (has0) { dma_weights(0, k0, c0); gather(B0, k0, c0); }
      compute(B1, 1, k1);
      sync_step();
      more = has0;
    }
#ifdef WCN_PROF
    if (tid == 0 && blockIdx.x < 8192)
      for (int q = 0; q < 4; ++q) g_prof2[blockIdx.x * 4 + q] = 2 * pq[q];  // only even half-steps are timed
#endif
  }
  __syncthreads();  // the slab and the mask words are rewritten by the next pass
  }  // word

  WCN_STAMP(3);
  // ---- epilogue: lane (h, n) holds out channels h*CO/2 + 16*b + q of row (rb, n).  Storing that straight to HBM
  // makes every lane write 16-B pieces of its own row (8 partial-line write requests per 128-B line); instead each
  // wave transposes 32 rows at a time through its own LDS stage and writes whole rows with adjacent lanes. ----
  constexpr int kPitch = CO * 2 + 16;               // bytes; +16 keeps the b128 stage writes conflict-free
  constexpr int kStage = 32 * kPitch;               // one 32-row block per wave
  constexpr bool kStaged = (size_t)kWaves * kStage <= 2 * (size_t)G::SLAB_BYTES + (size_t)TILE * kMaxKp * 4;
  if (out32) {
    // fp32 output (the fp32-feature path: fp16 operands, fp32 accumulate, unrounded result): every lane stores its
    // 16 contiguous channels per block straight from the accumulators
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
      const int32_t r = s_rows[wave * RPW + rb * 32 + n];
      if (r < 0) continue;
      float* dst = out32 + (int64_t)r * out_stride + h * (CO / 2);
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
  if (kStaged) __syncthreads();  // the weight / index slabs are dead from here on: reuse them as the stage
  char* stage = smem + wave * kStage;
  constexpr int kLanesPerRow = CO / 8;              // 16-B pieces per output row
  constexpr int kRowsPerInstr = 64 / kLanesPerRow;
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) {
    const int i = wave * RPW + rb * 32 + n;
    const int32_t r = s_rows[i];
    T* dst = out + (int64_t)r * out_stride + h * (CO / 2);
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      frag_t lo, hi;
      if (epi.bias) {  // fused epilogue: + bias[co] in fp32 before the rounding to the storage dtype
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
      if (!kStaged && epi.residual && r >= 0) {  // direct path: the lane owns 16 contiguous channels of its row
        const T* rp = reinterpret_cast<const T*>(epi.residual) + (int64_t)r * out_stride + h * (CO / 2) + 16 * b;
        const frag_t r0 = *reinterpret_cast<const frag_t*>(rp), r1 = *reinterpret_cast<const frag_t*>(rp + 8);
#pragma unroll
        for (int q = 0; q < 8; ++q) { acc[b][rb][q] += (float)r0[q]; acc[b][rb][8 + q] += (float)r1[q]; }
      }
      if (epi.relu && !(kStaged && epi.residual)) {  // (with a staged residual the activation follows the add below)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[b][rb][q] = fmaxf(acc[b][rb][q], 0.f);
      }
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        lo[q] = (T)acc[b][rb][q];
        hi[q] = (T)acc[b][rb][8 + q];
      }
      if (kStaged) {
        frag_t* sp = reinterpret_cast<frag_t*>(stage + n * kPitch + (h * (CO / 2) + 16 * b) * 2);
        sp[0] = lo;
        sp[1] = hi;
      } else if (r >= 0) {
        *reinterpret_cast<frag_t*>(dst + 16 * b) = lo;
        *reinterpret_cast<frag_t*>(dst + 16 * b + 8) = hi;
      }
    }
    if (kStaged) {
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
                  reinterpret_cast<const frag_t*>(reinterpret_cast<const T*>(epi.residual) + (int64_t)rr * out_stride + piece * 8));
#pragma unroll
              for (int q = 0; q < 8; ++q) {
                float f = (float)o[q] + (float)rv[q];
                if (epi.relu) f = fmaxf(f, 0.f);
                o[q] = (T)f;
              }
            }
            // streamed once: non-temporal, so the output does not push the gathered input out of the caches
            __builtin_nontemporal_store(o, reinterpret_cast<frag_t*>(out + (int64_t)rr * out_stride + piece * 8));
          }
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");  // next block overwrites the stage
    }
  }
#ifdef WCN_PROF
  wait_vmcnt<0>();
  WCN_STAMP(4);
#endif
}

template <typename T, int CIC, int CO, int RB>
static int launch_gather_gemm(const void* in, const void* wp, void* out, const int32_t* nbr, const uint32_t* mask,
                              const int32_t* perm, const ConvEpilogue& epi, int64_t n_out, int cin, int K, float* out32,
                              hipStream_t s, int groups = 1) {
  typedef GatherGemm<T, CIC, CO, RB> G;
  const int kp = wcn_kmap_row_pitch(K);
  const int mw = wcn_kmap_mask_words(K);
  static unsigned long long attr_done = 0ull;  // per device (wcn_common.h)
  const int rc = once_per_device(attr_done, [] {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(gather_gemm_mfma_kernel<T, CIC, CO, RB, false>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::LDS_BYTES) == hipSuccess &&
           hipFuncSetAttribute(reinterpret_cast<const void*>(gather_gemm_mfma_kernel<T, CIC, CO, RB, true>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::LDS_BYTES) == hipSuccess;
  });
  if (rc != WCN_SUCCESS) return rc;
  const dim3 grid((unsigned)ceil_div(n_out, G::TILE), (unsigned)groups);
  const int in_stride = cin * groups, out_stride = CO * groups;
  if (mw == 1)
    hipLaunchKernelGGL((gather_gemm_mfma_kernel<T, CIC, CO, RB, false>), grid, dim3(256), G::LDS_BYTES, s, (const T*)in,
                       (const T*)wp, (T*)out, nbr, mask, perm, epi, n_out, cin, K, kp, mw, out32, in_stride, out_stride);
  else
    hipLaunchKernelGGL((gather_gemm_mfma_kernel<T, CIC, CO, RB, true>), grid, dim3(256), G::LDS_BYTES, s, (const T*)in,
                       (const T*)wp, (T*)out, nbr, mask, perm, epi, n_out, cin, K, kp, mw, out32, in_stride, out_stride);
  return launch_status();
}

// Output widths 96 / 128 run 32 rows per wave (128-row tiles, 64 / 48 accumulator registers, three workgroups per CU
// instead of two): 4-6 % faster than 64 rows per wave on both scene types (round 2).  Width 64 keeps 64 rows per wave.
template <typename T, int CIC>
static int dispatch_co(int cout, const void* in, const void* wp, void* out, const int32_t* nbr, const uint32_t* mask,
                       const int32_t* perm, const ConvEpilogue& epi, int64_t n_out, int cin, int K, float* out32,
                       hipStream_t s, int groups) {
  switch (cout) {
    case 32: return launch_gather_gemm<T, CIC, 32, 2>(in, wp, out, nbr, mask, perm, epi, n_out, cin, K, out32, s, groups);
    case 64: return launch_gather_gemm<T, CIC, 64, 2>(in, wp, out, nbr, mask, perm, epi, n_out, cin, K, out32, s, groups);
    case 96: return launch_gather_gemm<T, CIC, 96, 1>(in, wp, out, nbr, mask, perm, epi, n_out, cin, K, out32, s, groups);
    case 128: return launch_gather_gemm<T, CIC, 128, 1>(in, wp, out, nbr, mask, perm, epi, n_out, cin, K, out32, s, groups);
    case 192: return launch_gather_gemm<T, CIC, 192, 1>(in, wp, out, nbr, mask, perm, epi, n_out, cin, K, out32, s, groups);
    case 256: return launch_gather_gemm<T, CIC, 256, 1>(in, wp, out, nbr, mask, perm, epi, n_out, cin, K, out32, s, groups);
    default: return WCN_ERROR_UNSUPPORTED_CONFIG;
  }
}

// conv_mfma_cs.hip: channel-split family, gathered rows staged through LDS (round 3)
bool gather_gemm_cs_supported(int cin, int cout, int K, int dtype);
int conv_gather_gemm_cs(const void* in, const void* wp, void* out, const int32_t* nbr, const uint32_t* mask,
                        const int32_t* perm, const ConvEpilogue& epi, int64_t n_out, int cin, int cout, int K, int dtype,
                        float* out32, hipStream_t s);
int pack_weight_cs(const void* w, int w_is_f32, int K, int cin, int cout, int dtype, int transpose, int flip, void* packed,
                   hipStream_t s);

// conv_mfma16.hip
bool mfma16_supported(int cin, int cout, int K, int dtype);
int conv_gather_gemm16(const void* in, const void* wp, void* out, const int32_t* nbr, const uint32_t* mask,
                       const int32_t* perm, const ConvEpilogue& epi, int64_t n_out, int cin, int cout, int K, int dtype,
                       float* out32, hipStream_t s);
int pack_weight16(const void* w, int w_is_f32, int K, int cin, int cout, int dtype, int transpose, int flip, void* packed,
                  hipStream_t s);

// reduction chunk per step: 64 channels when they divide cin (measured optimum between loads in flight per wave and steps
// per tile: a 128 -> 64 dgrad with one 128-channel step per offset 317 us, with 32-channel steps 314 us, vs 265 us)
int mfma_chunk_for(int cin, int cout, int K) {
  (void)cout; (void)K;
  if (cin % 64 == 0) return 64;
  if (cin % 32 == 0) return 32;
  if (cin % 16 == 0) return 16;
  return 0;
}

// channel shapes of the 32x32x16 kernels in this file
bool mfma32_shape(int cin, int cout) {
  if (mfma_chunk_for(cin, cout, 1) == 0) return false;
  return cout == 32 || cout == 64 || cout == 96 || cout == 128 || cout == 192 || cout == 256;
}

bool mfma_gather_supported(int cin, int cout, int K, int dtype) {
  if (gather_gemm_cs_supported(cin, cout, K, dtype)) return true;  // (incl. outputs in column blocks: 320 = 5 x 64, 384, 512 ...)
  if (mfma16_supported(cin, cout, K, dtype)) return true;
  if (dtype != WCN_F16 && dtype != WCN_BF16) return false;
  if (K < 1 || K > kMaxK) return false;
  return mfma32_shape(cin, cout);
}

template <typename T>
static int dispatch_cic(int cin, int cout, const void* in, const void* wp, void* out, const int32_t* nbr,
                        const uint32_t* mask, const int32_t* perm, const ConvEpilogue& epi, int64_t n_out, int K, float* out32,
                        hipStream_t s, int groups = 1) {
  switch (mfma_chunk_for(cin, cout, K)) {
    case 64: return dispatch_co<T, 64>(cout, in, wp, out, nbr, mask, perm, epi, n_out, cin, K, out32, s, groups);
    case 32: return dispatch_co<T, 32>(cout, in, wp, out, nbr, mask, perm, epi, n_out, cin, K, out32, s, groups);
    case 16: return dispatch_co<T, 16>(cout, in, wp, out, nbr, mask, perm, epi, n_out, cin, K, out32, s, groups);
    default: return WCN_ERROR_UNSUPPORTED_CONFIG;
  }
}

// channel groups in ONE launch: per-group widths cin x cout, the 32x32x16 kernels (group index on grid.y)
bool mfma_grouped_supported(int cin, int cout, int K, int dtype) {
  return (dtype == WCN_F16 || dtype == WCN_BF16) && K >= 1 && K <= kMaxK && mfma32_shape(cin, cout);
}
int conv_gather_gemm_grouped(const void* in, const void* wp, void* out, const int32_t* nbr, const uint32_t* mask,
                             const int32_t* perm, const ConvEpilogue& epi, int64_t n_out, int cin, int cout, int groups,
                             int K, int dtype, hipStream_t s) {
  if (groups < 1 || groups > 65535 || !mfma_grouped_supported(cin, cout, K, dtype)) return WCN_ERROR_UNSUPPORTED_CONFIG;
  if (dtype == WCN_BF16)
    return dispatch_cic<__bf16>(cin, cout, in, wp, out, nbr, mask, perm, epi, n_out, K, nullptr, s, groups);
  return dispatch_cic<_Float16>(cin, cout, in, wp, out, nbr, mask, perm, epi, n_out, K, nullptr, s, groups);
}
// packed images of all groups for the kernels above (always the 32x32x16 layout), one launch
int pack_weight_grouped(const void* w, int w_is_f32, int K, int groups, int cin, int cout, int dtype, int transpose, int flip,
                        void* packed, hipStream_t s);

int conv_gather_gemm_mfma(const void* in, const void* wp, void* out, const int32_t* nbr, const uint32_t* mask,
                          const int32_t* perm, const ConvEpilogue& epi, int64_t n_out, int cin, int cout, int K, int dtype,
                          float* out32, hipStream_t s) {
  if (!mfma_gather_supported(cin, cout, K, dtype)) return WCN_ERROR_UNSUPPORTED_CONFIG;
  if (gather_gemm_cs_supported(cin, cout, K, dtype))  // channel-split family (conv_mfma_cs.hip)
    return conv_gather_gemm_cs(in, wp, out, nbr, mask, perm, epi, n_out, cin, cout, K, dtype, out32, s);
  if (mfma16_supported(cin, cout, K, dtype))  // 16x16x32 shape: row-shaped gathers (conv_mfma16.hip)
    return conv_gather_gemm16(in, wp, out, nbr, mask, perm, epi, n_out, cin, cout, K, dtype, out32, s);
  if (dtype == WCN_BF16) return dispatch_cic<__bf16>(cin, cout, in, wp, out, nbr, mask, perm, epi, n_out, K, out32, s);
  return dispatch_cic<_Float16>(cin, cout, in, wp, out, nbr, mask, perm, epi, n_out, K, out32, s);
}

int pack_weight_mfma_f32(const float* w, int K, int cin, int cout, int dtype, int transpose, int flip, void* packed,
                         hipStream_t s) {
  // the layout of the packed image follows the kernel that will consume it (a pure function of the shape)
  if (gather_gemm_cs_supported(cin, cout, K, dtype)) return pack_weight_cs(w, 1, K, cin, cout, dtype, transpose, flip, packed, s);
  if (mfma16_supported(cin, cout, K, dtype)) return pack_weight16(w, 1, K, cin, cout, dtype, transpose, flip, packed, s);
  const int cic = mfma_chunk_for(cin, cout, K);
  if (cic == 0 || cout % 32 != 0 || (dtype != WCN_F16 && dtype != WCN_BF16)) return WCN_ERROR_UNSUPPORTED_CONFIG;
  const int64_t total = (int64_t)K * cin * cout;
  const dim3 grid((unsigned)ceil_div(total, 256)), block(256);
  if (dtype == WCN_BF16)
    hipLaunchKernelGGL(pack_weight_cast_kernel<__bf16>, grid, block, 0, s, w, (__bf16*)packed, K, cin, cout, cic, transpose, flip);
  else
    hipLaunchKernelGGL(pack_weight_cast_kernel<_Float16>, grid, block, 0, s, w, (_Float16*)packed, K, cin, cout, cic,
                       transpose, flip);
  return launch_status();
}

int pack_weight_mfma(const void* w, int K, int cin, int cout, int dtype, int transpose, int flip, void* packed,
                     hipStream_t s) {
  if (gather_gemm_cs_supported(cin, cout, K, dtype)) return pack_weight_cs(w, 0, K, cin, cout, dtype, transpose, flip, packed, s);
  if (mfma16_supported(cin, cout, K, dtype)) return pack_weight16(w, 0, K, cin, cout, dtype, transpose, flip, packed, s);
  const int cic = mfma_chunk_for(cin, cout, K);
  if (cic == 0 || cout % 32 != 0 || (dtype != WCN_F16 && dtype != WCN_BF16)) return WCN_ERROR_UNSUPPORTED_CONFIG;
  const int64_t total = (int64_t)K * cin * cout;
  // bf16 and f16 are both 2-byte moves
  hipLaunchKernelGGL(pack_weight_kernel<uint16_t>, dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, s,
                     (const uint16_t*)w, (uint16_t*)packed, K, cin, cout, cic, transpose, flip);
  return launch_status();
}

int pack_weight_grouped(const void* w, int w_is_f32, int K, int groups, int cin, int cout, int dtype, int transpose, int flip,
                        void* packed, hipStream_t s) {
  const int cic = mfma_chunk_for(cin, cout, K);
  if (groups < 1 || cic == 0 || !mfma32_shape(cin, cout) || (dtype != WCN_F16 && dtype != WCN_BF16))
    return WCN_ERROR_UNSUPPORTED_CONFIG;
  const int64_t total = (int64_t)groups * K * cin * cout;
  const dim3 grid((unsigned)ceil_div(total, 256)), block(256);
  if (w_is_f32) {
    if (dtype == WCN_BF16)
      hipLaunchKernelGGL((pack_weight_grouped_kernel<float, __bf16>), grid, block, 0, s, (const float*)w, (__bf16*)packed, K,
                         groups, cin, cout, cic, transpose, flip);
    else
      hipLaunchKernelGGL((pack_weight_grouped_kernel<float, _Float16>), grid, block, 0, s, (const float*)w, (_Float16*)packed,
                         K, groups, cin, cout, cic, transpose, flip);
  } else {
    hipLaunchKernelGGL((pack_weight_grouped_kernel<uint16_t, uint16_t>), grid, block, 0, s, (const uint16_t*)w,
                       (uint16_t*)packed, K, groups, cin, cout, cic, transpose, flip);
  }
  return launch_status();
}

#ifdef WCN_PROF
}  // namespace wcn
extern "C" int wcn_debug_read_prof2(void* dst, size_t bytes) {
  return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(wcn::g_prof2), bytes);
}
extern "C" int wcn_debug_read_prof(void* dst, size_t bytes) {
  return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(wcn::g_prof), bytes);
}
namespace wcn {
#endif
}  // namespace wcn

