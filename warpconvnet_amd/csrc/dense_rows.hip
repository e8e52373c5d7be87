// dense_rows.hip - y[N, cout] = x[N, cin] * W (+ bias) for the NARROW 1 x 1 x 1 layers of a network: the stem (3 -> 32) and the
// head (96 -> 20) of a MinkUNet and their input gradients (dy[N, 20] * W^T -> [N, 96]).
//
// The reference computes these as `feats @ weight[0]` (nn/functional/sparse_conv/helper.py:206-213) and leaves the shape to
// the vendor GEMM, which serves a 20- or 3-column operand at 0.09-0.15 of the HBM rate (profiles/r05_unet_1M_roofline.md:
// two `Cijk_*` launches of 205-210 us at 1 M rows for 232 MB of traffic each).  The channel-split gather kernel
// (conv_mfma_cs.hip) streams 1 x 1 x 1 layers at the HBM rate but needs cin >= 64, cin % 32 == 0 and 64 / 96 / 128 output
// columns; this kernel takes everything else up to 128 input and 96 output channels:
//   * one wave per 32 consecutive rows, transposed product (A = 32 output channels of W^T from LDS, B = the 32 rows): a lane
//     requests all of its row's k-slices up front (16-B pieces of one contiguous 32 x cin block), the result tile holds, per
//     lane, four runs of 4 consecutive channels of ONE row -> 8-B stores (cout % 8 == 0: the lane pair (l, l + 32) swaps its
//     middle runs with v_permlane32_swap and stores 16-B pieces), no LDS round trip for the output;
//   * W is read in its storage type (fp32 master weights or T) and layout ([cin, cout], or transposed for the input gradient)
//     and converted while the workgroup builds its fragment image in LDS: no packed image, no cast launch; rows may be fp32
//     (the stem under autocast: rounded to T on the fly, as the cast in front of the vendor GEMM would);
//   * channel counts that are not multiples of 4 (the 3-channel stem) take guarded element loads / stores.
// HBM-bound: cin + cout elements per row.
#include "wcn_common.h"

namespace wcn {

typedef __attribute__((ext_vector_type(8))) __bf16 d_bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 d_f16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 d_bf16x4;
typedef __attribute__((ext_vector_type(4))) _Float16 d_f16x4;
typedef __attribute__((ext_vector_type(16))) float d_f32x16;

template <typename T> struct DFrag;
template <> struct DFrag<__bf16> {
  typedef d_bf16x8 type;
  typedef d_bf16x4 half_type;
  static __device__ __forceinline__ d_f32x16 mfma(d_bf16x8 a, d_bf16x8 b, d_f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  }
};
template <> struct DFrag<_Float16> {
  typedef d_f16x8 type;
  typedef d_f16x4 half_type;
  static __device__ __forceinline__ d_f32x16 mfma(d_f16x8 a, d_f16x8 b, d_f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  }
};

constexpr int kDrMaxSteps = 8;   // k-steps of 16 channels: cin <= 128
constexpr int kDrMaxBlocks = 3;  // blocks of 32 output channels: cout <= 96 (four blocks leave one wave per SIMD: 231 us for
                                 // [1 M, 128] x [128, 128] against the vendor GEMM's 170)
constexpr int kDrThreads = 256;

// 4 consecutive channels c0 .. c0+3 of a row (zero past cin); `vec`: the row pitch and c0 keep the wide load aligned
template <typename T, typename XT>
__device__ __forceinline__ void dr_load4(const XT* __restrict__ row, int c0, int cin, bool vec, T* out) {
  if (vec && c0 + 4 <= cin) {
    if constexpr (sizeof(XT) == 4) {
      const float4 v = *reinterpret_cast<const float4*>(row + c0);
      out[0] = (T)v.x; out[1] = (T)v.y; out[2] = (T)v.z; out[3] = (T)v.w;
    } else {
      typename DFrag<T>::half_type v = *reinterpret_cast<const typename DFrag<T>::half_type*>(row + c0);
      out[0] = v[0]; out[1] = v[1]; out[2] = v[2]; out[3] = v[3];
    }
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i) out[i] = c0 + i < cin ? (T)row[c0 + i] : (T)0.f;
  }
}

// T: arithmetic / output type.  XT: type of the rows (T or float).  WT: type of the weight (float or T).  NB: blocks of 32
// output channels.  `w` is row-major with leading dimension ldw; transposed = 0: W[c][j] (c over cin), 1: W[j][c].
template <typename T, typename XT, typename WT, int NB>
__global__ __launch_bounds__(kDrThreads) void dense_rows_kernel(const XT* __restrict__ x, const WT* __restrict__ w, int ldw,
                                                                 int transposed, const float* __restrict__ bias,
                                                                 T* __restrict__ y, int64_t n, int cin, int cout) {
  typedef typename DFrag<T>::type frag;
  __shared__ frag s_w[NB * kDrMaxSteps * 64];
  const int steps = (cin + 15) >> 4;
  for (int e = threadIdx.x; e < NB * steps * 64; e += kDrThreads) {
    const int l = e & 63, s = (e >> 6) % steps, b = (e >> 6) / steps;
    const int j = b * 32 + (l & 31);
    frag f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int c = 16 * s + 8 * (l >> 5) + i;
      float v = 0.f;
      if (j < cout && c < cin) v = (float)(transposed ? w[(int64_t)j * ldw + c] : w[(int64_t)c * ldw + j]);
      f[i] = (T)v;
    }
    s_w[(b * kDrMaxSteps + s) * 64 + l] = f;
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int half = lane >> 5;
  const bool vec_in = (cin & 3) == 0, vec8 = (cin & 7) == 0, vec_out = (cout & 3) == 0, vec16_out = (cout & 7) == 0;
  float bv[NB][4][4];
#pragma unroll
  for (int b = 0; b < NB; ++b)
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int ch = b * 32 + 8 * g + 4 * half + q;
        bv[b][g][q] = (bias && ch < cout) ? bias[ch] : 0.f;
      }
  const int64_t tiles = (n + 31) >> 5;
  for (int64_t t = (int64_t)blockIdx.x * (kDrThreads / 64) + wave; t < tiles; t += (int64_t)gridDim.x * (kDrThreads / 64)) {
    const int64_t row = t * 32 + (lane & 31);
    const XT* xr = x + (row < n ? row : n - 1) * cin;
    frag xs[kDrMaxSteps];
#pragma unroll
    for (int s = 0; s < kDrMaxSteps; ++s) {
      if (s < steps) {
        const int c0 = 16 * s + 8 * half;
        if (sizeof(XT) == sizeof(T) && vec8 && c0 + 8 <= cin) {
          xs[s] = *reinterpret_cast<const frag*>(xr + c0);  // one 16-B piece
        } else {
          T v[8];
          dr_load4<T, XT>(xr, c0, cin, vec_in, v);
          dr_load4<T, XT>(xr, c0 + 4, cin, vec_in, v + 4);
#pragma unroll
          for (int i = 0; i < 8; ++i) xs[s][i] = v[i];
        }
      }
    }
    d_f32x16 acc[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[b][r] = bv[b][r >> 2][r & 3];
#pragma unroll
    for (int s = 0; s < kDrMaxSteps; ++s) {
      if (s < steps) {
#pragma unroll
        for (int b = 0; b < NB; ++b) acc[b] = DFrag<T>::mfma(s_w[(b * kDrMaxSteps + s) * 64 + lane], xs[s], acc[b]);
      }
    }
    if (vec16_out) {
      // cout % 8 == 0: a lane pair (l, l + 32) holds channels {0-3, 8-11} / {4-7, 12-15} of every 16 of its row; one
      // v_permlane32_swap per register exchanges the middle runs, and each lane stores 8 consecutive channels as one 16-B piece
      T* yr = y + (row < n ? row : n - 1) * cout;
#pragma unroll
      for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int m = 0; m < 2; ++m) {
          typedef __attribute__((ext_vector_type(2))) T pair_t;
          uint32_t ra[2], rb[2];
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            pair_t pa, pb;
            pa[0] = (T)acc[b][8 * m + 2 * q]; pa[1] = (T)acc[b][8 * m + 2 * q + 1];
            pb[0] = (T)acc[b][8 * m + 4 + 2 * q]; pb[1] = (T)acc[b][8 * m + 4 + 2 * q + 1];
            ra[q] = __builtin_bit_cast(uint32_t, pa);
            rb[q] = __builtin_bit_cast(uint32_t, pb);
            const auto sw = __builtin_amdgcn_permlane32_swap(ra[q], rb[q], false, false);  // ra lanes 32-63 <-> rb lanes 0-31
            ra[q] = sw[0];
            rb[q] = sw[1];
          }
          const int ch = b * 32 + 16 * m + 8 * half;
          if (row < n && ch < cout) *reinterpret_cast<uint4*>(yr + ch) = make_uint4(ra[0], ra[1], rb[0], rb[1]);
        }
    } else if (row < n) {
      T* yr = y + row * cout;
#pragma unroll
      for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int ch = b * 32 + 8 * g + 4 * half;  // channels ch .. ch+3 of this row: registers 4g .. 4g+3
          if (ch >= cout) continue;
          if (vec_out && ch + 4 <= cout) {
            typename DFrag<T>::half_type o;
#pragma unroll
            for (int q = 0; q < 4; ++q) o[q] = (T)acc[b][4 * g + q];
            *reinterpret_cast<typename DFrag<T>::half_type*>(yr + ch) = o;
          } else {
#pragma unroll
            for (int q = 0; q < 4; ++q)
              if (ch + q < cout) yr[ch + q] = (T)acc[b][4 * g + q];
          }
        }
    }
  }
}

template <typename T, typename XT, typename WT>
static int launch_dense_rows(const void* x, const void* w, int ldw, int transposed, const float* bias, void* y, int64_t n,
                             int cin, int cout, hipStream_t s) {
  const int nb = (cout + 31) / 32;
  const int64_t tiles = (n + 31) / 32;
  int64_t grid = (tiles + kDrThreads / 64 - 1) / (kDrThreads / 64);
  if (grid > 1024) grid = 1024;  // a workgroup builds its weight image once and keeps it for ~8+ tiles per wave at 1 M rows
#define WCN_DR_LAUNCH(NB)                                                                                                   \
  hipLaunchKernelGGL((dense_rows_kernel<T, XT, WT, NB>), dim3((unsigned)grid), dim3(kDrThreads), 0, s, (const XT*)x,          \
                     (const WT*)w, ldw, transposed, bias, (T*)y, n, cin, cout)
  switch (nb) {
    case 1: WCN_DR_LAUNCH(1); break;
    case 2: WCN_DR_LAUNCH(2); break;
    case 3: WCN_DR_LAUNCH(3); break;
    default: return WCN_ERROR_UNSUPPORTED_CONFIG;
  }
#undef WCN_DR_LAUNCH
  return hipGetLastError() == hipSuccess ? WCN_SUCCESS : WCN_ERROR_KERNEL_EXECUTION;
}

template <typename T>
static int dispatch_dense_rows(const void* x, int x_f32, const void* w, int w_f32, int ldw, int transposed, const float* bias,
                               void* y, int64_t n, int cin, int cout, hipStream_t s) {
  if (x_f32)
    return w_f32 ? launch_dense_rows<T, float, float>(x, w, ldw, transposed, bias, y, n, cin, cout, s)
                 : launch_dense_rows<T, float, T>(x, w, ldw, transposed, bias, y, n, cin, cout, s);
  return w_f32 ? launch_dense_rows<T, T, float>(x, w, ldw, transposed, bias, y, n, cin, cout, s)
               : launch_dense_rows<T, T, T>(x, w, ldw, transposed, bias, y, n, cin, cout, s);
}

}  // namespace wcn

using namespace wcn;

extern "C" {

int wcn_dense_rows_supported(int32_t cin, int32_t cout, int32_t dtype) {
  return (cin >= 1 && cin <= 16 * kDrMaxSteps && cout >= 1 && cout <= 32 * kDrMaxBlocks && (dtype == WCN_F16 || dtype == WCN_BF16))
             ? 1 : 0;
}

int wcn_dense_rows(const void* x, int32_t x_is_f32, const void* w, int32_t w_is_f32, int32_t w_transposed, const float* bias,
                   void* y, int64_t n, int32_t cin, int32_t cout, int32_t dtype, wcn_stream_t stream) {
  if (!wcn_dense_rows_supported(cin, cout, dtype)) return WCN_ERROR_UNSUPPORTED_CONFIG;
  if (n < 0 || !w || (n > 0 && (!x || !y))) return WCN_ERROR_INVALID_PARAMETERS;
  if (n == 0) return WCN_SUCCESS;
  // `w` is the layer's [rows, cols] matrix as stored: [cin, cout], or [cout, cin] when it is applied transposed
  const int ldw = w_transposed ? cin : cout;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == WCN_BF16)
    return dispatch_dense_rows<__bf16>(x, x_is_f32, w, w_is_f32, ldw, w_transposed, bias, y, n, cin, cout, s);
  return dispatch_dense_rows<_Float16>(x, x_is_f32, w, w_is_f32, ldw, w_transposed, bias, y, n, cin, cout, s);
}

}  // extern "C"
