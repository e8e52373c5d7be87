// pool.hip - sparse pooling over a kernel map (REDUCE_AND_STRIDE and the SparsePool modules).
//
//   wcn_pool_gather  out[m][c] = reduce_k in[tbl[m][k]][c]   reduce in {sum, mean, max, min}; rows of `tbl` with no
//                    neighbour give zero; max / min also return the winning input row (first extremum in offset
//                    order) for the backward pass, and the neighbour count per row on request.
//   wcn_pool_select  dx[n][c] = sum_k [arg[tbl[n][k]][c] == n] * dy[tbl[n][k]][c]   - gradient of max / min pooling,
//                    output-stationary over the REVERSE table (deterministic, no atomics).
//
// Both walk the row-major neighbour table the convolution kernels use ([rows][kp] int32, -1 = absent).  Adjacent lanes
// hold adjacent 16-B pieces of one feature row, so every neighbour access is a whole-row (coalesced) read; fp32
// accumulation.  The reference runs this as to_csr (a device sort) + feature gather + torch_scatter.segment_csr
// (warpconvnet/nn/functional/sparse_pool.py:84-110); here it is one pass over the table with no intermediate tensor.
#include <hip/hip_bf16.h>
#include <hip/hip_fp16.h>

#include <cfloat>

#include "wcn_common.h"

namespace wcn {

enum { kPoolSum = 0, kPoolMean = 1, kPoolMax = 2, kPoolMin = 3 };

template <typename T> struct PoolCvt;
template <> struct PoolCvt<float> {
  static __device__ __forceinline__ float ld(float v) { return v; }
  static __device__ __forceinline__ float st(float v) { return v; }
};
template <> struct PoolCvt<__half> {
  static __device__ __forceinline__ float ld(__half v) { return __half2float(v); }
  static __device__ __forceinline__ __half st(float v) { return __float2half(v); }
};
template <> struct PoolCvt<__hip_bfloat16> {
  static __device__ __forceinline__ float ld(__hip_bfloat16 v) { return __bfloat162float(v); }
  static __device__ __forceinline__ __hip_bfloat16 st(float v) { return __float2bfloat16(v); }
};

template <typename T, int VEC> struct alignas(sizeof(T) * VEC) PoolVec { T v[VEC]; };

template <typename T, int VEC>
__global__ __launch_bounds__(256) void pool_gather_kernel(const T* __restrict__ in, const int32_t* __restrict__ tbl,
                                                          int64_t m, int c, int K, int kp, int op, T* __restrict__ out,
                                                          int32_t* __restrict__ arg, int32_t* __restrict__ count) {
  const int cv = c / VEC;
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= m * cv) return;
  const int64_t row = idx / cv;
  const int v = (int)(idx % cv);
  float acc[VEC];
  int32_t best[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) { acc[j] = 0.f; best[j] = -1; }
  int n = 0;
  const int32_t* trow = tbl + row * kp;
  for (int k = 0; k < K; ++k) {
    const int32_t r = trow[k];
    if (r < 0) continue;
    const PoolVec<T, VEC> x = *reinterpret_cast<const PoolVec<T, VEC>*>(in + (int64_t)r * c + (int64_t)v * VEC);
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      const float f = PoolCvt<T>::ld(x.v[j]);
      if (op == kPoolMax) { if (n == 0 || f > acc[j]) { acc[j] = f; best[j] = r; } }
      else if (op == kPoolMin) { if (n == 0 || f < acc[j]) { acc[j] = f; best[j] = r; } }
      else acc[j] += f;
    }
    ++n;
  }
  PoolVec<T, VEC> y;
  const float scale = (op == kPoolMean && n > 0) ? 1.0f / (float)n : 1.0f;
#pragma unroll
  for (int j = 0; j < VEC; ++j) y.v[j] = PoolCvt<T>::st(acc[j] * scale);
  *reinterpret_cast<PoolVec<T, VEC>*>(out + row * c + (int64_t)v * VEC) = y;
  if (arg) {
#pragma unroll
    for (int j = 0; j < VEC; ++j) arg[row * c + (int64_t)v * VEC + j] = best[j];
  }
  if (count && v == 0) count[row] = n;
}

template <typename T, int VEC>
__global__ __launch_bounds__(256) void pool_select_kernel(const T* __restrict__ dy, const int32_t* __restrict__ arg,
                                                          const int32_t* __restrict__ tbl, int64_t n_rows, int c, int K,
                                                          int kp, T* __restrict__ dx) {
  const int cv = c / VEC;
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n_rows * cv) return;
  const int64_t row = idx / cv;
  const int v = (int)(idx % cv);
  float acc[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) acc[j] = 0.f;
  const int32_t* trow = tbl + row * kp;
  for (int k = 0; k < K; ++k) {
    const int32_t r = trow[k];
    if (r < 0) continue;
    const int64_t at = (int64_t)r * c + (int64_t)v * VEC;
    const PoolVec<T, VEC> g = *reinterpret_cast<const PoolVec<T, VEC>*>(dy + at);
#pragma unroll
    for (int j = 0; j < VEC; ++j)
      if (arg[at + j] == (int32_t)row) acc[j] += PoolCvt<T>::ld(g.v[j]);
  }
  PoolVec<T, VEC> y;
#pragma unroll
  for (int j = 0; j < VEC; ++j) y.v[j] = PoolCvt<T>::st(acc[j]);
  *reinterpret_cast<PoolVec<T, VEC>*>(dx + row * c + (int64_t)v * VEC) = y;
}

template <typename T>
static int pool_gather_t(const void* in, const int32_t* tbl, int64_t m, int c, int K, int kp, int op, void* out,
                         int32_t* arg, int32_t* count, hipStream_t s) {
  constexpr int VEC = 16 / (int)sizeof(T);
  if (c % VEC == 0) {
    hipLaunchKernelGGL((pool_gather_kernel<T, VEC>), dim3((unsigned)ceil_div(m * (c / VEC), 256)), dim3(256), 0, s,
                       (const T*)in, tbl, m, c, K, kp, op, (T*)out, arg, count);
  } else {
    hipLaunchKernelGGL((pool_gather_kernel<T, 1>), dim3((unsigned)ceil_div(m * c, 256)), dim3(256), 0, s, (const T*)in, tbl,
                       m, c, K, kp, op, (T*)out, arg, count);
  }
  return launch_status();
}

template <typename T>
static int pool_select_t(const void* dy, const int32_t* arg, const int32_t* tbl, int64_t n, int c, int K, int kp,
                         void* dx, hipStream_t s) {
  constexpr int VEC = 16 / (int)sizeof(T);
  if (c % VEC == 0) {
    hipLaunchKernelGGL((pool_select_kernel<T, VEC>), dim3((unsigned)ceil_div(n * (c / VEC), 256)), dim3(256), 0, s,
                       (const T*)dy, arg, tbl, n, c, K, kp, (T*)dx);
  } else {
    hipLaunchKernelGGL((pool_select_kernel<T, 1>), dim3((unsigned)ceil_div(n * c, 256)), dim3(256), 0, s, (const T*)dy, arg,
                       tbl, n, c, K, kp, (T*)dx);
  }
  return launch_status();
}

}  // namespace wcn

using namespace wcn;

extern "C" {

int wcn_pool_gather(const void* in, const int32_t* tbl, int64_t n_in, int64_t n_out, int32_t channels, int32_t num_offsets,
                    int32_t dtype, int32_t op, void* out, int32_t* arg, int32_t* count, wcn_stream_t stream) {
  if (n_in < 0 || n_out < 0 || channels < 0 || num_offsets < 1 || num_offsets > 4096 || op < kPoolSum || op > kPoolMin)
    return WCN_ERROR_INVALID_PARAMETERS;
  if (n_out == 0 || channels == 0) return WCN_SUCCESS;
  if (!tbl || !out || (n_in > 0 && !in)) return WCN_ERROR_INVALID_PARAMETERS;
  const int kp = wcn_kmap_row_pitch(num_offsets);
  hipStream_t s = (hipStream_t)stream;
  switch (dtype) {
    case WCN_F32: return pool_gather_t<float>(in, tbl, n_out, channels, num_offsets, kp, op, out, arg, count, s);
    case WCN_F16: return pool_gather_t<__half>(in, tbl, n_out, channels, num_offsets, kp, op, out, arg, count, s);
    case WCN_BF16: return pool_gather_t<__hip_bfloat16>(in, tbl, n_out, channels, num_offsets, kp, op, out, arg, count, s);
    default: return WCN_ERROR_UNSUPPORTED_CONFIG;
  }
}

int wcn_pool_select(const void* dy, const int32_t* arg, const int32_t* tbl, int64_t n_in, int64_t n_out, int32_t channels,
                    int32_t num_offsets, int32_t dtype, void* dx, wcn_stream_t stream) {
  if (n_in < 0 || n_out < 0 || channels < 0 || num_offsets < 1 || num_offsets > 4096) return WCN_ERROR_INVALID_PARAMETERS;
  if (n_in == 0 || channels == 0) return WCN_SUCCESS;
  if (!tbl || !dx || (n_out > 0 && (!dy || !arg))) return WCN_ERROR_INVALID_PARAMETERS;
  const int kp = wcn_kmap_row_pitch(num_offsets);
  hipStream_t s = (hipStream_t)stream;
  switch (dtype) {
    case WCN_F32: return pool_select_t<float>(dy, arg, tbl, n_in, channels, num_offsets, kp, dx, s);
    case WCN_F16: return pool_select_t<__half>(dy, arg, tbl, n_in, channels, num_offsets, kp, dx, s);
    case WCN_BF16: return pool_select_t<__hip_bfloat16>(dy, arg, tbl, n_in, channels, num_offsets, kp, dx, s);
    default: return WCN_ERROR_UNSUPPORTED_CONFIG;
  }
}

}  // extern "C"
