// dwconv.hip - depthwise sparse convolution (weight [K, C], one kernel per channel): HBM-bound
// gather-multiply-accumulate, no matrix cores involved.
//
//   gather : out[r][c] = sum_k in[tbl[r][kt]][c] * w[kw][c]      forward (kt = kw) and dgrad (submanifold: the forward
//            table with kt = K-1-kw; otherwise a reverse table) - output-stationary over the row-major neighbour
//            table, so no atomics and a fixed summation order.
//   wgrad  : dw[k][c] = sum over the pairs p of bucket k of x[in_p][c] * dy[out_p][c]; fixed split of every bucket,
//            partial sums reduced in order => deterministic.
//
// Lane layout of the fast path: C/VEC adjacent lanes cover one feature row with 16-B pieces (whole-row, coalesced
// requests - per-lane scattered accesses are bound by the texture addresser at ~1 lane per clock per CU), 64/(C/VEC)
// rows or pairs per wave.  The row's K neighbour ids are read once, coalesced, and handed out with shuffles.
//
// Reference semantics: warpconvnet/nn/functional/sparse_conv_depth.py:227-306 (explicit depthwise forward/backward);
// role of _C.fma.implicit_fma / implicit_reduction (csrc/implicit_fma_kernel.cu, csrc/implicit_reduction.cu).
#include <hip/hip_bf16.h>
#include <hip/hip_fp16.h>

#include "wcn_common.h"

namespace wcn {

template <typename T> struct DwCvt;
template <> struct DwCvt<float> {
  static __device__ __forceinline__ float ld(float v) { return v; }
  static __device__ __forceinline__ float st(float v) { return v; }
};
template <> struct DwCvt<__half> {
  static __device__ __forceinline__ float ld(__half v) { return __half2float(v); }
  static __device__ __forceinline__ __half st(float v) { return __float2half(v); }
};
template <> struct DwCvt<__hip_bfloat16> {
  static __device__ __forceinline__ float ld(__hip_bfloat16 v) { return __bfloat162float(v); }
  static __device__ __forceinline__ __hip_bfloat16 st(float v) { return __float2bfloat16(v); }
};

constexpr int kDwThreads = 256;
constexpr int kDwMaxLdsWeights = 12 * 1024;  // elements of w kept in LDS by the fast kernels

// ---- fast gather: LPR = C / VEC lanes per row (a power of two <= 64) ----------------------------------------------
template <typename T, int VEC>
__global__ __launch_bounds__(kDwThreads) void dwconv_gather_kernel(const T* __restrict__ in, const T* __restrict__ w,
                                                                   T* __restrict__ out, const int32_t* __restrict__ tbl,
                                                                   const float* __restrict__ bias, int64_t n_out, int C,
                                                                   int K, int kp, int k_flip, int lpr) {
  extern __shared__ __attribute__((aligned(16))) char dw_smem[];
  T* s_w = reinterpret_cast<T*>(dw_smem);  // [K][C]
  for (int e = threadIdx.x; e < K * C; e += kDwThreads) s_w[e] = w[e];
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int piece = lane % lpr, grp = lane / lpr;
  const int rows_per_wave = 64 / lpr;
  const int64_t wave_id = ((int64_t)blockIdx.x * kDwThreads + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * kDwThreads) >> 6;
  for (int64_t r0 = wave_id * rows_per_wave; r0 < n_out; r0 += nwaves * rows_per_wave) {
    const int64_t r = r0 + grp;
    const bool live = r < n_out;
    float acc[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) acc[v] = 0.f;
    for (int j0 = 0; j0 < kp; j0 += lpr) {
      // the row's next `lpr` neighbour ids: one coalesced read per row, then shuffles
      int32_t mine = -1;
      if (live && j0 + piece < kp) mine = tbl[r * kp + j0 + piece];
      for (int t = 0; t < lpr; ++t) {
        const int kt = j0 + t;
        if (kt >= K) break;  // uniform
        const int32_t idx = __shfl(mine, grp * lpr + t);
        if (idx < 0) continue;  // uniform inside the lane group
        const int kw = k_flip ? (K - 1 - kt) : kt;
        T xv[VEC], wv[VEC];
        *reinterpret_cast<uint4*>(xv) = *reinterpret_cast<const uint4*>(in + (int64_t)idx * C + piece * VEC);
        *reinterpret_cast<uint4*>(wv) = *reinterpret_cast<const uint4*>(s_w + kw * C + piece * VEC);
#pragma unroll
        for (int v = 0; v < VEC; ++v) acc[v] += DwCvt<T>::ld(xv[v]) * DwCvt<T>::ld(wv[v]);
      }
    }
    if (live) {
      T ov[VEC];
#pragma unroll
      for (int v = 0; v < VEC; ++v) ov[v] = DwCvt<T>::st(acc[v] + (bias ? bias[piece * VEC + v] : 0.f));
      *reinterpret_cast<uint4*>(out + r * C + piece * VEC) = *reinterpret_cast<const uint4*>(ov);
    }
  }
}

// ---- generic gather: any channel count, one thread per (row, channel) ------------------------------------------------
template <typename T>
__global__ __launch_bounds__(kDwThreads) void dwconv_gather_generic_kernel(const T* __restrict__ in, const T* __restrict__ w,
                                                                           T* __restrict__ out,
                                                                           const int32_t* __restrict__ tbl,
                                                                           const float* __restrict__ bias, int64_t n_out,
                                                                           int C, int K, int kp, int k_flip) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n_out * C) return;
  const int64_t r = e / C;
  const int c = (int)(e % C);
  float acc = 0.f;
  for (int kt = 0; kt < K; ++kt) {
    const int32_t idx = tbl[r * kp + kt];
    if (idx < 0) continue;
    const int kw = k_flip ? (K - 1 - kt) : kt;
    acc += DwCvt<T>::ld(in[(int64_t)idx * C + c]) * DwCvt<T>::ld(w[(int64_t)kw * C + c]);
  }
  out[e] = DwCvt<T>::st(acc + (bias ? bias[c] : 0.f));
}

// ---- wgrad ---------------------------------------------------------------------------------------------------------------
constexpr int kDwSplits = 64;  // fixed number of ranges per bucket

// grid (kDwSplits, K): partial[k][s][c] = sum over pairs of range s of bucket k, pairs ascending per lane group, lane
// groups combined through LDS in group order.
template <typename T, int VEC>
__global__ __launch_bounds__(kDwThreads) void dwconv_wgrad_kernel(const T* __restrict__ x, const T* __restrict__ dy,
                                                                  const int32_t* __restrict__ in_maps,
                                                                  const int32_t* __restrict__ out_maps,
                                                                  const int32_t* __restrict__ offsets, int C, int lpr,
                                                                  float* __restrict__ partial) {
  extern __shared__ __attribute__((aligned(16))) char dw_smem[];
  float* s_acc = reinterpret_cast<float*>(dw_smem);  // [groups][C]
  const int k = blockIdx.y, s = blockIdx.x;
  const int64_t b = offsets[k], e = offsets[k + 1];
  const int64_t len = e - b;
  const int64_t chunk = (len + kDwSplits - 1) / kDwSplits;
  const int64_t p0 = b + (int64_t)s * chunk;
  const int64_t p1 = (p0 + chunk < e) ? (p0 + chunk) : e;
  const int groups = kDwThreads / lpr;
  const int piece = threadIdx.x % lpr, grp = threadIdx.x / lpr;
  float acc[VEC];
#pragma unroll
  for (int v = 0; v < VEC; ++v) acc[v] = 0.f;
  for (int64_t p = p0 + grp; p < p1; p += groups) {
    const int32_t i = in_maps[p], o = out_maps[p];
    T xv[VEC], gv[VEC];
    *reinterpret_cast<uint4*>(xv) = *reinterpret_cast<const uint4*>(x + (int64_t)i * C + piece * VEC);
    *reinterpret_cast<uint4*>(gv) = *reinterpret_cast<const uint4*>(dy + (int64_t)o * C + piece * VEC);
#pragma unroll
    for (int v = 0; v < VEC; ++v) acc[v] += DwCvt<T>::ld(xv[v]) * DwCvt<T>::ld(gv[v]);
  }
#pragma unroll
  for (int v = 0; v < VEC; ++v) s_acc[grp * C + piece * VEC + v] = acc[v];
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += kDwThreads) {
    float t = 0.f;
    for (int g = 0; g < groups; ++g) t += s_acc[g * C + c];
    partial[((int64_t)k * kDwSplits + s) * C + c] = t;
  }
}

template <typename T>
__global__ __launch_bounds__(kDwThreads) void dwconv_wgrad_generic_kernel(const T* __restrict__ x, const T* __restrict__ dy,
                                                                          const int32_t* __restrict__ in_maps,
                                                                          const int32_t* __restrict__ out_maps,
                                                                          const int32_t* __restrict__ offsets, int C,
                                                                          float* __restrict__ partial) {
  // grid (kDwSplits, K); one thread per channel (strided), pairs of the range in ascending order
  const int k = blockIdx.y, s = blockIdx.x;
  const int64_t b = offsets[k], e = offsets[k + 1];
  const int64_t chunk = (e - b + kDwSplits - 1) / kDwSplits;
  const int64_t p0 = b + (int64_t)s * chunk;
  const int64_t p1 = (p0 + chunk < e) ? (p0 + chunk) : e;
  for (int c = threadIdx.x; c < C; c += kDwThreads) {
    float t = 0.f;
    for (int64_t p = p0; p < p1; ++p)
      t += DwCvt<T>::ld(x[(int64_t)in_maps[p] * C + c]) * DwCvt<T>::ld(dy[(int64_t)out_maps[p] * C + c]);
    partial[((int64_t)k * kDwSplits + s) * C + c] = t;
  }
}

__global__ __launch_bounds__(kDwThreads) void dwconv_wgrad_reduce_kernel(const float* __restrict__ partial, int K, int C,
                                                                         float* __restrict__ dw) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (int64_t)K * C) return;
  const int k = (int)(e / C), c = (int)(e % C);
  float t = 0.f;
  for (int s = 0; s < kDwSplits; ++s) t += partial[((int64_t)k * kDwSplits + s) * C + c];
  dw[e] = t;
}

// fast path: 16-B pieces, C/VEC a power of two in [1, 64], weights fit the LDS budget, 16-B aligned rows
template <typename T>
static bool dw_fast_ok(int C, int K, const void* a, const void* b) {
  constexpr int VEC = 16 / sizeof(T);
  if (C % VEC != 0) return false;
  const int lpr = C / VEC;
  if (lpr < 1 || lpr > 64 || (lpr & (lpr - 1)) != 0) return false;
  if ((int64_t)K * C > kDwMaxLdsWeights) return false;
  return ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b)) & 15) == 0;
}

template <typename T>
static int launch_dw_gather(const void* in, const void* w, void* out, const int32_t* tbl, const float* bias, int64_t n_out,
                            int C, int K, int k_flip, hipStream_t s) {
  constexpr int VEC = 16 / sizeof(T);
  const int kp = wcn_kmap_row_pitch(K);
  if (dw_fast_ok<T>(C, K, in, out) && (reinterpret_cast<uintptr_t>(w) & 15) == 0) {
    const int lpr = C / VEC;
    const int64_t rows_per_block = (int64_t)(kDwThreads / 64) * (64 / lpr);
    int64_t blocks = ceil_div(n_out, rows_per_block);
    if (blocks > 8192) blocks = 8192;  // grid-stride: the weight copy into LDS is amortised over several rows
    hipLaunchKernelGGL((dwconv_gather_kernel<T, VEC>), dim3((unsigned)blocks), dim3(kDwThreads),
                       (size_t)K * C * sizeof(T), s, (const T*)in, (const T*)w, (T*)out, tbl, bias, n_out, C, K, kp, k_flip,
                       lpr);
  } else {
    hipLaunchKernelGGL(dwconv_gather_generic_kernel<T>, dim3((unsigned)ceil_div(n_out * C, kDwThreads)), dim3(kDwThreads),
                       0, s, (const T*)in, (const T*)w, (T*)out, tbl, bias, n_out, C, K, kp, k_flip);
  }
  return launch_status();
}

int dwconv_gather(const void* in, const void* w, void* out, const int32_t* tbl, const float* bias, int64_t n_out, int C,
                  int K, int dtype, int k_flip, hipStream_t s) {
  switch (dtype) {
    case WCN_F32: return launch_dw_gather<float>(in, w, out, tbl, bias, n_out, C, K, k_flip, s);
    case WCN_F16: return launch_dw_gather<__half>(in, w, out, tbl, bias, n_out, C, K, k_flip, s);
    case WCN_BF16: return launch_dw_gather<__hip_bfloat16>(in, w, out, tbl, bias, n_out, C, K, k_flip, s);
    default: return WCN_ERROR_UNSUPPORTED_CONFIG;
  }
}

size_t dwconv_wgrad_workspace(int K, int C) { return (size_t)K * kDwSplits * C * sizeof(float); }

template <typename T>
static int launch_dw_wgrad(const void* x, const void* dy, float* dw, const int32_t* in_maps, const int32_t* out_maps,
                           const int32_t* offsets, int C, int K, float* partial, hipStream_t s) {
  constexpr int VEC = 16 / sizeof(T);
  const dim3 grid(kDwSplits, K);
  if (dw_fast_ok<T>(C, 1, x, dy)) {
    const int lpr = C / VEC;
    hipLaunchKernelGGL((dwconv_wgrad_kernel<T, VEC>), grid, dim3(kDwThreads), (size_t)(kDwThreads / lpr) * C * sizeof(float),
                       s, (const T*)x, (const T*)dy, in_maps, out_maps, offsets, C, lpr, partial);
  } else {
    hipLaunchKernelGGL(dwconv_wgrad_generic_kernel<T>, grid, dim3(kDwThreads), 0, s, (const T*)x, (const T*)dy, in_maps,
                       out_maps, offsets, C, partial);
  }
  hipLaunchKernelGGL(dwconv_wgrad_reduce_kernel, dim3((unsigned)ceil_div((int64_t)K * C, kDwThreads)), dim3(kDwThreads), 0, s,
                     (const float*)partial, K, C, dw);
  return launch_status();
}

int dwconv_wgrad(const void* x, const void* dy, float* dw, const int32_t* in_maps, const int32_t* out_maps,
                 const int32_t* offsets, int C, int K, int dtype, void* workspace, size_t workspace_bytes, hipStream_t s) {
  if (!workspace || workspace_bytes < dwconv_wgrad_workspace(K, C)) return WCN_ERROR_INVALID_PARAMETERS;
  float* partial = (float*)workspace;
  switch (dtype) {
    case WCN_F32: return launch_dw_wgrad<float>(x, dy, dw, in_maps, out_maps, offsets, C, K, partial, s);
    case WCN_F16: return launch_dw_wgrad<__half>(x, dy, dw, in_maps, out_maps, offsets, C, K, partial, s);
    case WCN_BF16: return launch_dw_wgrad<__hip_bfloat16>(x, dy, dw, in_maps, out_maps, offsets, C, K, partial, s);
    default: return WCN_ERROR_UNSUPPORTED_CONFIG;
  }
}

}  // namespace wcn
