// conv_ref.hip - straightforward HIP kernels for the three sparse-conv GEMMs (any channel count,
// f32/f16/bf16 storage, fp32 accumulation).  They are the bring-up / odd-shape path ("hip_ref"); the
// production path is conv_mfma.hip.  Semantics: reference explicit gather-matmul-scatter
// (warpconvnet/nn/functional/sparse_conv/detail/explicit.py:22-101), restated over the row-major
// neighbour table so no atomics are needed for forward/dgrad.
#include <hip/hip_bf16.h>
#include <hip/hip_fp16.h>

#include "wcn_common.h"

namespace wcn {

template <typename T> struct Cvt;
template <> struct Cvt<float> {
  static __device__ __forceinline__ float ld(const float* p) { return *p; }
  static __device__ __forceinline__ void st(float* p, float v) { *p = v; }
};
template <> struct Cvt<__half> {
  static __device__ __forceinline__ float ld(const __half* p) { return __half2float(*p); }
  static __device__ __forceinline__ void st(__half* p, float v) { *p = __float2half(v); }
};
template <> struct Cvt<__hip_bfloat16> {
  static __device__ __forceinline__ float ld(const __hip_bfloat16* p) { return __bfloat162float(*p); }
  static __device__ __forceinline__ void st(__hip_bfloat16* p, float v) { *p = __float2bfloat16(v); }
};

// out[r][co] = sum_k sum_ci in[tbl[r][kt]][ci] * W(kw, ci, co); one thread per (row, co).
template <typename T>
__global__ __launch_bounds__(256) void gather_gemm_ref_kernel(const T* __restrict__ in, const T* __restrict__ w,
                                                              T* __restrict__ out, const int32_t* __restrict__ nbr,
                                                              const float* __restrict__ bias, int64_t n_out, int cin,
                                                              int cout, int K, int kp, int w_transposed, int k_flip) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n_out * cout) return;
  const int64_t r = e / cout;
  const int co = (int)(e % cout);
  const int32_t* row_tbl = nbr + r * kp;
  float acc = 0.f;
  for (int kw = 0; kw < K; ++kw) {
    const int kt = k_flip ? (K - 1 - kw) : kw;
    const int idx = row_tbl[kt];
    if (idx < 0) continue;
    const T* xin = in + (int64_t)idx * cin;
    if (!w_transposed) {
      const T* wk = w + (int64_t)kw * cin * cout + co;  // w[kw][ci][co]
      for (int ci = 0; ci < cin; ++ci) acc += Cvt<T>::ld(xin + ci) * Cvt<T>::ld(wk + (int64_t)ci * cout);
    } else {
      const T* wk = w + ((int64_t)kw * cout + co) * cin;  // w_fwd[kw][co][ci]
      for (int ci = 0; ci < cin; ++ci) acc += Cvt<T>::ld(xin + ci) * Cvt<T>::ld(wk + ci);
    }
  }
  if (bias) acc += bias[co];
  Cvt<T>::st(out + e, acc);
}

// colsum[c] = sum_r in[r][c] (fp32): bias gradient.  Two deterministic passes: per-workgroup partial sums over a
// contiguous row range, then a fixed-order reduce.  VEC channels per thread (16-B loads for 2-byte types), lanes
// run along channels => every wave reads whole rows.
template <typename T, int VEC>
__global__ __launch_bounds__(256) void colsum_partial_kernel(const T* __restrict__ in, int64_t n, int c,
                                                             float* __restrict__ partial) {
  __shared__ float s_sum[256 * VEC];
  const int tid = threadIdx.x;
  const int cgroups = (c + VEC - 1) / VEC;                 // channel groups per row
  const int lanes_c = cgroups < 256 ? cgroups : 256;      // threads along channels
  const int rsteps = 256 / lanes_c;                       // rows handled concurrently
  const int cc = tid % lanes_c, rr = tid / lanes_c;
  const int64_t rows_per_block = (n + gridDim.x - 1) / gridDim.x;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
  const int64_t r1 = (r0 + rows_per_block < n) ? (r0 + rows_per_block) : n;
  for (int g0 = 0; g0 < cgroups; g0 += lanes_c) {
    const int ch0 = (g0 + cc) * VEC;
    float acc[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) acc[v] = 0.f;
    if (rr < rsteps && g0 + cc < cgroups) {
      for (int64_t r = r0 + rr; r < r1; r += rsteps) {
        const T* p = in + r * c + ch0;
        if (VEC > 1) {
          T tmp[VEC];
          *reinterpret_cast<uint4*>(tmp) = *reinterpret_cast<const uint4*>(p);  // VEC * sizeof(T) == 16
#pragma unroll
          for (int v = 0; v < VEC; ++v) acc[v] += Cvt<T>::ld(tmp + v);
        } else {
          acc[0] += Cvt<T>::ld(p);
        }
      }
    }
#pragma unroll
    for (int v = 0; v < VEC; ++v) s_sum[tid * VEC + v] = acc[v];
    __syncthreads();
    if (rr == 0 && g0 + cc < cgroups) {
#pragma unroll
      for (int v = 0; v < VEC; ++v) {
        float t = 0.f;
        for (int q = 0; q < rsteps; ++q) t += s_sum[(q * lanes_c + cc) * VEC + v];
        if (ch0 + v < c) partial[(int64_t)blockIdx.x * c + ch0 + v] = t;
      }
    }
    __syncthreads();
  }
}

// one wavefront per channel: lane l adds partials l, l+64, ... then a fixed butterfly => deterministic
__global__ __launch_bounds__(64) void colsum_final_kernel(const float* __restrict__ partial, int nblocks, int c,
                                                          float* __restrict__ out) {
  const int ch = blockIdx.x, lane = threadIdx.x;
  float t = 0.f;
  for (int b = lane; b < nblocks; b += 64) t += partial[(int64_t)b * c + ch];
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) t += __shfl_xor(t, d);
  if (lane == 0) out[ch] = t;
}

// dw[k][ci][co] = sum over pairs of bucket k (ascending) of x[in][ci] * dy[out][co]; one thread per element.
template <typename T>
__global__ __launch_bounds__(256) void wgrad_ref_kernel(const T* __restrict__ x, const T* __restrict__ dy,
                                                        float* __restrict__ dw, const int32_t* __restrict__ in_maps,
                                                        const int32_t* __restrict__ out_maps,
                                                        const int32_t* __restrict__ offsets, int cin, int cout, int K) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (int64_t)K * cin * cout) return;
  const int co = (int)(e % cout);
  const int ci = (int)((e / cout) % cin);
  const int k = (int)(e / ((int64_t)cin * cout));
  const int p0 = offsets[k], p1 = offsets[k + 1];
  float acc = 0.f;
  for (int p = p0; p < p1; ++p)
    acc += Cvt<T>::ld(x + (int64_t)in_maps[p] * cin + ci) * Cvt<T>::ld(dy + (int64_t)out_maps[p] * cout + co);
  dw[e] = acc;
}

template <typename T>
static int launch_gather_gemm_ref(const void* in, const void* w, void* out, const int32_t* nbr, const float* bias,
                                  int64_t n_out, int cin, int cout, int K, int w_transposed, int k_flip, hipStream_t s) {
  const int kp = wcn_kmap_row_pitch(K);
  const int64_t total = n_out * cout;
  hipLaunchKernelGGL(gather_gemm_ref_kernel<T>, dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, s, (const T*)in,
                     (const T*)w, (T*)out, nbr, bias, n_out, cin, cout, K, kp, w_transposed, k_flip);
  return launch_status();
}

int conv_gather_gemm_ref(const void* in, const void* w, void* out, const int32_t* nbr, const float* bias, int64_t n_out,
                         int cin, int cout, int K, int dtype, int w_transposed, int k_flip, hipStream_t s) {
  switch (dtype) {
    case WCN_F32: return launch_gather_gemm_ref<float>(in, w, out, nbr, bias, n_out, cin, cout, K, w_transposed, k_flip, s);
    case WCN_F16: return launch_gather_gemm_ref<__half>(in, w, out, nbr, bias, n_out, cin, cout, K, w_transposed, k_flip, s);
    case WCN_BF16:
      return launch_gather_gemm_ref<__hip_bfloat16>(in, w, out, nbr, bias, n_out, cin, cout, K, w_transposed, k_flip, s);
    default: return WCN_ERROR_UNSUPPORTED_CONFIG;
  }
}

constexpr int kColsumBlocks = 512;

size_t colsum_workspace(int c) { return (size_t)kColsumBlocks * c * sizeof(float); }

template <typename T>
static int launch_colsum(const void* in, int64_t n, int c, float* out, float* partial, hipStream_t s) {
  constexpr int kVec = 16 / sizeof(T);
  if (sizeof(T) == 2 && c % kVec == 0 && (reinterpret_cast<uintptr_t>(in) & 15) == 0)
    hipLaunchKernelGGL((colsum_partial_kernel<T, (sizeof(T) == 2 ? 8 : 1)>), dim3(kColsumBlocks), dim3(256), 0, s,
                       (const T*)in, n, c, partial);
  else
    hipLaunchKernelGGL((colsum_partial_kernel<T, 1>), dim3(kColsumBlocks), dim3(256), 0, s, (const T*)in, n, c, partial);
  hipLaunchKernelGGL(colsum_final_kernel, dim3((unsigned)c), dim3(64), 0, s, (const float*)partial, kColsumBlocks, c, out);
  return launch_status();
}

int colsum(const void* in, int64_t n, int c, int dtype, float* out, void* workspace, size_t workspace_bytes, hipStream_t s) {
  if (!workspace || workspace_bytes < colsum_workspace(c)) return WCN_ERROR_INVALID_PARAMETERS;
  switch (dtype) {
    case WCN_F32: return launch_colsum<float>(in, n, c, out, (float*)workspace, s);
    case WCN_F16: return launch_colsum<__half>(in, n, c, out, (float*)workspace, s);
    case WCN_BF16: return launch_colsum<__hip_bfloat16>(in, n, c, out, (float*)workspace, s);
    default: return WCN_ERROR_UNSUPPORTED_CONFIG;
  }
}

template <typename T>
static int launch_wgrad_ref(const void* x, const void* dy, float* dw, const int32_t* in_maps, const int32_t* out_maps,
                            const int32_t* offsets, int cin, int cout, int K, hipStream_t s) {
  const int64_t total = (int64_t)K * cin * cout;
  hipLaunchKernelGGL(wgrad_ref_kernel<T>, dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, s, (const T*)x,
                     (const T*)dy, dw, in_maps, out_maps, offsets, cin, cout, K);
  return launch_status();
}

int conv_wgrad_ref(const void* x, const void* dy, float* dw, const int32_t* in_maps, const int32_t* out_maps,
                   const int32_t* offsets, int cin, int cout, int K, int dtype, hipStream_t s) {
  switch (dtype) {
    case WCN_F32: return launch_wgrad_ref<float>(x, dy, dw, in_maps, out_maps, offsets, cin, cout, K, s);
    case WCN_F16: return launch_wgrad_ref<__half>(x, dy, dw, in_maps, out_maps, offsets, cin, cout, K, s);
    case WCN_BF16: return launch_wgrad_ref<__hip_bfloat16>(x, dy, dw, in_maps, out_maps, offsets, cin, cout, K, s);
    default: return WCN_ERROR_UNSUPPORTED_CONFIG;
  }
}

}  // namespace wcn
