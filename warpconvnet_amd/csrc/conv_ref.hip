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
                                                              int64_t n_out, int cin, int cout, int K, int kp,
                                                              int w_transposed, int k_flip) {
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
  Cvt<T>::st(out + e, acc);
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
static int launch_gather_gemm_ref(const void* in, const void* w, void* out, const int32_t* nbr, int64_t n_out, int cin,
                                  int cout, int K, int w_transposed, int k_flip, hipStream_t s) {
  const int kp = wcn_kmap_row_pitch(K);
  const int64_t total = n_out * cout;
  hipLaunchKernelGGL(gather_gemm_ref_kernel<T>, dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, s, (const T*)in,
                     (const T*)w, (T*)out, nbr, n_out, cin, cout, K, kp, w_transposed, k_flip);
  return launch_status();
}

int conv_gather_gemm_ref(const void* in, const void* w, void* out, const int32_t* nbr, int64_t n_out, int cin, int cout,
                         int K, int dtype, int w_transposed, int k_flip, hipStream_t s) {
  switch (dtype) {
    case WCN_F32: return launch_gather_gemm_ref<float>(in, w, out, nbr, n_out, cin, cout, K, w_transposed, k_flip, s);
    case WCN_F16: return launch_gather_gemm_ref<__half>(in, w, out, nbr, n_out, cin, cout, K, w_transposed, k_flip, s);
    case WCN_BF16:
      return launch_gather_gemm_ref<__hip_bfloat16>(in, w, out, nbr, n_out, cin, cout, K, w_transposed, k_flip, s);
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
