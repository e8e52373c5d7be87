// norm.hip - BatchNorm over sparse feature tensors [N, C] (the elementwise chain behind every sparse convolution:
// reference models/mink_unet.py:31-53 runs SparseConv3d -> nn.BatchNorm1d -> ReLU on the feature tensor).
//
// The stock BatchNorm kernels of the framework take 49 + 9 us forward and 50 + 10 us backward on a [200 k, 96] bf16
// tensor (0.8 TB/s) - a third of a MinkUNet iteration.  Here every pass is a streaming read with 16-B pieces per lane
// and several rows in flight per thread:
//   wcn_bn_stats            per-channel mean / biased variance.  One pass: sums of (x - p) and (x - p)^2 around the pivot
//                           p[c] = x[0][c] (a value from the distribution: no catastrophic cancellation in
//                           E[d^2] - E[d]^2), fp32, fixed-order two-level reduction => deterministic.
//   wcn_bn_apply            y = x * scale[c] + shift[c], optional ReLU.
//   wcn_bn_backward_reduce  sum_dy[c], sum_dy_xhat[c] with dy masked where the fused ReLU stored a zero; the mask is
//                           recomputed from x and the forward's scale / shift (bn_affine), the output y is not read again.
//   wcn_bn_backward_apply   dx = gamma * rstd * (dy - sum_dy / N - xhat * sum_dy_xhat / N).
//   wcn_bn_apply_residual / wcn_bn_backward_{reduce,apply}_masked   the tail of a residual block, z = ReLU(BN(x) + r)
//                           (reference models/mink_unet.py:160-172: `out += identity; out = relu(out)`), in the same two
//                           passes: the forward adds r while it applies, the backward masks with the stored z and also
//                           writes the masked gradient (the residual branch's share).
#include <hip/hip_bf16.h>
#include <hip/hip_fp16.h>

#include "wcn_common.h"

namespace wcn {

// First reduction level: workgroups (= partial sums per channel) and rows in flight per thread.  Measured on [1 M, 96] /
// [1 M, 32] / [290 k, 64] bf16, statistics and backward-reduce passes incl. the second level (tools/bench_bn.py, us):
// 1024 x 4: 39 / 31 / 19 and 101 / 44 / 32;  512 x 8: 37 / 26 / 16 and 86 / 37 / 26;  256 x 8: 41 / 26 / 16 and 106 / 39 / 27;
// 2048 x 4 and 512 x 16 are slower than 1024 x 4 (dev builds: -DWCN_NORM_BLOCKS / -DWCN_NORM_ROWS, tools/build_abl.sh).
#ifndef WCN_NORM_BLOCKS
#define WCN_NORM_BLOCKS 512
#endif
#ifndef WCN_NORM_ROWS
#define WCN_NORM_ROWS 8
#endif
constexpr int kNormBlocks = WCN_NORM_BLOCKS;
constexpr int kNormRowsInFlight = WCN_NORM_ROWS;

template <typename T> struct NCvt;
template <> struct NCvt<float> {
  static __device__ __forceinline__ float ld(float v) { return v; }
  static __device__ __forceinline__ float st(float v) { return v; }
};
template <> struct NCvt<__half> {
  static __device__ __forceinline__ float ld(__half v) { return __half2float(v); }
  static __device__ __forceinline__ __half st(float v) { return __float2half(v); }
};
template <> struct NCvt<__hip_bfloat16> {
  static __device__ __forceinline__ float ld(__hip_bfloat16 v) { return __bfloat162float(v); }
  static __device__ __forceinline__ __hip_bfloat16 st(float v) { return __float2bfloat16(v); }
};

template <typename T, int VEC> struct alignas(sizeof(T) * VEC) NVec { T v[VEC]; };

// y = x * scale + shift as ONE fused multiply-add, in the forward pass and wherever the backward passes need to know whether
// the ReLU behind it let a value through: the mask (stored y > 0) is recomputed from x with the very same operation instead
// of reading the forward output again (a third of the backward passes' traffic).
__device__ __forceinline__ float bn_affine(float xf, float sc, float sh) {
  float f = __builtin_fmaf(xf, sc, sh);
  // The fp32 result is materialised before anything rounds it to the storage type.  Without the barrier the compiler fuses the
  // fma with a following fp16 conversion into v_fma_mixlo_f16 (ONE rounding of the exact result) in some kernels and not in
  // others (fp32 fma, then v_cvt_f16_f32: two roundings) - 1 ulp apart in ~1 of 40 000 elements, i.e. the passes that must
  // agree on "what the forward stored" (the residual tail vs the plain apply, the backward's recomputed ReLU mask) did not for
  // fp16 (tools/soak_models.py).  bf16 has no such instruction and was never affected.
  asm volatile("" : "+v"(f));
  return f;
}
template <typename T>
__device__ __forceinline__ bool bn_relu_passes(float xf, float sc, float sh) {
  return NCvt<T>::ld(NCvt<T>::st(fmaxf(bn_affine(xf, sc, sh), 0.f))) > 0.f;  // what the forward stored, compared with zero
}

// Thread layout of the column reductions: lanes_c threads side by side cover one row (VEC channels each), the
// remaining factor of the 256 threads covers different rows; a workgroup owns a contiguous range of rows.
// MODE 0: a = x - pivot,                s0 += a,  s1 += a * a
// MODE 1: g = dy (0 where the fused ReLU stored a zero), xh = (x - mean) * rstd,   s0 += g,  s1 += g * xh
template <typename T, int VEC, int MODE>
__global__ __launch_bounds__(256) void norm_reduce_kernel(const T* __restrict__ x, const T* __restrict__ dy,
                                                          const float* __restrict__ rscale, const float* __restrict__ rshift,
                                                          int64_t n, int c,
                                                          const float* __restrict__ mean, const float* __restrict__ rstd,
                                                          float* __restrict__ partial, const T* __restrict__ zmask,
                                                          int64_t dy_ld) {
  // zmask (MODE 1, instead of rscale / rshift): the stored output of ReLU(BN(x) + residual); g = dy where it is positive
  // dy_ld: row pitch of dy in elements (a column slice of a wider tensor - the gradient of a channel concatenation - is read in place)
  __shared__ float s_red[2][256 * VEC];
  const int tid = threadIdx.x;
  const int cgroups = (c + VEC - 1) / VEC;
  const int lanes_c = cgroups < 256 ? cgroups : 256;
  const int rsteps = 256 / lanes_c;
  const int cc = tid % lanes_c, rr = tid / lanes_c;
  const int64_t rows_per_block = (n + gridDim.x - 1) / gridDim.x;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
  const int64_t r1 = (r0 + rows_per_block < n) ? (r0 + rows_per_block) : n;
  for (int g0 = 0; g0 < cgroups; g0 += lanes_c) {
    const int ch0 = (g0 + cc) * VEC;
    float s0[VEC], s1[VEC], a[VEC], b[VEC], sc[VEC], sh[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) { s0[v] = 0.f; s1[v] = 0.f; a[v] = 0.f; b[v] = 1.f; sc[v] = 0.f; sh[v] = 0.f; }
    const bool mine = rr < rsteps && g0 + cc < cgroups;
    if (mine) {
#pragma unroll
      for (int v = 0; v < VEC; ++v) {
        if (ch0 + v < c) {
          if (MODE == 0) a[v] = NCvt<T>::ld(x[ch0 + v]);  // pivot: row 0
          else {
            a[v] = mean[ch0 + v]; b[v] = rstd[ch0 + v];
            if (rscale) { sc[v] = rscale[ch0 + v]; sh[v] = rshift[ch0 + v]; }
          }
        }
      }
      for (int64_t r = r0 + rr; r < r1; r += (int64_t)rsteps * kNormRowsInFlight) {
        NVec<T, VEC> xv[kNormRowsInFlight], gv[kNormRowsInFlight], zv[kNormRowsInFlight];
        // clamped addresses: the loads of all rows in flight are issued before the first one is used
#pragma unroll
        for (int q = 0; q < kNormRowsInFlight; ++q) {
          const int64_t rq = r + (int64_t)q * rsteps;
          const int64_t rc = rq < r1 ? rq : r1 - 1;
          const int64_t at = rc * c + ch0;
          if (VEC > 1) {
            xv[q] = *reinterpret_cast<const NVec<T, VEC>*>(x + at);
            if (MODE == 1) gv[q] = *reinterpret_cast<const NVec<T, VEC>*>(dy + rc * dy_ld + ch0);
            if (MODE == 1 && zmask) zv[q] = *reinterpret_cast<const NVec<T, VEC>*>(zmask + at);
          } else {
            xv[q].v[0] = x[at];
            if (MODE == 1) gv[q].v[0] = dy[rc * dy_ld + ch0];
            if (MODE == 1 && zmask) zv[q].v[0] = zmask[at];
          }
        }
#pragma unroll
        for (int q = 0; q < kNormRowsInFlight; ++q) {
          if (r + (int64_t)q * rsteps >= r1) continue;
#pragma unroll
          for (int v = 0; v < VEC; ++v) {
            const float xf = NCvt<T>::ld(xv[q].v[v]);
            if (MODE == 0) {
              const float d = xf - a[v];
              s0[v] += d;
              s1[v] += d * d;
            } else {
              float g = NCvt<T>::ld(gv[q].v[v]);
              if (zmask) { if (!(NCvt<T>::ld(zv[q].v[v]) > 0.f)) g = 0.f; }
              else if (rscale && !bn_relu_passes<T>(xf, sc[v], sh[v])) g = 0.f;
              s0[v] += g;
              s1[v] += g * ((xf - a[v]) * b[v]);
            }
          }
        }
      }
    }
#pragma unroll
    for (int v = 0; v < VEC; ++v) { s_red[0][tid * VEC + v] = s0[v]; s_red[1][tid * VEC + v] = s1[v]; }
    __syncthreads();
    if (rr == 0 && g0 + cc < cgroups) {
#pragma unroll
      for (int v = 0; v < VEC; ++v) {
        float t0 = 0.f, t1 = 0.f;
        for (int q = 0; q < rsteps; ++q) {
          t0 += s_red[0][(q * lanes_c + cc) * VEC + v];
          t1 += s_red[1][(q * lanes_c + cc) * VEC + v];
        }
        if (ch0 + v < c) {
          partial[((int64_t)blockIdx.x * 2 + 0) * c + ch0 + v] = t0;
          partial[((int64_t)blockIdx.x * 2 + 1) * c + ch0 + v] = t1;
        }
      }
    }
    __syncthreads();
  }
}

// second level: one wave per channel, blocks summed in a fixed order.  MODE 0 finishes mean / biased variance.
struct BnFold {  // optional tail of the statistics pass: everything a training step derives from mean / var per channel
  const float* gamma = nullptr;   // [c] or null (= 1)
  const float* beta = nullptr;    // [c] or null (= 0)
  float* running_mean = nullptr;  // [c] fp32, updated in place, or null
  float* running_var = nullptr;   // [c] fp32, updated in place with the UNBIASED variance, or null
  float momentum = 0.f, eps = 1e-5f;
  float* rstd = nullptr;          // [c] out (null: only mean / var are produced)
  float* scale = nullptr;         // [c] out: gamma * rstd
  float* shift = nullptr;         // [c] out: beta - mean * scale
  long long* batches = nullptr;   // nn.BatchNorm's num_batches_tracked (int64 scalar), += 1, or null
};

template <typename T, int MODE>
__global__ __launch_bounds__(64) void norm_final_kernel(const float* __restrict__ partial, int nblocks, int c, int64_t n,
                                                        const T* __restrict__ x, float* __restrict__ out0,
                                                        float* __restrict__ out1, const BnFold f) {
  const int ch = blockIdx.x, lane = threadIdx.x;
  // all of a lane's partial sums requested before the first add (16 + 16 strided loads: issued one after the other they
  // were 16 L2 round trips, 8-9 us for a kernel that moves a few KB), summed in the same fixed order
  constexpr int kPer = kNormBlocks / 64;
  float v0[kPer], v1[kPer];
#pragma unroll
  for (int q = 0; q < kPer; ++q) {
    const int b = lane + 64 * q;
    const int bb = b < nblocks ? b : 0;
    v0[q] = partial[((int64_t)bb * 2 + 0) * c + ch];
    v1[q] = partial[((int64_t)bb * 2 + 1) * c + ch];
  }
  float t0 = 0.f, t1 = 0.f;
#pragma unroll
  for (int q = 0; q < kPer; ++q)
    if (lane + 64 * q < nblocks) { t0 += v0[q]; t1 += v1[q]; }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    t0 += __shfl_down(t0, d);
    t1 += __shfl_down(t1, d);
  }
  if (lane == 0) {
    if (MODE == 0) {
      const float inv = 1.0f / (float)n;
      const float md = t0 * inv;                       // mean of (x - pivot)
      const float var = fmaxf(t1 * inv - md * md, 0.f);
      const float mean = NCvt<T>::ld(x[ch]) + md;
      out0[ch] = mean;
      out1[ch] = var;
      if (f.rstd) {
        const float rstd = rsqrtf(var + f.eps);
        const float sc = f.gamma ? f.gamma[ch] * rstd : rstd;
        f.rstd[ch] = rstd;
        f.scale[ch] = sc;
        f.shift[ch] = (f.beta ? f.beta[ch] : 0.f) - mean * sc;
        if (f.running_mean) {
          const float unbias = n > 1 ? (float)n / (float)(n - 1) : 1.0f;
          f.running_mean[ch] = (1.0f - f.momentum) * f.running_mean[ch] + f.momentum * mean;
          f.running_var[ch] = (1.0f - f.momentum) * f.running_var[ch] + f.momentum * var * unbias;
        }
        if (f.batches && ch == 0) *f.batches += 1;
      }
    } else {
      out0[ch] = t0;
      out1[ch] = t1;
    }
  }
}

// inference: scale / shift (and mean / rstd for a backward pass) from the running statistics, one launch
__global__ void norm_fold_kernel(const float* __restrict__ mean_in, const float* __restrict__ var_in, int c, const BnFold f,
                                 float* __restrict__ mean_out) {
  const int ch = blockIdx.x * blockDim.x + threadIdx.x;
  if (ch >= c) return;
  const float mean = mean_in[ch];
  const float rstd = rsqrtf(var_in[ch] + f.eps);
  const float sc = f.gamma ? f.gamma[ch] * rstd : rstd;
  mean_out[ch] = mean;
  f.rstd[ch] = rstd;
  f.scale[ch] = sc;
  f.shift[ch] = (f.beta ? f.beta[ch] : 0.f) - mean * sc;
}

// The per-channel coefficients of the two elementwise passes are staged in LDS once per workgroup: read per element from
// global memory they cost one texture-addresser slot per lane and value (5 arrays x 8 channels per thread made the
// backward pass 343 us instead of 150 on [1 M, 96] bf16).
template <typename T, int VEC>
__global__ __launch_bounds__(256) void norm_apply_kernel(const T* __restrict__ x, int64_t n, int c,
                                                         const float* __restrict__ scale, const float* __restrict__ shift,
                                                         int relu, T* __restrict__ y, const T* __restrict__ res) {
  // res: y = [ReLU](round(x * scale + shift) + res) - the roundings of BatchNorm -> add -> ReLU run as three modules
  extern __shared__ float s_coef[];  // [2][c]: scale, shift
  for (int i = threadIdx.x; i < c; i += blockDim.x) {
    s_coef[i] = scale[i];
    s_coef[c + i] = shift[i];
  }
  __syncthreads();
  const int cv = c / VEC;
  const int64_t total = n * cv;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int ch0 = (int)(e % cv) * VEC;
    const NVec<T, VEC> xv = *reinterpret_cast<const NVec<T, VEC>*>(x + e * VEC);
    NVec<T, VEC> yv, rv;
    if (res) rv = *reinterpret_cast<const NVec<T, VEC>*>(res + e * VEC);
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
      float f = bn_affine(NCvt<T>::ld(xv.v[v]), s_coef[ch0 + v], s_coef[c + ch0 + v]);
      // round, then add in fp32 and round again: what the framework's elementwise add of two 16-bit tensors does (for fp16 the
      // fp32 sum is exact enough that this equals one rounding of the exact sum - checked on 84 M pairs)
      if (res) f = NCvt<T>::ld(NCvt<T>::st(NCvt<T>::ld(NCvt<T>::st(f)) + NCvt<T>::ld(rv.v[v])));
      if (relu) f = fmaxf(f, 0.f);
      yv.v[v] = NCvt<T>::st(f);
    }
    *reinterpret_cast<NVec<T, VEC>*>(y + e * VEC) = yv;
  }
}

// dx = gamma * rstd * (g - sum_dy / n - xhat * sum_dy_xhat / n) = A[c] * g + B[c] * x + C[c]
template <typename T, int VEC>
__global__ __launch_bounds__(256) void norm_bwd_apply_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                             const float* __restrict__ rscale,
                                                             const float* __restrict__ rshift, int64_t n, int c,
                                                             const float* __restrict__ mean, const float* __restrict__ rstd,
                                                             const float* __restrict__ gamma,
                                                             const float* __restrict__ sum_dy,
                                                             const float* __restrict__ sum_dy_xhat, T* __restrict__ dx,
                                                             const T* __restrict__ zmask, T* __restrict__ dres,
                                                             int64_t dy_ld) {
  // zmask / dres: residual tail - the mask is the stored output's sign, the masked gradient is also the residual branch's
  extern __shared__ float s_coef[];  // [5][c]: A, B, C, and scale / shift of the forward pass (ReLU mask)
  const float inv_n = 1.0f / (float)n;
  for (int i = threadIdx.x; i < c; i += blockDim.x) {
    const float r = rstd[i], w = (gamma ? gamma[i] : 1.0f) * r;
    const float k1 = r * sum_dy_xhat[i] * inv_n;  // coefficient of (x - mean)
    s_coef[i] = w;
    s_coef[c + i] = -w * k1;
    s_coef[2 * c + i] = w * (mean[i] * k1 - sum_dy[i] * inv_n);
    s_coef[3 * c + i] = rscale ? rscale[i] : 0.f;
    s_coef[4 * c + i] = rscale ? rshift[i] : 0.f;
  }
  __syncthreads();
  const int cv = c / VEC;
  const int64_t total = n * cv;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int ch0 = (int)(e % cv) * VEC;
    const NVec<T, VEC> gv = *reinterpret_cast<const NVec<T, VEC>*>(dy + (dy_ld == c ? e * VEC : (e / cv) * dy_ld + ch0));
    const NVec<T, VEC> xv = *reinterpret_cast<const NVec<T, VEC>*>(x + e * VEC);
    NVec<T, VEC> ov, zv, mv;
    if (zmask) zv = *reinterpret_cast<const NVec<T, VEC>*>(zmask + e * VEC);
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
      const int ch = ch0 + v;
      const float xf = NCvt<T>::ld(xv.v[v]);
      float g = NCvt<T>::ld(gv.v[v]);
      if (zmask) { if (!(NCvt<T>::ld(zv.v[v]) > 0.f)) g = 0.f; }
      else if (rscale && !bn_relu_passes<T>(xf, s_coef[3 * c + ch], s_coef[4 * c + ch])) g = 0.f;
      mv.v[v] = NCvt<T>::st(g);
      ov.v[v] = NCvt<T>::st(s_coef[ch] * g + s_coef[c + ch] * xf + s_coef[2 * c + ch]);
    }
    *reinterpret_cast<NVec<T, VEC>*>(dx + e * VEC) = ov;
    if (dres) *reinterpret_cast<NVec<T, VEC>*>(dres + e * VEC) = mv;
  }
}

// grid-stride: a few workgroups per CU, the LDS staging is amortised.  Cap measured with tools/bench_bn.py on one box ([1 M, 96] /
// [1 M, 32] / [290 k, 64] bf16, us): forward apply 2048: 70 / 21 / 13, 4096: 70 / 22 / 14; backward apply 2048: 128 / 33 / 21,
// 4096: 114 / 31 / 20.  (Non-temporal loads / stores: 5 % faster on the tensor that exceeds the Infinity Cache, 10-15 % slower on
// the ones a neighbouring kernel finds there - not used.)
static inline unsigned norm_grid(int64_t items, int cap = 2048) {
  const int64_t g = ceil_div(items, 256 * 4);
  return (unsigned)(g < cap ? (g < 1 ? 1 : g) : cap);
}

template <typename T>
static int bn_reduce_t(int mode, const void* x, const void* dy, const float* rscale, const float* rshift, int64_t n, int c,
                       const float* mean,
                       const float* rstd, float* out0, float* out1, float* partial, hipStream_t s,
                       const BnFold& fold = BnFold(), const void* zmask = nullptr, int64_t dy_ld = 0) {
  constexpr int VEC = 16 / (int)sizeof(T);
  if (dy_ld <= 0) dy_ld = c;
  // first-level workgroups: at least 128 rows each (small tensors: fewer partial sums for the second level), kNormBlocks at most
  int64_t nb = ceil_div(n < 1 ? 1 : n, 128);
  const int nblocks = (int)(nb < kNormBlocks ? nb : kNormBlocks);
  const bool vec = c % VEC == 0 && dy_ld % VEC == 0 && (reinterpret_cast<uintptr_t>(dy) & 15) == 0;
#define WCN_NR(V, M)                                                                                                  \
  hipLaunchKernelGGL((norm_reduce_kernel<T, V, M>), dim3(nblocks), dim3(256), 0, s, (const T*)x, (const T*)dy,          \
                     rscale, rshift, n, c, mean, rstd, partial, (const T*)zmask, dy_ld)
  if (mode == 0) { if (vec) WCN_NR(VEC, 0); else WCN_NR(1, 0); }
  else { if (vec) WCN_NR(VEC, 1); else WCN_NR(1, 1); }
#undef WCN_NR
  if (mode == 0)
    hipLaunchKernelGGL((norm_final_kernel<T, 0>), dim3((unsigned)c), dim3(64), 0, s, (const float*)partial, nblocks, c, n,
                       (const T*)x, out0, out1, fold);
  else
    hipLaunchKernelGGL((norm_final_kernel<T, 1>), dim3((unsigned)c), dim3(64), 0, s, (const float*)partial, nblocks, c, n,
                       (const T*)x, out0, out1, BnFold());
  return launch_status();
}

template <typename T>
static int bn_apply_t(const void* x, int64_t n, int c, const float* scale, const float* shift, int relu, void* y,
                      hipStream_t s, const void* res = nullptr) {
  constexpr int VEC = 16 / (int)sizeof(T);
  if (c % VEC == 0)
    hipLaunchKernelGGL((norm_apply_kernel<T, VEC>), dim3(norm_grid(n * (c / VEC))), dim3(256), (size_t)2 * c * 4, s,
                       (const T*)x, n, c, scale, shift, relu, (T*)y, (const T*)res);
  else
    hipLaunchKernelGGL((norm_apply_kernel<T, 1>), dim3(norm_grid(n * c)), dim3(256), (size_t)2 * c * 4, s, (const T*)x, n, c,
                       scale, shift, relu, (T*)y, (const T*)res);
  return launch_status();
}

template <typename T>
static int bn_bwd_apply_t(const void* dy, const void* x, const float* rscale, const float* rshift, int64_t n, int c,
                          const float* mean,
                          const float* rstd, const float* gamma, const float* sum_dy, const float* sum_dy_xhat, void* dx,
                          hipStream_t s, const void* zmask = nullptr, void* dres = nullptr, int64_t dy_ld = 0) {
  constexpr int VEC = 16 / (int)sizeof(T);
  if (dy_ld <= 0) dy_ld = c;
  if (c % VEC == 0 && dy_ld % VEC == 0 && (reinterpret_cast<uintptr_t>(dy) & 15) == 0)
    hipLaunchKernelGGL((norm_bwd_apply_kernel<T, VEC>), dim3(norm_grid(n * (c / VEC), 4096)), dim3(256), (size_t)5 * c * 4, s,
                       (const T*)dy, (const T*)x, rscale, rshift, n, c, mean, rstd, gamma, sum_dy, sum_dy_xhat, (T*)dx,
                       (const T*)zmask, (T*)dres, dy_ld);
  else
    hipLaunchKernelGGL((norm_bwd_apply_kernel<T, 1>), dim3(norm_grid(n * c, 4096)), dim3(256), (size_t)5 * c * 4, s, (const T*)dy,
                       (const T*)x, rscale, rshift, n, c, mean, rstd, gamma, sum_dy, sum_dy_xhat, (T*)dx, (const T*)zmask,
                       (T*)dres, dy_ld);
  return launch_status();
}

}  // namespace wcn

using namespace wcn;

extern "C" {

size_t wcn_bn_workspace(int32_t channels) { return channels > 0 ? (size_t)kNormBlocks * 2 * channels * sizeof(float) : 0; }

static bool bn_dtype_ok(int dtype) { return dtype == WCN_F32 || dtype == WCN_F16 || dtype == WCN_BF16; }

int wcn_bn_stats(const void* x, int64_t n, int32_t channels, int32_t dtype, float* mean, float* var, void* workspace,
                 size_t workspace_bytes, wcn_stream_t stream) {
  if (n < 1 || channels < 1 || !bn_dtype_ok(dtype) || !x || !mean || !var || !workspace ||
      workspace_bytes < wcn_bn_workspace(channels))
    return WCN_ERROR_INVALID_PARAMETERS;
  hipStream_t s = (hipStream_t)stream;
  float* p = (float*)workspace;
  switch (dtype) {
    case WCN_F32: return bn_reduce_t<float>(0, x, nullptr, nullptr, nullptr, n, channels, nullptr, nullptr, mean, var, p, s);
    case WCN_F16: return bn_reduce_t<__half>(0, x, nullptr, nullptr, nullptr, n, channels, nullptr, nullptr, mean, var, p, s);
    default: return bn_reduce_t<__hip_bfloat16>(0, x, nullptr, nullptr, nullptr, n, channels, nullptr, nullptr, mean, var, p, s);
  }
}

int wcn_bn_stats_fold(const void* x, int64_t n, int32_t channels, int32_t dtype, const float* gamma, const float* beta,
                      float* running_mean, float* running_var, float momentum, float eps, float* mean, float* var,
                      float* rstd, float* scale, float* shift, int64_t* num_batches_tracked, void* workspace,
                      size_t workspace_bytes, wcn_stream_t stream) {
  if (n < 1 || channels < 1 || !bn_dtype_ok(dtype) || !x || !mean || !var || !rstd || !scale || !shift || !workspace ||
      workspace_bytes < wcn_bn_workspace(channels) || ((running_mean == nullptr) != (running_var == nullptr)))
    return WCN_ERROR_INVALID_PARAMETERS;
  BnFold f;
  f.gamma = gamma; f.beta = beta; f.running_mean = running_mean; f.running_var = running_var;
  f.momentum = momentum; f.eps = eps; f.rstd = rstd; f.scale = scale; f.shift = shift;
  f.batches = reinterpret_cast<long long*>(num_batches_tracked);
  hipStream_t s = (hipStream_t)stream;
  float* p = (float*)workspace;
  switch (dtype) {
    case WCN_F32: return bn_reduce_t<float>(0, x, nullptr, nullptr, nullptr, n, channels, nullptr, nullptr, mean, var, p, s, f);
    case WCN_F16: return bn_reduce_t<__half>(0, x, nullptr, nullptr, nullptr, n, channels, nullptr, nullptr, mean, var, p, s, f);
    default: return bn_reduce_t<__hip_bfloat16>(0, x, nullptr, nullptr, nullptr, n, channels, nullptr, nullptr, mean, var, p, s, f);
  }
}

int wcn_bn_fold(const float* running_mean, const float* running_var, const float* gamma, const float* beta, float eps,
                int32_t channels, float* mean, float* rstd, float* scale, float* shift, wcn_stream_t stream) {
  if (channels < 1 || !running_mean || !running_var || !mean || !rstd || !scale || !shift) return WCN_ERROR_INVALID_PARAMETERS;
  BnFold f;
  f.gamma = gamma; f.beta = beta; f.eps = eps; f.rstd = rstd; f.scale = scale; f.shift = shift;
  hipLaunchKernelGGL(norm_fold_kernel, dim3((unsigned)ceil_div(channels, 256)), dim3(256), 0, (hipStream_t)stream,
                     running_mean, running_var, (int)channels, f, mean);
  return launch_status();
}

int wcn_bn_apply(const void* x, int64_t n, int32_t channels, int32_t dtype, const float* scale, const float* shift,
                 int32_t relu, void* y, wcn_stream_t stream) {
  if (n < 0 || channels < 1 || !bn_dtype_ok(dtype)) return WCN_ERROR_INVALID_PARAMETERS;
  if (n == 0) return WCN_SUCCESS;
  if (!x || !y || !scale || !shift) return WCN_ERROR_INVALID_PARAMETERS;
  hipStream_t s = (hipStream_t)stream;
  switch (dtype) {
    case WCN_F32: return bn_apply_t<float>(x, n, channels, scale, shift, relu, y, s);
    case WCN_F16: return bn_apply_t<__half>(x, n, channels, scale, shift, relu, y, s);
    default: return bn_apply_t<__hip_bfloat16>(x, n, channels, scale, shift, relu, y, s);
  }
}

static int bn_backward_reduce_ld(const void* dy, int64_t dy_ld, const void* x, const float* relu_scale, const float* relu_shift,
                                int64_t n, int32_t channels, int32_t dtype, const float* mean, const float* rstd, float* sum_dy,
                                float* sum_dy_xhat, void* workspace, size_t workspace_bytes, wcn_stream_t stream) {
  if (n < 1 || channels < 1 || !bn_dtype_ok(dtype) || !dy || !x || !mean || !rstd || !sum_dy || !sum_dy_xhat ||
      !workspace || workspace_bytes < wcn_bn_workspace(channels) || ((relu_scale == nullptr) != (relu_shift == nullptr)))
    return WCN_ERROR_INVALID_PARAMETERS;
  hipStream_t s = (hipStream_t)stream;
  float* p = (float*)workspace;
  switch (dtype) {
    case WCN_F32:
      return bn_reduce_t<float>(1, x, dy, relu_scale, relu_shift, n, channels, mean, rstd, sum_dy, sum_dy_xhat, p, s, BnFold(), nullptr, dy_ld);
    case WCN_F16:
      return bn_reduce_t<__half>(1, x, dy, relu_scale, relu_shift, n, channels, mean, rstd, sum_dy, sum_dy_xhat, p, s, BnFold(), nullptr, dy_ld);
    default:
      return bn_reduce_t<__hip_bfloat16>(1, x, dy, relu_scale, relu_shift, n, channels, mean, rstd, sum_dy, sum_dy_xhat, p, s, BnFold(),
                                         nullptr, dy_ld);
  }
}
int wcn_bn_backward_reduce(const void* dy, const void* x, const float* relu_scale, const float* relu_shift, int64_t n,
                           int32_t channels, int32_t dtype, const float* mean, const float* rstd, float* sum_dy,
                           float* sum_dy_xhat, void* workspace, size_t workspace_bytes, wcn_stream_t stream) {
  return bn_backward_reduce_ld(dy, 0, x, relu_scale, relu_shift, n, channels, dtype, mean, rstd, sum_dy, sum_dy_xhat, workspace,
                               workspace_bytes, stream);
}

static int bn_backward_apply_ld(const void* dy, int64_t dy_ld, const void* x, const float* relu_scale, const float* relu_shift,
                               int64_t n, int32_t channels, int32_t dtype, const float* mean, const float* rstd, const float* gamma,
                               const float* sum_dy, const float* sum_dy_xhat, void* dx, wcn_stream_t stream) {
  if (n < 0 || channels < 1 || !bn_dtype_ok(dtype) || ((relu_scale == nullptr) != (relu_shift == nullptr)))
    return WCN_ERROR_INVALID_PARAMETERS;
  if (n == 0) return WCN_SUCCESS;
  if (!dy || !x || !dx || !mean || !rstd || !sum_dy || !sum_dy_xhat) return WCN_ERROR_INVALID_PARAMETERS;
  hipStream_t s = (hipStream_t)stream;
  switch (dtype) {
    case WCN_F32:
      return bn_bwd_apply_t<float>(dy, x, relu_scale, relu_shift, n, channels, mean, rstd, gamma, sum_dy, sum_dy_xhat, dx, s, nullptr,
                                   nullptr, dy_ld);
    case WCN_F16:
      return bn_bwd_apply_t<__half>(dy, x, relu_scale, relu_shift, n, channels, mean, rstd, gamma, sum_dy, sum_dy_xhat, dx, s, nullptr,
                                    nullptr, dy_ld);
    default:
      return bn_bwd_apply_t<__hip_bfloat16>(dy, x, relu_scale, relu_shift, n, channels, mean, rstd, gamma, sum_dy, sum_dy_xhat, dx, s,
                                            nullptr, nullptr, dy_ld);
  }
}
int wcn_bn_backward_apply(const void* dy, const void* x, const float* relu_scale, const float* relu_shift, int64_t n,
                          int32_t channels, int32_t dtype, const float* mean, const float* rstd, const float* gamma,
                          const float* sum_dy, const float* sum_dy_xhat, void* dx, wcn_stream_t stream) {
  return bn_backward_apply_ld(dy, 0, x, relu_scale, relu_shift, n, channels, dtype, mean, rstd, gamma, sum_dy, sum_dy_xhat, dx, stream);
}

// ---- residual tail: z = [ReLU](BN(x) + residual), reference models/mink_unet.py:160-172 ----

int wcn_bn_apply_residual(const void* x, const void* residual, int64_t n, int32_t channels, int32_t dtype, const float* scale,
                          const float* shift, int32_t relu, void* y, wcn_stream_t stream) {
  if (n < 0 || channels < 1 || !bn_dtype_ok(dtype)) return WCN_ERROR_INVALID_PARAMETERS;
  if (n == 0) return WCN_SUCCESS;
  if (!x || !residual || !y || !scale || !shift) return WCN_ERROR_INVALID_PARAMETERS;
  hipStream_t s = (hipStream_t)stream;
  switch (dtype) {
    case WCN_F32: return bn_apply_t<float>(x, n, channels, scale, shift, relu, y, s, residual);
    case WCN_F16: return bn_apply_t<__half>(x, n, channels, scale, shift, relu, y, s, residual);
    default: return bn_apply_t<__hip_bfloat16>(x, n, channels, scale, shift, relu, y, s, residual);
  }
}

static int bn_backward_reduce_masked_ld(const void* dy, int64_t dy_ld, const void* x, const void* z, int64_t n, int32_t channels,
                                       int32_t dtype, const float* mean, const float* rstd, float* sum_dy, float* sum_dy_xhat,
                                       void* workspace, size_t workspace_bytes, wcn_stream_t stream) {
  if (n < 1 || channels < 1 || !bn_dtype_ok(dtype) || !dy || !x || !z || !mean || !rstd || !sum_dy || !sum_dy_xhat ||
      !workspace || workspace_bytes < wcn_bn_workspace(channels))
    return WCN_ERROR_INVALID_PARAMETERS;
  hipStream_t s = (hipStream_t)stream;
  float* p = (float*)workspace;
  const BnFold nf;
  switch (dtype) {
    case WCN_F32: return bn_reduce_t<float>(1, x, dy, nullptr, nullptr, n, channels, mean, rstd, sum_dy, sum_dy_xhat, p, s, nf, z, dy_ld);
    case WCN_F16: return bn_reduce_t<__half>(1, x, dy, nullptr, nullptr, n, channels, mean, rstd, sum_dy, sum_dy_xhat, p, s, nf, z, dy_ld);
    default:
      return bn_reduce_t<__hip_bfloat16>(1, x, dy, nullptr, nullptr, n, channels, mean, rstd, sum_dy, sum_dy_xhat, p, s, nf, z, dy_ld);
  }
}
int wcn_bn_backward_reduce_masked(const void* dy, const void* x, const void* z, int64_t n, int32_t channels, int32_t dtype,
                                  const float* mean, const float* rstd, float* sum_dy, float* sum_dy_xhat, void* workspace,
                                  size_t workspace_bytes, wcn_stream_t stream) {
  return bn_backward_reduce_masked_ld(dy, 0, x, z, n, channels, dtype, mean, rstd, sum_dy, sum_dy_xhat, workspace, workspace_bytes, stream);
}

static int bn_backward_apply_masked_ld(const void* dy, int64_t dy_ld, const void* x, const void* z, int64_t n, int32_t channels,
                                      int32_t dtype, const float* mean, const float* rstd, const float* gamma, const float* sum_dy,
                                      const float* sum_dy_xhat, void* dx, void* dres, wcn_stream_t stream) {
  if (n < 0 || channels < 1 || !bn_dtype_ok(dtype)) return WCN_ERROR_INVALID_PARAMETERS;
  if (n == 0) return WCN_SUCCESS;
  if (!dy || !x || !z || !dx || !mean || !rstd || !sum_dy || !sum_dy_xhat) return WCN_ERROR_INVALID_PARAMETERS;
  hipStream_t s = (hipStream_t)stream;
  switch (dtype) {
    case WCN_F32:
      return bn_bwd_apply_t<float>(dy, x, nullptr, nullptr, n, channels, mean, rstd, gamma, sum_dy, sum_dy_xhat, dx, s, z, dres, dy_ld);
    case WCN_F16:
      return bn_bwd_apply_t<__half>(dy, x, nullptr, nullptr, n, channels, mean, rstd, gamma, sum_dy, sum_dy_xhat, dx, s, z, dres, dy_ld);
    default:
      return bn_bwd_apply_t<__hip_bfloat16>(dy, x, nullptr, nullptr, n, channels, mean, rstd, gamma, sum_dy, sum_dy_xhat, dx, s, z,
                                            dres, dy_ld);
  }
}
int wcn_bn_backward_apply_masked(const void* dy, const void* x, const void* z, int64_t n, int32_t channels, int32_t dtype,
                                 const float* mean, const float* rstd, const float* gamma, const float* sum_dy,
                                 const float* sum_dy_xhat, void* dx, void* dres, wcn_stream_t stream) {
  return bn_backward_apply_masked_ld(dy, 0, x, z, n, channels, dtype, mean, rstd, gamma, sum_dy, sum_dy_xhat, dx, dres, stream);
}

// ---- layer entries: the BatchNorm of a training step in one call per direction ----
// `stats`: [5][channels] fp32 = mean | rstd | scale | shift | biased variance, written by the forward and read by the backward.

int wcn_bn_train_forward(const void* x, const void* residual, int64_t n, int32_t channels, int32_t dtype, const float* gamma,
                         const float* beta, float* running_mean, float* running_var, float momentum, float eps,
                         int64_t* num_batches_tracked, int32_t relu, float* stats, void* y, void* workspace,
                         size_t workspace_bytes, wcn_stream_t stream) {
  if (!stats) return WCN_ERROR_INVALID_PARAMETERS;
  float* mean = stats;
  float* rstd = stats + channels;
  float* scale = stats + 2 * (int64_t)channels;
  float* shift = stats + 3 * (int64_t)channels;
  float* var = stats + 4 * (int64_t)channels;
  const int rc = wcn_bn_stats_fold(x, n, channels, dtype, gamma, beta, running_mean, running_var, momentum, eps, mean, var, rstd,
                                   scale, shift, num_batches_tracked, workspace, workspace_bytes, stream);
  if (rc != WCN_SUCCESS) return rc;
  return residual ? wcn_bn_apply_residual(x, residual, n, channels, dtype, scale, shift, relu, y, stream)
                  : wcn_bn_apply(x, n, channels, dtype, scale, shift, relu, y, stream);
}

// `z`: the stored output of a residual tail (mask = its sign), or NULL (`relu`: the mask is recomputed from x and the forward's
// scale / shift).  `sums`: [2][channels] = sum_dy (bias gradient) | sum_dy_xhat (weight gradient).  `dx` NULL: sums only.
// `training` 0: the statistics were constants (eval mode) - dx without the mean terms.
// `dy_ld`: row pitch of dy in elements (>= channels; 0 = channels) - a column slice of a wider row-major tensor, e.g. the
// gradient a channel concatenation hands to one of its inputs, is read where it is instead of through a contiguous copy.
int wcn_bn_train_backward_ld(const void* dy, int64_t dy_ld, const void* x, const void* z, int32_t relu, int64_t n, int32_t channels,
                             int32_t dtype, const float* stats, const float* gamma, int32_t training, float* sums, void* dx,
                             void* dres, void* workspace, size_t workspace_bytes, wcn_stream_t stream) {
  if (!stats || !sums || channels < 1 || (dy_ld != 0 && dy_ld < channels)) return WCN_ERROR_INVALID_PARAMETERS;
  const float* mean = stats;
  const float* rstd = stats + channels;
  const float* rsc = (relu && !z) ? stats + 2 * (int64_t)channels : nullptr;
  const float* rsh = (relu && !z) ? stats + 3 * (int64_t)channels : nullptr;
  float* sum_dy = sums;
  float* sum_dy_xhat = sums + channels;
  const bool masked = z && relu;
  int rc = masked ? bn_backward_reduce_masked_ld(dy, dy_ld, x, z, n, channels, dtype, mean, rstd, sum_dy, sum_dy_xhat, workspace,
                                                 workspace_bytes, stream)
                  : bn_backward_reduce_ld(dy, dy_ld, x, rsc, rsh, n, channels, dtype, mean, rstd, sum_dy, sum_dy_xhat, workspace,
                                          workspace_bytes, stream);
  if (rc != WCN_SUCCESS || !dx) return rc;
  const float* a0 = sum_dy;
  const float* a1 = sum_dy_xhat;
  if (!training) {  // constants: the second half of the workspace is zeroed and stands in for the sums
    float* zeros = reinterpret_cast<float*>(workspace);
    if (hipMemsetAsync(zeros, 0, (size_t)channels * sizeof(float), (hipStream_t)stream) != hipSuccess)
      return WCN_ERROR_KERNEL_EXECUTION;
    a0 = a1 = zeros;
  }
  return masked ? bn_backward_apply_masked_ld(dy, dy_ld, x, z, n, channels, dtype, mean, rstd, gamma, a0, a1, dx, dres, stream)
                : bn_backward_apply_ld(dy, dy_ld, x, rsc, rsh, n, channels, dtype, mean, rstd, gamma, a0, a1, dx, stream);
}

int wcn_bn_train_backward(const void* dy, const void* x, const void* z, int32_t relu, int64_t n, int32_t channels, int32_t dtype,
                          const float* stats, const float* gamma, int32_t training, float* sums, void* dx, void* dres,
                          void* workspace, size_t workspace_bytes, wcn_stream_t stream) {
  return wcn_bn_train_backward_ld(dy, 0, x, z, relu, n, channels, dtype, stats, gamma, training, sums, dx, dres, workspace,
                                  workspace_bytes, stream);
}

}  // extern "C"
