// points.hip - point-cloud front-end kernels of the PointConv path:
//
//   wcn_knn_grid       exact k-nearest-neighbour search over a uniform cell grid.  The caller sorts the reference
//                      points by cell id (points of a cell are contiguous, cell_start[] is the CSR over cells); one
//                      thread per query walks the cell shells around its own cell, keeps the k best candidates in an
//                      insertion-sorted register list and stops as soon as the k-th distance is not larger than the
//                      distance to the boundary of the searched cube - every point outside is then farther away, so the
//                      result equals the brute-force answer (reference: chunked cdist + topk,
//                      warpconvnet/geometry/coords/search/knn.py:11-26, 108-142, O(M*N); this is O(M*k)).
//   wcn_segment_reduce out[m][c] = reduce over rows [row_splits[m], row_splits[m+1]) of in[.][c], reduce in
//                      {sum, mean, max, min}; max/min also return the arg row (first extremum) for the backward pass
//                      (role of torch_scatter.segment_csr in warpconvnet/ops/reductions.py:36-75).
#include <hip/hip_bf16.h>
#include <hip/hip_fp16.h>

#include <cfloat>

#include "wcn_common.h"

namespace wcn {

constexpr int kKnnMaxK = 64;

template <int KMAX>
__global__ __launch_bounds__(128) void knn_grid_kernel(const float* __restrict__ ref, const int32_t* __restrict__ ref_id,
                                                       const int32_t* __restrict__ cell_start, float ox, float oy,
                                                       float oz, float inv_h, float h, int gx, int gy, int gz,
                                                       const float* __restrict__ query, int64_t m, int k,
                                                       int64_t* __restrict__ out_idx, float* __restrict__ out_d2) {
  const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= m) return;
  const float qx = query[q * 3 + 0], qy = query[q * 3 + 1], qz = query[q * 3 + 2];
  int cx = (int)floorf((qx - ox) * inv_h), cy = (int)floorf((qy - oy) * inv_h), cz = (int)floorf((qz - oz) * inv_h);
  cx = cx < 0 ? 0 : (cx >= gx ? gx - 1 : cx);
  cy = cy < 0 ? 0 : (cy >= gy ? gy - 1 : cy);
  cz = cz < 0 ? 0 : (cz >= gz ? gz - 1 : cz);
  float bd[KMAX];
  int32_t bi[KMAX];
#pragma unroll
  for (int j = 0; j < KMAX; ++j) { bd[j] = FLT_MAX; bi[j] = -1; }
  float kth = FLT_MAX;  // current k-th best squared distance (FLT_MAX until k candidates were seen)
  auto consider = [&](float d2, int32_t id) {
    if (d2 >= kth) return;
    // insertion into the ascending list; indices are compile-time so the list stays in registers
    float cd = d2;
    int32_t ci = id;
#pragma unroll
    for (int j = 0; j < KMAX; ++j) {
      if (j < k && cd < bd[j]) {
        const float td = bd[j]; const int32_t ti = bi[j];
        bd[j] = cd; bi[j] = ci;
        cd = td; ci = ti;
      }
      if (j == k - 1) kth = bd[j];
    }
  };
  const int rmax = max(max(gx, gy), gz);
  for (int r = 0; r <= rmax; ++r) {
    // shell r: cells with max(|dx|, |dy|, |dz|) == r
    for (int dz = -r; dz <= r; ++dz) {
      const int z = cz + dz;
      if (z < 0 || z >= gz) continue;
      for (int dy = -r; dy <= r; ++dy) {
        const int y = cy + dy;
        if (y < 0 || y >= gy) continue;
        const bool face = (dz == -r || dz == r || dy == -r || dy == r);
        const int step = face ? 1 : (r == 0 ? 1 : 2 * r);  // interior rows of the shell: only the two end cells
        for (int dx = -r; dx <= r; dx += step) {
          const int x = cx + dx;
          if (x < 0 || x >= gx) continue;
          const int cell = (z * gy + y) * gx + x;
          const int p0 = cell_start[cell], p1 = cell_start[cell + 1];
          for (int p = p0; p < p1; ++p) {
            const float ex = ref[p * 3 + 0] - qx, ey = ref[p * 3 + 1] - qy, ez = ref[p * 3 + 2] - qz;
            consider(ex * ex + ey * ey + ez * ez, ref_id[p]);
          }
        }
      }
    }
    // distance from the query to the boundary of the searched cube [c - r, c + r]: everything outside is farther
    const float lox = ox + (cx - r) * h, hix = ox + (cx + r + 1) * h;
    const float loy = oy + (cy - r) * h, hiy = oy + (cy + r + 1) * h;
    const float loz = oz + (cz - r) * h, hiz = oz + (cz + r + 1) * h;
    float b = fminf(fminf(qx - lox, hix - qx), fminf(fminf(qy - loy, hiy - qy), fminf(qz - loz, hiz - qz)));
    const bool covers = (cx - r <= 0 && cx + r >= gx - 1 && cy - r <= 0 && cy + r >= gy - 1 && cz - r <= 0 && cz + r >= gz - 1);
    if (covers) break;
    if (b > 0.f && kth <= b * b) break;
  }
#pragma unroll
  for (int j = 0; j < KMAX; ++j) {
    if (j < k) {
      out_idx[q * k + j] = bi[j];
      if (out_d2) out_d2[q * k + j] = bd[j];
    }
  }
}

int knn_grid(const float* ref, const int32_t* ref_id, const int32_t* cell_start, const float origin[3], float h,
             const int32_t dims[3], const float* query, int64_t m, int k, int64_t* out_idx, float* out_d2, hipStream_t s) {
  if (k < 1 || k > kKnnMaxK || h <= 0.f || dims[0] < 1 || dims[1] < 1 || dims[2] < 1) return WCN_ERROR_INVALID_PARAMETERS;
  if (m == 0) return WCN_SUCCESS;
  const dim3 grid((unsigned)ceil_div(m, 128)), block(128);
#define WCN_KNN(KM)                                                                                                      \
  hipLaunchKernelGGL(knn_grid_kernel<KM>, grid, block, 0, s, ref, ref_id, cell_start, origin[0], origin[1], origin[2],   \
                     1.0f / h, h, dims[0], dims[1], dims[2], query, m, k, out_idx, out_d2)
  if (k <= 8) WCN_KNN(8);
  else if (k <= 16) WCN_KNN(16);
  else if (k <= 32) WCN_KNN(32);
  else WCN_KNN(64);
#undef WCN_KNN
  return launch_status();
}

// ---- radius search -----------------------------------------------------------------------------------------------------
// Cell-list radius search over the same sorted-by-cell layout as the kNN: cell size >= radius, so the 27 cells around the
// query's cell hold every point within the radius (reference: warpconvnet/csrc/radius_search_kernels.cu:30-134 - a
// hash-table cell list with the same 27-cell walk and the same `dist^2 <= radius^2` test; two passes: count, then
// write at the exclusive scan of the counts).  Rows come out in cell-walk order (z, y, x ascending), points of one cell
// in ascending original index (the cell sort is stable) - deterministic, unlike the reference's argsort order.
template <bool WRITE>
__global__ __launch_bounds__(128) void radius_grid_kernel(const float* __restrict__ ref, const int32_t* __restrict__ ref_id,
                                                          const int32_t* __restrict__ cell_start, float ox, float oy,
                                                          float oz, float inv_h, int gx, int gy, int gz,
                                                          const float* __restrict__ query, int64_t m, float r2,
                                                          int32_t* __restrict__ counts, const int64_t* __restrict__ splits,
                                                          int32_t* __restrict__ out_idx, float* __restrict__ out_dist) {
  const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= m) return;
  const float qx = query[q * 3 + 0], qy = query[q * 3 + 1], qz = query[q * 3 + 2];
  // queries may lie outside the grid: clamp the CELL RANGE, not the cell (float -> int conversion saturates)
  const float fx = floorf((qx - ox) * inv_h), fy = floorf((qy - oy) * inv_h), fz = floorf((qz - oz) * inv_h);
  const float big = 1.0e9f;
  const int cx = (int)fminf(fmaxf(fx, -big), big), cy = (int)fminf(fmaxf(fy, -big), big), cz = (int)fminf(fmaxf(fz, -big), big);
  const int x0 = max(cx - 1, 0), x1 = min(cx + 1, gx - 1);
  const int y0 = max(cy - 1, 0), y1 = min(cy + 1, gy - 1);
  const int z0 = max(cz - 1, 0), z1 = min(cz + 1, gz - 1);
  int n = 0;
  int64_t at = WRITE ? splits[q] : 0;
  for (int z = z0; z <= z1; ++z) {
    for (int y = y0; y <= y1; ++y) {
      if (x0 > x1) continue;
      // the cells x0..x1 of one (y, z) row are contiguous in the sorted array: one range instead of three
      const int row = (z * gy + y) * gx;
      const int p0 = cell_start[row + x0], p1 = cell_start[row + x1 + 1];
      for (int p = p0; p < p1; ++p) {
        const float ex = qx - ref[p * 3 + 0], ey = qy - ref[p * 3 + 1], ez = qz - ref[p * 3 + 2];
        const float d2 = ex * ex + ey * ey + ez * ez;
        if (d2 <= r2) {
          if (WRITE) {
            out_idx[at] = ref_id[p];
            if (out_dist) out_dist[at] = sqrtf(d2);
            ++at;
          }
          ++n;
        }
      }
    }
  }
  if (!WRITE) counts[q] = n;
}

static int radius_args_ok(float h, const int32_t dims[3], float radius) {
  return h > 0.f && radius >= 0.f && radius <= h && dims[0] >= 1 && dims[1] >= 1 && dims[2] >= 1;
}

int radius_grid_count(const float* ref, const int32_t* ref_id, const int32_t* cell_start, const float origin[3], float h,
                      const int32_t dims[3], const float* query, int64_t m, float radius, int32_t* counts, hipStream_t s) {
  if (!radius_args_ok(h, dims, radius)) return WCN_ERROR_INVALID_PARAMETERS;
  if (m == 0) return WCN_SUCCESS;
  hipLaunchKernelGGL(radius_grid_kernel<false>, dim3((unsigned)ceil_div(m, 128)), dim3(128), 0, s, ref, ref_id, cell_start,
                     origin[0], origin[1], origin[2], 1.0f / h, dims[0], dims[1], dims[2], query, m, radius * radius, counts,
                     (const int64_t*)nullptr, (int32_t*)nullptr, (float*)nullptr);
  return launch_status();
}

int radius_grid_write(const float* ref, const int32_t* ref_id, const int32_t* cell_start, const float origin[3], float h,
                      const int32_t dims[3], const float* query, int64_t m, float radius, const int64_t* splits,
                      int32_t* out_idx, float* out_dist, hipStream_t s) {
  if (!radius_args_ok(h, dims, radius)) return WCN_ERROR_INVALID_PARAMETERS;
  if (m == 0) return WCN_SUCCESS;
  hipLaunchKernelGGL(radius_grid_kernel<true>, dim3((unsigned)ceil_div(m, 128)), dim3(128), 0, s, ref, ref_id, cell_start,
                     origin[0], origin[1], origin[2], 1.0f / h, dims[0], dims[1], dims[2], query, m, radius * radius,
                     (int32_t*)nullptr, splits, out_idx, out_dist);
  return launch_status();
}

// ---- segment reduce ---------------------------------------------------------------------------------------------------
enum { kRedSum = 0, kRedMean = 1, kRedMax = 2, kRedMin = 3 };

template <typename T> struct PtCvt;
template <> struct PtCvt<float> {
  static __device__ __forceinline__ float ld(float v) { return v; }
  static __device__ __forceinline__ float st(float v) { return v; }
};
template <> struct PtCvt<__half> {
  static __device__ __forceinline__ float ld(__half v) { return __half2float(v); }
  static __device__ __forceinline__ __half st(float v) { return __float2half(v); }
};
template <> struct PtCvt<__hip_bfloat16> {
  static __device__ __forceinline__ float ld(__hip_bfloat16 v) { return __bfloat162float(v); }
  static __device__ __forceinline__ __hip_bfloat16 st(float v) { return __float2bfloat16(v); }
};

// one thread per (segment, channel); adjacent threads = adjacent channels => every step of the loop reads one row
// contiguously.  fp32 accumulation in ascending row order => deterministic.
template <typename T>
__global__ __launch_bounds__(256) void segment_reduce_kernel(const T* __restrict__ in, const int64_t* __restrict__ splits,
                                                             int64_t m, int c, int op, T* __restrict__ out,
                                                             int64_t* __restrict__ arg) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= m * c) return;
  const int64_t seg = e / c;
  const int ch = (int)(e % c);
  const int64_t r0 = splits[seg], r1 = splits[seg + 1];
  float acc = (op == kRedMax) ? -FLT_MAX : (op == kRedMin ? FLT_MAX : 0.f);
  int64_t best = -1;
  for (int64_t r = r0; r < r1; ++r) {
    const float v = PtCvt<T>::ld(in[r * c + ch]);
    if (op == kRedMax) { if (v > acc || best < 0) { acc = v; best = r; } }
    else if (op == kRedMin) { if (v < acc || best < 0) { acc = v; best = r; } }
    else acc += v;
  }
  if (op == kRedMean && r1 > r0) acc /= (float)(r1 - r0);
  if ((op == kRedMax || op == kRedMin) && r1 == r0) acc = 0.f;  // empty segment: zero (torch_scatter convention)
  out[e] = PtCvt<T>::st(acc);
  if (arg) arg[e] = best;
}

int segment_reduce(const void* in, const int64_t* splits, int64_t m, int c, int dtype, int op, void* out, int64_t* arg,
                   hipStream_t s) {
  if (op < kRedSum || op > kRedMin) return WCN_ERROR_INVALID_PARAMETERS;
  if (m == 0 || c == 0) return WCN_SUCCESS;
  const dim3 grid((unsigned)ceil_div(m * c, 256)), block(256);
  switch (dtype) {
    case WCN_F32:
      hipLaunchKernelGGL(segment_reduce_kernel<float>, grid, block, 0, s, (const float*)in, splits, m, c, op, (float*)out, arg);
      break;
    case WCN_F16:
      hipLaunchKernelGGL(segment_reduce_kernel<__half>, grid, block, 0, s, (const __half*)in, splits, m, c, op, (__half*)out, arg);
      break;
    case WCN_BF16:
      hipLaunchKernelGGL(segment_reduce_kernel<__hip_bfloat16>, grid, block, 0, s, (const __hip_bfloat16*)in, splits, m, c,
                         op, (__hip_bfloat16*)out, arg);
      break;
    default: return WCN_ERROR_UNSUPPORTED_CONFIG;
  }
  return launch_status();
}

}  // namespace wcn
