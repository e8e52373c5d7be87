// conv_api.hip - C-ABI entry points of the sparse-conv GEMMs (include/wcn.h); dispatch only.
#include "wcn_common.h"

namespace wcn {
// conv_ref.hip
int conv_gather_gemm_ref(const void* in, const void* w, void* out, const int32_t* nbr, const float* bias, int64_t n_out,
                         int cin, int cout, int K, int dtype, int w_transposed, int k_flip, hipStream_t s);
size_t colsum_workspace(int c);
int colsum(const void* in, int64_t n, int c, int dtype, float* out, void* workspace, size_t workspace_bytes, hipStream_t s);
int conv_wgrad_ref(const void* x, const void* dy, float* dw, const int32_t* in_maps, const int32_t* out_maps,
                   const int32_t* offsets, int cin, int cout, int K, int dtype, hipStream_t s);
// conv_mfma.hip
bool mfma_gather_supported(int cin, int cout, int K, int dtype);
bool gather_gemm_cs_supported(int cin, int cout, int K, int dtype);  // conv_mfma_cs.hip
int conv_gather_gemm_mfma(const void* in, const void* wp, void* out, const int32_t* nbr, const uint32_t* mask,
                          const int32_t* perm, const ConvEpilogue& epi, int64_t n_out, int cin, int cout, int K, int dtype,
                          float* out32, hipStream_t s);
int pack_weight_mfma(const void* w, int K, int cin, int cout, int dtype, int transpose, int flip, void* packed,
                     hipStream_t s);
int pack_weight_cs_pair(const float* w, int K, int cin, int cout, int dtype, int flip_dgrad, void* packed_fwd, void* packed_dgrad,
                        hipStream_t s);
int pack_weight_mfma_f32(const float* w, int K, int cin, int cout, int dtype, int transpose, int flip, void* packed,
                         hipStream_t s);
bool mfma_grouped_supported(int cin, int cout, int K, int dtype);
int conv_gather_gemm_grouped(const void* in, const void* wp, void* out, const int32_t* nbr, const uint32_t* mask,
                             const int32_t* perm, const ConvEpilogue& epi, int64_t n_out, int cin, int cout, int groups,
                             int K, int dtype, hipStream_t s);
int pack_weight_grouped(const void* w, int w_is_f32, int K, int groups, int cin, int cout, int dtype, int transpose, int flip,
                        void* packed, hipStream_t s);
// wgrad_mfma.hip
bool mfma_wgrad_supported(int cin, int cout, int dtype);
size_t wgrad_mfma_workspace(int K, int cin, int cout);
bool mfma_wgrad_bias_supported(int cin, int cout, int dtype);
int conv_wgrad_mfma(const void* x, const void* dy, float* dw, const int32_t* in_maps, const int32_t* out_maps,
                    const int32_t* offsets, int cin, int cout, int K, int dtype, void* workspace, size_t workspace_bytes,
                    int cs_k, float* bias_grad, int64_t pair_bound, hipStream_t s);
// dwconv.hip
int dwconv_gather(const void* in, const void* w, void* out, const int32_t* tbl, const float* bias, int64_t n_out, int C,
                  int K, int dtype, int k_flip, hipStream_t s);
size_t dwconv_wgrad_workspace(int K, int C);
int dwconv_wgrad(const void* x, const void* dy, float* dw, const int32_t* in_maps, const int32_t* out_maps,
                 const int32_t* offsets, int C, int K, int dtype, void* workspace, size_t workspace_bytes, hipStream_t s);
// points.hip
int knn_grid(const float* ref, const int32_t* ref_id, const int32_t* cell_start, const float origin[3], float h,
             const int32_t dims[3], const float* query, int64_t m, int k, int64_t* out_idx, float* out_d2, hipStream_t s);
int radius_grid_count(const float* ref, const int32_t* ref_id, const int32_t* cell_start, const float origin[3], float h,
                      const int32_t dims[3], const float* query, int64_t m, float radius, int32_t* counts, hipStream_t s);
int radius_grid_write(const float* ref, const int32_t* ref_id, const int32_t* cell_start, const float origin[3], float h,
                      const int32_t dims[3], const float* query, int64_t m, float radius, const int64_t* splits,
                      int32_t* out_idx, float* out_dist, hipStream_t s);
int segment_reduce(const void* in, const int64_t* splits, int64_t m, int c, int dtype, int op, void* out, int64_t* arg,
                   hipStream_t s);
}  // namespace wcn

using namespace wcn;

static inline size_t dtype_size(int dtype) { return dtype == WCN_F32 ? 4 : 2; }
static inline bool dtype_ok(int dtype) { return dtype == WCN_F32 || dtype == WCN_F16 || dtype == WCN_BF16; }

extern "C" {

int wcn_mfma_gather_supported(int32_t cin, int32_t cout, int32_t num_offsets, int32_t dtype) {
  return mfma_gather_supported(cin, cout, num_offsets, dtype) ? 1 : 0;
}
int wcn_conv_identity_supported(int32_t cin, int32_t cout, int32_t dtype) {
  return gather_gemm_cs_supported(cin, cout, 1, dtype) ? 1 : 0;
}
int wcn_mfma_wgrad_supported(int32_t cin, int32_t cout, int32_t dtype) {
  return mfma_wgrad_supported(cin, cout, dtype) ? 1 : 0;
}

size_t wcn_packed_weight_bytes(int32_t num_offsets, int32_t cin, int32_t cout, int32_t dtype, int32_t transpose) {
  (void)transpose;  // cin / cout are the kernel-side roles already (reduce over cin, produce cout)
  if (num_offsets < 1 || cin < 1 || cout < 1 || !dtype_ok(dtype)) return 0;
  // the channel-split kernels reduce in 64-channel chunks: a trailing 32-channel chunk is zero-padded in the image
  return (size_t)num_offsets * ((cin + 63) / 64 * 64) * cout * dtype_size(dtype);
}

int wcn_pack_weight(const void* w, int32_t num_offsets, int32_t cin, int32_t cout, int32_t dtype, int32_t transpose,
                    int32_t flip, void* packed, size_t packed_bytes, wcn_stream_t stream) {
  if (!w || !packed || num_offsets < 1 || cin < 1 || cout < 1) return WCN_ERROR_INVALID_PARAMETERS;
  if (packed_bytes < wcn_packed_weight_bytes(num_offsets, cin, cout, dtype, transpose)) return WCN_ERROR_INVALID_PARAMETERS;
  return pack_weight_mfma(w, num_offsets, cin, cout, dtype, transpose, flip, packed, (hipStream_t)stream);
}

int wcn_pack_weight_f32(const float* w, int32_t num_offsets, int32_t cin, int32_t cout, int32_t dtype, int32_t transpose,
                        int32_t flip, void* packed, size_t packed_bytes, wcn_stream_t stream) {
  if (!w || !packed || num_offsets < 1 || cin < 1 || cout < 1) return WCN_ERROR_INVALID_PARAMETERS;
  if (packed_bytes < wcn_packed_weight_bytes(num_offsets, cin, cout, dtype, transpose)) return WCN_ERROR_INVALID_PARAMETERS;
  return pack_weight_mfma_f32(w, num_offsets, cin, cout, dtype, transpose, flip, packed, (hipStream_t)stream);
}

// COMPACT tables (wcn_kmap_build_binned with compact = 1: 16 ints per row, the mask first): the channel-split gather kernels take
// them with `mask` = NULL and expand the rows into their index slab.
static bool compact_table_ok(int cin, int cout, int K, int dtype) {
  return wcn_kmap_compact_supported(K) && gather_gemm_cs_supported(cin, cout, K, dtype);
}
int wcn_conv_compact_table_supported(int32_t cin, int32_t cout, int32_t num_offsets, int32_t dtype) {
  return compact_table_ok(cin, cout, num_offsets, dtype) ? 1 : 0;
}

int wcn_pack_weight_pair_supported(int32_t num_offsets, int32_t cin, int32_t cout, int32_t dtype) {
  return (gather_gemm_cs_supported(cin, cout, num_offsets, dtype) && gather_gemm_cs_supported(cout, cin, num_offsets, dtype)) ? 1 : 0;
}

int wcn_pack_weight_f32_pair(const float* w, int32_t num_offsets, int32_t cin, int32_t cout, int32_t dtype, int32_t flip_dgrad,
                             void* packed_fwd, size_t packed_fwd_bytes, void* packed_dgrad, size_t packed_dgrad_bytes,
                             wcn_stream_t stream) {
  if (!w || !packed_fwd || !packed_dgrad || num_offsets < 1 || cin < 1 || cout < 1) return WCN_ERROR_INVALID_PARAMETERS;
  if (!wcn_pack_weight_pair_supported(num_offsets, cin, cout, dtype)) return WCN_ERROR_UNSUPPORTED_CONFIG;
  if (packed_fwd_bytes < wcn_packed_weight_bytes(num_offsets, cin, cout, dtype, 0) ||
      packed_dgrad_bytes < wcn_packed_weight_bytes(num_offsets, cout, cin, dtype, 1))
    return WCN_ERROR_INVALID_PARAMETERS;
  return pack_weight_cs_pair(w, num_offsets, cin, cout, dtype, flip_dgrad, packed_fwd, packed_dgrad, (hipStream_t)stream);
}

int wcn_conv_gather_gemm(const void* in, const void* w, void* out, const int32_t* nbr, const uint32_t* mask,
                         const int32_t* perm, const float* bias, int64_t n_in, int64_t n_out, int32_t cin,
                         int32_t cout, int32_t num_offsets, int32_t dtype, int32_t algo, int32_t w_transposed,
                         int32_t k_flip, wcn_stream_t stream) {
  if (n_in < 0 || n_out < 0 || cin < 1 || cout < 1 || num_offsets < 1 || !dtype_ok(dtype))
    return WCN_ERROR_INVALID_PARAMETERS;
  if (n_out == 0) return WCN_SUCCESS;
  // nbr == mask == NULL with one offset: the identity map (row r pairs with row r) - the dense product of a 1 x 1 x 1
  // convolution streamed through the channel-split gather kernel (shapes of wcn_conv_identity_supported)
  const bool identity = !nbr && !mask && !perm && num_offsets == 1 && algo == WCN_ALGO_MFMA && n_in == n_out &&
                        gather_gemm_cs_supported(cin, cout, 1, dtype);
  if (!w || !out || (!nbr && !identity) || (n_in > 0 && !in)) return WCN_ERROR_INVALID_PARAMETERS;
  hipStream_t s = (hipStream_t)stream;
  switch (algo) {
    case WCN_ALGO_REF:
      return conv_gather_gemm_ref(in, w, out, nbr, bias, n_out, cin, cout, num_offsets, dtype, w_transposed, k_flip, s);
    case WCN_ALGO_MFMA:
      // `w` must be the packed image (wcn_pack_weight already applied transpose / flip)
      if (!mask && !identity && !(nbr && compact_table_ok(cin, cout, num_offsets, dtype))) return WCN_ERROR_INVALID_PARAMETERS;
      {
        ConvEpilogue epi;
        epi.bias = bias;
        return conv_gather_gemm_mfma(in, w, out, nbr, mask, perm, epi, n_out, cin, cout, num_offsets, dtype, nullptr, s);
      }
    default:
      // AUTO cannot be resolved here because the two algorithms take different weight images.
      return WCN_ERROR_INVALID_PARAMETERS;
  }
}

int wcn_mfma_grouped_supported(int32_t cin_g, int32_t cout_g, int32_t num_offsets, int32_t dtype) {
  return mfma_grouped_supported(cin_g, cout_g, num_offsets, dtype) ? 1 : 0;
}

int wcn_pack_weight_grouped(const void* w, int32_t w_is_f32, int32_t num_offsets, int32_t groups, int32_t cin_g,
                            int32_t cout_g, int32_t dtype, int32_t transpose, int32_t flip, void* packed, wcn_stream_t stream) {
  if (!w || !packed || num_offsets < 1 || groups < 1 || cin_g < 1 || cout_g < 1) return WCN_ERROR_INVALID_PARAMETERS;
  return pack_weight_grouped(w, w_is_f32, num_offsets, groups, cin_g, cout_g, dtype, transpose, flip, packed,
                             (hipStream_t)stream);
}

int wcn_conv_gather_gemm_grouped(const void* in, const void* w_packed, void* out, const int32_t* nbr, const uint32_t* mask,
                                 const int32_t* perm, const float* bias, int64_t n_in, int64_t n_out, int32_t cin_g,
                                 int32_t cout_g, int32_t groups, int32_t num_offsets, int32_t dtype, wcn_stream_t stream) {
  if (n_in < 0 || n_out < 0 || cin_g < 1 || cout_g < 1 || groups < 1 || num_offsets < 1) return WCN_ERROR_INVALID_PARAMETERS;
  if (n_out == 0) return WCN_SUCCESS;
  if (!w_packed || !out || !nbr || !mask || (n_in > 0 && !in)) return WCN_ERROR_INVALID_PARAMETERS;
  ConvEpilogue epi;
  epi.bias = bias;
  return conv_gather_gemm_grouped(in, w_packed, out, nbr, mask, perm, epi, n_out, cin_g, cout_g, groups, num_offsets, dtype,
                                  (hipStream_t)stream);
}

int wcn_conv_gather_gemm_f32out(const void* in, const void* w, float* out, const int32_t* nbr, const uint32_t* mask,
                                const int32_t* perm, const float* bias, int64_t n_in, int64_t n_out, int32_t cin,
                                int32_t cout, int32_t num_offsets, int32_t dtype, wcn_stream_t stream) {
  if (n_in < 0 || n_out < 0 || cin < 1 || cout < 1 || num_offsets < 1 || (dtype != WCN_F16 && dtype != WCN_BF16))
    return WCN_ERROR_INVALID_PARAMETERS;
  if (n_out == 0) return WCN_SUCCESS;
  if (!w || !out || !nbr || (!mask && !compact_table_ok(cin, cout, num_offsets, dtype)) || (n_in > 0 && !in))
    return WCN_ERROR_INVALID_PARAMETERS;
  ConvEpilogue epi;
  epi.bias = bias;
  return conv_gather_gemm_mfma(in, w, nullptr, nbr, mask, perm, epi, n_out, cin, cout, num_offsets, dtype, out,
                               (hipStream_t)stream);
}

int wcn_conv_gather_gemm_fused(const void* in, const void* w_packed, void* out, const int32_t* nbr, const uint32_t* mask,
                               const int32_t* perm, const float* bias, const float* scale, const float* shift,
                               const void* residual, int32_t relu, int64_t n_in, int64_t n_out, int32_t cin, int32_t cout,
                               int32_t num_offsets, int32_t dtype, wcn_stream_t stream) {
  if (n_in < 0 || n_out < 0 || cin < 1 || cout < 1 || num_offsets < 1 || (dtype != WCN_F16 && dtype != WCN_BF16) ||
      ((scale == nullptr) != (shift == nullptr)))
    return WCN_ERROR_INVALID_PARAMETERS;
  if (n_out == 0) return WCN_SUCCESS;
  if (!w_packed || !out || !nbr || (!mask && !compact_table_ok(cin, cout, num_offsets, dtype)) || (n_in > 0 && !in) || residual == out)
    return WCN_ERROR_INVALID_PARAMETERS;
  ConvEpilogue epi;
  epi.bias = bias; epi.scale = scale; epi.shift = shift; epi.residual = residual; epi.relu = relu ? 1 : 0;
  return conv_gather_gemm_mfma(in, w_packed, out, nbr, mask, perm, epi, n_out, cin, cout, num_offsets, dtype, nullptr,
                               (hipStream_t)stream);
}

size_t wcn_colsum_workspace(int32_t channels) { return channels > 0 ? colsum_workspace(channels) : 0; }

int wcn_colsum(const void* in, int64_t n, int32_t channels, int32_t dtype, float* out, void* workspace,
               size_t workspace_bytes, wcn_stream_t stream) {
  if (n < 0 || channels < 1 || !dtype_ok(dtype) || !out) return WCN_ERROR_INVALID_PARAMETERS;
  if (n > 0 && !in) return WCN_ERROR_INVALID_PARAMETERS;
  return colsum(in, n, channels, dtype, out, workspace, workspace_bytes, (hipStream_t)stream);
}

size_t wcn_conv_wgrad_workspace(int32_t num_offsets, int32_t cin, int32_t cout, int32_t algo) {
  if (algo == WCN_ALGO_MFMA) return wgrad_mfma_workspace(num_offsets, cin, cout);
  return 256;
}

int wcn_conv_wgrad(const void* x, const void* dy, float* dw, const int32_t* in_maps, const int32_t* out_maps,
                   const int32_t* offsets, int64_t n_in, int64_t n_out, int32_t cin, int32_t cout, int32_t num_offsets,
                   int32_t dtype, int32_t algo, void* workspace, size_t workspace_bytes, wcn_stream_t stream) {
  if (n_in < 0 || n_out < 0 || cin < 1 || cout < 1 || num_offsets < 1 || !dtype_ok(dtype) || !dw || !offsets)
    return WCN_ERROR_INVALID_PARAMETERS;
  hipStream_t s = (hipStream_t)stream;
  switch (algo) {
    case WCN_ALGO_REF:
      return conv_wgrad_ref(x, dy, dw, in_maps, out_maps, offsets, cin, cout, num_offsets, dtype, s);
    case WCN_ALGO_MFMA:
      return conv_wgrad_mfma(x, dy, dw, in_maps, out_maps, offsets, cin, cout, num_offsets, dtype, workspace,
                             workspace_bytes, -1, nullptr, (int64_t)num_offsets * (n_in < n_out ? n_in : n_out), s);
    default:
      return WCN_ERROR_INVALID_PARAMETERS;
  }
}

// Layer entry: the backward of SparseConv3d -> BatchNorm (-> ReLU / residual tail) in one call - BatchNorm reduce + apply
// (wcn_bn_train_backward), ABt dgrad on the reverse tables, AtB wgrad.  The host side of a network pays per call.
int wcn_conv_bn_backward(const void* grad_out, const void* x, const void* y, const void* z, int32_t relu, const float* stats,
                         const float* gamma, int32_t training, float* sums, void* dy_conv, void* dres, const void* w_packed_dgrad,
                         const int32_t* rev_nbr, const uint32_t* rev_mask, const int32_t* rev_perm, int32_t flip, void* dx,
                         const int32_t* in_maps, const int32_t* out_maps, const int32_t* offsets, float* dw, void* wgrad_workspace,
                         size_t wgrad_workspace_bytes, int64_t n_in, int64_t n_out, int32_t cin, int32_t cout, int32_t num_offsets,
                         int32_t dtype, void* bn_workspace, size_t bn_workspace_bytes, wcn_stream_t stream) {
  return wcn_conv_bn_backward_ld(grad_out, 0, x, y, z, relu, stats, gamma, training, sums, dy_conv, dres, w_packed_dgrad, rev_nbr,
                                 rev_mask, rev_perm, flip, dx, in_maps, out_maps, offsets, dw, wgrad_workspace, wgrad_workspace_bytes,
                                 n_in, n_out, cin, cout, num_offsets, dtype, bn_workspace, bn_workspace_bytes, stream);
}

int wcn_conv_bn_backward_ld(const void* grad_out, int64_t grad_out_ld, const void* x, const void* y, const void* z, int32_t relu,
                            const float* stats, const float* gamma, int32_t training, float* sums, void* dy_conv, void* dres,
                            const void* w_packed_dgrad, const int32_t* rev_nbr, const uint32_t* rev_mask, const int32_t* rev_perm,
                            int32_t flip, void* dx, const int32_t* in_maps, const int32_t* out_maps, const int32_t* offsets,
                            float* dw, void* wgrad_workspace, size_t wgrad_workspace_bytes, int64_t n_in, int64_t n_out,
                            int32_t cin, int32_t cout, int32_t num_offsets, int32_t dtype, void* bn_workspace,
                            size_t bn_workspace_bytes, wcn_stream_t stream) {
  if (!dy_conv || (dtype != WCN_F16 && dtype != WCN_BF16)) return WCN_ERROR_INVALID_PARAMETERS;
  int rc = wcn_bn_train_backward_ld(grad_out, grad_out_ld, y, z, relu, n_out, cout, dtype, stats, gamma, training, sums, dy_conv, dres,
                                    bn_workspace, bn_workspace_bytes, stream);
  if (rc != WCN_SUCCESS) return rc;
  if (dx) {
    if (!w_packed_dgrad || !rev_nbr) return WCN_ERROR_INVALID_PARAMETERS;  // (rev_mask NULL: a compact table, wcn_conv_gather_gemm checks)
    rc = wcn_conv_gather_gemm(dy_conv, w_packed_dgrad, dx, rev_nbr, rev_mask, rev_perm, nullptr, n_out, n_in, cout, cin, num_offsets,
                              dtype, WCN_ALGO_MFMA, 1, flip, stream);
    if (rc != WCN_SUCCESS) return rc;
  }
  if (dw)
    rc = wcn_conv_wgrad(x, dy_conv, dw, in_maps, out_maps, offsets, n_in, n_out, cin, cout, num_offsets, dtype, WCN_ALGO_MFMA,
                        wgrad_workspace, wgrad_workspace_bytes, stream);
  return rc;
}

int wcn_mfma_wgrad_bias_supported(int32_t cin, int32_t cout, int32_t dtype) {
  return mfma_wgrad_bias_supported(cin, cout, dtype) ? 1 : 0;
}

int wcn_conv_wgrad_bias(const void* x, const void* dy, float* dw, const int32_t* in_maps, const int32_t* out_maps,
                        const int32_t* offsets, int64_t n_in, int64_t n_out, int32_t cin, int32_t cout,
                        int32_t num_offsets, int32_t dtype, int32_t self_offset, float* bias_grad, void* workspace,
                        size_t workspace_bytes, wcn_stream_t stream) {
  if (n_in < 0 || n_out < 0 || cin < 1 || cout < 1 || num_offsets < 1 || !dtype_ok(dtype) || !dw || !offsets || !bias_grad)
    return WCN_ERROR_INVALID_PARAMETERS;
  if (self_offset < 0 || self_offset >= num_offsets) return WCN_ERROR_INVALID_PARAMETERS;
  return conv_wgrad_mfma(x, dy, dw, in_maps, out_maps, offsets, cin, cout, num_offsets, dtype, workspace, workspace_bytes,
                         self_offset, bias_grad, (int64_t)num_offsets * (n_in < n_out ? n_in : n_out), (hipStream_t)stream);
}

int wcn_dwconv_gather(const void* in, const void* w, void* out, const int32_t* nbr, const float* bias, int64_t n_in,
                      int64_t n_out, int32_t channels, int32_t num_offsets, int32_t dtype, int32_t k_flip,
                      wcn_stream_t stream) {
  if (n_in < 0 || n_out < 0 || channels < 1 || num_offsets < 1 || !dtype_ok(dtype)) return WCN_ERROR_INVALID_PARAMETERS;
  if (n_out == 0) return WCN_SUCCESS;
  if (!w || !out || !nbr || (n_in > 0 && !in)) return WCN_ERROR_INVALID_PARAMETERS;
  return dwconv_gather(in, w, out, nbr, bias, n_out, channels, num_offsets, dtype, k_flip, (hipStream_t)stream);
}

size_t wcn_dwconv_wgrad_workspace(int32_t num_offsets, int32_t channels) {
  return (num_offsets > 0 && channels > 0) ? dwconv_wgrad_workspace(num_offsets, channels) : 0;
}

int wcn_dwconv_wgrad(const void* x, const void* dy, float* dw, const int32_t* in_maps, const int32_t* out_maps,
                     const int32_t* offsets, int64_t n_in, int64_t n_out, int32_t channels, int32_t num_offsets,
                     int32_t dtype, void* workspace, size_t workspace_bytes, wcn_stream_t stream) {
  if (n_in < 0 || n_out < 0 || channels < 1 || num_offsets < 1 || !dtype_ok(dtype) || !dw || !offsets)
    return WCN_ERROR_INVALID_PARAMETERS;
  return dwconv_wgrad(x, dy, dw, in_maps, out_maps, offsets, channels, num_offsets, dtype, workspace, workspace_bytes,
                      (hipStream_t)stream);
}

int wcn_knn_grid(const float* ref_sorted, const int32_t* ref_ids, const int32_t* cell_start, const float origin[3],
                 float cell_size, const int32_t dims[3], const float* query, int64_t num_query, int32_t k,
                 int64_t* out_index, float* out_dist2, wcn_stream_t stream) {
  if (num_query < 0 || !origin || !dims) return WCN_ERROR_INVALID_PARAMETERS;
  if (num_query == 0) return WCN_SUCCESS;
  if (!ref_sorted || !ref_ids || !cell_start || !query || !out_index) return WCN_ERROR_INVALID_PARAMETERS;
  return knn_grid(ref_sorted, ref_ids, cell_start, origin, cell_size, dims, query, num_query, k, out_index, out_dist2,
                  (hipStream_t)stream);
}

int wcn_radius_grid_count(const float* ref_sorted, const int32_t* ref_ids, const int32_t* cell_start, const float origin[3],
                          float cell_size, const int32_t dims[3], const float* query, int64_t num_query, float radius,
                          int32_t* counts, wcn_stream_t stream) {
  if (num_query < 0 || !origin || !dims) return WCN_ERROR_INVALID_PARAMETERS;
  if (num_query == 0) return WCN_SUCCESS;
  if (!ref_sorted || !ref_ids || !cell_start || !query || !counts) return WCN_ERROR_INVALID_PARAMETERS;
  return radius_grid_count(ref_sorted, ref_ids, cell_start, origin, cell_size, dims, query, num_query, radius, counts,
                           (hipStream_t)stream);
}

int wcn_radius_grid_write(const float* ref_sorted, const int32_t* ref_ids, const int32_t* cell_start, const float origin[3],
                          float cell_size, const int32_t dims[3], const float* query, int64_t num_query, float radius,
                          const int64_t* splits, int32_t* out_index, float* out_dist, wcn_stream_t stream) {
  if (num_query < 0 || !origin || !dims) return WCN_ERROR_INVALID_PARAMETERS;
  if (num_query == 0) return WCN_SUCCESS;
  if (!ref_sorted || !ref_ids || !cell_start || !query || !splits || !out_index) return WCN_ERROR_INVALID_PARAMETERS;
  return radius_grid_write(ref_sorted, ref_ids, cell_start, origin, cell_size, dims, query, num_query, radius, splits,
                           out_index, out_dist, (hipStream_t)stream);
}

int wcn_segment_reduce(const void* in, const int64_t* row_splits, int64_t num_segments, int32_t channels, int32_t dtype,
                       int32_t op, void* out, int64_t* arg_rows, wcn_stream_t stream) {
  if (num_segments < 0 || channels < 0 || !dtype_ok(dtype)) return WCN_ERROR_INVALID_PARAMETERS;
  if (num_segments == 0 || channels == 0) return WCN_SUCCESS;
  if (!row_splits || !out) return WCN_ERROR_INVALID_PARAMETERS;
  return segment_reduce(in, row_splits, num_segments, channels, dtype, op, out, arg_rows, (hipStream_t)stream);
}

}  // extern "C"
