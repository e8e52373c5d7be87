// mask_sort.hip - launch plan and stand-alone kernels of the mask radix argsort (bodies: mask_sort.h).
#include "mask_sort.h"

namespace wcn {

template <int ROLE>
__global__ __launch_bounds__(kRsThreads) void rs_kernel(RsArgs a) {
  extern __shared__ char s_rs[];
  rs_body<ROLE>(a, blockIdx.x, s_rs);
}

static size_t al256(size_t v) { return (v + 255) & ~(size_t)255; }

SortPlan sort_plan(void* workspace, int64_t n, int num_bits, bool tile_order) {
  SortPlan p;
  p.nblk = (int)ceil_div(n > 0 ? n : 1, kRsTile);
  char* ws = (char*)workspace;
  const size_t seg = al256((size_t)(n > 0 ? n : 1) * 8);
  p.pbuf[0] = (uint2*)ws;
  p.pbuf[1] = (uint2*)(ws ? ws + seg : nullptr);
  p.counts = (int32_t*)(ws ? ws + 2 * seg : nullptr);
  p.totals = p.counts ? p.counts + (int64_t)kRsMaxBins * p.nblk : nullptr;
  // only the low `num_bits` bits of word 0 can be set (num_bits = min(K, 32)): 3 passes of an exact sort for a 3x3x3 kernel
  if (num_bits > 32) num_bits = 32;
  if (num_bits < 1) num_bits = 1;
  if (tile_order && num_bits > 2 * kRsBits) {  // the tile order of the gather GEMMs: top 20 key bits, two wide passes (mask_sort.h)
    p.bits = kRsBitsWide;
    p.passes = 2;
    p.shift0 = num_bits > 2 * kRsBitsWide ? num_bits - 2 * kRsBitsWide : 0;
  } else {
    p.bits = kRsBits;
    p.passes = (num_bits + kRsBits - 1) / kRsBits;
    p.shift0 = 0;
  }
  p.bytes = 2 * seg + al256((size_t)kRsMaxBins * (p.nblk + 1) * 4) + 256;
  return p;
}

int sort_launches(const SortPlan& p, const uint32_t* mask, int mask_words, int64_t n, int32_t* perm, bool first_counted,
                  int kc, RsLaunch out[12]) {
  int count = 0;
  RsArgs a;
  a.kin = mask;
  a.kc = kc;
  a.bits = p.bits;
  a.stride = mask_words;
  a.pin = nullptr;
  a.n = n;
  a.nblk = p.nblk;
  a.counts = p.counts;
  a.totals = p.totals;
  for (int pass = 0; pass < p.passes; ++pass) {
    a.shift = p.shift0 + pass * p.bits;
    const bool last = pass == p.passes - 1;
    a.pout = last ? nullptr : p.pbuf[pass & 1];
    a.vout = last ? perm : nullptr;  // the last pass writes the permutation alone
    if (!(pass == 0 && first_counted)) {
      out[count++] = RsLaunch{kRsHist, p.nblk, a};
      out[count++] = RsLaunch{kRsScan, ((1 << p.bits) + kRsScanCols - 1) / kRsScanCols, a};
    }
    out[count++] = RsLaunch{kRsScatter, p.nblk, a};
    a.pin = a.pout;
  }
  return count;
}

void sort_run_range(const RsLaunch* l, int begin, int end, hipStream_t s) {
  for (int i = begin; i < end; ++i) {
    const dim3 grid((unsigned)l[i].blocks), block(kRsThreads);
    if (l[i].role == kRsHist) hipLaunchKernelGGL(rs_kernel<kRsHist>, grid, block, (size_t)4 << l[i].a.bits, s, l[i].a);
    else if (l[i].role == kRsScan) hipLaunchKernelGGL(rs_kernel<kRsScan>, grid, block, kRsThreads * 4, s, l[i].a);
    else hipLaunchKernelGGL(rs_kernel<kRsScatter>, grid, block, rs_scatter_lds(l[i].a.bits), s, l[i].a);
  }
}

}  // namespace wcn

using namespace wcn;

extern "C" {

size_t wcn_mask_argsort_workspace(int64_t n) { return sort_plan(nullptr, n, 32, false).bytes; }

int wcn_mask_argsort(const uint32_t* mask, int32_t mask_words, int32_t num_bits, int64_t n, int32_t* perm,
                     void* workspace, size_t workspace_bytes, wcn_stream_t stream) {
  if (n < 0 || mask_words < 1 || num_bits < 1) return WCN_ERROR_INVALID_PARAMETERS;
  if (n == 0) return WCN_SUCCESS;
  if (n >= (1ll << 31) || !mask || !perm || !workspace || workspace_bytes < wcn_mask_argsort_workspace(n))
    return WCN_ERROR_INVALID_PARAMETERS;
  RsLaunch l[12];
  const int count = sort_launches(sort_plan(workspace, n, num_bits, false), mask, mask_words, n, perm, false, 0, l);
  sort_run_range(l, 0, count, (hipStream_t)stream);
  return launch_status();
}

int wcn_mask_tile_order(const uint32_t* mask, int32_t mask_words, int32_t num_offsets, int64_t n, int32_t* perm,
                        void* workspace, size_t workspace_bytes, wcn_stream_t stream) {
  if (n < 0 || mask_words < 1 || num_offsets < 1) return WCN_ERROR_INVALID_PARAMETERS;
  if (n == 0) return WCN_SUCCESS;
  if (n >= (1ll << 31) || !mask || !perm || !workspace || workspace_bytes < wcn_mask_argsort_workspace(n))
    return WCN_ERROR_INVALID_PARAMETERS;
  RsLaunch l[12];
  const int kc = tile_key_centre(num_offsets, mask_words);
  const int count = sort_launches(sort_plan(workspace, n, num_offsets < 32 ? num_offsets : 32, kc > 0), mask, mask_words, n, perm,
                                  false, kc, l);
  sort_run_range(l, 0, count, (hipStream_t)stream);
  return launch_status();
}

}  // extern "C"
