"""HIP backends of the three sparse-conv GEMMs (through the C-ABI in ``include/wcn.h``).

* ``hip_mfma`` - fused gather -> MFMA -> store kernels (`csrc/conv_mfma.hip`, `csrc/wgrad_mfma.hip`)
* ``hip_ref``  - simple kernels for any channel count / fp32 (`csrc/conv_ref.hip`)
* ``auto``     - ``hip_mfma`` when the shape is covered, else ``hip_ref``; a pure function of
  (C_in, C_out, K, dtype), so every rank of a data-parallel job takes the same path.

fp32 feature tensors under ``auto`` / ``hip_mfma`` follow the reference's production treatment
(`mask_gemm.py:72-103, 696-745, 818-963`): operands are cast to fp16 - rescaled by an exact power of two when their
magnitude exceeds the fp16 range, computed on the device, no host sync - the matrix cores accumulate in fp32, the
fused kernels write fp32 results and the product of the operand scales is multiplied back.  ``hip_ref`` /
``explicit_gemm`` keep full fp32 operands.

Role of the reference's `_mask_gemm_forward_logic` / `_mask_gemm_backward_logic`
(`warpconvnet/nn/functional/sparse_conv/detail/mask_gemm.py:661-745, 818-963`).
"""
import functools
from typing import Optional, Tuple

import torch
from torch import Tensor

from warpconvnet_amd import _lib
from warpconvnet_amd.geometry.coords.search.search_results import IntSearchResult
from warpconvnet_amd.geometry.coords.search.torch_discrete import attach_tables_from_csr, reverse_tables


def _prep(t: Tensor, name: str) -> Tensor:
    t = t.contiguous()
    _lib.require_gpu_tensor(t, name)
    return t


_FP16_RESCALE_TARGET = 32768.0


@functools.lru_cache(maxsize=None)
def _gather_ok(cin: int, cout: int, K: int, code: int) -> bool:
    """`wcn_mfma_gather_supported`, memoised: a pure function of the shape (asked several times per layer and step)."""
    return bool(_lib.lib().wcn_mfma_gather_supported(cin, cout, K, code))


@functools.lru_cache(maxsize=None)
def _wgrad_ok(cin: int, cout: int, code: int) -> bool:
    return bool(_lib.lib().wcn_mfma_wgrad_supported(cin, cout, code))


@functools.lru_cache(maxsize=None)
def _wgrad_bias_ok(cin: int, cout: int, code: int) -> bool:
    return bool(_lib.lib().wcn_mfma_wgrad_bias_supported(cin, cout, code))


@functools.lru_cache(maxsize=None)
def _wgrad_workspace(K: int, cin: int, cout: int, code: int) -> int:
    return int(_lib.lib().wcn_conv_wgrad_workspace(K, cin, cout, code))


def fp16_safe_cast(t: Tensor) -> Tuple[Tensor, Tensor]:
    """``(t_fp16, scale)`` with ``t == t_fp16 * scale`` exact in the exponent; scale = 2^max(0, ceil(log2(absmax / 32768)))
    as a 0-dim device tensor (restates the reference's `_fp16_safe_cast`, mask_gemm.py:72-103)."""
    if t.numel() == 0:
        return t.half(), torch.ones((), dtype=torch.float32, device=t.device)
    m = t.abs().max().float()
    exp = torch.clamp(torch.ceil(torch.log2(m / _FP16_RESCALE_TARGET)), min=0.0)
    return (t * torch.exp2(-exp)).half(), torch.exp2(exp)


def _fp32_via_fp16(algo: str, kin: int, kout: int, K: int, dtype: torch.dtype) -> bool:
    """fp32 features take the fp16-operand fused kernels when the algorithm allows and the shape is covered."""
    if dtype != torch.float32 or algo == "hip_ref":
        return False
    return _gather_ok(kin, kout, K, _lib.WCN_F16)


def resolve_gather_algo(algo: str, cin: int, cout: int, K: int, dtype: torch.dtype) -> int:
    L = _lib.lib()
    ok = _gather_ok(cin, cout, K, _lib.dtype_code(dtype))
    if algo == "hip_ref":
        return _lib.WCN_ALGO_REF
    if algo == "hip_mfma":
        if not ok:
            raise RuntimeError(f"hip_mfma error: {_lib.status_string(-4)} (cin={cin}, cout={cout}, K={K}, {dtype})")
        return _lib.WCN_ALGO_MFMA
    return _lib.WCN_ALGO_MFMA if ok else _lib.WCN_ALGO_REF


def resolve_wgrad_algo(algo: str, cin: int, cout: int, dtype: torch.dtype) -> int:
    L = _lib.lib()
    ok = _wgrad_ok(cin, cout, _lib.dtype_code(dtype))
    if algo == "hip_ref":
        return _lib.WCN_ALGO_REF
    if algo == "hip_mfma":
        if not ok:
            raise RuntimeError(f"hip_mfma wgrad error: {_lib.status_string(-4)} (cin={cin}, cout={cout}, {dtype})")
        return _lib.WCN_ALGO_MFMA
    return _lib.WCN_ALGO_MFMA if ok else _lib.WCN_ALGO_REF


@functools.lru_cache(maxsize=None)
def _compact_ok(cin: int, cout: int, K: int, code: int) -> bool:
    return bool(_lib.lib().wcn_conv_compact_table_supported(cin, cout, K, code))


def own_tables(kernel_map, kin: int, kout: int, K: int, dtype: torch.dtype, mfma: bool = True):
    """``(table, mask)`` of a gather-GEMM launch on the map's OWN forward table (the forward product, or the dgrad of a
    submanifold map, which reads the same table with the offsets reversed).  The binned builder leaves COMPACT rows
    (``kernel_map._nbrc``, `csrc/kmap_cells.h`): where the channel-split kernels take the shape they are passed as they are, with
    ``mask`` = None - half the table bytes of the launch and no gather of ``mask[perm[i]]`` (a 128-B line per row).  Everything
    else gets the dense table (expanded from the compact rows on first use) and the mask array."""
    c = getattr(kernel_map, "_nbrc", None)
    if (mfma and c is not None and dtype in (torch.float16, torch.bfloat16)
            and _compact_ok(kin, kout, K, _lib.dtype_code(dtype))):
        return c, None
    return kernel_map._nbr, kernel_map._mask


@functools.lru_cache(maxsize=None)
def _pair_ok(K: int, cin: int, cout: int, code: int) -> bool:
    return bool(_lib.lib().wcn_pack_weight_pair_supported(K, cin, cout, code))


def pack_weight(weight: Tensor, transpose: bool, flip: bool, dtype: Optional[torch.dtype] = None,
                dgrad_flip: Optional[bool] = None) -> Tensor:
    """Fragment-ordered weight image for the MFMA gather-GEMM.  ``weight`` is the forward [K, Cin, Cout]; an fp32 master
    weight with a 16-bit ``dtype`` is rounded while it is packed (one launch instead of cast + pack).

    The image of a PARAMETER (a leaf that requires grad) is kept on the tensor object itself and reused while its version
    counter stands - inference, gradient accumulation, several micro-batches per optimizer step, a layer applied more than
    once; any in-place update bumps ``_version`` and the next use repacks.  The key also carries the storage address and
    the device, so ``module.to(device)`` / ``.double()`` / ``p.data = ...`` (which keep the Parameter object and its
    version) repack.  NOT seen: in-place writes through ``p.data`` (``p.data.copy_``, an EMA swap, an optimizer that
    updates ``p.data``) - they leave version, address and device unchanged; call ``invalidate_packed(module_or_params)``
    after such an update.  Temporaries (autocast copies, computed weights) are never remembered: the cache lives and
    dies with the parameter object, there is no global table."""
    K, c_in, c_out = weight.shape
    kin, kout = (c_out, c_in) if transpose else (c_in, c_out)  # kernel-side channel roles
    dtype = dtype or weight.dtype
    key = (dtype, bool(transpose), bool(flip))
    stamp = (weight._version, weight.data_ptr(), weight.device)
    cache = getattr(weight, "_wcn_packed", None)
    if cache is not None:
        hit = cache.get(key)
        if hit is not None and hit[0] == stamp:
            return hit[1]
    cacheable = weight.requires_grad and weight.is_leaf
    if (dgrad_flip is not None and not transpose and not flip and cacheable
            and weight.dtype == torch.float32 and dtype in (torch.float16, torch.bfloat16)
            and _pair_ok(K, c_in, c_out, _lib.dtype_code(dtype))):
        # the forward image of a parameter that will need its dgrad image in this step's backward (``dgrad_flip``: whether that
        # one is k-flipped - the caller's prediction; a wrong one only costs the ordinary pack later): both in ONE launch
        esz = 2
        fwd = torch.empty(_lib.lib().wcn_packed_weight_bytes(K, c_in, c_out, _lib.dtype_code(dtype), 0) // esz, dtype=dtype,
                          device=weight.device)
        bwd = torch.empty(_lib.lib().wcn_packed_weight_bytes(K, c_out, c_in, _lib.dtype_code(dtype), 1) // esz, dtype=dtype,
                          device=weight.device)
        _lib.check(
            _lib.lib().wcn_pack_weight_f32_pair(_lib.ptr(weight), K, c_in, c_out, _lib.dtype_code(dtype), int(bool(dgrad_flip)),
                                                _lib.ptr(fwd), fwd.numel() * esz, _lib.ptr(bwd), bwd.numel() * esz,
                                                _lib.stream_handle(weight.device)),
            "wcn_pack_weight_f32_pair",
        )
        if cache is None:
            cache = {}
            try:
                weight._wcn_packed = cache
            except AttributeError:
                return fwd
        cache[key] = (stamp, fwd)
        cache[(dtype, True, bool(dgrad_flip))] = (stamp, bwd)
        return fwd
    packed = _pack_weight_uncached(weight, K, kin, kout, transpose, flip, dtype)
    if weight.requires_grad and weight.is_leaf:
        if cache is None:
            cache = {}
            try:
                weight._wcn_packed = cache
            except AttributeError:
                return packed
        cache[key] = (stamp, packed)
    return packed


def invalidate_packed(obj) -> None:
    """Drop the cached packed weight images of a module's parameters (or of an iterable of tensors): needed after in-place
    updates through ``p.data`` (EMA swaps, ``p.data.copy_``), which no version counter records."""
    params = obj.parameters() if hasattr(obj, "parameters") else obj
    for p in params:
        for attr in ("_wcn_packed", "_wcn_packed_grouped", "_wcn_pc_packed"):
            if hasattr(p, attr):
                try:
                    delattr(p, attr)
                except AttributeError:
                    pass


def _pack_weight_uncached(weight: Tensor, K: int, kin: int, kout: int, transpose: bool, flip: bool,
                          dtype: torch.dtype) -> Tensor:
    nbytes = _lib.lib().wcn_packed_weight_bytes(K, kin, kout, _lib.dtype_code(dtype), int(transpose))
    packed = torch.empty(max(weight.numel(), nbytes // max(1, torch.empty((), dtype=dtype).element_size())), dtype=dtype,
                         device=weight.device)
    if weight.dtype == torch.float32 and dtype != torch.float32:
        _lib.check(
            _lib.lib().wcn_pack_weight_f32(_lib.ptr(weight), K, kin, kout, _lib.dtype_code(dtype), int(transpose), int(flip),
                                           _lib.ptr(packed), packed.numel() * packed.element_size(),
                                           _lib.stream_handle(weight.device)),
            "wcn_pack_weight_f32",
        )
        return packed
    _lib.check(
        _lib.lib().wcn_pack_weight(_lib.ptr(weight), K, kin, kout, _lib.dtype_code(weight.dtype), int(transpose),
                                   int(flip), _lib.ptr(packed), packed.numel() * packed.element_size(),
                                   _lib.stream_handle(weight.device)),
        "wcn_pack_weight",
    )
    return packed


# ---- channel counts outside the MFMA tiles (C in {3, 7, 13, 23, 33, 65}, ...): zero-padded channels --------------------------
# The reference covers them with scalar-load tile variants and a zero-filled K tail (`mask_gemm.py:495-541`,
# `warpgemm_a_loader_precomputed.cuh:176-224`).  Here the operands are padded with zero channels to the next shape the
# MFMA kernels take and the result is cut back: zeros contribute nothing to any of the three products, so the values are
# those of the unpadded problem (same fp32 accumulation), at the cost of one padded copy of the operands.
_MFMA_COUTS = (16, 32, 48, 64, 96, 128, 160, 192, 256, 384, 512)


def _code16(dtype: torch.dtype) -> int:
    return _lib.WCN_F16 if dtype == torch.float32 else _lib.dtype_code(dtype)


def _pad_plan(kin: int, kout: int, K: int, dtype: torch.dtype):
    """(kin_padded, kout_padded) of the smallest MFMA gather-GEMM shape that contains kin x kout, or None."""
    if dtype not in (torch.float16, torch.bfloat16):
        return None  # fp32 features keep full fp32 operands on the shapes the MFMA kernels do not take natively
    kin_p = max(32, (kin + 31) // 32 * 32)
    for kout_p in _MFMA_COUTS:
        if kout_p >= kout and _gather_ok(kin_p, kout_p, K, _code16(dtype)):
            return kin_p, kout_p
    return None


def _pad_cols(t: Tensor, width: int) -> Tensor:
    return t if t.shape[-1] == width else torch.nn.functional.pad(t, (0, width - t.shape[-1]))


def master_weight_ok(x_dtype: torch.dtype, weight: Tensor, algo: str, transposed: bool) -> bool:
    """May ``weight`` stay an fp32 master for 16-bit features?  Only when the MFMA kernel takes the shape: the packed
    image is then produced from fp32 directly; every other path multiplies in the storage dtype and needs the cast."""
    if weight.dtype != torch.float32 or x_dtype not in (torch.float16, torch.bfloat16) or weight.ndim != 3 or not weight.is_cuda:
        return False
    if algo not in ("auto", "hip_mfma"):
        return False
    K, cin, cout = weight.shape
    kin, kout = (cout, cin) if transposed else (cin, cout)
    return _gather_ok(kin, kout, K, _lib.dtype_code(x_dtype))


def _gather_gemm(inp: Tensor, weight: Tensor, nbr: Tensor, mask: Tensor, perm: Optional[Tensor], n_out: int,
                 cin: int, cout: int, K: int, algo_code: int, transposed: bool, flip: bool,
                 bias: Optional[Tensor] = None, f32_out: bool = False, dgrad_flip: Optional[bool] = None) -> Tensor:
    out = torch.empty((n_out, cout), dtype=torch.float32 if f32_out else inp.dtype, device=inp.device)
    if n_out == 0:
        return out
    if algo_code == _lib.WCN_ALGO_MFMA:
        w_arg = pack_weight(weight, transposed, flip, dtype=inp.dtype, dgrad_flip=dgrad_flip)
    else:
        w_arg = weight
    if f32_out:
        _lib.check(
            _lib.lib().wcn_conv_gather_gemm_f32out(
                _lib.ptr(inp), _lib.ptr(w_arg), _lib.ptr(out), _lib.ptr(nbr), _lib.ptr(mask), _lib.ptr(perm), _lib.ptr(bias),
                inp.shape[0], n_out, cin, cout, K, _lib.dtype_code(inp.dtype), _lib.stream_handle(inp.device)),
            "wcn_conv_gather_gemm_f32out",
        )
        return out
    _lib.check(
        _lib.lib().wcn_conv_gather_gemm(
            _lib.ptr(inp), _lib.ptr(w_arg), _lib.ptr(out), _lib.ptr(nbr), _lib.ptr(mask), _lib.ptr(perm), _lib.ptr(bias),
            inp.shape[0], n_out, cin, cout, K, _lib.dtype_code(inp.dtype), algo_code, int(transposed), int(flip),
            _lib.stream_handle(inp.device)),
        "wcn_conv_gather_gemm",
    )
    return out


def hip_forward(in_features: Tensor, weight: Tensor, kernel_map: IntSearchResult, num_out_coords: int,
                algo: str = "auto", bias: Optional[Tensor] = None, want_dgrad_image: bool = False) -> Tensor:
    """y[m] = sum_k x[nbr[m][k]] @ w[k] (+ bias, fused into the epilogue in fp32).  ``want_dgrad_image``: a backward with an input
    gradient follows (the autograd Function says so) - the dgrad weight image is packed in the forward image's launch."""
    if bias is not None:
        bias = bias.detach()
        if bias.dtype != torch.float32:
            bias = bias.float()
        bias = _prep(bias, "bias")
    x, w = _prep(in_features, "in_features"), _prep(weight, "weight")
    if x.dtype != w.dtype and not master_weight_ok(x.dtype, w, algo, False):
        raise RuntimeError(f"hip forward error: {_lib.status_string(-6)} ({x.dtype} vs {w.dtype})")
    K, cin, cout = w.shape
    assert K == len(kernel_map) and cin == x.shape[1]
    if algo == "auto" and x.is_cuda and not _gather_ok(cin, cout, K, _code16(x.dtype)):
        plan = _pad_plan(cin, cout, K, x.dtype)
        if plan is not None:  # zero-padded channels on the MFMA kernels instead of one thread per output element
            wp_ = torch.nn.functional.pad(w.to(x.dtype) if w.dtype != x.dtype else w, (0, plan[1] - cout, 0, plan[0] - cin))
            y = hip_forward(_pad_cols(x, plan[0]), wp_, kernel_map, num_out_coords, algo,
                            None if bias is None else _pad_cols(bias, plan[1]))
            return y[:, :cout].contiguous()
    attach_tables_from_csr(kernel_map, x.shape[0], num_out_coords)

    def launch():
        if _fp32_via_fp16(algo, cin, cout, K, x.dtype):
            x16, sx = fp16_safe_cast(x)
            w16, sw = fp16_safe_cast(w)
            tb, mk = own_tables(kernel_map, cin, cout, K, torch.float16)
            y = _gather_gemm(x16, w16, tb, mk, kernel_map._perm, num_out_coords, cin, cout, K,
                             _lib.WCN_ALGO_MFMA, transposed=False, flip=False, bias=None, f32_out=True)
            y = y * (sx * sw)  # exact power-of-two multiply-back, no host sync
            return y if bias is None else y + bias
        code = resolve_gather_algo(algo, cin, cout, K, x.dtype)
        # prediction of the dgrad image's k-flip (a submanifold map over distinct coordinates): exact once the map is validated,
        # "same row count, odd kernel" before that - a wrong guess only costs the ordinary dgrad pack in the backward
        if not want_dgrad_image:
            guess = None
        elif getattr(kernel_map, "_validate_fn", None) is None:
            guess = bool(kernel_map._symmetric)
        else:
            ks = getattr(kernel_map, "_kernel_size", None)
            guess = bool(ks is not None and all(int(k) % 2 == 1 for k in ks) and kernel_map._num_in == kernel_map._num_out)
        tb, mk = own_tables(kernel_map, cin, cout, K, x.dtype, mfma=code == _lib.WCN_ALGO_MFMA)
        return _gather_gemm(x, w, tb, mk, kernel_map._perm, num_out_coords, cin, cout, K, code,
                            transposed=False, flip=False, bias=bias, dgrad_flip=guess)

    # An optimistic map (built by the convolution itself this very call) has not had its status word read: the forward is
    # queued on its tables FIRST - so the GPU runs mask sort -> forward back to back while the host gets to the status -
    # and repeated in the rare case the build had to be redone (block table too small, duplicate coordinates).
    y = launch()
    if kernel_map.validate():
        y = launch()
    return y


def hip_colsum(t: Tensor) -> Tensor:
    """fp32 column sums of a [N, C] tensor (bias gradient), deterministic."""
    t = _prep(t, "grad_output")
    n, c = t.shape
    out = torch.empty(c, dtype=torch.float32, device=t.device)
    L = _lib.lib()
    ws_bytes = L.wcn_colsum_workspace(c)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=t.device)
    _lib.check(
        L.wcn_colsum(_lib.ptr(t), n, c, _lib.dtype_code(t.dtype), _lib.ptr(out), _lib.ptr(ws), ws_bytes,
                     _lib.stream_handle(t.device)),
        "wcn_colsum",
    )
    return out


def _dgrad_pair_lists(dy: Tensor, w: Tensor, kernel_map: IntSearchResult, num_in: int) -> Tensor:
    """Input gradient of a map built over REPEATED coordinates (degenerate input: `Voxels.unique()` removes it).  Several
    output rows then pair with one input row at the same offset, which neither the k-flipped forward table nor the
    one-slot-per-(row, offset) reverse table can express: scatter-add over the pair lists, fp32 accumulation - the
    reference's own explicit formulation (`explicit.py:60-92`).  Correct, not fast; submanifold maps with an odd kernel
    take `_dgrad_duplicates` (gather kernels) instead, this loop is left for the remaining shapes."""
    K, cin, cout = w.shape
    dx = torch.zeros(num_in, cin, dtype=torch.float32, device=dy.device)
    wf = w.float()
    for k in range(K):
        in_map, out_map = kernel_map[k]
        if in_map.shape[0]:
            dx.index_add_(0, in_map.long(), dy[out_map.long()].float() @ wf[k].T)
    return dx.to(dy.dtype)


def _dgrad_duplicates(dy: Tensor, w: Tensor, kernel_map: IntSearchResult, num_in: int, algo: str) -> Tensor:
    """Input gradient of a SUBMANIFOLD map over repeated coordinates (odd kernel, stride 1) on the gather kernels, no loop
    over the offsets and no host sync.  Every row of a coordinate has the same neighbours (the smallest row - the "winner" -
    of each neighbouring coordinate), and only winners appear as inputs, so with ``dy'[w] = sum of dy over the rows at w's
    coordinate`` (one index_add onto ``winner = nbr[:, K//2]``) the gradient of a winner row is the ordinary k-flipped
    product ``dx[i] = sum_k dy'[nbr[i][K-1-k]] . W[k]^T`` and every other row gets zero.  Same pair sums as the reference's
    explicit formulation (`explicit.py:60-92`)."""
    K = len(kernel_map)
    n = num_in
    winner = kernel_map._nbr[:, K // 2].long()
    dyp = torch.zeros((n, dy.shape[1]), dtype=torch.float32, device=dy.device).index_add_(0, winner, dy.float()).to(dy.dtype)
    shadow = IntSearchResult._blank(K, dy.device)  # the same tables, seen as a map without duplicates
    shadow._nbr, shadow._mask, shadow._perm = kernel_map._nbr, kernel_map._mask, kernel_map._perm
    shadow._nbrc = getattr(kernel_map, "_nbrc", None)
    shadow._offsets_dev, shadow._offsets = kernel_map._offsets_dev, kernel_map._offsets
    shadow._symmetric, shadow._has_duplicates = True, False
    shadow._in_maps, shadow._out_maps = kernel_map._in_maps, kernel_map._out_maps
    dx = hip_dgrad(dyp, w, shadow, n, algo)
    keep = winner == torch.arange(n, device=dy.device)
    return dx * keep.unsqueeze(1).to(dx.dtype)


def hip_dgrad(grad_output: Tensor, weight: Tensor, kernel_map: IntSearchResult, num_in_coords: int,
              algo: str = "auto") -> Tensor:
    """dx[n] = sum_k dy[rev[n][k]] @ w[k]^T; a submanifold map reuses the forward table with k reversed."""
    dy, w = _prep(grad_output, "grad_output"), _prep(weight, "weight")
    if dy.dtype != w.dtype and not master_weight_ok(dy.dtype, w, algo, True):
        raise RuntimeError(f"hip dgrad error: {_lib.status_string(-6)} ({dy.dtype} vs {w.dtype})")
    K, cin, cout = w.shape
    kernel_map.validate()
    if getattr(kernel_map, "_has_duplicates", False):
        if getattr(kernel_map, "_dup_symmetric", False) and kernel_map.has_tables and dy.shape[0] == num_in_coords:
            return _dgrad_duplicates(dy, w, kernel_map, num_in_coords, algo)
        return _dgrad_pair_lists(dy, w, kernel_map, num_in_coords)
    if algo == "auto" and dy.is_cuda and not _gather_ok(cout, cin, K, _code16(dy.dtype)):
        plan = _pad_plan(cout, cin, K, dy.dtype)  # kernel-side roles: reduce over cout, produce cin
        if plan is not None:
            wp_ = torch.nn.functional.pad(w.to(dy.dtype) if w.dtype != dy.dtype else w, (0, plan[0] - cout, 0, plan[1] - cin))
            return hip_dgrad(_pad_cols(dy, plan[0]), wp_, kernel_map, num_in_coords, algo)[:, :cin].contiguous()
    attach_tables_from_csr(kernel_map, num_in_coords, dy.shape[0])
    via16 = _fp32_via_fp16(algo, cout, cin, K, dy.dtype)
    code = _lib.WCN_ALGO_MFMA if via16 else resolve_gather_algo(algo, cout, cin, K, dy.dtype)
    if kernel_map._symmetric:
        tbl, mask = own_tables(kernel_map, cout, cin, K, torch.float16 if via16 else dy.dtype, mfma=code == _lib.WCN_ALGO_MFMA)
        perm, flip = kernel_map._perm, True
    else:
        tbl, mask, perm = reverse_tables(kernel_map, num_in_coords)
        flip = False
    if via16:
        g16, sg = fp16_safe_cast(dy)
        w16, sw = fp16_safe_cast(w)
        dx = _gather_gemm(g16, w16, tbl, mask, perm, num_in_coords, cout, cin, K, _lib.WCN_ALGO_MFMA, transposed=True,
                          flip=flip, f32_out=True)
        return dx * (sg * sw)
    return _gather_gemm(dy, w, tbl, mask, perm, num_in_coords, cout, cin, K, code, transposed=True, flip=flip)


def hip_wgrad(in_features: Tensor, grad_output: Tensor, kernel_map: IntSearchResult, weight_shape, algo: str = "auto",
              want_bias_grad: bool = False, out: Optional[Tensor] = None):
    """dw[k] = x[in_k]^T @ dy[out_k], fp32 [K, Cin, Cout].

    ``want_bias_grad``: returns ``(dw, bias_grad_or_None)``; the fp32 column sums of ``grad_output`` come out of the same
    kernel (ones-row MFMA on the centre bucket) when the map is a submanifold map over distinct coordinates and the
    MFMA tile layout supports it, else ``None`` (the caller then uses :func:`hip_colsum`).
    ``out``: fp32 [K, Cin, Cout] destination (a gradient-bucket slot, `dist.GradientBuckets`): the kernel writes there and the
    same tensor is returned; ignored on the paths that post-process the result (zero-padded channels, fp32 operands
    through fp16).
    """
    x, dy = _prep(in_features, "in_features"), _prep(grad_output, "grad_output")
    if x.dtype != dy.dtype:
        raise RuntimeError(f"hip wgrad error: {_lib.status_string(-6)} ({x.dtype} vs {dy.dtype})")
    K, cin, cout = weight_shape
    dev = x.device
    if algo == "auto" and x.is_cuda and not _wgrad_ok(cin, cout, _code16(x.dtype)):
        cin_p, cout_p = (cin + 31) // 32 * 32, (cout + 31) // 32 * 32
        if x.dtype != torch.float32 and _wgrad_ok(cin_p, cout_p, _code16(x.dtype)):  # zero-padded channels: the padded rows / columns of dw are zero
            r = hip_wgrad(_pad_cols(x, cin_p), _pad_cols(dy, cout_p), kernel_map, (K, cin_p, cout_p), algo, want_bias_grad)
            dw_p, db = r if want_bias_grad else (r, None)
            dw_c = dw_p[:, :cin, :cout].contiguous()
            return (dw_c, None if db is None else db[:cout].contiguous()) if want_bias_grad else dw_c
    kernel_map.validate()
    scale = None
    if x.dtype == torch.float32 and algo != "hip_ref" and _wgrad_ok(cin, cout, _lib.WCN_F16):
        x, sx = fp16_safe_cast(x)      # fp16 operands, fp32 accumulate and output; scales multiplied back below
        dy, sg = fp16_safe_cast(dy)
        scale = sx * sg
    if out is not None and scale is None and out.shape == (K, cin, cout) and out.dtype == torch.float32 and out.is_contiguous() \
            and out.device == dev:
        dw = out
    else:
        dw = torch.empty((K, cin, cout), dtype=torch.float32, device=dev)
    if kernel_map._offsets_dev is None:
        kernel_map._offsets_dev = kernel_map.offsets.to(device=dev, dtype=torch.int32)
    L = _lib.lib()
    code = resolve_wgrad_algo(algo, cin, cout, x.dtype)
    ws_bytes = _wgrad_workspace(K, cin, cout, code)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    # (not for fp32 features routed through fp16 operands: the bias gradient then stays an exact fp32 column sum)
    fuse = (want_bias_grad and scale is None and code == _lib.WCN_ALGO_MFMA and getattr(kernel_map, "_self_exact", False)
            and x.shape[0] == dy.shape[0] and dy.shape[0] > 0
            and _wgrad_bias_ok(cin, cout, _lib.dtype_code(x.dtype)))
    if fuse:
        db = torch.empty(cout, dtype=torch.float32, device=dev)
        _lib.check(
            L.wcn_conv_wgrad_bias(_lib.ptr(x), _lib.ptr(dy), _lib.ptr(dw), _lib.ptr(kernel_map.in_maps_device),
                                  _lib.ptr(kernel_map.out_maps_device), _lib.ptr(kernel_map._offsets_dev), x.shape[0],
                                  dy.shape[0], cin, cout, K, _lib.dtype_code(x.dtype), K // 2, _lib.ptr(db), _lib.ptr(ws),
                                  ws_bytes, _lib.stream_handle(dev)),
            "wcn_conv_wgrad_bias",
        )
        return dw, db
    _lib.check(
        L.wcn_conv_wgrad(_lib.ptr(x), _lib.ptr(dy), _lib.ptr(dw), _lib.ptr(kernel_map.in_maps_device),
                         _lib.ptr(kernel_map.out_maps_device), _lib.ptr(kernel_map._offsets_dev), x.shape[0], dy.shape[0], cin,
                         cout, K, _lib.dtype_code(x.dtype), code, _lib.ptr(ws), ws_bytes, _lib.stream_handle(dev)),
        "wcn_conv_wgrad",
    )
    if scale is not None:
        dw = dw * scale
    return (dw, None) if want_bias_grad else dw


# ---- channel groups -----------------------------------------------------------------------------------------------
@functools.lru_cache(maxsize=None)
def _grouped_ok(kin: int, kout: int, K: int, code: int) -> bool:
    return bool(_lib.lib().wcn_mfma_grouped_supported(kin, kout, K, code))


def grouped_supported(weight: Tensor, dtype: torch.dtype, transposed: bool) -> bool:
    """Can the grouped gather-GEMM take this weight ([K, G, Cin/G, Cout/G]) in ONE launch?  Per-group widths must be a
    shape of the 32x32x16 kernels, storage 16-bit."""
    if weight.ndim != 4 or not weight.is_cuda or dtype not in (torch.float16, torch.bfloat16):
        return False
    K, G, cg_in, cg_out = weight.shape
    kin, kout = (cg_out, cg_in) if transposed else (cg_in, cg_out)
    return _grouped_ok(kin, kout, K, _lib.dtype_code(dtype))


def _pack_grouped(weight: Tensor, transpose: bool, flip: bool, dtype: torch.dtype) -> Tensor:
    """G packed images back to back, one launch; cached on the parameter per version like `pack_weight`."""
    K, G, cg_in, cg_out = weight.shape
    kin, kout = (cg_out, cg_in) if transpose else (cg_in, cg_out)
    key = ("grouped", dtype, bool(transpose), bool(flip))
    stamp = (weight._version, weight.data_ptr(), weight.device)
    cache = getattr(weight, "_wcn_packed", None)
    if cache is not None:
        hit = cache.get(key)
        if hit is not None and hit[0] == stamp:
            return hit[1]
    w = weight.contiguous()
    if w.dtype not in (torch.float32, dtype):
        w = w.to(dtype)
    packed = torch.empty(w.numel(), dtype=dtype, device=w.device)
    _lib.check(
        _lib.lib().wcn_pack_weight_grouped(_lib.ptr(w), int(w.dtype == torch.float32), K, G, kin, kout, _lib.dtype_code(dtype),
                                           int(transpose), int(flip), _lib.ptr(packed), _lib.stream_handle(w.device)),
        "wcn_pack_weight_grouped",
    )
    if weight.requires_grad and weight.is_leaf:
        if cache is None:
            cache = {}
            try:
                weight._wcn_packed = cache
            except AttributeError:
                return packed
        cache[key] = (stamp, packed)
    return packed


def _grouped_gather(x: Tensor, packed: Tensor, tbl: Tensor, mask: Tensor, perm: Tensor, n_out: int, kin: int, kout: int,
                    G: int, K: int, bias: Optional[Tensor]) -> Tensor:
    out = torch.empty((n_out, G * kout), dtype=x.dtype, device=x.device)
    if n_out == 0:
        return out
    _lib.check(
        _lib.lib().wcn_conv_gather_gemm_grouped(_lib.ptr(x), _lib.ptr(packed), _lib.ptr(out), _lib.ptr(tbl), _lib.ptr(mask),
                                                _lib.ptr(perm), _lib.ptr(bias), x.shape[0], n_out, kin, kout, G, K,
                                                _lib.dtype_code(x.dtype), _lib.stream_handle(x.device)),
        "wcn_conv_gather_gemm_grouped",
    )
    return out


def hip_forward_grouped(in_features: Tensor, weight: Tensor, kernel_map: IntSearchResult, num_out_coords: int,
                        bias: Optional[Tensor] = None) -> Tensor:
    """y[m][g] = sum_k x[nbr[m][k]][g] @ w[k][g]: all groups in one launch, the group index on grid.y (reference: one launch
    with the group on grid.z, `MaskGemm_forward_64x64x32_1s_flat.h:117-123`).  No channel-slice copies, no concatenation."""
    x = _prep(in_features, "in_features")
    K, G, cg_in, cg_out = weight.shape
    assert K == len(kernel_map) and x.shape[1] == G * cg_in
    if bias is not None:
        bias = _prep(bias.detach().float(), "bias")
    kernel_map.validate()
    attach_tables_from_csr(kernel_map, x.shape[0], num_out_coords)
    packed = _pack_grouped(weight, False, False, x.dtype)
    return _grouped_gather(x, packed, kernel_map._nbr, kernel_map._mask, kernel_map._perm, num_out_coords, cg_in, cg_out, G, K, bias)


def hip_dgrad_grouped(grad_output: Tensor, weight: Tensor, kernel_map: IntSearchResult, num_in_coords: int) -> Tensor:
    dy = _prep(grad_output, "grad_output")
    K, G, cg_in, cg_out = weight.shape
    kernel_map.validate()
    if getattr(kernel_map, "_has_duplicates", False):
        return torch.cat([_dgrad_pair_lists(dy[:, g * cg_out : (g + 1) * cg_out], weight[:, g].to(dy.dtype), kernel_map,
                                            num_in_coords) for g in range(G)], dim=1)
    attach_tables_from_csr(kernel_map, num_in_coords, dy.shape[0])
    if kernel_map._symmetric:
        tbl, mask, perm, flip = kernel_map._nbr, kernel_map._mask, kernel_map._perm, True
    else:
        tbl, mask, perm = reverse_tables(kernel_map, num_in_coords)
        flip = False
    packed = _pack_grouped(weight, True, flip, dy.dtype)
    return _grouped_gather(dy, packed, tbl, mask, perm, num_in_coords, cg_out, cg_in, G, K, None)
