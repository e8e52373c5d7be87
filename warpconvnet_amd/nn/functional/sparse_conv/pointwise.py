"""1x1 (kernel volume 1, stride 1) sparse convolution: a dense ``[N, Cin] x [Cin, Cout]`` product on the feature tensor.

The reference's shortcut is a plain ``feats @ weight[0]`` (`helper.py:206-213`) and leaves the backward to the
framework.  Forward and input gradient are fine that way; the WEIGHT gradient ``X^T dY`` has a tiny output
(``Cin x Cout``) and a reduction over all N rows, a shape the vendor GEMM serves with a non-split kernel (measured
300-600 us at N = 200 k for 32->64 ... 96->20, the largest single kernels of a MinkUNet backward).  It is exactly
the AtB problem of the sparse path with the identity pair list, so it goes through `wcn_conv_wgrad` (fixed grid of pair
ranges + ordered slab reduction, deterministic): ~15 us.  Channel counts outside the MFMA tiles are zero-padded to the
next multiple of 32 for that one product.
"""
import functools
import os
from typing import Optional

import torch
from torch import Tensor
from torch.autograd import Function

from warpconvnet_amd import _lib
from warpconvnet_amd.geometry.coords.search.search_results import IntSearchResult

_ARANGE = {}         # device -> int32 arange, grown geometrically: every identity map of a device is a VIEW of it
_IDENTITY_MAPS = {}  # (n, device) -> map object over a view (a few hundred bytes each; the rows are shared)
_MAX_CACHED_MAPS = 64


def _identity_map(n: int, dev: torch.device) -> IntSearchResult:
    """Pair list (r, r), r = 0..n-1, as a one-offset kernel map.  A training loop with a different row count every iteration
    neither re-creates the rows nor pins one buffer per size: all maps of a device are views of one arange that only grows
    (4 bytes per row of the largest tensor seen); the map OBJECTS are cached by size and dropped oldest-first."""
    key = (n, str(dev))
    km = _IDENTITY_MAPS.get(key)
    rows = _ARANGE.get(key[1])
    if km is not None and rows is not None and km._in_maps.data_ptr() == rows.data_ptr():
        return km
    if rows is None or rows.shape[0] < n:
        rows = torch.arange(max(n, 2 * (rows.shape[0] if rows is not None else 0)), dtype=torch.int32, device=dev)
        _ARANGE[key[1]] = rows
        for k in [k for k in _IDENTITY_MAPS if k[1] == key[1]]:  # (views of the old, smaller buffer)
            del _IDENTITY_MAPS[k]
    while len(_IDENTITY_MAPS) >= _MAX_CACHED_MAPS:
        del _IDENTITY_MAPS[next(iter(_IDENTITY_MAPS))]
    view = rows[:n]
    km = IntSearchResult(view, view, torch.tensor([0, n], dtype=torch.int32))
    km._offsets_dev = torch.tensor([0, n], dtype=torch.int32, device=dev)
    _IDENTITY_MAPS[key] = km
    return km


def _pad_channels(t: Tensor, mult: int = 32, dtype: Optional[torch.dtype] = None) -> Tensor:
    c = t.shape[1]
    cp = (c + mult - 1) // mult * mult
    dtype = t.dtype if dtype is None else dtype
    if cp == c:
        return t.contiguous() if t.dtype == dtype else t.to(dtype).contiguous()
    if t.dtype == dtype:
        return torch.nn.functional.pad(t, (0, cp - c))  # (one launch)
    out = torch.zeros((t.shape[0], cp), dtype=dtype, device=t.device)
    out[:, :c] = t  # (rounds fp32 rows to the compute type in the same copy)
    return out


def dense_wgrad(x: Tensor, dy: Tensor) -> Tensor:
    """fp32 ``x^T @ dy`` ([Cin, Cout]) of two ``[N, C]`` GPU tensors through the sparse wgrad kernel; ``dy`` is 16-bit, ``x`` is
    of the same type or fp32 (the rows a narrow stem read unrounded: rounded like its forward did)."""
    from warpconvnet_amd.nn.functional.sparse_conv.detail import hip_gemm

    n, cin, cout = x.shape[0], x.shape[1], dy.shape[1]
    xp, gp = _pad_channels(x, dtype=dy.dtype), _pad_channels(dy)
    dw = hip_gemm.hip_wgrad(xp, gp, _identity_map(n, x.device), (1, xp.shape[1], gp.shape[1]), "hip_mfma")
    return dw[0, :cin, :cout]


@functools.lru_cache(maxsize=None)
def _identity_ok(cin: int, cout: int, code: int) -> bool:
    return bool(_lib.lib().wcn_conv_identity_supported(cin, cout, code))


@functools.lru_cache(maxsize=None)
def _narrow_ok(cin: int, cout: int, code: int) -> bool:
    # (WARPCONVNET_AMD_NARROW=0: the vendor GEMM for these layers, for A/B timing)
    return os.environ.get("WARPCONVNET_AMD_NARROW", "1") != "0" and bool(_lib.lib().wcn_dense_rows_supported(cin, cout, code))


def narrow_takes_fp32_rows(cin: int, cout: int, code: int) -> bool:
    """True where a 1 x 1 x 1 layer on fp32 rows under 16-bit autocast runs as `narrow_rows` on the fp32 rows themselves."""
    return _narrow_ok(cin, cout, code) and not _identity_ok(cin, cout, code)


def narrow_rows(x: Tensor, weight3: Tensor, transposed: bool, bias: Optional[Tensor] = None,
                dtype: Optional[torch.dtype] = None) -> Optional[Tensor]:
    """The same product for the NARROW layers (stem 3 -> 32, head 96 -> 20 and their input gradients; any cin <= 128, cout <= 96)
    through `wcn_dense_rows`: one streaming launch that reads the weight where it is (fp32 master or 16-bit, either
    orientation) - no packed image, no cast copy.  ``x``: 16-bit rows, or fp32 rows with ``dtype`` = the 16-bit compute type
    (rounded in the kernel, as the cast under autocast would).  None when the kernel does not apply."""
    _, cin, cout = weight3.shape
    kin, kout = (cout, cin) if transposed else (cin, cout)
    out_dtype = dtype if dtype is not None else x.dtype
    if (not x.is_cuda or out_dtype not in (torch.float16, torch.bfloat16) or x.dtype not in (out_dtype, torch.float32)
            or x.ndim != 2 or x.shape[0] == 0 or x.shape[1] != kin or weight3.dtype not in (out_dtype, torch.float32)
            or weight3.device != x.device or not _narrow_ok(kin, kout, _lib.dtype_code(out_dtype))):
        return None
    x = x.contiguous()
    if x.data_ptr() & 15:  # (the kernel's 16-B / 8-B row loads: a contiguous view at an odd storage offset gets its own buffer)
        x = x.clone()
    w = weight3.detach()
    if not w.is_contiguous():
        w = w.contiguous()
    n = x.shape[0]
    out = torch.empty((n, kout), dtype=out_dtype, device=x.device)
    if bias is not None:
        bias = bias.detach().float().contiguous()
    _lib.check(
        _lib.lib().wcn_dense_rows(_lib.ptr(x), int(x.dtype == torch.float32), _lib.ptr(w), int(w.dtype == torch.float32),
                                  int(transposed), _lib.ptr(bias), _lib.ptr(out), n, kin, kout, _lib.dtype_code(out_dtype),
                                  _lib.stream_handle(x.device)),
        "wcn_dense_rows",
    )
    return out


def dense_rows(x: Tensor, weight3: Tensor, transposed: bool, bias: Optional[Tensor] = None) -> Optional[Tensor]:
    """``x @ weight3[0]`` (``transposed``: ``x @ weight3[0].T``) for 16-bit ``[N, C]`` GPU rows through the channel-split gather
    kernel with the identity map - a streaming kernel at its HBM rate where the vendor GEMM serves these skinny shapes at
    0.15-0.33 of it.  ``weight3`` is the convolution's ``[1, Cin, Cout]`` weight (its packed image is cached per parameter
    version like every other layer's).  Shapes outside that kernel (narrow stems and heads) take `narrow_rows`; None when
    neither applies."""
    from warpconvnet_amd.nn.functional.sparse_conv.detail import hip_gemm

    _, cin, cout = weight3.shape
    kin, kout = (cout, cin) if transposed else (cin, cout)
    if not x.is_cuda or x.dtype not in (torch.float16, torch.bfloat16) or x.shape[0] == 0 or x.shape[1] != kin:
        return None
    if not _identity_ok(kin, kout, _lib.dtype_code(x.dtype)):
        return narrow_rows(x, weight3, transposed, bias)
    x = x.contiguous()
    wp = hip_gemm.pack_weight(weight3, transposed, False, dtype=x.dtype)
    n = x.shape[0]
    out = torch.empty((n, kout), dtype=x.dtype, device=x.device)
    if bias is not None:
        bias = bias.detach().float().contiguous()
    _lib.check(
        _lib.lib().wcn_conv_gather_gemm(_lib.ptr(x), _lib.ptr(wp), _lib.ptr(out), None, None, None, _lib.ptr(bias), n, n, kin, kout, 1,
                                        _lib.dtype_code(x.dtype), _lib.WCN_ALGO_MFMA, int(transposed), 0,
                                        _lib.stream_handle(x.device)),
        "wcn_conv_gather_gemm",
    )
    return out


class _PointwiseConv(Function):
    @staticmethod
    def forward(ctx, feats: Tensor, weight: Tensor, bias: Optional[Tensor]) -> Tensor:
        """``weight``: ``[Cin, Cout]``, or the convolution's own ``[1, Cin, Cout]`` parameter (then the products take the
        streaming kernel of `dense_rows` where the shape allows)."""
        ctx.is3 = weight.ndim == 3
        ctx.save_for_backward(feats, weight)  # (the parameter itself: no copy, autograd's version check applies)
        ctx.weight_dtype, ctx.has_bias = weight.dtype, bias is not None
        ctx.bias_dtype = bias.dtype if bias is not None else None
        if ctx.is3:
            out = dense_rows(feats, weight, False, bias)
            if out is not None:
                return out
        w2 = weight[0] if ctx.is3 else weight
        out = feats @ (w2 if w2.dtype == feats.dtype else w2.to(feats.dtype))  # (the cast copy only on the vendor-GEMM path)
        return out if bias is None else out + bias.to(out.dtype)

    @staticmethod
    def backward(ctx, grad_out: Tensor):
        feats, weight = ctx.saved_tensors
        dy = grad_out.contiguous()
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = dense_rows(dy, weight, True) if ctx.is3 else None
            if dx is None:
                w2 = weight[0] if ctx.is3 else weight
                dx = dy @ (w2 if w2.dtype == dy.dtype else w2.to(dy.dtype)).t()
        if ctx.needs_input_grad[1]:
            if (dy.is_cuda and dy.dtype in (torch.float16, torch.bfloat16) and feats.dtype == dy.dtype and feats.shape[0] > 0):
                dw = dense_wgrad(feats.contiguous(), dy).to(ctx.weight_dtype)
            else:
                dw = (feats.t() @ dy).to(ctx.weight_dtype)
            if ctx.is3:
                dw = dw.unsqueeze(0)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            if dy.is_cuda and dy.shape[0] > 0 and dy.dtype in (torch.float32, torch.float16, torch.bfloat16):
                from warpconvnet_amd.nn.functional.sparse_conv.detail.hip_gemm import hip_colsum

                db = hip_colsum(dy).to(ctx.bias_dtype)
            else:
                db = dy.sum(0).to(ctx.bias_dtype)
        return dx, dw, db


def pointwise_conv(feats: Tensor, weight: Tensor, bias: Optional[Tensor] = None) -> Tensor:
    """``feats [N, Cin] @ weight [Cin, Cout] (+ bias)`` with the weight gradient through the sparse wgrad kernel."""
    return _PointwiseConv.apply(feats, weight, bias)
