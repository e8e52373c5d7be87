"""Depthwise sparse convolution: ``out[o] += x[i] * w[k]`` per kernel-map pair, weight ``[K, C]``.

Interface of the reference (`warpconvnet/nn/functional/sparse_conv_depth.py:48-58, 657-957`): algorithm enums
``SPARSE_DEPTHWISE_CONV_{FWD,BWD}_ALGO_MODE`` (``explicit`` / ``implicit`` / ``auto``), the autograd function
``UnifiedSpatiallySparseDepthwiseConvFunction(in_features, weight, kernel_map, num_out_coords, fwd_algo, bwd_algo,
compute_dtype)`` and ``spatially_sparse_depthwise_conv``.  Backends of this build:

* ``explicit`` - gather / multiply / index_add on the tensor's device (reference `:227-306` semantics; the only
  backend for CPU tensors)
* ``implicit`` - the HIP kernels behind ``wcn_dwconv_gather`` / ``wcn_dwconv_wgrad`` (``csrc/dwconv.hip``; they take the
  role of the reference's ``_C.fma.implicit_fma`` / ``implicit_reduction``): output-stationary over the neighbour table,
  fp32 accumulation, deterministic
* ``auto``     - ``implicit`` on the GPU, ``explicit`` on the CPU; a static choice (the reference times both and caches
  the winner; nothing is timed here, so ranks cannot diverge).  A failing backend raises, it never falls back.
"""
from enum import Enum
from typing import List, Optional, Tuple, Union

import torch

from warpconvnet_amd.utils.compile_guard import eager_unless_compiling
from torch import Tensor
from torch.autograd import Function

from warpconvnet_amd import _lib
from warpconvnet_amd.constants import (
    WARPCONVNET_DEPTHWISE_CONV_BWD_ALGO_MODE,
    WARPCONVNET_DEPTHWISE_CONV_FWD_ALGO_MODE,
)
from warpconvnet_amd.geometry.coords.search.search_results import IntSearchResult


class SPARSE_DEPTHWISE_CONV_FWD_ALGO_MODE(Enum):
    EXPLICIT = "explicit"
    IMPLICIT = "implicit"
    AUTO = "auto"


class SPARSE_DEPTHWISE_CONV_BWD_ALGO_MODE(Enum):
    EXPLICIT = "explicit"
    IMPLICIT = "implicit"
    AUTO = "auto"


_ALIASES = {"explicit_gemm": "explicit", "implicit_gemm": "implicit"}


def _parse(algo, enum_cls):
    if isinstance(algo, (list, tuple)):  # the reference accepts a list to bound its autotune search: first entry wins
        algo = algo[0]
    if isinstance(algo, str):
        a = algo.strip().lower()
        return enum_cls(_ALIASES.get(a, a))
    return algo


def _explicit_depthwise_forward_logic(in_features: Tensor, weight: Tensor, kernel_map: IntSearchResult, num_out_coords: int,
                                      compute_dtype: Optional[torch.dtype] = None) -> Tensor:
    x = in_features if compute_dtype is None else in_features.to(compute_dtype)
    w = weight if compute_dtype is None else weight.to(compute_dtype)
    iden = kernel_map.identity_map_index
    if iden is not None:
        out = x * w[iden].unsqueeze(0)
    else:
        out = torch.zeros(num_out_coords, w.shape[-1], device=x.device, dtype=x.dtype)
    for k in range(len(kernel_map)):
        if k == iden:
            continue
        in_map, out_map = kernel_map[k]
        if in_map.shape[0] == 0:
            continue
        out.index_add_(0, out_map.to(x.device).long(), x[in_map.to(x.device).long()] * w[k].unsqueeze(0))
    return out.to(in_features.dtype)


def _explicit_depthwise_backward_logic(grad_output: Tensor, in_features: Tensor, weight: Tensor, kernel_map: IntSearchResult,
                                       compute_dtype: Optional[torch.dtype] = None,
                                       device: Optional[torch.device] = None) -> Tuple[Tensor, Tensor]:
    device = device or grad_output.device
    dt = compute_dtype if compute_dtype is not None else in_features.dtype
    x, w, g = in_features.to(device, dt), weight.to(device, dt), grad_output.to(device, dt)
    dw = torch.zeros_like(w)
    iden = kernel_map.identity_map_index
    if iden is not None:
        dx = g * w[iden].unsqueeze(0)
        dw[iden] = torch.sum(x * g, dim=0)
    else:
        dx = torch.zeros_like(x)
    for k in range(len(kernel_map)):
        if k == iden:
            continue
        in_map, out_map = kernel_map[k]
        if in_map.shape[0] == 0:
            continue
        i, o = in_map.to(device).long(), out_map.to(device).long()
        go = g[o]
        dx.index_add_(0, i, go * w[k].unsqueeze(0))
        dw[k] += torch.sum(x[i] * go, dim=0)
    return dx.to(in_features.dtype), dw.to(weight.dtype)


def _tables(kernel_map: IntSearchResult, num_in: int, num_out: int):
    from warpconvnet_amd.geometry.coords.search.torch_discrete import attach_tables_from_csr

    kernel_map.validate()
    attach_tables_from_csr(kernel_map, num_in, num_out)
    return kernel_map._nbr


def _hip_gather(inp: Tensor, weight: Tensor, tbl: Tensor, n_out: int, K: int, flip: bool) -> Tensor:
    C = inp.shape[1]
    out = torch.empty((n_out, C), dtype=inp.dtype, device=inp.device)
    if n_out == 0:
        return out
    _lib.check(
        _lib.lib().wcn_dwconv_gather(_lib.ptr(inp), _lib.ptr(weight), _lib.ptr(out), _lib.ptr(tbl), None, inp.shape[0], n_out,
                                     C, K, _lib.dtype_code(inp.dtype), int(flip), _lib.stream_handle(inp.device)),
        "wcn_dwconv_gather",
    )
    return out


def _implicit_depthwise_forward_logic(in_features: Tensor, weight: Tensor, kernel_map: IntSearchResult, num_out_coords: int,
                                      compute_dtype: Optional[torch.dtype] = None) -> Tensor:
    dt = compute_dtype or in_features.dtype
    x, w = in_features.to(dt).contiguous(), weight.to(dt).contiguous()
    tbl = _tables(kernel_map, x.shape[0], num_out_coords)
    return _hip_gather(x, w, tbl, num_out_coords, w.shape[0], False).to(in_features.dtype)


def _implicit_depthwise_backward_logic(grad_output: Tensor, in_features: Tensor, weight: Tensor, kernel_map: IntSearchResult,
                                       compute_dtype: Optional[torch.dtype] = None,
                                       needs: Tuple[bool, bool] = (True, True)) -> Tuple[Optional[Tensor], Optional[Tensor]]:
    from warpconvnet_amd.geometry.coords.search.torch_discrete import reverse_tables

    dt = compute_dtype or in_features.dtype
    x, w, g = in_features.to(dt).contiguous(), weight.to(dt).contiguous(), grad_output.to(dt).contiguous()
    K, C = w.shape
    n_in, n_out = x.shape[0], g.shape[0]
    dev = x.device
    fwd_tbl = _tables(kernel_map, n_in, n_out)
    dx = dw = None
    if needs[0]:
        if getattr(kernel_map, "_has_duplicates", False):
            # repeated coordinates (degenerate input): several output rows pair with one input row per offset, which the
            # one-slot-per-(row, offset) tables cannot express -> scatter-add over the pair lists (reference formulation)
            dxf = torch.zeros(n_in, C, dtype=torch.float32, device=dev)
            for k in range(K):
                in_map, out_map = kernel_map[k]
                if in_map.shape[0]:
                    dxf.index_add_(0, in_map.long(), g[out_map.long()].float() * w[k].float())
            dx = dxf
        elif kernel_map._symmetric:
            dx = _hip_gather(g, w, fwd_tbl, n_in, K, True)  # rev[n][k] == nbr[n][K-1-k] for a submanifold map
        else:
            rev_tbl, _, _ = reverse_tables(kernel_map, n_in)
            dx = _hip_gather(g, w, rev_tbl, n_in, K, False)
        dx = dx.to(in_features.dtype)
    if needs[1]:
        L = _lib.lib()
        dwf = torch.empty((K, C), dtype=torch.float32, device=dev)
        if kernel_map._offsets_dev is None:
            kernel_map._offsets_dev = kernel_map.offsets.to(device=dev, dtype=torch.int32)
        ws_bytes = L.wcn_dwconv_wgrad_workspace(K, C)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        _lib.check(
            L.wcn_dwconv_wgrad(_lib.ptr(x), _lib.ptr(g), _lib.ptr(dwf), _lib.ptr(kernel_map.in_maps_device),
                               _lib.ptr(kernel_map.out_maps_device), _lib.ptr(kernel_map._offsets_dev), n_in, n_out, C, K,
                               _lib.dtype_code(x.dtype), _lib.ptr(ws), ws_bytes, _lib.stream_handle(dev)),
            "wcn_dwconv_wgrad",
        )
        dw = dwf.to(weight.dtype)
    return dx, dw


def _resolve(algo, on_gpu: bool) -> str:
    v = algo.value
    if v == "auto":
        return "implicit" if on_gpu else "explicit"
    if v == "implicit" and not on_gpu:
        raise RuntimeError("depthwise 'implicit' backend is the HIP path (no CPU fallback); use 'explicit' for CPU tensors")
    return v


class UnifiedSpatiallySparseDepthwiseConvFunction(Function):
    @staticmethod
    def forward(ctx, in_features: Tensor, weight: Tensor, kernel_map: IntSearchResult, num_out_coords: int,
                fwd_algo, bwd_algo, compute_dtype: Optional[torch.dtype]) -> Tensor:
        fwd_algo = _parse(fwd_algo, SPARSE_DEPTHWISE_CONV_FWD_ALGO_MODE)
        bwd_algo = _parse(bwd_algo, SPARSE_DEPTHWISE_CONV_BWD_ALGO_MODE)
        if weight.ndim != 2 or weight.shape[1] != in_features.shape[1]:
            raise ValueError(f"depthwise weight must be [K, C] with C = {in_features.shape[1]}, got {tuple(weight.shape)}")
        ctx.kernel_map, ctx.bwd_algo, ctx.compute_dtype = kernel_map, bwd_algo, compute_dtype
        ctx.save_for_backward(in_features, weight)
        if num_out_coords == 0 or in_features.shape[0] == 0:
            return torch.zeros(num_out_coords, weight.shape[1], dtype=in_features.dtype, device=in_features.device)
        if _resolve(fwd_algo, in_features.is_cuda) == "implicit":
            return _implicit_depthwise_forward_logic(in_features, weight, kernel_map, num_out_coords, compute_dtype)
        return _explicit_depthwise_forward_logic(in_features, weight, kernel_map, num_out_coords, compute_dtype)

    @staticmethod
    def backward(ctx, grad_output: Tensor):
        in_features, weight = ctx.saved_tensors
        need_dx, need_dw = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        grad_in = grad_w = None
        if grad_output.shape[0] == 0 or in_features.shape[0] == 0 or not (need_dx or need_dw):
            grad_in = torch.zeros_like(in_features) if need_dx else None
            grad_w = torch.zeros_like(weight) if need_dw else None
        elif _resolve(ctx.bwd_algo, grad_output.is_cuda) == "implicit":
            grad_in, grad_w = _implicit_depthwise_backward_logic(grad_output.contiguous(), in_features, weight, ctx.kernel_map,
                                                                 ctx.compute_dtype, (need_dx, need_dw))
        else:
            grad_in, grad_w = _explicit_depthwise_backward_logic(grad_output, in_features, weight, ctx.kernel_map,
                                                                 ctx.compute_dtype, grad_output.device)
            grad_in = grad_in if need_dx else None
            grad_w = grad_w if need_dw else None
        ctx.kernel_map = None
        return grad_in, grad_w, None, None, None, None, None


@eager_unless_compiling
def spatially_sparse_depthwise_conv(
    in_features: Tensor,
    weight: Tensor,
    kernel_map: IntSearchResult,
    num_out_coords: int,
    fwd_algo: Union[SPARSE_DEPTHWISE_CONV_FWD_ALGO_MODE, List, str, None] = None,
    bwd_algo: Union[SPARSE_DEPTHWISE_CONV_BWD_ALGO_MODE, List, str, None] = None,
    compute_dtype: Optional[torch.dtype] = None,
) -> Tensor:
    """Depthwise sparse convolution on feature rows (reference `sparse_conv_depth.py:957-1011`)."""
    if fwd_algo is None:
        fwd_algo = WARPCONVNET_DEPTHWISE_CONV_FWD_ALGO_MODE
    if bwd_algo is None:
        bwd_algo = WARPCONVNET_DEPTHWISE_CONV_BWD_ALGO_MODE
    return UnifiedSpatiallySparseDepthwiseConvFunction.apply(in_features, weight, kernel_map, num_out_coords, fwd_algo, bwd_algo,
                                                             compute_dtype)
