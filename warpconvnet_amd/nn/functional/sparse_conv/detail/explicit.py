"""`explicit_gemm` backend: per-offset gather -> torch.matmul -> scatter on the tensor's own device.

Same operation order as the reference (`warpconvnet/nn/functional/sparse_conv/detail/explicit.py:22-101`):
identity offset as one dense matmul, then for every non-empty offset ``X[in_map] @ W[k]`` accumulated at
``out_map``; backward ``dY[out_map] @ W[k]^T`` into ``in_map`` and ``X[in_map]^T @ dY[out_map]`` per offset.
It is a user-selectable algorithm (``fwd_algo="explicit_gemm"``), never a silent fallback.
"""
from typing import Optional, Tuple

import torch
from torch import Tensor

from warpconvnet_amd.geometry.coords.search.search_results import IntSearchResult
from warpconvnet_amd.utils.type_cast import _maybe_cast


def _w(weight: Tensor, k: int) -> Tensor:
    return weight[k]


def _explicit_gemm_forward_logic(in_features: Tensor, weight: Tensor, kernel_map: IntSearchResult, num_out_coords: int,
                                 compute_dtype: Optional[torch.dtype] = None) -> Tensor:
    x = _maybe_cast(in_features, compute_dtype)
    w = _maybe_cast(weight, compute_dtype)
    iden = kernel_map.identity_map_index
    if iden is not None:
        out = torch.matmul(x, w[iden])
    else:
        out = torch.zeros(num_out_coords, w.shape[-1], device=x.device, dtype=x.dtype)
    for k in range(len(kernel_map)):
        if k == iden:
            continue
        in_map, out_map = kernel_map[k]
        if in_map.shape[0] == 0:
            continue
        # every output row appears at most once per offset, so index_add_ == the reference's index-put
        out.index_add_(0, out_map.long(), torch.matmul(x[in_map.long()], w[k]))
    return out.to(dtype=in_features.dtype) if compute_dtype is not None else out


def _explicit_gemm_backward_logic(grad_output: Tensor, in_features: Tensor, weight: Tensor, kernel_map: IntSearchResult,
                                  compute_dtype: Optional[torch.dtype] = None, device=None,
                                  needs_input_grad: Tuple[bool, bool] = (True, True)) -> Tuple[Optional[Tensor], Optional[Tensor]]:
    dt = compute_dtype if compute_dtype is not None else in_features.dtype
    x, w, dy = in_features.to(dt), weight.to(dt), grad_output.to(dt)
    need_dx, need_dw = needs_input_grad
    dx = torch.zeros_like(x) if need_dx else None
    dw = torch.zeros_like(w) if need_dw else None
    iden = kernel_map.identity_map_index
    if iden is not None:
        if need_dx:
            dx = torch.matmul(dy, w[iden].T)
        if need_dw:
            dw[iden] = torch.matmul(x.T, dy)
    for k in range(len(kernel_map)):
        if k == iden:
            continue
        in_map, out_map = kernel_map[k]
        if in_map.shape[0] == 0:
            continue
        g = dy[out_map.long()]
        if need_dx:
            dx.index_add_(0, in_map.long(), torch.matmul(g, w[k].T))
        if need_dw:
            dw[k] += torch.matmul(x[in_map.long()].T, g)
    return (
        dx.to(dtype=in_features.dtype) if dx is not None else None,
        dw.to(dtype=weight.dtype) if dw is not None else None,
    )
