"""Explicit gather-matmul-scatter sparse convolution on the CPU.  TEST INFRASTRUCTURE ONLY.

Restates the reference's explicit path (`warpconvnet/nn/functional/sparse_conv/detail/explicit.py`):
  forward  :22-57   identity offset as a dense matmul, then per non-empty offset X[in_map] @ W[k] added at out_map
  backward :60-101  dX[in_map] += dY[out_map] @ W[k]^T ; dW[k] += X[in_map]^T @ dY[out_map]
in the same operation order, on torch CPU tensors (fp32 or fp64).  Pinned by tests/golden/explicit_*.npz,
which were produced by importing the reference itself.
"""
from typing import Optional, Tuple

import numpy as np
import torch


def _t(a, dtype=None):
    t = a if isinstance(a, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(a))
    return t.to(dtype) if dtype is not None else t


def forward(x, w, in_maps, out_maps, offsets, num_out: int, identity_map_index: Optional[int] = None) -> torch.Tensor:
    x, w = _t(x), _t(w)
    in_maps, out_maps = _t(in_maps).long(), _t(out_maps).long()
    offsets = [int(v) for v in np.asarray(offsets).tolist()]
    K = len(offsets) - 1
    if identity_map_index is not None:
        y = torch.matmul(x, w[identity_map_index])
    else:
        y = torch.zeros(num_out, w.shape[-1], dtype=x.dtype)
    for k in range(K):
        if k == identity_map_index or offsets[k + 1] == offsets[k]:
            continue
        i, o = in_maps[offsets[k] : offsets[k + 1]], out_maps[offsets[k] : offsets[k + 1]]
        y[o] += torch.matmul(x[i], w[k])  # each output row appears at most once per offset
    return y


def backward(dy, x, w, in_maps, out_maps, offsets, identity_map_index: Optional[int] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    dy, x, w = _t(dy), _t(x), _t(w)
    in_maps, out_maps = _t(in_maps).long(), _t(out_maps).long()
    offsets = [int(v) for v in np.asarray(offsets).tolist()]
    K = len(offsets) - 1
    dw = torch.zeros_like(w)
    if identity_map_index is not None:
        dx = torch.matmul(dy, w[identity_map_index].T)
        dw[identity_map_index] = torch.matmul(x.T, dy)
    else:
        dx = torch.zeros_like(x)
    for k in range(K):
        if k == identity_map_index or offsets[k + 1] == offsets[k]:
            continue
        i, o = in_maps[offsets[k] : offsets[k + 1]], out_maps[offsets[k] : offsets[k + 1]]
        g = dy[o]
        dx.index_add_(0, i, torch.matmul(g, w[k].T))
        dw[k] += torch.matmul(x[i].T, g)
    return dx, dw


# ---- depthwise (weight [K, C]) -----------------------------------------------------------------------------------
# Restates `warpconvnet/nn/functional/sparse_conv_depth.py`:
#   forward  :227-257   identity offset as X * w[iden], then per non-empty offset X[in_map] * w[k] added at out_map
#   backward :260-306   dX[in_map] += dY[out_map] * w[k] ; dw[k] += sum_rows X[in_map] * dY[out_map]
# Pinned by tests/golden/depthwise_*.npz (produced by importing the reference).
def depthwise_forward(x, w, in_maps, out_maps, offsets, num_out: int, identity_map_index: Optional[int] = None) -> torch.Tensor:
    x, w = _t(x), _t(w)
    in_maps, out_maps = _t(in_maps).long(), _t(out_maps).long()
    offsets = [int(v) for v in np.asarray(offsets).tolist()]
    K = len(offsets) - 1
    if identity_map_index is not None:
        y = x * w[identity_map_index].unsqueeze(0)
    else:
        y = torch.zeros(num_out, w.shape[-1], dtype=x.dtype)
    for k in range(K):
        if k == identity_map_index or offsets[k + 1] == offsets[k]:
            continue
        i, o = in_maps[offsets[k] : offsets[k + 1]], out_maps[offsets[k] : offsets[k + 1]]
        y[o] += x[i] * w[k].unsqueeze(0)
    return y


def depthwise_backward(dy, x, w, in_maps, out_maps, offsets, identity_map_index: Optional[int] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    dy, x, w = _t(dy), _t(x), _t(w)
    in_maps, out_maps = _t(in_maps).long(), _t(out_maps).long()
    offsets = [int(v) for v in np.asarray(offsets).tolist()]
    K = len(offsets) - 1
    dw = torch.zeros_like(w)
    if identity_map_index is not None:
        dx = dy * w[identity_map_index].unsqueeze(0)
        dw[identity_map_index] = torch.sum(x * dy, dim=0)
    else:
        dx = torch.zeros_like(x)
    for k in range(K):
        if k == identity_map_index or offsets[k + 1] == offsets[k]:
            continue
        i, o = in_maps[offsets[k] : offsets[k + 1]], out_maps[offsets[k] : offsets[k + 1]]
        g = dy[o]
        dx.index_add_(0, i, g * w[k].unsqueeze(0))
        dw[k] += torch.sum(x[i] * g, dim=0)
    return dx, dw
