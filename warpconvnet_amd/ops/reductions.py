"""Row (segment) reductions over CSR row splits.

Interface of the reference (`warpconvnet/ops/reductions.py:13-75`): ``REDUCTIONS`` and
``row_reduction(features [N, F], row_offsets [M+1], reduction)``.  The reference delegates to ``torch_scatter.segment_csr``;
here GPU tensors go through ``wcn_segment_reduce`` (`csrc/points.hip`: one thread per (segment, channel), fp32
accumulation in row order, argmax rows kept for the backward pass) and CPU tensors through plain torch.  The max / min
gradient flows to the FIRST extremum only - the torch_scatter behaviour the reference relies on (`:58-63`).
"""
from enum import Enum
from typing import Literal

import torch
from torch import Tensor
from torch.autograd import Function

from warpconvnet_amd import _lib


class REDUCTIONS(Enum):
    MIN = "min"
    MAX = "max"
    MEAN = "mean"
    SUM = "sum"
    MUL = "mul"
    VAR = "var"
    STD = "std"
    RANDOM = "random"


REDUCTION_TYPES_STR = Literal["min", "max", "mean", "sum", "mul", "var", "std", "random"]
_OP = {"sum": 0, "mean": 1, "max": 2, "min": 3}


def _segment_cpu(features: Tensor, splits: Tensor, op: str):
    m = splits.numel() - 1
    counts = (splits[1:] - splits[:-1]).long()
    seg = torch.repeat_interleave(torch.arange(m), counts)
    out = torch.zeros((m, features.shape[1]), dtype=features.dtype)
    arg = None
    if op in ("sum", "mean"):
        out.index_add_(0, seg, features)
        if op == "mean":
            out = out / counts.clamp_min(1).to(features.dtype).unsqueeze(1)
    else:
        arg = torch.full((m, features.shape[1]), -1, dtype=torch.int64)
        for i in range(m):  # small inputs only (tests / CPU plumbing)
            a, b = int(splits[i]), int(splits[i + 1])
            if b > a:
                v, j = (features[a:b].max(0) if op == "max" else features[a:b].min(0))
                out[i], arg[i] = v, j + a
    return out, arg


class _SegmentReduce(Function):
    @staticmethod
    def forward(ctx, features: Tensor, splits: Tensor, op: str) -> Tensor:
        features = features.contiguous()
        splits64 = splits.to(device=features.device, dtype=torch.int64).contiguous()
        m, c = splits64.numel() - 1, features.shape[1]
        if features.is_cuda:
            out = torch.empty((m, c), dtype=features.dtype, device=features.device)
            arg = torch.empty((m, c), dtype=torch.int64, device=features.device) if op in ("max", "min") else None
            _lib.check(
                _lib.lib().wcn_segment_reduce(_lib.ptr(features), _lib.ptr(splits64), m, c, _lib.dtype_code(features.dtype),
                                              _OP[op], _lib.ptr(out), _lib.ptr(arg), _lib.stream_handle(features.device)),
                "wcn_segment_reduce",
            )
        else:
            out, arg = _segment_cpu(features, splits64, op)
        ctx.op, ctx.n = op, features.shape[0]
        ctx.save_for_backward(splits64, arg if arg is not None else torch.empty(0))
        return out

    @staticmethod
    def backward(ctx, grad_out: Tensor):
        splits, arg = ctx.saved_tensors
        counts = splits[1:] - splits[:-1]
        if ctx.op in ("sum", "mean"):
            g = grad_out
            if ctx.op == "mean":
                g = g / counts.clamp_min(1).to(g.dtype).unsqueeze(1)
            return torch.repeat_interleave(g, counts, dim=0, output_size=ctx.n), None, None
        grad_in = torch.zeros((ctx.n, grad_out.shape[1]), dtype=grad_out.dtype, device=grad_out.device)
        valid = arg >= 0
        cols = torch.arange(grad_out.shape[1], device=grad_out.device).expand_as(arg)
        grad_in[arg[valid], cols[valid]] = grad_out[valid]  # every (row, column) target is unique: first extremum only
        return grad_in, None, None


def row_reduction(features: Tensor, row_offsets: Tensor, reduction, eps: float = 1e-6) -> Tensor:
    if isinstance(reduction, str):
        reduction = REDUCTIONS(reduction)
    assert len(features) == int(row_offsets[-1]), (
        f"Features length {len(features)} must match the last row split {int(row_offsets[-1])}"
    )
    if reduction in (REDUCTIONS.MIN, REDUCTIONS.MAX, REDUCTIONS.MEAN, REDUCTIONS.SUM):
        return _SegmentReduce.apply(features, row_offsets, reduction.value)
    if reduction in (REDUCTIONS.VAR, REDUCTIONS.STD):
        mean = _SegmentReduce.apply(features, row_offsets, "mean")
        var = _SegmentReduce.apply(features**2, row_offsets, "mean") - mean**2
        return var if reduction == REDUCTIONS.VAR else torch.sqrt(var + eps)
    if reduction == REDUCTIONS.RANDOM:
        num = row_offsets[1:] - row_offsets[:-1]
        idx = (torch.rand(len(num), device=num.device) * num).floor().long() + row_offsets[:-1]
        return features[idx.to(features.device)]
    raise ValueError(f"Invalid reduction: {reduction}")
