"""Backend registry of the sparse-conv dispatch.

Same plug-in interface as the reference (`warpconvnet/nn/functional/sparse_conv/detail/backends.py:90-131,
443-510`): ``FORWARD_BACKENDS[name](FwdCtx) -> Tensor | int``, ``BACKWARD_BACKENDS[name](BwdCtx) ->
(Tensor | int | None, Tensor | None)``; a non-zero int status becomes a ``RuntimeError`` in
``run_forward`` / ``run_backward``.  Registered names: ``explicit_gemm``, ``hip_ref``, ``hip_mfma``,
``auto`` (static shape-class choice between the two HIP paths).
"""
import dataclasses
from dataclasses import dataclass
from typing import Any, Callable, Dict, Optional, Tuple

import torch
from torch import Tensor

from warpconvnet_amd import _lib
from warpconvnet_amd.geometry.coords.search.search_results import IntSearchResult

from . import hip_gemm
from .explicit import _explicit_gemm_backward_logic, _explicit_gemm_forward_logic


@dataclass
class FwdCtx:
    in_features: Tensor
    weight: Tensor
    kernel_map: IntSearchResult
    num_out_coords: int
    compute_dtype: Optional[torch.dtype]
    params: Dict[str, Any]
    fwd_block_size: Optional[int] = None
    groups: int = 1
    use_fp16_accum: bool = False
    bias: Optional[Tensor] = None  # fused epilogue (build extension; the reference adds the bias outside)
    needs_dgrad: bool = False      # a backward with an input gradient will follow: pack the dgrad weight image with the forward's


@dataclass
class BwdCtx:
    grad_output: Tensor
    in_features: Tensor
    weight: Tensor
    kernel_map: IntSearchResult
    num_out_coords: int
    compute_dtype: Optional[torch.dtype]
    device: torch.device
    needs_input_grad: Tuple[bool, ...]
    params: Dict[str, Any]
    weight_T: Optional[Tensor] = None
    groups: int = 1
    use_fp16_accum: bool = False
    scratch: Optional[Dict[int, Any]] = None
    want_bias_grad: bool = False        # build extension: the caller would like sum_rows(grad_output) as well
    bias_grad: Optional[Tensor] = None  # set by a backend that produced it in the same pass (fp32 [Cout])
    dw_out: Optional[Tensor] = None     # fp32 destination for the weight gradient (a gradient-bucket slot, dist.py)


FwdFn = Callable[[FwdCtx], Any]
BwdFn = Callable[[BwdCtx], Tuple[Any, Any]]


def _fwd_explicit(ctx: FwdCtx):
    out = _explicit_gemm_forward_logic(ctx.in_features, ctx.weight, ctx.kernel_map, ctx.num_out_coords, ctx.compute_dtype)
    return out if ctx.bias is None else out + ctx.bias.to(out.dtype)


def _bwd_explicit(ctx: BwdCtx):
    return _explicit_gemm_backward_logic(ctx.grad_output, ctx.in_features, ctx.weight, ctx.kernel_map, ctx.compute_dtype,
                                         ctx.device, needs_input_grad=tuple(ctx.needs_input_grad[:2]))


def _make_hip_fwd(algo: str) -> FwdFn:
    def fn(ctx: FwdCtx):
        dt = ctx.compute_dtype or ctx.in_features.dtype
        # (an fp32 master weight stays as it is when the MFMA kernel takes the shape: its packed image is rounded from it)
        w = ctx.weight if hip_gemm.master_weight_ok(dt, ctx.weight, algo, False) else ctx.weight.to(dt)
        out = hip_gemm.hip_forward(ctx.in_features.to(dt), w, ctx.kernel_map, ctx.num_out_coords, algo, bias=ctx.bias,
                                   want_dgrad_image=ctx.needs_dgrad)
        return out.to(ctx.in_features.dtype) if ctx.compute_dtype is not None else out

    return fn


def _make_hip_bwd(algo: str) -> BwdFn:
    def fn(ctx: BwdCtx):
        dt = ctx.compute_dtype or ctx.in_features.dtype
        need_dx, need_dw = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        dy = ctx.grad_output.to(dt)
        dx = dw = None
        # Weight gradient FIRST, input gradient second (round 5): the forward has just gathered every row of x; the weight gradient
        # gathers them again and finds more of them in the 256 MB Infinity Cache before dgrad streams 1 GB of dy rows through
        # it, and dgrad then finds dy warm - 1 069 -> 1 084 M voxels/s on the headline step (three A/B rounds on one box; wgrad
        # 246 -> 248 us, dgrad 249 -> 238 us in-step).  The fused conv -> BatchNorm node keeps dgrad first: there dy has just been
        # written by the BatchNorm backward, and MinkUNet-14 measured the same either way.
        if need_dw:
            # the bias gradient rides along only if it is the column sum of the very tensor autograd handed us
            fuse_db = ctx.want_bias_grad and dy.dtype == ctx.grad_output.dtype
            if fuse_db:
                dw, ctx.bias_grad = hip_gemm.hip_wgrad(ctx.in_features.to(dt), dy, ctx.kernel_map, tuple(ctx.weight.shape),
                                                       algo, want_bias_grad=True, out=getattr(ctx, "dw_out", None))
            else:
                dw = hip_gemm.hip_wgrad(ctx.in_features.to(dt), dy, ctx.kernel_map, tuple(ctx.weight.shape), algo,
                                        out=getattr(ctx, "dw_out", None))
            # stays fp32 here: the autograd function casts once to the dtype of the weight it was given
        if need_dx:
            w = ctx.weight if hip_gemm.master_weight_ok(dt, ctx.weight, algo, True) else ctx.weight.to(dt)
            dx = hip_gemm.hip_dgrad(dy, w, ctx.kernel_map, ctx.in_features.shape[0], algo)
            dx = dx.to(ctx.in_features.dtype)
        return dx, dw

    return fn


FORWARD_BACKENDS: Dict[str, FwdFn] = {
    "explicit_gemm": _fwd_explicit,
    "hip_ref": _make_hip_fwd("hip_ref"),
    "hip_mfma": _make_hip_fwd("hip_mfma"),
    "auto": _make_hip_fwd("auto"),
}

BACKWARD_BACKENDS: Dict[str, BwdFn] = {
    "explicit_gemm": _bwd_explicit,
    "hip_ref": _make_hip_bwd("hip_ref"),
    "hip_mfma": _make_hip_bwd("hip_mfma"),
    "auto": _make_hip_bwd("auto"),
}


def _hip_algo(algo: str, t: Tensor) -> bool:
    return algo in ("auto", "hip_mfma") and t.is_cuda


def _is_depthwise(ctx) -> bool:
    w = ctx.weight
    return w.ndim == 4 and w.shape[2] == 1 and w.shape[3] == 1 and w.shape[1] == ctx.groups


def _grouped_fast_forward(algo: str, ctx: FwdCtx):
    """Single-launch paths of a grouped convolution: the depthwise kernels when every group is one channel
    (`SparseConv3d(groups=C)`: weight [K, C, 1, 1] is the depthwise weight [K, C]), the grouped gather-GEMM (group on
    grid.y) when the per-group widths are an MFMA shape.  None = take the per-group loop."""
    if not _hip_algo(algo, ctx.in_features):
        return None
    dt = ctx.compute_dtype or ctx.in_features.dtype
    if _is_depthwise(ctx):
        from warpconvnet_amd.nn.functional.sparse_conv_depth import _implicit_depthwise_forward_logic

        K, C = ctx.weight.shape[0], ctx.groups
        out = _implicit_depthwise_forward_logic(ctx.in_features, ctx.weight.reshape(K, C), ctx.kernel_map, ctx.num_out_coords, dt)
        return out if ctx.bias is None else out + ctx.bias.to(out.dtype)
    if hip_gemm.grouped_supported(ctx.weight, dt, False) and hip_gemm.grouped_supported(ctx.weight, dt, True):
        out = hip_gemm.hip_forward_grouped(ctx.in_features.to(dt), ctx.weight, ctx.kernel_map, ctx.num_out_coords, ctx.bias)
        return out.to(ctx.in_features.dtype) if ctx.compute_dtype is not None else out
    return None


def _grouped_fast_backward(algo: str, ctx: BwdCtx):
    if not _hip_algo(algo, ctx.grad_output):
        return None
    dt = ctx.compute_dtype or ctx.in_features.dtype
    need_dx, need_dw = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
    if _is_depthwise(ctx):
        from warpconvnet_amd.nn.functional.sparse_conv_depth import _implicit_depthwise_backward_logic

        K, C = ctx.weight.shape[0], ctx.groups
        dx, dw = _implicit_depthwise_backward_logic(ctx.grad_output, ctx.in_features, ctx.weight.reshape(K, C), ctx.kernel_map,
                                                    dt, (need_dx, need_dw))
        return dx, (None if dw is None else dw.reshape(K, C, 1, 1))
    if not (hip_gemm.grouped_supported(ctx.weight, dt, False) and hip_gemm.grouped_supported(ctx.weight, dt, True)):
        return None
    G = ctx.groups
    dy = ctx.grad_output.to(dt)
    dx = dw = None
    if need_dx:
        dx = hip_gemm.hip_dgrad_grouped(dy, ctx.weight, ctx.kernel_map, ctx.in_features.shape[0]).to(ctx.in_features.dtype)
    if need_dw:
        # the AtB kernel tiles the [Cin, Cout] plane; the block-diagonal problem runs as G calls on channel slices
        x = ctx.in_features.to(dt)
        K, _, cg_in, cg_out = ctx.weight.shape
        dws = []
        for g in range(G):
            xs, gs = _group_slices(x, G, g), _group_slices(dy, G, g)
            dws.append(hip_gemm.hip_wgrad(xs, gs, ctx.kernel_map, (K, cg_in, cg_out), algo))
        dw = torch.stack(dws, dim=1)
    return dx, dw


def _group_slices(t: Tensor, groups: int, g: int) -> Tensor:
    c = t.shape[1] // groups
    return t[:, g * c : (g + 1) * c].contiguous()


def run_forward(algo: str, ctx: FwdCtx):
    try:
        fn = FORWARD_BACKENDS[algo]
    except KeyError:
        raise ValueError(f"Unsupported forward algorithm: {algo}")
    if ctx.groups > 1:
        fast = _grouped_fast_forward(algo, ctx)
        if fast is not None:
            return fast
        # Channel groups (weight [K, G, Cin/G, Cout/G], reference sparse_conv.py:147-157): G independent problems on
        # channel slices, same kernel map.  Slices are made contiguous (the kernels take dense [N, C] rows).
        G = ctx.groups
        outs = []
        for g in range(G):
            sub = dataclasses.replace(
                ctx, in_features=_group_slices(ctx.in_features, G, g), weight=ctx.weight[:, g].contiguous(), groups=1,
                bias=None if ctx.bias is None else ctx.bias.reshape(G, -1)[g].contiguous())
            r = fn(sub)
            if isinstance(r, int) and r != 0:
                raise RuntimeError(f"{algo} fwd error: {_lib.status_string(r)}")
            outs.append(r)
        return torch.cat(outs, dim=1)
    result = fn(ctx)
    if isinstance(result, int) and result != 0:
        raise RuntimeError(f"{algo} fwd error: {_lib.status_string(result)}")
    return result


def run_backward(algo: str, ctx: BwdCtx):
    try:
        fn = BACKWARD_BACKENDS[algo]
    except KeyError:
        raise ValueError(f"Unsupported backward algorithm: {algo}")
    if ctx.groups > 1:
        fast = _grouped_fast_backward(algo, ctx)
        if fast is not None:
            return fast
        G = ctx.groups
        dxs, dws, dbs = [], [], []
        for g in range(G):
            sub = dataclasses.replace(
                ctx, grad_output=_group_slices(ctx.grad_output, G, g), in_features=_group_slices(ctx.in_features, G, g),
                weight=ctx.weight[:, g].contiguous(), groups=1, bias_grad=None)
            dx, dw = fn(sub)
            if isinstance(dx, int) and dx != 0:
                raise RuntimeError(f"{algo} bwd error: {_lib.status_string(dx)}")
            dxs.append(dx)
            dws.append(dw)
            dbs.append(sub.bias_grad)
        dx = torch.cat(dxs, dim=1) if dxs[0] is not None else None
        dw = torch.stack(dws, dim=1) if dws[0] is not None else None  # [K, G, Cin/G, Cout/G]
        ctx.bias_grad = torch.cat(dbs) if all(b is not None for b in dbs) else None
        return dx, dw
    result = fn(ctx)
    if isinstance(result[0], int) and result[0] != 0:
        raise RuntimeError(f"{algo} bwd error: {_lib.status_string(result[0])}")
    return result
