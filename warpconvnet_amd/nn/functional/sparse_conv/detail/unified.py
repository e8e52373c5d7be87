"""Autograd function of the sparse convolution.

Counterpart of `UnifiedSpatiallySparseConvFunction` (`warpconvnet/nn/functional/sparse_conv/detail/
unified.py:143-785`) with the same call signature and the same backward contract: ``(in_features,
weight)`` are saved (already in compute precision), read from ``ctx.saved_tensors`` exactly once, dgrad
and wgrad are dispatched independently, grads return in the dtypes of the inputs.  Differences: the
algorithm is resolved statically (no run-time autotune sweep, so ranks never diverge) and a failing
backend raises instead of silently falling back to ``explicit_gemm``.
"""
from enum import Enum
from typing import Any, Optional, Tuple, Union

import torch
from torch import Tensor
from torch.autograd import Function

from warpconvnet_amd.dist import claim_grad_slot
from warpconvnet_amd.geometry.coords.search.search_results import IntSearchResult
from warpconvnet_amd.utils.ntuple import _pad_values

from .backends import BwdCtx, FwdCtx, run_backward, run_forward


class SPARSE_CONV_AB_ALGO_MODE(Enum):
    EXPLICIT_GEMM = "explicit_gemm"
    HIP_REF = "hip_ref"
    HIP_MFMA = "hip_mfma"
    AUTO = "auto"


class SPARSE_CONV_ATB_ALGO_MODE(Enum):
    EXPLICIT_GEMM = "explicit_gemm"
    HIP_REF = "hip_ref"
    HIP_MFMA = "hip_mfma"
    AUTO = "auto"


def _algo_name(algo: Any, on_gpu: bool) -> str:
    if isinstance(algo, Enum):
        algo = algo.value
    if isinstance(algo, (list, tuple)):
        algo = algo[0].value if isinstance(algo[0], Enum) else algo[0]
    algo = str(algo).lower()
    if algo == "auto" and not on_gpu:
        # CPU tensors can only run the torch-op backend; "auto" resolves to it explicitly (config 1 plumbing)
        return "explicit_gemm"
    return algo


class UnifiedSpatiallySparseConvFunction(Function):
    @staticmethod
    def forward(
        ctx,
        in_features: Tensor,
        weight: Tensor,
        kernel_map: IntSearchResult,
        num_out_coords: int,
        fwd_algo: Any = "auto",
        dgrad_algo: Any = "auto",
        wgrad_algo: Any = "auto",
        compute_dtype: Optional[torch.dtype] = None,
        fwd_block_size: Optional[int] = None,
        bwd_block_size: Optional[int] = None,
        voxel_size: Optional[Tuple[int, ...]] = None,
        conv_cache_metadata: Optional[dict] = None,
        groups: int = 1,
        use_fp16_accum: bool = False,
        bias: Optional[Tensor] = None,
    ) -> Tensor:
        """``bias`` (15th argument, optional) is a build extension: the bias add is fused into the GEMM epilogue and
        its gradient is a HIP column-sum; the 14-argument reference call signature is unchanged."""
        on_gpu = in_features.is_cuda
        ctx.kernel_map = kernel_map
        ctx.num_out_coords = num_out_coords
        ctx.compute_dtype = compute_dtype
        ctx.groups = groups
        ctx.dgrad_algo = _algo_name(dgrad_algo, on_gpu)
        ctx.wgrad_algo = _algo_name(wgrad_algo, on_gpu)
        ctx.weight_dtype = weight.dtype
        if compute_dtype is not None and weight.dtype != compute_dtype and weight.is_floating_point():
            from .hip_gemm import master_weight_ok

            fa, da = _algo_name(fwd_algo, on_gpu), ctx.dgrad_algo
            keep_master = (on_gpu and groups == 1 and master_weight_ok(compute_dtype, weight, fa, False)
                           and master_weight_ok(compute_dtype, weight, da, True))
            if not keep_master:  # saved in compute precision (reference helper.py:256-262 casts before apply)
                weight = weight.to(compute_dtype)
            # else: the MFMA kernels' packed weight image is rounded from the fp32 master directly (same values, one launch
            # less per direction) and the parameter itself is what is saved for the backward pass
        ctx.save_for_backward(in_features, weight)
        ctx.has_bias = bias is not None
        cout = weight.shape[-1] * (groups if weight.ndim == 4 else 1)
        if num_out_coords == 0 or in_features.shape[0] == 0 or in_features.shape[1] == 0 or cout == 0:
            out = torch.zeros((num_out_coords, cout), dtype=in_features.dtype, device=in_features.device)
            return out if bias is None else out + bias.to(out.dtype)
        fctx = FwdCtx(in_features, weight, kernel_map, num_out_coords, compute_dtype, {}, fwd_block_size, groups,
                      bool(use_fp16_accum), bias, needs_dgrad=bool(ctx.needs_input_grad[0]))
        return run_forward(_algo_name(fwd_algo, on_gpu), fctx)

    @staticmethod
    def backward(ctx, grad_output: Tensor):
        in_features, weight = ctx.saved_tensors  # read exactly once (activation-checkpointing contract)
        need_dx, need_dw = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        need_db = ctx.has_bias and len(ctx.needs_input_grad) > 14 and ctx.needs_input_grad[14]
        grad_in = grad_w = grad_b = None
        empty = ctx.num_out_coords == 0 or in_features.shape[0] == 0 or grad_output.shape[1] == 0
        if empty or not (need_dx or need_dw):
            if need_dx:
                grad_in = torch.zeros_like(in_features)
            if need_dw:
                grad_w = torch.zeros_like(weight)
        else:
            grad_output = grad_output.contiguous()

            # data parallelism (`dist.GradientBuckets`): the parameter publishes its slot in the flat gradient bucket; with no
            # gradient accumulated yet the weight-gradient kernel writes there directly and autograd adopts the alias
            slot = getattr(weight, "_wcn_grad_slot", None)
            if slot is not None and (weight.grad is not None or not need_dw or ctx.groups != 1 or slot.dtype != torch.float32
                                     or ctx.weight_dtype != torch.float32 or slot.shape != weight.shape):
                slot = None
            if slot is not None:
                slot = claim_grad_slot(weight)  # first producer of this parameter in this backward pass only

            def _ctx(needs, want_db=False):
                b = BwdCtx(grad_output, in_features, weight, ctx.kernel_map, ctx.num_out_coords, ctx.compute_dtype,
                           grad_output.device, needs, {}, None, ctx.groups, False, {}, want_db)
                if slot is not None and needs[1]:
                    b.dw_out = slot.detach()  # a fresh alias: the engine may steal it as .grad (sole owner)
                return b

            if ctx.dgrad_algo == ctx.wgrad_algo:
                bctx = _ctx((need_dx, need_dw), need_db)
                grad_in, grad_w = run_backward(ctx.dgrad_algo, bctx)
                grad_b = bctx.bias_grad
            else:
                if need_dx:
                    grad_in, _ = run_backward(ctx.dgrad_algo, _ctx((True, False)))
                if need_dw:
                    bctx = _ctx((False, True), need_db)
                    _, grad_w = run_backward(ctx.wgrad_algo, bctx)
                    grad_b = bctx.bias_grad
        if need_db and grad_b is None:  # the wgrad kernel did not produce it in the same pass
            if grad_output.is_cuda and grad_output.shape[0] > 0:
                from .hip_gemm import hip_colsum

                grad_b = hip_colsum(grad_output.contiguous())
            else:
                grad_b = grad_output.sum(0, dtype=torch.float64 if grad_output.dtype == torch.float64 else torch.float32)
        if grad_w is not None and grad_w.dtype != ctx.weight_dtype:
            grad_w = grad_w.to(ctx.weight_dtype)
        ctx.kernel_map = None  # release eagerly (reference unified.py:779-783)
        out = list(_pad_values(15, grad_in, grad_w))
        out[14] = grad_b
        return tuple(out)
