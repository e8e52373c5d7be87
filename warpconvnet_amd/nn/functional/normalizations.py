"""BatchNorm over sparse feature tensors ``[N, C]`` through the HIP kernels of `csrc/norm.hip`.

The reference applies ``torch.nn.BatchNorm1d`` to the feature tensor (`nn/modules/normalizations.py:30-68`, and
`models/mink_unet.py:31-53` inside every ConvBlock).  The framework's stock kernels run that at ~0.8 TB/s on
``[200 k, 96]`` bf16 (49 + 9 us forward, 50 + 10 us backward - a third of a MinkUNet iteration); ``hip_batch_norm`` is the
same function (training: batch statistics, running-statistics update with the unbiased variance, momentum /
cumulative average; eval: running statistics) as four streaming passes, with the ReLU that follows in a ConvBlock
fused into the apply pass and its mask into the backward passes.  fp32 statistics, fixed-order reductions.
"""
import os
from typing import Optional

import torch
from torch import Tensor
from torch.autograd import Function

from warpconvnet_amd import _lib


def hip_batch_norm_supported(x: Tensor) -> bool:
    """2-D non-empty f32 / f16 / bf16 GPU tensor, and not switched off with WARPCONVNET_AMD_HIP_BATCHNORM=0."""
    return (x.is_cuda and x.ndim == 2 and x.shape[0] > 0
            and x.dtype in (torch.float32, torch.float16, torch.bfloat16)
            and os.environ.get("WARPCONVNET_AMD_HIP_BATCHNORM", "1") not in ("0", "false"))


_WS = {}


def _workspace(c: int, dev) -> Tensor:
    """Partial-sum workspace of the two reduction passes: kept per (device, stream, width) - the passes of one stream are
    ordered, so they can share it - instead of a fresh allocation per call."""
    key = (str(dev), _lib.stream_handle(dev), int(c))
    ws = _WS.get(key)
    if ws is None:
        if len(_WS) >= 64:
            _WS.clear()
        ws = _WS[key] = torch.empty(_lib.lib().wcn_bn_workspace(c), dtype=torch.uint8, device=dev)
    return ws


def _apply(x: Tensor, scale: Tensor, shift: Tensor, relu: bool) -> Tensor:
    y = torch.empty_like(x)
    _lib.check(
        _lib.lib().wcn_bn_apply(_lib.ptr(x), x.shape[0], x.shape[1], _lib.dtype_code(x.dtype), _lib.ptr(scale), _lib.ptr(shift),
                                int(relu), _lib.ptr(y), _lib.stream_handle(x.device)),
        "wcn_bn_apply",
    )
    return y


class _HipBatchNorm(Function):
    @staticmethod
    def forward(ctx, x: Tensor, weight: Optional[Tensor], bias: Optional[Tensor], running_mean: Optional[Tensor],
                running_var: Optional[Tensor], training: bool, momentum: float, eps: float, relu: bool,
                batches: Optional[Tensor] = None) -> Tensor:
        x = x.contiguous()
        n, c = x.shape
        dev = x.device
        L = _lib.lib()
        def f32(t):  # parameters / buffers are fp32 in practice: no copy
            return None if t is None else (t.detach() if t.dtype == torch.float32 and t.is_contiguous() else t.detach().float().contiguous())

        gamma, beta = f32(weight), f32(bias)
        stats = torch.empty((5, c), dtype=torch.float32, device=dev)  # one allocation for the five per-channel vectors
        mean, rstd, scale, shift = stats[0], stats[1], stats[2], stats[3]
        if training:
            var = stats[4]
            ws = _workspace(c, dev)
            # running statistics are updated inside the kernel when they are fp32 (the usual case), else below
            fused_running = (running_mean is not None and running_mean.dtype == torch.float32
                             and running_var.dtype == torch.float32 and running_mean.is_contiguous() and running_var.is_contiguous())
            _lib.check(
                L.wcn_bn_stats_fold(_lib.ptr(x), n, c, _lib.dtype_code(x.dtype), _lib.ptr(gamma), _lib.ptr(beta),
                                    _lib.ptr(running_mean) if fused_running else None,
                                    _lib.ptr(running_var) if fused_running else None, momentum, eps, _lib.ptr(mean),
                                    _lib.ptr(var), _lib.ptr(rstd), _lib.ptr(scale), _lib.ptr(shift), _lib.ptr(batches), _lib.ptr(ws),
                                    ws.numel(), _lib.stream_handle(dev)),
                "wcn_bn_stats_fold",
            )
            if running_mean is not None and not fused_running:  # in place, like F.batch_norm: unbiased variance
                unbias = float(n) / float(max(n - 1, 1))
                running_mean.mul_(1.0 - momentum).add_(mean.to(running_mean.dtype), alpha=momentum)
                running_var.mul_(1.0 - momentum).add_(var.to(running_var.dtype), alpha=momentum * unbias)
        else:
            _lib.check(
                L.wcn_bn_fold(_lib.ptr(f32(running_mean)), _lib.ptr(f32(running_var)), _lib.ptr(gamma), _lib.ptr(beta), eps, c,
                              _lib.ptr(mean), _lib.ptr(rstd), _lib.ptr(scale), _lib.ptr(shift), _lib.stream_handle(dev)),
                "wcn_bn_fold",
            )
        y = _apply(x, scale, shift, relu)
        # (the ReLU mask is recomputed from x and scale / shift in the backward passes: the output is not saved for it)
        ctx.save_for_backward(x, stats, gamma)
        ctx.training, ctx.relu, ctx.has_bias = training, relu, bias is not None
        ctx.wdtype = weight.dtype if weight is not None else None
        ctx.bdtype = bias.dtype if bias is not None else None
        return y

    @staticmethod
    def backward(ctx, grad_out: Tensor):
        x, stats, gamma = ctx.saved_tensors
        mean, rstd = stats[0], stats[1]
        rsc, rsh = (stats[2], stats[3]) if ctx.relu else (None, None)
        n, c = x.shape
        dev = x.device
        L = _lib.lib()
        dy = grad_out.contiguous()
        if dy.dtype != x.dtype:
            dy = dy.to(x.dtype)
        sums = torch.empty((2, c), dtype=torch.float32, device=dev)
        sum_dy, sum_dy_xhat = sums[0], sums[1]
        ws = _workspace(c, dev)
        _lib.check(
            L.wcn_bn_backward_reduce(_lib.ptr(dy), _lib.ptr(x), _lib.ptr(rsc), _lib.ptr(rsh), n, c, _lib.dtype_code(x.dtype), _lib.ptr(mean),
                                     _lib.ptr(rstd), _lib.ptr(sum_dy), _lib.ptr(sum_dy_xhat), _lib.ptr(ws), ws.numel(),
                                     _lib.stream_handle(dev)),
            "wcn_bn_backward_reduce",
        )
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            if ctx.training:
                s0, s1 = sum_dy, sum_dy_xhat
            else:  # eval: the statistics are constants, dx = gamma * rstd * g
                s0 = s1 = torch.zeros(c, dtype=torch.float32, device=dev)
            _lib.check(
                L.wcn_bn_backward_apply(_lib.ptr(dy), _lib.ptr(x), _lib.ptr(rsc), _lib.ptr(rsh), n, c, _lib.dtype_code(x.dtype), _lib.ptr(mean),
                                        _lib.ptr(rstd), _lib.ptr(gamma), _lib.ptr(s0), _lib.ptr(s1), _lib.ptr(dx),
                                        _lib.stream_handle(dev)),
                "wcn_bn_backward_apply",
            )
        dw = sum_dy_xhat.to(ctx.wdtype) if (ctx.wdtype is not None and ctx.needs_input_grad[1]) else None
        db = sum_dy.to(ctx.bdtype) if (ctx.has_bias and ctx.needs_input_grad[2]) else None
        return dx, dw, db, None, None, None, None, None, None, None


def hip_batch_norm(x: Tensor, running_mean: Optional[Tensor], running_var: Optional[Tensor], weight: Optional[Tensor] = None,
                   bias: Optional[Tensor] = None, training: bool = False, momentum: float = 0.1, eps: float = 1e-5,
                   relu: bool = False, num_batches_tracked: Optional[Tensor] = None) -> Tensor:
    """``F.batch_norm`` for ``[N, C]`` GPU features (+ optional fused ReLU).  ``training`` without running statistics uses
    batch statistics only; eval needs running statistics.  ``num_batches_tracked`` (training, int64 scalar on the same
    device): incremented by the statistics kernel instead of a launch of its own."""
    if not hip_batch_norm_supported(x):
        raise RuntimeError(f"hip_batch_norm needs a non-empty 2-D f32/f16/bf16 GPU tensor, got {tuple(x.shape)} {x.dtype} {x.device}")
    if not training and (running_mean is None or running_var is None):
        raise ValueError("hip_batch_norm in eval mode needs running statistics")
    if training and x.shape[0] == 1:  # same refusal as F.batch_norm: a single row has no variance
        raise ValueError(f"Expected more than 1 value per channel when training, got input size {tuple(x.shape)}")
    if num_batches_tracked is not None and not (training and num_batches_tracked.dtype == torch.int64
                                                and num_batches_tracked.device == x.device and num_batches_tracked.numel() == 1):
        raise ValueError("num_batches_tracked must be an int64 scalar on the features' device, training mode only")
    return _HipBatchNorm.apply(x, weight, bias, running_mean, running_var, bool(training), float(momentum), float(eps), bool(relu),
                               num_batches_tracked)


def bn_module_state(norm: torch.nn.modules.batchnorm._BatchNorm, x: Tensor):
    """The bookkeeping of ``nn.BatchNorm1d.forward`` (``num_batches_tracked``, cumulative average when ``momentum is None``,
    which statistics to use) -> ``(use_batch_stats, momentum, eps, running_mean, running_var, counter, fused_running)``:
    ``counter`` is the module's batch counter when the statistics kernel can bump it, ``fused_running`` says the running
    statistics are fp32 and get updated inside that kernel."""
    momentum = 0.0 if norm.momentum is None else norm.momentum
    counter = None
    rm_ok = (norm.running_mean is not None and norm.running_mean.dtype == torch.float32 and norm.running_var.dtype == torch.float32
             and norm.running_mean.is_contiguous() and norm.running_var.is_contiguous())
    if norm.training and norm.track_running_stats and norm.num_batches_tracked is not None:
        nbt = norm.num_batches_tracked
        if norm.momentum is not None and nbt.dtype == torch.int64 and nbt.device == x.device and rm_ok:
            counter = nbt  # the statistics kernel adds the one (it updates the running statistics in the same place)
        else:
            nbt.add_(1)
            if norm.momentum is None:
                momentum = 1.0 / float(nbt)
    use_batch_stats = norm.training or (norm.running_mean is None and norm.running_var is None)
    rm = norm.running_mean if (not norm.training or norm.track_running_stats) else None
    rv = norm.running_var if (not norm.training or norm.track_running_stats) else None
    if not use_batch_stats and not rm_ok:  # eval on non-fp32 running statistics: fp32 copies for the fold kernel
        rm, rv = rm.float().contiguous(), rv.float().contiguous()
    return bool(use_batch_stats), float(momentum), float(norm.eps), rm, rv, counter, bool(rm is not None and rm_ok)


def batch_norm_module_forward(norm: torch.nn.modules.batchnorm._BatchNorm, x: Tensor, relu: bool = False) -> Tensor:
    """``nn.BatchNorm1d.forward`` on ``[N, C]`` features through the HIP kernels: the module's bookkeeping
    (``num_batches_tracked``, cumulative average when ``momentum is None``) followed by :func:`hip_batch_norm`."""
    use_batch_stats, momentum, eps, rm, rv, counter, _ = bn_module_state(norm, x)
    return hip_batch_norm(x, rm, rv, norm.weight, norm.bias, use_batch_stats, momentum, eps, relu, counter)
