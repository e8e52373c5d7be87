"""``SparseConv3d -> BatchNorm1d (-> ReLU)`` as ONE autograd node with direct C-ABI launches.

The reference's models run this chain as three modules (`warpconvnet/models/mink_unet.py:31-53`), each with its own
orchestration (`nn/functional/sparse_conv/helper.py:147-358` -> `detail/unified.py:143-785` -> a backend) and its own
autograd node.  Below ~1 M voxels a network is bound by exactly that host work: on the MI355X box one
conv -> BN -> ReLU block costs 320 us of host time forward + backward for 11 kernel launches (55 us of launch calls;
`tools/host_micro.py`), and MinkUNet-14 has 28 such blocks.  Here the chain is a single `torch.autograd.Function`:

* forward: gather GEMM (queued on the tables of an optimistic map build BEFORE its status word is read, repeated in the
  rare case the build had to be redone) -> BatchNorm statistics + fold -> apply (+ ReLU); the packed weight image comes
  from the per-parameter cache of `hip_gemm.pack_weight`;
* backward: BatchNorm reduce -> apply -> dgrad -> wgrad (written straight into a gradient-bucket slot when the parameter
  publishes one, `dist.GradientBuckets`).

Same kernels, same math and the same saved tensors as the three-module path; only the Python between the launches is
gone.  The fast path takes the layers it can serve without any of the special cases of the general path - 16-bit compute
dtype, shapes of the MFMA kernels in both directions, groups = 1, no convolution bias (it would be cancelled by the
BatchNorm anyway), non-generative, no hooks - and returns ``None`` otherwise: `Sequential` then runs the modules one by one.
A residual tail (``residual=``) and transposed convolutions onto a given tensor's coordinates (``out_spatial=``) are served
for `models/mink_unet.py`'s ``BasicBlock`` and ``ConvTrBlock``.
"""
import os
from typing import Optional

import torch
from torch import Tensor
from torch.autograd import Function

from warpconvnet_amd import _lib
from warpconvnet_amd.dist import claim_grad_slot
from warpconvnet_amd.geometry.coords.search.torch_discrete import reverse_tables
from warpconvnet_amd.nn.functional.normalizations import _workspace as _bn_workspace
from warpconvnet_amd.nn.functional.normalizations import bn_module_state

from .detail import hip_gemm


# Test instrumentation: a callable ``(phase, tensors: dict)`` that sees the convolution INSIDE every fused node - its inputs and
# its raw output in the forward, the gradient that reaches it and the gradients it returns in the backward - so that the tests can
# hold each node's three GEMMs against the fp64 oracle on the node's own inputs (tests/test_gpu_minkunet.py).  None in production.
_OBSERVER = None


class _Plan:
    """Per-call constants of one fused block (plain attributes: cheaper than threading a dozen Function arguments)."""

    __slots__ = ("km", "num_in", "num_out", "cin", "cout", "K", "code", "relu", "training", "momentum", "eps", "running_mean",
                 "running_var", "counter", "fused_running")


def _bn_forward(plan: _Plan, y: Tensor, gamma: Optional[Tensor], beta: Optional[Tensor], residual: Optional[Tensor] = None):
    """BatchNorm (+ residual) (+ ReLU) of the convolution's output ``y``: statistics + fold (training) or fold of the running
    statistics (eval), then apply - one C call in training (`wcn_bn_train_forward`).
    -> (out, stats [5, C] = mean, rstd, scale, shift, var)"""
    L = _lib.lib()
    dev = y.device
    stream = _lib.stream_handle(dev)
    M, cout, code = plan.num_out, plan.cout, plan.code
    stats = torch.empty((5, cout), dtype=torch.float32, device=dev)
    out = torch.empty_like(y)
    gp, bp, yp = _lib.ptr(gamma), _lib.ptr(beta), _lib.ptr(y)
    s0 = stats.data_ptr()
    if plan.training:
        ws = _bn_workspace(cout, dev)
        fr = plan.fused_running
        _lib.check(L.wcn_bn_train_forward(yp, _lib.ptr(residual), M, cout, code, gp, bp, _lib.ptr(plan.running_mean) if fr else None,
                                          _lib.ptr(plan.running_var) if fr else None, plan.momentum, plan.eps,
                                          _lib.ptr(plan.counter), int(plan.relu), s0, _lib.ptr(out), _lib.ptr(ws), ws.numel(),
                                          stream), "wcn_bn_train_forward")
        if plan.running_mean is not None and not fr:  # (non-fp32 running statistics: the module's own arithmetic)
            unbias = float(M) / float(max(M - 1, 1))
            plan.running_mean.mul_(1.0 - plan.momentum).add_(stats[0].to(plan.running_mean.dtype), alpha=plan.momentum)
            plan.running_var.mul_(1.0 - plan.momentum).add_(stats[4].to(plan.running_var.dtype), alpha=plan.momentum * unbias)
        return out, stats
    mean, rstd, scale, shift = (s0 + 4 * cout * i for i in range(4))
    _lib.check(L.wcn_bn_fold(_lib.ptr(plan.running_mean), _lib.ptr(plan.running_var), gp, bp, plan.eps, cout, mean, rstd,
                             scale, shift, stream), "wcn_bn_fold")
    if residual is not None:
        _lib.check(L.wcn_bn_apply_residual(yp, _lib.ptr(residual), M, cout, code, scale, shift, int(plan.relu), _lib.ptr(out),
                                           stream), "wcn_bn_apply_residual")
    else:
        _lib.check(L.wcn_bn_apply(yp, M, cout, code, scale, shift, int(plan.relu), _lib.ptr(out), stream), "wcn_bn_apply")
    return out, stats


def _grad_rows(grad_out: Tensor, like: Tensor):
    """-> (tensor to read, row pitch in elements).  The gradient a channel concatenation hands to one of its inputs is a column
    slice of a wider row-major tensor (`torch.cat` backward: `narrow`): the BatchNorm backward kernels read it in place."""
    g = grad_out
    if g.dtype != like.dtype:
        return g.contiguous().to(like.dtype), like.shape[1]
    if g.is_contiguous():
        return g, g.shape[1]
    if (g.ndim == 2 and g.stride(1) == 1 and g.stride(0) >= g.shape[1] and g.shape[0] > 0
            and g.data_ptr() % 16 == 0 and (g.stride(0) * g.element_size()) % 16 == 0):
        # (16-B aligned rows only: a misaligned slice would take the kernels' element path, whose reduction order differs from
        # the one a contiguous tensor gets - results must not depend on the layout of the incoming gradient)
        return g, g.stride(0)
    return g.contiguous(), g.shape[1]


def _bn_backward(plan: _Plan, grad_out: Tensor, y: Tensor, stats: Tensor, gamma: Optional[Tensor], need_dy: bool,
                 z: Optional[Tensor] = None, need_dres: bool = False):
    """One C call (`wcn_bn_train_backward`: reduce + apply).  -> (gradient of the convolution's output or None, sums [2, C] =
    sum_dy (bias gradient), sum_dy_xhat (weight gradient), gradient of the residual or None).  ``z``: the stored output of a
    residual tail ReLU(BN(y) + r) - the ReLU mask is its sign."""
    L = _lib.lib()
    dev = y.device
    M, cout, code = plan.num_out, plan.cout, plan.code
    g, g_ld = _grad_rows(grad_out, y)
    sums = torch.empty((2, cout), dtype=torch.float32, device=dev)
    ws = _bn_workspace(cout, dev)
    masked = z is not None and plan.relu  # (BN(y) + r without activation: the gradient reaches both branches unmasked)
    want_dx = need_dy or (masked and need_dres)
    dyc = torch.empty_like(y) if want_dx else None
    dres = torch.empty_like(y) if (masked and need_dres) else None
    _lib.check(L.wcn_bn_train_backward_ld(_lib.ptr(g), g_ld, _lib.ptr(y), _lib.ptr(z) if masked else None, int(plan.relu), M, cout,
                                          code, stats.data_ptr(), _lib.ptr(gamma), int(plan.training), sums.data_ptr(),
                                          _lib.ptr(dyc), _lib.ptr(dres), _lib.ptr(ws), ws.numel(), _lib.stream_handle(dev)),
               "wcn_bn_train_backward_ld")
    if need_dres and not masked:
        dres = g
    return (dyc if need_dy else None), sums, dres


def _affine_grads(ctx, sums: Tensor):
    dgamma = dbeta = None
    if ctx.gdtype is not None and ctx.needs_input_grad[2]:
        dgamma = sums[1] if ctx.gdtype == torch.float32 else sums[1].to(ctx.gdtype)
    if ctx.bdtype is not None and ctx.needs_input_grad[3]:
        dbeta = sums[0] if ctx.bdtype == torch.float32 else sums[0].to(ctx.bdtype)
    return dgamma, dbeta


class _ConvBnAct(Function):
    """K > 1: gather GEMM on the kernel map."""

    @staticmethod
    def forward(ctx, x: Tensor, w: Tensor, gamma: Optional[Tensor], beta: Optional[Tensor], plan: _Plan,
                residual: Optional[Tensor] = None) -> Tensor:
        L = _lib.lib()
        dev = x.device
        stream = _lib.stream_handle(dev)
        km, M, cin, cout, K, code = plan.km, plan.num_out, plan.cin, plan.cout, plan.K, plan.code
        # (forward and dgrad image in one launch when this step has a backward; k-flip of the latter predicted as in hip_forward)
        ks = getattr(km, "_kernel_size", None)
        guess = (bool(km._symmetric) if getattr(km, "_validate_fn", None) is None
                 else bool(ks is not None and all(int(k) % 2 == 1 for k in ks) and km._num_in == km._num_out))
        wp = hip_gemm.pack_weight(w, False, False, dtype=x.dtype, dgrad_flip=guess if ctx.needs_input_grad[0] else None)
        y = torch.empty((M, cout), dtype=x.dtype, device=dev)
        xp, wpp, yp = _lib.ptr(x), _lib.ptr(wp), _lib.ptr(y)

        def launch():
            tb, mk = hip_gemm.own_tables(km, cin, cout, K, x.dtype)  # (compact rows and no mask where the kernel takes them)
            _lib.check(L.wcn_conv_gather_gemm(xp, wpp, yp, _lib.ptr(tb), _lib.ptr(mk), _lib.ptr(km._perm), None,
                                              plan.num_in, M, cin, cout, K, code, _lib.WCN_ALGO_MFMA, 0, 0, stream),
                       "wcn_conv_gather_gemm")

        launch()
        if km.validate():  # an optimistic map whose build the device rejected: rebuilt - repeat on the new tables
            launch()
        if _OBSERVER is not None:
            _OBSERVER("forward", dict(x=x, w=w, km=km, y=y, num_in=plan.num_in, num_out=M))
        out, stats = _bn_forward(plan, y, gamma, beta, residual)
        ctx.has_res = residual is not None
        if ctx.has_res and plan.relu:
            ctx.save_for_backward(x, w, y, stats, gamma, out)  # (the output's sign is the ReLU mask of a residual tail)
        else:
            ctx.save_for_backward(x, w, y, stats, gamma)
        ctx.plan = plan
        ctx.gdtype = gamma.dtype if gamma is not None else None
        ctx.bdtype = beta.dtype if beta is not None else None
        return out

    @staticmethod
    def backward(ctx, grad_out: Tensor):
        saved = ctx.saved_tensors
        x, w, y, stats, gamma = saved[:5]
        z = saved[5] if len(saved) > 5 else None  # (stored output of a residual tail with ReLU: its sign is the mask)
        plan = ctx.plan
        km, M, cin, cout, K, code = plan.km, plan.num_out, plan.cin, plan.cout, plan.K, plan.code
        L = _lib.lib()
        dev = y.device
        stream = _lib.stream_handle(dev)
        need_dx, need_dw = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        need_dres = ctx.has_res and ctx.needs_input_grad[5]
        if km._has_duplicates or not (need_dx or need_dw):
            # repeated coordinates (the general path's pair-list dgrad), or nothing behind the BatchNorm: the passes one by one
            dyc, sums, dres = _bn_backward(plan, grad_out, y, stats, gamma, need_dx or need_dw, z if ctx.has_res else None, need_dres)
            dx = hip_gemm.hip_dgrad(dyc, w, km, plan.num_in, "auto") if need_dx else None
            dw = hip_gemm.hip_wgrad(x, dyc, km, (K, cin, cout), "auto").to(w.dtype) if need_dw else None
            if _OBSERVER is not None:
                _OBSERVER("backward", dict(x=x, w=w, km=km, dy=dyc, dx=dx, dw=dw, num_in=plan.num_in, num_out=M))
            dgamma, dbeta = _affine_grads(ctx, sums)
            ctx.plan = None
            return dx, dw, dgamma, dbeta, None, dres
        # one C call: BatchNorm reduce + apply -> dgrad on the reverse tables -> wgrad (wcn_conv_bn_backward)
        g, g_ld = _grad_rows(grad_out, y)
        masked = z is not None and plan.relu
        sums = torch.empty((2, cout), dtype=torch.float32, device=dev)
        dyc = torch.empty_like(y)
        dres = torch.empty_like(y) if (masked and need_dres) else None
        tbl = msk = perm = wpd = dx = None
        flip = False
        if need_dx:
            if km._symmetric:
                (tbl, msk), perm, flip = hip_gemm.own_tables(km, cout, cin, K, y.dtype), km._perm, True
            else:
                tbl, msk, perm = reverse_tables(km, plan.num_in)
            wpd = hip_gemm.pack_weight(w, True, flip, dtype=y.dtype)
            dx = torch.empty((plan.num_in, cin), dtype=y.dtype, device=dev)
        dw = wws = None
        ws_bytes = 0
        if need_dw:
            slot = getattr(w, "_wcn_grad_slot", None)
            if slot is not None and slot.dtype == torch.float32 and slot.shape == w.shape:
                slot = claim_grad_slot(w)  # first producer of this parameter in this backward pass only
            else:
                slot = None
            if slot is not None:
                dw = slot.detach()  # data parallelism: straight into the gradient bucket (dist.GradientBuckets)
            else:
                dw = torch.empty((K, cin, cout), dtype=torch.float32, device=dev)
            ws_bytes = hip_gemm._wgrad_workspace(K, cin, cout, _lib.WCN_ALGO_MFMA)
            wws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        bws = _bn_workspace(cout, dev)
        _lib.check(L.wcn_conv_bn_backward_ld(
            _lib.ptr(g), g_ld, _lib.ptr(x), _lib.ptr(y), _lib.ptr(z) if masked else None, int(plan.relu), stats.data_ptr(), _lib.ptr(gamma),
            int(plan.training), sums.data_ptr(), _lib.ptr(dyc), _lib.ptr(dres), _lib.ptr(wpd), _lib.ptr(tbl), _lib.ptr(msk),
            _lib.ptr(perm), int(flip), _lib.ptr(dx), _lib.ptr(km.in_maps_device) if need_dw else None,
            _lib.ptr(km.out_maps_device) if need_dw else None, _lib.ptr(km._offsets_dev) if need_dw else None, _lib.ptr(dw),
            _lib.ptr(wws), ws_bytes, plan.num_in, M, cin, cout, K, code, _lib.ptr(bws), bws.numel(), stream), "wcn_conv_bn_backward_ld")
        if need_dres and not masked:
            dres = g
        if dw is not None and dw.dtype != w.dtype:
            dw = dw.to(w.dtype)
        if _OBSERVER is not None:
            _OBSERVER("backward", dict(x=x, w=w, km=km, dy=dyc, dx=dx, dw=dw, num_in=plan.num_in, num_out=M))
        dgamma, dbeta = _affine_grads(ctx, sums)
        ctx.plan = None
        return dx, dw, dgamma, dbeta, None, dres


class _PointwiseBnAct(Function):
    """Kernel volume 1, stride 1: dense products on the feature tensor (reference shortcut `helper.py:206-213`); the weight
    gradient through the sparse AtB kernel with the identity pair list (`pointwise.dense_wgrad`)."""

    @staticmethod
    def forward(ctx, x: Tensor, w: Tensor, gamma: Optional[Tensor], beta: Optional[Tensor], plan: _Plan) -> Tensor:
        from .pointwise import dense_rows, narrow_rows

        y = narrow_rows(x, w, False, dtype=_lib.torch_dtype(plan.code)) if x.dtype == torch.float32 else dense_rows(x, w, False)
        if y is None:  # (shapes outside the streaming kernels: the vendor GEMM on a cast copy of the weight)
            xc = x if x.dtype == _lib.torch_dtype(plan.code) else x.to(_lib.torch_dtype(plan.code))  # (fp32 rows under autocast)
            y = xc @ (w[0] if w.dtype == xc.dtype else w[0].to(xc.dtype))
        out, stats = _bn_forward(plan, y, gamma, beta)
        ctx.save_for_backward(x, y, stats, gamma, w)  # (w: the parameter itself - no copy, and autograd's version check applies)
        ctx.plan, ctx.wdtype = plan, w.dtype
        ctx.gdtype = gamma.dtype if gamma is not None else None
        ctx.bdtype = beta.dtype if beta is not None else None
        return out

    @staticmethod
    def backward(ctx, grad_out: Tensor):
        from .pointwise import dense_wgrad

        x, y, stats, gamma, w3 = ctx.saved_tensors
        plan = ctx.plan
        need_dx, need_dw = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        dyc, sums, _ = _bn_backward(plan, grad_out, y, stats, gamma, need_dx or need_dw)
        dx = dw = None
        if need_dx:
            from .pointwise import dense_rows

            dx = dense_rows(dyc, w3, True)
            if dx is None:
                dx = dyc @ (w3[0] if w3.dtype == dyc.dtype else w3[0].to(dyc.dtype)).t()
        if need_dw:
            dw = dense_wgrad(x, dyc).unsqueeze(0)
            if dw.dtype != ctx.wdtype:
                dw = dw.to(ctx.wdtype)
        dgamma, dbeta = _affine_grads(ctx, sums)
        ctx.plan = None
        return dx, dw, dgamma, dbeta, None


def _static_ok(conv, norm) -> int:
    """Module properties that do not change between calls (memoised on the convolution): 0 = general path, 1 = gather GEMM,
    2 = pointwise, 3 = transposed gather GEMM (onto the coordinates of a given tensor)."""
    from warpconvnet_amd.nn.functional.sparse_conv.helper import STRIDED_CONV_MODE
    from warpconvnet_amd.nn.modules.sparse_conv import SpatiallySparseConv

    if type(norm) is not torch.nn.BatchNorm1d or not isinstance(conv, SpatiallySparseConv):
        return 0
    if conv.groups != 1 or conv.bias is not None or conv.generative or conv.weight.ndim != 3:
        return 0
    if conv.num_spatial_dims != 3 or conv.order is not None or conv.compute_dtype is not None:
        return 0
    if conv.stride_mode != STRIDED_CONV_MODE.STRIDE_ONLY:
        return 0
    for a in (conv.fwd_algo, conv.dgrad_algo, conv.wgrad_algo):
        if str(getattr(a, "value", a)).lower() not in ("auto", "hip_mfma"):
            return 0
    if conv.weight.dtype != torch.float32 or norm.num_features != conv.out_channels:
        return 0
    if all(k == 1 for k in conv.kernel_size):
        return 2 if all(s == 1 for s in conv.stride) and not conv.transposed else 0
    return 3 if conv.transposed else 1


def conv_bn_act(x, conv, norm, relu: bool, residual=None, out_spatial=None):
    """``relu(norm(conv(x)) [+ residual])`` (``relu`` optional) through the fused node, or ``None`` when the layer needs the
    general path.

    ``x``: Voxels on the GPU; the compute dtype is the autocast dtype (bf16 / fp16) or a 16-bit feature dtype.
    ``residual``: Voxels on the output coordinates (or their ``[M, Cout]`` feature tensor) - the identity branch of a residual
    block (reference `models/mink_unet.py:160-172`), added between the normalisation and the activation in the same pass.
    ``out_spatial``: transposed convolutions - the Voxels whose coordinates the output takes (`mink_unet.py:83-90`); the map is
    the cached forward map of the matching strided layer with in / out exchanged, its tables are that map's reverse tables."""
    from warpconvnet_amd.geometry.types.voxels import Voxels
    from warpconvnet_amd.nn.functional.sparse_conv.helper import generate_output_coords_and_kernel_map, wrap_conv_output

    if os.environ.get("WARPCONVNET_AMD_FUSED_BLOCK", "1") == "0":
        return None
    ok = conv.__dict__.get("_wcn_block_ok")
    if ok is None or ok[0] is not norm:
        ok = conv.__dict__["_wcn_block_ok"] = (norm, _static_ok(conv, norm))
    kind = ok[1]
    if not kind or not isinstance(x, Voxels):
        return None
    raw = x.batched_features.batched_tensor
    if not raw.is_cuda or raw.shape[0] < 2:
        return None
    dtype = torch.get_autocast_dtype("cuda") if torch.is_autocast_enabled() else raw.dtype
    if dtype not in (torch.bfloat16, torch.float16):
        return None
    if not norm.training and (norm.running_mean is None or norm.running_var is None):
        return None
    # the kernels read gamma / beta (and, in eval, the running statistics) as `const float*`: anything else - a module cast
    # with `.half()`, a non-contiguous view after a checkpoint load - goes the general way, which converts (`_HipBatchNorm`)
    raw_f32 = [norm.weight, norm.bias] + ([] if norm.training else [norm.running_mean, norm.running_var])
    for t in raw_f32:
        if t is not None and (t.dtype != torch.float32 or not t.is_contiguous() or t.device != raw.device):
            return None
    cin, cout = conv.in_channels, conv.out_channels
    K = conv.weight.shape[0]
    code = _lib.dtype_code(dtype)
    plan = _Plan()
    plan.cin, plan.cout, plan.K, plan.code, plan.relu = cin, cout, K, code, bool(relu)
    if residual is not None and kind != 1:
        return None  # (a residual tail behind a 1 x 1 x 1 convolution: the modules one by one)
    if (kind == 3) != (out_spatial is not None):
        return None
    if kind == 2:
        from .pointwise import narrow_takes_fp32_rows

        if raw.dtype == torch.float32 and narrow_takes_fp32_rows(cin, cout, code):
            feats = raw  # (a narrow stem under autocast: the kernel rounds the fp32 rows itself, no cast launch)
        else:
            feats = x.feature_tensor
            if feats.dtype != dtype:
                feats = feats.to(dtype)
        feats = feats.contiguous()
        plan.km, plan.num_in, plan.num_out = None, feats.shape[0], feats.shape[0]
        (plan.training, plan.momentum, plan.eps, plan.running_mean, plan.running_var, plan.counter,
         plan.fused_running) = bn_module_state(norm, feats)
        return x.replace(batched_features=_PointwiseBnAct.apply(feats, conv.weight, norm.weight, norm.bias, plan))
    if not (hip_gemm._gather_ok(cin, cout, K, code) and hip_gemm._gather_ok(cout, cin, K, code) and hip_gemm._wgrad_ok(cin, cout, code)):
        return None
    in_ts = x.tensor_stride or (1, 1, 1)
    if kind == 3:
        from warpconvnet_amd.geometry.coords.search.torch_discrete import attach_tables_from_csr

        if not isinstance(out_spatial, Voxels):
            return None
        out_ts = out_spatial.tensor_stride or (1, 1, 1)
        if not any(o < i for o, i in zip(out_ts, in_ts)):
            return None  # (the general path raises the reference's assertion)
        bcoords_out, out_offsets, km = generate_output_coords_and_kernel_map(
            x, conv.kernel_size, conv.dilation, conv.stride, transposed=True, output_spatially_sparse_tensor=out_spatial,
            need_pairs=torch.is_grad_enabled())
        attach_tables_from_csr(km, raw.shape[0], bcoords_out.shape[0])
    else:
        out_ts = tuple(o * s for o, s in zip(conv.stride, in_ts))
        bcoords_out, out_offsets, km = generate_output_coords_and_kernel_map(
            x, conv.kernel_size, conv.dilation, conv.stride, need_pairs=torch.is_grad_enabled(), optimistic=True)
    M = bcoords_out.shape[0]
    if M < 2 or not km.has_tables:
        # degenerate sizes, or a map that only has its CSR form (user-made): the general path (it finds the map in the cache)
        return None
    feats = x.feature_tensor
    if feats.dtype != dtype:
        feats = feats.to(dtype)
    feats = feats.contiguous()
    plan.km, plan.num_in, plan.num_out = km, feats.shape[0], M
    res = None
    if residual is not None:
        res = residual.feature_tensor if isinstance(residual, Voxels) else residual
        if res.shape != (M, cout) or res.device != feats.device:
            return None
        if res.dtype != dtype:
            res = res.to(dtype)
    # every `return None` is above this line: bn_module_state() bumps `num_batches_tracked` when the kernel cannot, and a
    # bail-out after it would count the batch twice (the modules then run one by one)
    (plan.training, plan.momentum, plan.eps, plan.running_mean, plan.running_var, plan.counter,
     plan.fused_running) = bn_module_state(norm, feats)
    if res is not None:
        out = _ConvBnAct.apply(feats, conv.weight, norm.weight, norm.bias, plan, res.contiguous())
    else:
        out = _ConvBnAct.apply(feats, conv.weight, norm.weight, norm.bias, plan)
    return wrap_conv_output(x, bcoords_out, out_offsets, out, out_ts)
