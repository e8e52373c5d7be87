"""GPU: HIP BatchNorm over sparse feature tensors vs torch.nn.BatchNorm1d evaluated in fp64 on the CPU (same parameters,
same inputs): forward, input / weight / bias gradients, running statistics, fused ReLU, eval mode, the Sequential
fast path."""
import copy

import numpy as np
import pytest
import torch
import torch.nn as nn

from tests.util import rel_max_err, scene_u

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _ref(bn: nn.BatchNorm1d, x: torch.Tensor, g: torch.Tensor, relu: bool):
    """fp64 reference on the values the kernel saw (x, g already rounded to the storage dtype)."""
    ref = copy.deepcopy(bn).double().cpu()
    xr = x.detach().double().cpu().requires_grad_(True)
    y = ref(xr)
    if relu:
        y = torch.relu(y)
    y.backward(g.double().cpu())
    return ref, y.detach(), xr.grad


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("n,c", [(5000, 96), (20000, 13), (257, 256), (7, 32), (1, 8)])
@pytest.mark.parametrize("relu", [False, True])
def test_training_forward_backward(n, c, dtype, relu):
    from warpconvnet_amd.nn.functional.normalizations import batch_norm_module_forward

    torch.manual_seed(n + c)
    bn = nn.BatchNorm1d(c).to(DEV)
    with torch.no_grad():
        bn.weight.normal_(1.0, 0.3)
        bn.bias.normal_(0.0, 0.3)
    x = (torch.randn(n, c, device=DEV) * 2.0 + 0.7).to(dtype).requires_grad_(True)
    g = torch.randn(n, c, device=DEV).to(dtype)
    if n == 1:  # a single value per channel has no variance: refused in training mode, like torch
        with pytest.raises(ValueError):
            batch_norm_module_forward(bn, x, relu=relu)
        return
    ref, y_ref, dx_ref = _ref(bn, x, g, relu)
    y = batch_norm_module_forward(bn, x, relu=relu)
    y.backward(g)
    tol = 1e-5 if dtype == torch.float32 else 1.5e-2
    assert y.dtype == dtype and rel_max_err(y.detach(), y_ref) < tol
    assert rel_max_err(x.grad, dx_ref) < (1e-4 if dtype == torch.float32 else 3e-2)
    assert rel_max_err(bn.weight.grad, ref.weight.grad) < (1e-4 if dtype == torch.float32 else 2e-2)
    assert rel_max_err(bn.bias.grad, ref.bias.grad) < (1e-4 if dtype == torch.float32 else 2e-2)
    # running statistics (unbiased variance, momentum 0.1) and the step counter
    assert rel_max_err(bn.running_mean, ref.running_mean) < 1e-4 and rel_max_err(bn.running_var, ref.running_var) < 1e-4
    assert int(bn.num_batches_tracked) == 1 == int(ref.num_batches_tracked)


def test_statistics_are_robust_and_deterministic():
    """A large common offset must not cancel the variance (sums are taken around a pivot row); bitwise repeatable."""
    from warpconvnet_amd.nn.functional.normalizations import hip_batch_norm

    torch.manual_seed(0)
    x = (torch.randn(100_000, 64, device=DEV) + 1000.0)
    rm, rv = torch.zeros(64, device=DEV), torch.ones(64, device=DEV)
    y1 = hip_batch_norm(x, rm.clone(), rv.clone(), training=True)
    y2 = hip_batch_norm(x, rm.clone(), rv.clone(), training=True)
    assert torch.equal(y1, y2)
    want = (x.double() - x.double().mean(0)) / x.double().var(0, unbiased=False).add(1e-5).sqrt()
    assert rel_max_err(y1, want) < 1e-3  # fp32 storage of x at 1000 +- 1 limits this, not the reduction


def test_eval_mode_cumulative_momentum_and_sequential():
    from warpconvnet_amd.geometry.types.voxels import Voxels
    from warpconvnet_amd.nn.functional.normalizations import batch_norm_module_forward
    from warpconvnet_amd.nn.modules import BatchNorm, Sequential
    from warpconvnet_amd.nn.modules.sparse_conv import SparseConv3d

    torch.manual_seed(1)
    # momentum=None: cumulative moving average over the batches seen
    bn = nn.BatchNorm1d(24, momentum=None).to(DEV)
    ref = copy.deepcopy(bn).double().cpu()
    for step in range(3):
        x = torch.randn(3000, 24, device=DEV) * (step + 1)
        batch_norm_module_forward(bn, x)
        ref(x.double().cpu())
    assert int(bn.num_batches_tracked) == 3
    assert rel_max_err(bn.running_mean, ref.running_mean) < 1e-4 and rel_max_err(bn.running_var, ref.running_var) < 1e-4
    # eval: running statistics, gradients flow with constant statistics
    bn.eval(); ref.eval()
    x = torch.randn(1000, 24, device=DEV, requires_grad=True)
    g = torch.randn(1000, 24, device=DEV)
    _, y_ref, dx_ref = _ref(bn, x, g, True)
    y = batch_norm_module_forward(bn, x, relu=True)
    y.backward(g)
    assert rel_max_err(y.detach(), y_ref) < 1e-5 and rel_max_err(x.grad, dx_ref) < 1e-5
    # Sequential(conv, BatchNorm1d, ReLU) == the same chain with the stock BatchNorm module (fast path switched off)
    p = torch.from_numpy(scene_u(4000, 3)[:, 1:])
    vox = Voxels([p], [torch.randn(len(p), 16)], device=torch.device(DEV))
    net = Sequential(SparseConv3d(16, 32, 3, bias=False), nn.BatchNorm1d(32), nn.ReLU(inplace=True)).to(DEV)
    net2 = copy.deepcopy(net)
    import os

    y_fast = net(vox).feature_tensor
    os.environ["WARPCONVNET_AMD_HIP_BATCHNORM"] = "0"
    try:
        y_stock = net2(vox).feature_tensor
    finally:
        del os.environ["WARPCONVNET_AMD_HIP_BATCHNORM"]
    assert rel_max_err(y_fast, y_stock) < 1e-4 and (y_fast >= 0).all()
    assert rel_max_err(net[1].running_var, net2[1].running_var) < 1e-4
    # the reference-style wrapper module
    wrap = BatchNorm(32).to(DEV)
    out = wrap(net(vox))
    assert out.feature_tensor.shape == (len(p), 32) and int(wrap.norm.num_batches_tracked) == 1


@pytest.mark.parametrize("cin,cout,ksize,stride,relu", [(64, 64, 3, 1, True), (64, 128, 3, 1, False), (32, 32, 2, 2, True), (96, 96, 3, 1, True),
                                                       (128, 64, 3, 1, True), (3, 32, 1, 1, True), (192, 128, 1, 1, False)])
def test_fused_conv_bn_relu_block_equals_the_module_chain(cin, cout, ksize, stride, relu):
    """`Sequential(SparseConv3d, BatchNorm1d, ReLU)` runs as ONE autograd node with direct launches
    (`nn/functional/sparse_conv/block.py`); a forward hook on the convolution sends the same modules down the general
    path.  Same kernels in the same order: outputs, input / weight / BatchNorm gradients and the running statistics are
    bit-identical, in training and in eval mode."""
    import copy

    from warpconvnet_amd.geometry.types.voxels import Voxels
    from warpconvnet_amd.nn.functional.sparse_conv import block as blk
    from warpconvnet_amd.nn.modules.sequential import Sequential
    from warpconvnet_amd.nn.modules.sparse_conv import SparseConv3d

    dev = torch.device("cuda:0")
    c = scene_u(9000, 71)[:, 1:]
    torch.manual_seed(0)
    fused = Sequential(SparseConv3d(cin, cout, ksize, stride, bias=False), nn.BatchNorm1d(cout), nn.ReLU() if relu else nn.Identity()).to(dev)
    chain = copy.deepcopy(fused)
    chain[0].register_forward_hook(lambda m, i, o: None)  # hooks present -> module-by-module path
    feats = torch.randn(len(c), cin, device=dev)
    calls = []
    Fn = blk._PointwiseBnAct if ksize == 1 else blk._ConvBnAct
    real = Fn.apply
    Fn.apply = staticmethod(lambda *a: (calls.append(1), real(*a))[1])
    try:
        res = []
        for net in (fused, chain):
            for mode in ("train", "eval"):
                net.train(mode == "train")
                x = Voxels([torch.from_numpy(c)], [feats], device=dev)
                x = x.replace(batched_features=x.feature_tensor.detach().clone().requires_grad_(True))
                with torch.autocast("cuda", dtype=torch.bfloat16):
                    y = net(x)
                g = torch.randn(y.feature_tensor.shape, device=dev, generator=torch.Generator(dev).manual_seed(3)).to(y.feature_tensor.dtype)
                net.zero_grad(set_to_none=True)
                y.feature_tensor.backward(g)
                res.append((y.feature_tensor.detach().clone(), x.batched_features.batched_tensor.grad.clone(),
                            net[0].weight.grad.clone(), net[1].weight.grad.clone(), net[1].bias.grad.clone(),
                            net[1].running_mean.clone(), net[1].running_var.clone(), int(net[1].num_batches_tracked)))
    finally:
        Fn.apply = real
    assert len(calls) == 2, "the hook-free Sequential must take the fused node (train + eval), the hooked one must not"
    for a, b in zip(res[:2], res[2:]):
        for u, v in zip(a[:-1], b[:-1]):
            assert torch.equal(u, v)
        assert a[-1] == b[-1]


@pytest.mark.parametrize("amp", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("cin,cout,block", [(64, 64, "BasicBlock"), (96, 128, "BasicBlock"), (128, 128, "BottleneckBlock")])
def test_residual_tail_equals_the_module_chain(cin, cout, block, amp):
    """`models.mink_unet.BasicBlock`: conv2's BatchNorm, `out += identity` and the ReLU (reference `mink_unet.py:160-172`) ride
    on the fused node's two BatchNorm passes (`wcn_bn_apply_residual`, `wcn_bn_backward_*_masked`).  With hooks on conv2 the same
    block runs its modules one by one: outputs, every gradient (input - both branches meet there -, weights, BatchNorm
    parameters) and the running statistics are bit-identical, in training and in eval mode, under bf16 and fp16 autocast (fp16:
    the fp32 affine result must be rounded in a separate step in EVERY kernel - `bn_affine`'s barrier against v_fma_mixlo_f16).  (A bottleneck's last convolution is
    1 x 1 x 1: it keeps the module chain, the test pins that it still computes the same.)"""
    import copy

    from warpconvnet_amd.geometry.types.voxels import Voxels
    from warpconvnet_amd.models import mink_unet
    from warpconvnet_amd.nn.functional.sparse_conv import block as blk

    dev = torch.device("cuda:0")
    c = scene_u(7000, 72)[:, 1:]
    torch.manual_seed(1)
    fused = getattr(mink_unet, block)(cin, cout).to(dev)
    chain = copy.deepcopy(fused)
    last = chain.conv2 if block == "BasicBlock" else chain.conv3
    last[0].register_forward_hook(lambda m, i, o: None)  # hooks present -> conv2 / add / ReLU as separate modules
    feats = torch.randn(len(c), cin, device=dev)
    with_res = []
    real = blk._ConvBnAct.apply
    blk._ConvBnAct.apply = staticmethod(lambda *a: (with_res.append(len(a) > 5 and a[5] is not None), real(*a))[1])
    try:
        res = []
        for net in (fused, chain):
            for mode in ("train", "eval"):
                net.train(mode == "train")
                x = Voxels([torch.from_numpy(c)], [feats], device=dev)
                x = x.replace(batched_features=x.feature_tensor.detach().clone().requires_grad_(True))
                with torch.autocast("cuda", dtype=amp):
                    y = net(x)
                g = torch.randn(y.feature_tensor.shape, device=dev, generator=torch.Generator(dev).manual_seed(5)).to(y.feature_tensor.dtype)
                net.zero_grad(set_to_none=True)
                y.feature_tensor.backward(g)
                out = [y.feature_tensor.detach().clone(), x.batched_features.batched_tensor.grad.clone()]
                out += [p.grad.clone() for p in net.parameters()]
                out += [b.clone() for b in net.buffers()]
                res.append(out)
    finally:
        blk._ConvBnAct.apply = real
    assert sum(with_res) == (2 if block == "BasicBlock" else 0), "the hook-free BasicBlock takes the residual tail (train + eval)"
    assert (res[0][0] >= 0).all() and float(res[0][0].float().abs().max()) > 0
    for a, b in zip(res[:2], res[2:]):
        assert len(a) == len(b)
        for u, v in zip(a, b):
            assert torch.equal(u, v)


def test_minkunet14_module_matches_its_module_by_module_run():
    """`models.mink_unet.MinkUNet14` end to end (stem, four stride-2 stages, transposed stages onto the encoder tensors,
    concatenations, 1 x 1 x 1 head): the fused blocks against WARPCONVNET_AMD_FUSED_BLOCK=0 on the same weights - same logits and
    the same gradients bit for bit; activation checkpointing leaves values and BatchNorm statistics unchanged."""
    import copy
    import os

    from warpconvnet_amd.geometry.types.voxels import Voxels
    from warpconvnet_amd.models.mink_unet import MinkUNet14

    dev = torch.device("cuda:0")
    c = scene_u(30000, 73)[:, 1:]
    torch.manual_seed(2)
    net = MinkUNet14(3, 20).to(dev)
    feats = torch.randn(len(c), 3, device=dev)

    def run(model):
        x = Voxels([torch.from_numpy(c)], [feats], device=dev)
        model.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y = model(x)
        y.feature_tensor.float().square().mean().backward()
        return ([y.feature_tensor.detach().clone()] + [p.grad.clone() for p in model.parameters()],
                [b.clone() for b in model.buffers()])

    a, abuf = run(copy.deepcopy(net))
    os.environ["WARPCONVNET_AMD_FUSED_BLOCK"] = "0"
    try:
        b, bbuf = run(copy.deepcopy(net))
    finally:
        del os.environ["WARPCONVNET_AMD_FUSED_BLOCK"]
    assert a[0].shape == (len(c), 20)
    for u, v in zip(a + abuf, b + bbuf):
        assert torch.equal(u, v)
    ck = copy.deepcopy(net)
    ck.gradient_checkpointing_enable()
    cgrads, cbuf = run(ck)
    for u, v in zip(a + abuf, cgrads + cbuf):
        assert torch.equal(u, v)


@pytest.mark.parametrize("freeze", ["weight", "input", "both"])
def test_fused_node_with_frozen_weight_or_input(freeze):
    """`wcn_conv_bn_backward` skips the products nobody asked for: a frozen convolution weight (no wgrad), an input that does not
    require grad (no dgrad), both (BatchNorm sums only) - the gradients that remain equal the module chain's bit for bit."""
    import copy

    from warpconvnet_amd.geometry.types.voxels import Voxels
    from warpconvnet_amd.nn.modules.sequential import Sequential
    from warpconvnet_amd.nn.modules.sparse_conv import SparseConv3d

    dev = torch.device("cuda:0")
    c = scene_u(6000, 74)[:, 1:]
    torch.manual_seed(3)
    fused = Sequential(SparseConv3d(64, 96, 3, bias=False), nn.BatchNorm1d(96), nn.ReLU()).to(dev)
    chain = copy.deepcopy(fused)
    chain[0].register_forward_hook(lambda m, i, o: None)
    feats = torch.randn(len(c), 64, device=dev)
    res = []
    for net in (fused, chain):
        net[0].weight.requires_grad_(freeze not in ("weight", "both"))
        x = Voxels([torch.from_numpy(c)], [feats], device=dev)
        x = x.replace(batched_features=x.feature_tensor.detach().clone().requires_grad_(freeze not in ("input", "both")))
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y = net(x)
        g = torch.randn(y.feature_tensor.shape, device=dev, generator=torch.Generator(dev).manual_seed(7)).to(y.feature_tensor.dtype)
        net.zero_grad(set_to_none=True)
        y.feature_tensor.backward(g)
        xg = x.batched_features.batched_tensor.grad
        res.append((y.feature_tensor.detach().clone(), xg, net[0].weight.grad, net[1].weight.grad.clone(), net[1].bias.grad.clone()))
    a, b = res
    assert torch.equal(a[0], b[0]) and torch.equal(a[3], b[3]) and torch.equal(a[4], b[4])
    assert (a[1] is None) == (freeze in ("input", "both")) and (b[1] is None) == (a[1] is None)
    assert (a[2] is None) == (freeze in ("weight", "both")) and (b[2] is None) == (a[2] is None)
    if a[1] is not None:
        assert torch.equal(a[1], b[1])
    if a[2] is not None:
        assert torch.equal(a[2], b[2])


@pytest.mark.parametrize("name", ["MinkUNet18", "MinkUNet50"])
def test_minkunet_family_runs_and_matches_the_module_chain(name):
    """The deeper members of `models.mink_unet` (two BasicBlocks per stage; BottleneckBlocks with 1 x 1 x 1 convolutions and a
    residual tail behind a pointwise layer): logits and every parameter gradient with the fused nodes equal the
    WARPCONVNET_AMD_FUSED_BLOCK=0 run bit for bit, gradients finite and non-zero."""
    import copy
    import os

    from warpconvnet_amd.geometry.types.voxels import Voxels
    from warpconvnet_amd.models import mink_unet

    dev = torch.device("cuda:0")
    c = scene_u(12000, 75)[:, 1:]
    torch.manual_seed(4)
    net = getattr(mink_unet, name)(3, 7).to(dev)
    feats = torch.randn(len(c), 3, device=dev)

    def run(model):
        model.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y = model(Voxels([torch.from_numpy(c)], [feats], device=dev))
        y.feature_tensor.float().square().mean().backward()
        return [y.feature_tensor.detach().clone()] + [p.grad.clone() for p in model.parameters()]

    a = run(copy.deepcopy(net))
    os.environ["WARPCONVNET_AMD_FUSED_BLOCK"] = "0"
    try:
        b = run(copy.deepcopy(net))
    finally:
        del os.environ["WARPCONVNET_AMD_FUSED_BLOCK"]
    assert a[0].shape == (len(c), 7)
    for u, v in zip(a, b):
        assert torch.isfinite(u).all() and torch.equal(u, v)
    assert sum(float(g.abs().sum()) > 0 for g in a[1:]) > 0.9 * (len(a) - 1)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16, torch.float32])
def test_recomputed_relu_mask_is_the_stored_outputs_sign(dtype):
    """The backward passes never read the forward's output: the ReLU mask is recomputed from x and the scale / shift the forward
    applied, with the same fma and the same rounding to the storage type.  Property: with dy = 1 the masked column sums equal the
    count of positive stored outputs per channel EXACTLY, on values crowded around the threshold (a 1-ulp disagreement between
    two kernels' roundings - the fp16 fma + convert fusion `bn_affine` guards against - shows up here)."""
    from warpconvnet_amd import _lib

    dev = torch.device("cuda:0")
    L = _lib.lib()
    st = _lib.stream_handle(dev)
    code = _lib.dtype_code(dtype)
    n, c = 200_000, 64
    for seed in range(3):
        g = torch.Generator(device=dev).manual_seed(seed)
        x = (torch.randn(n, c, device=dev, generator=g) * 0.05).to(dtype)
        scale = torch.rand(c, device=dev, generator=g) + 0.5
        shift = torch.randn(c, device=dev, generator=g) * 0.01
        y = torch.empty_like(x)
        _lib.check(L.wcn_bn_apply(_lib.ptr(x), n, c, code, _lib.ptr(scale), _lib.ptr(shift), 1, _lib.ptr(y), st), "wcn_bn_apply")
        dy = torch.ones_like(x)
        mean, rstd = torch.zeros(c, device=dev), torch.ones(c, device=dev)
        s0, s1 = torch.empty(c, device=dev), torch.empty(c, device=dev)
        ws = torch.empty(L.wcn_bn_workspace(c), dtype=torch.uint8, device=dev)
        _lib.check(L.wcn_bn_backward_reduce(_lib.ptr(dy), _lib.ptr(x), _lib.ptr(scale), _lib.ptr(shift), n, c, code, _lib.ptr(mean),
                                            _lib.ptr(rstd), _lib.ptr(s0), _lib.ptr(s1), _lib.ptr(ws), ws.numel(), st),
                   "wcn_bn_backward_reduce")
        want = (y > 0).sum(0).float()
        assert 0.1 * n < float(want.min()) and float(want.max()) < 0.9 * n  # (the threshold runs through the data)
        assert torch.equal(s0, want)


@pytest.mark.parametrize("case", ["duplicates", "table_full"])
def test_fused_node_on_scenes_the_optimistic_build_rejects(case):
    """The fused conv -> BatchNorm -> ReLU node queues its gather GEMM on the tables of an optimistic build: (i) duplicate
    coordinates whose later copy comes first (rebuild with the strict insert, and the backward takes the pair-list dgrad of the
    general path), (ii) a scene too sparse for the first block table (TABLE_FULL rebuild).  Hints reset so the branches are taken;
    outputs, gradients and running statistics equal the module chain's bit for bit."""
    import copy

    from warpconvnet_amd.geometry.coords.search.torch_discrete import default_hints
    from warpconvnet_amd.geometry.types.voxels import Voxels
    from warpconvnet_amd.nn.modules.sequential import Sequential
    from warpconvnet_amd.nn.modules.sparse_conv import SparseConv3d

    dev = torch.device("cuda:0")
    rng = np.random.default_rng(18)
    if case == "table_full":
        cells = rng.permutation(36 * 36 * 36)[:24000]
        c = np.stack([cells // 1296 * 8 + 7 * (cells % 2), cells // 36 % 36 * 8, cells % 36 * 8], 1).astype(np.int32)
    else:
        base = scene_u(5000, 32)[:, 1:]
        c = np.concatenate([base[:400][::-1], base], 0).astype(np.int32)
    torch.manual_seed(5)
    fused = Sequential(SparseConv3d(64, 128, 3, bias=False), nn.BatchNorm1d(128), nn.ReLU()).to(dev)
    chain = copy.deepcopy(fused)
    chain[0].register_forward_hook(lambda m, i, o: None)
    feats = torch.randn(len(c), 64)
    res = []
    for net in (fused, chain):
        default_hints().reset()
        vox = Voxels([torch.from_numpy(c)], [feats], device=dev)
        x = vox.replace(batched_features=vox.feature_tensor.detach().clone().requires_grad_(True))
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y = net(x)
        g = torch.randn(y.feature_tensor.shape, device=dev, generator=torch.Generator(dev).manual_seed(9)).to(y.feature_tensor.dtype)
        net.zero_grad(set_to_none=True)
        y.feature_tensor.backward(g)
        km = next(iter(x.cache.values()))
        assert km._has_duplicates == (case == "duplicates")
        res.append([y.feature_tensor.detach().clone(), x.batched_features.batched_tensor.grad.clone(), net[0].weight.grad.clone(),
                    net[1].weight.grad.clone(), net[1].bias.grad.clone(), net[1].running_mean.clone(), net[1].running_var.clone()])
    for u, v in zip(*res):
        assert torch.isfinite(u.float()).all() and torch.equal(u, v)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16, torch.float32])
@pytest.mark.parametrize("c,left,right", [(96, 32, 0), (32, 0, 96), (64, 3, 5), (20, 4, 0), (40, 8, 8)])
@pytest.mark.parametrize("tail", [False, True])
def test_backward_reads_a_column_slice_of_a_wider_gradient_in_place(dtype, c, left, right, tail):
    """`wcn_bn_train_backward_ld` (include/wcn.h): the gradient `torch.cat` hands to one of its inputs (reference
    models/mink_unet.py:392-404) is a column slice of a wider row-major tensor; the reduce and apply passes read it with its row
    pitch.  Same sums, same gradients, bit for bit, as on a contiguous copy - 16-B pieces where pitch and offset allow, the
    element path otherwise; `tail`: the masked passes of a residual tail."""
    from warpconvnet_amd import _lib

    dev = torch.device(DEV)
    L = _lib.lib()
    st = _lib.stream_handle(dev)
    code = _lib.dtype_code(dtype)
    n = 30_011
    torch.manual_seed(c + left)
    y = torch.randn(n, c, device=dev).to(dtype)
    z = torch.randn(n, c, device=dev).to(dtype) if tail else None
    wide = torch.randn(n, left + c + right, device=dev).to(dtype)
    g = wide[:, left:left + c]
    assert not g.is_contiguous() or left + right == 0
    gamma = torch.rand(c, device=dev) + 0.5
    stats = torch.empty(5, c, device=dev)
    stats[0], stats[1] = y.float().mean(0), 1.0 / (y.float().var(0, unbiased=False) + 1e-5).sqrt()
    stats[2], stats[3] = gamma * stats[1], -stats[0] * gamma * stats[1]
    ws = torch.empty(L.wcn_bn_workspace(c), dtype=torch.uint8, device=dev)

    def run(rows, ld):
        sums = torch.empty(2, c, device=dev)
        dx, dres = torch.empty_like(y), (torch.empty_like(y) if tail else None)
        _lib.check(L.wcn_bn_train_backward_ld(_lib.ptr(rows), ld, _lib.ptr(y), _lib.ptr(z), 1, n, c, code, _lib.ptr(stats),
                                              _lib.ptr(gamma), 1, _lib.ptr(sums), _lib.ptr(dx), _lib.ptr(dres), _lib.ptr(ws),
                                              ws.numel(), st), "wcn_bn_train_backward_ld")
        return sums, dx, dres

    a = run(g, wide.shape[1])
    b = run(g.contiguous(), 0)
    vec = 16 // y.element_size()
    if c % vec != 0 or (g.data_ptr() % 16 == 0 and (wide.shape[1] * y.element_size()) % 16 == 0):
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and (not tail or torch.equal(a[2], b[2]))
    else:
        # a misaligned slice takes the element path: another reduction order than the 16-B path of the contiguous copy
        # (`block._grad_rows` hands such slices over as copies, so a network never sees the difference)
        assert rel_max_err(a[0], b[0]) < 1e-5 and rel_max_err(a[1].float(), b[1].float()) < 2e-2
        assert not tail or torch.equal(a[2], b[2])  # (the masked gradient is elementwise)
    assert L.wcn_bn_train_backward_ld(_lib.ptr(g), c - 1, _lib.ptr(y), None, 1, n, c, code, _lib.ptr(stats), _lib.ptr(gamma), 1,
                                      _lib.ptr(a[0]), None, None, _lib.ptr(ws), ws.numel(), st) == -5  # pitch below the row length


def test_fused_blocks_under_a_channel_concatenation_take_the_sliced_gradient():
    """Two fused conv -> BN -> ReLU blocks whose outputs are concatenated and fed to a third (the decoder pattern of
    `models/mink_unet.py`): gradients equal the module-by-module run bit for bit - the sliced gradients reach the BatchNorm
    backward without a contiguous copy (`block._grad_rows`)."""
    import os

    from warpconvnet_amd.geometry.types.voxels import Voxels
    from warpconvnet_amd.nn.modules.sequential import Sequential
    from warpconvnet_amd.nn.modules.sparse_conv import SparseConv3d

    dev = torch.device(DEV)
    coords = torch.from_numpy(scene_u(20000, 14)[:, 1:]).to(dev)
    n = coords.shape[0]
    torch.manual_seed(5)
    a = Sequential(SparseConv3d(32, 96, 3, bias=False), nn.BatchNorm1d(96), nn.ReLU()).to(dev)
    b = Sequential(SparseConv3d(32, 32, 3, bias=False), nn.BatchNorm1d(32), nn.ReLU()).to(dev)
    c = Sequential(SparseConv3d(128, 64, 3, bias=False), nn.BatchNorm1d(64), nn.ReLU()).to(dev)
    feats = torch.randn(n, 32, device=dev)
    off = torch.tensor([0, n], dtype=torch.int32)

    def run(fused):
        os.environ["WARPCONVNET_AMD_FUSED_BLOCK"] = "1" if fused else "0"
        try:
            for m in (a, b, c):
                m.zero_grad(set_to_none=True)
            f = feats.clone().requires_grad_(True)
            x = Voxels(coords, f, offsets=off)
            with torch.autocast("cuda", dtype=torch.bfloat16):
                ya, yb = a(x), b(x)
                cat = ya.replace(batched_features=torch.cat([ya.feature_tensor, yb.feature_tensor], dim=1))
                out = c(cat)
            out.feature_tensor.float().square().mean().backward()
            return [p.grad.clone() for m in (a, b, c) for p in m.parameters()] + [f.grad.clone()]
        finally:
            os.environ.pop("WARPCONVNET_AMD_FUSED_BLOCK", None)

    for u, v in zip(run(True), run(False)):
        assert torch.equal(u, v)


@pytest.mark.parametrize("cin,cout", [(3, 32), (20, 48), (64, 20)])
def test_narrow_stem_node_reads_fp32_rows_and_returns_an_fp32_gradient(cin, cout):
    """A 1 x 1 x 1 conv -> BN -> ReLU block on fp32 features under bf16 autocast whose shape belongs to the narrow-layer kernel
    (`wcn_dense_rows`): the fused node hands the fp32 rows to the kernel (rounded there) instead of casting them first.  Output,
    parameter gradients and the gradient of the fp32 input leaf equal the module-by-module run bit for bit."""
    import os

    from warpconvnet_amd.geometry.types.voxels import Voxels
    from warpconvnet_amd.nn.modules.sequential import Sequential
    from warpconvnet_amd.nn.modules.sparse_conv import SparseConv3d

    dev = torch.device(DEV)
    coords = torch.from_numpy(scene_u(9000, 21)[:, 1:]).to(dev)
    n = coords.shape[0]
    torch.manual_seed(cin + cout)
    blk = Sequential(SparseConv3d(cin, cout, 1, bias=False), nn.BatchNorm1d(cout), nn.ReLU()).to(dev)
    feats = torch.randn(n, cin, device=dev)
    off = torch.tensor([0, n], dtype=torch.int32)

    def run(fused):
        os.environ["WARPCONVNET_AMD_FUSED_BLOCK"] = "1" if fused else "0"
        try:
            blk.zero_grad(set_to_none=True)
            f = feats.clone().requires_grad_(True)
            with torch.autocast("cuda", dtype=torch.bfloat16):
                out = blk(Voxels(coords, f, offsets=off)).feature_tensor
            out.float().square().mean().backward()
            assert f.grad.dtype == torch.float32
            return [out.detach().clone(), f.grad.clone()] + [p.grad.clone() for p in blk.parameters()]
        finally:
            os.environ.pop("WARPCONVNET_AMD_FUSED_BLOCK", None)

    for u, v in zip(run(True), run(False)):
        assert torch.equal(u, v)
