"""GPU: conv + BatchNorm(inference) + residual + ReLU folded into the gather-GEMM epilogue (`wcn_conv_gather_gemm_fused`)
against the fp64 oracle chain on the same (storage-rounded) operands."""
import numpy as np
import pytest
import torch
import torch.nn as nn

from oracle import conv as oconv
from oracle import kmap as okmap
from tests.util import rel_max_err, scene_u

pytestmark = pytest.mark.gpu


def _setup(cin, cout, ksize, stride, dtype, seed=0, n=(3000, 2200)):
    from warpconvnet_amd.geometry.types.voxels import Voxels
    from warpconvnet_amd.nn.modules.sparse_conv import SparseConv3d

    dev = torch.device("cuda:0")
    parts = [scene_u(m, seed + b)[:, 1:] - 5 for b, m in enumerate(n)]
    g = torch.Generator().manual_seed(seed)
    vox = Voxels([torch.from_numpy(p.copy()) for p in parts], [torch.randn(len(p), cin, generator=g) for p in parts], device=dev)
    vox = vox.replace(batched_features=vox.feature_tensor.to(dtype))
    torch.manual_seed(seed)
    conv = SparseConv3d(cin, cout, ksize, stride=stride).to(dev)
    bn = nn.BatchNorm1d(cout).to(dev)
    with torch.no_grad():
        bn.running_mean.normal_(0, 0.3)
        bn.running_var.uniform_(0.5, 2.0)
        bn.weight.normal_(1.0, 0.2)
        bn.bias.normal_(0, 0.2)
    return dev, vox, conv, bn.eval()


def _oracle_chain(vox, conv, bn, ksize, stride, residual, relu, out_coords):
    bc = vox.batch_indexed_coordinates.cpu().numpy().astype(np.int32)
    oc = out_coords.cpu().numpy().astype(np.int32)
    r = okmap.kernel_map(bc, oc, ksize, stride)
    w = conv.weight.detach().to(vox.feature_tensor.dtype).double().cpu()  # the kernel multiplies storage-rounded weights
    y = oconv.forward(vox.feature_tensor.double().cpu(), w, r["in_maps"], r["out_maps"], r["offsets"], len(oc))
    y = y + conv.bias.detach().double().cpu()
    if bn is not None:
        s = bn.weight.double().cpu() / torch.sqrt(bn.running_var.double().cpu() + bn.eps)
        y = (y - bn.running_mean.double().cpu()) * s + bn.bias.double().cpu()
    if residual is not None:
        y = y + residual.double().cpu()
    return torch.relu(y) if relu else y


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("cin,cout", [(64, 128), (32, 64), (16, 32), (64, 256), (128, 96)])
def test_fused_block_matches_oracle(cin, cout, dtype):
    from warpconvnet_amd.nn.modules import FusedSparseConvBlock

    dev, vox, conv, bn = _setup(cin, cout, 3, 1, dtype, seed=cin + cout)
    n = vox.feature_tensor.shape[0]
    res = vox.replace(batched_features=torch.randn(n, cout, device=dev).to(dtype))
    for use_bn, use_res, relu in [(True, True, True), (True, False, True), (False, True, False), (True, False, False), (False, False, True)]:
        block = FusedSparseConvBlock(conv, bn if use_bn else None, relu=relu).eval()
        y = block(vox, res if use_res else None)
        assert y.feature_tensor.dtype == dtype and y.feature_tensor.shape == (n, cout)
        want = _oracle_chain(vox, conv, bn if use_bn else None, (3, 3, 3), (1, 1, 1), res.feature_tensor if use_res else None, relu,
                             y.batch_indexed_coordinates)
        assert rel_max_err(y.feature_tensor, want) < 6e-3, (use_bn, use_res, relu)
        if relu:
            assert (y.feature_tensor >= 0).all()


def test_fused_equals_unfused_modules_and_strided():
    """Against the module chain the reference runs (SparseConv3d -> BatchNorm1d -> ReLU), stride-2 k=2 and k=3."""
    from warpconvnet_amd.nn.modules import FusedSparseConvBlock

    for ksize, stride in ((2, 2), (3, 2), (3, 1)):
        dev, vox, conv, bn = _setup(32, 64, ksize, stride, torch.bfloat16, seed=ksize * 10 + stride)
        conv.eval()
        y_conv = conv(vox)
        unfused = torch.relu(bn(y_conv.feature_tensor.float())).to(torch.bfloat16)
        fused = FusedSparseConvBlock(conv, bn).eval()(vox)
        assert torch.equal(fused.coordinate_tensor, y_conv.coordinate_tensor) and fused.tensor_stride == y_conv.tensor_stride
        assert rel_max_err(fused.feature_tensor, unfused) < 1.5e-2  # the unfused chain rounds to bf16 twice
        want = _oracle_chain(vox, conv, bn, (ksize,) * 3, (stride,) * 3, None, True, fused.batch_indexed_coordinates)
        assert rel_max_err(fused.feature_tensor, want) < 6e-3


def test_fused_fallback_shapes_and_errors():
    """fp32 features and channel counts outside the MFMA set run the same chain unfused (HIP conv + elementwise)."""
    from warpconvnet_amd.nn.functional.sparse_conv.fused import fused_sparse_conv_inference
    from warpconvnet_amd.nn.modules import FusedSparseConvBlock

    dev, vox, conv, bn = _setup(3, 20, 3, 1, torch.float32, seed=9)
    y = FusedSparseConvBlock(conv, bn).eval()(vox)
    want = _oracle_chain(vox, conv, bn, (3, 3, 3), (1, 1, 1), None, True, y.batch_indexed_coordinates)
    assert y.feature_tensor.dtype == torch.float32 and rel_max_err(y.feature_tensor, want) < 1e-3
    block = FusedSparseConvBlock(conv, bn)
    with pytest.raises(RuntimeError):
        block.train()(vox)
    with pytest.raises(ValueError):
        fused_sparse_conv_inference(vox, conv.weight, 3, scale=torch.ones(20, device=dev))  # scale without shift
    # autocast picks the compute dtype like spatially_sparse_conv
    dev, vox, conv, bn = _setup(64, 64, 3, 1, torch.float32, seed=4)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y = FusedSparseConvBlock(conv, bn).eval()(vox)
    assert y.feature_tensor.dtype == torch.bfloat16
