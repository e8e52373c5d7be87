"""GPU: depthwise sparse convolution (HIP, through the C-ABI) vs the CPU oracle and the reference's golden vectors."""
import glob
import os

import numpy as np
import pytest
import torch

from oracle import conv as oconv
from oracle import kmap as okmap
from tests.util import rel_max_err, scene_u

pytestmark = pytest.mark.gpu

# same tolerances as the dense-weight GEMM tests (reference tests/nn/test_kernel_correctness.py:64-65, 139-145)
TOL = {torch.float32: 1e-5, torch.float16: 2e-2, torch.bfloat16: 2e-2}


def _dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _kmap(in_np, out_np, ksize, stride=(1, 1, 1), same=False):
    from warpconvnet_amd.geometry.coords.search.torch_discrete import generate_kernel_map

    a = torch.from_numpy(in_np).to(_dev())
    b = a if same else torch.from_numpy(out_np).to(_dev())
    return generate_kernel_map(a, b, stride, ksize)


@pytest.mark.parametrize("name", sorted(os.path.basename(p) for p in glob.glob(os.path.join(os.path.dirname(__file__), "golden", "depthwise_*f32*.npz"))))
def test_depthwise_golden_vectors_fp32(golden_dir, name):
    from warpconvnet_amd.nn.functional.sparse_conv_depth import spatially_sparse_depthwise_conv

    g = np.load(os.path.join(golden_dir, name))
    dev = _dev()
    same = g["in_coords"].shape == g["out_coords"].shape and (g["in_coords"] == g["out_coords"]).all()
    km = _kmap(g["in_coords"], g["out_coords"], tuple(int(v) for v in g["ksize"]), tuple(int(v) for v in g["stride"]), same=bool(same))
    np.testing.assert_array_equal(km.offsets.numpy(), g["offsets"])
    X = torch.from_numpy(g["X"]).to(dev).requires_grad_(True)
    W = torch.from_numpy(g["W"]).to(dev).requires_grad_(True)
    Y = spatially_sparse_depthwise_conv(X, W, km, g["out_coords"].shape[0], fwd_algo="implicit", bwd_algo="implicit")
    Y.backward(torch.from_numpy(g["dY"]).to(dev))
    for got, want in ((Y.detach(), g["Y"]), (X.grad, g["dX"]), (W.grad, g["dW"])):
        assert rel_max_err(got, torch.from_numpy(want)) < TOL[torch.float32]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("C", [64, 128, 8, 13, 96, 256])
def test_depthwise_vs_oracle_submanifold(dtype, C):
    """fast path (C/VEC a power of two) and generic path (13, 96); dgrad through the k-flipped forward table."""
    from warpconvnet_amd.nn.functional.sparse_conv_depth import spatially_sparse_depthwise_conv

    s = np.concatenate([scene_u(3000, 41, 0), scene_u(1111, 42, 1)], 0)
    km = _kmap(s, s, (3, 3, 3), same=True)
    r = okmap.kernel_map(s, s, (3, 3, 3))
    dev = _dev()
    g = torch.Generator(device="cpu").manual_seed(C)
    X = torch.randn(len(s), C, generator=g).to(dev, dtype).requires_grad_(True)
    W = (torch.randn(27, C, generator=g) * 0.2).to(dev, dtype).requires_grad_(True)
    dY = torch.randn(len(s), C, generator=g).to(dev, dtype)
    Y = spatially_sparse_depthwise_conv(X, W, km, len(s))  # auto -> implicit on the GPU
    Y.backward(dY)
    Xd, Wd, dYd = X.detach().double().cpu(), W.detach().double().cpu(), dY.double().cpu()
    Yr = oconv.depthwise_forward(Xd, Wd, r["in_maps"], r["out_maps"], r["offsets"], len(s))
    dXr, dWr = oconv.depthwise_backward(dYd, Xd, Wd, r["in_maps"], r["out_maps"], r["offsets"])
    assert Y.dtype == dtype and X.grad.dtype == dtype and W.grad.dtype == dtype
    assert rel_max_err(Y.detach(), Yr) < TOL[dtype]
    assert rel_max_err(X.grad, dXr) < TOL[dtype]
    assert rel_max_err(W.grad, dWr) < TOL[dtype]
    # run-to-run determinism (fixed summation order, no atomics)
    X2, W2 = X.detach().clone().requires_grad_(True), W.detach().clone().requires_grad_(True)
    Y2 = spatially_sparse_depthwise_conv(X2, W2, km, len(s))
    Y2.backward(dY)
    assert torch.equal(Y2, Y) and torch.equal(X2.grad, X.grad) and torch.equal(W2.grad, W.grad)
    # explicit (torch) backend on the GPU agrees as well
    X3, W3 = X.detach().clone().requires_grad_(True), W.detach().clone().requires_grad_(True)
    Y3 = spatially_sparse_depthwise_conv(X3, W3, km, len(s), fwd_algo="explicit", bwd_algo="explicit")
    Y3.backward(dY)
    assert rel_max_err(Y3.detach(), Yr) < max(TOL[dtype], 3e-2 if dtype != torch.float32 else 0)


def test_depthwise_strided_and_module():
    """k=2 s=2 down-sampling (reverse table in dgrad) through the module API, bias included."""
    from warpconvnet_amd.geometry.types.voxels import Voxels
    from warpconvnet_amd.nn.modules.sparse_conv_depth import SparseDepthwiseConv3d

    dev = _dev()
    s = scene_u(5000, 43, 0)
    coarse, _ = okmap.stride_coords(s, (2, 2, 2))
    r = okmap.kernel_map(s, coarse, (2, 2, 2), (2, 2, 2))
    torch.manual_seed(1)
    conv = SparseDepthwiseConv3d(32, kernel_size=2, stride=2).to(dev)
    X = torch.randn(len(s), 32, device=dev, requires_grad=True)
    x = Voxels(torch.from_numpy(s[:, 1:]).to(dev), X, offsets=torch.tensor([0, len(s)]))
    y = conv(x)
    assert y.tensor_stride == (2, 2, 2) and y.feature_tensor.shape == (len(coarse), 32)
    # output rows follow the build's deterministic coarse ordering == the oracle's stride_coords order
    np.testing.assert_array_equal(y.batch_indexed_coordinates.cpu().numpy(), coarse)
    dY = torch.randn(len(coarse), 32, device=dev)
    y.feature_tensor.backward(dY)
    Xd, Wd = X.detach().double().cpu(), conv.weight.detach().double().cpu()
    Yr = oconv.depthwise_forward(Xd, Wd, r["in_maps"], r["out_maps"], r["offsets"], len(coarse)) + conv.bias.detach().double().cpu()
    dXr, dWr = oconv.depthwise_backward(dY.double().cpu(), Xd, Wd, r["in_maps"], r["out_maps"], r["offsets"])
    assert rel_max_err(y.feature_tensor.detach(), Yr) < 1e-5
    assert rel_max_err(X.grad, dXr) < 1e-5
    assert rel_max_err(conv.weight.grad, dWr) < 1e-5
    assert rel_max_err(conv.bias.grad, dY.double().sum(0).cpu()) < 1e-5
