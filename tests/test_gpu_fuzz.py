"""GPU: randomised sweep of the kernel-map builders and the conv kernels against the oracle.

Each case draws a scene (size, density, number of batch items, origin shift, optional duplicate rows), a kernel
shape, stride and dilation from a seeded generator; indices are compared bit-exact, features within the
tolerances of tests/test_gpu_conv.py.  The seeds are fixed, so a failure reproduces with `-k "seed_<n>"`."""
import numpy as np
import pytest
import torch

from oracle import conv as oconv
from oracle import kmap as okmap
from tests.util import rel_max_err

pytestmark = pytest.mark.gpu

KSIZES = [(3, 3, 3), (3, 3, 3), (2, 2, 2), (5, 5, 5), (1, 1, 1), (3, 1, 3), (1, 3, 5), (2, 3, 2), (5, 3, 1), (4, 4, 4)]
STRIDES = [(1, 1, 1), (1, 1, 1), (2, 2, 2), (2, 1, 2), (3, 3, 3), (4, 4, 4)]
DILATIONS = [(1, 1, 1), (1, 1, 1), (2, 2, 2), (1, 2, 3), (4, 1, 2)]  # halos up to 8 cells stay on the cell-table builder


def _draw_scene(rng, duplicates=False):
    nb = int(rng.integers(1, 4))
    parts = []
    for b in range(nb):
        n = int(rng.choice([1, 7, 63, 64, 65, 500, 4097, 20000]))
        density = rng.choice([0.02, 0.125, 0.5, 0.95])
        extent = max(1, int(np.ceil((n / density) ** (1.0 / 3.0))))
        shape = np.maximum(1, (extent * rng.choice([0.3, 1.0, 1.0, 3.0], size=3)).astype(np.int64))
        c = np.stack([rng.integers(0, s, size=int(1.3 * n) + 1) for s in shape], 1)
        _, first = np.unique(c, axis=0, return_index=True)
        c = c[np.sort(first)][:n]
        c = c + rng.integers(-3000, 3000, size=3) * int(rng.integers(0, 2))
        if duplicates and len(c) > 4 and rng.integers(2):
            c = np.concatenate([c, c[rng.integers(0, len(c), size=len(c) // 5)]], 0)  # repeated voxels: smallest row wins
        parts.append(np.concatenate([np.full((len(c), 1), b, np.int64), c], 1))
    return np.concatenate(parts, 0).astype(np.int32)


def _gen(a_np, b_np, ksize, stride, dilation, same):
    from warpconvnet_amd.geometry.coords.search.torch_discrete import generate_kernel_map

    dev = torch.device("cuda:0")
    a = torch.from_numpy(a_np).to(dev)
    b = a if same else torch.from_numpy(b_np).to(dev)
    return generate_kernel_map(a, b, stride, ksize, dilation)


@pytest.mark.parametrize("method", ["auto", "hash"])  # auto = binned builder whenever it supports the case
@pytest.mark.parametrize("seed", range(40), ids=lambda s: f"seed_{s}")
def test_kernel_map_fuzz(seed, method, monkeypatch):
    from warpconvnet_amd.geometry.coords.ops.stride import stride_coords

    monkeypatch.setenv("WARPCONVNET_AMD_KMAP_METHOD", method)
    rng = np.random.default_rng(1000 + seed)
    s = _draw_scene(rng, duplicates=True)
    ksize = KSIZES[int(rng.integers(len(KSIZES)))]
    stride = STRIDES[int(rng.integers(len(STRIDES)))]
    dilation = DILATIONS[int(rng.integers(len(DILATIONS)))]
    if stride == (1, 1, 1):
        out, same = s, True
    else:
        out, _ = okmap.stride_coords(s, stride)
        got, _ = stride_coords(torch.from_numpy(s).cuda(), stride)
        np.testing.assert_array_equal(got.cpu().numpy(), out)
        same = False
    km = _gen(s, out, ksize, stride, dilation, same)
    r = okmap.kernel_map(s, out, ksize, stride, dilation)
    K = len(r["offsets"]) - 1
    np.testing.assert_array_equal(km._pair_table.cpu().numpy(), r["found"])
    np.testing.assert_array_equal(km._nbr.cpu().numpy()[:, :K].T, r["found"])
    np.testing.assert_array_equal(km.offsets.numpy(), r["offsets"])
    np.testing.assert_array_equal(km.in_maps.cpu().numpy(), r["in_maps"])
    np.testing.assert_array_equal(km.out_maps.cpu().numpy(), r["out_maps"])
    np.testing.assert_array_equal(km._mask.cpu().numpy().view(np.uint32), r["mask"])
    perm = km._perm.cpu().numpy()
    assert np.array_equal(np.sort(perm), np.arange(len(out)))


@pytest.mark.parametrize("seed", range(32), ids=lambda s: f"seed_{s}")
def test_conv_fwd_bwd_fuzz(seed):
    """SparseConv3d forward + backward through the MFMA kernels on a random configuration vs the fp64 oracle."""
    from warpconvnet_amd.geometry.types.voxels import Voxels
    from warpconvnet_amd.nn.modules.sparse_conv import SparseConv3d

    rng = np.random.default_rng(7000 + seed)
    s = _draw_scene(rng)
    ksize = KSIZES[int(rng.integers(len(KSIZES)))]
    stride = STRIDES[int(rng.integers(3))]
    cin = int(rng.choice([3, 8, 16, 32, 48, 64, 96, 128]))
    cout = int(rng.choice([8, 16, 32, 64, 96, 128, 256]))
    dtype = [torch.float16, torch.bfloat16, torch.float32][int(rng.integers(3))]
    bias = bool(rng.integers(2))
    dev = torch.device("cuda:0")
    parts = [s[s[:, 0] == b, 1:] for b in range(int(s[:, 0].max()) + 1)]
    conv = SparseConv3d(cin, cout, ksize, stride=stride, bias=bias).to(dev)
    if dtype != torch.float32:
        conv = conv.to(dtype)
    vox = Voxels([torch.from_numpy(p.copy()) for p in parts], [torch.from_numpy(rng.standard_normal((len(p), cin)).astype(np.float32)) for p in parts], device=dev)
    x = vox.replace(batched_features=vox.feature_tensor.detach().to(dtype).requires_grad_(True))
    out = conv(x)
    # a strided output lives on the floor(c / stride) grid (tensor_stride = stride), the grid the oracle maps on
    o_np = out.batch_indexed_coordinates.cpu().numpy().astype(np.int32)
    dy = torch.from_numpy(rng.standard_normal((o_np.shape[0], cout)).astype(np.float32)).to(dev, dtype)
    (out.feature_tensor.float() * dy.float()).sum().backward()

    # oracle on the values the kernels saw (operands already rounded to the compute dtype)
    w = conv.weight.detach().double().cpu()
    xq = x.feature_tensor.detach().double().cpu()
    dyq = dy.double().cpu()
    bc = vox.batch_indexed_coordinates.cpu().numpy().astype(np.int32)
    r = okmap.kernel_map(bc, o_np, ksize, stride)
    y_ref = oconv.forward(xq, w, r["in_maps"], r["out_maps"], r["offsets"], o_np.shape[0])
    if bias:
        y_ref = y_ref + conv.bias.detach().double().cpu()
    dx_ref, dw_ref = oconv.backward(dyq, xq, w, r["in_maps"], r["out_maps"], r["offsets"])
    gx = x.batched_features.batched_tensor.grad
    tol = {torch.float16: 2e-2, torch.bfloat16: 2e-2, torch.float32: 2e-2}[dtype]
    assert rel_max_err(out.feature_tensor.detach(), y_ref) < tol
    assert rel_max_err(gx, dx_ref) < tol
    assert rel_max_err(conv.weight.grad, dw_ref) < tol
    if bias:
        assert rel_max_err(conv.bias.grad, dyq.sum(0)) < tol
