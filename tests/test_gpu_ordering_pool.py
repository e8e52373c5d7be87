"""GPU: Morton ordering (codes, per-batch permutation, Voxels.sort, `order=` of the convolution) and sparse pooling
(sparse_reduce / unpool / REDUCE_AND_STRIDE / global_pool) against the numpy oracle, through the C-ABI."""
import numpy as np
import pytest
import torch

from oracle import kmap as okmap
from oracle import serialization as oser
from tests.util import rel_max_err, scene_u

pytestmark = pytest.mark.gpu

ORDERS = ["morton_xyz", "morton_xzy", "morton_yxz", "morton_yzx", "morton_zxy", "morton_zyx"]


def _dev():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return torch.device("cuda:0")


def _scene(batches=(3000, 1, 2500), seed=0, shift=-40):
    parts = [scene_u(n, seed + b, b) for b, n in enumerate(batches)]
    s = np.concatenate(parts, 0)
    s[:, 1:] += shift
    offs = np.concatenate([[0], np.cumsum([len(p) for p in parts])]).astype(np.int64)
    return s, offs


@pytest.mark.parametrize("order", ORDERS)
def test_morton_codes_bit_exact(order):
    from warpconvnet_amd.geometry.coords.ops.serialization import POINT_ORDERING, encode, morton_code

    s, _ = _scene()
    dev = _dev()
    c3 = torch.from_numpy(s[:, 1:].copy()).to(dev)
    got = morton_code(c3, order=POINT_ORDERING(order))
    assert got.dtype == torch.int64
    np.testing.assert_array_equal(got.cpu().numpy(), oser.morton_code(s[:, 1:], order))
    c4 = torch.from_numpy(s).to(dev)
    np.testing.assert_array_equal(encode(c4, order=order).cpu().numpy(), oser.morton_code(s, order))
    # float grids: normalise, then truncate
    f = s[:, 1:].astype(np.float32) + 0.75
    np.testing.assert_array_equal(encode(torch.from_numpy(f).to(dev), order=order).cpu().numpy(),
                                  oser.morton_code((f - f.min(0)).astype(np.int32), order))


def test_morton_limits_and_empty():
    from warpconvnet_amd.geometry.coords.ops.serialization import encode, morton_code

    dev = _dev()
    big = np.array([[0, 0, 0], [65535, 65535, 65535], [1000, 2000, 3000], [(1 << 20) - 1, 5, (1 << 20) - 1]], np.int32)
    got = morton_code(torch.from_numpy(big).to(dev)).cpu().numpy()
    np.testing.assert_array_equal(got, oser.morton_code(big))
    assert len(np.unique(got)) == 4 and (got >= 0).all()
    b4 = np.array([[0, 0, 0, 0], [3, 65535, 65535, 65535], [511, 1, 2, 3]], np.int32)
    np.testing.assert_array_equal(morton_code(torch.from_numpy(b4).to(dev)).cpu().numpy(), oser.morton_code(b4))
    e = torch.empty((0, 3), dtype=torch.int32, device=dev)
    assert encode(e).shape == (0,) and morton_code(e).shape == (0,)
    r = encode(e, return_perm=True, return_inverse=True)
    assert r.perm.shape == (0,) and r.inverse_perm.shape == (0,)


def test_encode_permutations():
    from warpconvnet_amd.geometry.coords.ops.serialization import encode

    s, offs = _scene()
    dev = _dev()
    c3 = torch.from_numpy(s[:, 1:].copy()).to(dev)
    r = encode(c3, batch_offsets=torch.from_numpy(offs), order="morton_xyz", return_perm=True, return_inverse=True)
    codes, perm = oser.encode_perm(s[:, 1:], offs, "morton_xyz")
    np.testing.assert_array_equal(r.codes.cpu().numpy(), codes)
    np.testing.assert_array_equal(r.perm.cpu().numpy(), perm)  # coordinates are unique per batch element => unique answer
    assert r.perm.dtype == torch.int64
    sorted_c = c3[r.perm]
    assert torch.equal(sorted_c[r.inverse_perm], c3)
    for b in range(len(offs) - 1):  # every batch element stays in place and is sorted
        seg = r.codes[r.perm][offs[b] : offs[b + 1]]
        assert ((r.perm[offs[b] : offs[b + 1]] >= offs[b]) & (r.perm[offs[b] : offs[b + 1]] < offs[b + 1])).all()
        assert (seg[1:] >= seg[:-1]).all()
    # without offsets: one global sort
    r2 = encode(c3, order="morton_zyx", return_perm=True)
    np.testing.assert_array_equal(r2.perm.cpu().numpy(), oser.encode_perm(s[:, 1:], None, "morton_zyx")[1])
    # random order: a permutation
    r3 = encode(c3, order="random", return_perm=True)
    assert sorted(r3.perm.cpu().tolist()) == list(range(len(s)))


def test_voxels_sort_and_conv_order():
    """`Voxels.sort` and `SparseConv3d(order=...)`: same voxels, same values, rows in z-order per batch element."""
    from warpconvnet_amd.geometry.types.voxels import Voxels
    from warpconvnet_amd.nn.modules.sparse_conv import SparseConv3d

    s, offs = _scene((2500, 1800), seed=5)
    dev = _dev()
    parts = [s[offs[b] : offs[b + 1], 1:] for b in range(2)]
    feats = [torch.randn(len(p), 16) for p in parts]
    vox = Voxels([torch.from_numpy(p.copy()) for p in parts], feats, device=dev)
    sv = vox.sort("morton_xyz")
    _, perm = oser.encode_perm(s[:, 1:], offs, "morton_xyz")
    np.testing.assert_array_equal(sv.coordinate_tensor.cpu().numpy(), s[perm, 1:])
    assert torch.equal(sv.feature_tensor, vox.feature_tensor[torch.from_numpy(perm).to(dev)])
    assert sv.sort("morton_xyz") is sv and torch.equal(sv.offsets, vox.offsets)

    torch.manual_seed(0)
    for stride in (1, 2):
        plain = SparseConv3d(16, 32, 3, stride=stride).to(dev)
        ordered = SparseConv3d(16, 32, 3, stride=stride, order="morton_xyz").to(dev)
        ordered.load_state_dict(plain.state_dict())
        y0, y1 = plain(vox), ordered(vox)
        c0 = y0.batch_indexed_coordinates.cpu().numpy()
        c1 = y1.batch_indexed_coordinates.cpu().numpy()
        o_offs = y0.offsets.numpy()
        _, p = oser.encode_perm(c0[:, 1:], o_offs, "morton_xyz")
        np.testing.assert_array_equal(c1, c0[p])
        assert rel_max_err(y1.feature_tensor, y0.feature_tensor[torch.from_numpy(p).to(dev)]) < 1e-3
        assert torch.equal(y1.offsets, y0.offsets)


def _pool_case(dev, C=24, dtype=torch.float32, seed=3):
    from warpconvnet_amd.geometry.types.voxels import Voxels

    s, offs = _scene((3000, 2000), seed=seed, shift=-9)
    parts = [s[offs[b] : offs[b + 1], 1:] for b in range(2)]
    rng = np.random.default_rng(seed)
    x_np = rng.standard_normal((len(s), C)).astype(np.float32)
    vox = Voxels([torch.from_numpy(p.copy()) for p in parts],
                 [torch.from_numpy(x_np[offs[b] : offs[b + 1]]) for b in range(2)], device=dev)
    if dtype != torch.float32:
        vox = vox.replace(batched_features=vox.feature_tensor.to(dtype))
    return s, x_np, vox


@pytest.mark.parametrize("reduction", ["max", "min", "mean", "sum"])
@pytest.mark.parametrize("ksize,stride", [(2, 2), (3, 2), ((2, 1, 2), (2, 1, 2))])
def test_sparse_reduce_forward_backward(reduction, ksize, stride):
    from warpconvnet_amd.nn.functional.sparse_pool import sparse_reduce

    dev = _dev()
    s, x_np, vox = _pool_case(dev)
    x = vox.replace(batched_features=vox.feature_tensor.detach().clone().requires_grad_(True))
    y = sparse_reduce(x, ksize, stride, reduction)
    ks = (ksize,) * 3 if isinstance(ksize, int) else ksize
    st = (stride,) * 3 if isinstance(stride, int) else stride
    out_np, _ = okmap.stride_coords(s, st)
    np.testing.assert_array_equal(y.batch_indexed_coordinates.cpu().numpy(), out_np)
    assert y.tensor_stride == st
    r = okmap.kernel_map(s, out_np, ks, st)
    want = oser.sparse_reduce(x_np, r["in_maps"], r["out_maps"], len(out_np), reduction)
    assert rel_max_err(y.feature_tensor.detach(), torch.from_numpy(want)) < 1e-5
    # backward against autograd on the same reduction written with torch index ops (fp64, CPU)
    g_np = np.random.default_rng(1).standard_normal(want.shape).astype(np.float32)
    y.feature_tensor.backward(torch.from_numpy(g_np).to(dev))
    xt = torch.from_numpy(x_np).double().requires_grad_(True)
    im, om = torch.from_numpy(r["in_maps"]).long(), torch.from_numpy(r["out_maps"]).long()
    if reduction in ("sum", "mean"):
        ref = torch.zeros(len(out_np), x_np.shape[1], dtype=torch.float64).index_add(0, om, xt[im])
        if reduction == "mean":
            ref = ref / torch.bincount(om, minlength=len(out_np)).clamp_min(1).unsqueeze(1)
    else:
        ref = torch.zeros(len(out_np), x_np.shape[1], dtype=torch.float64).scatter_reduce(
            0, om.unsqueeze(1).expand(-1, x_np.shape[1]), xt[im], "amax" if reduction == "max" else "amin", include_self=False)
    ref.backward(torch.from_numpy(g_np).double())
    assert rel_max_err(x.feature_tensor.grad, xt.grad) < 1e-5  # random floats: no ties, so amax's tie rule is moot


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_sparse_reduce_half_and_odd_channels(dtype):
    from warpconvnet_amd.nn.functional.sparse_pool import sparse_avg_pool, sparse_max_pool

    dev = _dev()
    for C in (64, 13):  # 16-B vector path and the scalar path
        s, x_np, vox = _pool_case(dev, C=C, dtype=dtype, seed=C)
        xq = vox.feature_tensor.double().cpu().numpy()
        out_np, _ = okmap.stride_coords(s, (2, 2, 2))
        r = okmap.kernel_map(s, out_np, (2, 2, 2), (2, 2, 2))
        ymax = sparse_max_pool(vox, 2, 2)
        np.testing.assert_array_equal(ymax.feature_tensor.double().cpu().numpy(),
                                      oser.sparse_reduce(xq, r["in_maps"], r["out_maps"], len(out_np), "max"))  # exact
        yavg = sparse_avg_pool(vox, 2)
        assert rel_max_err(yavg.feature_tensor, torch.from_numpy(oser.sparse_reduce(xq, r["in_maps"], r["out_maps"], len(out_np), "mean"))) < 1e-2


def test_unpool_and_modules():
    from warpconvnet_amd.nn.modules import GlobalPool, SparseMaxPool, SparseUnpool

    dev = _dev()
    s, x_np, vox = _pool_case(dev, C=16)
    x = vox.replace(batched_features=vox.feature_tensor.detach().clone().requires_grad_(True))
    pooled = SparseMaxPool(2, 2)(x)
    up = SparseUnpool(2, 2, concat_unpooled_st=True)(pooled, x)
    assert up.feature_tensor.shape == (len(s), 32) and torch.equal(up.coordinate_tensor, x.coordinate_tensor)
    out_np, _ = okmap.stride_coords(s, (2, 2, 2))
    r = okmap.kernel_map(s, out_np, (2, 2, 2), (2, 2, 2))
    parent = np.empty(len(s), np.int64)
    parent[r["in_maps"]] = r["out_maps"]  # every fine voxel lies in exactly one 2^3 window
    assert torch.equal(up.feature_tensor[:, :16], x.feature_tensor)
    assert torch.equal(up.feature_tensor[:, 16:], pooled.feature_tensor[torch.from_numpy(parent).to(dev)])
    # gradient of (unpool o maxpool): every window's winner collects the gradients of the window's voxels
    g = torch.randn(len(s), 32, device=dev)
    up.feature_tensor.backward(g)
    xt = torch.from_numpy(x_np).double().requires_grad_(True)
    om = torch.from_numpy(parent)
    pm = torch.zeros(len(out_np), 16, dtype=torch.float64).scatter_reduce(0, om.unsqueeze(1).expand(-1, 16), xt, "amax", include_self=False)
    torch.cat([xt, pm[om]], 1).backward(g.double().cpu())
    assert rel_max_err(x.feature_tensor.grad, xt.grad) < 1e-5
    # global pooling: one row per batch element
    gp = GlobalPool("mean")(vox)
    offs = vox.offsets.tolist()
    want = torch.stack([vox.feature_tensor[offs[b] : offs[b + 1]].mean(0) for b in range(2)])
    assert gp.feature_tensor.shape == (2, 16) and rel_max_err(gp.feature_tensor, want) < 1e-5
    assert gp.offsets.tolist() == [0, 1, 2]


def test_reduce_and_stride_conv():
    """stride_mode=REDUCE_AND_STRIDE == max-pool over stride windows, then the same convolution at stride 1."""
    from warpconvnet_amd.nn.functional.sparse_conv import STRIDED_CONV_MODE
    from warpconvnet_amd.nn.functional.sparse_pool import sparse_reduce
    from warpconvnet_amd.nn.modules.sparse_conv import SparseConv3d

    dev = _dev()
    s, x_np, vox = _pool_case(dev, C=16)
    torch.manual_seed(0)
    conv = SparseConv3d(16, 32, 3, stride=2, stride_mode=STRIDED_CONV_MODE.REDUCE_AND_STRIDE).to(dev)
    ref = SparseConv3d(16, 32, 3, stride=1).to(dev)
    ref.load_state_dict(conv.state_dict())
    x = vox.replace(batched_features=vox.feature_tensor.detach().clone().requires_grad_(True))
    y = conv(x)
    assert y.tensor_stride == (2, 2, 2)
    x2 = vox.replace(batched_features=vox.feature_tensor.detach().clone().requires_grad_(True))
    y2 = ref(sparse_reduce(x2, 2, 2, "max"))
    assert torch.equal(y.coordinate_tensor, y2.coordinate_tensor)
    assert rel_max_err(y.feature_tensor.detach(), y2.feature_tensor.detach()) < 1e-6
    y.feature_tensor.square().sum().backward()
    y2.feature_tensor.square().sum().backward()
    assert rel_max_err(x.feature_tensor.grad, x2.feature_tensor.grad) < 1e-6
    assert rel_max_err(conv.weight.grad, ref.weight.grad) < 1e-6
    # oracle for the forward: max-pool, kernel map on the pooled coordinates, gather-GEMM-scatter
    from oracle import conv as oconv

    out_np, _ = okmap.stride_coords(s, (2, 2, 2))
    r = okmap.kernel_map(s, out_np, (2, 2, 2), (2, 2, 2))
    pooled = oser.sparse_reduce(x_np, r["in_maps"], r["out_maps"], len(out_np), "max")
    r2 = okmap.kernel_map(out_np, out_np, (3, 3, 3))
    want = oconv.forward(pooled, conv.weight.detach().double().cpu(), r2["in_maps"], r2["out_maps"], r2["offsets"], len(out_np))
    want = want + conv.bias.detach().double().cpu()
    assert rel_max_err(y.feature_tensor.detach(), want) < 2e-2
