"""CPU: host-side mirror of the reference API (geometry types, containers, module init, explicit backend)."""
import os

import numpy as np
import pytest
import torch

from oracle import kmap as okmap
from tests.util import scene_u
from warpconvnet_amd.geometry.coords.search.cache import IntSearchCache, IntSearchCacheKey
from warpconvnet_amd.geometry.coords.search.search_results import IntSearchResult, RealSearchResult
from warpconvnet_amd.geometry.coords.search.torch_discrete import kernel_offsets_from_size
from warpconvnet_amd.geometry.types.points import Points
from warpconvnet_amd.geometry.types.voxels import Voxels
from warpconvnet_amd.nn.modules.sparse_conv import SparseConv2d, SparseConv3d


def test_kernel_offsets_match_reference_tables(golden_dir):
    g = np.load(os.path.join(golden_dir, "offset_tables.npz"))
    for name in g.files:
        ks, dl = tuple(int(c) for c in name[1:4]), tuple(int(c) for c in name[6:9])
        np.testing.assert_array_equal(kernel_offsets_from_size(ks, dl).numpy(), g[name])


def test_int_search_result_container_matches_reference(golden_dir):
    g = np.load(os.path.join(golden_dir, "search_result_api.npz"))
    r = IntSearchResult(torch.from_numpy(g["in_maps"]), torch.from_numpy(g["out_maps"]), torch.from_numpy(g["offsets"]))
    assert len(r) == int(g["length"]) and [r.numel(i) for i in range(3)] == g["numel"].tolist()
    assert r.offsets.device.type == "cpu"
    np.testing.assert_array_equal(r[1][0].numpy(), g["item1_in"])
    np.testing.assert_array_equal(r[2][0].numpy(), g["item2_in"])
    np.testing.assert_array_equal(r[2][1].numpy(), g["item2_out"])
    ci, co, coff = r.to_csr()
    np.testing.assert_array_equal(co.numpy(), g["csr_out"])
    np.testing.assert_array_equal(coff.numpy(), g["csr_off"])
    # the reference's torch.sort is not stable: compare in-rows as sets per output row
    for j in range(len(g["csr_out"])):
        s, e = int(g["csr_off"][j]), int(g["csr_off"][j + 1])
        assert sorted(ci[s:e].tolist()) == sorted(g["csr_in"][s:e].tolist())
    np.testing.assert_array_equal(r.neighbor_count_per_output(5).numpy(), g["counts"])
    assert [len(a) for a, _ in r] == g["numel"].tolist()
    with pytest.raises(AssertionError):
        IntSearchResult(torch.zeros(3, dtype=torch.int32), torch.zeros(3, dtype=torch.int32), torch.tensor([0, 2]))
    knn = RealSearchResult(torch.arange(12).reshape(4, 3))
    assert knn.neighbor_row_splits.tolist() == [0, 3, 6, 9, 12]


def test_cache_key_truth_table(golden_dir):
    g = np.load(os.path.join(golden_dir, "cache_key.npz"))
    o1, o2 = torch.tensor([0, 5, 9]), torch.tensor([0, 5, 10])
    base = dict(kernel_size=(3, 3, 3), kernel_dilation=(1, 1, 1), transposed=False, generative=False,
                stride_mode="stride_only", skip_symmetric_kernel_map=False, in_offsets=o1, out_offsets=o1)
    variants = [dict(), dict(kernel_size=(2, 2, 2)), dict(kernel_dilation=(2, 2, 2)), dict(transposed=True),
                dict(in_offsets=o2), dict(out_offsets=o2)]
    k0 = IntSearchCacheKey(**base)
    assert [bool(k0 == IntSearchCacheKey(**{**base, **v})) for v in variants] == g["equal"].tolist()
    assert hash(k0) == hash(IntSearchCacheKey(**base))
    c = IntSearchCache()
    assert c.get(k0) is None
    c.put(k0, "x")
    assert c.get(IntSearchCacheKey(**base)) == "x"


def test_module_init_matches_reference_state_dict(golden_dir):
    """Same torch version + same RNG stream => the seeded initial weights equal the reference's bit for bit."""
    g = np.load(os.path.join(golden_dir, "module_init.npz"))
    for name, kwargs in [("c16x32_k3", dict(in_channels=16, out_channels=32, kernel_size=3)),
                         ("c64x128_k3_g4", dict(in_channels=64, out_channels=128, kernel_size=3, groups=4)),
                         ("c32x16_k2_s2_tr", dict(in_channels=32, out_channels=16, kernel_size=2, stride=2, transposed=True))]:
        torch.manual_seed(0)
        m = SparseConv3d(**kwargs)
        np.testing.assert_array_equal(m.weight.detach().numpy(), g[name + "_weight"])
        np.testing.assert_array_equal(m.bias.detach().numpy(), g[name + "_bias"])
    m = SparseConv3d(16, 32, 3)
    bound = (3 ** 0.5) * (2.0 / (1 + 5)) ** 0.5 / (16 * 27) ** 0.5
    assert m.weight.shape == (27, 16, 32) and float(m.weight.detach().abs().max()) <= bound + 1e-7
    assert SparseConv2d(8, 8, 3).weight.shape == (9, 8, 8)
    assert SparseConv3d(8, 8, 3, bias=False).bias is None
    with pytest.raises(ValueError):
        SparseConv3d(10, 8, 3, groups=4)
    with pytest.raises(ValueError):
        SparseConv3d(8, 8, 3, fwd_algo="not_an_algo")


def test_voxels_behaviour():
    """Behaviours pinned by the reference's type tests (tests/types/test_voxels.py:27-255)."""
    c = [torch.randint(0, 10, (5, 3), dtype=torch.int32), torch.randint(0, 10, (7, 3), dtype=torch.int32)]
    f = [torch.randn(5, 4), torch.randn(7, 4)]
    v = Voxels(c, f, voxel_size=0.02, tag="x")
    assert v.offsets.tolist() == [0, 5, 12] and v.offsets.dtype == torch.int32 and v.batch_size == 2
    assert v.batch_indexed_coordinates.shape == (12, 4) and v.batch_indexed_coordinates[:, 0].tolist() == [0] * 5 + [1] * 7
    v2 = v.replace(batched_features=torch.zeros(12, 8))
    assert v2.extra_attributes == {"voxel_size": 0.02, "tag": "x"} and v2.num_channels == 8 and v.num_channels == 4
    assert torch.equal((v + 1.0).feature_tensor, v.feature_tensor + 1.0) and torch.equal((2 * v).feature_tensor, 2 * v.feature_tensor)
    h = v.to(dtype=torch.float16)
    assert h.dtype == torch.float16 and v.dtype == torch.float32
    assert v.feature_tensor.dtype == torch.float32  # (autocast behaviour is covered by the GPU module test)
    dup = Voxels([torch.tensor([[1, 1, 1], [2, 2, 2], [1, 1, 1]], dtype=torch.int32)], [torch.arange(6.0).reshape(3, 2)])
    u = dup.unique()
    assert len(u) == 2 and u.feature_tensor.tolist() == [[0.0, 1.0], [2.0, 3.0]]  # first occurrence is kept
    d = v.to_dense()
    back = Voxels.from_dense(d, target_spatial_sparse_tensor=v)
    assert back.feature_tensor.shape == (12, 4)
    assert v[1].feature_tensor.shape == (7, 4) and v.tensor_stride is None
    v.set_tensor_stride(2)
    assert v.tensor_stride == (2, 2, 2)
    with pytest.raises(AssertionError):
        Voxels(torch.zeros(3, 3, dtype=torch.int32), torch.zeros(3, 2))  # tensor input needs offsets
    p = Points([torch.rand(10, 3), torch.rand(12, 3)], [torch.rand(10, 2), torch.rand(12, 2)])
    assert p.to_voxels(0.3).num_channels == 2 and p.voxel_downsample(0.5).batch_size == 2


def _attach_oracle_map(vox, ksize=(3, 3, 3)):
    """CPU plumbing (BASELINE config 0): the kernel map comes from the oracle and is put in the cache, the product's
    explicit_gemm backend does the arithmetic."""
    from warpconvnet_amd.nn.functional.sparse_conv import STRIDED_CONV_MODE

    bc = vox.batch_indexed_coordinates.numpy().astype(np.int32)
    r = okmap.kernel_map(bc, bc, ksize)
    km = IntSearchResult(torch.from_numpy(r["in_maps"]), torch.from_numpy(r["out_maps"]), torch.from_numpy(r["offsets"]), 13)
    key = IntSearchCacheKey(ksize, (1, 1, 1), False, False, str(STRIDED_CONV_MODE.STRIDE_ONLY), False, vox.offsets, vox.offsets)
    vox._extra_attributes["_cache"] = IntSearchCache()
    vox.cache.put(key, km)
    return r


def test_config0_cpu_explicit_module_vs_oracle():
    """8k synthetic voxels, SparseConv3d 16 -> 32 k=3, explicit gather-matmul-scatter on PyTorch CPU."""
    from oracle import conv as oconv

    s = scene_u(8000, 0)
    feats = torch.randn(len(s), 16)
    vox = Voxels([torch.from_numpy(s[:, 1:])], [feats])
    r = _attach_oracle_map(vox)
    torch.manual_seed(0)
    conv = SparseConv3d(16, 32, 3)
    x = vox.replace(batched_features=feats.clone().requires_grad_(True))
    y = conv(x)
    assert y.feature_tensor.shape == (len(s), 32) and y.cache is x.cache
    y.feature_tensor.square().sum().backward()
    Yr = oconv.forward(feats, conv.weight.detach(), r["in_maps"], r["out_maps"], r["offsets"], len(s), 13) + conv.bias.detach()
    torch.testing.assert_close(y.feature_tensor.detach(), Yr, rtol=1e-5, atol=1e-5)
    dXr, dWr = oconv.backward(2 * Yr, feats, conv.weight.detach(), r["in_maps"], r["out_maps"], r["offsets"], 13)
    torch.testing.assert_close(conv.weight.grad, dWr, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(x.batched_features.batched_tensor.grad, dXr, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(conv.bias.grad, (2 * Yr).sum(0), rtol=1e-4, atol=1e-3)


def test_explicit_backend_matches_golden(golden_dir):
    from warpconvnet_amd.nn.functional.sparse_conv.detail.explicit import (
        _explicit_gemm_backward_logic,
        _explicit_gemm_forward_logic,
    )

    for name in ("explicit_u2048_16x32_f32.npz", "explicit_stride2_k2_16x32_f32.npz", "explicit_b2_7x13_f32_noiden.npz"):
        g = np.load(os.path.join(golden_dir, name))
        iden = None if int(g["identity"]) < 0 else int(g["identity"])
        km = IntSearchResult(torch.from_numpy(g["in_maps"]), torch.from_numpy(g["out_maps"]), torch.from_numpy(g["offsets"]), iden)
        X, W, dY = (torch.from_numpy(g[k]) for k in ("X", "W", "dY"))
        Y = _explicit_gemm_forward_logic(X, W, km, g["out_coords"].shape[0])
        dX, dW = _explicit_gemm_backward_logic(dY, X, W, km)
        torch.testing.assert_close(Y, torch.from_numpy(g["Y"]), rtol=1e-5, atol=1e-5)
        torch.testing.assert_close(dX, torch.from_numpy(g["dX"]), rtol=1e-5, atol=1e-5)
        torch.testing.assert_close(dW, torch.from_numpy(g["dW"]), rtol=1e-5, atol=1e-4)


def test_one_by_one_conv_and_errors():
    v = Voxels([torch.randint(0, 5, (6, 3), dtype=torch.int32)], [torch.randn(6, 4)])
    conv = SparseConv3d(4, 8, 1)
    torch.testing.assert_close(conv(v).feature_tensor, v.feature_tensor @ conv.weight[0] + conv.bias)
    with pytest.raises(TypeError):
        conv(torch.zeros(3, 4))
    with pytest.raises(RuntimeError):  # CPU voxels without a cached map: kernel-map construction is GPU-only
        SparseConv3d(4, 8, 3)(v)
    with pytest.raises(ValueError, match="no CPU fallback"):  # generative: coordinate expansion is GPU-only, like the reference
        SparseConv3d(4, 8, 3, generative=True)(v)


# ---- depthwise (CPU: explicit backend) -----------------------------------------------------------------------------
def test_depthwise_module_init_matches_reference_state_dict(golden_dir):
    from warpconvnet_amd.nn.modules.sparse_conv_depth import SparseDepthwiseConv3d

    g = np.load(os.path.join(golden_dir, "module_init_depthwise.npz"))
    for name, kwargs in [("c64_k3", dict(channels=64, kernel_size=3)),
                         ("c32_k2_s2_tr", dict(channels=32, kernel_size=2, stride=2, transposed=True))]:
        torch.manual_seed(0)
        m = SparseDepthwiseConv3d(**kwargs)
        np.testing.assert_array_equal(m.weight.detach().numpy(), g[name + "_weight"])
        np.testing.assert_array_equal(m.bias.detach().numpy(), g[name + "_bias"])
        assert m.weight.shape == (int(np.prod(m.kernel_size)), kwargs["channels"])


@pytest.mark.parametrize("name", ["depthwise_u600_c64_f32.npz", "depthwise_u600_c64_f64.npz", "depthwise_b2_c13_f32_noiden.npz",
                                  "depthwise_stride2_k2_c32_f32.npz"])
def test_depthwise_explicit_backend_matches_golden(golden_dir, name):
    """CPU tensors take the explicit backend; autograd function == the reference's explicit depthwise outputs."""
    from warpconvnet_amd.nn.functional.sparse_conv_depth import spatially_sparse_depthwise_conv

    g = np.load(os.path.join(golden_dir, name))
    iden = int(g["identity"])
    km = IntSearchResult(torch.from_numpy(g["in_maps"]), torch.from_numpy(g["out_maps"]), torch.from_numpy(g["offsets"]),
                         None if iden < 0 else iden)
    X = torch.from_numpy(g["X"]).requires_grad_(True)
    W = torch.from_numpy(g["W"]).requires_grad_(True)
    Y = spatially_sparse_depthwise_conv(X, W, km, g["out_coords"].shape[0], fwd_algo="auto", bwd_algo="explicit_gemm")
    Y.backward(torch.from_numpy(g["dY"]))
    tol = 1e-12 if g["X"].dtype == np.float64 else 1e-6
    for got, want in ((Y.detach(), g["Y"]), (X.grad, g["dX"]), (W.grad, g["dW"])):
        want = torch.from_numpy(want)
        assert got.dtype == want.dtype and got.shape == want.shape
        assert (got - want).abs().max().item() <= tol * max(1.0, want.abs().max().item())
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        spatially_sparse_depthwise_conv(X, W, km, g["out_coords"].shape[0], fwd_algo="implicit")
    with pytest.raises(ValueError):
        spatially_sparse_depthwise_conv(X, W[:, :-1], km, g["out_coords"].shape[0])


def _grouped_oracle(X, W4, dY, r, n_out, bias=None):
    """Channel groups = G independent convolutions on channel slices (reference weight layout [K, G, Cin/G, Cout/G])."""
    from oracle import conv as oconv

    G = W4.shape[1]
    cg_in, cg_out = X.shape[1] // G, dY.shape[1] // G
    Ys, dXs, dWs = [], [], []
    for g in range(G):
        Xg, Wg, dYg = X[:, g * cg_in:(g + 1) * cg_in], W4[:, g], dY[:, g * cg_out:(g + 1) * cg_out]
        Ys.append(oconv.forward(Xg, Wg, r["in_maps"], r["out_maps"], r["offsets"], n_out))
        dx, dw = oconv.backward(dYg, Xg, Wg, r["in_maps"], r["out_maps"], r["offsets"])
        dXs.append(dx)
        dWs.append(dw)
    Y = torch.cat(Ys, 1)
    if bias is not None:
        Y = Y + bias
    return Y, torch.cat(dXs, 1), torch.stack(dWs, 1)


def test_grouped_conv_cpu_explicit_module_vs_oracle():
    """SparseConv3d(groups=4) on CPU tensors (explicit backend) == 4 independent convolutions on channel slices."""
    bc = scene_u(1500, 9)
    r = okmap.kernel_map(bc, bc, (3, 3, 3))
    torch.manual_seed(3)
    conv = SparseConv3d(16, 24, 3, groups=4, fwd_algo="explicit_gemm", dgrad_algo="explicit_gemm", wgrad_algo="explicit_gemm").double()
    X = torch.randn(len(bc), 16, dtype=torch.float64, requires_grad=True)
    km = IntSearchResult(torch.from_numpy(r["in_maps"]), torch.from_numpy(r["out_maps"]), torch.from_numpy(r["offsets"]))
    from warpconvnet_amd.nn.functional.sparse_conv.detail.unified import UnifiedSpatiallySparseConvFunction

    Y = UnifiedSpatiallySparseConvFunction.apply(X, conv.weight, km, len(bc), "explicit_gemm", "explicit_gemm", "explicit_gemm",
                                                 None, None, None, None, None, 4, False, conv.bias)
    dY = torch.randn(len(bc), 24, dtype=torch.float64)
    Y.backward(dY)
    Yr, dXr, dWr = _grouped_oracle(X.detach(), conv.weight.detach(), dY, r, len(bc), conv.bias.detach())
    assert conv.weight.shape == (27, 4, 4, 6)
    assert (Y.detach() - Yr).abs().max() < 1e-10 and (X.grad - dXr).abs().max() < 1e-10
    assert (conv.weight.grad - dWr).abs().max() < 1e-10 and (conv.bias.grad - dY.sum(0)).abs().max() < 1e-10


# ---- PointConv (CPU tensors: torch kNN + torch segment reduce = the reference's algorithm) ----------------------------
def _pointconv_case(g, name, device, kw):
    from warpconvnet_amd.geometry.coords.search.search_configs import RealSearchConfig
    from warpconvnet_amd.nn.modules.point_conv import PointConv

    feats = torch.from_numpy(g[name + "_feats"]).to(device).requires_grad_(True)
    pc = Points(torch.from_numpy(g[name + "_coords"]).to(device), feats, offsets=torch.tensor([0, 180, 320]))
    conv = PointConv(8, 16, RealSearchConfig(mode="knn", knn_k=8), **kw)
    state = {k[len(name) + 7:]: torch.from_numpy(g[k]) for k in g.files if k.startswith(name + "_param_")}
    conv.load_state_dict(state, strict=True)  # same parameter / buffer names as the reference module
    conv = conv.to(device)
    y = conv(pc).feature_tensor
    y.backward(torch.from_numpy(g[name + "_dY"]).to(device))
    grads = {k: p.grad for k, p in conv.named_parameters()}
    return y.detach(), feats.grad, grads


@pytest.mark.parametrize("name,kw", [("knn8_relpos_mean_max", dict(use_rel_pos=True, reductions=("mean", "max"))),
                                     ("knn8_plain_sum", dict(reductions=("sum",)))])
def test_pointconv_cpu_matches_reference_golden(golden_dir, name, kw):
    g = np.load(os.path.join(golden_dir, "pointconv.npz"))
    y, dx, grads = _pointconv_case(g, name, torch.device("cpu"), kw)
    torch.testing.assert_close(y, torch.from_numpy(g[name + "_Y"]), rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(dx, torch.from_numpy(g[name + "_dX"]), rtol=1e-4, atol=1e-5)
    for k, v in grads.items():
        torch.testing.assert_close(v, torch.from_numpy(g[f"{name}_grad_{k}"]), rtol=1e-3, atol=1e-4)


def test_row_reduction_cpu_and_voxel_downsample():
    from warpconvnet_amd.ops.reductions import row_reduction

    f = torch.randn(12, 5, dtype=torch.float64, requires_grad=True)
    splits = torch.tensor([0, 4, 4, 9, 12])
    for op, ref in (("sum", lambda t: t.sum(0)), ("mean", lambda t: t.mean(0)), ("max", lambda t: t.max(0).values),
                    ("min", lambda t: t.min(0).values)):
        out = row_reduction(f, splits, op)
        want = torch.stack([ref(f[a:b]) if b > a else torch.zeros(5, dtype=f.dtype) for a, b in zip(splits[:-1], splits[1:])])
        torch.testing.assert_close(out, want)
    assert torch.autograd.gradcheck(lambda t: row_reduction(t, splits, "mean"), (f,))
    assert torch.autograd.gradcheck(lambda t: row_reduction(t, splits, "max"), (f,))
    pc = Points([torch.rand(300, 3)], [torch.randn(300, 4)])
    down = pc.voxel_downsample(0.25, reduction="mean")
    q = torch.floor(pc.coordinate_tensor / 0.25).int()
    uniq, inv = torch.unique(q, dim=0, return_inverse=True)
    want = torch.zeros(len(uniq), 4).index_add_(0, inv, pc.feature_tensor) / torch.bincount(inv).unsqueeze(1)
    assert down.feature_tensor.shape == want.shape
    torch.testing.assert_close(down.feature_tensor, want, rtol=1e-5, atol=1e-6)


def test_intsearchresult_pytree_roundtrip():
    """IntSearchResult is a pytree node with children (in_maps, out_maps, offsets)
    (reference tests/nn/test_torch_compile.py:190-207)."""
    import warpconvnet_amd  # noqa: F401  (registers the node)
    from torch.utils._pytree import tree_flatten, tree_unflatten
    from warpconvnet_amd.geometry.coords.search.search_results import IntSearchResult

    isr = IntSearchResult(torch.arange(10, dtype=torch.int32), torch.arange(10, dtype=torch.int32),
                          torch.tensor([0, 3, 7, 10]), identity_map_index=1)
    flat, spec = tree_flatten(isr)
    assert len(flat) == 3
    back = tree_unflatten(flat, spec)
    assert isinstance(back, IntSearchResult) and back.identity_map_index == 1 and len(back) == 3
    assert torch.equal(back.in_maps, isr.in_maps) and torch.equal(back.offsets, isr.offsets)
    assert back[1][0].tolist() == [3, 4, 5, 6]


def test_radius_search_cpu_contract():
    """Result types / shapes of the radius search front end (reference radius.py:162-291, continuous.py:36-53) on the
    CPU path: int32 index + split, fp32 distance; batched: global int64 ids, no neighbour across batch elements."""
    from warpconvnet_amd.geometry.coords.search.continuous import neighbor_search
    from warpconvnet_amd.geometry.coords.search.radius import batched_radius_search, radius_search
    from warpconvnet_amd.geometry.coords.search.search_configs import RealSearchConfig

    g = torch.Generator().manual_seed(0)
    p, q = torch.rand(300, 3, generator=g), torch.rand(40, 3, generator=g)
    idx, dist, split = radius_search(p, q, 0.25)
    assert idx.dtype == torch.int32 and dist.dtype == torch.float32 and split.dtype == torch.int32 and split.shape == (41,)
    d = torch.cdist(q, p)
    for i in (0, 7, 39):
        row = idx[split[i] : split[i + 1]].long()
        assert set(row.tolist()) == set(torch.nonzero(d[i] <= 0.25).view(-1).tolist())
        assert torch.allclose(dist[split[i] : split[i + 1]], d[i, row])
    e = radius_search(p, torch.empty(0, 3), 0.1)
    assert e[0].shape == (0,) and e[2].tolist() == [0]
    offs_p, offs_q = torch.tensor([0, 100, 300]), torch.tensor([0, 10, 40])
    bi, bd, bs = batched_radius_search(p, offs_p, q, offs_q, 0.25)
    assert bi.dtype == torch.int64 and bs.dtype == torch.int64 and bs.shape == (41,) and int(bs[-1]) == len(bi)
    assert (bi[: bs[10]] < 100).all() and (bi[bs[10] :] >= 100).all()
    r = neighbor_search(p, offs_p, q, offs_q, RealSearchConfig("radius", radius=0.25))
    assert torch.equal(r.neighbor_indices, bi) and torch.equal(r.neighbor_row_splits, bs)


@pytest.mark.parametrize("tag", ["r030", "r075"])
def test_radius_search_cpu_matches_reference_golden(golden_dir, tag):
    """CPU path == the reference's own CPU radius search (tests/golden/radius_search.npz, generated by running
    warpconvnet/geometry/coords/search/radius.py:127-225): same neighbours in the same (ascending index) order."""
    import os

    from warpconvnet_amd.geometry.coords.search.radius import radius_search

    gz = np.load(os.path.join(golden_dir, "radius_search.npz"))
    idx, dist, split = radius_search(torch.from_numpy(gz["points"]), torch.from_numpy(gz["queries"]), float(gz[f"{tag}_radius"]))
    np.testing.assert_array_equal(split.numpy(), gz[f"{tag}_split"])
    np.testing.assert_array_equal(idx.numpy(), gz[f"{tag}_index"])
    np.testing.assert_allclose(dist.numpy(), gz[f"{tag}_distance"], rtol=1e-6, atol=1e-7)


def test_grad_slot_claim_survives_an_abandoned_backward():
    """`dist.claim_grad_slot`: one producer per parameter and backward pass; a claim left behind by a pass that never reached
    AccumulateGrad (`torch.autograd.grad`, an exception mid-backward) must not switch the slot off for the passes that follow."""
    from warpconvnet_amd import dist as wdist

    p = torch.nn.Parameter(torch.zeros(4))
    p._wcn_grad_slot = torch.zeros(4)
    got = []

    class Probe(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x, w):
            return x * 1.0

        @staticmethod
        def backward(ctx, g):
            got.append(wdist.claim_grad_slot(p) is not None)
            got.append(wdist.claim_grad_slot(p) is not None)  # a second producer in the same pass
            return g, None

    x = torch.ones(4, requires_grad=True)
    torch.autograd.grad(Probe.apply(x, p).sum(), x)   # no AccumulateGrad for p: the claim is never released
    assert got == [True, False] and p._wcn_grad_claimed is not False
    got.clear()
    torch.autograd.grad(Probe.apply(x, p).sum(), x)   # a NEW backward pass: the stale claim is recognised
    assert got == [True, False]
    assert wdist.claim_grad_slot(p) is None           # outside a backward pass a held claim stays held
