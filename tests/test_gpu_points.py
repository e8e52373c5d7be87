"""GPU: point-cloud front end - exact grid kNN, segment reduce, PointConv - vs brute force / torch / the reference golden."""
import os

import numpy as np
import pytest
import torch

from tests.util import rel_max_err

pytestmark = pytest.mark.gpu


def _dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


@pytest.mark.parametrize("n,m,k,shape", [(20000, 5000, 16, "cube"), (3000, 3000, 1, "cube"), (50000, 2000, 32, "slab"),
                                         (700, 700, 64, "line"), (40, 100, 8, "cube")])
def test_knn_grid_equals_brute_force(n, m, k, shape):
    """Distances of the k neighbours equal cdist + topk exactly-ish (fp32 rounding of the squared distance only); index
    sets agree wherever the k-th and (k+1)-th distances differ."""
    from warpconvnet_amd.geometry.coords.search.knn import _knn_cdist, _knn_grid

    g = torch.Generator().manual_seed(n + k)
    scale = {"cube": torch.tensor([1.0, 1.0, 1.0]), "slab": torch.tensor([50.0, 50.0, 4.0]), "line": torch.tensor([100.0, 0.01, 0.01])}[shape]
    ref = (torch.rand(n, 3, generator=g) * scale).to(_dev())
    qry = ((torch.rand(m, 3, generator=g) * 1.2 - 0.1) * scale).to(_dev())  # some queries outside the bounding box
    idx, d2 = _knn_grid(ref, qry, k, return_dist2=True)
    assert idx.shape == (m, k) and idx.dtype == torch.int64 and (idx >= 0).all() and (idx < n).all()
    want = _knn_cdist(ref.double(), qry.double(), k)
    dw = ((ref.double()[want] - qry.double().unsqueeze(1)) ** 2).sum(-1)
    dg = ((ref.double()[idx] - qry.double().unsqueeze(1)) ** 2).sum(-1)
    assert (dg[:, 1:] >= dg[:, :-1] * (1 - 1e-5) - 1e-7).all()  # ascending (the kernel orders by the fp32 squared distance)
    torch.testing.assert_close(dg, dw, rtol=1e-5, atol=1e-7)
    torch.testing.assert_close(d2.double(), dg, rtol=1e-4, atol=1e-6)
    same = (torch.sort(idx, 1).values == torch.sort(want, 1).values).all(1)
    assert same.float().mean() > 0.99  # differences only from exact ties


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_segment_reduce_vs_torch(dtype):
    from warpconvnet_amd.ops.reductions import row_reduction

    dev = _dev()
    g = torch.Generator().manual_seed(3)
    counts = torch.randint(0, 9, (4000,), generator=g)
    splits = torch.cat([torch.zeros(1, dtype=torch.long), counts.cumsum(0)])
    f = torch.randn(int(splits[-1]), 48, generator=g).to(dtype)
    for op in ("sum", "mean", "max", "min"):
        fc = f.float().clone().requires_grad_(True)
        fg = f.to(dev).requires_grad_(True)
        oc, og = row_reduction(fc, splits, op), row_reduction(fg, splits.to(dev), op)
        tol = 1e-5 if dtype == torch.float32 else 2e-2
        torch.testing.assert_close(og.float().cpu(), oc, rtol=tol, atol=tol)
        w = torch.randn(oc.shape, generator=g)
        oc.backward(w)
        og.backward(w.to(dev, dtype))
        torch.testing.assert_close(fg.grad.float().cpu(), fc.grad, rtol=tol, atol=tol)


@pytest.mark.parametrize("name,kw", [("knn8_relpos_mean_max", dict(use_rel_pos=True, reductions=("mean", "max"))),
                                     ("knn8_plain_sum", dict(reductions=("sum",)))])
def test_pointconv_gpu_matches_reference_golden(golden_dir, name, kw):
    from tests.test_host_api import _pointconv_case

    g = np.load(os.path.join(golden_dir, "pointconv.npz"))
    y, dx, grads = _pointconv_case(g, name, _dev(), kw)
    torch.testing.assert_close(y.cpu(), torch.from_numpy(g[name + "_Y"]), rtol=1e-3, atol=1e-4)
    torch.testing.assert_close(dx.cpu(), torch.from_numpy(g[name + "_dX"]), rtol=1e-3, atol=1e-4)
    for k, v in grads.items():
        torch.testing.assert_close(v.cpu(), torch.from_numpy(g[f"{name}_grad_{k}"]), rtol=5e-3, atol=5e-4)


def test_pointconv_config5_full_size_parity():
    """BASELINE config 5 at FULL size: 200 k fp32 points, PointConv(32 -> 64, knn 16), voxelise, depthwise k=3 conv.
    kNN against brute force and PointConv output rows against the CPU evaluation of the same module on a 2 000-query
    subsample (the full problem is 4e10 distance evaluations on the CPU); the depthwise convolution, forward and both
    gradients, against oracle.conv.depthwise_* on the oracle's kernel map for ALL voxels."""
    import copy

    from oracle import conv as oconv
    from oracle import kmap as okmap
    from warpconvnet_amd.geometry.coords.search.search_configs import RealSearchConfig
    from warpconvnet_amd.geometry.types.points import Points
    from warpconvnet_amd.nn.modules import PointConv, SparseDepthwiseConv3d
    from warpconvnet_amd.ops.reductions import row_reduction

    dev = _dev()
    g = torch.Generator().manual_seed(5)
    n, k = 200_000, 16
    coords = (torch.rand(n, 3, generator=g) * torch.tensor([50.0, 50.0, 4.0])).to(dev)
    feats = torch.randn(n, 32, generator=g)
    pc = Points(coords, feats.to(dev), offsets=torch.tensor([0, n]))
    cfg = RealSearchConfig(mode="knn", knn_k=k)
    torch.manual_seed(0)
    conv = PointConv(32, 64, cfg).to(dev)
    out = conv(pc)
    assert out.feature_tensor.shape == (n, 64) and torch.isfinite(out.feature_tensor).all()

    # ---- kNN of all 200 k queries, checked on a subsample against brute force in fp64 ----
    nb = pc.neighbors(query_coords=pc.batched_coordinates, search_args=cfg)
    idx = nb.neighbor_indices.long().view(n, k)
    sub = torch.randperm(n, generator=g)[:2000].to(dev)
    cd = coords.double()
    for lo in range(0, len(sub), 500):
        q = sub[lo : lo + 500]
        d2 = ((cd[q].unsqueeze(1) - cd.unsqueeze(0)) ** 2).sum(-1)  # [500, n]
        want_d, want_i = torch.topk(d2, k, dim=1, largest=False)
        got_d = torch.gather(d2, 1, idx[q])
        torch.testing.assert_close(torch.sort(got_d, 1).values, want_d, rtol=1e-5, atol=1e-9)
        same = (torch.sort(idx[q], 1).values == torch.sort(want_i, 1).values).all(1)
        assert same.float().mean() > 0.99  # differences only from exact distance ties

    # ---- PointConv rows of the subsample: the same module evaluated on the CPU from the (verified) neighbour lists ----
    ref = copy.deepcopy(conv).cpu()
    q = sub.cpu()
    fi = feats[idx[sub].cpu().view(-1)]                     # neighbour features  [2000*k, 32]
    fq = feats[q].repeat_interleave(k, dim=0)              # query features      [2000*k, 32]
    edge = [fi, fq]
    if conv.use_rel_pos or conv.use_rel_pos_encode:
        rel = coords.cpu()[idx[sub].cpu().view(-1)] - coords.cpu()[q].repeat_interleave(k, dim=0)
        edge.append(ref.positional_encoding(rel) if conv.use_rel_pos_encode else rel)
    with torch.no_grad():
        e = ref.edge_transform_mlp(torch.cat(edge, 1))
        splits = torch.arange(0, len(q) * k + 1, k)
        want = ref.out_transform_mlp(torch.cat([row_reduction(e, splits, reduction=r) for r in conv.reductions], -1))
    torch.testing.assert_close(out.feature_tensor[sub].detach().cpu(), want, rtol=1e-3, atol=1e-4)

    # ---- voxelise + depthwise convolution: all voxels against the oracle (kernel map bit-exact, features fp64) ----
    vox = out.to_voxels(0.25)
    torch.manual_seed(1)
    dw = SparseDepthwiseConv3d(64, 3, bias=False).to(dev)
    x = vox.replace(batched_features=vox.feature_tensor.detach().clone().requires_grad_(True))
    y = dw(x)
    gy = torch.randn(y.feature_tensor.shape, generator=g).to(dev)
    y.feature_tensor.backward(gy)
    bc = vox.batch_indexed_coordinates.cpu().numpy()
    r = okmap.kernel_map(bc, bc, (3, 3, 3))
    km = next(iter(x.cache.values()))
    np.testing.assert_array_equal(km.in_maps.cpu().numpy(), r["in_maps"])
    np.testing.assert_array_equal(km.out_maps.cpu().numpy(), r["out_maps"])
    xd, wd = x.feature_tensor.detach().double().cpu(), dw.weight.detach().double().cpu()
    yr = oconv.depthwise_forward(xd, wd, r["in_maps"], r["out_maps"], r["offsets"], len(bc))
    dxr, dwr = oconv.depthwise_backward(gy.double().cpu(), xd, wd, r["in_maps"], r["out_maps"], r["offsets"])
    assert rel_max_err(y.feature_tensor.detach(), yr) < 1e-3
    assert rel_max_err(x.feature_tensor.grad, dxr) < 1e-3 and rel_max_err(dw.weight.grad, dwr) < 1e-3


@pytest.mark.parametrize("n,m,radius,shape", [(20000, 4000, 0.05, "cube"), (5000, 5000, 0.0, "cube"), (30000, 1500, 2.0, "slab"),
                                              (300, 200, 5.0, "cube"), (900, 400, 0.7, "line")])
def test_radius_grid_equals_brute_force(n, m, radius, shape):
    """Every point with dist <= radius, nothing else; per-query rows compared as sets (the order inside a row is
    implementation-defined in the reference too), distances = sqrt of the fp32 squared distance."""
    from warpconvnet_amd.geometry.coords.search.radius import radius_search

    g = torch.Generator().manual_seed(n + m)
    scale = {"cube": torch.tensor([1.0, 1.0, 1.0]), "slab": torch.tensor([50.0, 50.0, 4.0]), "line": torch.tensor([100.0, 0.01, 0.01])}[shape]
    ref = (torch.rand(n, 3, generator=g) * scale).to(_dev())
    qry = ((torch.rand(m, 3, generator=g) * 1.2 - 0.1) * scale).to(_dev())
    if radius == 0.0:
        qry = ref[:m].clone()  # radius 0: a query finds exactly the points at its own position
    idx, dist, split = radius_search(ref, qry, radius)
    assert idx.dtype == torch.int32 and dist.dtype == torch.float32 and split.dtype == torch.int32
    assert split.shape == (m + 1,) and int(split[-1]) == idx.shape[0] == dist.shape[0]
    # brute force with the kernel's own arithmetic (fp32 differences, fp32 sum of squares)
    d2 = ((qry.unsqueeze(1) - ref.unsqueeze(0)) ** 2).sum(-1) if n * m <= 4e7 else None
    if d2 is None:
        d2 = torch.cat([((qry[s : s + 512].unsqueeze(1) - ref.unsqueeze(0)) ** 2).sum(-1) for s in range(0, m, 512)])
    inside = d2 <= radius * radius
    np.testing.assert_array_equal((split[1:] - split[:-1]).cpu().numpy(), inside.sum(1).cpu().numpy())
    rows = torch.repeat_interleave(torch.arange(m, device=_dev()), (split[1:] - split[:-1]).long())
    got = torch.zeros_like(inside)
    got[rows, idx.long()] = True
    # fp32 summation order of the three squares may differ by an ulp from torch's: allow disagreement only on the boundary
    diff = got != inside
    assert (diff.sum() == 0) or ((d2[diff] - radius * radius).abs() <= 1e-6 * max(radius * radius, 1e-12)).all()
    torch.testing.assert_close(dist, d2[rows, idx.long()].sqrt(), rtol=1e-5, atol=1e-7)
    assert (dist <= radius * (1 + 1e-6) + 1e-12).all()


def test_batched_radius_and_pointconv_radius():
    """Batched search (global row ids, no pair across batch elements), `Points.neighbors`, and PointConv on a radius
    graph (ragged rows) against the same module on CPU tensors (cdist brute force + torch reductions)."""
    from warpconvnet_amd.geometry.coords.search.radius import batched_radius_search
    from warpconvnet_amd.geometry.coords.search.search_configs import RealSearchConfig
    from warpconvnet_amd.geometry.types.points import Points
    from warpconvnet_amd.nn.modules.point_conv import PointConv

    g = torch.Generator().manual_seed(3)
    pts = [torch.rand(1500, 3, generator=g), torch.rand(900, 3, generator=g) + 0.5]  # overlapping boxes: batches must not mix
    feats = [torch.randn(len(p), 8, generator=g) for p in pts]
    ref = torch.cat(pts).to(_dev())
    offs = torch.tensor([0, 1500, 2400])
    idx, dist, split = batched_radius_search(ref, offs, ref, offs, 0.08)
    assert idx.dtype == torch.int64 and split.dtype == torch.int64 and split.shape == (2401,) and int(split[-1]) == len(idx)
    rows = torch.repeat_interleave(torch.arange(2400, device=_dev()), split[1:] - split[:-1])
    assert ((rows < 1500) == (idx < 1500)).all()
    want = torch.cat([(torch.cdist(p, p) <= 0.08).sum(1) for p in pts])
    assert ((split[1:] - split[:-1]).cpu() - want).abs().max() <= 1  # cdist's own rounding at the boundary
    cfg = RealSearchConfig("radius", radius=0.08)
    pc_gpu = Points(pts, feats, device=_dev())
    pc_cpu = Points(pts, feats)
    nb = pc_gpu.neighbors(cfg)
    assert int(nb.neighbor_row_splits[-1]) == len(nb.neighbor_indices)
    torch.manual_seed(0)
    conv = PointConv(8, 16, neighbor_search_args=cfg, out_point_type="same")
    y_cpu = conv(pc_cpu)
    conv_gpu = conv.to(_dev())
    x = pc_gpu.replace(batched_features=pc_gpu.feature_tensor.clone().requires_grad_(True))
    y_gpu = conv_gpu(x)
    torch.testing.assert_close(y_gpu.feature_tensor.cpu(), y_cpu.feature_tensor, rtol=2e-3, atol=2e-4)
    y_gpu.feature_tensor.square().sum().backward()
    assert torch.isfinite(x.feature_tensor.grad).all() and x.feature_tensor.grad.abs().sum() > 0


@pytest.mark.parametrize("tag", ["r030", "r075"])
def test_radius_grid_matches_reference_golden(golden_dir, tag):
    """HIP cell-list search vs the reference's CPU radius search on the same points (golden vectors): per-query neighbour
    SETS and distances (the order inside a row is implementation-defined).  The reference thresholds torch.cdist's
    distance (|x|^2 + |y|^2 - 2xy in fp32, ~1e-5 relative error), the kernel the directly summed squared differences: a
    pair that close to the radius may fall on either side, and distances agree to that accuracy."""
    from warpconvnet_amd.geometry.coords.search.radius import radius_search

    gz = np.load(os.path.join(golden_dir, "radius_search.npz"))
    radius = float(gz[f"{tag}_radius"])
    p, q = torch.from_numpy(gz["points"]).to(_dev()), torch.from_numpy(gz["queries"]).to(_dev())
    idx, dist, split = radius_search(p, q, radius)
    idx, dist, split = idx.cpu().numpy(), dist.cpu().numpy(), split.cpu().numpy()
    w_idx, w_dist, w_split = gz[f"{tag}_index"], gz[f"{tag}_distance"], gz[f"{tag}_split"]
    mismatched = 0
    for i in range(len(q)):
        got = dict(zip(idx[split[i] : split[i + 1]].tolist(), dist[split[i] : split[i + 1]].tolist()))
        want = dict(zip(w_idx[w_split[i] : w_split[i + 1]].tolist(), w_dist[w_split[i] : w_split[i + 1]].tolist()))
        for j in set(got) ^ set(want):
            d = got.get(j, want.get(j))
            assert abs(d * d - radius * radius) <= 3e-5, (i, j, d)  # cdist's error lives in d^2 (~eps * |x|^2)
            mismatched += 1
        for j in set(got) & set(want):
            assert abs(got[j] ** 2 - want[j] ** 2) <= 3e-5
    assert mismatched <= 2


@pytest.mark.gpu
@pytest.mark.parametrize("n,cin,cout,k,red,rel", [(5000, 32, 64, 16, "mean", False), (1234, 8, 16, 8, "sum", False),
                                                  (3001, 16, 32, 4, "mean", False), (2000, 8, 19, 32, "mean", True),
                                                  (700, 4, 8, 1, "sum", False), (4097, 24, 48, 2, "mean", False),
                                                  # widths that differ -> Linear shortcut of the MLPBlock
                                                  (3000, 24, 64, 16, "mean", False), (2500, 16, 16, 8, "sum", False),
                                                  (2000, 16, 32, 4, "mean", True), (1500, 30, 12, 8, "mean", False)])
def test_pointconv_fused_edge_kernel_vs_fp64(n, cin, cout, k, red, rel):
    """The one-pass edge pipeline (csrc/pointconv.hip: gather -> Linear -> LN -> ReLU -> Linear -> LN + shortcut ->
    reduction, forward and backward; identity and Linear shortcut) against the same module evaluated op by op in fp64 on the CPU from the same
    neighbour lists: output rows, input-feature gradient, every parameter gradient."""
    import copy

    from warpconvnet_amd.geometry.coords.search.search_configs import RealSearchConfig
    from warpconvnet_amd.geometry.types.points import Points
    from warpconvnet_amd.nn.functional import point_conv as fpc
    from warpconvnet_amd.nn.modules import PointConv
    from warpconvnet_amd.ops.reductions import row_reduction

    dev = _dev()
    g = torch.Generator().manual_seed(n + k)
    coords = torch.rand(n, 3, generator=g) * 4.0
    feats = torch.randn(n, cin, generator=g)
    cfg = RealSearchConfig(mode="knn", knn_k=k)
    torch.manual_seed(1)
    conv = PointConv(cin, cout, cfg, reductions=(red,), use_rel_pos=rel)
    with torch.no_grad():  # non-trivial LayerNorm weights
        for m in conv.modules():
            if isinstance(m, torch.nn.LayerNorm):
                m.weight.uniform_(0.5, 1.5)
                m.bias.uniform_(-0.5, 0.5)
    ref = copy.deepcopy(conv).double()
    conv = conv.to(dev)
    x = feats.to(dev).requires_grad_(True)
    pc = Points(coords.to(dev), x, offsets=torch.tensor([0, n // 3, n]))
    nb = pc.neighbors(query_coords=pc.batched_coordinates, search_args=cfg)
    assert fpc.fused_edge_supported(conv.edge_transform_mlp, x, x, 3 if rel else 0, k, red)
    calls = []
    orig = fpc._FusedEdge.apply
    fpc._FusedEdge.apply = lambda *a: (calls.append(1), orig(*a))[1]
    try:
        out = conv(pc).feature_tensor
    finally:
        fpc._FusedEdge.apply = orig
    assert calls, "the fused edge kernel was not used"
    dy = torch.randn(n, cout, generator=g)
    out.backward(dy.to(dev))

    idx = nb.neighbor_indices.cpu().view(-1)
    xr = feats.double().requires_grad_(True)
    edge = [xr[idx], xr.repeat_interleave(k, dim=0)]
    if rel:
        edge.append((coords[idx] - coords.repeat_interleave(k, dim=0)).double())
    e = ref.edge_transform_mlp(torch.cat(edge, 1))
    want = ref.out_transform_mlp(row_reduction(e, torch.arange(0, n * k + 1, k), reduction=red))
    want.backward(dy.double())
    torch.testing.assert_close(out.detach().cpu().double(), want.detach(), rtol=1e-4, atol=1e-4)
    # input gradient: a ReLU whose pre-activation is within fp32 rounding of zero takes the other branch than in fp64 and
    # moves the 64 gradient entries of that edge - a handful of the 10^7 activations; everything else must agree tightly
    got, wantg = x.grad.cpu().double(), xr.grad
    bad = (got - wantg).abs() > 1e-4 + 1e-3 * wantg.abs()
    assert bad.double().mean() < 2e-3, f"{int(bad.sum())} of {bad.numel()} input-gradient entries off"
    assert float((got - wantg).norm() / wantg.norm()) < 1e-3
    for (name, p), (_, pr) in zip(conv.named_parameters(), ref.named_parameters()):
        scale = float(pr.grad.abs().max()) + 1e-12
        err = float((p.grad.cpu().double() - pr.grad).abs().max()) / scale
        assert err < 1e-3, f"{name}: relative max error {err:.2e}"


@pytest.mark.parametrize("m,cin,cout,hid", [(5000, 64, 64, 128), (777, 16, 32, 48), (33, 7, 5, 9)])
def test_mlp_block_one_kernel_vs_fp64(m, cin, cout, hid):
    """MLPBlock on a plain [M, C] fp32 tensor runs as the edge kernel with k = 1 (identity and Linear shortcut)."""
    import copy

    from warpconvnet_amd.nn.functional import point_conv as fpc
    from warpconvnet_amd.nn.modules.mlp import MLPBlock

    dev = _dev()
    torch.manual_seed(m)
    mlp = MLPBlock(cin, cout, hid)
    ref = copy.deepcopy(mlp).double()
    mlp = mlp.to(dev)
    x = torch.randn(m, cin)
    xg = x.to(dev).requires_grad_(True)
    assert fpc.fused_mlp_block_supported(mlp, xg)
    y = mlp(xg)
    dy = torch.randn(m, cout)
    y.backward(dy.to(dev))
    xr = x.double().requires_grad_(True)
    yr = ref(xr)
    yr.backward(dy.double())
    torch.testing.assert_close(y.detach().cpu().double(), yr.detach(), rtol=1e-4, atol=1e-4)
    assert float((xg.grad.cpu().double() - xr.grad).norm() / xr.grad.norm()) < 1e-3
    for (name, p), (_, pr) in zip(mlp.named_parameters(), ref.named_parameters()):
        err = float((p.grad.cpu().double() - pr.grad).abs().max()) / (float(pr.grad.abs().max()) + 1e-12)
        assert err < 1e-3, f"{name}: relative max error {err:.2e}"


@pytest.mark.parametrize("n,cin,cout,mode,arg,red,rel", [(3000, 32, 64, "radius", 0.45, "mean", False),
                                                         (2000, 16, 32, "radius", 0.6, "sum", False),
                                                         (2500, 24, 64, "radius", 0.5, "mean", True),
                                                         (1800, 32, 64, "knn", 20, "mean", False),
                                                         (900, 8, 16, "radius", 0.05, "mean", False)])
def test_pointconv_fused_edge_kernel_ragged_lists_vs_fp64(n, cin, cout, mode, arg, red, rel):
    """Ragged neighbour lists (radius search; kNN with a list length that is not a power of two; queries with empty lists)
    through the one-pass edge pipeline: list segments that straddle 32-edge tiles are added to their rows."""
    import copy

    from warpconvnet_amd.geometry.coords.search.search_configs import RealSearchConfig
    from warpconvnet_amd.geometry.types.points import Points
    from warpconvnet_amd.nn.functional import point_conv as fpc
    from warpconvnet_amd.nn.modules import PointConv
    from warpconvnet_amd.ops.reductions import row_reduction

    dev = _dev()
    g = torch.Generator().manual_seed(n)
    coords = torch.rand(n, 3, generator=g) * 4.0
    feats = torch.randn(n, cin, generator=g)
    cfg = RealSearchConfig(mode="radius", radius=arg) if mode == "radius" else RealSearchConfig(mode="knn", knn_k=arg)
    torch.manual_seed(2)
    conv = PointConv(cin, cout, cfg, reductions=(red,), use_rel_pos=rel)
    ref = copy.deepcopy(conv).double()
    conv = conv.to(dev)
    x = feats.to(dev).requires_grad_(True)
    pc = Points(coords.to(dev), x, offsets=torch.tensor([0, n // 2, n]))
    nb = pc.neighbors(query_coords=pc.batched_coordinates, search_args=cfg)
    calls = []
    orig = fpc._FusedEdge.apply
    fpc._FusedEdge.apply = lambda *a: (calls.append(a[-2] is not None), orig(*a))[1]
    try:
        out = conv(pc).feature_tensor
    finally:
        fpc._FusedEdge.apply = orig
    assert calls and calls[0], "the ragged form of the fused edge kernel was not used"
    dy = torch.randn(n, cout, generator=g)
    out.backward(dy.to(dev))

    idx = nb.neighbor_indices.cpu().view(-1)
    splits = nb.neighbor_row_splits.cpu()
    counts = splits[1:] - splits[:-1]
    if mode == "radius" and arg < 0.1:
        assert (counts <= 1).any()  # (nearly) empty lists are part of the case
    xr = feats.double().requires_grad_(True)
    edge = [xr[idx], xr.repeat_interleave(counts, dim=0)]
    if rel:
        edge.append((coords[idx] - coords.repeat_interleave(counts, dim=0)).double())
    e = ref.edge_transform_mlp(torch.cat(edge, 1))
    want = ref.out_transform_mlp(row_reduction(e, splits, reduction=red))
    want.backward(dy.double())
    torch.testing.assert_close(out.detach().cpu().double(), want.detach(), rtol=1e-4, atol=1e-4)
    got, wantg = x.grad.cpu().double(), xr.grad
    bad = (got - wantg).abs() > 1e-4 + 1e-3 * wantg.abs()
    assert bad.double().mean() < 2e-3 and float((got - wantg).norm() / wantg.norm()) < 1e-3
    for (name, p), (_, pr) in zip(conv.named_parameters(), ref.named_parameters()):
        err = float((p.grad.cpu().double() - pr.grad).abs().max()) / (float(pr.grad.abs().max()) + 1e-12)
        assert err < 1e-3, f"{name}: relative max error {err:.2e}"


@pytest.mark.parametrize("kind", ["provided", "downsample"])
def test_pointconv_fused_edge_kernel_other_query_sets(kind):
    """Query points that are not the input points (`out_point_type` "provided" with its own feature width, "downsample" with
    pooled queries): the one-pass edge pipeline against the fp64 op-by-op evaluation on the same neighbour lists."""
    import copy

    from warpconvnet_amd.geometry.coords.search.search_configs import RealSearchConfig
    from warpconvnet_amd.geometry.types.points import Points
    from warpconvnet_amd.nn.functional import point_conv as fpc
    from warpconvnet_amd.nn.modules import PointConv
    from warpconvnet_amd.ops.reductions import row_reduction

    dev = _dev()
    g = torch.Generator().manual_seed(7)
    n, cin, k = 4000, 24, 8
    coords = torch.rand(n, 3, generator=g) * 3.0
    feats = torch.randn(n, cin, generator=g)
    cfg = RealSearchConfig(mode="knn", knn_k=k)
    torch.manual_seed(3)
    if kind == "provided":
        m, cq, cout = 1500, 16, 40
        conv = PointConv(cin, cout, cfg, out_point_type="provided", provided_in_channels=cq)
        qcoords = torch.rand(m, 3, generator=g) * 3.0
        qfeats = torch.randn(m, cq, generator=g)
    else:
        cout = 48
        conv = PointConv(cin, cout, cfg, out_point_type="downsample", pooling_reduction="mean", pooling_voxel_size=0.3)
    ref = copy.deepcopy(conv).double()
    conv = conv.to(dev)
    x = feats.to(dev).requires_grad_(True)
    pc = Points(coords.to(dev), x, offsets=torch.tensor([0, n]))
    calls = []
    orig = fpc._FusedEdge.apply
    fpc._FusedEdge.apply = lambda *a: (calls.append(1), orig(*a))[1]
    try:
        if kind == "provided":
            qx = qfeats.to(dev).requires_grad_(True)
            qpc = Points(qcoords.to(dev), qx, offsets=torch.tensor([0, m]))
            out_pc = conv(pc, qpc)
        else:
            qpc = pc.voxel_downsample(0.3, reduction="mean")
            out_pc = conv(pc)
    finally:
        fpc._FusedEdge.apply = orig
    assert calls, "the fused edge kernel was not used"
    out = out_pc.feature_tensor
    dy = torch.randn(out.shape, generator=g)
    out.backward(dy.to(dev))

    nb = pc.neighbors(query_coords=qpc.batched_coordinates, search_args=cfg)
    idx = nb.neighbor_indices.cpu().view(-1)
    mq = qpc.coordinate_tensor.shape[0]
    assert out.shape[0] == mq
    xr = feats.double().requires_grad_(True)
    if kind == "provided":
        qr = qfeats.double().requires_grad_(True)
        qf = qr
    else:  # pooled query features are a function of the inputs: the same pooling on the CPU in fp64, rows matched by position
        cpu_q = Points(coords.double(), xr, offsets=torch.tensor([0, n])).voxel_downsample(0.3, reduction="mean")
        got_xyz = qpc.coordinate_tensor.detach().cpu().double()
        order = torch.cdist(got_xyz, cpu_q.coordinate_tensor.double()).argmin(1)
        assert torch.allclose(cpu_q.coordinate_tensor.double()[order], got_xyz, atol=1e-4)
        qf = cpu_q.feature_tensor[order]
    e = ref.edge_transform_mlp(torch.cat([xr[idx], qf.repeat_interleave(k, dim=0)], 1))
    want = ref.out_transform_mlp(row_reduction(e, torch.arange(0, mq * k + 1, k), reduction="mean"))
    want.backward(dy.double())
    torch.testing.assert_close(out.detach().cpu().double(), want.detach(), rtol=1e-4, atol=1e-4)
    assert float((x.grad.cpu().double() - xr.grad).norm() / xr.grad.norm()) < 1e-3
    if kind == "provided":
        assert float((qx.grad.cpu().double() - qr.grad).norm() / qr.grad.norm()) < 1e-3
    for (name, p), (_, pr) in zip(conv.named_parameters(), ref.named_parameters()):
        err = float((p.grad.cpu().double() - pr.grad).abs().max()) / (float(pr.grad.abs().max()) + 1e-12)
        assert err < 1e-3, f"{name}: relative max error {err:.2e}"


@pytest.mark.gpu
def test_pointconv_backward_bitwise_repeatable_when_deterministic():
    """The input gradient of the one-kernel PointConv backward is a scatter-add over the neighbour lists: fp32 atomics by
    default (like `index_add` in the reference's autograd).  Under `torch.use_deterministic_algorithms(True)` the kernel
    writes per-edge rows instead and the rows of every input point are added in ascending edge order: two runs are
    bitwise identical, and equal to the atomic path within fp32 summation error."""
    from warpconvnet_amd.geometry.coords.search.search_configs import RealSearchConfig
    from warpconvnet_amd.geometry.types.points import Points
    from warpconvnet_amd.nn.modules import PointConv

    dev = _dev()
    n, cin, cout, k = 6000, 32, 64, 16
    g = torch.Generator().manual_seed(11)
    coords = (torch.rand(n, 3, generator=g) * 6.0).to(dev)
    feats = torch.randn(n, cin, generator=g).to(dev)
    torch.manual_seed(3)
    conv = PointConv(cin, cout, RealSearchConfig(mode="knn", knn_k=k), reductions=("mean",)).to(dev)
    dy = torch.randn(n, cout, generator=g).to(dev)

    def grads(deterministic):
        conv.zero_grad(set_to_none=True)
        x = feats.clone().requires_grad_(True)
        prev = torch.are_deterministic_algorithms_enabled()
        torch.use_deterministic_algorithms(deterministic)
        try:
            out = conv(Points(coords, x, offsets=torch.tensor([0, n]))).feature_tensor
            out.backward(dy)
        finally:
            torch.use_deterministic_algorithms(prev)
        return x.grad.clone(), [p.grad.clone() for p in conv.parameters()]

    a, pa = grads(True)
    b, pb = grads(True)
    assert torch.equal(a, b), "deterministic input gradient differs between two runs"
    for u, v in zip(pa, pb):
        assert torch.equal(u, v)  # parameter gradients: fixed-order partial sums in both modes
    c, _ = grads(False)
    assert (a - c).abs().max() <= 1e-4 * max(1.0, float(c.abs().max()))
