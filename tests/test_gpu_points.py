"""GPU: point-cloud front end - exact grid kNN, segment reduce, PointConv - vs brute force / torch / the reference golden."""
import os

import numpy as np
import pytest
import torch

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


def test_pointconv_config5_shape_runs():
    """BASELINE config 5 shape: 200 k fp32 points, PointConv(32 -> 64, knn 16), then voxelise and a depthwise k=3 conv."""
    from warpconvnet_amd.geometry.coords.search.search_configs import RealSearchConfig
    from warpconvnet_amd.geometry.types.points import Points
    from warpconvnet_amd.nn.modules import PointConv, SparseDepthwiseConv3d

    dev = _dev()
    g = torch.Generator().manual_seed(5)
    n = 200_000
    coords = (torch.rand(n, 3, generator=g) * torch.tensor([50.0, 50.0, 4.0])).to(dev)
    pc = Points(coords, torch.randn(n, 32, generator=g).to(dev), offsets=torch.tensor([0, n]))
    torch.manual_seed(0)
    conv = PointConv(32, 64, RealSearchConfig(mode="knn", knn_k=16)).to(dev)
    out = conv(pc)
    assert out.feature_tensor.shape == (n, 64) and torch.isfinite(out.feature_tensor).all()
    vox = out.to_voxels(0.25)
    dw = SparseDepthwiseConv3d(64, 3).to(dev)
    y = dw(vox)
    assert y.feature_tensor.shape == vox.feature_tensor.shape and torch.isfinite(y.feature_tensor).all()
    y.feature_tensor.sum().backward()
    assert conv.edge_transform_mlp.block[0].weight.grad is not None and dw.weight.grad is not None
