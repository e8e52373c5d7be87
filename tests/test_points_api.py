"""Drop-in surface of ``Points`` (reference `warpconvnet/geometry/types/points.py:92-326`), mirroring the reference's own
`tests/types/test_points.py:24-222`.  The CPU half runs everywhere; ``sort`` needs the device (as in the reference)."""
import dataclasses

import pytest
import torch

from warpconvnet_amd.geometry.coords.real import RealCoords
from warpconvnet_amd.geometry.coords.sample import random_sample_per_batch
from warpconvnet_amd.geometry.features.cat import CatFeatures
from warpconvnet_amd.geometry.types.points import Points


def _points(device=None, seed=0):
    g = torch.Generator().manual_seed(seed)
    Ns = torch.randint(100, 1000, (3,), generator=g)
    coords = [torch.rand((int(N), 3), generator=g) for N in Ns]
    feats = [torch.rand((int(N), 7), generator=g) for N in Ns]
    return Points(coords, feats, device=device), Ns


def test_point_indexing_and_construction():  # reference test_points.py:24-58
    points, Ns = _points()
    cum = Ns.cumsum(0).tolist()
    for i in range(len(Ns)):
        assert points[i].batched_coordinates.batch_size == 1
        assert points[i].batched_coordinates.batched_tensor.shape[0] == Ns[i]
        assert points[i].batched_features.batched_tensor.shape[0] == Ns[i]
    assert points.batched_coordinates.batch_size == len(Ns)
    assert points.batched_coordinates.offsets.tolist() == [0] + cum
    assert points.batched_coordinates.batched_tensor.shape == (cum[-1], 3)
    assert points.batched_features.batched_tensor.shape == (cum[-1], 7)
    cat = Points(torch.rand(cum[-1], 3), torch.rand(cum[-1], 7), offsets=torch.IntTensor([0] + cum))
    assert cat.batched_coordinates.batch_size == len(Ns)


def test_from_list_of_coordinates_sinusoidal_features():  # reference test_points.py:151-158
    B, N, D = 4, 500, 3
    coords = [torch.rand(N, D) for _ in range(B)]
    p = Points.from_list_of_coordinates(coords, encoding_channels=10, encoding_range=1)
    assert p.batched_coordinates.batched_tensor.shape == (B * N, D)
    assert p.batched_features.batched_tensor.shape == (B * N, 10 * D)
    # the features are cos | sin of 2*pi*2^i * x per axis (reference nn/functional/encodings.py:31-75)
    x = coords[0][:, :1]
    f = 2 * torch.pi * 2.0 ** torch.arange(5)
    want = torch.cat([(x * f).cos(), (x * f).sin()], 1)
    assert torch.allclose(p.feature_tensor[:N, :10], want, atol=1e-5)
    # a [B, N, D] tensor is split along its first axis; explicit features are kept as they are
    q = Points.from_list_of_coordinates(torch.stack(coords), features=[torch.ones(N, 2) for _ in range(B)])
    assert q.batch_size == B and q.feature_tensor.shape == (B * N, 2)
    with pytest.raises(AssertionError):
        Points.from_list_of_coordinates(coords, encoding_channels=10)


def test_random_downsample_draws_inside_every_batch_element():  # reference points.py:189-208, coords/sample.py:11-31
    points, Ns = _points()
    torch.manual_seed(0)
    idx, off = random_sample_per_batch(points.offsets, 64)
    assert off.tolist() == [0, 64, 128, 192] and idx.shape == (192,)
    for b in range(3):
        sel = idx[64 * b : 64 * (b + 1)]
        assert int(sel.min()) >= int(points.offsets[b]) and int(sel.max()) < int(points.offsets[b + 1])
    tagged = Points(points.batched_coordinates, points.batched_features, voxel_size=0.25)
    down = tagged.random_downsample(64)
    assert down.batch_size == 3 and len(down) == 192 and down.offsets.tolist() == [0, 64, 128, 192]
    assert down.voxel_size == 0.25  # extra attributes travel
    # every sampled row is a row of the same batch element of the source
    for b in range(3):
        src = points[b].coordinate_tensor
        got = down[b].coordinate_tensor
        assert bool(((got[:, None, :] == src[None, :, :]).all(-1)).any(1).all())


def test_contiguous():  # reference test_points.py:161-206
    points, _ = _points()
    assert points.contiguous() is points
    wide_c = torch.rand(len(points), 6)
    wide_f = torch.rand(len(points), 14)
    nc = Points(RealCoords(wide_c[:, ::2], points.offsets.clone()), CatFeatures(wide_f[:, ::2], points.offsets.clone()),
                tag="kept")
    assert not nc.coordinate_tensor.is_contiguous() and not nc.feature_tensor.is_contiguous()
    c = nc.contiguous()
    assert c is not nc and c.coordinate_tensor.is_contiguous() and c.feature_tensor.is_contiguous()
    assert torch.equal(c.coordinate_tensor, nc.coordinate_tensor) and torch.equal(c.feature_tensor, nc.feature_tensor)
    assert c.offsets.tolist() == nc.offsets.tolist() and c.extra_attributes["tag"] == "kept"


def test_binary_operations_and_extra_attributes():  # reference test_points.py:117-148, 209-222
    points, _ = _points()
    assert torch.allclose((points + 1).feature_tensor, points.feature_tensor + 1)
    assert torch.allclose((points * 2).feature_tensor, points.feature_tensor * 2)
    sq = points**2
    assert torch.allclose(sq.feature_tensor, points.feature_tensor**2)
    assert (points + sq).feature_tensor.shape == points.feature_tensor.shape
    assert (points * sq).coordinate_tensor.shape == points.coordinate_tensor.shape
    tagged = Points([torch.rand(5, 3)], [torch.rand(5, 2)], test_attribute="test")
    assert tagged.replace(batched_features=tagged.batched_features + 1).extra_attributes["test_attribute"] == "test"
    assert tagged.ordering is None


def test_sort_is_device_only_like_the_reference():
    points, _ = _points()
    with pytest.raises(AssertionError):
        points.sort(0.1)


@pytest.mark.gpu
def test_point_dataclass_serialization_gpu():  # reference test_points.py:61-74
    points, _ = _points("cuda:0")
    down = points.voxel_downsample(0.1)
    d = dataclasses.asdict(down)
    assert d["_extra_attributes"]["voxel_size"] == 0.1
    assert "voxel_size" in dataclasses.replace(down).extra_attributes


@pytest.mark.gpu
@pytest.mark.parametrize("ordering", ["morton_xyz", "morton_zyx"])
def test_sort_orders_every_batch_element_along_the_curve(ordering):
    """``sort`` permutes rows inside each batch element only, and the permuted rows' Morton codes ascend."""
    from warpconvnet_amd.geometry.coords.ops.serialization import encode

    points, _ = _points("cuda:0", seed=3)
    tagged = Points(points.batched_coordinates, points.batched_features, tag=7)
    s = tagged.sort(0.05, ordering)
    assert s.offsets.tolist() == points.offsets.tolist() and s.extra_attributes["tag"] == 7
    for b in range(points.batch_size):
        src = torch.cat([points[b].coordinate_tensor, points[b].feature_tensor], 1).cpu()
        got = torch.cat([s[b].coordinate_tensor, s[b].feature_tensor], 1).cpu()
        key = lambda t: t[torch.argsort(t[:, 0] * 1e3 + t[:, 3], stable=True)]  # noqa: E731
        assert torch.equal(torch.sort(src.flatten()).values, torch.sort(got.flatten()).values)  # same multiset of rows
        assert torch.equal(key(src), key(got))
    # codes of the sorted rows ascend inside every batch element (same quantisation the method used)
    q = torch.floor(s.coordinate_tensor / 0.05).int()
    origin = torch.floor(points.coordinate_tensor / 0.05).int().min(0).values
    codes = encode(torch.cat([q, origin[None]], 0), order=ordering)[:-1].cpu()
    for b in range(points.batch_size):
        seg = codes[int(s.offsets[b]) : int(s.offsets[b + 1])]
        assert bool((seg[1:] >= seg[:-1]).all())


def test_geometry_types_are_dataclasses_like_the_reference():  # reference geometry.py:38-63, test_points.py:61-74
    p = Points([torch.rand(10, 3)], [torch.rand(10, 2)], voxel_size=0.1)
    d = dataclasses.asdict(p)
    assert set(d) == {"batched_coordinates", "batched_features", "_extra_attributes"}
    assert d["_extra_attributes"]["voxel_size"] == 0.1
    r = dataclasses.replace(p)
    assert type(r) is Points and r.extra_attributes["voxel_size"] == 0.1 and r is not p
    assert hash(p) != hash(r) and p != r  # identity semantics are kept
