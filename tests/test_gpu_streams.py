"""GPU: the C-ABI on NON-default streams (SURVEY §8b "Threading / streams").

Every entry point takes the caller's stream; per-device one-time state (`once_per_device` function attributes), the pinned
READY word of `wcn_kmap_tally_sort` and the packed-weight cache must not assume the default stream.  Mirrors the
reference's `tests/coords/test_packed_hashmap.py:431-462` (concurrent searches on one table from several streams) and adds
the whole hot path - map build, forward, backward - on a side stream, bit-equal to the default-stream run."""
import numpy as np
import pytest
import torch

from oracle import kmap as okmap
from tests.util import scene_u

pytestmark = pytest.mark.gpu


def _dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def test_concurrent_searches_across_streams():
    from warpconvnet_amd.geometry.coords.search.packed_hashmap import PackedHashTable

    dev = _dev()
    coords = torch.from_numpy(scene_u(20000, 10)).to(dev)
    ht = PackedHashTable.from_coords(coords)
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream() for _ in range(4)]
    chunks = torch.chunk(coords, 4)
    results = [None] * 4
    for _ in range(3):  # interleaved issue, several rounds: the searches of different streams overlap on the device
        for i, (stream, q) in enumerate(zip(streams, chunks)):
            with torch.cuda.stream(stream):
                results[i] = ht.search(q)
    for s in streams:
        s.synchronize()
    torch.cuda.synchronize()
    for i, (q, r) in enumerate(zip(chunks, results)):
        assert (r >= 0).all(), f"stream {i}: missed keys"
        assert torch.equal(coords[r.long()], q), f"stream {i}: mismatch"
    ref = ht.search(coords)
    torch.cuda.synchronize()
    assert torch.equal(torch.cat(results, 0), ref)
    assert torch.equal(ref.cpu(), torch.arange(len(coords), dtype=ref.dtype))


def _step(conv, coords_np, feats, stream=None):
    """map build + forward + backward of one SparseConv3d on `stream` (None: the default stream)."""
    from warpconvnet_amd.geometry.types.voxels import Voxels

    dev = _dev()
    ctx = torch.cuda.stream(stream) if stream is not None else torch.cuda.stream(torch.cuda.default_stream(dev))
    with ctx:
        vox = Voxels([torch.from_numpy(coords_np[:, 1:].copy())], [feats.clone()], device=dev)
        x = vox.replace(batched_features=vox.feature_tensor.detach().clone().requires_grad_(True))
        conv.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y = conv(x)
        y.feature_tensor.float().square().mean().backward()
        km = next(iter(x.cache.values()))
        out = (y.feature_tensor.detach().clone(), x.feature_tensor.grad.clone(), conv.weight.grad.clone(),
               km.in_maps.clone(), km.out_maps.clone(), km.offsets.clone())
    (stream or torch.cuda.current_stream(dev)).synchronize()
    return out


def test_hot_path_on_a_side_stream_equals_the_default_stream():
    from warpconvnet_amd.nn.modules.sparse_conv import SparseConv3d

    dev = _dev()
    c = scene_u(30000, 4)
    feats = torch.randn(len(c), 64)
    torch.manual_seed(0)
    conv = SparseConv3d(64, 128, 3).to(dev)
    torch.cuda.synchronize()
    # FIRST use of the library's kernels in this test is on the side stream (function attributes, packed weights)
    side = torch.cuda.Stream()
    got_side = _step(conv, c, feats, side)
    got_main = _step(conv, c, feats, None)
    side2 = torch.cuda.Stream()
    got_side2 = _step(conv, c, feats, side2)
    r = okmap.kernel_map(c, c, (3, 3, 3))
    for got in (got_side, got_main, got_side2):
        np.testing.assert_array_equal(got[3].cpu().numpy(), r["in_maps"])
        np.testing.assert_array_equal(got[4].cpu().numpy(), r["out_maps"])
        np.testing.assert_array_equal(got[5].cpu().numpy(), r["offsets"])
    for a, b in zip(got_side[:3], got_main[:3]):
        assert torch.equal(a, b)  # same kernels, same order of accumulation: bit-equal
    for a, b in zip(got_side2[:3], got_main[:3]):
        assert torch.equal(a, b)


def test_two_streams_build_and_convolve_different_scenes_concurrently():
    """Two scenes in flight at once on two streams (each with its own pinned status mirror and workspaces): both equal
    their single-stream results."""
    from warpconvnet_amd.nn.modules.sparse_conv import SparseConv3d

    dev = _dev()
    scenes = [scene_u(25000, 21), scene_u(41000, 22)]
    feats = [torch.randn(len(s), 32) for s in scenes]
    torch.manual_seed(1)
    convs = [SparseConv3d(32, 64, 3).to(dev) for _ in scenes]
    want = [_step(cv, s, f, None) for cv, s, f in zip(convs, scenes, feats)]
    from warpconvnet_amd.geometry.types.voxels import Voxels

    streams = [torch.cuda.Stream() for _ in scenes]
    held = []
    for rounds in range(2):
        held.clear()
        for cv, s, f, st in zip(convs, scenes, feats, streams):
            with torch.cuda.stream(st):  # queue everything of scene i, then move on without waiting
                vox = Voxels([torch.from_numpy(s[:, 1:].copy())], [f.clone()], device=dev)
                x = vox.replace(batched_features=vox.feature_tensor.detach().clone().requires_grad_(True))
                cv.zero_grad(set_to_none=True)
                with torch.autocast("cuda", dtype=torch.bfloat16):
                    y = cv(x)
                y.feature_tensor.float().square().mean().backward()
                held.append((x, y))
        for st in streams:
            st.synchronize()
    for (x, y), cv, w in zip(held, convs, want):
        km = next(iter(x.cache.values()))
        assert torch.equal(km.in_maps.cpu(), w[3].cpu()) and torch.equal(km.out_maps.cpu(), w[4].cpu())
        assert torch.equal(y.feature_tensor.detach(), w[0])
        assert torch.equal(x.feature_tensor.grad, w[1]) and torch.equal(cv.weight.grad, w[2])
