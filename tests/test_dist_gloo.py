"""CPU, world_size 2 over gloo: scene sharding + flat-bucket gradient all-reduce of the data-parallel path.

The GPU run uses the same code with backend "nccl" (= RCCL); here the convolution arithmetic runs on the
product's explicit_gemm backend with oracle-built kernel maps, so the test checks the N>1 plumbing:
disjoint scene shards, identical parameters on every rank, all-reduced gradients equal to the single-process
gradient over all scenes, one collective per bucket.
"""
import os
import tempfile

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests.util import scene_u

NUM_SCENES = 4


def _scene_batch(scene_ids):
    from tests.test_host_api import _attach_oracle_map
    from warpconvnet_amd.geometry.types.voxels import Voxels

    coords = [torch.from_numpy(scene_u(400 + 50 * i, 100 + i)[:, 1:]) for i in scene_ids]
    feats = [torch.randn(len(c), 8, generator=torch.Generator().manual_seed(200 + i)) for c, i in zip(coords, scene_ids)]
    vox = Voxels(coords, feats)
    _attach_oracle_map(vox)
    return vox


def _loss_and_grads(model, vox):
    model.zero_grad()
    y = model(vox)
    loss = y.feature_tensor.square().sum()
    loss.backward()
    return loss.item()


def _make_model():
    from warpconvnet_amd.nn.modules.sparse_conv import SparseConv3d

    torch.manual_seed(1234)
    return SparseConv3d(8, 16, 3, fwd_algo="explicit_gemm", dgrad_algo="explicit_gemm", wgrad_algo="explicit_gemm")


def _worker(rank, world, init_file, out_dir):
    from warpconvnet_amd.dist import allreduce_gradients, shard_scenes

    dist.init_process_group("gloo", init_method=f"file://{init_file}", rank=rank, world_size=world)
    torch.set_num_threads(1)
    mine = shard_scenes(NUM_SCENES)
    assert mine == list(range(rank, NUM_SCENES, world))
    model = _make_model()
    _loss_and_grads(model, _scene_batch(mine))
    calls = allreduce_gradients(model.parameters(), average=False)
    torch.save({"w": model.weight.grad.clone(), "b": model.bias.grad.clone(), "calls": calls, "mine": mine},
               os.path.join(out_dir, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_data_parallel_gradients_match_single_process():
    with tempfile.TemporaryDirectory() as tmp:
        init_file = os.path.join(tmp, "rendezvous")
        mp.spawn(_worker, args=(2, init_file, tmp), nprocs=2, join=True)
        r0 = torch.load(os.path.join(tmp, "rank0.pt"))
        r1 = torch.load(os.path.join(tmp, "rank1.pt"))
    assert sorted(r0["mine"] + r1["mine"]) == list(range(NUM_SCENES)) and not set(r0["mine"]) & set(r1["mine"])
    assert r0["calls"] == 1 and r1["calls"] == 1  # weight + bias travel in ONE flat bucket
    torch.testing.assert_close(r0["w"], r1["w"], rtol=0, atol=0)
    torch.testing.assert_close(r0["b"], r1["b"], rtol=0, atol=0)
    model = _make_model()
    _loss_and_grads(model, _scene_batch(list(range(NUM_SCENES))))
    torch.testing.assert_close(r0["w"], model.weight.grad, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(r0["b"], model.bias.grad, rtol=1e-4, atol=1e-3)


def _make_net():
    """Three layers so that several buckets and a reverse-order launch sequence exist; `unused` never sees data."""
    from warpconvnet_amd.nn.modules.sparse_conv import SparseConv3d

    torch.manual_seed(4321)
    kw = dict(fwd_algo="explicit_gemm", dgrad_algo="explicit_gemm", wgrad_algo="explicit_gemm")
    # (`unused` first: buckets fill in reverse parameter order, so it ends up in the LAST bucket and the strict launch
    # order does not hold the others back)
    net = torch.nn.ModuleDict({"unused": SparseConv3d(4, 4, 3, **kw), "a": SparseConv3d(8, 16, 3, **kw),
                               "b": SparseConv3d(16, 16, 3, **kw), "c": SparseConv3d(16, 4, 3, **kw)})
    return net


def _net_step(net, vox):
    y = net["c"](net["b"](net["a"](vox)))
    y.feature_tensor.square().sum().backward()


def _bucket_worker(rank, world, init_file, out_dir):
    from warpconvnet_amd.dist import GradientBuckets, shard_scenes

    dist.init_process_group("gloo", init_method=f"file://{init_file}", rank=rank, world_size=world)
    torch.set_num_threads(1)
    net = _make_net()
    params = list(net.parameters())
    buckets = GradientBuckets(params, average=False, bucket_bytes=8 * 1024)  # tiny buckets: several collectives
    launches = []
    real_launch = buckets._launch
    buckets._launch = lambda b: (launches.append(next(i for i, x in enumerate(buckets._buckets) if x is b)), real_launch(b))[1]
    calls = []
    for it in range(2):  # second iteration: views stay attached, hooks fire again
        buckets.zero_grad()
        _net_step(net, _scene_batch(shard_scenes(NUM_SCENES)))
        in_backward = list(launches)
        calls.append(buckets.finish())
    grads = {n: p.grad.clone() for n, p in net.named_parameters()}
    views_ok = all(p.grad.data_ptr() == buckets._buckets[buckets._where[id(p)][0]]["views"][buckets._where[id(p)][1]].data_ptr()
                   for p in params)
    torch.save({"grads": grads, "calls": calls, "launch_order": launches, "in_backward": len(in_backward),
                "nbuckets": len(buckets._buckets), "views_ok": views_ok}, os.path.join(out_dir, f"b{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_gradient_buckets_overlap_hooks_and_unused_parameters():
    """GradientBuckets (the N > 1 path of bench.py's step): gradients are views into persistent flat buckets, buckets are
    all-reduced from autograd hooks in reverse layer order while the backward pass is still running, launches are in
    bucket order on every rank, parameters that took no part contribute zeros; result = single-process gradient."""
    with tempfile.TemporaryDirectory() as tmp:
        init_file = os.path.join(tmp, "rendezvous")
        mp.spawn(_bucket_worker, args=(2, init_file, tmp), nprocs=2, join=True)
        r0, r1 = torch.load(os.path.join(tmp, "b0.pt")), torch.load(os.path.join(tmp, "b1.pt"))
    assert r0["nbuckets"] >= 3 and r0["calls"] == [r0["nbuckets"]] * 2 == r1["calls"]
    assert r0["launch_order"] == r1["launch_order"] == list(range(r0["nbuckets"])) * 2  # strictly in bucket order
    # second iteration: all but the bucket(s) of the unused layer left from INSIDE the backward pass, before finish()
    assert r0["nbuckets"] < r0["in_backward"] < 2 * r0["nbuckets"]
    assert r0["views_ok"] and r1["views_ok"]
    net = _make_net()
    _net_step(net, _scene_batch(list(range(NUM_SCENES))))
    for n, p in net.named_parameters():
        torch.testing.assert_close(r0["grads"][n], r1["grads"][n], rtol=0, atol=0)
        want = p.grad if p.grad is not None else torch.zeros_like(p)
        torch.testing.assert_close(r0["grads"][n], want, rtol=1e-4, atol=1e-3)


def test_allreduce_is_noop_without_process_group():
    from warpconvnet_amd.dist import allreduce_gradients, rank_and_world, shard_scenes

    assert rank_and_world() == (0, 1)
    assert shard_scenes(5) == [0, 1, 2, 3, 4] and shard_scenes(5, 1, 2) == [1, 3]
    p = torch.nn.Parameter(torch.ones(3))
    p.grad = torch.ones(3)
    assert allreduce_gradients([p]) == 0 and p.grad.tolist() == [1.0, 1.0, 1.0]
