"""CPU, world_size 2 over gloo: scene sharding + flat-bucket gradient all-reduce of the data-parallel path.

The GPU run uses the same code with backend "nccl" (= RCCL); here the convolution arithmetic runs on the
product's explicit_gemm backend with oracle-built kernel maps, so the test checks the N>1 plumbing:
disjoint scene shards, identical parameters on every rank, all-reduced gradients equal to the single-process
gradient over all scenes, one collective per bucket.
"""
import os
import tempfile

import numpy as np
import pytest
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


def _bench_worker(rank, world, init_file, out_dir):
    """bench.py's own N > 1 code path (`build_workload` -> `make_step` -> `timed_loop`), with the process group on gloo, the
    tensors on the CPU and the convolution on the explicit backend with an oracle-built map: the functions the driver's
    `torch.distributed.run ... bench.py --gpus N` executes, minus the device."""
    import bench
    from tests.test_host_api import _attach_oracle_map

    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    dist.init_process_group("gloo", init_method=f"file://{init_file}", rank=rank, world_size=world)
    torch.set_num_threads(1)
    args = bench.parse_args(["--gpus", str(world), "--steps", "3", "--warmup", "1", "--voxels", "600"])
    dev = torch.device("cpu")
    kw = dict(fwd_algo="explicit_gemm", dgrad_algo="explicit_gemm", wgrad_algo="explicit_gemm")
    coords, feats, grad_out, offsets, conv, params = bench.build_workload(args, dev, rank, conv_kwargs=kw)
    step, buckets = bench.make_step(dev, world, coords, feats, grad_out, offsets, conv, params, attach=_attach_oracle_map)
    assert buckets is not None and buckets.world == world
    elapsed, ms_half = bench.timed_loop(step, args, dev, world)
    diag = bench.multi_gpu_diag(dev, rank, world, params, iters=2)  # (what the N > 1 line carries as `multi_gpu`)
    torch.save({"w": conv.weight.grad.clone(), "b": conv.bias.grad.clone(), "elapsed": elapsed, "ms_half": ms_half,
                "n": coords.shape[0], "coords": coords.clone(), "diag": diag}, os.path.join(out_dir, f"bench{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_bench_step_wiring_under_gloo():
    """`bench.py --gpus 2` end to end on CPU: different scenes per rank (weak scaling), the step's gradient buckets reduce to
    the same averaged gradient on both ranks = the mean of the two single-process gradients, the timed loop returns the
    maximum over ranks."""
    import bench
    from tests.test_host_api import _attach_oracle_map

    with tempfile.TemporaryDirectory() as tmp:
        init_file = os.path.join(tmp, "rendezvous")
        mp.spawn(_bench_worker, args=(2, init_file, tmp), nprocs=2, join=True)
        r0, r1 = torch.load(os.path.join(tmp, "bench0.pt")), torch.load(os.path.join(tmp, "bench1.pt"))
    assert not torch.equal(r0["coords"], r1["coords"])  # one scene per rank
    assert r0["elapsed"] == r1["elapsed"] > 0 and r0["ms_half"] > 0  # max over ranks, agreed
    for r in (r0, r1):  # the self-diagnosis of the first multi-GPU run: every rank seen, one all-reduce of the gradient's size timed
        assert r["diag"]["ranks_seen"] == [0, 1] and r["diag"]["backend"] == "gloo"
        assert r["diag"]["grad_allreduce_bytes"] == 4 * (27 * 64 * 128 + 128) and r["diag"]["grad_allreduce_ms"] > 0
    torch.testing.assert_close(r0["w"], r1["w"], rtol=0, atol=0)
    torch.testing.assert_close(r0["b"], r1["b"], rtol=0, atol=0)
    # single process: the same two workloads, gradients averaged by hand
    kw = dict(fwd_algo="explicit_gemm", dgrad_algo="explicit_gemm", wgrad_algo="explicit_gemm")
    args = bench.parse_args(["--gpus", "1", "--steps", "1", "--warmup", "0", "--voxels", "600"])
    gw, gb = [], []
    for rank in range(2):
        coords, feats, grad_out, offsets, conv, params = bench.build_workload(args, torch.device("cpu"), rank, conv_kwargs=kw)
        step, buckets = bench.make_step(torch.device("cpu"), 1, coords, feats, grad_out, offsets, conv, params, attach=_attach_oracle_map)
        assert buckets is None
        step()
        gw.append(conv.weight.grad.clone())
        gb.append(conv.bias.grad.clone())
    torch.testing.assert_close(r0["w"], (gw[0] + gw[1]) / 2, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(r0["b"], (gb[0] + gb[1]) / 2, rtol=1e-4, atol=1e-3)


def test_gradient_buckets_refuse_unsynchronised_accumulation():
    """Two backward() passes before finish() used to leave the second pass's gradients out of the (already launched)
    all-reduce silently; now the second pass raises, and `no_sync()` is the way to accumulate."""
    from warpconvnet_amd.dist import GradientBuckets

    lin = torch.nn.Linear(4, 3)
    buckets = GradientBuckets(lin.parameters(), average=False)
    x = torch.randn(5, 4)
    lin(x).sum().backward()
    with pytest.raises(RuntimeError, match="second gradient before finish"):
        lin(x).sum().backward()
    buckets.finish()
    buckets.zero_grad()
    with buckets.no_sync():
        lin(x).sum().backward()
    lin(x).sum().backward()
    buckets.finish()
    ref = torch.nn.Linear(4, 3)
    ref.load_state_dict(lin.state_dict())
    (ref(x).sum() * 2).backward()
    torch.testing.assert_close(lin.weight.grad, ref.weight.grad)
    buckets.remove()
    assert buckets._handles == []
