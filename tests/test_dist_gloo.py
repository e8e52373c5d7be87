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


def test_allreduce_is_noop_without_process_group():
    from warpconvnet_amd.dist import allreduce_gradients, rank_and_world, shard_scenes

    assert rank_and_world() == (0, 1)
    assert shard_scenes(5) == [0, 1, 2, 3, 4] and shard_scenes(5, 1, 2) == [1, 3]
    p = torch.nn.Parameter(torch.ones(3))
    p.grad = torch.ones(3)
    assert allreduce_gradients([p]) == 0 and p.grad.tolist() == [1.0, 1.0, 1.0]
