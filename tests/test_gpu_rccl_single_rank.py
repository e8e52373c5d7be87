"""GPU: the data-parallel gradient path on the REAL backend ("nccl" = RCCL on ROCm) with a process group of one rank.

The 8-GPU tier has never been available to this repository's test boxes, so `GradientBuckets` - persistent flat buckets, the
all-reduce launched asynchronously from the autograd hook of a bucket's last gradient, weight gradients written by the HIP
kernels straight into their bucket slot, `finish()` waiting on the work handles - had only ever run on gloo / CPU
(tests/test_dist_gloo.py).  A one-rank RCCL group cannot test the arithmetic of a sum over ranks (gloo does), but it runs every
backend-specific piece: communicator creation, RCCL's own stream and its event hand-off with the compute stream, asynchronous
work objects, in-place collectives on views of a flat buffer the convolution kernels are still writing other parts of."""
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist

from tests.util import scene_u

pytestmark = pytest.mark.gpu


@pytest.fixture
def rccl_group():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=dev)
    try:
        yield dev
    finally:
        dist.destroy_process_group()


def test_gradient_buckets_on_rccl_with_one_rank(rccl_group):
    from warpconvnet_amd.dist import GradientBuckets
    from warpconvnet_amd.geometry.types.voxels import Voxels
    from warpconvnet_amd.nn.modules.sparse_conv import SparseConv3d

    dev = rccl_group
    c = torch.from_numpy(scene_u(30_000, 7)[:, 1:]).to(dev)
    n = c.shape[0]
    g = torch.Generator().manual_seed(3)
    feats = torch.randn(n, 64, generator=g).to(dev, torch.bfloat16).requires_grad_(True)
    grad_out = torch.randn(n, 128, generator=g).to(dev, torch.bfloat16)
    offsets = torch.tensor([0, n], dtype=torch.int32)
    torch.manual_seed(0)
    conv_a = SparseConv3d(64, 128, 3, bias=True).to(dev)
    conv_b = SparseConv3d(128, 128, 3, bias=True).to(dev)
    params = list(conv_a.parameters()) + list(conv_b.parameters())

    def backward():
        feats.grad = None
        x = Voxels(c, feats, offsets=offsets)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y = conv_b(conv_a(x))
        y.batched_features.batched_tensor.backward(grad_out)

    for p in params:
        p.grad = None
    backward()
    want = [p.grad.detach().clone() for p in params]
    want_dx = feats.grad.detach().clone()

    # two small buckets, so that the first collective is launched from a hook while conv_a's backward is still to run
    buckets = GradientBuckets(params, bucket_bytes=1 << 20, collective_when_alone=True)
    assert len(buckets._buckets) >= 2
    for it in range(3):
        buckets.zero_grad()
        backward()
        calls = buckets.finish()
        assert calls == len(buckets._buckets)
        torch.cuda.synchronize()
        for p, w in zip(params, want):
            assert p.grad is not None and p.grad.data_ptr() == p._wcn_grad_slot.data_ptr()  # the gradient lives in its bucket slot
            assert torch.equal(p.grad, w), it  # sum over one rank, averaged by one: bit-identical to the plain backward
        assert torch.equal(feats.grad, want_dx)
    # the world-size reduction the benchmark's timed loop ends with (bench.timed_loop), on the same communicator
    t = torch.tensor([1.25], device=dev, dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.barrier()
    assert float(t.item()) == 1.25
    buckets.remove()
