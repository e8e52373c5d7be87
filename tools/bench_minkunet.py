"""MinkUNet-14 (BASELINE config 3 shape) forward + backward timing on a surface-like scene: wall time and host enqueue
time per iteration (the network is host-bound when they coincide).  GPU box only.

    python tools/bench_minkunet.py [--voxels 200000] [--iters 10]
    WARPCONVNET_AMD_HIP_BATCHNORM=0 python tools/bench_minkunet.py      # stock BatchNorm kernels for comparison
"""
import argparse
import gc
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from bench import scene_surface
from bench_models import MinkUNet14
from warpconvnet_amd.geometry.types.voxels import Voxels


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--voxels", type=int, default=200_000)
    ap.add_argument("--iters", type=int, default=10)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    c = torch.from_numpy(scene_surface(args.voxels, seed=3)).to(dev)
    n = c.shape[0]
    feats = torch.randn(n, 3, device=dev)
    torch.manual_seed(0)
    net = MinkUNet14(3, 20).to(dev)

    def step():
        net.zero_grad(set_to_none=True)
        x = Voxels(c, feats, offsets=torch.tensor([0, n], dtype=torch.int32))
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y = net(x)
        y.feature_tensor.float().square().mean().backward()

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    gc.collect()
    gc.freeze()  # (a full collection over the import-time heap is a 100+ ms host stall if it lands in the window: bench.py)
    t = time.perf_counter()
    for _ in range(args.iters):
        step()
    host = (time.perf_counter() - t) / args.iters
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t) / args.iters
    print(f"MinkUNet-14, {n} voxels, bf16 autocast, fwd+bwd: wall {wall * 1e3:.2f} ms/iter, host enqueue {host * 1e3:.2f} ms/iter")


if __name__ == "__main__":
    main()
