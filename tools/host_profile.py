"""Dev tool: where the HOST time of a MinkUNet-14 iteration goes (the network is host-bound below ~1 M voxels).
Prints forward / backward host-enqueue time and cProfile tables of the forward and of the backward (run with the autograd
engine's worker threads off, so its nodes execute under the profiler); the .pstats files land in gpurun_out/.  GPU box only.

    python tools/host_profile.py [voxels] [noprof]
"""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from bench import scene_surface
from bench_models import MinkUNet14
from warpconvnet_amd.geometry.types.voxels import Voxels

dev = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
c = torch.from_numpy(scene_surface(N, seed=3)).to(dev)
n = c.shape[0]
feats = torch.randn(n, 3, device=dev)
torch.manual_seed(0)
net = MinkUNet14(3, 20).to(dev)
off = torch.tensor([0, n], dtype=torch.int32)


def fwd():
    net.zero_grad(set_to_none=True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y = net(Voxels(c, feats, offsets=off))
    return y.feature_tensor.float().square().mean()


for _ in range(3):
    fwd().backward()
torch.cuda.synchronize()
tf = tb = 0.0
for _ in range(10):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    loss = fwd()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    loss.backward()
    t3 = time.perf_counter()
    tf += t1 - t0
    tb += t3 - t2
print(f"{n} voxels: host enqueue forward {tf * 100:.2f} ms, backward {tb * 100:.2f} ms per iteration")
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    fwd().backward()
torch.cuda.synchronize()
print(f"free-running: {(time.perf_counter() - t0) * 50:.3f} ms per iteration")
if len(sys.argv) > 2 and sys.argv[2] == "noprof":  # (for a kernel trace: tools/gap_report.py)
    sys.exit(0)

pr = cProfile.Profile()
pr.enable()
losses = [fwd() for _ in range(5)]
pr.disable()
os.makedirs("gpurun_out", exist_ok=True)
pr.dump_stats("gpurun_out/host_fwd.pstats")
print("---- forward (5 iterations), by tottime")
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
print("---- forward (5 iterations), by cumulative time")
pstats.Stats(pr).sort_stats("cumulative").print_stats(45)

# backward: with the engine's worker threads switched off the backward nodes run on this thread, under this profiler
pb = cProfile.Profile()
with torch.autograd.set_multithreading_enabled(False):
    pb.enable()
    for l in losses:
        l.backward()
    pb.disable()
torch.cuda.synchronize()
pb.dump_stats("gpurun_out/host_bwd.pstats")
print("---- backward (5 iterations, single-threaded engine), by tottime")
pstats.Stats(pb).sort_stats("tottime").print_stats(25)
