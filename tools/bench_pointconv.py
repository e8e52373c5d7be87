"""Config 5 (PointConv 32->64 kNN 16 on 200 k points -> voxelise -> depthwise k=3), forward + backward: wall / host time per step.
GPU box only.   python tools/bench_pointconv.py [--iters 10]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from warpconvnet_amd.geometry.coords.search.search_configs import RealSearchConfig
from warpconvnet_amd.geometry.types.points import Points
from warpconvnet_amd.nn.modules.point_conv import PointConv
from warpconvnet_amd.nn.modules.sparse_conv_depth import SparseDepthwiseConv3d

ap = argparse.ArgumentParser(); ap.add_argument("--iters", type=int, default=10); args = ap.parse_args()
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(7)
n = 200_000
pts = (torch.rand(n, 3, generator=g) * torch.tensor([50.0, 50.0, 4.0])).to(dev)
pf = torch.randn(n, 32, generator=g).to(dev)
torch.manual_seed(0)
pconv = PointConv(32, 64, RealSearchConfig(mode="knn", knn_k=16)).to(dev)
dw = SparseDepthwiseConv3d(64, 3).to(dev)

def step():
    pconv.zero_grad(set_to_none=True); dw.zero_grad(set_to_none=True)
    o = pconv(Points(pts, pf, offsets=torch.tensor([0, n])))
    y = dw(o.to_voxels(0.25))
    y.feature_tensor.sum().backward()

for _ in range(3): step()
torch.cuda.synchronize()
import gc; gc.collect(); gc.freeze()
t = time.perf_counter()
for _ in range(args.iters): step()
host = (time.perf_counter() - t) / args.iters
torch.cuda.synchronize()
wall = (time.perf_counter() - t) / args.iters
print(f"PointConv + depthwise, {n} points: wall {wall*1e3:.2f} ms/step, host enqueue {host*1e3:.2f} ms/step")
