"""Fused ConvBlock tail vs the unfused module chain on the bench scene (1 M voxels, bf16).  GPU box only."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn as nn

from bench import scene_u, time_events
from warpconvnet_amd.geometry.types.voxels import Voxels
from warpconvnet_amd.nn.modules import FusedSparseConvBlock
from warpconvnet_amd.nn.modules.sparse_conv import SparseConv3d

dev = torch.device("cuda:0")
for cin, cout in ((64, 128), (64, 64), (128, 128)):
    c = torch.from_numpy(scene_u(1_000_000, seed=1000)).to(dev)
    N = c.shape[0]
    x = Voxels(c, torch.randn(N, cin, device=dev).to(torch.bfloat16), offsets=torch.tensor([0, N], dtype=torch.int32))
    res = x.replace(batched_features=torch.randn(N, cout, device=dev).to(torch.bfloat16))
    conv = SparseConv3d(cin, cout, 3).to(dev).eval()
    bn = nn.BatchNorm1d(cout).to(dev).eval()
    fused = FusedSparseConvBlock(conv, bn).eval()
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        conv(x)  # kernel map into the cache
        def unfused():
            return torch.relu(bn(conv(x).feature_tensor))
        def unfused_res():
            return torch.relu(bn(conv(x).feature_tensor) + res.feature_tensor)
        t_u, t_f = time_events(unfused, 20), time_events(lambda: fused(x), 20)
        t_ur, t_fr = time_events(unfused_res, 20), time_events(lambda: fused(x, res), 20)
        t_c = time_events(lambda: conv(x), 20)
    print(f"{cin}->{cout}: conv only {t_c:.3f} ms | conv+bn+relu unfused {t_u:.3f} fused {t_f:.3f} | +residual unfused {t_ur:.3f} fused {t_fr:.3f}")
