"""Dev experiment: how much of the weight-gradient kernel's time is memory.  The same launch (1 M-voxel uniform scene, 64 -> 128,
bf16) on (a) the real pair lists, (b) lists folded into the first 8 192 rows (every gather an L2 hit), (c) input row = output
row (both gathers ascending within an offset), (d) output rows folded only, (e) input rows folded only.  GPU box only.

    python tools/exp_wgrad_floor.py
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from bench import scene_u
from warpconvnet_amd import _lib
from warpconvnet_amd.geometry.coords.search.torch_discrete import generate_kernel_map

dev = torch.device("cuda:0")
N, CIN, COUT, K = 1_000_000, 64, 128, 27
coords = torch.from_numpy(scene_u(N, 1)).to(dev)
n = coords.shape[0]
bc = torch.cat([torch.zeros(n, 1, dtype=torch.int32, device=dev), coords], 1).contiguous()
km = generate_kernel_map(bc, bc, (1, 1, 1), (3, 3, 3))
x = torch.randn(n, CIN, device=dev).to(torch.bfloat16)
dy = torch.randn(n, COUT, device=dev).to(torch.bfloat16)
L = _lib.lib()
st = _lib.stream_handle(dev)
ws_bytes = L.wcn_conv_wgrad_workspace(K, CIN, COUT, _lib.WCN_ALGO_MFMA)
ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
dw = torch.empty(K, CIN, COUT, device=dev)
im, om, off = km.in_maps_device, km.out_maps_device, km._offsets_dev


def run(a, b):
    def f():
        _lib.check(L.wcn_conv_wgrad(_lib.ptr(x), _lib.ptr(dy), _lib.ptr(dw), _lib.ptr(a), _lib.ptr(b), _lib.ptr(off), n, n, CIN, COUT,
                                    K, _lib.WCN_BF16, _lib.WCN_ALGO_MFMA, _lib.ptr(ws), ws_bytes, st), "wgrad")
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 20 * 1e3


fold = 8191
print(f"pairs {int(km.offsets[-1])}")
print(f"(a) real lists                      {run(im, om):7.1f} us")
print(f"(b) both folded into 8 192 rows     {run(im & fold, om & fold):7.1f} us")
print(f"(c) input row = output row          {run(om, om):7.1f} us")
print(f"(d) output rows folded              {run(im, om & fold):7.1f} us")
print(f"(e) input rows folded               {run(im & fold, om):7.1f} us")
perm = torch.randperm(n, device=dev, dtype=torch.int64)
print(f"(f) output rows through a random permutation (no order inside an offset) {run(im, perm[om.long()].int()):7.1f} us")
