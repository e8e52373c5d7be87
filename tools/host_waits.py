"""Dev tool: where a free-running MinkUNet-14 iteration spends HOST time - inside kernel-map validation (status word waits),
inside the down-sampling passes (output-count reads), inside everything else.  GPU box only.

    python tools/host_waits.py [voxels]
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from bench import scene_surface
from bench_models import MinkUNet14
from warpconvnet_amd.geometry.coords.ops import stride as stride_mod
from warpconvnet_amd.geometry.coords.search.search_results import IntSearchResult
from warpconvnet_amd.geometry.types.voxels import Voxels

dev = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
c = torch.from_numpy(scene_surface(N, seed=3)).to(dev)
n = c.shape[0]
feats = torch.randn(n, 3, device=dev)
torch.manual_seed(0)
net = MinkUNet14(3, 20).to(dev)
off = torch.tensor([0, n], dtype=torch.int32)

acc = {"validate": [0.0, 0, 0], "stride": [0.0, 0, 0]}
_validate = IntSearchResult.validate
_stride = stride_mod._stride_coords_from_cells


def validate(self):
    pending = getattr(self, "_validate_fn", None) is not None
    t = time.perf_counter()
    r = _validate(self)
    if pending:
        acc["validate"][0] += time.perf_counter() - t
        acc["validate"][1] += 1
    return r


def stride_cells(*a, **k):
    t = time.perf_counter()
    r = _stride(*a, **k)
    acc["stride"][0] += time.perf_counter() - t
    acc["stride"][1] += 1
    return r


IntSearchResult.validate = validate
stride_mod._stride_coords_from_cells = stride_cells


def step():
    net.zero_grad(set_to_none=True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y = net(Voxels(c, feats, offsets=off))
    y.feature_tensor.float().square().mean().backward()


for _ in range(5):
    step()
torch.cuda.synchronize()
for v in acc.values():
    v[0], v[1] = 0.0, 0
iters = 30
t0 = time.perf_counter()
for _ in range(iters):
    step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"{n} voxels: {(t2 - t0) / iters * 1e3:.2f} ms per iteration (host loop {(t1 - t0) / iters * 1e3:.2f})")
for k, (t, cnt, _) in acc.items():
    print(f"  {k:9s}: {cnt / iters:5.1f} calls, {t / iters * 1e3:6.3f} ms per iteration ({t / max(cnt, 1) * 1e6:6.1f} us each)")
