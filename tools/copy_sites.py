"""Dev tool: which Python call sites issue memory copies (clone / cpu / to / copy_ / torch.tensor on a device) in a MinkUNet-14
iteration - the `__amd_rocclr_copyBuffer` rows of a kernel trace.  GPU box only.   python tools/copy_sites.py [voxels]"""
import collections
import os
import sys
import traceback

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
sites = collections.Counter()
on = [False]


def site():
    for fr in reversed(traceback.extract_stack()[:-2]):
        if "warpconvnet_amd" in fr.filename or "bench" in fr.filename:
            return f"{os.path.relpath(fr.filename)}:{fr.lineno} {fr.line.strip()[:90]}"
    return "?"


def wrap(name):
    orig = getattr(torch.Tensor, name)

    def w(self, *a, **k):
        r = orig(self, *a, **k)
        if on[0] and isinstance(r, torch.Tensor) and (self.is_cuda or r.is_cuda) and r is not self and r.data_ptr() != self.data_ptr():
            sites[f"{name:6s} {tuple(self.shape)} {self.dtype} -> {r.device.type}  @ {site()}"] += 1
        return r
    setattr(torch.Tensor, name, w)


for m in ("clone", "cpu", "to", "contiguous", "int", "float", "long"):
    wrap(m)
_copy = torch.Tensor.copy_


def copy_(self, src, *a, **k):
    if on[0] and (self.is_cuda or src.is_cuda):
        sites[f"copy_  {tuple(src.shape)} {src.dtype}/{src.device.type} -> {self.dtype}/{self.device.type}  @ {site()}"] += 1
    return _copy(self, src, *a, **k)


torch.Tensor.copy_ = copy_
_tensor = torch.tensor


def tensor(*a, **k):
    r = _tensor(*a, **k)
    if on[0] and r.is_cuda:
        sites[f"tensor {tuple(r.shape)} -> cuda  @ {site()}"] += 1
    return r


torch.tensor = tensor


def step():
    net.zero_grad(set_to_none=True)
    x = Voxels(c, feats, offsets=_tensor([0, n], dtype=torch.int32))
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y = net(x)
    y.feature_tensor.float().square().mean().backward()


for _ in range(3):
    step()
on[0] = True
iters = 4
for _ in range(iters):
    step()
torch.cuda.synchronize()
for k, v in sites.most_common(40):
    print(f"{v / iters:5.1f}  {k}")
