"""dev: host time of a MinkUNet-14 iteration attributed to a handful of functions with lightweight timers (no cProfile
inflation): per-iteration totals and call counts.  GPU box.   python tools/host_attrib.py [voxels]"""
import os, sys, time, functools, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import scene_surface
from warpconvnet_amd.models.mink_unet import MinkUNet14
from warpconvnet_amd.geometry.types.voxels import Voxels

acc = collections.defaultdict(lambda: [0, 0.0])
def wrap(mod, name, label=None):
    fn = getattr(mod, name)
    label = label or name
    @functools.wraps(fn)
    def w(*a, **k):
        t = time.perf_counter()
        try:
            return fn(*a, **k)
        finally:
            e = acc[label]; e[0] += 1; e[1] += time.perf_counter() - t
    setattr(mod, name, w)

from warpconvnet_amd.nn.functional.sparse_conv import block as blk, helper
from warpconvnet_amd.geometry.coords.search import torch_discrete as td
from warpconvnet_amd.geometry.coords.ops import stride as st
import warpconvnet_amd.models.mink_unet as mu
wrap(blk, "generate_output_coords_and_kernel_map") if hasattr(blk, "generate_output_coords_and_kernel_map") else None
wrap(helper, "generate_output_coords_and_kernel_map")
wrap(helper, "generate_kernel_map")
wrap(helper, "wrap_conv_output")
wrap(helper, "stride_coords")
wrap(blk, "_bn_forward"); wrap(blk, "_bn_backward")
wrap(mu, "conv_bn_act")
wrap(mu, "cat")
for cls, nm in ((blk._ConvBnAct, "forward"), (blk._ConvBnAct, "backward"), (blk._PointwiseBnAct, "forward"), (blk._PointwiseBnAct, "backward")):
    f = getattr(cls, nm)
    def mk(f, label):
        def w(*a, **k):
            t = time.perf_counter()
            try: return f(*a, **k)
            finally:
                e = acc[label]; e[0] += 1; e[1] += time.perf_counter() - t
        return staticmethod(w)
    setattr(cls, nm, mk(f, f"{cls.__name__}.{nm}"))

dev = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
c = torch.from_numpy(scene_surface(N, seed=3)).to(dev); n = c.shape[0]
feats = torch.randn(n, 3, device=dev)
torch.manual_seed(0)
net = MinkUNet14(3, 20).to(dev)
off = torch.tensor([0, n], dtype=torch.int32)
def it():
    net.zero_grad(set_to_none=True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y = net(Voxels(c, feats, offsets=off))
    y.feature_tensor.float().square().mean().backward()
for _ in range(5): it()
torch.cuda.synchronize(); acc.clear()
import gc; gc.collect(); gc.freeze()
K = 20
t0 = time.perf_counter()
for _ in range(K): it()
host = (time.perf_counter() - t0) / K
torch.cuda.synchronize()
print(f"host per iteration {host*1e3:.2f} ms")
for k, (cnt, t) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    print(f"{k:45s} {cnt / K:6.1f} calls  {t / K * 1e3:7.3f} ms/iter  {t / max(cnt,1) * 1e6:7.1f} us/call")
