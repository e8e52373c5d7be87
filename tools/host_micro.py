"""Dev: host-side cost of the building blocks of a layer call (microseconds per call, GPU queue kept short)."""
import sys, time
sys.path.insert(0, ".")
import torch
from bench import scene_surface
from warpconvnet_amd import _lib
from warpconvnet_amd.geometry.types.voxels import Voxels
from warpconvnet_amd.nn.modules.sparse_conv import SparseConv3d
from warpconvnet_amd.nn.modules.sequential import Sequential

dev = torch.device("cuda:0")
L = _lib.lib()

def timeit(fn, n=300):
    for _ in range(20): fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for i in range(n):
        fn()
        if i % 50 == 49: torch.cuda.synchronize()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e6

x = torch.randn(4096, 64, device=dev).bfloat16(); y = torch.empty_like(x)
sc = torch.ones(64, device=dev); sh = torch.zeros(64, device=dev)
st = _lib.stream_handle(dev)
print("ctypes wcn_bn_apply (1 launch)      %.1f us" % timeit(lambda: L.wcn_bn_apply(_lib.ptr(x), 4096, 64, 2, _lib.ptr(sc), _lib.ptr(sh), 0, _lib.ptr(y), st)))
print("torch.empty                          %.1f us" % timeit(lambda: torch.empty((4096, 64), dtype=torch.bfloat16, device=dev)))
print("torch add_ (1 launch)                %.1f us" % timeit(lambda: y.add_(1)))
print("_lib.ptr x 10                        %.1f us" % timeit(lambda: [_lib.ptr(x) for _ in range(10)]))
class Id(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a): return a.view_as(a)
    @staticmethod
    def backward(ctx, g): return g
xa = x.float().requires_grad_(True)
print("autograd Function.apply (identity)   %.1f us" % timeit(lambda: Id.apply(xa)))
c = torch.from_numpy(scene_surface(40000, seed=3)).to(dev); n = c.shape[0]
f = torch.randn(n, 64, device=dev)
v = Voxels(c, f, offsets=torch.tensor([0, n], dtype=torch.int32))
conv = SparseConv3d(64, 64, 3).to(dev)
bn = torch.nn.BatchNorm1d(64).to(dev)
blk = Sequential(conv, bn, torch.nn.ReLU())
with torch.autocast("cuda", dtype=torch.bfloat16):
    o = conv(v)  # builds the map
    print("Voxels.replace(features)             %.1f us" % timeit(lambda: v.replace(batched_features=f)))
    with torch.no_grad():
        print("SparseConv3d fwd, cached map, no_grad %.1f us" % timeit(lambda: conv(v)))
    print("SparseConv3d fwd, cached map, grad    %.1f us" % timeit(lambda: conv(v)))
    print("Sequential(conv, BN, ReLU) fwd, grad  %.1f us" % timeit(lambda: blk(v)))
    def fb():
        out = blk(v)
        out.feature_tensor.backward(out.feature_tensor)
    print("Sequential(conv, BN, ReLU) fwd+bwd    %.1f us" % timeit(fb, 100))

import cProfile, pstats
if len(sys.argv) > 1 and sys.argv[1] == "prof":
    with torch.autocast("cuda", dtype=torch.bfloat16):
        with torch.autograd.set_multithreading_enabled(False):
            for _ in range(20): fb()
            torch.cuda.synchronize()
            pr = cProfile.Profile(); pr.enable()
            for i in range(200):
                fb()
                if i % 50 == 49: torch.cuda.synchronize()
            pr.disable()
    pstats.Stats(pr).sort_stats("tottime").print_stats(45)
