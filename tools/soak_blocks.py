# dev soak: Sequential(SparseConv3d, BatchNorm1d[, ReLU]) as ONE fused node vs the same modules one by one (a forward hook on the
# convolution) on random shapes / strides / dtypes / modes - outputs, gradients, running statistics bit for bit (GPU box):
#     python tools/soak_blocks.py [cases]
import copy, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, torch.nn as nn
from tests.util import scene_u
from warpconvnet_amd.geometry.types.voxels import Voxels
from warpconvnet_amd.nn.modules.sequential import Sequential
from warpconvnet_amd.nn.modules.sparse_conv import SparseConv3d
dev = torch.device("cuda:0")
bad = 0
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
for seed in range(cases):
    rng = np.random.default_rng(500 + seed)
    cin = int(rng.choice([32, 64, 96, 128, 192, 256])); cout = int(rng.choice([32, 64, 96, 128, 192, 256, 320]))
    ks, stride = [(1, 1), (3, 1), (3, 1), (2, 2), (3, 2)][int(rng.integers(0, 5))]
    relu = bool(rng.integers(0, 2)); amp = [torch.bfloat16, torch.float16][int(rng.integers(0, 2))]
    parts = [scene_u(int(rng.integers(500, 20000)), 900 + 3 * seed + b)[:, 1:] for b in range(int(rng.integers(1, 3)))]
    torch.manual_seed(seed)
    fused = Sequential(SparseConv3d(cin, cout, ks, stride, bias=False), nn.BatchNorm1d(cout), nn.ReLU() if relu else nn.Identity()).to(dev)
    chain = copy.deepcopy(fused); chain[0].register_forward_hook(lambda m, i, o: None)
    feats = [torch.randn(len(p), cin) for p in parts]
    res = []
    for net in (fused, chain):
        out = []
        for mode in (True, False):
            net.train(mode)
            x = Voxels([torch.from_numpy(p) for p in parts], feats, device=dev)
            x = x.replace(batched_features=x.feature_tensor.detach().clone().requires_grad_(True))
            with torch.autocast("cuda", dtype=amp):
                y = net(x)
            g = torch.randn(y.feature_tensor.shape, device=dev, generator=torch.Generator(dev).manual_seed(3)).to(y.feature_tensor.dtype)
            net.zero_grad(set_to_none=True)
            y.feature_tensor.backward(g)
            out += [y.feature_tensor.detach().clone(), x.batched_features.batched_tensor.grad.clone(), net[0].weight.grad.clone(),
                    net[1].weight.grad.clone(), net[1].bias.grad.clone(), net[1].running_mean.clone(), net[1].running_var.clone()]
        res.append(out)
    eq = [torch.equal(u, v) for u, v in zip(*res)]
    ok = all(eq)
    bad += not ok
    if not ok:
        print(f"case {seed}: {cin}->{cout} k{ks} s{stride} relu={relu} {amp} scenes {[len(p) for p in parts]} MISMATCH at", [i for i, e in enumerate(eq) if not e])
print(f"soak done: {cases} cases, failures: {bad}")
