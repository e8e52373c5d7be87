# dev soak: PointConv with the one-kernel edge pipeline (csrc/pointconv.hip) vs the same module composed from separate ops
# (WARPCONVNET_AMD_POINTCONV_FUSED = 0 semantics) on random shapes, neighbour counts, reductions, kNN / radius lists, with and
# without relative positions - outputs, input gradient, every parameter gradient (GPU box):   python tools/soak_pointconv.py [cases]
import copy, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from warpconvnet_amd.geometry.coords.search.search_configs import RealSearchConfig
from warpconvnet_amd.geometry.types.points import Points
from warpconvnet_amd.nn.functional import point_conv as fpc
from warpconvnet_amd.nn.modules import PointConv
dev = torch.device("cuda:0")
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
bad = used = 0
for seed in range(cases):
    rng = np.random.default_rng(4000 + seed)
    n = int(rng.integers(200, 12000)); cin = int(rng.choice([4, 8, 16, 24, 32])); cout = int(rng.choice([8, 16, 32, 48, 64]))
    red = ["mean", "sum"][int(rng.integers(0, 2))]; rel = bool(rng.integers(0, 2))
    if rng.integers(0, 3) == 0:
        cfg = RealSearchConfig(mode="radius", radius=float(rng.uniform(0.15, 0.4)))
    else:
        cfg = RealSearchConfig(mode="knn", knn_k=int(rng.choice([1, 2, 4, 8, 16, 32])))
    g = torch.Generator().manual_seed(seed)
    coords = torch.rand(n, 3, generator=g) * 4.0
    feats = torch.randn(n, cin, generator=g)
    torch.manual_seed(seed)
    conv = PointConv(cin, cout, cfg, reductions=(red,), use_rel_pos=rel).to(dev)
    offs = torch.tensor([0, n // 3, n])
    res = []
    try:
        for enabled in (True, False):
            fpc._ENABLED = enabled
            net = copy.deepcopy(conv)
            x = feats.to(dev).requires_grad_(True)
            calls = []
            orig = fpc._FusedEdge.apply
            fpc._FusedEdge.apply = lambda *a: (calls.append(1), orig(*a))[1]
            try:
                out = net(Points(coords.to(dev), x, offsets=offs)).feature_tensor
            finally:
                fpc._FusedEdge.apply = orig
            gy = torch.randn(out.shape, generator=torch.Generator().manual_seed(seed + 1)).to(dev)
            out.backward(gy)
            res.append([out.detach(), x.grad] + [p.grad for p in net.parameters()])
            if enabled: used += bool(calls)
        errs = [float((a.double() - b.double()).abs().max() / (b.double().abs().max() + 1e-9)) for a, b in zip(*res)]
        # (a ReLU pre-activation within fp32 rounding of zero takes the other branch in the other op order: gradients by norm)
        nerr = [float((a.double() - b.double()).norm() / (b.double().norm() + 1e-12)) for a, b in zip(*res)]
        ok = errs[0] < 2e-4 and all(e < 4e-3 for e in nerr)
    except Exception as e:
        ok, errs, nerr = False, repr(e)[:300], None
    finally:
        fpc._ENABLED = True
    bad += not ok
    if not ok:
        print(f"case {seed}: n={n} {cin}->{cout} {cfg.mode} k={getattr(cfg, 'knn_k', None)} {red} rel={rel} FAIL {errs} {nerr}")
print(f"soak done: {cases} cases ({used} through the fused kernel), failures: {bad}")
