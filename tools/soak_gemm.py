# dev soak: the three GEMMs on random channel shapes (incl. outputs wider than 128 channels: column blocks of the channel-split
# kernels), kernel volumes, strides and 16-bit dtypes against the fp64 oracle (GPU box):   python tools/soak_gemm.py [cases]
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import kmap as okmap
from tests.util import scene_u, rel_max_err
from tests.test_gpu_conv import _kmap, _run_all, _oracle, TOL
dev = torch.device("cuda:0")
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
bad = 0
for seed in range(cases):
    rng = np.random.default_rng(300 + seed)
    cin = int(rng.choice([32, 64, 96, 128, 160, 192, 256, 320, 384]))
    cout = int(rng.choice([32, 64, 96, 128, 160, 192, 256, 320, 384, 512]))
    ks, stride = [((3, 3, 3), (1, 1, 1)), ((2, 2, 2), (2, 2, 2)), ((3, 3, 3), (2, 2, 2)), ((3, 1, 3), (1, 1, 1))][int(rng.integers(0, 4))]
    dtype = [torch.bfloat16, torch.float16][int(rng.integers(0, 2))]
    s = np.concatenate([scene_u(int(rng.integers(300, 5000)), 40 + seed, b) for b in range(int(rng.integers(1, 3)))], 0)
    same = stride == (1, 1, 1)
    out = s if same else okmap.stride_coords(s, stride)[0]
    km = _kmap(s, out, ks, stride, same=same)
    r = okmap.kernel_map(s, out, ks, stride)
    K = ks[0] * ks[1] * ks[2]
    g = torch.Generator().manual_seed(seed)
    X = torch.randn(len(s), cin, generator=g).to(dev, dtype)
    W = (torch.randn(K, cin, cout, generator=g) * (1.0 / (cin * 4) ** 0.5)).to(dev, dtype)
    dY = torch.randn(len(out), cout, generator=g).to(dev, dtype)
    try:
        Y, dX, dW = _run_all(km, X, W, dY, "auto", len(s), len(out))
        Yr, dXr, dWr = _oracle(r, X, W, dY, len(out))
        errs = (rel_max_err(Y, Yr), rel_max_err(dX, dXr), rel_max_err(dW, dWr))
        ok = all(e < TOL[dtype] for e in errs)
    except Exception as e:
        ok, errs = False, repr(e)[:200]
    bad += not ok
    if not ok:
        print(f"case {seed}: {cin}->{cout} k{ks} s{stride} {dtype} n={len(s)} FAIL {errs}")
print(f"soak done: {cases} cases, failures: {bad}")
