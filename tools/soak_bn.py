# dev soak: hip_batch_norm (+ fused ReLU) on random [N, C] shapes / dtypes / modes vs torch's batch_norm in fp64 on the same
# values - output, input / weight / bias gradients, running statistics (GPU box):   python tools/soak_bn.py [cases]
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, torch.nn.functional as F
from warpconvnet_amd.nn.functional.normalizations import hip_batch_norm
dev = torch.device("cuda:0")
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
bad = 0
for seed in range(cases):
    rng = np.random.default_rng(seed)
    n = int(rng.choice([2, 3, 17, 127, 128, 129, 1000, 4097, 60000, int(rng.integers(2, 300000))]))
    c = int(rng.choice([1, 3, 7, 8, 16, 20, 32, 33, 64, 96, 100, 128, 256, 384]))
    dt = [torch.float32, torch.bfloat16, torch.float16][int(rng.integers(0, 3))]
    training, relu, affine = bool(rng.integers(0, 4)), bool(rng.integers(0, 2)), bool(rng.integers(0, 4))
    g = torch.Generator().manual_seed(seed)
    x = (torch.randn(n, c, generator=g) * float(rng.uniform(0.1, 5)) + float(rng.uniform(-3, 3))).to(dev, dt).requires_grad_(True)
    w = (torch.rand(c, generator=g) + 0.5).to(dev).requires_grad_(True) if affine else None
    b = (torch.randn(c, generator=g) * 0.3).to(dev).requires_grad_(True) if affine else None
    rm, rv = (torch.randn(c, generator=g) * 0.1).to(dev), (torch.rand(c, generator=g) + 0.5).to(dev)
    rm2, rv2 = rm.double().cpu().clone(), rv.double().cpu().clone()
    gy = torch.randn(n, c, generator=g).to(dev, dt)
    try:
        y = hip_batch_norm(x, rm, rv, w, b, training, 0.1, 1e-5, relu)
        y.backward(gy)
        xr = x.detach().double().cpu().requires_grad_(True)
        wr = w.detach().double().cpu().requires_grad_(True) if affine else None
        br = b.detach().double().cpu().requires_grad_(True) if affine else None
        yr = F.batch_norm(xr, rm2, rv2, wr, br, training, 0.1, 1e-5)
        if relu:  # the mask the GPU applied (a pre-activation within rounding of zero may take the other branch in fp64)
            yr = yr * (y.detach() > 0).double().cpu()
        yr.backward(gy.double().cpu())
        tol = {torch.float32: 2e-5, torch.bfloat16: 1.6e-2, torch.float16: 2e-3}[dt]
        def err(a, r): return float((a.double().cpu() - r).abs().max() / (r.abs().max() + 1e-6))
        errs = {"y": err(y.detach(), yr.detach())}
        gx, gxr = x.grad.double().cpu(), xr.grad
        errs["dx"] = float((gx - gxr).norm() / (gxr.norm() + 1e-3 * gy.double().norm().item()))  # (n = 2: dx is a difference of nearly equal terms)
        if affine:
            errs["dw"], errs["db"] = err(w.grad, wr.grad), err(b.grad, br.grad)
        if training:
            errs["rm"], errs["rv"] = err(rm, rm2), err(rv, rv2)
        # (n <= 4 in training mode: dx is a difference of nearly equal terms scaled by a large rstd - conditioning, any fp32 kernel)
        lim = {"y": tol, "dx": (80 if n <= 4 else 4) * tol, "dw": 6 * tol, "db": 6 * tol, "rm": max(tol, 1e-4), "rv": max(tol, 1e-4)}
        ok = all(np.isfinite(v) and v <= lim[k] for k, v in errs.items())
    except Exception as e:
        ok, errs = False, repr(e)[:200]
    bad += not ok
    if not ok:
        print(f"case {seed}: [{n}, {c}] {dt} training={training} relu={relu} affine={affine} FAIL {errs}")
print(f"soak done: {cases} cases, failures: {bad}")
