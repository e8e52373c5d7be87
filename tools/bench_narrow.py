"""Dev tool: the narrow-layer kernel (wcn_dense_rows) against the vendor GEMM on stem / head shapes.  GPU box only.

    python tools/bench_narrow.py [rows]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from warpconvnet_amd.nn.functional.sparse_conv.pointwise import narrow_rows

dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


for cin, cout in ((96, 20), (3, 32), (32, 20), (48, 64)):
    w = torch.randn(1, cin, cout, device=dev) / cin ** 0.5
    wb = w[0].to(torch.bfloat16)
    x = torch.randn(n, cin, device=dev).to(torch.bfloat16)
    dy = torch.randn(n, cout, device=dev).to(torch.bfloat16)
    mb_f = n * (cin + cout) * 2 / 1e6
    t_f = timed(lambda: narrow_rows(x, w, False))
    t_b = timed(lambda: narrow_rows(dy, w, True))
    v_f = timed(lambda: x @ w[0].to(torch.bfloat16))
    v_b = timed(lambda: dy @ w[0].to(torch.bfloat16).t())
    print(f"{n} rows {cin:3d} -> {cout:3d}: forward {t_f:7.1f} us ({mb_f / t_f:5.2f} TB/s; vendor + cast {v_f:7.1f}), "
          f"input gradient {t_b:7.1f} us ({mb_f / t_b:5.2f} TB/s; vendor + cast {v_b:7.1f})")
