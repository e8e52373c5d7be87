"""dev: BatchNorm passes on [N, C] bf16 tensors, us per launch and TB/s of their own traffic (GPU box).
    python tools/bench_bn.py            (WARPCONVNET_AMD_LIB=... for another build)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from warpconvnet_amd import _lib

dev = torch.device("cuda:0")
L = _lib.lib()
st = _lib.stream_handle(dev)
for n, c in [(1_000_000, 96), (1_000_000, 32), (290_000, 64), (290_000, 96), (80_000, 128), (21_000, 256)]:
    x = torch.randn(n, c, device=dev).bfloat16(); dy = torch.randn(n, c, device=dev).bfloat16(); out = torch.empty_like(x)
    stats = torch.rand(5, c, device=dev) + 0.5
    sums = torch.empty(2, c, device=dev)
    ws = torch.empty(L.wcn_bn_workspace(c), dtype=torch.uint8, device=dev)
    p = lambda t, i=0: t.data_ptr() + 4 * c * i
    code = _lib.WCN_BF16
    def stats_():
        _lib.check(L.wcn_bn_stats(_lib.ptr(x), n, c, code, p(stats, 0), p(stats, 4), _lib.ptr(ws), ws.numel(), st), "s")
    def apply_():
        _lib.check(L.wcn_bn_apply(_lib.ptr(x), n, c, code, p(stats, 2), p(stats, 3), 1, _lib.ptr(out), st), "a")
    def red_():
        _lib.check(L.wcn_bn_backward_reduce(_lib.ptr(dy), _lib.ptr(x), p(stats, 2), p(stats, 3), n, c, code, p(stats, 0), p(stats, 1),
                                            p(sums, 0), p(sums, 1), _lib.ptr(ws), ws.numel(), st), "r")
    def bapply_():
        _lib.check(L.wcn_bn_backward_apply(_lib.ptr(dy), _lib.ptr(x), p(stats, 2), p(stats, 3), n, c, code, p(stats, 0), p(stats, 1),
                                           None, p(sums, 0), p(sums, 1), _lib.ptr(out), st), "b")
    row = [f"[{n}, {c}]"]
    for name, fn, passes in (("stats", stats_, 1), ("apply", apply_, 2), ("bwd_reduce", red_, 2), ("bwd_apply", bapply_, 3)):
        for _ in range(3): fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        # alternate two tensors' worth of other traffic between launches? no: back-to-back, tensors > Infinity Cache at 1M x 96
        e0.record()
        for _ in range(20): fn()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1000 / 20
        row.append(f"{name} {us:6.1f} us {passes * n * c * 2 / us / 1e6:5.2f} TB/s")
    print(" | ".join(row))
