"""dev: BatchNorm passes on [N, C] bf16 tensors that do NOT sit in the Infinity Cache: every launch works on the next of
`sets` tensor sets (sets x traffic of a pass >> 256 MB), us per launch and TB/s of the pass's own traffic.
    python tools/bench_bn_cold.py            (WARPCONVNET_AMD_LIB=... for another build)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from warpconvnet_amd import _lib

dev = torch.device("cuda:0")
L = _lib.lib()
st = _lib.stream_handle(dev)
code = _lib.WCN_BF16
for n, c in [(1_000_000, 96), (1_000_000, 32), (290_000, 96), (290_000, 64), (80_000, 128)]:
    sets = max(4, int(2.0e9 // (n * c * 2 * 3)))
    xs = [torch.randn(n, c, device=dev).bfloat16() for _ in range(sets)]
    dys = [torch.randn(n, c, device=dev).bfloat16() for _ in range(sets)]
    outs = [torch.empty(n, c, device=dev, dtype=torch.bfloat16) for _ in range(sets)]
    stats = torch.rand(5, c, device=dev) + 0.5
    sums = torch.empty(2, c, device=dev)
    ws = torch.zeros(L.wcn_bn_workspace(c), dtype=torch.uint8, device=dev)
    p = lambda t, i=0: t.data_ptr() + 4 * c * i
    def stats_(i):
        _lib.check(L.wcn_bn_stats(_lib.ptr(xs[i]), n, c, code, p(stats, 0), p(stats, 4), _lib.ptr(ws), ws.numel(), st), "s")
    def apply_(i):
        _lib.check(L.wcn_bn_apply(_lib.ptr(xs[i]), n, c, code, p(stats, 2), p(stats, 3), 1, _lib.ptr(outs[i]), st), "a")
    def red_(i):
        _lib.check(L.wcn_bn_backward_reduce(_lib.ptr(dys[i]), _lib.ptr(xs[i]), p(stats, 2), p(stats, 3), n, c, code, p(stats, 0), p(stats, 1),
                                            p(sums, 0), p(sums, 1), _lib.ptr(ws), ws.numel(), st), "r")
    def bapply_(i):
        _lib.check(L.wcn_bn_backward_apply(_lib.ptr(dys[i]), _lib.ptr(xs[i]), p(stats, 2), p(stats, 3), n, c, code, p(stats, 0), p(stats, 1),
                                           None, p(sums, 0), p(sums, 1), _lib.ptr(outs[i]), st), "b")
    row = [f"[{n}, {c}] x{sets}"]
    for name, fn, passes in (("stats", stats_, 1), ("apply", apply_, 2), ("bwd_reduce", red_, 2), ("bwd_apply", bapply_, 3)):
        for i in range(sets): fn(i)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 3 * sets
        e0.record()
        for j in range(reps): fn(j % sets)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1000 / reps
        row.append(f"{name} {us:6.1f} us {passes * n * c * 2 / us / 1e6:5.2f} TB/s")
    print(" | ".join(row))
    del xs, dys, outs
# reference point: a plain device copy of the same size, cold
n, c = 1_000_000, 96
a = [torch.randn(n, c, device=dev).bfloat16() for _ in range(6)]
b = [torch.empty_like(a[0]) for _ in range(6)]
for i in range(6): b[i].copy_(a[i])
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for j in range(18): b[j % 6].copy_(a[j % 6])
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1000 / 18
print(f"torch copy [{n}, {c}] cold: {us:.1f} us {2 * n * c * 2 / us / 1e6:.2f} TB/s")
