"""Dev experiment: how much of the forward / dgrad gather kernels' time is memory.  The shipped launches (1 M-voxel uniform scene,
64 -> 128 and 128 -> 64, bf16) on (a) the real neighbour table, (b) the table with every neighbour id folded into the first 8 192
rows (every gather an L2 hit; same masks, same steps), (c) real table, output rows in natural order (perm = identity: tiles of
unrelated masks, more steps).  GPU box only.

    python tools/exp_gather_floor.py
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from bench import scene_u
from warpconvnet_amd import _lib
from warpconvnet_amd.geometry.coords.search.torch_discrete import generate_kernel_map
from warpconvnet_amd.nn.functional.sparse_conv.detail import hip_gemm

dev = torch.device("cuda:0")
N, CIN, COUT, K = 1_000_000, 64, 128, 27
coords = torch.from_numpy(scene_u(N, 1)).to(dev)
n = coords.shape[0]
bc = torch.cat([torch.zeros(n, 1, dtype=torch.int32, device=dev), coords], 1).contiguous()
km = generate_kernel_map(bc, bc, (1, 1, 1), (3, 3, 3))
x = torch.randn(n, CIN, device=dev).to(torch.bfloat16)
dy = torch.randn(n, COUT, device=dev).to(torch.bfloat16)
w = torch.randn(K, CIN, COUT, device=dev) * 0.05
L = _lib.lib()
st = _lib.stream_handle(dev)
wp_f = hip_gemm.pack_weight(w.to(torch.bfloat16), False, False)
wp_d = hip_gemm.pack_weight(w.to(torch.bfloat16), True, True)
y = torch.empty(n, COUT, dtype=torch.bfloat16, device=dev)
dx = torch.empty(n, CIN, dtype=torch.bfloat16, device=dev)


def timed(f):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 20 * 1e3


def fwd(nbr, perm):
    return timed(lambda: _lib.check(L.wcn_conv_gather_gemm(_lib.ptr(x), _lib.ptr(wp_f), _lib.ptr(y), _lib.ptr(nbr), _lib.ptr(km._mask),
                                                          _lib.ptr(perm), None, n, n, CIN, COUT, K, _lib.WCN_BF16, _lib.WCN_ALGO_MFMA,
                                                          0, 0, st), "fwd"))


def dgrad(nbr, perm):
    return timed(lambda: _lib.check(L.wcn_conv_gather_gemm(_lib.ptr(dy), _lib.ptr(wp_d), _lib.ptr(dx), _lib.ptr(nbr), _lib.ptr(km._mask),
                                                          _lib.ptr(perm), None, n, n, COUT, CIN, K, _lib.WCN_BF16, _lib.WCN_ALGO_MFMA,
                                                          1, 1, st), "dgrad"))


nbr = km._nbr
folded = torch.where(nbr >= 0, nbr & 8191, nbr).contiguous()
folded[:, 31] = nbr[:, 31]  # (the mask column of the row)
ident = torch.arange(n, dtype=torch.int32, device=dev)
print(f"(a) real table, tile order                  fwd {fwd(nbr, km._perm):7.1f} us   dgrad {dgrad(nbr, km._perm):7.1f} us")
print(f"(b) neighbour ids folded into 8 192 rows    fwd {fwd(folded, km._perm):7.1f} us   dgrad {dgrad(folded, km._perm):7.1f} us")
print(f"(c) real table, rows in natural order       fwd {fwd(nbr, ident):7.1f} us   dgrad {dgrad(nbr, ident):7.1f} us")
print(f"(d) folded ids, rows in natural order       fwd {fwd(folded, ident):7.1f} us   dgrad {dgrad(folded, ident):7.1f} us")
