"""Dev experiment: is wgrad sensitive to the order of the pairs inside a bucket?  Row order (today) vs mask-sorted-tile
order (what a per-tile emitter would produce).  Usage: python tools/exp_wgrad_order.py [uniform|surface]"""
import sys
import numpy as np, torch
sys.path.insert(0, ".")
from bench import scene_u, scene_surface, time_events, CIN, COUT, KVOL
from warpconvnet_amd import _lib
from warpconvnet_amd.geometry.coords.search.torch_discrete import generate_kernel_map

dev = torch.device("cuda:0")
scene = sys.argv[1] if len(sys.argv) > 1 else "uniform"
N = 1_000_000
c = torch.from_numpy((scene_u if scene == "uniform" else scene_surface)(N, seed=1000)).to(dev)
N = len(c)
bc = torch.cat([torch.zeros(N, 1, dtype=c.dtype, device=dev), c], 1).int().contiguous()
km = generate_kernel_map(bc, bc, (1, 1, 1), (3, 3, 3))
X = torch.randn(N, CIN, device=dev).bfloat16()
dY = torch.randn(N, COUT, device=dev).bfloat16()
L = _lib.lib()
stream = _lib.stream_handle(dev)
dw = torch.empty(KVOL, CIN, COUT, dtype=torch.float32, device=dev)
db = torch.empty(COUT, dtype=torch.float32, device=dev)
ws_bytes = L.wcn_conv_wgrad_workspace(KVOL, CIN, COUT, _lib.WCN_ALGO_MFMA)
ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)

def run(i, o):
    L.wcn_conv_wgrad_bias(_lib.ptr(X), _lib.ptr(dY), _lib.ptr(dw), _lib.ptr(i), _lib.ptr(o), _lib.ptr(km._offsets_dev), N, N,
                          CIN, COUT, KVOL, _lib.WCN_BF16, KVOL // 2, _lib.ptr(db), _lib.ptr(ws), ws_bytes, stream)

i0, o0 = km.in_maps_device, km.out_maps_device
t0 = time_events(lambda: run(i0, o0), 20, 3)
ref = dw.clone()
# reorder every bucket by the position of the output row in the mask-sorted permutation
pos = torch.empty(N, dtype=torch.int64, device=dev)
pos[km._perm.long()] = torch.arange(N, device=dev)
offs = km.offsets.tolist()
bucket = torch.repeat_interleave(torch.arange(KVOL, device=dev), torch.tensor([offs[k + 1] - offs[k] for k in range(KVOL)], device=dev))
order = torch.argsort(bucket * N + pos[o0.long()])
i1, o1 = i0[order].contiguous(), o0[order].contiguous()
t1 = time_events(lambda: run(i1, o1), 20, 3)
err = (dw - ref).abs().max().item() / ref.abs().max().item()
# and fully random order inside the bucket
order2 = torch.argsort(bucket * N + torch.randperm(N, device=dev)[o0.long()])
i2, o2 = i0[order2].contiguous(), o0[order2].contiguous()
t2 = time_events(lambda: run(i2, o2), 20, 3)
print(f"{scene}: wgrad row-order {t0*1e3:.1f} us, sorted-tile order {t1*1e3:.1f} us (rel diff {err:.2e}), random {t2*1e3:.1f} us")
