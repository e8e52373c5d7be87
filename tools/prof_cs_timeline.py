"""Dev tool: start / end wall-clock of every workgroup of one gather-GEMM launch (library built with `make prof`): how full
the chip is over the launch, and what the last partial round of workgroups costs.   usage: prof_cs_timeline.py fwd|dgrad"""
import ctypes, sys
import numpy as np, torch
sys.path.insert(0, ".")
import bench
import warpconvnet_amd._lib as L
from warpconvnet_amd.geometry.types.voxels import Voxels

dev = torch.device("cuda:0")
which = sys.argv[1] if len(sys.argv) > 1 else "fwd"
args = bench.parse_args([])
coords, feats, grad_out, offsets, conv, params = bench.build_workload(args, dev, 0)
lib = L.lib()._h if hasattr(L.lib(), "_h") else L.lib()
lib.wcn_debug_read_time_cs.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
for _ in range(4):
    x = Voxels(coords, feats, offsets=offsets)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y = conv(x)
    if which == "dgrad":
        y.batched_features.batched_tensor.backward(grad_out)
torch.cuda.synchronize()
buf = np.zeros((16384, 4), dtype=np.uint64)
assert lib.wcn_debug_read_time_cs(buf.ctypes.data, buf.nbytes) == 0
t = buf[buf[:, 1] > 0].astype(np.int64)
t0 = t[:, 0].min()
st, en = (t[:, 0] - t0) / 100.0, (t[:, 1] - t0) / 100.0
dur = en - st
span = en.max()
print(which, "workgroups", len(t), "span us", span, "mean workgroup us", dur.mean().round(2), "p10/p50/p90", np.percentile(dur, [10, 50, 90]).round(2))
# resident workgroups over time
edges = np.linspace(0, span, 41)
act = [((st < b) & (en > a)).sum() for a, b in zip(edges[:-1], edges[1:])]
print("resident workgroups per 1/40 of the span:", act)
peak = max(act)
print("sum of workgroup time / (peak resident x span) =", (dur.sum() / (peak * span)).round(3))
print("time at which 90 / 95 / 99 / 100 % of the workgroups have ended:", np.percentile(en, [90, 95, 99, 100]).round(1))
# turnover: per CU (XCC id, SE, CU id from HW_ID), every start after the first round pairs with the earliest unpaired end
hw = t[:, 2]
xcc = (hw >> 32) & 0xF
hwid = hw & 0xFFFFFFFF
cu = (hwid >> 8) & 0xF
sh = (hwid >> 12) & 0x1
se = (hwid >> 13) & 0x7
key = ((xcc * 8 + se) * 2 + sh) * 16 + cu
gaps = []
for kcu in np.unique(key):
    sel = key == kcu
    s_k, e_k = np.sort(st[sel]), np.sort(en[sel])
    # slots: starts in order; the j-th start beyond the resident count follows the (j - resident)-th end
    resident = int((s_k < s_k[0] + 3.0).sum())
    for j in range(resident, len(s_k)):
        gaps.append(s_k[j] - e_k[j - resident])
gaps = np.array(gaps)
print("CUs seen", len(np.unique(key)), "turnover gap us (next start - freeing end): mean", gaps.mean().round(2), "p10/p50/p90", np.percentile(gaps, [10, 50, 90]).round(2))
