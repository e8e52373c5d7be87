"""Dev tool: per-workgroup phase times of the fwd/dgrad gather-GEMM kernel (needs a library built with -DWCN_PROF)."""
import ctypes, sys
import numpy as np
import torch
sys.path.insert(0, ".")
from bench import scene_u
import warpconvnet_amd._lib as L
from warpconvnet_amd.geometry.types.voxels import Voxels
from warpconvnet_amd.nn.modules.sparse_conv import SparseConv3d

dev = torch.device("cuda:0")
N = 1_000_000
import os
TILE = 128 if os.environ.get('WCN_GG_RB1') else 256
coords = torch.from_numpy(scene_u(N, seed=1000)).to(dev)
feats = torch.randn(coords.shape[0], 64, device=dev).to(torch.bfloat16)
conv = SparseConv3d(64, 128, kernel_size=3, bias=True).to(dev)
lib = L.lib()
lib.wcn_debug_read_prof.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
lib.wcn_debug_read_prof2.argtypes = [ctypes.c_void_p, ctypes.c_size_t]




def read(n):
    buf = np.zeros((8192, 8), dtype=np.uint64)
    rc = lib.wcn_debug_read_prof(buf.ctypes.data, buf.nbytes)
    assert rc == 0
    b2 = np.zeros((8192, 4), dtype=np.uint64)
    lib.wcn_debug_read_prof2(b2.ctypes.data, b2.nbytes)
    return np.concatenate([buf[:n].astype(np.int64), b2[:n].astype(np.int64)], axis=1)

def report(name, p):
    t0 = p[:, 0].min()
    span = (p[:, 4].max() - t0) / 100.0
    life = (p[:, 4] - p[:, 0]) / 100.0
    print(f"{name}: kernel span {span:.1f} us, {len(p)} WGs, mean WG life {life.mean():.2f} us, concurrency {life.sum()/span:.0f}")
    for a, b, nm in ((0, 1, "perm/mask"), (1, 2, "nbr slab + mask OR"), (2, 3, "main loop"), (3, 4, "epilogue+drain")):
        d = (p[:, b] - p[:, a]) / 100.0
        print(f"   {nm:22s} mean {d.mean():6.2f} us  p50 {np.median(d):6.2f}  p95 {np.percentile(d,95):6.2f}")
    steps = p[:, 5]
    loop = (p[:, 3] - p[:, 2]) / 100.0
    print(f"   steps/WG mean {steps.mean():.2f}; loop us per step {loop.sum()/steps.sum():.3f}")
    cyc = p[:, 8:12].sum(0).astype(float); tot = cyc.sum()
    print('   in-loop split (wave 0): issue %.1f%%  compute %.1f%%  vmcnt wait %.1f%%  barrier %.1f%%; cycles/step %.0f' % (*(100*cyc/tot), tot/steps.sum()))
    st = (p[:, 0] - t0) / 100.0
    print(f"   WGs started within 2us: {(st<2).sum()}  within 20us: {(st<20).sum()}")

with torch.autocast("cuda", dtype=torch.bfloat16):
    for it in range(3):
        x = Voxels(batched_coordinates=coords, batched_features=feats.clone().requires_grad_(True), offsets=torch.tensor([0, N]))
        y = conv(x)
        torch.cuda.synchronize()
        nwg = min(8192, (N + TILE - 1) // TILE)
        pf = read(nwg)
        y.features.backward(torch.ones_like(y.features))
        torch.cuda.synchronize()
        pb = read(nwg)
report("fwd  (64->128)", pf)
report("dgrad(128->64)", pb)
