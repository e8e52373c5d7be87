"""Dev tool: per-workgroup phase times of the channel-split gather-GEMM kernel (csrc/conv_mfma_cs.hip).
Needs the phase-stamp build: make -C warpconvnet_amd/csrc prof; WARPCONVNET_AMD_LIB=.../libwcn_hip_prof.so python tools/prof_phases.py"""
import ctypes, sys
import numpy as np
import torch
sys.path.insert(0, ".")
from bench import scene_u, scene_surface as scene_s
import warpconvnet_amd._lib as L
from warpconvnet_amd.geometry.types.voxels import Voxels
from warpconvnet_amd.nn.modules.sparse_conv import SparseConv3d

dev = torch.device("cuda:0")
N = 1_000_000
scene = scene_s if (len(sys.argv) > 1 and sys.argv[1] == "surface") else scene_u
coords = torch.from_numpy(scene(N, seed=1000)).to(dev)
feats = torch.randn(coords.shape[0], 64, device=dev).to(torch.bfloat16)
conv = SparseConv3d(64, 128, kernel_size=3, bias=True).to(dev)
lib = L.lib()
rd = lib._h.wcn_debug_read_prof_cs
rd.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
rst = lib._h.wcn_debug_reset_prof_cs


def read():
    buf = np.zeros((2048, 8), dtype=np.uint64)
    assert rd(buf.ctypes.data, buf.nbytes) == 0
    assert rst() == 0
    return buf.astype(np.float64)


def report(name, p, ms):
    p = p[p[:, 3] > 0]
    tiles, steps = p[:, 3].sum(), p[:, 4].sum()
    tot = p[:, 0] + p[:, 1] + p[:, 2]
    clk = tot.mean() / (ms * 1e3)  # clocks per us, if a workgroup lives for the whole kernel
    print(f"   WG total clocks: mean {tot.mean():.0f}  min {tot.min():.0f}  p50 {np.median(tot):.0f}  p95 {np.percentile(tot,95):.0f}  max {tot.max():.0f}")
    print(f"{name}: {ms*1e3:.1f} us, {len(p)} WGs, {tiles/len(p):.2f} tiles/WG, {steps/tiles:.2f} steps/tile, WG clocks {tot.mean():.0f} (~{clk:.0f} MHz)")
    for i, nm in enumerate(("install + masks", "step loop", "next-tile requests + epilogue")):
        print(f"   {nm:30s} {100*p[:, i].sum()/tot.sum():5.1f}%   {p[:, i].sum()/tiles:8.0f} clk/tile")
    lp = p[:, 5:8].sum(0)
    print("   in-loop (wave 0): wait+barrier %.1f%%  issue %.1f%%  compute %.1f%%; clk/step %.0f" % (*(100 * lp / lp.sum()), lp.sum() / steps))


with torch.autocast("cuda", dtype=torch.bfloat16):
    for it in range(3):
        x = Voxels(batched_coordinates=coords, batched_features=feats.clone().requires_grad_(True), offsets=torch.tensor([0, N]))
        read()
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        y = conv(x)
        torch.cuda.synchronize()
        pf = read()
        g = torch.ones_like(y.features)
        y.features.backward(g)
        torch.cuda.synchronize()
        pb = read()
# kernel times from a plain timed pair of launches (the stamps cost ~10 %)
import time
from warpconvnet_amd.nn.functional.sparse_conv.detail import hip_gemm
report("fwd  (64->128)", pf, 0.2)
report("dgrad(128->64)", pb, 0.26)
