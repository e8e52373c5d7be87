"""Dev tool: per-wave phase times of cell_neighbors (needs kmap_binned.hip built with -DWCN_PROF)."""
import ctypes, sys
import numpy as np, torch
sys.path.insert(0, ".")
from bench import scene_u
import warpconvnet_amd._lib as L
from warpconvnet_amd.geometry.coords.search.torch_discrete import generate_kernel_map
dev = torch.device("cuda:0")
N = 1_000_000
c = torch.from_numpy(scene_u(N, seed=1000)).to(dev)
c = torch.cat([torch.zeros(len(c), 1, dtype=c.dtype, device=dev), c], 1).int().contiguous()
lib = L.lib()._h if hasattr(L.lib(), "_h") else L.lib()
lib.wcn_debug_read_bprof.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
for _ in range(3):
    km = generate_kernel_map(c, c, (1, 1, 1), (3, 3, 3))
torch.cuda.synchronize()
buf = np.zeros((4096, 8), dtype=np.uint64)
assert lib.wcn_debug_read_bprof(buf.ctypes.data, buf.nbytes) == 0
p = buf[buf[:, 3] > 0].astype(np.int64)  # stamps of the LAST block every wave processed
t0 = p[:, 0].min()
print("waves sampled", len(p), "last-block span us", (p[:, 3].max() - t0) / 100.0)
for a, b, nm in ((0, 1, "halo gather"), (1, 2, "own cells + enumerate + prefetch"), (2, 3, "probe")):
    d = (p[:, b] - p[:, a]) / 100.0
    print(f"  {nm:34s} mean {d.mean():6.2f} p50 {np.median(d):6.2f} p95 {np.percentile(d,95):6.2f}")
life = (p[:, 3] - p[:, 0]) / 100.0
print("  block life mean", life.mean(), "own_cnt mean", p[:, 4].mean(), "max", p[:, 4].max())
st = (p[:, 5] - p[:, 5].min()) / 100.0
en = (p[:, 3] - p[:, 5].min()) / 100.0
print("  wave start percentiles [0,10,50,90,99,100]", np.percentile(st, [0, 10, 50, 90, 99, 100]).round(1))
print("  wave end   percentiles [0,10,50,90,99,100]", np.percentile(en, [0, 10, 50, 90, 99, 100]).round(1))
