"""Dev tool: per-wave phase times of cell_neighbors (needs kmap_binned.hip built with -DWCN_PROF: make prof)."""
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
buf = np.zeros((4096, 12), dtype=np.uint64)
assert lib.wcn_debug_read_bprof(buf.ctypes.data, buf.nbytes) == 0
p = buf[buf[:, 11] > 0].astype(np.int64)
nb = p[:, 8]
print("waves", len(p), "blocks per wave mean", nb.mean(), "max", nb.max())
t0 = p[:, 9].min()
for i, nm in ((0, "wait at the loop top"), (1, "halo scatter"), (4, "own cells + enumerate"), (5, "flush previous rows"), (6, "gather issue (next block)"), (2, "head loads issue + fence"), (3, "probe")):
    d = p[:, i] / 100.0
    print(f"  {nm:36s} per wave {d.mean():6.2f} us   per block {(d / np.maximum(nb, 1)).mean():6.2f}")
print("  prologue (kernel entry -> loop)      per wave", ((p[:, 10] - p[:, 9]) / 100.0).mean())
print("  wave start percentiles [0,50,100]", np.percentile((p[:, 9] - t0) / 100.0, [0, 50, 100]).round(1))
print("  wave end   percentiles [0,10,50,90,100]", np.percentile((p[:, 11] - t0) / 100.0, [0, 10, 50, 90, 100]).round(1))
