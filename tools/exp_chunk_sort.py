"""Dev experiment: does a CHUNK-LOCAL mask sort - key = (spatial chunk of C rows, mask) on
block-ordered input - buy back the L2 locality the global mask sort destroys?  Prints, per variant, the offsets issued per
128-row tile (the GEMM's step count) and the forward / dgrad kernel times.  Under `rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum`
one variant per process gives the L2 hit rate (tools/_run.sh drives that).

    python tools/exp_chunk_sort.py <scene> <chunk_rows | 0 = global sort> [iters]
"""
import sys
import numpy as np, torch
sys.path.insert(0, ".")
from bench import scene_u, scene_surface, time_events, CIN, COUT
from warpconvnet_amd.geometry.coords.search.torch_discrete import generate_kernel_map
from warpconvnet_amd.nn.functional.sparse_conv.detail.hip_gemm import hip_forward, hip_dgrad

dev = torch.device("cuda:0")
scene = sys.argv[1] if len(sys.argv) > 1 else "uniform"
chunk = int(sys.argv[2]) if len(sys.argv) > 2 else 0
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 20
N = 1_000_000
c_np = (scene_u if scene == "uniform" else scene_surface)(N, seed=1000)
key = (((c_np[:, 0] >> 4) * 4096 + (c_np[:, 1] >> 4)) * 4096 + (c_np[:, 2] >> 4)).astype(np.int64)
c_np = c_np[np.lexsort((c_np[:, 2], c_np[:, 1], c_np[:, 0], key))]  # 16^3-block order: consecutive rows are neighbours in space
c = torch.from_numpy(np.ascontiguousarray(c_np)).to(dev)
c = torch.cat([torch.zeros(len(c), 1, dtype=c.dtype, device=dev), c], 1).int().contiguous()
km = generate_kernel_map(c, c, (1, 1, 1), (3, 3, 3))
n = len(c)
mask = km._mask[:, 0].long() & 0xFFFFFFFF
if chunk > 0:
    rows = torch.arange(n, device=dev)
    k2 = ((rows // chunk) << 32) | ((~mask) & 0xFFFFFFFF)  # ascending chunk, descending mask, ties in row order (stable)
    km._perm = torch.sort(k2, stable=True).indices.int().contiguous()
perm = km._perm.long()
tiles = mask[perm][: n // 128 * 128].view(-1, 128)
union = tiles[:, 0].clone()
for j in range(1, 128):
    union |= tiles[:, j]
steps = sum(((union >> b) & 1).sum().item() for b in range(27)) / union.numel()
x = torch.randn(n, CIN, device=dev).bfloat16()
dy = torch.randn(n, COUT, device=dev).bfloat16()
w = (torch.randn(27, CIN, COUT, device=dev) * 0.05).bfloat16()
t_f = time_events(lambda: hip_forward(x, w, km, n, "hip_mfma"), iters, warmup=3)
t_d = time_events(lambda: hip_dgrad(dy, w, km, n, "hip_mfma"), iters, warmup=3)
print(f"{scene} chunk={chunk or 'global'}: offsets issued per 128-row tile {steps:.2f}, fwd {t_f * 1e3:.1f} us, dgrad {t_d * 1e3:.1f} us")
