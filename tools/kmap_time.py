"""Dev tool: time generate_kernel_map on the headline scene (HIP events) and, optionally, under rocprofv3."""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from bench import scene_u, scene_surface, time_events
from warpconvnet_amd.geometry.coords.search.torch_discrete import generate_kernel_map

dev = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
scene = sys.argv[2] if len(sys.argv) > 2 else "uniform"
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 20
c = torch.from_numpy((scene_u if scene == "uniform" else scene_surface)(N, seed=1000)).to(dev)
c = torch.cat([torch.zeros(len(c), 1, dtype=c.dtype, device=dev), c], 1).int().contiguous()
km = generate_kernel_map(c, c, (1, 1, 1), (3, 3, 3))
print("N", len(c), "pairs", int(km.offsets[-1]))
t = time_events(lambda: generate_kernel_map(c, c, (1, 1, 1), (3, 3, 3)), iters, warmup=3)
print(f"kmap build {scene}: {t * 1e3:.1f} us")
