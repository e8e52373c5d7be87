import sys, torch
sys.path.insert(0, ".")
from bench import scene_u
from warpconvnet_amd.geometry.coords.search.torch_discrete import generate_kernel_map
dev = torch.device("cuda:0")
for n in (5000, 100000, 1000000):
    c = torch.from_numpy(scene_u(n, seed=1000)).to(dev)
    c = torch.cat([torch.zeros(len(c), 1, dtype=c.dtype, device=dev), c], 1).int().contiguous()
    km = generate_kernel_map(c, c, (1, 1, 1), (3, 3, 3))
    print(n, km._symmetric, km._self_exact, (km._nbr[:, 13] != torch.arange(len(c), device=dev)).sum().item())
