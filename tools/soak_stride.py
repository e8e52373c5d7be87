# dev soak: down-sampling + strided kernel maps answered from the cell table (csrc/kmap_stride.hip) on random scenes - negative
# coordinates, several batch elements, duplicates, random strides / kernels - bit-exact vs the oracle (GPU box):
#     python tools/soak_stride.py [cases]
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import kmap as okmap
from tests.util import scene_u
from tests.test_gpu_kmap import _check_against_oracle
from warpconvnet_amd.geometry.coords.ops.stride import stride_coords
from warpconvnet_amd.geometry.coords.search.torch_discrete import generate_kernel_map, default_hints
dev = torch.device("cuda:0")
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
bad = 0
for seed in range(cases):
    rng = np.random.default_rng(2000 + seed)
    nb = int(rng.integers(1, 4))
    s = np.concatenate([scene_u(int(rng.integers(50, 9000)), 11 * seed + b, b) for b in range(nb)], 0)
    s[:, 1:] -= int(rng.integers(0, 40))
    if rng.integers(0, 2):  # duplicated rows inside the first batch element
        k = int(rng.integers(1, 50)); first = int((s[:, 0] == 0).sum())
        s = np.concatenate([s[:first], s[:k], s[first:]], 0)
    s = s.astype(np.int32)
    stride = tuple(int(v) for v in rng.choice([1, 2, 2, 2, 4], size=3))
    if stride == (1, 1, 1): stride = (2, 2, 2)
    ksize = stride if rng.integers(0, 2) else tuple(int(v) for v in rng.choice([2, 3], size=3))
    prebuild = bool(rng.integers(0, 2))
    if rng.integers(0, 4) == 0: default_hints().reset()
    try:
        a = torch.from_numpy(s).to(dev)
        if prebuild: generate_kernel_map(a, a, (1, 1, 1), (3, 3, 3))
        want, _ = okmap.stride_coords(s, stride)
        got, offs = stride_coords(a, stride, num_batches=nb, with_map=tuple(ksize) == tuple(stride))
        np.testing.assert_array_equal(got.cpu().numpy(), want)
        np.testing.assert_array_equal(offs.numpy(), np.concatenate([[0], np.cumsum(np.bincount(want[:, 0], minlength=nb))]))
        km = generate_kernel_map(a, got, stride, ksize)
        _check_against_oracle(km, s, want, ksize, stride)
    except Exception as e:
        bad += 1
        print(f"case {seed}: n={len(s)} batches={nb} stride={stride} ksize={ksize} prebuild={prebuild} FAIL {repr(e)[:300]}")
print(f"soak done: {cases} cases, failures: {bad}")
