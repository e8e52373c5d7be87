"""Dictionary brute-force kernel map, independent of the hash-table restatement.  TEST INFRASTRUCTURE ONLY.

Method of the reference's own test (`tests/coords/test_kernel_map_invariants.py:205-230`): put every input
coordinate in a dict, enumerate ``out * stride + offset[k]`` for every output row and offset.
Pure-Python loops: small inputs only.
"""
from typing import Dict, List, Tuple

import numpy as np


def kernel_offsets(kernel_size, dilation=(1, 1, 1)) -> np.ndarray:
    """[K, 3] offsets, k = (i*ky + j)*kz + l, centre (s-1)//2 for odd s, 0 for even s
    (reference `geometry/coords/search/torch_discrete.py:24-56`)."""
    kx, ky, kz = (int(v) for v in kernel_size)
    c = [(s - 1) // 2 if s % 2 == 1 else 0 for s in (kx, ky, kz)]
    out = []
    for i in range(kx):
        for j in range(ky):
            for l in range(kz):
                out.append(((i - c[0]) * dilation[0], (j - c[1]) * dilation[1], (l - c[2]) * dilation[2]))
    return np.asarray(out, dtype=np.int32)


def kernel_map(in_coords: np.ndarray, out_coords: np.ndarray, kernel_size, stride=(1, 1, 1), dilation=(1, 1, 1)):
    """Returns (found [K, M], offsets [K+1], in_maps, out_maps) with buckets ordered by output row."""
    table: Dict[Tuple[int, int, int, int], int] = {}
    for i, c in enumerate(in_coords.tolist()):
        table.setdefault(tuple(c), i)  # first occurrence wins
    offs = kernel_offsets(kernel_size, dilation).tolist()
    K, M = len(offs), out_coords.shape[0]
    found = np.full((K, M), -1, dtype=np.int32)
    outs = out_coords.tolist()
    for k, (ox, oy, oz) in enumerate(offs):
        for j, (b, x, y, z) in enumerate(outs):
            found[k, j] = table.get((b, x * stride[0] + ox, y * stride[1] + oy, z * stride[2] + oz), -1)
    in_maps: List[int] = []
    out_maps: List[int] = []
    offsets = [0]
    for k in range(K):
        rows = np.nonzero(found[k] >= 0)[0]
        in_maps.extend(found[k, rows].tolist())
        out_maps.extend(rows.tolist())
        offsets.append(len(in_maps))
    return found, np.asarray(offsets, np.int32), np.asarray(in_maps, np.int32), np.asarray(out_maps, np.int32)
