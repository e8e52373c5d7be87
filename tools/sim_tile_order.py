"""Dev tool (CPU): offsets issued per GEMM tile (= size of the union of the tile's row masks) for candidate row orders, on the
bench scenes' real masks (C oracle).  This is how `tile_key` (csrc/mask_sort.h, round 5) was chosen: within the mask's own
width (26 + 1 bits, three radix passes) the pair / Gray key gives 7.19 offsets per 128-row tile on the uniform 1 M scene against
8.82 for the reference's descending-mask order (64-row tiles: 6.73 / 7.77), and leaves the surface scene where it was (9.11 / 9.08).

    python tools/sim_tile_order.py [voxels]
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import scene_surface, scene_u
from oracle import kmap as okmap
from tests.util import tile_key


def masks_of(c):
    c4 = np.concatenate([np.zeros((len(c), 1), np.int32), c], 1)
    r = okmap.kernel_map(c4, c4, (3, 3, 3))
    return r["mask"][:, 0].astype(np.int64)


def steps(mask, order, tile):
    n = len(mask) // tile * tile
    u = np.bitwise_or.reduce(mask[order][:n].reshape(-1, tile), axis=1)
    return float(sum(((u >> b) & 1).mean() for b in range(27)))


def rev13(x):
    out = np.zeros_like(x)
    for i in range(13):
        out |= ((x >> (12 - i)) & 1) << i
    return out


def igray(k, bits):
    s = 1
    k = k.copy()
    while s < bits:
        k ^= k >> s
        s <<= 1
    return k


n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
for name, gen in (("uniform", scene_u), ("surface", scene_surface)):
    m = masks_of(gen(n, seed=1000))
    lo, rh = m & 0x1FFF, rev13((m >> 14) & 0x1FFF)
    orr = lo | rh
    cands = {
        "descending mask (reference order)": m,
        "top 18 bits of the mask (2 passes)": m >> 9,
        "pairs | low member": (orr << 13) | lo,
        "pairs | low member, 18 bits (2 passes)": (orr << 5) | (lo >> 8),
        "Gray rank of the mask": igray(m, 27),
        "Gray rank of (pairs | low member)  = tile_key": tile_key(m, 27),
        "Gray rank of (pairs | both | low member), 39 bits (5 passes)": igray((orr << 26) | ((lo & rh) << 13) | lo, 39),
        "tile_key, top 22 bits": tile_key(m, 27) >> 5,
        "tile_key, top 20 bits (2 x 10-bit passes) = the builder's order": tile_key(m, 27) >> 7,
        "tile_key, top 18 bits (2 x 9-bit passes)": tile_key(m, 27) >> 9,
    }
    print(f"{name}: {len(m)} rows, {np.mean([bin(int(x)).count('1') for x in m[:50000]]):.2f} offsets per row")
    for label, key in cands.items():
        o = np.argsort(-key, kind="stable")
        print(f"  {label:62s} 128-row tiles {steps(m, o, 128):6.2f}   64-row tiles {steps(m, o, 64):6.2f}")
