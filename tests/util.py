"""Shared helpers of the test-suite: synthetic scenes and oracle-backed kernel maps."""
import numpy as np
import torch


def scene_u(n, seed, batch=0):
    """Reference-style uniform scene: extent 2*ceil(n^(1/3)), draw 1.3n, unique (first occurrence), truncate."""
    rng = np.random.default_rng(seed)
    extent = 2 * int(np.ceil(n ** (1.0 / 3.0)))
    c = rng.integers(0, extent, size=(int(1.3 * n), 3))
    _, first = np.unique(c, axis=0, return_index=True)
    c = c[np.sort(first)][:n].astype(np.int32)
    return np.concatenate([np.full((len(c), 1), batch, np.int32), c], axis=1)


def scene_surface(side, seed, batch=0):
    """Surface-like scene: two height-field sheets over a side x side grid."""
    rng = np.random.default_rng(seed)
    xs, ys = np.meshgrid(np.arange(side), np.arange(side), indexing="ij")
    sheets = []
    for l in range(2):
        a1, a2 = rng.uniform(3, 9), rng.uniform(1, 4)
        f = rng.uniform(0.02, 0.15, size=4)
        ph = rng.uniform(0, 6.28, size=3)
        z = np.round(40 * l + a1 * np.sin(f[0] * xs + ph[0]) * np.cos(f[1] * ys + ph[1]) + a2 * np.sin(f[2] * xs + f[3] * ys + ph[2]))
        sheets.append(np.stack([xs.ravel(), ys.ravel(), z.ravel().astype(np.int64)], 1))
    c = np.unique(np.concatenate(sheets, 0), axis=0).astype(np.int32)
    rng.shuffle(c)
    return np.concatenate([np.full((len(c), 1), batch, np.int32), c], axis=1)


def rel_max_err(a: torch.Tensor, ref: torch.Tensor) -> float:
    """max|a - ref| / max|ref|  (metric of the reference's tests/nn/test_kernel_correctness.py:139-145)."""
    a, ref = a.double().cpu(), ref.double().cpu()
    denom = ref.abs().max().item()
    return (a - ref).abs().max().item() / (denom if denom > 0 else 1.0)


def sort_buckets(in_maps, out_maps, offsets):
    """Sort every bucket by (out, in) so pair sets can be compared independent of order."""
    in_maps, out_maps = np.asarray(in_maps).copy(), np.asarray(out_maps).copy()
    for k in range(len(offsets) - 1):
        s, e = int(offsets[k]), int(offsets[k + 1])
        order = np.lexsort((in_maps[s:e], out_maps[s:e]))
        in_maps[s:e], out_maps[s:e] = in_maps[s:e][order], out_maps[s:e][order]
    return in_maps, out_maps


def tile_key(mask: np.ndarray, num_offsets: int) -> np.ndarray:
    """numpy restatement of `tile_key` (warpconvnet_amd/csrc/mask_sort.h): the key the builder's row order (`perm`) is sorted by -
    descending, ties in ascending row order.  Odd kernel volumes K = 2c + 1 <= 31: mirror-image offsets (k, K-1-k) paired, touched
    pairs in the high half, the low member of each pair in the low half, ranked in reflected-Gray order, centre offset last;
    any other volume: the mask word itself (the reference's descending-mask order, mask_data_kernels.cu:187-220)."""
    m = mask.astype(np.int64) & 0xFFFFFFFF
    K = int(num_offsets)
    if not (3 <= K <= 31 and K % 2 == 1):
        return m
    c = K // 2
    half = (1 << c) - 1
    lo = m & half
    hi = (m >> (c + 1)) & half
    rh = np.zeros_like(hi)
    for i in range(c):
        rh |= ((hi >> (c - 1 - i)) & 1) << i
    k = ((lo | rh) << c) | lo
    for s in (1, 2, 4, 8, 16):
        k ^= k >> s
    return (k << 1) | ((m >> c) & 1)


def tile_order_key(mask: np.ndarray, num_offsets: int) -> np.ndarray:
    """What `perm` (wcn_kmap_tally_sort / wcn_mask_tile_order) is sorted by, descending and stable: `tile_key`, and for keys above
    18 bits its top 20 bits only (two 10-bit radix passes instead of three 9-bit ones: `sort_plan`, csrc/mask_sort.hip)."""
    K = int(num_offsets)
    k = tile_key(mask, K)
    if 3 <= K <= 31 and K % 2 == 1 and K > 18:
        k = k >> max(K - 20, 0)
    return k
