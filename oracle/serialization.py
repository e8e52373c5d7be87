"""Z-order codes, orderings and sparse pooling restated in numpy.  TEST INFRASTRUCTURE ONLY.

Parity status: the reference computes these with CUDA kernels (`warpconvnet/csrc/morton_code.cu:14-58`,
`_C.utils.segmented_sort`) that cannot run in the build container, so this restatement is pinned by the reference's
*tests' semantics* (`tests/coords/test_serialization.py`: codes sort inside every batch element, axis permutation
= encoding the permuted columns with MORTON_XYZ, inverse permutation restores the input, distinct coordinates get
distinct codes) and by hand-computed known answers (tests/test_oracle_golden.py) - **parity unpinned** by reference
outputs.
"""
from typing import Optional, Tuple

import numpy as np

AXES = {"morton_xyz": (0, 1, 2), "morton_xzy": (0, 2, 1), "morton_yxz": (1, 0, 2), "morton_yzx": (1, 2, 0),
        "morton_zxy": (2, 0, 1), "morton_zyx": (2, 1, 0), "morton": (0, 1, 2)}


def _interleave3(a: np.ndarray, b: np.ndarray, c: np.ndarray, bits: int) -> np.ndarray:
    """bit i of a -> 3i, of b -> 3i+1, of c -> 3i+2 (morton_code.cu:33, 57: z << 2 | y << 1 | x)."""
    out = np.zeros(a.shape, dtype=np.uint64)
    for i in range(bits):
        out |= ((a.astype(np.uint64) >> np.uint64(i)) & np.uint64(1)) << np.uint64(3 * i)
        out |= ((b.astype(np.uint64) >> np.uint64(i)) & np.uint64(1)) << np.uint64(3 * i + 1)
        out |= ((c.astype(np.uint64) >> np.uint64(i)) & np.uint64(1)) << np.uint64(3 * i + 2)
    return out


def morton_code(coords: np.ndarray, order: str = "morton_xyz") -> np.ndarray:
    """[N,3] (x,y,z): 21-bit interleave; [N,4] (b,x,y,z): (b << 48) | 16-bit interleave; per-column minimum removed
    first (serialization.py:211-212), axis permutation applied to the spatial columns (serialization.py:215-225)."""
    coords = np.asarray(coords)
    if coords.shape[0] == 0:
        return np.empty(0, np.int64)
    c = (coords - coords.min(0)).astype(np.int64)
    ax = AXES[order]
    sp = c.shape[1] - 3
    x, y, z = (c[:, sp + ax[j]] for j in range(3))
    if c.shape[1] == 3:
        return _interleave3(x, y, z, 21).astype(np.int64)
    m = _interleave3(x, y, z, 16)  # the & 0xFFFFFFFFFFFF of morton_code.cu:39
    return ((c[:, 0].astype(np.uint64) << np.uint64(48)) | m).astype(np.int64)


def encode_perm(coords: np.ndarray, offsets: Optional[np.ndarray] = None, order: str = "morton_xyz") -> Tuple[np.ndarray, np.ndarray]:
    """(codes, perm): perm sorts the codes ascending, inside every [offsets[b], offsets[b+1]) when offsets are given
    (serialization.py:155-161), stable."""
    codes = morton_code(coords, order)
    if offsets is None:
        return codes, np.argsort(codes, kind="stable")
    perm = np.empty(len(codes), np.int64)
    for b in range(len(offsets) - 1):
        s, e = int(offsets[b]), int(offsets[b + 1])
        perm[s:e] = s + np.argsort(codes[s:e], kind="stable")
    return codes, perm


def sparse_reduce(feats: np.ndarray, in_maps: np.ndarray, out_maps: np.ndarray, num_out: int, reduction: str) -> np.ndarray:
    """Pooling over a kernel map: out[m] = reduce over the pairs (i, m) of feats[i]; outputs without a pair are zero
    (`warpconvnet/nn/functional/sparse_pool.py:84-110`: to_csr -> row_reduction -> zero fill)."""
    feats = np.asarray(feats, np.float64)
    out = np.zeros((num_out, feats.shape[1]), np.float64)
    if reduction in ("sum", "mean"):
        np.add.at(out, out_maps, feats[in_maps])
        if reduction == "mean":
            cnt = np.bincount(out_maps, minlength=num_out).astype(np.float64)
            out /= np.maximum(cnt, 1.0)[:, None]
        return out
    fill = -np.inf if reduction == "max" else np.inf
    acc = np.full_like(out, fill)
    (np.maximum if reduction == "max" else np.minimum).at(acc, out_maps, feats[in_maps])
    has = np.bincount(out_maps, minlength=num_out) > 0
    out[has] = acc[has]
    return out
