"""ctypes front-end of ``kmap_oracle.c`` (numpy in / numpy out).  TEST INFRASTRUCTURE ONLY."""
import ctypes
import os
import subprocess
from typing import Dict, Tuple

import numpy as np

_DIR = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_DIR, "_build", "libkmap_oracle.so")
_LIB = None


def build() -> str:
    res = subprocess.run(["make", "-C", _DIR], capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError(f"building the C oracle failed:\n{res.stdout}\n{res.stderr}")
    return _SO


def _lib():
    global _LIB
    if _LIB is None:
        if not os.path.exists(_SO):
            build()
        L = ctypes.CDLL(_SO)
        vp, i64, i32 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int
        L.oracle_next_pow2.restype = i64
        L.oracle_next_pow2.argtypes = [i64]
        L.oracle_check_range.restype = i32
        L.oracle_check_range.argtypes = [vp, i64]
        L.oracle_hash_build.restype = i32
        L.oracle_hash_build.argtypes = [vp, i64, vp, vp, i64]
        L.oracle_hash_search.restype = None
        L.oracle_hash_search.argtypes = [vp, vp, i64, vp, i64, vp]
        L.oracle_kernel_map_found.restype = None
        L.oracle_kernel_map_found.argtypes = [vp, vp, i64, vp, i64, vp, vp, vp, vp]
        L.oracle_compact.restype = i64
        L.oracle_compact.argtypes = [vp, i64, i32, vp, vp, vp]
        L.oracle_pair_mask.restype = None
        L.oracle_pair_mask.argtypes = [vp, i64, i32, i32, vp]
        L.oracle_reverse.restype = None
        L.oracle_reverse.argtypes = [vp, i64, i64, i32, i32, vp, vp]
        L.oracle_stride_coords.restype = i64
        L.oracle_stride_coords.argtypes = [vp, i64, vp, vp, vp]
        _LIB = L
    return _LIB


def _p(a: np.ndarray):
    return a.ctypes.data_as(ctypes.c_void_p)


def _c(a, dtype=np.int32) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=dtype)


class HashTable:
    """Reference-layout table: separate ``keys`` (uint64) and ``values`` (int32) arrays."""

    def __init__(self, coords: np.ndarray, capacity: int = None):
        L = _lib()
        coords = _c(coords)
        assert coords.ndim == 2 and coords.shape[1] == 4
        n = coords.shape[0]
        if L.oracle_check_range(_p(coords), n):
            raise ValueError("coordinate out of packed range")
        self.capacity = int(L.oracle_next_pow2(capacity if capacity is not None else max(16, 2 * n)))
        self.keys = np.zeros(self.capacity, dtype=np.uint64)
        self.values = np.zeros(self.capacity, dtype=np.int32)
        if L.oracle_hash_build(_p(coords), n, _p(self.keys), _p(self.values), self.capacity):
            raise RuntimeError("hash table is full")
        self.coords = coords

    def search(self, queries: np.ndarray) -> np.ndarray:
        q = _c(queries)
        out = np.empty(q.shape[0], dtype=np.int32)
        _lib().oracle_hash_search(_p(self.keys), _p(self.values), self.capacity, _p(q), q.shape[0], _p(out))
        return out


def kernel_map(in_coords, out_coords, kernel_size, stride=(1, 1, 1), dilation=(1, 1, 1)) -> Dict[str, np.ndarray]:
    """Full oracle kernel map.  Returns found [K, M], offsets [K+1], in_maps, out_maps (buckets ordered by output
    row), mask [M, mw], and the reverse table rev [K, N_in] / rev_mask [N_in, mw]."""
    L = _lib()
    in_coords, out_coords = _c(in_coords), _c(out_coords)
    ks, st, dl = _c(kernel_size), _c(stride), _c(dilation)
    table = HashTable(in_coords)
    M, N = out_coords.shape[0], in_coords.shape[0]
    K = int(np.prod(ks))
    mw = (K + 31) // 32
    found = np.empty((K, M), dtype=np.int32)
    L.oracle_kernel_map_found(_p(table.keys), _p(table.values), table.capacity, _p(out_coords), M, _p(ks), _p(st), _p(dl),
                              _p(found))
    offsets = np.zeros(K + 1, dtype=np.int32)
    total = int(L.oracle_compact(_p(found), M, K, _p(offsets), None, None))
    in_maps = np.empty(total, dtype=np.int32)
    out_maps = np.empty(total, dtype=np.int32)
    L.oracle_compact(_p(found), M, K, _p(offsets), _p(in_maps), _p(out_maps))
    mask = np.zeros((M, mw), dtype=np.uint32)
    L.oracle_pair_mask(_p(found), M, K, mw, _p(mask))
    rev = np.empty((K, N), dtype=np.int32)
    rev_mask = np.zeros((N, mw), dtype=np.uint32)
    L.oracle_reverse(_p(found), M, N, K, mw, _p(rev), _p(rev_mask))
    return dict(found=found, offsets=offsets, in_maps=in_maps, out_maps=out_maps, mask=mask, rev=rev, rev_mask=rev_mask)


def stride_coords(coords, stride) -> Tuple[np.ndarray, np.ndarray]:
    """(unique floor(coords/stride) in first-occurrence order [M, 4], first source row [M])."""
    coords = _c(coords)
    st = _c(stride)
    out = np.empty_like(coords)
    first = np.empty(coords.shape[0], dtype=np.int32)
    cnt = int(_lib().oracle_stride_coords(_p(coords), coords.shape[0], _p(st), _p(out), _p(first)))
    return out[:cnt].copy(), first[:cnt].copy()


def compact_rows(found: np.ndarray, pitch: int = 16):
    """The COMPACT neighbour rows the product's binned builder writes for one-word masks (`warpconvnet_amd/csrc/kmap_cells.h`;
    no reference counterpart - the reference keeps the dense `found_in_coord_index` [K, M] of `cuhash_kernel_map.cu:93-134`):
    row m = [mask, ids of its SET offsets in ascending k ..., don't-care].  ``found``: the oracle's [K, M] table (-1 = absent).
    Returns ``(rows [M, pitch] int32 with the don't-care words zeroed, mask [M] uint32, fits [M] bool)`` - a row with more
    than ``pitch - 1`` neighbours does not fit (the product then rebuilds with dense rows)."""
    K, M = found.shape
    assert K <= 31
    present = found >= 0
    mask = (present.astype(np.uint32) << np.arange(K, dtype=np.uint32)[:, None]).sum(0).astype(np.uint32)
    count = present.sum(0)
    rows = np.zeros((M, pitch), np.int32)
    rows[:, 0] = mask.view(np.int32)
    rank = np.cumsum(present, 0) - 1  # position of offset k among the set offsets of its row
    ks, ms = np.nonzero(present)
    ok = rank[ks, ms] < pitch - 1
    rows[ms[ok], 1 + rank[ks, ms][ok]] = found[ks, ms][ok]
    return rows, mask, count <= pitch - 1


def densify_rows(rows: np.ndarray, num_offsets: int) -> np.ndarray:
    """Inverse of :func:`compact_rows` for rows that fit: the dense [M, K] table (-1 = absent)."""
    M = rows.shape[0]
    mask = rows[:, 0].view(np.uint32)
    out = np.full((M, num_offsets), -1, np.int32)
    at = np.ones(M, np.int64)
    for k in range(num_offsets):
        has = ((mask >> np.uint32(k)) & np.uint32(1)).astype(bool)
        out[has, k] = rows[has, at[has]]
        at += has
    return out
