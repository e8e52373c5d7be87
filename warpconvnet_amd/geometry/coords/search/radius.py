"""Radius search: all reference points within ``radius`` of every query (``dist <= radius``).

Interface of the reference (`warpconvnet/geometry/coords/search/radius.py:162-291`): ``radius_search`` returns
``(neighbor_index [Q] int32, neighbor_distance [Q] fp32, neighbor_split [M+1] int32)``, ``batched_radius_search`` the
concatenation over batch elements with global row ids (int64).  GPU tensors: a dense cell list over the bounding box of
the reference points with cell size >= radius (shared with the kNN, `knn.build_cell_grid`), then the reference's two
passes - count per query, exclusive scan on the device, write (`wcn_radius_grid_count / _write`, csrc/points.hip).  The
reference hashes the occupied cells instead (`radius.py:33-61`); the 27-cell walk and the ``dist^2 <= radius^2`` test are
the same.  Within a query the neighbours come in cell-walk order (deterministic); the reference's order depends on an
unstable argsort, so callers must not rely on it.  CPU tensors: the reference's own brute force (cdist), also the
oracle of the GPU tests.
"""
import ctypes
from typing import Optional, Tuple

import torch
from torch import Tensor

from warpconvnet_amd import _lib
from warpconvnet_amd.geometry.coords.search.knn import build_cell_grid


def _radius_search_cdist(points: Tensor, queries: Tensor, radius: float, chunk_size: int = 4096):
    idx, dist, counts = [], [], []
    for s in range(0, queries.shape[0], chunk_size):
        d = torch.cdist(queries[s : s + chunk_size], points)
        mask = d <= radius
        q, p = mask.nonzero(as_tuple=True)  # row-major: ascending query, then ascending point index
        idx.append(p)
        dist.append(d[q, p])
        counts.append(mask.sum(1))
    counts = torch.cat(counts) if counts else torch.zeros(0, dtype=torch.int64, device=queries.device)
    split = torch.zeros(queries.shape[0] + 1, dtype=torch.int32, device=queries.device)
    split[1:] = torch.cumsum(counts, 0)
    index = torch.cat(idx).int() if idx else torch.zeros(0, dtype=torch.int32, device=queries.device)
    distance = torch.cat(dist).float() if dist else torch.zeros(0, dtype=torch.float32, device=queries.device)
    return index, distance, split


def _radius_search_grid(points: Tensor, queries: Tensor, radius: float):
    dev = points.device
    p32, q32 = points.float().contiguous(), queries.float().contiguous()
    m = q32.shape[0]
    # cell a hair larger than the radius: a neighbour is then at most one cell away however the divisions round
    ref_sorted, ref_ids, cell_start, lo, h, dims = build_cell_grid(p32, cell_size=radius * (1.0 + 1e-4) + 1e-12)
    args = (_lib.ptr(ref_sorted), _lib.ptr(ref_ids), _lib.ptr(cell_start), (ctypes.c_float * 3)(*lo), ctypes.c_float(h),
            _lib.i3(dims), _lib.ptr(q32), m, ctypes.c_float(radius))
    L, stream = _lib.lib(), _lib.stream_handle(dev)
    counts = torch.empty(m, dtype=torch.int32, device=dev)
    _lib.check(L.wcn_radius_grid_count(*args, _lib.ptr(counts), stream), "wcn_radius_grid_count")
    splits = torch.zeros(m + 1, dtype=torch.int64, device=dev)
    torch.cumsum(counts, 0, out=splits[1:])
    total = int(splits[-1].item())  # the one host read the reference also makes (radius.py:84)
    index = torch.empty(total, dtype=torch.int32, device=dev)
    distance = torch.empty(total, dtype=torch.float32, device=dev)
    if total > 0:
        _lib.check(L.wcn_radius_grid_write(*args, _lib.ptr(splits), _lib.ptr(index), _lib.ptr(distance), stream),
                   "wcn_radius_grid_write")
    return index, distance, splits.int()


@torch.no_grad()
def radius_search(points: Tensor, queries: Tensor, radius: float, grid_dim=None, chunk_size: int = 4096
                  ) -> Tuple[Tensor, Tensor, Tensor]:
    """(neighbor_index [Q] int32, neighbor_distance [Q] fp32, neighbor_split [M+1] int32); ``grid_dim`` is unused, kept
    for API compatibility like in the reference."""
    assert points.is_contiguous() and queries.is_contiguous(), "points and queries must be contiguous"
    assert radius >= 0, "radius must be non-negative"
    dev = queries.device
    if queries.shape[0] == 0 or points.shape[0] == 0:
        return (torch.zeros(0, dtype=torch.int32, device=dev), torch.zeros(0, dtype=torch.float32, device=dev),
                torch.zeros(queries.shape[0] + 1, dtype=torch.int32, device=dev))
    if points.is_cuda:
        return _radius_search_grid(points, queries, float(radius))
    return _radius_search_cdist(points, queries, float(radius), chunk_size)


@torch.no_grad()
def batched_radius_search(ref_positions: Tensor, ref_offsets: Tensor, query_positions: Tensor, query_offsets: Tensor,
                          radius: float, grid_dim=None) -> Tuple[Tensor, Tensor, Tensor]:
    """Per batch element ``radius_search``; indices are global reference rows (int64), splits [M+1] int64."""
    B = len(ref_offsets) - 1
    assert B == len(query_offsets) - 1
    assert int(ref_offsets[-1]) == ref_positions.shape[0], f"Last offset {int(ref_offsets[-1])} != {ref_positions.shape[0]}"
    assert int(query_offsets[-1]) == query_positions.shape[0], f"Last offset {int(query_offsets[-1])} != {query_positions.shape[0]}"
    idx, dist, splits = [], [], []
    base = 0
    for b in range(B):
        r0, r1 = int(ref_offsets[b]), int(ref_offsets[b + 1])
        q0, q1 = int(query_offsets[b]), int(query_offsets[b + 1])
        i, d, s = radius_search(ref_positions[r0:r1].contiguous(), query_positions[q0:q1].contiguous(), radius, grid_dim)
        idx.append(i.long() + r0)
        dist.append(d)
        splits.append((s if b == B - 1 else s[:-1]).long() + base)
        base += i.shape[0]
    return torch.cat(idx), torch.cat(dist), torch.cat(splits)
