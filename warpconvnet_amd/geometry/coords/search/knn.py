"""k-nearest-neighbour search.

GPU tensors: exact grid search in HIP (``wcn_knn_grid``, `csrc/points.hip`): the reference points of a batch element
are binned into a uniform grid sized for a few points per cell (device-side sort by cell id + CSR over cells), then one
thread per query walks the cell shells around its cell and stops when the k-th distance is inside the searched cube -
O(M*k) instead of the reference's O(M*N) chunked ``cdist`` + ``topk``
(`warpconvnet/geometry/coords/search/knn.py:11-26, 108-142`), same neighbours (ties may come back in another order).
CPU tensors: the reference's own algorithm (cdist + topk), which is also the oracle of the GPU tests.
Indices are global row ids (local index + batch offset).
"""
import ctypes
import math

import torch
from torch import Tensor

from warpconvnet_amd import _lib

_MAX_CELLS = 1 << 24


def _knn_cdist(ref: Tensor, query: Tensor, k: int, chunk: int = 4096) -> Tensor:
    out = []
    for s in range(0, query.shape[0], chunk):
        d = torch.cdist(query[s : s + chunk], ref)
        out.append(torch.topk(d, k, dim=1, largest=False).indices)
    return torch.cat(out, 0) if out else torch.zeros((0, k), dtype=torch.int64, device=ref.device)


def build_cell_grid(ref32: Tensor, cell_size: float = None, min_cell_size: float = 0.0):
    """Bin ``ref32`` [N, 3] fp32 into a uniform grid over its bounding box: (ref_sorted, ref_ids, cell_start, lo, h, dims).

    ``cell_size=None`` sizes the cells for a few points each (kNN); otherwise the cell is at least ``cell_size`` (radius
    search needs cell >= radius).  Cells grow until the grid fits ``_MAX_CELLS``.  One host read (the bounding box)."""
    dev = ref32.device
    n = ref32.shape[0]
    # bounding box: reductions along the long axis of a [3, N] copy (a column reduction over [N, 3] runs at a fraction of
    # the bandwidth: 0.14 + 0.09 ms for 200 k points), both bounds in one host read
    lo_t, hi_t = torch.aminmax(ref32.t().contiguous(), dim=1)
    lo_hi = torch.stack([lo_t, hi_t]).cpu().tolist()
    lo, hi = lo_hi[0], lo_hi[1]
    ext = [max(h - l, 1e-6) for l, h in zip(lo, hi)]
    if cell_size is None:
        # a few points per cell on average: shell 1 (27 cells) then usually holds k <= 32 candidates
        h = (4.0 * ext[0] * ext[1] * ext[2] / max(n, 1)) ** (1.0 / 3.0)
        h = max(h, max(ext) / 1024.0, 1e-6)
    else:
        h = max(float(cell_size), min_cell_size, 1e-6)
    dims = [int(math.floor(e / h)) + 1 for e in ext]
    while dims[0] * dims[1] * dims[2] > _MAX_CELLS:
        h *= 1.5
        dims = [int(math.floor(e / h)) + 1 for e in ext]
    origin = torch.tensor(lo, device=dev, dtype=torch.float32)
    cell3 = torch.floor((ref32 - origin) / h).to(torch.int64)
    for a in range(3):
        cell3[:, a].clamp_(0, dims[a] - 1)
    cell = (cell3[:, 2] * dims[1] + cell3[:, 1]) * dims[0] + cell3[:, 0]
    ncells = dims[0] * dims[1] * dims[2]
    if ref32.is_cuda:
        # in-house radix argsort on the (< 2^24) cell ids instead of the framework's merge sort of int64 keys
        from warpconvnet_amd.geometry.coords.search.torch_discrete import argsort_u32_ascending

        order = argsort_u32_ascending(cell, max(1, (ncells - 1).bit_length())).long()
    else:
        order = torch.argsort(cell, stable=True)
    sorted_cell = cell[order]
    cell_start = torch.searchsorted(sorted_cell, torch.arange(ncells + 1, device=dev, dtype=torch.int64)).to(torch.int32)
    return ref32[order].contiguous(), order.to(torch.int32).contiguous(), cell_start, lo, h, dims


def _knn_grid(ref: Tensor, query: Tensor, k: int, return_dist2: bool = False):
    dev = ref.device
    ref32, q32 = ref.float().contiguous(), query.float().contiguous()
    ref_sorted, ref_ids, cell_start, lo, h, dims = build_cell_grid(ref32)
    m = q32.shape[0]
    out = torch.empty((m, k), dtype=torch.int64, device=dev)
    d2 = torch.empty((m, k), dtype=torch.float32, device=dev) if return_dist2 else None
    _lib.check(
        _lib.lib().wcn_knn_grid(_lib.ptr(ref_sorted), _lib.ptr(ref_ids), _lib.ptr(cell_start), (ctypes.c_float * 3)(*lo),
                                ctypes.c_float(h), _lib.i3(dims), _lib.ptr(q32), m, k, _lib.ptr(out), _lib.ptr(d2),
                                _lib.stream_handle(dev)),
        "wcn_knn_grid",
    )
    return (out, d2) if return_dist2 else out


@torch.no_grad()
def knn_search(ref: Tensor, query: Tensor, k: int, chunk: int = 4096) -> Tensor:
    """[M, k] int64 indices into ``ref`` of the k nearest reference points of every query, ascending by distance."""
    assert k <= ref.shape[0], f"knn_k={k} exceeds the number of reference points {ref.shape[0]}"
    if ref.is_cuda and k <= 64:
        return _knn_grid(ref, query, k)
    return _knn_cdist(ref, query, k, chunk)


@torch.no_grad()
def batched_knn_search(ref: Tensor, ref_offsets: Tensor, query: Tensor, query_offsets: Tensor, k: int,
                       chunk: int = 4096) -> Tensor:
    assert len(ref_offsets) == len(query_offsets)
    outs = []
    for b in range(len(ref_offsets) - 1):
        r0, r1 = int(ref_offsets[b]), int(ref_offsets[b + 1])
        q0, q1 = int(query_offsets[b]), int(query_offsets[b + 1])
        outs.append(knn_search(ref[r0:r1], query[q0:q1], k, chunk) + r0)
    return torch.cat(outs, 0)
