"""Coordinate down-sampling for strided convolution.

Reference: `warpconvnet/geometry/coords/ops/stride.py:18-56` (floor-divide, hash de-duplicate,
``torch.unique`` of winner indices, UNSTABLE argsort by batch).  Here the surviving rows are the first
occurrences in input order; because inputs are batch-sorted, so are the outputs - no argsort, and the
output row order is deterministic.  A Morton ``order`` re-sorts the survivors by (batch, z-order code).
"""
from typing import Tuple

import torch
from torch import Tensor

from warpconvnet_amd.geometry.coords.ops.batch_index import offsets_from_batch_index
from warpconvnet_amd.utils.ntuple import device_const_i32, ntuple
from warpconvnet_amd.utils.unique import unique_first_indices, unique_first_indices_with_offsets


def _stride_coords_from_cells(bcoords: Tensor, stride: Tuple[int, ...], num_batches: int, with_map: bool):
    """Down-sampling through the cell table the submanifold layers of this level built (`csrc/kmap_stride.hip`): one block
    lookup + a few cell reads per voxel, four launches and ONE host read (the output offsets) - instead of a hash insert of
    every candidate, a search, and a chain of torch ops with three reads.  -> (out [M, 4], CPU offsets [B + 1])"""
    import ctypes

    from warpconvnet_amd import _lib

    from warpconvnet_amd.geometry.coords.search.cell_handle import attach_cells, attach_stride_map, cells_of
    from warpconvnet_amd.geometry.coords.search.packed_hashmap import PackedHashTable
    from warpconvnet_amd.geometry.coords.search.torch_discrete import default_hints

    n, dev = bcoords.shape[0], bcoords.device
    L = _lib.lib()
    stream = _lib.stream_handle(dev)
    st = _lib.i3(stride)
    ntile = int(L.wcn_cells_stride_tiles(n))
    flags = torch.empty(max(1, (n + 63) // 64), dtype=torch.int64, device=dev)
    counts = torch.empty(ntile + 1, dtype=torch.int32, device=dev)
    meta = torch.zeros(num_batches + 2, dtype=torch.int32, device=dev)  # [out_offsets (B + 1) | status of a table built here]
    cells = cells_of(bcoords)  # None once the tensor has been edited in place since the build
    hints = default_hints()
    while True:
        built = cells is None
        if built:
            # no submanifold layer has run on these coordinates (a network whose first spatial layer is strided): the table
            # alone, strict insert - four launches instead of a hash insert of every candidate
            max_blocks = hints.max_blocks(n)
            ws_bytes = L.wcn_kmap_binned_workspace(n, max_blocks)
            ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
            scratch = torch.empty(n, dtype=torch.int32, device=dev)
            _lib.check(L.wcn_kmap_cells_build(_lib.ptr(bcoords), n, max_blocks, _lib.ptr(ws), ws_bytes, _lib.ptr(scratch),
                                              _lib.ptr(meta[num_batches + 1 :]), stream), "wcn_kmap_cells_build")
            cells = (ws, n, max_blocks)
        ws, _, max_blocks = cells
        _lib.check(L.wcn_cells_stride_count(_lib.ptr(ws), n, max_blocks, _lib.ptr(bcoords), st, _lib.ptr(flags), _lib.ptr(counts),
                                            num_batches, _lib.ptr(meta), stream), "wcn_cells_stride_count")
        host = meta.cpu()  # the one host read: output offsets (+ the status of a table built here)
        status = int(host[-1])
        if built and (status & _lib.WCN_FLAG_TABLE_FULL) and max_blocks < n:
            hints.div = 4 if hints.div > 4 else 1  # sparser scenes than the bound assumed: larger table, remembered
            cells = None
            meta.zero_()
            continue
        PackedHashTable.raise_for_flags(status, n, 2 * n)
        break
    if built:
        attach_cells(bcoords, *cells)  # complete and validated: later layers on this tensor reuse it
    offsets = host[: num_batches + 1].clone()
    m = int(offsets[-1])
    out = torch.empty((m, 4), dtype=torch.int32, device=dev)
    K = int(stride[0]) * int(stride[1]) * int(stride[2])
    nbr = mask = None
    if with_map and K <= 32 and m > 0:
        nbr = torch.empty((m, int(L.wcn_kmap_row_pitch(K))), dtype=torch.int32, device=dev)
        mask = torch.empty((m, 1), dtype=torch.int32, device=dev)
    _lib.check(L.wcn_cells_stride_emit(_lib.ptr(ws), n, max_blocks, _lib.ptr(bcoords), st, _lib.ptr(flags), _lib.ptr(counts),
                                       _lib.ptr(out), None, _lib.ptr(nbr), _lib.ptr(mask), stream), "wcn_cells_stride_emit")
    if nbr is not None:
        # the kernel map of a convolution with kernel_size == stride is exactly these cells: generate_kernel_map picks it up
        attach_stride_map(out, bcoords, stride, nbr, mask)
    return out, offsets


@torch.no_grad()
def stride_coords(batch_indexed_coords: Tensor, stride: Tuple[int, ...], order=None, num_batches: int = None,
                  with_map: bool = False) -> Tuple[Tensor, Tensor]:
    """[N, D+1] -> (unique floor(coords / stride) [M, D+1], CPU offsets [B+1]).

    ``num_batches`` (known by the caller from the input's offsets) enables the cell-table route: the table of an earlier
    submanifold build on the coordinate tensor (`_wcn_cells`, set by `generate_kernel_map`) or, without one, a table built
    here; ``with_map`` additionally emits the kernel map of a convolution whose kernel is the stride window."""
    nd = batch_indexed_coords.shape[1] - 1
    stride = ntuple(stride, nd)
    if all(s == 1 for s in stride):
        return batch_indexed_coords, offsets_from_batch_index(batch_indexed_coords[:, 0])
    from warpconvnet_amd.geometry.coords.ops.serialization import POINT_ORDERING, encode, to_point_ordering

    order = to_point_ordering(order)
    if (num_batches is not None and order == POINT_ORDERING.RANDOM and batch_indexed_coords.is_cuda
            and batch_indexed_coords.shape[1] == 4 and batch_indexed_coords.shape[0] > 0
            and batch_indexed_coords.dtype == torch.int32 and batch_indexed_coords.is_contiguous()):
        from warpconvnet_amd import _lib

        if _lib.lib().wcn_cells_stride_supported(_lib.i3(stride)):
            return _stride_coords_from_cells(batch_indexed_coords, stride, int(num_batches), with_map)
    div = device_const_i32([1, *stride], batch_indexed_coords.device)
    coarse = torch.div(batch_indexed_coords, div, rounding_mode="floor").to(torch.int32)
    if coarse.is_cuda and coarse.shape[1] == 4 and coarse.shape[0] > 0:
        idx, offsets = unique_first_indices_with_offsets(coarse)  # one host read for status, row count and batch counts
        out = coarse[idx].contiguous()
    else:
        idx = unique_first_indices(coarse)
        out = coarse[idx].contiguous()
        offsets = offsets_from_batch_index(out[:, 0])
    if order != POINT_ORDERING.RANDOM:  # (batch, z-order) sort: the batch index sits above the code bits (stride.py:52-54)
        out = out[encode(out, order=order, return_perm=True).perm].contiguous()
    return out, offsets
