"""Coordinate de-duplication.

Reference: `warpconvnet/utils/unique.py:124-143` (``unique_hashmap``: hash-insert then
``torch.unique`` of the winner indices).  On the GPU this build uses its own HIP hash table
(insert keeps the smallest row index per key, so the result is deterministic); on the CPU it
uses a lexicographic sort.  Both return the ascending row indices of first occurrences.
"""
from typing import Tuple

import torch
from torch import Tensor


@torch.no_grad()
def unique_first_indices(bcoords: Tensor) -> Tensor:
    """bcoords [N, 4|3] int -> int64 ascending indices of the first occurrence of every distinct row."""
    if bcoords.shape[0] == 0:
        return torch.zeros(0, dtype=torch.int64, device=bcoords.device)
    if bcoords.is_cuda:
        return unique_hashmap(bcoords)[0]
    _, inverse = torch.unique(bcoords, dim=0, return_inverse=True)
    n = bcoords.shape[0]
    first = torch.full((int(inverse.max()) + 1,), n, dtype=torch.int64)
    first.scatter_reduce_(0, inverse, torch.arange(n, dtype=torch.int64), reduce="amin")
    return torch.sort(first).values


@torch.no_grad()
def unique_hashmap(bcoords: Tensor, **kwargs) -> Tuple[Tensor, "PackedHashTable"]:  # noqa: F821
    """GPU path: returns ``(unique_indices int64 ascending, table)`` (reference signature)."""
    from warpconvnet_amd.geometry.coords.search.packed_hashmap import PackedHashTable

    assert bcoords.is_cuda, f"Batched coordinates must be on a GPU device, got {bcoords.device}"
    if bcoords.shape[1] == 3:
        bcoords = torch.nn.functional.pad(bcoords, (0, 1), value=0)
    table = PackedHashTable.from_coords(bcoords, device=bcoords.device)
    return table.unique_index, table


_ARANGE = {}


def arange_i32(n: int, dev) -> Tensor:
    """``torch.arange(n, int32)`` on ``dev``, kept per size (read-only by contract): identity pair lists and first-occurrence
    tests ask for the same few sizes every iteration."""
    key = (int(n), str(dev))
    t = _ARANGE.get(key)
    if t is None:
        if len(_ARANGE) >= 32:
            _ARANGE.clear()
        t = _ARANGE[key] = torch.arange(n, dtype=torch.int32, device=dev)
    return t


@torch.no_grad()
def unique_first_indices_with_offsets(bcoords: Tensor) -> Tuple[Tensor, Tensor]:
    """GPU: ``(ascending int64 rows of the first occurrence of every distinct [b, x, y, z] row, CPU int32 offsets [B+1] of
    the surviving rows per batch index)`` with ONE host read.

    The generic route (`unique_hashmap` -> `PackedHashTable.insert` -> `unique_index` -> `offsets_from_batch_index`) reads
    the status word, the number of survivors and the per-batch counts back separately - three queue-draining waits per
    strided convolution; here the status word and a 512-bin per-batch histogram of the survivors travel together and the
    index list is compacted with a known size (`nonzero_static`)."""
    from warpconvnet_amd.geometry.coords.search.packed_hashmap import PackedHashTable

    assert bcoords.is_cuda and bcoords.ndim == 2 and bcoords.shape[1] == 4
    coords = bcoords.contiguous().to(torch.int32)
    n, dev = coords.shape[0], coords.device
    table = PackedHashTable(max(16, 2 * n), device=dev)
    meta = torch.zeros(1 + PackedHashTable.BATCH_MAX + 1, dtype=torch.int32, device=dev)  # [status, counts[512]]
    table._launch_insert(coords, meta[:1])
    first = table.search(coords) == arange_i32(n, dev)
    b = coords[:, 0].long().clamp_(0, PackedHashTable.BATCH_MAX)  # out-of-range batch ids are reported through the status
    meta[1:].index_add_(0, b, first.to(torch.int32))
    host = meta.cpu()  # the one host read
    PackedHashTable.raise_for_flags(int(host[0]), n, table.capacity)
    counts = host[1:]
    nz = torch.nonzero(counts)
    num_batches = int(nz[-1]) + 1 if len(nz) else 0
    offsets = torch.cat([torch.zeros(1, dtype=torch.int64), counts[:num_batches].to(torch.int64).cumsum(0)]).to(torch.int32)
    idx = torch.nonzero_static(first, size=int(offsets[-1])).squeeze(1)
    return idx, offsets
