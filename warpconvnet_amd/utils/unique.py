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
