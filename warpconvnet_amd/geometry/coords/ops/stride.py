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


@torch.no_grad()
def stride_coords(batch_indexed_coords: Tensor, stride: Tuple[int, ...], order=None) -> Tuple[Tensor, Tensor]:
    """[N, D+1] -> (unique floor(coords / stride) [M, D+1], CPU offsets [B+1])."""
    nd = batch_indexed_coords.shape[1] - 1
    stride = ntuple(stride, nd)
    if all(s == 1 for s in stride):
        return batch_indexed_coords, offsets_from_batch_index(batch_indexed_coords[:, 0])
    div = device_const_i32([1, *stride], batch_indexed_coords.device)
    coarse = torch.div(batch_indexed_coords, div, rounding_mode="floor").to(torch.int32)
    from warpconvnet_amd.geometry.coords.ops.serialization import POINT_ORDERING, encode, to_point_ordering

    order = to_point_ordering(order)
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
