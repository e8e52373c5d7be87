"""Voxel de-duplication of (real or integer) coordinates.

Reference: `warpconvnet/geometry/coords/ops/voxel.py:112` (``voxel_downsample_random_indices``): the
reference keeps an unspecified ("random") representative per voxel; this build keeps the FIRST row
of every voxel (smallest row index) so results are deterministic.
"""
from typing import Optional, Tuple

import torch
from torch import Tensor

from warpconvnet_amd.geometry.coords.ops.batch_index import (
    batch_indexed_coordinates,
    offsets_from_batch_index,
)


@torch.no_grad()
def voxel_downsample_random_indices(
    batched_points: Tensor, offsets: Tensor, voxel_size: Optional[float] = None
) -> Tuple[Tensor, Tensor]:
    """Returns ``(unique_row_indices sorted ascending, new CPU offsets)``."""
    from warpconvnet_amd.utils.unique import unique_first_indices

    if voxel_size is not None:
        coords = torch.floor(batched_points / voxel_size).to(torch.int32)
    else:
        coords = batched_points.to(torch.int32)
    bcoords = batch_indexed_coordinates(coords, offsets)
    idx = unique_first_indices(bcoords)
    new_offsets = offsets_from_batch_index(bcoords[idx, 0], num_batches=len(offsets) - 1)
    return idx, new_offsets
