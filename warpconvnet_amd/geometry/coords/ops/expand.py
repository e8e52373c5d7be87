"""Coordinate expansion for generative convolution: the output set is the input set plus every kernel offset of it.

Reference: `warpconvnet/geometry/coords/ops/expand.py:17-75` (hash table seeded with the inputs, offsets inserted in
batches with table re-growth, then an UNSTABLE argsort by batch).  Here the candidates ``coords + offset_k`` are
de-duplicated with the build's own HIP hash table (insert keeps the smallest candidate row per key), so the surviving
rows are the first occurrences in the order [inputs, offset 0 of every input, offset 1 ...]; inputs are batch-sorted and
a stable batch sort keeps that order inside every batch - the output order is deterministic.
"""
from typing import Optional, Tuple

import numpy as np
import torch
from torch import Tensor

from warpconvnet_amd.geometry.coords.ops.batch_index import offsets_from_batch_index
from warpconvnet_amd.utils.ntuple import ntuple
from warpconvnet_amd.utils.unique import unique_first_indices


@torch.no_grad()
def expand_coords(batch_indexed_coords: Tensor, kernel_size: Tuple[int, ...], kernel_dilation: Tuple[int, ...],
                  kernel_batch: Optional[int] = None) -> Tuple[Tensor, Tensor]:
    """[N, D+1] -> (unique union of coords and coords + kernel offsets [M, D+1] batch-sorted, CPU offsets [B+1]).

    ``kernel_batch`` bounds the number of offsets materialised at once (memory: N * kernel_batch candidate rows).
    """
    from warpconvnet_amd.geometry.coords.search.torch_discrete import kernel_offsets_from_size

    if batch_indexed_coords.device.type != "cuda":
        raise ValueError(f"expand_coords requires GPU tensors (HIP path, no CPU fallback), got {batch_indexed_coords.device}")
    nd = batch_indexed_coords.shape[1] - 1
    kernel_size = ntuple(kernel_size, nd)
    kernel_dilation = ntuple(kernel_dilation, nd)
    coords = batch_indexed_coords.to(torch.int32).contiguous()
    K = int(np.prod(kernel_size))
    if kernel_batch is None:
        kernel_batch = max(1, K // kernel_size[0])
    kernel_batch = max(1, min(int(kernel_batch), K))
    offsets = kernel_offsets_from_size(kernel_size, kernel_dilation, device=coords.device).to(torch.int32)  # [K, D+1]
    current = coords
    for k0 in range(0, K, kernel_batch):
        off = offsets[k0 : k0 + kernel_batch]
        cand = (coords.unsqueeze(0) + off.unsqueeze(1)).reshape(-1, nd + 1)  # offset-major: all rows of offset k0, ...
        merged = torch.cat([current, cand], 0)
        current = merged[unique_first_indices(merged)].contiguous()
    order = torch.sort(current[:, 0], stable=True).indices
    out = current[order].contiguous()
    return out, offsets_from_batch_index(out[:, 0])
