"""Batch-index <-> offsets conversions.

Reference: `warpconvnet/geometry/coords/ops/batch_index.py:90-148` (``batch_indexed_coordinates``,
``offsets_from_batch_index``, ``batch_index_from_offset``).  The reference launches a small CUDA
kernel for offsets -> batch index; here it is a host-side ``repeat_interleave`` on the (tiny, CPU)
offsets followed by one H2D copy, because offsets live on the host by contract.
"""
from typing import Optional

import torch
from torch import Tensor


@torch.no_grad()
def batch_index_from_offset(offsets: Tensor, device=None) -> Tensor:
    """offsets [B+1] -> int32 batch index per row [N], produced ON ``device`` (only the B counts cross the bus;
    ``output_size`` is known from the host offsets, so there is no device->host sync)."""
    off = offsets.detach().cpu().to(torch.int64)
    counts = off[1:] - off[:-1]
    n = int(off[-1] - off[0])
    dev = torch.device(device) if device is not None else torch.device("cpu")
    if len(counts) == 1:
        return torch.zeros(n, dtype=torch.int32, device=dev)
    return torch.repeat_interleave(torch.arange(len(counts), dtype=torch.int32, device=dev), counts.to(dev), output_size=n)


@torch.no_grad()
def batch_indexed_coordinates(batched_coords: Tensor, offsets: Tensor) -> Tensor:
    """[N, D] + offsets -> [N, D+1] with the batch index in column 0 (same dtype/device)."""
    if batched_coords.is_cuda and batched_coords.dtype == torch.int32 and batched_coords.ndim == 2:
        from warpconvnet_amd import _lib  # one HIP launch (wcn_batch_indexed_coords) instead of fill + copy kernels

        c = batched_coords.contiguous()
        n, d = c.shape
        B = offsets.numel() - 1
        out = torch.empty((n, d + 1), dtype=torch.int32, device=c.device)
        off_dev = offsets.to(device=c.device, dtype=torch.int32) if B > 1 else None
        _lib.check(_lib.lib().wcn_batch_indexed_coords(_lib.ptr(c), n, d, _lib.ptr(off_dev), max(B, 1), _lib.ptr(out),
                                                       _lib.stream_handle(c.device)), "wcn_batch_indexed_coords")
        return out
    bidx = batch_index_from_offset(offsets, device=batched_coords.device).to(batched_coords.dtype)
    return torch.cat([bidx.unsqueeze(1), batched_coords], dim=1)


@torch.no_grad()
def offsets_from_batch_index(batch_index: Tensor, num_batches: Optional[int] = None) -> Tensor:
    """Sorted batch index [N] -> CPU offsets [B+1] (trailing empty batches kept if num_batches given)."""
    counts = torch.bincount(batch_index.to(torch.int64), minlength=num_batches or 0).cpu()
    return torch.cat([torch.zeros(1, dtype=torch.int64), counts.cumsum(0)]).to(torch.int32)


@torch.no_grad()
def offsets_from_offsets(offsets: Tensor, sorted_indices: Tensor, device=None) -> Tensor:
    """Offsets of the sub-batch selected by (batch-sorted) row ``sorted_indices``."""
    B = offsets.shape[0] - 1
    if B == 1:
        return torch.tensor([0, len(sorted_indices)], dtype=torch.int32)
    bidx = batch_index_from_offset(offsets, device=sorted_indices.device)
    return offsets_from_batch_index(bidx[sorted_indices.long()], num_batches=B)
