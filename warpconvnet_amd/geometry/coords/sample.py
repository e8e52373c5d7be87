"""Per-batch random row sampling (reference `warpconvnet/geometry/coords/sample.py:11-31`)."""
from typing import Tuple

import torch
from torch import Tensor


def random_sample_per_batch(offsets: Tensor, num_samples: int) -> Tuple[Tensor, Tensor]:
    """``num_samples`` row indices per batch element, drawn uniformly WITH replacement, and the offsets of the sampled
    batch (``arange(B + 1) * num_samples``)."""
    offsets = offsets.cpu()
    counts = offsets.diff()
    draws = torch.floor(torch.rand(len(counts), num_samples) * counts.view(-1, 1)).to(torch.int32)
    indices = (draws + offsets[:-1].view(-1, 1).to(torch.int32)).view(-1)
    return indices, torch.arange(len(counts) + 1) * num_samples
