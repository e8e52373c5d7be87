"""List -> concatenated tensor + CPU offsets (reference `geometry/utils/list_to_batch.py:11-28`)."""
from typing import List, Tuple

import torch
from torch import Tensor


def list_to_cat_tensor(tensor_list: List[Tensor]) -> Tuple[Tensor, Tensor, int]:
    """Returns ``(cat, offsets[B+1] int32 on CPU, B)``."""
    sizes = torch.tensor([0] + [int(t.shape[0]) for t in tensor_list], dtype=torch.int64)
    offsets = sizes.cumsum(0).to(torch.int32)
    return torch.cat(list(tensor_list), dim=0), offsets, len(tensor_list)
