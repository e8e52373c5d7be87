"""dtype helpers (mirrors reference `warpconvnet/utils/type_cast.py`)."""
from typing import Optional

import torch

TYPE_ORDER = [torch.bfloat16, torch.float16, torch.float32, torch.float64]


def _as_dtype(d):
    return d.dtype if isinstance(d, torch.Tensor) else d


def _min_dtype(*dtypes):
    ds = [_as_dtype(d) for d in dtypes]
    assert all(d in TYPE_ORDER for d in ds), f"Invalid dtype: {ds}"
    return TYPE_ORDER[min(TYPE_ORDER.index(d) for d in ds)]


def _max_dtype(*dtypes):
    ds = [_as_dtype(d) for d in dtypes]
    assert all(d in TYPE_ORDER for d in ds), f"Invalid dtype: {ds}"
    return TYPE_ORDER[max(TYPE_ORDER.index(d) for d in ds)]


def _maybe_cast(tensor: torch.Tensor, dtype: Optional[torch.dtype] = None) -> torch.Tensor:
    """Cast only when a dtype is given and differs."""
    if dtype is None or tensor.dtype == dtype:
        return tensor
    return tensor.to(dtype=dtype)
