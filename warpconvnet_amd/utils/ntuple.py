"""Tuple-broadcast helpers (mirrors reference `warpconvnet/utils/ntuple.py:10-50`)."""
from typing import Any, Sequence, Tuple, Union

import torch


def ntuple(x: Union[int, Sequence[int], torch.Tensor], ndim: int) -> Tuple[int, ...]:
    """Broadcast an int (or validate a sequence) to an ``ndim``-tuple of ints."""
    if isinstance(x, torch.Tensor):
        x = [int(v) for v in x.reshape(-1).tolist()]
    if isinstance(x, int):
        return (x,) * ndim
    out = tuple(int(v) for v in x)
    assert len(out) == ndim, f"expected {ndim} values, got {out}"
    return out


def _pad_values(number_of_outputs: int, *values: Any) -> Tuple[Any, ...]:
    """Left-align ``values`` in a tuple of ``number_of_outputs`` slots, rest ``None``."""
    assert number_of_outputs >= len(values) >= 0
    return tuple(values) + (None,) * (number_of_outputs - len(values))


def _pad_tuple(x: Any, y: Any, number_of_outputs: int) -> Tuple[Any, ...]:
    return _pad_values(number_of_outputs, x, y)
