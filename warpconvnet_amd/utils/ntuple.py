"""Tuple-broadcast helpers (mirrors reference `warpconvnet/utils/ntuple.py:10-50`)."""
from typing import Any, Sequence, Tuple, Union

import torch


def ntuple(x: Union[int, Sequence[int], torch.Tensor], ndim: int) -> Tuple[int, ...]:
    """Broadcast an int (or validate a sequence) to an ``ndim``-tuple of ints."""
    if type(x) is tuple:  # the per-layer case (module attributes): one dict lookup
        hit = _NTUPLES.get((x, ndim))
        if hit is not None:
            return hit
    if isinstance(x, torch.Tensor):
        x = [int(v) for v in x.reshape(-1).tolist()]
    if isinstance(x, int):
        return (x,) * ndim
    out = tuple(int(v) for v in x)
    assert len(out) == ndim, f"expected {ndim} values, got {out}"
    if type(x) is tuple and len(_NTUPLES) < 4096:
        _NTUPLES[(x, ndim)] = out
    return out


_NTUPLES = {}


def _pad_values(number_of_outputs: int, *values: Any) -> Tuple[Any, ...]:
    """Left-align ``values`` in a tuple of ``number_of_outputs`` slots, rest ``None``."""
    assert number_of_outputs >= len(values) >= 0
    return tuple(values) + (None,) * (number_of_outputs - len(values))


def _pad_tuple(x: Any, y: Any, number_of_outputs: int) -> Tuple[Any, ...]:
    return _pad_values(number_of_outputs, x, y)


_DEVICE_CONSTS = {}


def device_const_i32(values: Sequence[int], device: torch.device) -> torch.Tensor:
    """Small int32 constant vector on ``device``, built once per (values, device): ``torch.tensor(list, device=gpu)`` is a
    synchronous pageable host-to-device copy (~0.3 ms) - five of them per MinkUNet forward before this cache."""
    key = (tuple(int(v) for v in values), str(device))
    t = _DEVICE_CONSTS.get(key)
    if t is None:
        t = _DEVICE_CONSTS[key] = torch.tensor(key[0], dtype=torch.int32, device=device)
    return t
