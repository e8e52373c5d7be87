"""Sinusoidal positional encodings (reference `warpconvnet/nn/functional/encodings.py:12-75`)."""
from typing import Optional

import numpy as np
import torch
from torch import Tensor


def get_freqs(num_freqs: int, data_range: float = 2.0, device: Optional[torch.device] = None) -> Tensor:
    """``2*pi/data_range * 2^i`` for ``i < num_freqs``."""
    freqs = 2 ** torch.arange(start=0, end=num_freqs, device=device or torch.device("cpu"))
    return (2 * np.pi / data_range) * freqs


def sinusoidal_encoding(x: Tensor, num_channels: Optional[int] = None, data_range: Optional[float] = None,
                        encoding_axis: int = -1, freqs: Optional[Tensor] = None, concat_input: bool = False) -> Tensor:
    """[..., C] -> [..., C * num_channels] (``cos`` block, ``sin`` block, optionally the input itself, per channel)."""
    assert encoding_axis == -1, "Only encoding_axis=-1 is supported at the moment"
    x = x.unsqueeze(encoding_axis)
    if freqs is None:
        assert num_channels is not None and data_range is not None, "num_channels and data_range must be provided if freqs are not given"
        assert num_channels % 2 == 0, f"num_channels must be even for sin/cos, got {num_channels}"
        freqs = get_freqs(num_channels // 2, data_range, device=x.device)
    freqs = freqs.reshape((1,) * (len(x.shape) - 1) + freqs.shape)
    fx = x * freqs
    parts = [fx.cos(), fx.sin()] + ([x] if concat_input else [])
    return torch.cat(parts, dim=encoding_axis).flatten(start_dim=-2)
