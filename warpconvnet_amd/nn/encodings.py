"""Encoding modules (reference `warpconvnet/nn/encodings.py:33-62`)."""
import torch.nn as nn
from torch import Tensor

from warpconvnet_amd.nn.functional.encodings import get_freqs, sinusoidal_encoding


class SinusoidalEncoding(nn.Module):
    def __init__(self, num_channels: int, data_range: float = 2.0, concat_input: bool = True):
        super().__init__()
        assert num_channels % 2 == 0, f"num_channels must be even for sin/cos, got {num_channels}"
        self.num_channels = num_channels
        self.concat_input = concat_input
        self.register_buffer("freqs", get_freqs(num_channels // 2, data_range))

    def num_output_channels(self, num_input_channels: int) -> int:
        return (num_input_channels + 1) * self.num_channels if self.concat_input else num_input_channels * self.num_channels

    def forward(self, x: Tensor) -> Tensor:
        return sinusoidal_encoding(x, freqs=self.freqs, concat_input=self.concat_input)
