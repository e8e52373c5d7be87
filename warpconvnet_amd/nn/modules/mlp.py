"""``MLPBlock``: Linear - LayerNorm - activation - Linear - LayerNorm plus a (projected) shortcut
(reference `warpconvnet/nn/modules/mlp.py:124-177`).  On fp32 CUDA features whose widths fit (in <= 64, hidden <= 128, out <= 64)
the whole block runs as one HIP kernel per direction (the PointConv edge kernel with one row per "edge"); otherwise the
linears are library GEMMs."""
from typing import Union

import torch.nn as nn
from torch import Tensor

from warpconvnet_amd.geometry.base.geometry import Geometry
from warpconvnet_amd.nn.functional.point_conv import fused_mlp_block, fused_mlp_block_supported
from warpconvnet_amd.nn.modules.base_module import BaseSpatialModule


class MLPBlock(BaseSpatialModule):
    def __init__(self, in_channels: int, out_channels: int = None, hidden_channels: int = None, activation=nn.ReLU,
                 bias: bool = True):
        super().__init__()
        hidden_channels = in_channels if hidden_channels is None else hidden_channels
        out_channels = in_channels if out_channels is None else out_channels
        self.in_channels = in_channels
        self.block = nn.Sequential(
            nn.Linear(in_channels, hidden_channels, bias=bias),
            nn.LayerNorm(hidden_channels),
            activation(),
            nn.Linear(hidden_channels, out_channels, bias=bias),
            nn.LayerNorm(out_channels),
        )
        self.shortcut = nn.Linear(in_channels, out_channels, bias=bias) if in_channels != out_channels else nn.Identity()

    def _forward_feature(self, x: Tensor) -> Tensor:
        if x.is_cuda and fused_mlp_block_supported(self, x):  # one HIP kernel per direction (csrc/pointconv.hip, k = 1)
            return fused_mlp_block(self, x)
        return self.block(x) + self.shortcut(x)

    def forward(self, x: Union[Tensor, Geometry]):
        if isinstance(x, Geometry):
            return x.replace(batched_features=self._forward_feature(x.feature_tensor))
        return self._forward_feature(x)
