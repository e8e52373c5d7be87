"""``Sequential`` that threads Geometry objects through spatial and plain tensor layers.

Reference: `warpconvnet/nn/modules/sequential.py:14-62`: a plain ``nn.Module`` receives the feature tensor,
a spatial module receives the geometry; consecutive plain layers do not rebuild the geometry in between.
"""
import torch
import torch.nn as nn

from warpconvnet_amd.geometry.base.geometry import Geometry

from .base_module import BaseSpatialModule


class Sequential(nn.Sequential, BaseSpatialModule):
    def forward(self, x: Geometry):
        assert isinstance(x, Geometry), f"Expected a Geometry, got {type(x)}"
        carrier = x  # last geometry seen: supplies coordinates when a tensor re-enters a spatial layer
        for module in self:
            spatial = isinstance(module, BaseSpatialModule)
            if isinstance(x, Geometry):
                if spatial:
                    x = module(x)
                else:
                    carrier, x = x, module(x.feature_tensor)
            else:
                x = module(carrier.replace(batched_features=x)) if spatial else module(x)
        if isinstance(x, torch.Tensor):
            x = carrier.replace(batched_features=x)
        return x
