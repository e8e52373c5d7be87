"""Base classes of modules that consume ``Geometry`` objects (reference `nn/modules/base_module.py:12-45`)."""
import torch.nn as nn

from warpconvnet_amd.geometry.base.geometry import Geometry


class BaseSpatialModule(nn.Module):
    @property
    def device(self):
        return next(self.parameters()).device

    def forward(self, x: Geometry):
        raise NotImplementedError


class BaseSpatialModel(BaseSpatialModule):
    pass
