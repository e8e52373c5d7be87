"""``Sequential`` that threads Geometry objects through spatial and plain tensor layers.

Reference: `warpconvnet/nn/modules/sequential.py:14-62`: a plain ``nn.Module`` receives the feature tensor,
a spatial module receives the geometry; consecutive plain layers do not rebuild the geometry in between.
"""
import torch
import torch.nn as nn

from warpconvnet_amd.geometry.base.geometry import Geometry

from warpconvnet_amd.nn.functional.normalizations import batch_norm_module_forward, hip_batch_norm_supported

from .base_module import BaseSpatialModule


class Sequential(nn.Sequential, BaseSpatialModule):
    def forward(self, x: Geometry):
        assert isinstance(x, Geometry), f"Expected a Geometry, got {type(x)}"
        carrier = x  # last geometry seen: supplies coordinates when a tensor re-enters a spatial layer
        mods = list(self)
        i = 0
        while i < len(mods):
            module = mods[i]
            i += 1
            spatial = isinstance(module, BaseSpatialModule)
            if not spatial and type(module) is nn.BatchNorm1d:
                # BatchNorm1d on the feature tensor (and the ReLU behind it, the ConvBlock pattern of the reference's
                # models) goes through the HIP kernels: same function, state and gradients (functional/normalizations.py)
                feats = x.feature_tensor if isinstance(x, Geometry) else x
                if hip_batch_norm_supported(feats) and feats.is_floating_point():
                    fuse_relu = i < len(mods) and type(mods[i]) is nn.ReLU
                    if isinstance(x, Geometry):
                        carrier = x
                    x = batch_norm_module_forward(module, feats, relu=fuse_relu)
                    i += int(fuse_relu)
                    continue
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
