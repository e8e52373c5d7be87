"""``Sequential`` that threads Geometry objects through spatial and plain tensor layers.

Reference: `warpconvnet/nn/modules/sequential.py:14-62`: a plain ``nn.Module`` receives the feature tensor,
a spatial module receives the geometry; consecutive plain layers do not rebuild the geometry in between.
"""
import torch
import torch.nn as nn

from warpconvnet_amd.geometry.base.geometry import Geometry

from warpconvnet_amd.nn.functional.normalizations import batch_norm_module_forward, hip_batch_norm_supported

from warpconvnet_amd.nn.functional.sparse_conv.block import conv_bn_act

from .base_module import BaseSpatialModule
from .sparse_conv import SpatiallySparseConv


def _has_hooks(module: nn.Module) -> bool:
    """The fused BatchNorm(+ReLU) path calls the kernels, not ``module(x)``: a module that carries forward hooks (activation
    capture, profilers, quantisation observers) - or any global module hook - runs the ordinary way so the hooks fire."""
    import torch.nn.modules.module as _m

    return bool(module._forward_hooks or module._forward_pre_hooks or module._backward_hooks or module._backward_pre_hooks
                or _m._global_forward_hooks or _m._global_forward_pre_hooks or _m._global_backward_hooks
                or _m._global_backward_pre_hooks)


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
            if (spatial and i < len(mods) and type(mods[i]) is nn.BatchNorm1d and isinstance(x, Geometry)
                    and isinstance(module, SpatiallySparseConv) and not _has_hooks(module) and not _has_hooks(mods[i])):
                # SparseConv3d -> BatchNorm1d (-> ReLU): one autograd node, direct launches (functional/sparse_conv/block.py);
                # None = a layer the fused node does not take, the modules then run one by one below
                nxt = mods[i + 1] if i + 1 < len(mods) else None
                fuse_relu = type(nxt) is nn.ReLU and not _has_hooks(nxt)
                y = conv_bn_act(x, module, mods[i], fuse_relu)
                if y is not None:
                    x = carrier = y
                    i += 1 + int(fuse_relu)
                    continue
            if not spatial and type(module) is nn.BatchNorm1d and not _has_hooks(module):
                # BatchNorm1d on the feature tensor (and the ReLU behind it, the ConvBlock pattern of the reference's
                # models) goes through the HIP kernels: same function, state and gradients (functional/normalizations.py)
                feats = x.feature_tensor if isinstance(x, Geometry) else x
                if hip_batch_norm_supported(feats) and feats.is_floating_point():
                    fuse_relu = i < len(mods) and type(mods[i]) is nn.ReLU and not _has_hooks(mods[i])
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
