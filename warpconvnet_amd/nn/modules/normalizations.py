"""Normalisation modules over ``Geometry`` features (reference `nn/modules/normalizations.py:30-68`)."""
from typing import Union

import torch.nn as nn
from torch import Tensor

from warpconvnet_amd.geometry.base.geometry import Geometry
from warpconvnet_amd.nn.functional.normalizations import batch_norm_module_forward, hip_batch_norm_supported
from warpconvnet_amd.nn.modules.base_module import BaseSpatialModule


def apply_batch_norm(norm: nn.modules.batchnorm._BatchNorm, x: Tensor, relu: bool = False) -> Tensor:
    """BatchNorm (+ ReLU) on a feature tensor: HIP kernels for 2-D GPU tensors, the framework's module otherwise."""
    if hip_batch_norm_supported(x) and type(norm) in (nn.BatchNorm1d,):
        return batch_norm_module_forward(norm, x, relu)
    y = norm(x)
    return nn.functional.relu(y) if relu else y


class NormalizationBase(BaseSpatialModule):
    """Applies a normalisation module to the feature tensor of a geometry."""

    def __init__(self, norm: nn.Module):
        super().__init__()
        self.norm = norm

    def __repr__(self):
        return f"{self.__class__.__name__}({self.norm})"

    def forward(self, input: Union[Geometry, Tensor]):
        feats = input.feature_tensor if isinstance(input, Geometry) else input
        out = apply_batch_norm(self.norm, feats) if isinstance(self.norm, nn.BatchNorm1d) else self.norm(feats)
        return input.replace(batched_features=out) if isinstance(input, Geometry) else out


class BatchNorm(NormalizationBase):
    """``torch.nn.BatchNorm1d`` over ``Geometry`` features (same parameters / state dict as the reference's module)."""

    def __init__(self, num_features: int, eps: float = 1e-5, momentum: float = 0.1):
        super().__init__(nn.BatchNorm1d(num_features, eps=eps, momentum=momentum))
