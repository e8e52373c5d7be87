"""``FusedSparseConvBlock``: conv -> BatchNorm (inference) -> (+ residual) -> ReLU as ONE kernel launch."""
from typing import Optional

import torch
import torch.nn as nn

from warpconvnet_amd.geometry.types.voxels import Voxels
from warpconvnet_amd.nn.functional.sparse_conv.fused import fold_batchnorm, fused_sparse_conv_inference
from warpconvnet_amd.nn.modules.base_module import BaseSpatialModule
from warpconvnet_amd.nn.modules.sparse_conv import SpatiallySparseConv


class FusedSparseConvBlock(BaseSpatialModule):
    """Inference-time replacement of ``Sequential(SparseConv3d, BatchNorm1d, ReLU)`` (the reference's ``ConvBlock``,
    `models/mink_unet.py:31-53`) and of the tail of a residual block (``relu(bn(conv(h)) + x)``, `:161-175`): the
    BatchNorm is folded to a per-channel scale / shift and applied, with the optional residual add and ReLU, inside the
    convolution kernel's epilogue.  Shares the parameters of the modules it was built from (no copies)."""

    def __init__(self, conv: SpatiallySparseConv, norm: Optional[nn.modules.batchnorm._BatchNorm] = None, relu: bool = True):
        super().__init__()
        if conv.transposed or conv.generative or conv.groups != 1:
            raise ValueError("FusedSparseConvBlock covers non-transposed, non-generative, ungrouped convolutions")
        self.conv, self.norm, self.relu = conv, norm, relu
        self._folded = None  # (versions of the BatchNorm tensors, scale, shift): refolded only when one of them changes

    def forward(self, x: Voxels, residual: Optional[Voxels] = None) -> Voxels:
        if self.training and self.norm is not None:
            raise RuntimeError("FusedSparseConvBlock folds BatchNorm running statistics: call .eval() first")
        scale = shift = None
        if self.norm is not None:
            n = self.norm
            key = tuple((t.data_ptr(), t._version) if t is not None else None
                        for t in (n.running_mean, n.running_var, n.weight, n.bias)) + (n.eps,)
            if self._folded is None or self._folded[0] != key:
                self._folded = (key, *fold_batchnorm(n))  # ~8 tiny launches: not worth repeating per call
            scale, shift = self._folded[1], self._folded[2]
        c = self.conv
        return fused_sparse_conv_inference(x, c.weight, c.kernel_size, c.stride, c.dilation, bias=c.bias, scale=scale,
                                           shift=shift, residual=residual, relu=self.relu, compute_dtype=c.compute_dtype,
                                           order=c.order)
