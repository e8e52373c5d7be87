"""``SparseDepthwiseConv3d`` / ``SparseDepthwiseConv2d`` modules: one ``[K]`` kernel per channel, weight ``[K, C]``.

Constructor and ``forward`` of the reference (`warpconvnet/nn/modules/sparse_conv_depth.py:34-338`): uniform init with
bound ``sqrt(num_spatial_dims) * gain(leaky_relu, sqrt 5) / sqrt(K)`` (fan_in = fan_out = kernel volume), bias
``U(+-1/sqrt(K))``; the bias is added after the convolution; ``tensor_stride`` bookkeeping as for ``SparseConv3d``.
"""
import math
from typing import Optional, Tuple, Union

import numpy as np
import torch
import torch.nn as nn
from torch.nn.init import calculate_gain

from warpconvnet_amd.constants import (
    WARPCONVNET_DEPTHWISE_CONV_BWD_ALGO_MODE,
    WARPCONVNET_DEPTHWISE_CONV_FWD_ALGO_MODE,
)
from warpconvnet_amd.geometry.coords.integer import IntCoords
from warpconvnet_amd.geometry.types.voxels import Voxels
from warpconvnet_amd.nn.functional.sparse_conv import STRIDED_CONV_MODE, generate_output_coords_and_kernel_map
from warpconvnet_amd.nn.functional.sparse_conv_depth import (
    SPARSE_DEPTHWISE_CONV_BWD_ALGO_MODE,
    SPARSE_DEPTHWISE_CONV_FWD_ALGO_MODE,
    _parse,
    spatially_sparse_depthwise_conv,
)
from warpconvnet_amd.nn.modules.base_module import BaseSpatialModule
from warpconvnet_amd.utils.ntuple import ntuple


class SpatiallySparseDepthwiseConv(BaseSpatialModule):
    def __init__(
        self,
        channels: int,
        kernel_size: Union[int, Tuple[int, ...]],
        stride: Union[int, Tuple[int, ...]] = 1,
        dilation: Union[int, Tuple[int, ...]] = 1,
        bias: bool = True,
        transposed: bool = False,
        generative: bool = False,
        num_spatial_dims: int = 3,
        fwd_algo: Optional[Union[SPARSE_DEPTHWISE_CONV_FWD_ALGO_MODE, str]] = None,
        bwd_algo: Optional[Union[SPARSE_DEPTHWISE_CONV_BWD_ALGO_MODE, str]] = None,
        stride_mode: STRIDED_CONV_MODE = STRIDED_CONV_MODE.STRIDE_ONLY,
        stride_reduce: str = "max",
        order=None,
        compute_dtype: Optional[torch.dtype] = None,
    ):
        super().__init__()
        self.num_spatial_dims = num_spatial_dims
        self.channels = self.in_channels = self.out_channels = channels
        self.kernel_size = ntuple(kernel_size, ndim=num_spatial_dims)
        self.stride = ntuple(stride, ndim=num_spatial_dims)
        self.dilation = ntuple(dilation, ndim=num_spatial_dims)
        self.transposed, self.generative, self.stride_reduce = transposed, generative, stride_reduce
        self.fwd_algo = _parse(fwd_algo if fwd_algo is not None else WARPCONVNET_DEPTHWISE_CONV_FWD_ALGO_MODE,
                               SPARSE_DEPTHWISE_CONV_FWD_ALGO_MODE)
        self.bwd_algo = _parse(bwd_algo if bwd_algo is not None else WARPCONVNET_DEPTHWISE_CONV_BWD_ALGO_MODE,
                               SPARSE_DEPTHWISE_CONV_BWD_ALGO_MODE)
        self.stride_mode, self.order, self.compute_dtype = stride_mode, order, compute_dtype
        K = int(np.prod(self.kernel_size))
        # randn first, like the reference (:115-121): the seeded RNG stream - and so the initial state_dict - matches
        self.weight = nn.Parameter(torch.randn(K, channels))
        self.bias = nn.Parameter(torch.randn(channels)) if bias else None
        self.reset_parameters()

    def __repr__(self):
        s = f"{self.__class__.__name__}(channels={self.channels}, kernel_size={self.kernel_size}"
        if any(v != 1 for v in self.stride):
            s += f", stride={self.stride}"
        if any(v != 1 for v in self.dilation):
            s += f", dilation={self.dilation}"
        if self.transposed:
            s += ", transposed=True"
        if self.bias is None:
            s += ", bias=False"
        return s + ")"

    def _calculate_fan_in_and_fan_out(self):
        rf = int(np.prod(self.kernel_size))
        return rf, rf

    @torch.no_grad()
    def reset_parameters(self):
        fan_in, fan_out = self._calculate_fan_in_and_fan_out()
        fan = fan_out if self.transposed else fan_in
        bound = math.sqrt(self.num_spatial_dims) * calculate_gain("leaky_relu", math.sqrt(5)) / math.sqrt(fan)
        self.weight.uniform_(-bound, bound)
        if self.bias is not None:
            b = 1.0 / math.sqrt(fan_in) if fan_in > 0 else 0.0
            self.bias.uniform_(-b, b)

    def forward(self, input_sparse_tensor: Voxels, output_spatially_sparse_tensor: Optional[Voxels] = None) -> Voxels:
        bcoords_out, out_offsets, kernel_map = generate_output_coords_and_kernel_map(
            input_sparse_tensor=input_sparse_tensor, kernel_size=self.kernel_size, kernel_dilation=self.dilation,
            stride=self.stride, generative=self.generative, transposed=self.transposed,
            output_spatially_sparse_tensor=output_spatially_sparse_tensor, stride_mode=self.stride_mode, order=self.order,
        )
        out = spatially_sparse_depthwise_conv(input_sparse_tensor.feature_tensor, self.weight, kernel_map, bcoords_out.shape[0],
                                              fwd_algo=self.fwd_algo, bwd_algo=self.bwd_algo, compute_dtype=self.compute_dtype)
        if self.bias is not None:
            out = out + self.bias
        in_ts = input_sparse_tensor.tensor_stride or (1,) * self.num_spatial_dims
        if not self.transposed:
            out_ts = tuple(o * s for o, s in zip(self.stride, in_ts))
        elif output_spatially_sparse_tensor is not None and output_spatially_sparse_tensor.tensor_stride is not None:
            out_ts = output_spatially_sparse_tensor.tensor_stride
        else:
            out_ts = (1,) * self.num_spatial_dims
        return input_sparse_tensor.replace(
            batched_coordinates=IntCoords(bcoords_out[:, 1:], offsets=out_offsets.cpu().int()),
            batched_features=out,
            tensor_stride=out_ts,
        )


class SparseDepthwiseConv2d(SpatiallySparseDepthwiseConv):
    def __init__(self, channels, kernel_size, stride=1, dilation=1, bias=True, transposed=False, generative=False, **kwargs):
        kwargs.pop("num_spatial_dims", None)
        super().__init__(channels, kernel_size, stride=stride, dilation=dilation, bias=bias, transposed=transposed,
                         generative=generative, num_spatial_dims=2, **kwargs)


class SparseDepthwiseConv3d(SpatiallySparseDepthwiseConv):
    def __init__(self, channels, kernel_size, stride=1, dilation=1, bias=True, transposed=False, generative=False, **kwargs):
        kwargs.pop("num_spatial_dims", None)
        super().__init__(channels, kernel_size, stride=stride, dilation=dilation, bias=bias, transposed=transposed,
                         generative=generative, num_spatial_dims=3, **kwargs)
