"""``SparseConv3d`` / ``SparseConv2d`` modules.

Constructor and ``forward`` signature of the reference (`warpconvnet/nn/modules/sparse_conv.py:31-391`):
weight ``[K, Cin, Cout]`` (groups: ``[K, G, Cin/G, Cout/G]``), optional bias, uniform init with bound
``sqrt(num_spatial_dims) * gain(leaky_relu, sqrt 5) / sqrt(fan)``, fan = ``(Cin/G) * K`` (fan_out when
transposed), bias ``U(+-1/sqrt(fan_in))``; algorithm knobs default to the ``WARPCONVNET_*_ALGO_MODE``
environment variables.
"""
import math
from typing import Optional, Tuple, Union

import numpy as np
import torch
import torch.nn as nn
from torch.nn.init import calculate_gain

from warpconvnet_amd.constants import (
    WARPCONVNET_DGRAD_ALGO_MODE,
    WARPCONVNET_FWD_ALGO_MODE,
    WARPCONVNET_WGRAD_ALGO_MODE,
)
from warpconvnet_amd.geometry.types.voxels import Voxels
from warpconvnet_amd.nn.functional.sparse_conv import (
    SPARSE_CONV_AB_ALGO_MODE,
    SPARSE_CONV_ATB_ALGO_MODE,
    STRIDED_CONV_MODE,
    spatially_sparse_conv,
)
from warpconvnet_amd.nn.modules.base_module import BaseSpatialModule
from warpconvnet_amd.utils.ntuple import ntuple


def _parse_algo(algo, enum_cls):
    return enum_cls(algo) if isinstance(algo, str) else algo


class SpatiallySparseConv(BaseSpatialModule):
    def __init__(
        self,
        in_channels: int,
        out_channels: int,
        kernel_size: Union[int, Tuple[int, ...]],
        stride: Union[int, Tuple[int, ...]] = 1,
        dilation: Union[int, Tuple[int, ...]] = 1,
        bias: bool = True,
        transposed: bool = False,
        generative: bool = False,
        groups: int = 1,
        kernel_matmul_batch_size: int = 2,
        num_spatial_dims: Optional[int] = 3,
        fwd_algo: Optional[Union[SPARSE_CONV_AB_ALGO_MODE, str]] = None,
        dgrad_algo: Optional[Union[SPARSE_CONV_AB_ALGO_MODE, str]] = None,
        wgrad_algo: Optional[Union[SPARSE_CONV_ATB_ALGO_MODE, str]] = None,
        stride_mode: STRIDED_CONV_MODE = STRIDED_CONV_MODE.STRIDE_ONLY,
        order=None,
        compute_dtype: Optional[torch.dtype] = None,
        use_fp16_accum: Optional[bool] = None,
        implicit_matmul_fwd_block_size: Optional[int] = None,
        implicit_matmul_bwd_block_size: Optional[int] = None,
    ):
        super().__init__()
        if in_channels % groups != 0:
            raise ValueError(f"in_channels ({in_channels}) must be divisible by groups ({groups})")
        if out_channels % groups != 0:
            raise ValueError(f"out_channels ({out_channels}) must be divisible by groups ({groups})")
        self.num_spatial_dims = num_spatial_dims
        self.in_channels, self.out_channels, self.groups = in_channels, out_channels, groups
        self.kernel_size = ntuple(kernel_size, ndim=num_spatial_dims)
        self.stride = ntuple(stride, ndim=num_spatial_dims)
        self.dilation = ntuple(dilation, ndim=num_spatial_dims)
        self.transposed, self.generative = transposed, generative
        self.kernel_matmul_batch_size = kernel_matmul_batch_size
        self.fwd_algo = _parse_algo(fwd_algo if fwd_algo is not None else WARPCONVNET_FWD_ALGO_MODE, SPARSE_CONV_AB_ALGO_MODE)
        self.dgrad_algo = _parse_algo(dgrad_algo if dgrad_algo is not None else WARPCONVNET_DGRAD_ALGO_MODE, SPARSE_CONV_AB_ALGO_MODE)
        self.wgrad_algo = _parse_algo(wgrad_algo if wgrad_algo is not None else WARPCONVNET_WGRAD_ALGO_MODE, SPARSE_CONV_ATB_ALGO_MODE)
        self.stride_mode, self.order = stride_mode, order
        self.compute_dtype, self.use_fp16_accum = compute_dtype, use_fp16_accum
        self.implicit_matmul_fwd_block_size = implicit_matmul_fwd_block_size
        self.implicit_matmul_bwd_block_size = implicit_matmul_bwd_block_size

        K = int(np.prod(self.kernel_size))
        # randn first, like the reference (sparse_conv.py:147-161): keeps the seeded RNG stream, hence the
        # initial state_dict, identical for a given torch.manual_seed
        if groups == 1:
            self.weight = nn.Parameter(torch.randn(K, in_channels, out_channels))
        else:
            self.weight = nn.Parameter(torch.randn(K, groups, in_channels // groups, out_channels // groups))
        self.bias = nn.Parameter(torch.randn(out_channels)) if bias else None
        self.reset_parameters()

    def __repr__(self):
        s = (f"{self.__class__.__name__}(in_channels={self.in_channels}, out_channels={self.out_channels}, "
             f"kernel_size={self.kernel_size}")
        if any(v != 1 for v in self.stride):
            s += f", stride={self.stride}"
        if any(v != 1 for v in self.dilation):
            s += f", dilation={self.dilation}"
        if self.groups != 1:
            s += f", groups={self.groups}"
        if self.transposed:
            s += ", transposed=True"
        return s + ")"

    def _calculate_fan_in_and_fan_out(self):
        rf = int(np.prod(self.kernel_size))
        return (self.in_channels // self.groups) * rf, (self.out_channels // self.groups) * rf

    @torch.no_grad()
    def reset_parameters(self):
        fan_in, fan_out = self._calculate_fan_in_and_fan_out()
        fan = fan_out if self.transposed else fan_in
        bound = math.sqrt(self.num_spatial_dims) * calculate_gain("leaky_relu", math.sqrt(5)) / math.sqrt(fan)
        self.weight.uniform_(-bound, bound)
        if self.bias is not None:
            b = 1.0 / math.sqrt(fan_in)
            self.bias.uniform_(-b, b)

    def forward(self, input_sparse_tensor: Voxels, output_spatially_sparse_tensor: Optional[Voxels] = None):
        return spatially_sparse_conv(
            input_sparse_tensor=input_sparse_tensor,
            weight=self.weight,
            kernel_size=self.kernel_size,
            stride=self.stride,
            kernel_dilation=self.dilation,
            bias=self.bias,
            groups=self.groups,
            kernel_matmul_batch_size=self.kernel_matmul_batch_size,
            output_spatially_sparse_tensor=output_spatially_sparse_tensor,
            transposed=self.transposed,
            generative=self.generative,
            fwd_algo=self.fwd_algo,
            dgrad_algo=self.dgrad_algo,
            wgrad_algo=self.wgrad_algo,
            stride_mode=self.stride_mode,
            order=self.order,
            compute_dtype=self.compute_dtype,
            use_fp16_accum=self.use_fp16_accum,
            implicit_matmul_fwd_block_size=self.implicit_matmul_fwd_block_size,
            implicit_matmul_bwd_block_size=self.implicit_matmul_bwd_block_size,
        )


class SparseConv3d(SpatiallySparseConv):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, dilation=1, bias=True, transposed=False,
                 generative=False, groups=1, **kwargs):
        kwargs.pop("num_spatial_dims", None)
        super().__init__(in_channels, out_channels, kernel_size, stride=stride, dilation=dilation, bias=bias,
                         transposed=transposed, generative=generative, groups=groups, num_spatial_dims=3, **kwargs)


class SparseConv2d(SpatiallySparseConv):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, dilation=1, bias=True, transposed=False,
                 generative=False, groups=1, **kwargs):
        kwargs.pop("num_spatial_dims", None)
        super().__init__(in_channels, out_channels, kernel_size, stride=stride, dilation=dilation, bias=bias,
                         transposed=transposed, generative=generative, groups=groups, num_spatial_dims=2, **kwargs)
