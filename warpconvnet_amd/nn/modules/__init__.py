from .base_module import BaseSpatialModel, BaseSpatialModule
from .fused_block import FusedSparseConvBlock
from .mlp import MLPBlock
from .normalizations import BatchNorm, NormalizationBase
from .point_conv import PointConv
from .sequential import Sequential
from .sparse_conv import SparseConv2d, SparseConv3d, SpatiallySparseConv
from .sparse_pool import GlobalPool, SparseMaxPool, SparseMinPool, SparsePool, SparseUnpool
from .sparse_conv_depth import SparseDepthwiseConv2d, SparseDepthwiseConv3d, SpatiallySparseDepthwiseConv

__all__ = ["BaseSpatialModel", "BaseSpatialModule", "MLPBlock", "PointConv", "Sequential", "SparseConv2d", "SparseConv3d", "SpatiallySparseConv",
           "SparseDepthwiseConv2d", "SparseDepthwiseConv3d", "SpatiallySparseDepthwiseConv",
           "BatchNorm", "NormalizationBase", "FusedSparseConvBlock", "GlobalPool", "SparseMaxPool", "SparseMinPool", "SparsePool", "SparseUnpool"]
