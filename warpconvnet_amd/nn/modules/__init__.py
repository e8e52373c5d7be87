from .base_module import BaseSpatialModel, BaseSpatialModule
from .sequential import Sequential
from .sparse_conv import SparseConv2d, SparseConv3d, SpatiallySparseConv
from .sparse_conv_depth import SparseDepthwiseConv2d, SparseDepthwiseConv3d, SpatiallySparseDepthwiseConv

__all__ = ["BaseSpatialModel", "BaseSpatialModule", "Sequential", "SparseConv2d", "SparseConv3d", "SpatiallySparseConv",
           "SparseDepthwiseConv2d", "SparseDepthwiseConv3d", "SpatiallySparseDepthwiseConv"]
