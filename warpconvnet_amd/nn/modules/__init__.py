from .base_module import BaseSpatialModel, BaseSpatialModule
from .sequential import Sequential
from .sparse_conv import SparseConv2d, SparseConv3d, SpatiallySparseConv

__all__ = ["BaseSpatialModel", "BaseSpatialModule", "Sequential", "SparseConv2d", "SparseConv3d", "SpatiallySparseConv"]
