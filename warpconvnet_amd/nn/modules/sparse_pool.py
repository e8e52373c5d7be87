"""Pooling modules over ``Voxels`` (reference `warpconvnet/nn/modules/sparse_pool.py:20-138`)."""
from typing import Literal

from warpconvnet_amd.geometry.base.geometry import Geometry
from warpconvnet_amd.geometry.types.voxels import Voxels
from warpconvnet_amd.nn.functional.sparse_pool import global_pool, sparse_reduce, sparse_unpool
from warpconvnet_amd.nn.modules.base_module import BaseSpatialModule


class SparsePool(BaseSpatialModule):
    """Reduce the features of every ``kernel_size`` window placed with ``stride`` (``max`` / ``min`` / ``mean`` / ``sum``)."""

    def __init__(self, kernel_size: int, stride: int, reduce: Literal["max", "min", "mean", "sum"] = "max"):
        super().__init__()
        self.kernel_size, self.stride, self.reduce = kernel_size, stride, reduce

    def __repr__(self):
        return f"{self.__class__.__name__}(kernel_size={self.kernel_size}, stride={self.stride}, reduce={self.reduce})"

    def forward(self, st: Voxels) -> Voxels:
        return sparse_reduce(st, self.kernel_size, self.stride, self.reduce)


class SparseMaxPool(SparsePool):
    def __init__(self, kernel_size: int, stride: int):
        super().__init__(kernel_size, stride, "max")


class SparseMinPool(SparsePool):
    def __init__(self, kernel_size: int, stride: int):
        super().__init__(kernel_size, stride, "min")


class GlobalPool(BaseSpatialModule):
    """One feature row per batch element."""

    def __init__(self, reduce: Literal["min", "max", "mean", "sum"] = "max"):
        super().__init__()
        self.reduce = reduce

    def forward(self, x: Geometry) -> Geometry:
        return global_pool(x, self.reduce)


class SparseUnpool(BaseSpatialModule):
    """Copy pooled features back onto the fine voxels (optionally concatenated with the fine features)."""

    def __init__(self, kernel_size: int, stride: int, concat_unpooled_st: bool = True):
        super().__init__()
        self.kernel_size, self.stride, self.concat_unpooled_st = kernel_size, stride, concat_unpooled_st

    def forward(self, st: Voxels, unpooled_st: Voxels) -> Voxels:
        return sparse_unpool(st, unpooled_st, self.kernel_size, self.stride, self.concat_unpooled_st)
