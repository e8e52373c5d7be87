"""Real-valued point coordinates (reference `warpconvnet/geometry/coords/real.py:16-91`)."""
from typing import List, Optional, Union

import torch
from torch import Tensor

from warpconvnet_amd.geometry.base.coords import Coords


class RealCoords(Coords):
    def check(self):
        super().check()
        assert self.batched_tensor.is_floating_point(), "Real coordinates must be floating point"
        assert self.batched_tensor.ndim == 2

    def voxel_downsample(self, voxel_size: float):
        """Keep one point per voxel (first occurrence, deterministic). Returns ``(RealCoords, row indices)``."""
        from warpconvnet_amd.geometry.coords.ops.voxel import voxel_downsample_random_indices

        idx, offsets = voxel_downsample_random_indices(self.batched_tensor, self.offsets, voxel_size)
        return self.__class__(self.batched_tensor[idx], offsets), idx

    def neighbors(self, search_args, query_coords: Optional["RealCoords"] = None):
        from warpconvnet_amd.geometry.coords.search.continuous import neighbor_search

        if query_coords is None:
            query_coords = self
        return neighbor_search(
            self.batched_tensor, self.offsets, query_coords.batched_tensor, query_coords.offsets, search_args
        )
