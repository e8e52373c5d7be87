"""Coordinate batch base (reference `geometry/base/coords.py:12-35`)."""
from torch import Tensor

from .batched import BatchedTensor


class Coords(BatchedTensor):
    @property
    def num_spatial_dims(self) -> int:
        return self.batched_tensor.shape[1]

    @property
    def batch_indexed_coordinates(self) -> Tensor:
        from warpconvnet_amd.geometry.coords.ops.batch_index import batch_indexed_coordinates

        return batch_indexed_coordinates(self.batched_tensor, self.offsets)

    def neighbors(self, query_coords: "Coords", search_args):
        raise NotImplementedError
