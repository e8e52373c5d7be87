"""Coordinate batch base (reference `geometry/base/coords.py:12-35`)."""
from torch import Tensor

from .batched import BatchedTensor


class Coords(BatchedTensor):
    @property
    def num_spatial_dims(self) -> int:
        return self.batched_tensor.shape[1]

    @property
    def batch_indexed_coordinates(self) -> Tensor:
        """[N, D+1] with the batch index in column 0; cached on the object (coordinates are immutable here), which
        also lets the kernel-map builder recognise "same coordinate tensor" by pointer."""
        from warpconvnet_amd.geometry.coords.ops.batch_index import batch_indexed_coordinates

        cached = self.__dict__.get("_bcoords")
        if cached is None or cached.shape[0] != self.batched_tensor.shape[0] or cached.device != self.batched_tensor.device:
            cached = batch_indexed_coordinates(self.batched_tensor, self.offsets)
            self.__dict__["_bcoords"] = cached
        return cached

    def neighbors(self, query_coords: "Coords", search_args):
        raise NotImplementedError
