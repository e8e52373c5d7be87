"""Integer voxel coordinates (reference `warpconvnet/geometry/coords/integer.py:23-211`)."""
from typing import List, Optional, Tuple, Union

import torch
from torch import Tensor

from warpconvnet_amd.geometry.base.coords import Coords
from warpconvnet_amd.geometry.coords.ops.batch_index import batch_index_from_offset, batch_indexed_coordinates
from warpconvnet_amd.geometry.utils.list_to_batch import list_to_cat_tensor
from warpconvnet_amd.utils.ntuple import ntuple


class IntCoords(Coords):
    def __init__(
        self,
        batched_tensor: Union[List[Tensor], Tensor],
        offsets: Optional[Union[List[int], Tensor]] = None,
        voxel_size: Optional[float] = None,
        tensor_stride: Optional[Union[int, Tuple[int, ...]]] = None,
        device: Optional[str] = None,
    ):
        if isinstance(batched_tensor, (list, tuple)):
            assert offsets is None, "If batched_tensors is a list, offsets must be None."
            batched_tensor, offsets, _ = list_to_cat_tensor(batched_tensor)
        self.voxel_size = voxel_size
        self.tensor_stride = None
        self._hashmap = None
        super().__init__(batched_tensor, offsets, device=device)
        if tensor_stride is not None:
            self.tensor_stride = ntuple(tensor_stride, ndim=self.batched_tensor.shape[1])

    def check(self):
        super().check()
        assert self.batched_tensor.dtype in (torch.int32, torch.int64), "Discrete coordinates must be integers"

    @property
    def stride(self):
        return self.tensor_stride

    def set_tensor_stride(self, tensor_stride: Union[int, Tuple[int, ...]]):
        self.tensor_stride = ntuple(tensor_stride, ndim=self.num_spatial_dims)

    def _like(self, tensor: Tensor, offsets: Tensor) -> "IntCoords":
        return self.__class__(tensor, offsets, voxel_size=self.voxel_size, tensor_stride=self.tensor_stride)

    def _new(self, tensor: Tensor) -> "IntCoords":
        out = super()._new(tensor)
        out._hashmap = None
        return out

    def unique(self) -> "IntCoords":
        from warpconvnet_amd.geometry.coords.ops.voxel import voxel_downsample_random_indices

        idx, offsets = voxel_downsample_random_indices(self.batched_tensor, self.offsets)
        return self._like(self.batched_tensor[idx], offsets)

    def prune(self, mask: Tensor) -> "IntCoords":
        """Keep rows where ``mask`` is true; the number of batches is preserved."""
        assert mask.shape[0] == self.batched_tensor.shape[0], "Mask must match tensor shape"
        mask = mask.to(self.batched_tensor.device).bool()
        bidx = batch_index_from_offset(self.offsets, device=self.batched_tensor.device)
        counts = torch.bincount(bidx[mask].long(), minlength=self.batch_size).cpu()
        offsets = torch.cat([torch.zeros(1, dtype=torch.int64), counts.cumsum(0)]).to(self.offsets.dtype)
        return self._like(self.batched_tensor[mask], offsets)

    def expand(self, kernel_size: Union[int, Tuple[int, ...]], dilation: Union[int, Tuple[int, ...]] = 1) -> "IntCoords":
        """Coordinates plus all their kernel offsets, de-duplicated, batch-sorted (reference `integer.py:147-190`);
        the tensor stride is kept - expansion does not scale coordinates."""
        from warpconvnet_amd.geometry.coords.ops.expand import expand_coords

        nd = self.num_spatial_dims
        bcoords = batch_indexed_coordinates(self.batched_tensor, self.offsets)
        out, offsets = expand_coords(bcoords, ntuple(kernel_size, ndim=nd), ntuple(dilation, ndim=nd))
        return self._like(out[:, 1:].contiguous(), offsets)

    @property
    def hashmap(self):
        from warpconvnet_amd.geometry.coords.search.packed_hashmap import PackedHashTable

        if self._hashmap is None:
            bcoords = batch_indexed_coordinates(self.batched_tensor, self.offsets).to(torch.int32)
            if bcoords.shape[1] == 3:
                bcoords = torch.nn.functional.pad(bcoords, (0, 1), value=0)
            self._hashmap = PackedHashTable.from_coords(bcoords)
        return self._hashmap
