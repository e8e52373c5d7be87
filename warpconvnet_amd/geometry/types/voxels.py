"""Sparse voxel tensor (reference `warpconvnet/geometry/types/voxels.py:23-317`)."""
from typing import List, Optional, Tuple, Union

import torch
from torch import Tensor

from warpconvnet_amd.geometry.base.geometry import Geometry
from warpconvnet_amd.geometry.coords.integer import IntCoords
from warpconvnet_amd.geometry.coords.ops.batch_index import offsets_from_batch_index
from warpconvnet_amd.geometry.coords.real import RealCoords
from warpconvnet_amd.geometry.features.cat import CatFeatures, to_batched_features


def _ravel(bcoords: Tensor, shape: Tuple[int, ...]) -> Tensor:
    idx = torch.zeros(bcoords.shape[0], dtype=torch.int64, device=bcoords.device)
    for d, s in enumerate(shape):
        idx = idx * int(s) + bcoords[:, d].long()
    return idx


class Voxels(Geometry):
    def __init__(
        self,
        batched_coordinates: Union[List[Tensor], Tensor, IntCoords],
        batched_features: Union[List[Tensor], Tensor, CatFeatures],
        offsets: Optional[Tensor] = None,
        device: Optional[str] = None,
        **kwargs,
    ):
        tensor_stride = kwargs.pop("tensor_stride", None) or kwargs.pop("stride", None)
        if isinstance(batched_coordinates, list):
            assert isinstance(batched_features, list), "If coords is a list, features must be a list too."
            assert len(batched_coordinates) == len(batched_features)
            assert all(len(c) == len(f) for c, f in zip(batched_coordinates, batched_features))
            batched_coordinates = IntCoords(batched_coordinates, device=device, tensor_stride=tensor_stride)
        elif isinstance(batched_coordinates, Tensor):
            assert isinstance(batched_features, Tensor) and offsets is not None, (
                "If coordinate is a tensor, features must be a tensor and offsets must be provided."
            )
            batched_coordinates = IntCoords(batched_coordinates, offsets=offsets, device=device, tensor_stride=tensor_stride)
        elif tensor_stride is not None:
            batched_coordinates.set_tensor_stride(tensor_stride)
        if isinstance(batched_features, list):
            batched_features = CatFeatures(batched_features, device=device)
        elif isinstance(batched_features, Tensor):
            batched_features = to_batched_features(batched_features, batched_coordinates.offsets, device=device)
        Geometry.__init__(self, batched_coordinates, batched_features, **kwargs)

    # ---- metadata -----------------------------------------------------------------------------
    @property
    def tensor_stride(self):
        return self.batched_coordinates.tensor_stride

    stride = tensor_stride

    def set_tensor_stride(self, tensor_stride):
        self.batched_coordinates.set_tensor_stride(tensor_stride)

    @property
    def voxel_size(self):
        return self._extra_attributes.get("voxel_size", None)

    @property
    def ordering(self):
        return self._extra_attributes.get("ordering", None)

    @property
    def coordinate_hashmap(self):
        return self.batched_coordinates.hashmap

    @property
    def spatial_cache(self) -> dict:
        return self._extra_attributes.setdefault("_spatial_cache", {})

    # ---- de-duplication / conversions -----------------------------------------------------------
    def unique(self) -> "Voxels":
        from warpconvnet_amd.geometry.coords.ops.voxel import voxel_downsample_random_indices

        idx, offsets = voxel_downsample_random_indices(self.coordinate_tensor, self.offsets)
        coords = IntCoords(self.coordinate_tensor[idx], offsets, tensor_stride=self.tensor_stride)
        feats = CatFeatures(self.batched_features.batched_tensor[idx], offsets)
        return self.__class__(coords, feats, **self.extra_attributes)

    def sort(self, ordering="morton_xyz") -> "Voxels":
        """Rows of every batch element in space-filling-curve order (reference `types/voxels.py:248-269`); the
        codes are kept as the ``code`` attribute, the ordering as ``ordering``."""
        from warpconvnet_amd.geometry.coords.ops.serialization import encode, to_point_ordering

        ordering = to_point_ordering(ordering)
        if ordering == self.ordering:
            return self
        assert isinstance(self.batched_features, CatFeatures), "Features must be a CatFeatures to sort."
        res = encode(self.coordinate_tensor, batch_offsets=self.offsets, order=ordering, return_perm=True)
        kwargs = self.extra_attributes
        kwargs["ordering"], kwargs["code"] = ordering, res.codes
        kwargs.pop("_cache", None)  # cached kernel maps index rows of the old order
        kwargs.pop("_spatial_cache", None)
        coords = IntCoords(self.coordinate_tensor[res.perm], self.offsets, tensor_stride=self.tensor_stride)
        feats = CatFeatures(self.batched_features.batched_tensor[res.perm], self.offsets)
        return self.__class__(coords, feats, **kwargs)

    def to_dense(self, channel_dim: int = 1, spatial_shape: Optional[Tuple[int, ...]] = None,
                 min_coords: Optional[Tuple[int, ...]] = None, max_coords: Optional[Tuple[int, ...]] = None) -> Tensor:
        """[B, C, *spatial] dense tensor (channel position selectable), zeros where no voxel exists."""
        bcoords = self.batch_indexed_coordinates.clone().long()
        feats = self.batched_features.batched_tensor
        nd = self.num_spatial_dims
        if min_coords is None:
            if bcoords.shape[0] == 0:
                lo = torch.zeros(nd, dtype=torch.long, device=bcoords.device)
                hi = lo - 1
            else:
                lo, hi = bcoords[:, 1:].min(0).values, bcoords[:, 1:].max(0).values
            if spatial_shape is None:
                spatial_shape = tuple(int(v) for v in (hi - lo + 1).tolist())
                bcoords[:, 1:] -= lo
            # else: coordinates are taken as already aligned with spatial_shape
        else:
            lo = torch.tensor(min_coords, dtype=torch.long, device=bcoords.device)
            if max_coords is not None:
                spatial_shape = tuple(int(b - a + 1) for a, b in zip(min_coords, max_coords))
            assert spatial_shape is not None, "give max_coords or spatial_shape together with min_coords"
            bcoords[:, 1:] -= lo
            shp = torch.tensor(spatial_shape, dtype=torch.long, device=bcoords.device)
            keep = ((bcoords[:, 1:] >= 0) & (bcoords[:, 1:] < shp)).all(1)
            bcoords, feats = bcoords[keep], feats[keep]
        dense = torch.zeros((self.batch_size, *spatial_shape, self.num_channels), dtype=feats.dtype, device=feats.device)
        if bcoords.shape[0] > 0:
            dense.flatten(0, -2)[_ravel(bcoords, (self.batch_size, *spatial_shape))] = feats
        return dense.moveaxis(-1, channel_dim) if channel_dim != -1 else dense

    @classmethod
    def from_dense(cls, dense_tensor: Tensor, dense_tensor_channel_dim: int = 1,
                   target_spatial_sparse_tensor: Optional["Voxels"] = None, **kwargs) -> "Voxels":
        dense = dense_tensor.moveaxis(dense_tensor_channel_dim, -1)
        flat = dense.flatten(0, -2)
        if target_spatial_sparse_tensor is None:
            nz = torch.nonzero(dense.abs().sum(-1)).int()
            offsets = offsets_from_batch_index(nz[:, 0], num_batches=dense.shape[0])
            feats = flat[_ravel(nz.long(), dense.shape[:-1])]
            return cls(IntCoords(nz[:, 1:].contiguous(), offsets=offsets), CatFeatures(feats, offsets), **kwargs)
        t = target_spatial_sparse_tensor
        bcoords = t.batch_indexed_coordinates.clone().long()
        bcoords[:, 1:] -= t.coordinate_tensor.min(0).values.long()
        return t.replace(batched_features=flat[_ravel(bcoords, dense.shape[:-1])])

    def to_point(self, voxel_size: Optional[float] = None):
        from warpconvnet_amd.geometry.types.points import Points

        voxel_size = voxel_size if voxel_size is not None else self.voxel_size
        assert voxel_size is not None, "voxel_size is required to convert voxels to points"
        scale = torch.tensor([[voxel_size * s for s in (self.tensor_stride or (1,) * self.num_spatial_dims)]],
                             device=self.device, dtype=torch.float32)
        return Points(RealCoords(self.coordinate_tensor.float() * scale, self.offsets), self.batched_features)
