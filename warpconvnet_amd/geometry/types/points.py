"""Point cloud with real coordinates (reference `warpconvnet/geometry/types/points.py:33-326`).

The surface the sparse-conv hot path and its immediate callers use: construction (incl. ``from_list_of_coordinates`` with
sinusoidal features), neighbour search, voxel / random down-sampling, space-filling-curve ``sort``, ``contiguous`` and the
conversion to ``Voxels``; patch ops belong to other model families (out of scope).
"""
from typing import List, Optional, Tuple, Union

import torch
from torch import Tensor

from warpconvnet_amd.geometry.base.coords import Coords
from warpconvnet_amd.geometry.base.geometry import Geometry
from warpconvnet_amd.geometry.coords.real import RealCoords
from warpconvnet_amd.geometry.features.cat import CatFeatures, to_batched_features


class Points(Geometry):
    def __init__(
        self,
        batched_coordinates: Union[List[Tensor], Tensor, RealCoords],
        batched_features: Union[List[Tensor], Tensor, CatFeatures],
        offsets: Optional[Tensor] = None,
        device: Optional[str] = None,
        **kwargs,
    ):
        if isinstance(batched_coordinates, list):
            assert isinstance(batched_features, list), "If coords is a list, features must be a list too."
            assert len(batched_coordinates) == len(batched_features)
            assert all(len(c) == len(f) for c, f in zip(batched_coordinates, batched_features)), (
                "All elements in coords and features must have same length"
            )
            batched_coordinates = RealCoords(batched_coordinates, device=device)
        elif isinstance(batched_coordinates, Tensor):
            assert isinstance(batched_features, Tensor) and offsets is not None, (
                "If coordinate is a tensor, features must be a tensor and offsets must be provided."
            )
            batched_coordinates = RealCoords(batched_coordinates, offsets=offsets, device=device)
        if isinstance(batched_features, list):
            batched_features = CatFeatures(batched_features, device=device)
        elif isinstance(batched_features, Tensor):
            batched_features = to_batched_features(batched_features, batched_coordinates.offsets, device=device)
        Geometry.__init__(self, batched_coordinates, batched_features, **kwargs)

    @property
    def voxel_size(self):
        return self._extra_attributes.get("voxel_size", None)

    @property
    def ordering(self):
        return self._extra_attributes.get("ordering", None)

    @classmethod
    def from_list_of_coordinates(
        cls,
        coordinates: Union[List[Tensor], Tensor],
        features: Optional[List[Tensor]] = None,
        encoding_channels: Optional[int] = None,
        encoding_range: Optional[Union[float, Tuple[float, float]]] = None,
        encoding_dim: Optional[int] = -1,
    ) -> "Points":
        """Points from per-batch coordinate tensors (a [B, N, D] tensor is split along its first axis); without ``features``
        every point gets the sinusoidal encoding of its coordinates, ``encoding_channels`` per axis
        (reference `points.py:283-317`)."""
        from warpconvnet_amd.nn.functional.encodings import sinusoidal_encoding

        if isinstance(coordinates, Tensor):
            coordinates = list(coordinates)
        if features is None:
            assert encoding_range is not None, "Encoding range must be provided if encoding channels are provided"
            features = [sinusoidal_encoding(c, encoding_channels, encoding_range, encoding_dim) for c in coordinates]
        return cls(RealCoords(coordinates), CatFeatures(features))

    def sort(self, voxel_size: float, ordering="morton_xyz") -> "Points":
        """Rows of every batch element in space-filling-curve order of their ``voxel_size`` cells; points of one cell keep
        their input order (the sort is stable).  Reference `points.py:92-120` (device only there too: the codes come from a
        device kernel)."""
        from warpconvnet_amd.geometry.coords.ops.serialization import encode

        assert self.device.type != "cpu", "Sorting is only supported on GPU"
        result = encode(torch.floor(self.coordinate_tensor / voxel_size).int(), batch_offsets=self.offsets,
                        order=ordering, return_perm=True)
        return self.__class__(
            RealCoords(self.coordinate_tensor[result.perm], self.offsets),
            CatFeatures(self.feature_tensor[result.perm], self.offsets),
            **self.extra_attributes.copy(),
        )

    def random_downsample(self, num_sample_points: int) -> "Points":
        """``num_sample_points`` rows drawn with replacement from every batch element (reference `points.py:189-208`,
        `coords/sample.py:11-31`): the result holds ``batch_size * num_sample_points`` points."""
        from warpconvnet_amd.geometry.coords.sample import random_sample_per_batch

        idx, offsets = random_sample_per_batch(self.offsets, num_sample_points)
        idx = idx.to(self.coordinate_tensor.device, torch.int64)
        return self.__class__(
            RealCoords(self.coordinate_tensor[idx], offsets),
            CatFeatures(self.feature_tensor[idx], offsets),
            **self.extra_attributes,
        )

    def contiguous(self) -> "Points":
        """``self`` when coordinates and features are already contiguous, else a copy that is (reference `points.py:210-236`)."""
        if self.coordinate_tensor.is_contiguous() and self.feature_tensor.is_contiguous():
            return self
        return self.__class__(
            RealCoords(self.coordinate_tensor.contiguous(), self.offsets),
            CatFeatures(self.feature_tensor.contiguous(), self.offsets),
            **self.extra_attributes,
        )

    def neighbors(self, search_args, query_coords: Optional[Coords] = None):
        """CSR neighbour lists (``RealSearchResult``); cached per (config, offsets) like the reference."""
        from warpconvnet_amd.geometry.coords.search.continuous import neighbor_search

        if query_coords is None:
            query_coords = self.batched_coordinates
        assert isinstance(query_coords, Coords), "query_coords must be Coords"
        cache = self._extra_attributes.setdefault("_cache", {})
        key = (search_args, tuple(self.offsets.tolist()), tuple(query_coords.offsets.tolist()))
        if key not in cache:
            cache[key] = neighbor_search(
                self.coordinate_tensor, self.offsets, query_coords.batched_tensor, query_coords.offsets, search_args
            )
        return cache[key]

    def voxel_downsample(self, voxel_size: float, reduction="random") -> "Points":
        """One point per voxel of edge ``voxel_size`` (reference `points.py:122-190`): ``random`` keeps the first point
        and its features; any other reduction pools the features of the voxel's points (``row_reduction``) and keeps the
        first point's coordinates."""
        from warpconvnet_amd.geometry.coords.ops.batch_index import batch_indexed_coordinates, offsets_from_batch_index
        from warpconvnet_amd.geometry.coords.ops.voxel import voxel_downsample_random_indices
        from warpconvnet_amd.ops.reductions import REDUCTIONS, row_reduction

        if isinstance(reduction, str):
            reduction = REDUCTIONS(reduction)
        extra = {**self.extra_attributes, "voxel_size": voxel_size}
        if reduction == REDUCTIONS.RANDOM:
            idx, offsets = voxel_downsample_random_indices(self.coordinate_tensor, self.offsets, voxel_size)
            return self.__class__(
                RealCoords(self.coordinate_tensor[idx], offsets),
                CatFeatures(self.batched_features.batched_tensor[idx], offsets),
                **extra,
            )
        q = torch.floor(self.coordinate_tensor / voxel_size).to(torch.int32)
        bq = batch_indexed_coordinates(q, self.offsets)
        uniq = inverse = None
        if bq.is_cuda and bq.shape[0] > 0:
            # Rows (b, x, y, z) inside the packed range of the convolution's own keys (b < 512, |x| < 2^17, `csrc/wcn_common.h`) -
            # every input the sparse layers accept - are de-duplicated as ONE 64-bit key per row: the framework's unique over
            # rows is a merge sort with a row comparator (0.31 ms for 200 k points), over int64 keys a radix sort.  Biased fields
            # keep the lexicographic order of the signed rows, so the voxels come out in the same order either way.
            b64 = bq.to(torch.int64)
            key = (b64[:, 0] << 54) | ((b64[:, 1] + 131072) << 36) | ((b64[:, 2] + 131072) << 18) | (b64[:, 3] + 131072)
            in_range = ((bq[:, 0] >= 0) & (bq[:, 0] < 512) & (bq[:, 1:] >= -131072).all(1) & (bq[:, 1:] <= 131071).all(1)).all()
            ukey, inv = torch.unique(key, return_inverse=True)
            if bool(in_range):
                uniq = torch.stack([ukey >> 54, ((ukey >> 36) & 0x3FFFF) - 131072, ((ukey >> 18) & 0x3FFFF) - 131072,
                                    (ukey & 0x3FFFF) - 131072], 1).to(torch.int32)
                inverse = inv
        if uniq is None:
            uniq, inverse = torch.unique(bq, dim=0, return_inverse=True)  # lexicographic => batch-sorted voxels
        perm = torch.argsort(inverse, stable=True)                    # points grouped by voxel, input order inside
        counts = torch.bincount(inverse, minlength=uniq.shape[0])
        splits = torch.cat([counts.new_zeros(1), counts.cumsum(0)])
        feats = row_reduction(self.feature_tensor[perm], splits, reduction)
        offsets = offsets_from_batch_index(uniq[:, 0], num_batches=self.batch_size)
        return self.__class__(
            RealCoords(self.coordinate_tensor[perm[splits[:-1]]], offsets),
            CatFeatures(feats, offsets),
            **extra,
        )

    def to_voxels(self, voxel_size: float, reduction: str = "mean"):
        """Quantise to integer voxels, reducing the features of points that share a voxel."""
        from warpconvnet_amd.geometry.coords.integer import IntCoords
        from warpconvnet_amd.geometry.coords.ops.batch_index import batch_indexed_coordinates, offsets_from_batch_index
        from warpconvnet_amd.geometry.types.voxels import Voxels

        q = torch.floor(self.coordinate_tensor / voxel_size).to(torch.int32)
        bq = batch_indexed_coordinates(q, self.offsets)
        uniq, inverse = torch.unique(bq, dim=0, return_inverse=True)  # lexicographic => batch-sorted
        feats = self.batched_features.batched_tensor
        out = torch.zeros((uniq.shape[0], feats.shape[1]), dtype=feats.dtype, device=feats.device)
        if reduction in ("mean", "sum"):
            out.index_add_(0, inverse, feats)
            if reduction == "mean":
                counts = torch.bincount(inverse, minlength=uniq.shape[0]).clamp_min(1).to(feats.dtype)
                out = out / counts.unsqueeze(1)
        elif reduction == "max":
            out = torch.full_like(out, float("-inf")).scatter_reduce(0, inverse[:, None].expand_as(feats), feats, "amax")
        else:
            raise ValueError(f"unsupported reduction {reduction!r}")
        offsets = offsets_from_batch_index(uniq[:, 0], num_batches=self.batch_size)
        return Voxels(IntCoords(uniq[:, 1:].contiguous(), offsets), CatFeatures(out, offsets), voxel_size=voxel_size)
