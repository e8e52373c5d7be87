"""Geometry = coordinates + features + free-form extra attributes.

API of the reference `warpconvnet/geometry/base/geometry.py:38-388`: ``replace()`` carries
``_extra_attributes`` (incl. the kernel-map ``_cache``) forward, ``feature_tensor`` follows the autocast
dtype, arithmetic acts on features, ``to()`` returns new objects.
"""
from dataclasses import dataclass, field
from typing import Any, Dict, Optional, Union

import torch
from torch import Tensor

from warpconvnet_amd.geometry.base.coords import Coords
from warpconvnet_amd.geometry.base.features import Features
from warpconvnet_amd.geometry.features.cat import to_batched_features


@dataclass(eq=False)
class Geometry:
    # a dataclass like the reference's (geometry.py:38-63): `dataclasses.asdict / replace / fields` work on every geometry
    # type (reference tests/types/test_points.py:61-74); the hand-written __init__ below is kept by the decorator.  eq=False:
    # identity comparison and hashing stay (the generated __eq__ would compare tensors element-wise and drop __hash__).
    batched_coordinates: Coords
    batched_features: Features
    _extra_attributes: Dict[str, Any] = field(default_factory=dict, init=True)

    def __init__(self, batched_coordinates: Union[Coords, Tensor], batched_features, **kwargs):
        offsets = kwargs.pop("offsets", None)
        device = kwargs.pop("device", None)
        if isinstance(batched_coordinates, Tensor):
            assert offsets is not None, "offsets must be provided when batched_coordinates is a tensor"
            batched_coordinates = Coords(batched_coordinates, offsets)
        self.batched_coordinates = batched_coordinates
        self.batched_features = to_batched_features(batched_features, batched_coordinates.offsets, device=device)
        assert bool((batched_coordinates.offsets == self.batched_features.offsets).all()), "coords/features offsets differ"
        if "_extra_attributes" in kwargs:  # flatten (happens when an attribute dict is forwarded)
            attr = kwargs.pop("_extra_attributes")
            assert isinstance(attr, dict)
            kwargs = {**attr, **kwargs}
        self._extra_attributes: Dict[str, Any] = kwargs

    # ---- tensors ------------------------------------------------------------------------------
    @property
    def coordinate_tensor(self) -> Tensor:
        return self.batched_coordinates.batched_tensor

    coordinates = coordinate_tensor

    @property
    def batch_indexed_coordinates(self) -> Tensor:
        return self.batched_coordinates.batch_indexed_coordinates

    @property
    def coords(self) -> Tensor:
        return self.batch_indexed_coordinates

    @property
    def feature_tensor(self) -> Tensor:
        t = self.batched_features.batched_tensor
        if torch.is_autocast_enabled():
            amp = torch.get_autocast_dtype("cuda")
            if t.dtype != amp:
                return t.to(dtype=amp)
        return t

    features = feature_tensor
    feats = feature_tensor

    @property
    def nested_coordinates(self):
        return self.batched_coordinates.to_nested()

    @property
    def nested_features(self):
        return self.batched_features.to_nested()

    # ---- metadata -----------------------------------------------------------------------------
    @property
    def num_spatial_dims(self) -> int:
        return self.batched_coordinates.num_spatial_dims

    @property
    def offsets(self) -> Tensor:
        return self.batched_features.offsets

    @property
    def device(self):
        return self.batched_features.device

    @property
    def num_channels(self) -> int:
        return self.batched_features.num_channels

    @property
    def batch_size(self) -> int:
        return len(self.offsets) - 1

    @property
    def dtype(self):
        return self.batched_features.dtype

    @property
    def extra_attributes(self) -> Dict[str, Any]:
        return self._extra_attributes.copy()

    @property
    def cache(self):
        return self._extra_attributes.get("_cache")

    def __len__(self) -> int:
        return len(self.batched_coordinates)

    def numel(self):
        return int(self.offsets[-1]) * self.num_channels

    def __getitem__(self, idx: int) -> "Geometry":
        coords = self.batched_coordinates[idx]
        feats = self.batched_features[idx]
        return self.__class__(coords, feats, offsets=torch.tensor([0, len(coords)]), **self._extra_attributes)

    # ---- functional updates -------------------------------------------------------------------
    def replace(self, batched_coordinates: Optional[Coords] = None, batched_features=None, **kwargs) -> "Geometry":
        if (batched_coordinates is None and not kwargs and isinstance(batched_features, Tensor) and batched_features.ndim == 2
                and batched_features.shape[0] == self.batched_features.batched_tensor.shape[0]
                and batched_features.device == self.batched_features.batched_tensor.device):
            # the per-layer case (a module swaps the feature tensor): same coordinates, same offsets, same attributes - the
            # constructor's host-side validation has nothing new to look at (~15 us per call, ~80 calls per MinkUNet forward)
            out = object.__new__(self.__class__)
            out.batched_coordinates = self.batched_coordinates
            out.batched_features = self.batched_features._new(batched_features)
            out._extra_attributes = dict(self._extra_attributes)
            return out
        if "_extra_attributes" in kwargs:
            kwargs = {**kwargs.pop("_extra_attributes"), **kwargs}
        coords = batched_coordinates if batched_coordinates is not None else self.batched_coordinates
        feats = batched_features if batched_features is not None else self.batched_features
        if isinstance(feats, Tensor):
            feats = to_batched_features(feats, coords.offsets)
        return self.__class__(coords, feats, **{**self._extra_attributes, **kwargs})

    def replace_features(self, new_features) -> "Geometry":
        return self.replace(batched_features=new_features)

    def to(self, device=None, dtype: Optional[torch.dtype] = None) -> "Geometry":
        if isinstance(device, torch.dtype):
            device, dtype = None, device
        if device is None:
            device = self.device
        extra = self._extra_attributes
        if torch.device(device) != torch.device(self.device):
            # cached kernel maps / spatial caches hold tensors (and raw table pointers) of the old device
            extra = {k: v for k, v in extra.items() if k not in ("_cache", "_spatial_cache")}
        return self.__class__(
            self.batched_coordinates.to(device=device),
            self.batched_features.to(device=device, dtype=dtype),
            **extra,
        )

    def _apply_feature_transform(self, fn) -> "Geometry":
        return self.replace(batched_features=fn(self.feature_tensor))

    def half(self):
        return self._apply_feature_transform(lambda x: x.half())

    def float(self):
        return self._apply_feature_transform(lambda x: x.float())

    def double(self):
        return self._apply_feature_transform(lambda x: x.double())

    def to_cat(self) -> "Geometry":
        return self

    # ---- arithmetic on features -----------------------------------------------------------------
    def equal_shape(self, value) -> bool:
        if isinstance(value, Geometry):
            return self.batched_coordinates.equal_shape(value.batched_coordinates) and self.batched_features.equal_shape(
                value.batched_features
            )
        return isinstance(value, Tensor) and tuple(value.shape) == tuple(self.batched_features.batched_tensor.shape)

    def binary_op(self, value, op: str) -> "Geometry":
        if isinstance(value, Geometry):
            assert self.equal_shape(value), f"Shapes do not match. {self} != {value}"
            return self._apply_feature_transform(lambda x: getattr(x, op)(value.feature_tensor))
        if isinstance(value, (int, float)) or (torch.is_tensor(value) and value.numel() == 1):
            return self._apply_feature_transform(lambda x: getattr(x, op)(value))
        if isinstance(value, Tensor):
            assert self.equal_shape(value)
            return self._apply_feature_transform(lambda x: getattr(x, op)(value))
        raise NotImplementedError

    def __add__(self, v): return self.binary_op(v, "__add__")
    def __sub__(self, v): return self.binary_op(v, "__sub__")
    def __mul__(self, v): return self.binary_op(v, "__mul__")
    def __truediv__(self, v): return self.binary_op(v, "__truediv__")
    def __floordiv__(self, v): return self.binary_op(v, "__floordiv__")
    def __mod__(self, v): return self.binary_op(v, "__mod__")
    def __pow__(self, v): return self.binary_op(v, "__pow__")
    def __radd__(self, v): return self.binary_op(v, "__add__")
    def __rmul__(self, v): return self.binary_op(v, "__mul__")
    def __rsub__(self, v): return self._apply_feature_transform(lambda x: -x).binary_op(v, "__add__")
    def __rtruediv__(self, v): return self._apply_feature_transform(lambda x: x.reciprocal()).binary_op(v, "__mul__")
    def __neg__(self): return self._apply_feature_transform(lambda x: -x)

    def __str__(self) -> str:
        return (
            f"{self.__class__.__name__}(feature_shape={tuple(self.batched_features.shape)}, "
            f"coords_shape={tuple(self.batched_coordinates.shape)})"
        )

    def __repr__(self) -> str:
        extra = {k: v for k, v in self._extra_attributes.items() if v is not None and not k.startswith("_")}
        tail = "".join(f", {k}={v}" for k, v in extra.items())
        return (
            f"{self.__class__.__name__}(offsets={self.offsets.tolist()}, feature_shape={tuple(self.batched_features.shape)}, "
            f"coords_shape={tuple(self.batched_coordinates.shape)}, device={self.device}, dtype={self.dtype}{tail})"
        )
