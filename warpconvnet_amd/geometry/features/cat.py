"""Concatenated features [N, C] (reference `geometry/features/cat.py:11-33`).

Padded features (`PadFeatures`) belong to the dense-grid / FIGConv families and are out of scope
(SURVEY.md §2a P5).
"""
from typing import Optional

import torch
from torch import Tensor

from warpconvnet_amd.geometry.base.features import Features


class CatFeatures(Features):
    def check(self):
        super().check()
        assert self.batched_tensor.ndim == 2, "Batched tensor must be 2D"
        assert self.batched_tensor.shape[0] == int(self.offsets[-1]), (
            f"Offsets {self.offsets.tolist()} do not match tensor {tuple(self.batched_tensor.shape)}"
        )

    def equal_shape(self, value: object) -> bool:
        return (
            isinstance(value, CatFeatures)
            and bool((self.offsets == value.offsets).all())
            and self.num_channels == value.num_channels
        )


def to_batched_features(features, offsets: Tensor, device: Optional[str] = None) -> CatFeatures:
    """Wrap a raw [N, C] tensor (or pass an existing Features through)."""
    if isinstance(features, Features):
        return features if device is None else features.to(device)
    assert isinstance(features, Tensor) and features.ndim == 2, "features must be [N, C]"
    return CatFeatures(features, offsets, device=device)
