"""Continuous-search configuration (reference `geometry/coords/search/search_configs.py`)."""
from dataclasses import dataclass
from enum import Enum
from typing import Optional, Union


class RealSearchMode(Enum):
    RADIUS = "radius"
    KNN = "knn"
    VOXEL = "voxel"


@dataclass(frozen=True)
class RealSearchConfig:
    mode: Union[RealSearchMode, str] = RealSearchMode.KNN
    radius: Optional[float] = None
    knn_k: Optional[int] = None
    grid_dim: Optional[int] = None

    def __post_init__(self):
        if isinstance(self.mode, str):
            object.__setattr__(self, "mode", RealSearchMode(self.mode))

    def replace(self, **kw) -> "RealSearchConfig":
        vals = dict(mode=self.mode, radius=self.radius, knn_k=self.knn_k, grid_dim=self.grid_dim)
        vals.update(kw)
        return RealSearchConfig(**vals)
