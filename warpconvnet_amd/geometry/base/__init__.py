from .batched import BatchedTensor
from .coords import Coords
from .features import Features
from .geometry import Geometry

__all__ = ["BatchedTensor", "Coords", "Features", "Geometry"]
