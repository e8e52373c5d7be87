from .points import Points
from .voxels import Voxels

__all__ = ["Points", "Voxels"]
