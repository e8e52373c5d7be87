from .detail.unified import (
    SPARSE_CONV_AB_ALGO_MODE,
    SPARSE_CONV_ATB_ALGO_MODE,
    UnifiedSpatiallySparseConvFunction,
)
from .helper import STRIDED_CONV_MODE, generate_output_coords_and_kernel_map, spatially_sparse_conv

__all__ = [
    "SPARSE_CONV_AB_ALGO_MODE",
    "SPARSE_CONV_ATB_ALGO_MODE",
    "STRIDED_CONV_MODE",
    "UnifiedSpatiallySparseConvFunction",
    "generate_output_coords_and_kernel_map",
    "spatially_sparse_conv",
]
