"""Continuous neighbour search front-end (reference `geometry/coords/search/continuous.py:19-66`)."""
from torch import Tensor

from .knn import batched_knn_search
from .radius import batched_radius_search
from .search_configs import RealSearchConfig, RealSearchMode
from .search_results import RealSearchResult


def neighbor_search(ref_positions: Tensor, ref_offsets: Tensor, query_positions: Tensor, query_offsets: Tensor,
                    search_args: RealSearchConfig) -> RealSearchResult:
    if search_args.mode == RealSearchMode.KNN:
        assert search_args.knn_k is not None, "knn_k must be provided for knn search"
        return RealSearchResult(
            batched_knn_search(ref_positions, ref_offsets, query_positions, query_offsets, search_args.knn_k)
        )
    if search_args.mode == RealSearchMode.RADIUS:
        assert search_args.radius is not None, "Radius must be provided for radius search"
        index, _, split = batched_radius_search(ref_positions, ref_offsets, query_positions, query_offsets,
                                                search_args.radius, search_args.grid_dim)
        return RealSearchResult(index, split)
    raise ValueError(f"search_args.mode {search_args.mode} not supported.")
