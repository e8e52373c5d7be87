"""``PointConv``: edge MLP over the k nearest (or provided / down-sampled) neighbours, row reduction, output MLP.

Constructor and checks of the reference (`warpconvnet/nn/modules/point_conv.py:36-282`).  What is native here: the neighbour
search (``wcn_knn_grid``, exact grid kNN instead of O(M*N) cdist + topk), and for the default configuration the whole edge
pipeline - gather, edge MLP, reduction over the neighbours - as one HIP kernel per direction (``csrc/pointconv.hip``), so no
``[M*k, C]`` edge tensor exists.  Other configurations (custom edge MLPs, max / several reductions, ragged radius lists,
sinusoidal encodings) compose the same result from ``wcn_segment_reduce`` and library GEMMs.
"""
import warnings
from typing import List, Literal, Optional

import torch
import torch.nn as nn

from warpconvnet_amd.geometry.base.coords import Coords
from warpconvnet_amd.geometry.coords.search.search_configs import RealSearchConfig, RealSearchMode
from warpconvnet_amd.geometry.types.points import Points
from warpconvnet_amd.nn.encodings import SinusoidalEncoding
from warpconvnet_amd.nn.functional.point_conv import fused_edge_supported, fused_point_conv_edge
from warpconvnet_amd.nn.modules.base_module import BaseSpatialModule
from warpconvnet_amd.nn.modules.mlp import MLPBlock
from warpconvnet_amd.ops.reductions import REDUCTIONS, row_reduction


def _get_module_input_channel(module: nn.Module) -> int:
    if isinstance(module, nn.Linear):
        return module.in_features
    if isinstance(module, nn.Sequential):
        return _get_module_input_channel(module[0])
    if isinstance(module, nn.Module):
        for _, child in module.named_children():
            return _get_module_input_channel(child)
    raise ValueError(f"Unsupported module type: {type(module)}")


class PointConv(BaseSpatialModule):
    def __init__(
        self,
        in_channels: int,
        out_channels: int,
        neighbor_search_args: RealSearchConfig,
        pooling_reduction: Optional[REDUCTIONS] = None,
        pooling_voxel_size: Optional[float] = None,
        edge_transform_mlp: Optional[nn.Module] = None,
        out_transform_mlp: Optional[nn.Module] = None,
        mlp_block: nn.Module = MLPBlock,
        hidden_dim: Optional[int] = None,
        channel_multiplier: int = 2,
        use_rel_pos: bool = False,
        use_rel_pos_encode: bool = False,
        pos_encode_dim: int = 32,
        pos_encode_range: float = 4,
        reductions: List[str] = ("mean",),
        out_point_type: Literal["provided", "downsample", "same"] = "same",
        provided_in_channels: Optional[int] = None,
        bias: bool = True,
    ):
        super().__init__()
        assert isinstance(reductions, (tuple, list)) and len(reductions) > 0, (
            f"reductions must be a list or tuple of length > 0, got {reductions}"
        )
        if out_point_type == "provided":
            assert pooling_reduction is None and pooling_voxel_size is None
            assert provided_in_channels is not None, "provided_in_channels must be provided for provided type"
        elif out_point_type == "downsample":
            assert pooling_reduction is not None and pooling_voxel_size is not None, (
                "pooling_reduction and pooling_voxel_size must be provided for downsample type"
            )
            assert provided_in_channels is None, "provided_in_channels must be None for downsample type"
            if (neighbor_search_args.mode == RealSearchMode.RADIUS
                    and neighbor_search_args.radius < pooling_voxel_size * (3**0.5)):
                warnings.warn(
                    f"neighbor search radius {neighbor_search_args.radius} is less than sqrt(3) times the downsample "
                    f"voxel size {pooling_voxel_size}", stacklevel=2)
        elif out_point_type == "same":
            assert pooling_reduction is None and pooling_voxel_size is None, (
                "pooling_reduction and pooling_voxel_size must be None for same type"
            )
            assert provided_in_channels is None, "provided_in_channels must be None for same type"
        if (pooling_reduction is not None and pooling_voxel_size is not None
                and neighbor_search_args.mode == RealSearchMode.RADIUS and pooling_voxel_size > neighbor_search_args.radius):
            raise ValueError(f"downsample_voxel_size {pooling_voxel_size} must be <= radius {neighbor_search_args.radius}")
        assert isinstance(neighbor_search_args, RealSearchConfig)
        self.reductions = reductions
        self.in_channels, self.out_channels = in_channels, out_channels
        self.use_rel_pos, self.use_rel_pos_encode = use_rel_pos, use_rel_pos_encode
        self.out_point_feature_type = out_point_type
        self.neighbor_search_args = neighbor_search_args
        self.pooling_reduction, self.pooling_voxel_size = pooling_reduction, pooling_voxel_size
        self.positional_encoding = SinusoidalEncoding(pos_encode_dim, data_range=pos_encode_range)
        if provided_in_channels is None:
            provided_in_channels = in_channels
        if hidden_dim is None:
            hidden_dim = channel_multiplier * max(out_channels, in_channels)
        if edge_transform_mlp is None:
            edge_in = in_channels + provided_in_channels
            if use_rel_pos_encode:
                edge_in += pos_encode_dim * 3
            elif use_rel_pos:
                edge_in += 3
            edge_transform_mlp = mlp_block(in_channels=edge_in, out_channels=out_channels, hidden_channels=hidden_dim, bias=bias)
        self.edge_transform_mlp = edge_transform_mlp
        self.edge_mlp_in_channels = _get_module_input_channel(edge_transform_mlp)
        if out_transform_mlp is None:
            out_transform_mlp = mlp_block(in_channels=out_channels * len(reductions), out_channels=out_channels,
                                          hidden_channels=hidden_dim, bias=bias)
        self.out_transform_mlp = out_transform_mlp

    def __repr__(self):
        s = f"{self.__class__.__name__}(in_channels={self.in_channels} out_channels={self.out_channels}"
        if self.use_rel_pos_encode:
            s += f" rel_pos_encode={self.use_rel_pos_encode}"
        if self.pooling_reduction is not None:
            s += f" pooling={self.pooling_reduction}"
        if self.neighbor_search_args is not None:
            s += f" neighbor={self.neighbor_search_args}"
        return s + ")"

    def _fused_edge(self, in_pc: Points, query_pc: Points, neighbors):
        """gather -> edge MLP -> reduction as ONE HIP kernel (`nn/functional/point_conv.py`) when the configuration allows it:
        kNN lists (uniform power-of-two length) or ragged lists (radius search, other k) with per-edge query ids, default edge
        MLP (identity or Linear shortcut), one mean / sum reduction, no sinusoidal encoding.  None = take the composed path below (same result, edge tensors in HBM)."""
        nidx = neighbors.neighbor_indices
        if len(self.reductions) != 1 or self.use_rel_pos_encode or nidx.numel() == 0:
            return None
        uniform = nidx.ndim == 2
        k = nidx.shape[1] if uniform else 1
        if uniform and (k & (k - 1)) != 0:  # kNN with a list length that is not a power of two: the ragged form of the kernel
            uniform, k = False, 1
        fin, fq = in_pc.feature_tensor, query_pc.feature_tensor.view(-1, query_pc.num_channels)
        nrel = 3 if self.use_rel_pos else 0
        if torch.is_autocast_enabled():  # the one-kernel path computes in fp32: under autocast the composed path sets the dtypes
            return None
        if nrel and (in_pc.coordinate_tensor.requires_grad or query_pc.coordinate_tensor.requires_grad):
            return None  # the kernel has no coordinate gradient: the composed path differentiates the relative positions
        if not fused_edge_supported(self.edge_transform_mlp, fin, fq, nrel, k, self.reductions[0]):
            return None
        if nidx.ndim == 2:
            counts = in_pc.offsets[1:] - in_pc.offsets[:-1]
            if int(counts.min()) < nidx.shape[1]:  # lists padded with -1 (fewer than k points in a batch element): composed path
                return None
        xyz = (in_pc.coordinate_tensor.view(-1, 3), query_pc.coordinate_tensor.view(-1, 3)) if nrel else (None, None)
        splits = None if uniform else neighbors.neighbor_row_splits
        return fused_point_conv_edge(self.edge_transform_mlp, fin, fq, nidx, k, self.reductions[0], *xyz, row_splits=splits)

    def forward(self, in_pc: Points, query_pc: Optional[Points] = None) -> Points:
        if self.out_point_feature_type == "provided":
            assert query_pc is not None, "query_point_features must be provided for the provided type"
        elif self.out_point_feature_type == "downsample":
            assert query_pc is None
            query_pc = in_pc.voxel_downsample(self.pooling_voxel_size, reduction=self.pooling_reduction)
        elif self.out_point_feature_type == "same":
            assert query_pc is None
            query_pc = in_pc
        cin, cq = in_pc.num_channels, query_pc.num_channels
        assert (cin + cq + self.use_rel_pos_encode * self.positional_encoding.num_channels * 3
                + (not self.use_rel_pos_encode) * self.use_rel_pos * 3 == self.edge_mlp_in_channels), (
            f"input features shape {in_pc.feature_tensor.shape} and query feature shape {query_pc.feature_tensor.shape} "
            f"does not match the edge_transform_mlp input channels {self.edge_mlp_in_channels}"
        )
        neighbors = in_pc.neighbors(query_coords=query_pc.batched_coordinates, search_args=self.neighbor_search_args)
        fused = self._fused_edge(in_pc, query_pc, neighbors)
        if fused is not None:
            return Points(
                batched_coordinates=Coords(batched_tensor=query_pc.coordinate_tensor, offsets=query_pc.offsets),
                batched_features=self.out_transform_mlp(fused),
                **query_pc.extra_attributes,
            )
        idx = neighbors.neighbor_indices.long().view(-1)
        splits = neighbors.neighbor_row_splits
        num_reps = splits[1:] - splits[:-1]
        edge = [in_pc.feature_tensor[idx],
                torch.repeat_interleave(query_pc.feature_tensor.view(-1, cq).contiguous(), num_reps, dim=0)]
        if self.use_rel_pos or self.use_rel_pos_encode:
            rel = in_pc.coordinate_tensor.view(-1, 3)[idx] - torch.repeat_interleave(
                query_pc.coordinate_tensor.view(-1, 3).contiguous(), num_reps, dim=0)
            edge.append(self.positional_encoding(rel) if self.use_rel_pos_encode else rel)
        edge = self.edge_transform_mlp(torch.cat(edge, dim=1))
        out = torch.cat([row_reduction(edge, splits, reduction=r) for r in self.reductions], dim=-1)
        out = self.out_transform_mlp(out)
        return Points(
            batched_coordinates=Coords(batched_tensor=query_pc.coordinate_tensor, offsets=query_pc.offsets),
            batched_features=out,
            **query_pc.extra_attributes,
        )
