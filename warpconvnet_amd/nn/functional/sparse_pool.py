"""Sparse pooling / unpooling over a strided kernel map.

Counterpart of `warpconvnet/nn/functional/sparse_pool.py:25-185` (``sparse_reduce``, ``sparse_max_pool``,
``sparse_avg_pool``, ``sparse_unpool``) and ``global_pool`` (`nn/functional/global_pool.py:30-55`).  The reference turns
the map into CSR with a device sort, gathers ``features[in_maps]`` into a temporary and reduces it with
``torch_scatter.segment_csr``; here the forward is ONE kernel over the map's row-major neighbour table
(`wcn_pool_gather`, csrc/pool.hip) and the backward is output-stationary over the reverse table (sum / mean: the
same kernel on the pre-scaled gradient; max / min: `wcn_pool_select` on the saved arg rows) - no atomics, so both
directions are deterministic.  The map is cached under the same key as a strided convolution's
(`IntSearchCacheKey`, STRIDE_ONLY), so ``sparse_unpool`` and a following transposed convolution find it.
"""
from typing import Optional, Tuple, Union

import torch

from warpconvnet_amd.utils.compile_guard import eager_unless_compiling
from torch import Tensor
from torch.autograd import Function

from warpconvnet_amd import _lib
from warpconvnet_amd.geometry.base.geometry import Geometry
from warpconvnet_amd.geometry.coords.integer import IntCoords
from warpconvnet_amd.geometry.coords.ops.stride import stride_coords
from warpconvnet_amd.geometry.coords.search.cache import IntSearchCache, IntSearchCacheKey
from warpconvnet_amd.geometry.coords.search.search_results import IntSearchResult
from warpconvnet_amd.geometry.coords.search.torch_discrete import attach_tables_from_csr, generate_kernel_map, reverse_tables
from warpconvnet_amd.geometry.types.voxels import Voxels
from warpconvnet_amd.ops.reductions import REDUCTIONS, row_reduction
from warpconvnet_amd.utils.ntuple import ntuple

_OP = {"sum": 0, "mean": 1, "max": 2, "min": 3}


def _pool_gather(x: Tensor, tbl: Tensor, n_out: int, K: int, op: int, want_arg: bool, want_count: bool):
    c = x.shape[1]
    out = torch.empty((n_out, c), dtype=x.dtype, device=x.device)
    arg = torch.empty((n_out, c), dtype=torch.int32, device=x.device) if want_arg else None
    cnt = torch.empty(n_out, dtype=torch.int32, device=x.device) if want_count else None
    _lib.check(
        _lib.lib().wcn_pool_gather(_lib.ptr(x), _lib.ptr(tbl), x.shape[0], n_out, c, K, _lib.dtype_code(x.dtype), op,
                                   _lib.ptr(out), _lib.ptr(arg), _lib.ptr(cnt), _lib.stream_handle(x.device)),
        "wcn_pool_gather",
    )
    return out, arg, cnt


class _SparsePoolFunction(Function):
    """features [N_in, C] -> [N_out, C] over ``kernel_map`` (pairs (in, out)), reduction in {sum, mean, max, min}."""

    @staticmethod
    def forward(ctx, features: Tensor, kernel_map: IntSearchResult, num_out: int, op_name: str) -> Tensor:
        x = features.contiguous()
        _lib.require_gpu_tensor(x, "features")
        if x.dtype not in (torch.float32, torch.float16, torch.bfloat16):
            raise RuntimeError(f"sparse pooling: unsupported feature dtype {x.dtype}")
        attach_tables_from_csr(kernel_map, x.shape[0], num_out)
        op = _OP[op_name]
        K = len(kernel_map)
        out, arg, cnt = _pool_gather(x, kernel_map._nbr, num_out, K, op, want_arg=op >= 2, want_count=op == 1)
        ctx.kernel_map, ctx.op, ctx.num_in, ctx.K = kernel_map, op, x.shape[0], K
        ctx.save_for_backward(*(t for t in (arg, cnt) if t is not None))
        return out

    @staticmethod
    def backward(ctx, grad_out: Tensor):
        kernel_map, op, n_in, K = ctx.kernel_map, ctx.op, ctx.num_in, ctx.K
        g = grad_out.contiguous()
        rev_tbl, _, _ = reverse_tables(kernel_map, n_in)
        if op >= 2:
            (arg,) = ctx.saved_tensors
            dx = torch.empty((n_in, g.shape[1]), dtype=g.dtype, device=g.device)
            _lib.check(
                _lib.lib().wcn_pool_select(_lib.ptr(g), _lib.ptr(arg), _lib.ptr(rev_tbl), n_in, g.shape[0], g.shape[1], K,
                                           _lib.dtype_code(g.dtype), _lib.ptr(dx), _lib.stream_handle(g.device)),
                "wcn_pool_select",
            )
        else:
            if op == 1:
                (cnt,) = ctx.saved_tensors
                g = (g.float() / cnt.clamp_min(1).unsqueeze(1)).to(g.dtype)
            dx, _, _ = _pool_gather(g, rev_tbl, n_in, K, 0, False, False)
        ctx.kernel_map = None
        return dx, None, None, None


def _pool_map(voxels: Voxels, kernel_size: Tuple[int, ...], stride: Tuple[int, ...], order=None):
    """(batch-indexed output coords, CPU offsets, map) of a strided window; cached like a strided convolution's map."""
    from warpconvnet_amd.nn.functional.sparse_conv.helper import STRIDED_CONV_MODE

    nd = voxels.num_spatial_dims
    bcoords_in = voxels.batch_indexed_coordinates
    if bcoords_in.dtype != torch.int32:
        bcoords_in = bcoords_in.to(torch.int32)
    from warpconvnet_amd.geometry.coords.ops.serialization import POINT_ORDERING, to_point_ordering

    bcoords_out, out_offsets = stride_coords(bcoords_in, stride, order=order)
    # same key construction as the convolution (helper.py): the ordering of the OUTPUT rows is part of the key, or an
    # ordered pool and an unordered strided layer on one tensor would share a map whose rows follow the other's order
    order_e = to_point_ordering(order)
    mode_key = str(STRIDED_CONV_MODE.STRIDE_ONLY)
    if order_e != POINT_ORDERING.RANDOM:
        mode_key = f"{mode_key}|{order_e.value}"
    key = IntSearchCacheKey(kernel_size, ntuple(1, nd), False, False, mode_key, False, voxels.offsets, out_offsets)
    kernel_map = voxels.cache.get(key) if voxels.cache is not None else None
    if kernel_map is None:
        kernel_map = generate_kernel_map(bcoords_in, bcoords_out, stride, kernel_size, ntuple(1, nd))
    if voxels.cache is None:
        voxels._extra_attributes["_cache"] = IntSearchCache()
    voxels.cache.put(key, kernel_map)
    return bcoords_out, out_offsets, kernel_map


@eager_unless_compiling
def sparse_reduce(voxels: Voxels, kernel_size: Union[int, Tuple[int, ...]],
                  stride: Optional[Union[int, Tuple[int, ...]]] = None,
                  reduction: Union[REDUCTIONS, str] = REDUCTIONS.MAX, order=None) -> Voxels:
    """Pool the features of every ``kernel_size`` window placed with ``stride``; output coordinates =
    unique ``floor(coords / stride)``, tensor stride multiplied by ``stride``."""
    if isinstance(reduction, str):
        reduction = REDUCTIONS(reduction)
    if reduction.value not in _OP:
        raise NotImplementedError(f"sparse_reduce supports {sorted(_OP)}; got {reduction.value!r}")
    if stride is None:
        stride = kernel_size
    nd = voxels.num_spatial_dims
    stride, kernel_size = ntuple(stride, nd), ntuple(kernel_size, nd)
    in_ts = voxels.tensor_stride or ntuple(1, nd)
    out_ts = tuple(o * s for o, s in zip(stride, in_ts))
    bcoords_out, out_offsets, kernel_map = _pool_map(voxels, kernel_size, stride, order)
    out = _SparsePoolFunction.apply(voxels.feature_tensor, kernel_map, bcoords_out.shape[0], reduction.value)
    return voxels.replace(
        batched_coordinates=IntCoords(bcoords_out[:, 1:], offsets=out_offsets.cpu().int()),
        batched_features=out,
        tensor_stride=out_ts,
    )


def sparse_max_pool(voxels: Voxels, kernel_size, stride=None) -> Voxels:
    return sparse_reduce(voxels, kernel_size, stride, reduction=REDUCTIONS.MAX)


def sparse_avg_pool(voxels: Voxels, kernel_size, stride=None) -> Voxels:
    return sparse_reduce(voxels, kernel_size, stride, reduction=REDUCTIONS.MEAN)


@eager_unless_compiling
def sparse_unpool(pooled_voxels: Voxels, unpooled_voxels: Voxels, kernel_size, stride,
                  concat_unpooled_voxels: bool = False) -> Voxels:
    """Copy every pooled feature back to the fine voxels of its window (the map of the matching ``sparse_reduce`` /
    strided convolution must be in the cache of either tensor, as in the reference `sparse_pool.py:156-171`)."""
    from warpconvnet_amd.nn.functional.sparse_conv.helper import STRIDED_CONV_MODE

    nd = pooled_voxels.num_spatial_dims
    stride, kernel_size = ntuple(stride, nd), ntuple(kernel_size, nd)
    key = IntSearchCacheKey(kernel_size, ntuple(1, nd), False, False, str(STRIDED_CONV_MODE.STRIDE_ONLY), False,
                            unpooled_voxels.offsets, pooled_voxels.offsets)
    kernel_map = None
    for source in (pooled_voxels, unpooled_voxels):
        if kernel_map is None and source.cache is not None:
            kernel_map = source.cache.get(key)
    assert kernel_map is not None, "sparse_unpool: no cached fine->coarse kernel map for this kernel_size / stride"
    # a fine voxel lies in exactly one window when kernel_size == stride: "sum over the windows that contain it" is the
    # copy the reference performs with argsort(in_maps) (`sparse_pool.py:174-180`); its gradient is the matching sum-pool
    rep = _SparsePoolFunction.apply(pooled_voxels.feature_tensor, _swapped(kernel_map, unpooled_voxels.feature_tensor.shape[0]),
                                    unpooled_voxels.feature_tensor.shape[0], "sum")
    if concat_unpooled_voxels:
        rep = torch.cat([unpooled_voxels.feature_tensor, rep], dim=-1)
    return unpooled_voxels.replace(batched_features=rep)


def _swapped(kernel_map: IntSearchResult, num_fine: int) -> IntSearchResult:
    """The coarse->fine view of a fine->coarse map: forward table = the map's reverse table and vice versa (cached)."""
    sw = getattr(kernel_map, "_pool_swapped", None)
    if sw is None:
        rev_tbl, rev_mask, rev_perm = reverse_tables(kernel_map, num_fine)
        sw = IntSearchResult(kernel_map.out_maps, kernel_map.in_maps, kernel_map.offsets)
        sw._nbr, sw._mask, sw._perm = rev_tbl, rev_mask, rev_perm
        sw._offsets_dev = kernel_map._offsets_dev
        sw._rev = (kernel_map._nbr, kernel_map._mask, kernel_map._perm)
        kernel_map._pool_swapped = sw
    return sw


def global_pool(x: Geometry, reduce: str = "max") -> Geometry:
    """One feature row per batch element (coordinates: the zero vector), reference `global_pool.py:30-55`."""
    B, nd = x.batch_size, x.num_spatial_dims
    offsets = torch.arange(B + 1, dtype=torch.int32)
    feats = row_reduction(x.feature_tensor, x.offsets, reduce)
    coords = torch.zeros(B, nd, dtype=x.coordinate_tensor.dtype, device=x.device)
    return x.replace(
        batched_coordinates=x.batched_coordinates.__class__(coords, offsets),
        batched_features=x.batched_features.__class__(feats, offsets),
    )
