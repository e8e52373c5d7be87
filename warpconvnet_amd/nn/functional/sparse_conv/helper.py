"""Orchestration of one sparse convolution: output coordinates, kernel-map cache, dtype policy, GEMM dispatch.

Counterpart of `warpconvnet/nn/functional/sparse_conv/helper.py:147-567` (``spatially_sparse_conv``,
``generate_output_coords_and_kernel_map``) with the same argument surface, incl. generative convolution,
``REDUCE_AND_STRIDE`` (pool, then convolve at stride 1) and Morton orderings of the output rows.
"""
from enum import Enum
from typing import List, Optional, Tuple, Union

import math

import torch

from warpconvnet_amd.utils.compile_guard import eager_unless_compiling
from torch import Tensor

from warpconvnet_amd.constants import get_fp16_accum
from warpconvnet_amd.geometry.base.geometry import Geometry
from warpconvnet_amd.geometry.coords.integer import IntCoords
from warpconvnet_amd.geometry.coords.ops.stride import stride_coords
from warpconvnet_amd.geometry.coords.search.cache import IntSearchCache, IntSearchCacheKey
from warpconvnet_amd.geometry.coords.search.search_results import IntSearchResult
from warpconvnet_amd.geometry.coords.search.torch_discrete import generate_kernel_map
from warpconvnet_amd.geometry.types.voxels import Voxels
from warpconvnet_amd.utils.ntuple import device_const_i32, ntuple

from .detail.unified import (
    SPARSE_CONV_AB_ALGO_MODE,
    SPARSE_CONV_ATB_ALGO_MODE,
    UnifiedSpatiallySparseConvFunction,
)


class STRIDED_CONV_MODE(Enum):
    REDUCE_AND_STRIDE = "reduce_and_stride"
    STRIDE_ONLY = "stride_only"


def _swap(kernel_map: IntSearchResult) -> IntSearchResult:
    """Transposed convolution uses the forward map with in/out exchanged (reference helper.py:487-497)."""
    swapped = IntSearchResult(in_maps=kernel_map.out_maps, out_maps=kernel_map.in_maps, offsets=kernel_map.offsets)
    swapped._offsets_dev = kernel_map._offsets_dev
    swapped._twin = kernel_map  # (its gather tables are the forward map's reverse tables and vice versa: torch_discrete.py)
    return swapped


def _cpu_kernel_map(in_coords: Tensor, out_coords: Tensor, stride, kernel_size, kernel_dilation) -> IntSearchResult:
    raise RuntimeError(
        "kernel-map construction needs GPU coordinates (HIP path, no CPU fallback). Move the Voxels to a GPU, or "
        "attach a pre-built IntSearchResult to the input's cache."
    )


@torch.no_grad()
@eager_unless_compiling
def generate_output_coords_and_kernel_map(
    input_sparse_tensor: Voxels,
    kernel_size: Tuple[int, ...],
    kernel_dilation: Tuple[int, ...],
    stride: Tuple[int, ...],
    generative: bool = False,
    transposed: bool = False,
    output_spatially_sparse_tensor: Optional[Voxels] = None,
    stride_mode: STRIDED_CONV_MODE = STRIDED_CONV_MODE.STRIDE_ONLY,
    order=None,
    kernel_search_batch_size: Optional[int] = None,
    out_code_backend: Optional[str] = None,
    need_pairs: bool = True,
    optimistic: bool = False,
) -> Tuple[Tensor, Tensor, IntSearchResult]:
    """Returns ``(batch_indexed_out_coords [M, D+1], out_offsets (CPU), kernel_map)``.  ``need_pairs``: the caller will
    run a weight gradient on the map (pair lists written with the build); False defers them to their first use."""
    bcoords_in = input_sparse_tensor.batch_indexed_coordinates
    if bcoords_in.dtype != torch.int32:
        bcoords_in = bcoords_in.to(torch.int32)
    map_stride = stride  # in -> out ratio used for the kernel map

    if output_spatially_sparse_tensor is not None:
        assert not generative, "Output spatially sparse tensor is not supported with generative convolution"
        bcoords_out = output_spatially_sparse_tensor.batch_indexed_coordinates.to(torch.int32)
        out_offsets = output_spatially_sparse_tensor.offsets
    elif generative:
        # reference `_apply_generative_policy` (helper.py:58-146): expand the (strided / up-scaled) coordinates
        from warpconvnet_amd.geometry.coords.ops.expand import expand_coords

        if all(s == 1 for s in stride):
            base = bcoords_in
        elif transposed:  # up-sampling + densification: scale the coordinates, then expand; map built at stride 1
            scale = device_const_i32([1, *stride], bcoords_in.device)
            base = bcoords_in * scale
            bcoords_in = base
        else:             # stride first, then expand; the map still pairs the original inputs with the outputs
            base, _ = stride_coords(bcoords_in, stride)
        bcoords_out, out_offsets = expand_coords(base, kernel_size, kernel_dilation)
        if transposed:
            map_stride = tuple(1 for _ in stride)
    elif any(s != 1 for s in stride):
        # (with the cell table of this level's submanifold layers on the coordinate tensor: no hash table, and for
        # kernel_size == stride the kernel map comes out of the same pass - geometry/coords/ops/stride.py)
        bcoords_out, out_offsets = stride_coords(
            bcoords_in, stride, num_batches=len(input_sparse_tensor.offsets) - 1,
            with_map=(not transposed and tuple(kernel_size) == tuple(stride) and all(d == 1 for d in kernel_dilation)))
    else:
        bcoords_out, out_offsets = bcoords_in, input_sparse_tensor.offsets

    # optional space-filling-curve order of the OUTPUT rows (reference helper.py:435-443): every batch element is sorted
    # on its own, so the offsets stay valid
    from warpconvnet_amd.geometry.coords.ops.serialization import POINT_ORDERING, encode, to_point_ordering

    order = to_point_ordering(order)
    if order != POINT_ORDERING.RANDOM and bcoords_out.shape[0] > 0:
        perm = encode(bcoords_out[:, 1:], batch_offsets=out_offsets, order=order, return_perm=True).perm
        bcoords_out = bcoords_out[perm].contiguous()

    # (the reference's key ignores the ordering, so an ordered and an unordered layer on the same tensor would share one
    # map and one of them would read rows of the other's order; here the ordering is part of the key)
    mode_key = str(stride_mode) if order == POINT_ORDERING.RANDOM else f"{stride_mode}|{order.value}"
    key = IntSearchCacheKey(kernel_size, kernel_dilation, transposed, generative, mode_key, False,
                            input_sparse_tensor.offsets, out_offsets)
    if input_sparse_tensor.cache is not None:
        hit = input_sparse_tensor.cache.get(key)
        if hit is not None:
            return bcoords_out, out_offsets, hit

    if transposed:
        # the forward (fine -> coarse) map may sit in either tensor's cache; reuse it swapped
        fwd_key = IntSearchCacheKey(kernel_size, kernel_dilation, False, generative, str(stride_mode), False,
                                    out_offsets, input_sparse_tensor.offsets)
        for source in (input_sparse_tensor, output_spatially_sparse_tensor):
            if source is not None and source.cache is not None:
                fwd = source.cache.get(fwd_key)
                if fwd is not None:
                    return bcoords_out, out_offsets, _swap(fwd)
        kernel_map = _swap(generate_kernel_map(bcoords_out, bcoords_in, map_stride, kernel_size, kernel_dilation,
                                               need_pairs=need_pairs))
    else:
        # optimistic (the convolution passes it): the caller queues its forward kernel on the map's tables and THEN calls
        # kernel_map.validate() - no host round trip between the map build and the forward
        kernel_map = generate_kernel_map(bcoords_in, bcoords_out, map_stride, kernel_size, kernel_dilation, need_pairs=need_pairs,
                                         optimistic=optimistic)

    if input_sparse_tensor.cache is None:
        input_sparse_tensor._extra_attributes["_cache"] = IntSearchCache()
    cache = input_sparse_tensor.cache
    cache.put(key, kernel_map)
    if getattr(kernel_map, "_validate_fn", None) is not None:
        # an optimistic map is cached before its status word is read: if the device rejects the build for good, validate()
        # raises (every time) and the entry goes away, so a retry on the same tensor builds - and fails - afresh
        kernel_map._on_invalid = lambda: cache.evict(key, kernel_map)
    return bcoords_out, out_offsets, kernel_map


@eager_unless_compiling
def spatially_sparse_conv(
    input_sparse_tensor: Geometry,
    weight: Tensor,
    kernel_size: Union[int, List[int], Tuple[int, ...]],
    stride: Union[int, List[int], Tuple[int, ...]] = 1,
    kernel_dilation: Union[int, List[int], Tuple[int, ...]] = 1,
    bias: Optional[Tensor] = None,
    groups: int = 1,
    use_fp16_accum: Optional[bool] = None,
    kernel_matmul_batch_size: int = 2,
    generative: bool = False,
    output_spatially_sparse_tensor: Optional[Geometry] = None,
    transposed: bool = False,
    fwd_algo=SPARSE_CONV_AB_ALGO_MODE.AUTO,
    dgrad_algo=SPARSE_CONV_AB_ALGO_MODE.AUTO,
    wgrad_algo=SPARSE_CONV_ATB_ALGO_MODE.AUTO,
    stride_mode: STRIDED_CONV_MODE = STRIDED_CONV_MODE.STRIDE_ONLY,
    stride_reduce: str = "max",
    order=None,
    compute_dtype: Optional[torch.dtype] = None,
    implicit_matmul_fwd_block_size: Optional[int] = 16,
    implicit_matmul_bwd_block_size: Optional[int] = 16,
) -> Geometry:
    if not isinstance(input_sparse_tensor, Voxels):
        raise TypeError(f"spatially_sparse_conv expects input_sparse_tensor of type Voxels, got {type(input_sparse_tensor)}")
    if output_spatially_sparse_tensor is not None and not isinstance(output_spatially_sparse_tensor, Voxels):
        raise TypeError(
            f"spatially_sparse_conv expects output_spatially_sparse_tensor of type Voxels or None, got {type(output_spatially_sparse_tensor)}"
        )
    nd = input_sparse_tensor.num_spatial_dims
    _kernel_size = ntuple(kernel_size, ndim=nd)
    _dilation = ntuple(kernel_dilation, ndim=nd)
    _stride = ntuple(stride, ndim=nd)

    # 1x1 kernel, stride 1: a plain matmul on the features (reference helper.py:206-213)
    if math.prod(_kernel_size) == 1 and math.prod(_stride) == 1:
        from warpconvnet_amd.nn.functional.sparse_conv.pointwise import pointwise_conv

        feats = input_sparse_tensor.feature_tensor
        if weight.ndim == 3:  # forward / dX: dense products (streaming gather kernel where the shape allows); dW: sparse AtB kernel
            out = pointwise_conv(feats, weight, bias)
        else:
            out = feats @ weight[0].to(feats.dtype)
            if bias is not None:
                out = out + bias.to(out.dtype)
        return input_sparse_tensor.replace(batched_features=out)

    in_tensor_stride = input_sparse_tensor.tensor_stride or ntuple(1, ndim=nd)
    if transposed and not generative:
        assert output_spatially_sparse_tensor is not None, (
            "Output spatially sparse tensor is required for transposed convolution without generative"
        )
    if not transposed:
        out_tensor_stride = tuple(o * s for o, s in zip(_stride, in_tensor_stride))
    elif generative:
        out_tensor_stride = tuple(i // s for i, s in zip(in_tensor_stride, _stride))
    else:
        out_tensor_stride = output_spatially_sparse_tensor.tensor_stride or ntuple(1, ndim=nd)
        assert any(o < i for o, i in zip(out_tensor_stride, in_tensor_stride)), "Output stride is larger than input stride"

    # compute dtype: explicit argument > autocast dtype > feature dtype (reference helper.py:246-254)
    if compute_dtype is not None:
        effective_dtype = compute_dtype
    elif torch.is_autocast_enabled():
        effective_dtype = torch.get_autocast_dtype("cuda")
    else:
        effective_dtype = input_sparse_tensor.feature_tensor.dtype
    if use_fp16_accum is None:
        use_fp16_accum = get_fp16_accum()  # accepted for API parity: MFMA accumulates in fp32 regardless

    # REDUCE_AND_STRIDE: pool the input over stride-sized windows first, then convolve the pooled tensor at stride 1
    # (reference helper.py:273-289); the pooled tensor carries the output tensor stride already
    if stride_mode == STRIDED_CONV_MODE.REDUCE_AND_STRIDE and any(s != 1 for s in _stride):
        from warpconvnet_amd.nn.functional.sparse_pool import sparse_reduce

        input_sparse_tensor = sparse_reduce(input_sparse_tensor, kernel_size=_stride, stride=_stride, reduction=stride_reduce)
        _stride = ntuple(1, ndim=nd)

    bcoords_out, out_offsets, kernel_map = generate_output_coords_and_kernel_map(
        input_sparse_tensor, _kernel_size, _dilation, _stride, generative=generative, transposed=transposed,
        output_spatially_sparse_tensor=output_spatially_sparse_tensor, stride_mode=stride_mode, order=order,
        need_pairs=torch.is_grad_enabled(),  # (the map builders run under no_grad: the caller's mode is read here)
        optimistic=input_sparse_tensor.batched_features.batched_tensor.is_cuda,  # (not feature_tensor: that casts under autocast)
    )
    num_out = bcoords_out.shape[0]
    # (training: the builder has queued the pair-list scatter on a helper stream - right behind the scan, while the
    # neighbour table is still in the Infinity Cache - and the weight gradient joins it on first use of the lists; a map
    # built under no_grad writes them on first use; a forward-only pass never writes them.)

    # cast BEFORE Function.apply so the tensors saved for backward are in compute precision
    feats = input_sparse_tensor.feature_tensor
    if feats.dtype != effective_dtype:
        feats = feats.to(effective_dtype)
    w = weight  # cast to the compute dtype INSIDE the autograd function: the low-precision copy is what gets saved, and the
    #            weight gradient goes back to the fp32 master without a bf16 round trip (fp32 -> bf16 -> fp32, two launches)

    out_feats = UnifiedSpatiallySparseConvFunction.apply(
        feats, w, kernel_map, num_out, fwd_algo, dgrad_algo, wgrad_algo, effective_dtype,
        implicit_matmul_fwd_block_size, implicit_matmul_bwd_block_size, in_tensor_stride,
        {"conv_stride": _stride, "transposed": transposed, "generative": generative,
         "stride_mode": getattr(stride_mode, "value", str(stride_mode))},
        groups, use_fp16_accum, bias,  # bias: fused epilogue + HIP column-sum gradient (reference: helper.py:339-342)
    )

    return wrap_conv_output(input_sparse_tensor, bcoords_out, out_offsets, out_feats, out_tensor_stride)


def wrap_conv_output(input_sparse_tensor: Voxels, bcoords_out: Tensor, out_offsets: Tensor, out_feats: Tensor,
                     out_tensor_stride) -> Voxels:
    """The output geometry of a convolution: the input's coordinate object when the output set is the input set, a new
    ``IntCoords`` (with its batch-indexed form attached) otherwise."""
    if bcoords_out is input_sparse_tensor.batch_indexed_coordinates:
        # same output set (stride-1, non-generative layers): keep the coordinate object - no [N, 3] copy, no new batch-index
        # pass for the next layer, and the next map build recognises "same coordinate tensor" by pointer
        out_coords = input_sparse_tensor.batched_coordinates
    else:
        out_offsets_cpu = out_offsets.cpu().int() if out_offsets.dtype != torch.int32 or out_offsets.device.type != "cpu" else out_offsets
        out_coords = IntCoords(bcoords_out[:, 1:], offsets=out_offsets_cpu)
        if bcoords_out.dtype == torch.int32 and bcoords_out.is_contiguous():
            out_coords.__dict__["_bcoords"] = bcoords_out  # (base/coords.py: the cached batch-indexed form)
    if out_coords is input_sparse_tensor.batched_coordinates and input_sparse_tensor.tensor_stride == out_tensor_stride:
        return input_sparse_tensor.replace(batched_features=out_feats)  # nothing but the features changed: no re-validation
    return input_sparse_tensor.replace(
        batched_coordinates=out_coords,
        batched_features=out_feats,
        tensor_stride=out_tensor_stride,
    )
