"""Space-filling-curve ordering of integer coordinates.

Counterpart of `warpconvnet/geometry/coords/ops/serialization.py:22-245` (``POINT_ORDERING``, ``encode``,
``morton_code``).  The codes come from one HIP kernel (`wcn_morton_code`, csrc/kmap.hip) instead of the reference's
`_C.coords.morton_code_{20,16}bit`; the permutation of a batched tensor (`batch_offsets` given) is a segmented sort -
every batch element is ordered on its own and stays in place - done with two stable device sorts (code, then batch
index), which is what `_C.utils.segmented_sort` produces for distinct codes.
"""
from enum import Enum
from typing import NamedTuple, Optional, Union

import torch
from torch import Tensor

from warpconvnet_amd import _lib


class POINT_ORDERING(Enum):
    RANDOM = "random"
    MORTON_XYZ = "morton_xyz"
    MORTON_XZY = "morton_xzy"
    MORTON_YXZ = "morton_yxz"
    MORTON_YZX = "morton_yzx"
    MORTON_ZXY = "morton_zxy"
    MORTON_ZYX = "morton_zyx"


STR2POINT_ORDERING = {
    "random": POINT_ORDERING.RANDOM,
    "morton": POINT_ORDERING.MORTON_XYZ,
    **{o.value: o for o in POINT_ORDERING if o is not POINT_ORDERING.RANDOM},
}

# spatial column that feeds the x / y / z slot of the interleave
POINT_ORDERING_TO_MORTON_PERMUTATIONS = {
    o: ["xyz".index(ch) for ch in o.value.split("_")[1]] for o in POINT_ORDERING if o is not POINT_ORDERING.RANDOM
}


class SerializationResult(NamedTuple):
    """codes [N] int64; perm sorts the rows (sorted = data[perm]); inverse_perm restores them (data = sorted[inverse_perm])."""

    codes: Tensor
    perm: Optional[Tensor] = None
    inverse_perm: Optional[Tensor] = None


def to_point_ordering(order: Union[POINT_ORDERING, str, None]) -> POINT_ORDERING:
    if order is None:
        return POINT_ORDERING.RANDOM
    if isinstance(order, POINT_ORDERING):
        return order
    key = str(order).lower()
    if key not in STR2POINT_ORDERING:
        raise ValueError(f"unknown point ordering {order!r}; one of {sorted(STR2POINT_ORDERING)}")
    return STR2POINT_ORDERING[key]


@torch.no_grad()
def morton_code(coords: Tensor, threads_per_block: int = 256,
                order: Union[POINT_ORDERING, str] = POINT_ORDERING.MORTON_XYZ) -> Tensor:
    """Z-order codes [N] int64 of ``coords`` [N, 3] (x, y, z) or [N, 4] (b, x, y, z), after subtracting the per-column
    minimum.  [N, 3]: 21 bits per axis; [N, 4]: 16 bits per axis below ``b << 48``."""
    order = to_point_ordering(order)
    assert order in POINT_ORDERING_TO_MORTON_PERMUTATIONS, f"Order '{order}' not supported for morton code"
    if coords.shape[0] == 0:
        return torch.empty(0, dtype=torch.int64)
    assert coords.ndim == 2 and coords.shape[1] in (3, 4), "coords must be [N, 3] or [N, 4]"
    if coords.dtype == torch.int32:
        c = coords.contiguous()
        origin = c.min(0).values.contiguous()  # subtracted inside the kernel; stays on the device, no host sync
    else:  # float / int64 grids: normalise first, then truncate - the order the reference uses (serialization.py:211-212)
        c = (coords - coords.min(0).values).to(torch.int32).contiguous()
        origin = None
    _lib.require_gpu_tensor(c, "coords")
    codes = torch.empty(c.shape[0], dtype=torch.int64, device=c.device)
    _lib.check(
        _lib.lib().wcn_morton_code(_lib.ptr(c), c.shape[0], c.shape[1], _lib.ptr(origin),
                                   _lib.i3(POINT_ORDERING_TO_MORTON_PERMUTATIONS[order]), _lib.ptr(codes),
                                   _lib.stream_handle(c.device)),
        "wcn_morton_code",
    )
    return codes


@torch.no_grad()
def segmented_argsort(keys: Tensor, offsets: Tensor) -> Tensor:
    """Permutation [N] int64 that sorts ``keys`` ascending inside every segment [offsets[b], offsets[b+1]) and leaves the
    segments where they are (stable)."""
    n = keys.shape[0]
    perm = torch.sort(keys, stable=True).indices
    if len(offsets) <= 2:
        return perm
    offs = offsets.to(device=keys.device, dtype=torch.int64)
    seg = torch.bucketize(perm, offs[1:-1], right=True)  # segment of every (already code-sorted) row
    assert int(offsets[-1]) == n, f"offsets[-1] ({int(offsets[-1])}) must equal the number of rows ({n})"
    return perm[torch.sort(seg, stable=True).indices]


@torch.no_grad()
def encode(grid_coord: Tensor, batch_offsets: Optional[Tensor] = None,
           order: Union[POINT_ORDERING, str] = POINT_ORDERING.MORTON_XYZ, return_perm: bool = False,
           return_inverse: bool = False) -> Union[Tensor, SerializationResult]:
    """Codes of ``grid_coord`` under ``order``; optionally the sorting permutation (per batch element when
    ``batch_offsets`` is given) and its inverse."""
    order = to_point_ordering(order)
    if grid_coord.shape[0] == 0:
        codes = torch.empty(0, dtype=torch.int64)
    elif order in POINT_ORDERING_TO_MORTON_PERMUTATIONS:
        codes = morton_code(grid_coord, order=order)
    else:  # RANDOM
        codes = torch.randperm(grid_coord.shape[0], device=grid_coord.device)
    if not return_perm and not return_inverse:
        return codes
    if codes.shape[0] == 0:
        empty = torch.empty(0, dtype=torch.int64)
        return SerializationResult(codes, empty if return_perm else None, empty if return_inverse else None)
    perm = segmented_argsort(codes, batch_offsets) if batch_offsets is not None else torch.sort(codes, stable=True).indices
    inverse = None
    if return_inverse:
        inverse = torch.empty_like(perm)
        inverse[perm] = torch.arange(len(perm), device=perm.device)
    return SerializationResult(codes, perm, inverse)
