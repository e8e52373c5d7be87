"""Kernel-map cache carried on ``Voxels`` (reference `geometry/coords/search/cache.py:94-163`).

Keyed on kernel / dilation / flags and the HOST offsets of input and output (not on coordinates; the
reference accepts the aliasing this allows).
"""
from typing import Optional, Tuple

from torch import Tensor

from .search_results import IntSearchResult


_OFFSETS_MEMO = {}  # id(offsets tensor) -> (tensor, int32 CPU copy, tuple): the layers of a network pass the SAME offsets object


def _offsets_key(t: Tensor):
    """(int32 CPU tensor, tuple of its values) of an offsets tensor, memoised per tensor OBJECT: a key is built twice per
    convolution call and the conversion (detach + cpu + int + tolist) was most of its cost.  The memo holds the tensor, so the
    id stays valid; offsets tensors are immutable by convention (every geometry operation makes a new one)."""
    hit = _OFFSETS_MEMO.get(id(t))
    if hit is not None and hit[0] is t:
        return hit[1], hit[2]
    if len(_OFFSETS_MEMO) > 256:
        _OFFSETS_MEMO.clear()
    c = t.detach().cpu().int()
    tup = tuple(c.tolist())
    _OFFSETS_MEMO[id(t)] = (t, c, tup)
    return c, tup


class IntSearchCacheKey:
    def __init__(self, kernel_size, kernel_dilation, transposed, generative, stride_mode,
                 skip_symmetric_kernel_map, in_offsets: Tensor, out_offsets: Tensor):
        self.kernel_size = tuple(kernel_size)
        self.kernel_dilation = tuple(kernel_dilation)
        self.transposed = bool(transposed)
        self.generative = bool(generative)
        self.stride_mode = str(stride_mode)
        self.skip_symmetric_kernel_map = bool(skip_symmetric_kernel_map)
        self.in_offsets, in_t = _offsets_key(in_offsets)
        self.out_offsets, out_t = _offsets_key(out_offsets)
        self._tuple = (
            self.kernel_size, self.kernel_dilation, self.transposed, self.generative, self.stride_mode,
            self.skip_symmetric_kernel_map, in_t, out_t,
        )
        self._hash = hash(self._tuple)

    def __hash__(self):
        return self._hash

    def __eq__(self, other):
        return isinstance(other, IntSearchCacheKey) and self._tuple == other._tuple

    def __repr__(self):
        return (
            f"IntSearchCacheKey(kernel_size={self.kernel_size}, kernel_dilation={self.kernel_dilation}, "
            f"transposed={self.transposed}, generative={self.generative}, stride_mode={self.stride_mode}, "
            f"num_in={int(self.in_offsets[-1])}, num_out={int(self.out_offsets[-1])})"
        )


class IntSearchCache(dict):
    def get(self, key: IntSearchCacheKey) -> Optional[IntSearchResult]:
        return super().get(key, None)

    def put(self, key: IntSearchCacheKey, value: IntSearchResult):
        super().__setitem__(key, value)

    def evict(self, key: IntSearchCacheKey, value: Optional[IntSearchResult] = None) -> None:
        """Drop `key` (only if it still maps to `value`, when given): a map whose optimistic build the device rejected must
        not be served again - the reference never caches a failed build."""
        if key in self and (value is None or super().get(key) is value):
            super().__delitem__(key)

    def __repr__(self):
        return f"{self.__class__.__name__}({len(self)} keys)"
