"""Kernel-map cache carried on ``Voxels`` (reference `geometry/coords/search/cache.py:94-163`).

Keyed on kernel / dilation / flags and the HOST offsets of input and output (not on coordinates; the
reference accepts the aliasing this allows).
"""
from typing import Optional, Tuple

from torch import Tensor

from .search_results import IntSearchResult


class IntSearchCacheKey:
    def __init__(self, kernel_size, kernel_dilation, transposed, generative, stride_mode,
                 skip_symmetric_kernel_map, in_offsets: Tensor, out_offsets: Tensor):
        self.kernel_size = tuple(kernel_size)
        self.kernel_dilation = tuple(kernel_dilation)
        self.transposed = bool(transposed)
        self.generative = bool(generative)
        self.stride_mode = str(stride_mode)
        self.skip_symmetric_kernel_map = bool(skip_symmetric_kernel_map)
        self.in_offsets = in_offsets.detach().cpu().int()
        self.out_offsets = out_offsets.detach().cpu().int()
        self._tuple = (
            self.kernel_size, self.kernel_dilation, self.transposed, self.generative, self.stride_mode,
            self.skip_symmetric_kernel_map, tuple(self.in_offsets.tolist()), tuple(self.out_offsets.tolist()),
        )

    def __hash__(self):
        return hash(self._tuple)

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

    def __repr__(self):
        return f"{self.__class__.__name__}({len(self)} keys)"
