"""Packed 64-bit coordinate hash table on the GPU.

Public surface of the reference's ``PackedHashTable`` (`warpconvnet/geometry/coords/search/
packed_hashmap.py:27-260`, `_packed_base.py:32-134`): ``from_coords``, ``insert``, ``search``,
``unique_index``, ``capacity`` (power of two, >= 2N), ``num_entries``, range validation
(batch in [0, 511], coords in [-131072, 131071] -> ``ValueError``), full table -> ``RuntimeError``.

Differences by design: one interleaved 16-B slot array instead of separate key/value arrays, a single
device status word (one host read instead of five), and duplicates resolve to the SMALLEST row index.
"""
import enum
from typing import Optional, Union

import torch
from torch import Tensor

from warpconvnet_amd import _lib


def _next_power_of_2(n: int) -> int:
    return 1 if n <= 1 else 1 << (int(n) - 1).bit_length()


class SearchMode(enum.IntEnum):
    LINEAR = 0
    DOUBLE_HASH = 1  # accepted for API parity; the table always uses linear probing
    WARP_COOP = 2


class PackedHashTable:
    BATCH_MAX = 511
    COORD_MIN = -131072
    COORD_MAX = 131071

    def __init__(self, capacity: int, device: Union[str, torch.device] = "cuda", use_double_hash: bool = False):
        self._capacity = _next_power_of_2(max(int(capacity), 1))
        self._device = torch.device(device)
        self._slots: Optional[Tensor] = None  # int64 [capacity, 2] == 16-B slots
        self._coords: Optional[Tensor] = None
        self._num_entries = 0

    # ---- properties ----------------------------------------------------------------------
    @property
    def capacity(self) -> int:
        return self._capacity

    @property
    def device(self) -> torch.device:
        return self._slots.device if self._slots is not None else self._device

    @property
    def num_entries(self) -> int:
        return self._num_entries

    @property
    def key_dim(self) -> int:
        return 4

    @property
    def slots_tensor(self) -> Tensor:
        return self._slots

    @property
    def keys_tensor(self) -> Tensor:
        """int64 [capacity] view of the packed keys (0 = empty)."""
        return self._slots[:, 0]

    @property
    def values_tensor(self) -> Tensor:
        """int32 [capacity] view of the stored row indices (-1 = empty)."""
        return self._slots.view(torch.int32)[:, 2]

    @property
    def vector_keys(self) -> Tensor:
        if self._coords is None:
            raise RuntimeError("No coordinates stored. Call insert() first.")
        return self._coords[: self._num_entries]

    # ---- construction -----------------------------------------------------------------------
    @classmethod
    def from_coords(cls, coords: Tensor, device: Union[str, torch.device, None] = None,
                    capacity: Optional[int] = None, use_double_hash: bool = False) -> "PackedHashTable":
        target = torch.device(device) if device is not None else coords.device
        coords = coords.contiguous().to(dtype=torch.int32, device=target)
        cap = capacity if capacity is not None else max(16, coords.shape[0] * 2)
        obj = cls(capacity=cap, device=target, use_double_hash=use_double_hash)
        obj.insert(coords)
        return obj

    def _launch_insert(self, coords: Tensor, status: Tensor) -> None:
        """prepare + insert, asynchronous; flags are OR-ed into ``status[0]``."""
        L = _lib.lib()
        stream = _lib.stream_handle(coords.device)
        self._slots = torch.empty((self._capacity, 2), dtype=torch.int64, device=coords.device)
        _lib.check(L.wcn_hash_prepare(_lib.ptr(self._slots), self._capacity, stream), "wcn_hash_prepare")
        _lib.check(
            L.wcn_hash_insert(_lib.ptr(self._slots), self._capacity, _lib.ptr(coords), coords.shape[0],
                              _lib.ptr(status), stream),
            "wcn_hash_insert",
        )
        self._coords = coords
        self._num_entries = coords.shape[0]

    @staticmethod
    def raise_for_flags(flags: int, num_keys: int, capacity: int) -> None:
        if flags & _lib.WCN_FLAG_COORD_RANGE:
            raise ValueError(
                f"Coordinate out of packed range: batch must be in [0, {PackedHashTable.BATCH_MAX}] and spatial "
                f"coords in [{PackedHashTable.COORD_MIN}, {PackedHashTable.COORD_MAX}]"
            )
        if flags & _lib.WCN_FLAG_TABLE_FULL:
            raise RuntimeError(
                f"PackedHashTable.insert failed: hash table is full (num_keys={num_keys}, capacity={capacity}). "
                "Increase capacity or reduce load factor."
            )

    def insert(self, coords: Tensor) -> None:
        if not coords.is_cuda:
            raise RuntimeError("PackedHashTable lives on the GPU (HIP path, no CPU fallback); got CPU coordinates")
        assert coords.ndim == 2 and coords.shape[1] == 4
        coords = coords.contiguous().to(dtype=torch.int32)
        n = coords.shape[0]
        assert n <= self._capacity // 2, f"num_keys={n} exceeds capacity/2={self._capacity // 2}"
        status = torch.zeros(1, dtype=torch.int32, device=coords.device)
        self._launch_insert(coords, status)
        self.raise_for_flags(int(status.item()), n, self._capacity)  # the single host read

    def search(self, query_coords: Tensor, mode: SearchMode = SearchMode.LINEAR) -> Tensor:
        """int32 [M]: row index of each query coordinate in the inserted tensor, -1 on miss."""
        assert self._slots is not None, "Call insert() first"
        assert query_coords.ndim == 2 and query_coords.shape[1] == 4
        q = query_coords.contiguous().to(dtype=torch.int32, device=self._slots.device)
        out = torch.empty(q.shape[0], dtype=torch.int32, device=q.device)
        _lib.check(
            _lib.lib().wcn_hash_search(_lib.ptr(self._slots), self._capacity, _lib.ptr(q), q.shape[0], _lib.ptr(out),
                                       _lib.stream_handle(q.device)),
            "wcn_hash_search",
        )
        return out

    @property
    def unique_index(self) -> Tensor:
        """Ascending int64 rows that are the first occurrence of their coordinate."""
        assert self._slots is not None, "Call insert() first"
        found = self.search(self._coords)
        rows = torch.arange(found.shape[0], dtype=torch.int32, device=found.device)
        return torch.nonzero(found == rows).squeeze(1)
