"""Search-result containers.

``IntSearchResult`` keeps the reference's public contract (`warpconvnet/geometry/coords/search/
search_results.py:54-203`): CSR-by-offset ``in_maps`` / ``out_maps`` (device int32), ``offsets`` on the
HOST, ``identity_map_index``, ``__getitem__`` slices, ``to_csr``, ``neighbor_count_per_output``.

Build-specific device tables used by the HIP GEMMs ride along as private attributes:

* ``_nbr [M, kp]``  row-major neighbour table (input row per (output row, offset), -1 if absent; every column of a row beyond
  K holds -1 too).  A property: the binned builder writes COMPACT rows instead (``_nbrc [M, 16]``: word 0 = the row's mask,
  words 1 .. popcount = the neighbour rows of its set offsets in ascending k - `csrc/kmap_cells.h`), which the channel-split
  gather GEMMs and the pair scatter read directly; the dense table is expanded from them on first use (`wcn_kmap_densify`) for
  every other consumer.  ``has_tables`` asks without expanding.
* ``_mask [M, mw]`` neighbour bitmask, ``_perm [M]`` rows in tile order (`csrc/mask_sort.h`)
* ``_offsets_dev [K+1]`` device copy of ``offsets``
* ``_symmetric``    True for a submanifold map (same coordinate tensor, stride 1, odd kernel): the
  reverse table needed by dgrad is then ``_nbr`` with the offset index reversed, nothing to build
* ``_rev``          lazily built ``(rev_nbr, rev_mask, rev_perm)`` otherwise
"""
from typing import List, Literal, Optional, Tuple

import torch
from torch import Tensor


class RealSearchResult:
    """Neighbour lists of a continuous search (reference `search_results.py:14-51`)."""

    def __init__(self, *args):
        if len(args) == 2:
            self.neighbor_indices = args[0].long()
            self.neighbor_row_splits = args[1].long()
        elif len(args) == 1:
            knn = args[0]
            assert isinstance(knn, Tensor) and knn.ndim == 2, "expected an [M, k] index tensor"
            M, k = knn.shape
            self.neighbor_indices = knn.long()
            self.neighbor_row_splits = torch.arange(0, M * k + 1, k, device=knn.device, dtype=torch.long)
        else:
            raise ValueError("RealSearchResult takes (indices, row_splits) or a single [M, k] tensor")
        self.neighbor_distances = None

    def to(self, device):
        self.neighbor_indices = self.neighbor_indices.to(device)
        self.neighbor_row_splits = self.neighbor_row_splits.to(device)
        return self

    def __repr__(self):
        return (
            f"{self.__class__.__name__}(neighbor_indices={tuple(self.neighbor_indices.shape)}, "
            f"neighbor_row_splits={tuple(self.neighbor_row_splits.shape)})"
        )


class IntSearchResult:
    """CSR-by-offset kernel map.  ``in_maps`` / ``out_maps`` / ``offsets`` may be *pending*: the builder launches every
    kernel asynchronously, scatters the pairs into worst-case sized buffers and copies ``offsets`` + status flags to
    pinned host memory behind an event; the first access of one of those attributes (or of ``len(in_maps)``) waits for
    the event, validates the flags and narrows the buffers.  The HIP GEMMs never need the host copy (they read the
    device offsets), so the hot path has no host synchronisation at all; ``poll()`` raises a pending error as soon
    as the event has completed without ever blocking."""

    def __init__(
        self,
        in_maps: Tensor,
        out_maps: Tensor,
        offsets: Tensor,
        identity_map_index: Optional[int] = None,
    ):
        offsets_cpu = offsets.detach().cpu()
        assert len(in_maps) == len(out_maps) == int(offsets_cpu[-1]), (
            f"in_maps ({len(in_maps)}), out_maps ({len(out_maps)}) and offsets[-1] ({int(offsets_cpu[-1])}) disagree"
        )
        self._in_maps = in_maps
        self._out_maps = out_maps
        self._offsets = offsets_cpu
        self._lazy_pairs = None
        self._num_offsets = len(offsets_cpu) - 1
        self._init_tables()
        self.identity_map_index = identity_map_index

    def _init_tables(self):
        # build-specific device tables (see module docstring)
        self._nbr_dense: Optional[Tensor] = None
        self._nbrc: Optional[Tensor] = None  # compact rows [M, 16] (binned builder): read by the GEMMs that can, else densified
        self._mask: Optional[Tensor] = None
        self._perm: Optional[Tensor] = None
        self._offsets_dev: Optional[Tensor] = None
        self._symmetric: bool = False
        self._self_exact: bool = False  # symmetric AND no duplicate coordinates: nbr[r][K//2] == r for every row
        self._has_duplicates: bool = False  # submanifold map over repeated coordinates: dgrad goes through the pair lists
        self._dup_symmetric: bool = False   # ... with an odd kernel at stride 1: dgrad on the gather kernels after all
        self._rev: Optional[Tuple[Tensor, Tensor, Tensor]] = None
        self._num_in: Optional[int] = None
        self._num_out: Optional[int] = None
        self._pair_table_cache: Optional[Tensor] = None
        self._validate_fn = None  # optimistic builds: reads the status word, finishes / repeats the build (see validate)
        self._stride_window = False  # kernel_size == stride map from the down-sampling pass: one (output, offset) per input row
        self._validating = False
        self._identity: Optional[int] = None
        self._validate_error: Optional[Exception] = None  # a build the device rejected for good: raised by every validate()
        self._on_invalid = None  # called once when validation fails (the convolution evicts the map from its cache)
        self._twin = None  # a map made by exchanging in / out of another one (transposed convolution): that map - its reverse
        #                    tables are this map's forward tables and vice versa, nothing is rebuilt from the pair lists

    @property
    def has_tables(self) -> bool:
        """Do the device tables of the gather GEMMs exist (dense or compact)?  Never expands anything."""
        return self._nbr_dense is not None or self._nbrc is not None

    @property
    def _nbr(self) -> Optional[Tensor]:
        if self._nbr_dense is None and self._nbrc is not None:
            from warpconvnet_amd import _lib

            c = self._nbrc
            K = self._num_offsets
            dense = torch.empty((c.shape[0], _lib.lib().wcn_kmap_row_pitch(K)), dtype=torch.int32, device=c.device)
            _lib.check(_lib.lib().wcn_kmap_densify(_lib.ptr(c), c.shape[0], K, _lib.ptr(dense), _lib.stream_handle(c.device)),
                       "wcn_kmap_densify")
            self._nbr_dense = dense
        return self._nbr_dense

    @_nbr.setter
    def _nbr(self, value: Optional[Tensor]) -> None:
        self._nbr_dense = value
        self._nbrc = None  # (a new dense table replaces whatever the builder left)

    @classmethod
    def _blank(cls, num_offsets: int, device) -> "IntSearchResult":
        """Builder hook: an empty container the kernel-map builder fills (tables at once, offsets / pair lists / identities
        when the build's status word has been read - `validate`)."""
        self = object.__new__(cls)
        self._in_maps = self._out_maps = None
        self._offsets = None
        self._lazy_pairs = None
        self._device = torch.device(device)
        self._num_offsets = num_offsets
        self._init_tables()
        self.identity_map_index = None
        return self

    @classmethod
    def _from_deferred_pairs(cls, scatter, offsets_host: Tensor, device, identity_map_index: Optional[int]) -> "IntSearchResult":
        """Builder hook: offsets are known, the pair lists are not written yet.  ``scatter()`` allocates and fills
        ``(in_maps, out_maps)`` on first use - the forward / dgrad kernels read the neighbour table, only the weight gradient
        (and the container API) needs the lists, so a forward-only pass never pays for them."""
        self = cls._blank(len(offsets_host) - 1, device)
        self._offsets = offsets_host.detach().cpu()
        self._lazy_pairs = scatter
        self.identity_map_index = identity_map_index
        return self

    @property
    def identity_map_index(self) -> Optional[int]:
        """Offset whose bucket is the identity (output row i pairs with input row i), or None.  Known once the status word of
        an optimistic build has been read, so the getter validates first."""
        if (self._validate_fn is not None or self._validate_error is not None) and not self._validating:
            self.validate()
        return self._identity

    @identity_map_index.setter
    def identity_map_index(self, value: Optional[int]) -> None:
        self._identity = value

    def validate(self) -> bool:
        """Read the status word of an OPTIMISTIC build (`generate_kernel_map(..., optimistic=True)`): raises the build-time
        errors (coordinate range, table capacity), fills offsets / identities / pair lists, and rebuilds the tables if the
        device rejected the first attempt.  Returns True when the device tables were REPLACED - launches made on the old
        ones must be repeated.  Idempotent and free once done; a no-op for every other map."""
        err = getattr(self, "_validate_error", None)
        if err is not None:
            raise err  # the reference raises at every use of a failed build; so does this container
        fn = getattr(self, "_validate_fn", None)
        if fn is None or self._validating:
            return False
        self._validating = True
        try:
            rebuilt = bool(fn(self))
        except Exception as e:
            # the device rejected the build for good (coordinate range, table capacity): the container stays unusable - its
            # tables hold defined "no neighbour" rows, its lists do not exist - and says so every time
            self._validate_fn = None
            self._validate_error = e
            cb, self._on_invalid = getattr(self, "_on_invalid", None), None
            if cb is not None:
                cb()
            raise
        finally:
            self._validating = False
        # (an interrupt - KeyboardInterrupt - leaves _validate_fn in place: the next call settles the same build)
        self._validate_fn = None
        self._on_invalid = None
        return rebuilt

    def _ensure_pairs(self):
        self.validate()
        fn = getattr(self, "_lazy_pairs", None)
        if fn is not None:
            self._lazy_pairs = None
            self._in_maps, self._out_maps = fn()

    def _materialize(self):
        self._ensure_pairs()

    def poll(self):
        """Non-blocking hook kept for callers of earlier builds: nothing to do (see `validate`)."""

    @property
    def in_maps(self) -> Tensor:
        self._materialize()
        return self._in_maps

    @property
    def out_maps(self) -> Tensor:
        self._materialize()
        return self._out_maps

    @property
    def offsets(self) -> Tensor:
        self._materialize()
        return self._offsets

    @property
    def in_maps_device(self) -> Tensor:
        """Pair buffer without forcing the host copy (may be longer than the number of pairs)."""
        self._ensure_pairs()
        return self._in_maps

    @property
    def out_maps_device(self) -> Tensor:
        self._ensure_pairs()
        return self._out_maps

    # ---- reference container API --------------------------------------------------------
    @torch.no_grad()
    def __getitem__(self, idx: int) -> Tuple[Tensor, Tensor]:
        start, end = int(self.offsets[idx]), int(self.offsets[idx + 1])
        return self.in_maps[start:end], self.out_maps[start:end]

    def __len__(self) -> int:
        return self._num_offsets

    def __iter__(self):
        for i in range(len(self)):
            yield self[i]

    def __repr__(self):
        return f"{self.__class__.__name__}(len={len(self)}, iden_map={self.identity_map_index})"

    def numel(self, i: int) -> int:
        return int(self.offsets[i + 1] - self.offsets[i])

    @property
    def device(self):
        if self._in_maps is None:
            return self._device
        return self._in_maps.device

    @torch.no_grad()
    def get_batch(self, start_idx: int, end_idx: int, out_format: Literal["list", "tensor"] = "list"):
        ins = [self[i][0] for i in range(start_idx, end_idx)]
        outs = [self[i][1] for i in range(start_idx, end_idx)]
        if out_format == "list":
            return ins, outs
        if out_format != "tensor":
            raise ValueError(f"Invalid output format: {out_format}")
        width = max(len(t) for t in ins)
        in_t = torch.full((len(ins), width), -1, dtype=torch.int64, device=self.in_maps.device)
        out_t = torch.full((len(ins), width), -1, dtype=torch.int64, device=self.in_maps.device)
        for i, (a, b) in enumerate(zip(ins, outs)):
            in_t[i, : len(a)] = a
            out_t[i, : len(b)] = b
        return in_t, out_t

    @torch.no_grad()
    def to_csr(self) -> Tuple[Tensor, Tensor, Tensor]:
        """(in rows sorted by out row, unique out rows, CPU offsets over the unique out rows)."""
        out_sorted, order = torch.sort(self.out_maps, stable=True)
        uniq, counts = torch.unique_consecutive(out_sorted, return_counts=True)
        offsets = torch.cat([torch.zeros(1, dtype=torch.int32), counts.cpu().cumsum(0).to(torch.int32)])
        return self.in_maps[order], uniq, offsets

    def clone(self) -> "IntSearchResult":
        return IntSearchResult(self.in_maps.clone(), self.out_maps.clone(), self.offsets.clone(), self.identity_map_index)

    @torch.no_grad()
    def neighbor_count_per_output(self, num_out: int) -> Tensor:
        return torch.bincount(self.out_maps.long(), minlength=num_out)

    # ---- reference-layout view of the neighbour table ---------------------------------------
    @property
    def _pair_table(self) -> Optional[Tensor]:
        """[K, M] table in the reference's layout (`torch_discrete.py:289-291`), built on demand."""
        if self._nbr is None:
            return None
        if self._pair_table_cache is None:
            from warpconvnet_amd.geometry.coords.search.torch_discrete import nbr_to_pair_table

            self._pair_table_cache = nbr_to_pair_table(self._nbr, len(self))
        return self._pair_table_cache
