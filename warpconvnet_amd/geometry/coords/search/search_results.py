"""Search-result containers.

``IntSearchResult`` keeps the reference's public contract (`warpconvnet/geometry/coords/search/
search_results.py:54-203`): CSR-by-offset ``in_maps`` / ``out_maps`` (device int32), ``offsets`` on the
HOST, ``identity_map_index``, ``__getitem__`` slices, ``to_csr``, ``neighbor_count_per_output``.

Build-specific device tables used by the HIP GEMMs ride along as private attributes:

* ``_nbr [M, kp]``  row-major neighbour table (input row per (output row, offset), -1 if absent)
* ``_mask [M, mw]`` neighbour bitmask, ``_perm [M]`` rows sorted by descending mask
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
        self.in_maps = in_maps
        self.out_maps = out_maps
        self.offsets = offsets_cpu
        self.identity_map_index = identity_map_index
        # build-specific device tables (see module docstring)
        self._nbr: Optional[Tensor] = None
        self._mask: Optional[Tensor] = None
        self._perm: Optional[Tensor] = None
        self._offsets_dev: Optional[Tensor] = None
        self._symmetric: bool = False
        self._rev: Optional[Tuple[Tensor, Tensor, Tensor]] = None
        self._num_in: Optional[int] = None
        self._num_out: Optional[int] = None
        self._pair_table_cache: Optional[Tensor] = None

    # ---- reference container API --------------------------------------------------------
    @torch.no_grad()
    def __getitem__(self, idx: int) -> Tuple[Tensor, Tensor]:
        start, end = int(self.offsets[idx]), int(self.offsets[idx + 1])
        return self.in_maps[start:end], self.out_maps[start:end]

    def __len__(self) -> int:
        return len(self.offsets) - 1

    def __iter__(self):
        for i in range(len(self)):
            yield self[i]

    def __repr__(self):
        return f"{self.__class__.__name__}(len={len(self)}, iden_map={self.identity_map_index})"

    def numel(self, i: int) -> int:
        return int(self.offsets[i + 1] - self.offsets[i])

    @property
    def device(self):
        return self.in_maps.device

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
