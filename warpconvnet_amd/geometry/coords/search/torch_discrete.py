"""Kernel-map generation (the integer half of the SparseConv3d hot path).

Mirrors `warpconvnet/geometry/coords/search/torch_discrete.py:24-56, 296-432` (``kernel_offsets_from_size``,
``generate_kernel_map``).  One call = hash build + probe (neighbour table, masks, per-block counts) +
scan + deterministic per-offset compaction + mask argsort, all on the current HIP stream through the
C-ABI, with ONE host read (offsets + status flags) where the reference does six.
"""
import ctypes
import os
from typing import Literal, Optional, Sequence, Tuple

import numpy as np
import torch

from warpconvnet_amd.utils.compile_guard import eager_unless_compiling
from torch import Tensor

from warpconvnet_amd import _lib
from warpconvnet_amd.geometry.coords.search.cell_handle import attach_cells, cells_of, stride_map_of
from warpconvnet_amd.geometry.coords.search.packed_hashmap import PackedHashTable, _next_power_of_2
from warpconvnet_amd.geometry.coords.search.search_results import IntSearchResult
from warpconvnet_amd.utils.ntuple import ntuple


@torch.no_grad()
def kernel_offsets_from_size(
    kernel_size: Tuple[int, ...],
    kernel_dilation: Tuple[int, ...],
    center_offset: Optional[Tuple[int, ...]] = None,
    device: Optional[torch.device] = None,
) -> Tensor:
    """[K, D+1] int32 offsets, k = (i*ky + j)*kz + l, centre (s-1)//2 for odd s and 0 for even s."""
    assert len(kernel_size) == len(kernel_dilation)
    if center_offset is None:
        center_offset = [(s - 1) // 2 if s % 2 == 1 else 0 for s in kernel_size]
    assert len(center_offset) == len(kernel_size)
    grids = np.indices(tuple(int(s) for s in kernel_size)).reshape(len(kernel_size), -1)  # C order: last axis fastest
    cols = [np.zeros(grids.shape[1], dtype=np.int64)]
    for d in range(len(kernel_size)):
        cols.append((grids[d] - int(center_offset[d])) * int(kernel_dilation[d]))
    out = torch.from_numpy(np.stack(cols, axis=1).astype(np.int32))
    return out.to(device) if device is not None else out


@torch.no_grad()
def nbr_to_pair_table(nbr: Tensor, num_offsets: int) -> Tensor:
    """Row-major neighbour table [M, kp] -> reference layout [K, M] (`cuhash_kernel_map.cu:133`)."""
    M = nbr.shape[0]
    out = torch.empty((num_offsets, M), dtype=torch.int32, device=nbr.device)
    _lib.check(
        _lib.lib().wcn_kmap_transpose(_lib.ptr(nbr), M, num_offsets, _lib.ptr(out), _lib.stream_handle(nbr.device)),
        "wcn_kmap_transpose",
    )
    return out


@torch.no_grad()
def mask_argsort(mask: Tensor, num_offsets: int = 32) -> Tensor:
    """Row order for the gather GEMMs' tiles (`wcn_mask_tile_order`): rows with similar neighbour sets adjacent - for odd
    kernel volumes up to 31 a stable sort by the pair / Gray key of `csrc/mask_sort.h`, else by descending mask word 0."""
    n, mw = mask.shape
    perm = torch.empty(n, dtype=torch.int32, device=mask.device)
    if n == 0:
        return perm
    L = _lib.lib()
    ws_bytes = L.wcn_mask_argsort_workspace(n)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=mask.device)
    _lib.check(
        L.wcn_mask_tile_order(_lib.ptr(mask), mw, int(num_offsets), n, _lib.ptr(perm), _lib.ptr(ws), ws_bytes,
                              _lib.stream_handle(mask.device)),
        "wcn_mask_tile_order",
    )
    return perm


@torch.no_grad()
def argsort_u32_ascending(keys: Tensor, num_bits: int) -> Tensor:
    """Stable ascending argsort of non-negative int32 keys below ``2 ** num_bits`` (int32 permutation) with the in-house LSD
    radix sort (`wcn_mask_argsort` orders DESCENDING, ties in ascending row order: the keys go in complemented) - 3 passes
    for 27 bits where the framework's sort of int64 keys is a block sort + seven merge launches (0.3 ms for 200 k keys)."""
    n = keys.shape[0]
    perm = torch.empty(n, dtype=torch.int32, device=keys.device)
    if n == 0:
        return perm
    num_bits = max(1, min(int(num_bits), 31))
    flipped = (((1 << num_bits) - 1) - keys.to(torch.int32)).contiguous()
    L = _lib.lib()
    ws_bytes = L.wcn_mask_argsort_workspace(n)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=keys.device)
    _lib.check(
        L.wcn_mask_argsort(_lib.ptr(flipped), 1, num_bits, n, _lib.ptr(perm), _lib.ptr(ws), ws_bytes, _lib.stream_handle(keys.device)),
        "wcn_mask_argsort",
    )
    return perm


@torch.no_grad()
def attach_tables_from_csr(kmap: IntSearchResult, num_in: int, num_out: int) -> IntSearchResult:
    """Build nbr / mask / perm for a map that only has its CSR form (swapped or user-made maps).

    Role of the reference's `_build_pair_table` + `_build_mask_and_argsort`
    (`nn/functional/sparse_conv/detail/mask_gemm.py:127-254`).
    """
    kmap.validate()  # an optimistic map another consumer left unvalidated: settle (and possibly rebuild) it first
    if kmap.has_tables:
        return kmap
    twin = kmap._twin
    if twin is not None and not twin._has_duplicates:
        # in / out exchanged (transposed convolution on the cached forward map): this map's gather table is the forward map's
        # reverse table - built once and shared with the forward layer's dgrad - instead of a second pass over the pair lists
        twin.validate()
        if twin.has_tables and twin._num_out == num_in:
            kmap._nbr, kmap._mask, kmap._perm = reverse_tables(twin, num_out)
            kmap._offsets_dev = twin._offsets_dev
            kmap._num_in, kmap._num_out = num_in, num_out
            return kmap
    dev = kmap.in_maps_device.device
    K = len(kmap)
    L = _lib.lib()
    kp, mw = L.wcn_kmap_row_pitch(K), L.wcn_kmap_mask_words(K)
    nbr = torch.empty((num_out, kp), dtype=torch.int32, device=dev)
    mask = torch.empty((num_out, mw), dtype=torch.int32, device=dev)
    offsets_dev = kmap._offsets_dev if kmap._offsets_dev is not None else kmap.offsets.to(device=dev, dtype=torch.int32)
    _lib.check(
        L.wcn_kmap_from_csr(_lib.ptr(kmap.in_maps_device), _lib.ptr(kmap.out_maps_device), _lib.ptr(offsets_dev), K,
                            kmap.in_maps_device.shape[0], num_out, _lib.ptr(nbr), _lib.ptr(mask), _lib.stream_handle(dev)),
        "wcn_kmap_from_csr",
    )
    kmap._nbr, kmap._mask, kmap._offsets_dev = nbr, mask, offsets_dev
    kmap._perm = mask_argsort(mask, K)
    kmap._num_in, kmap._num_out = num_in, num_out
    return kmap


@torch.no_grad()
def reverse_tables(kmap: IntSearchResult, num_in: int) -> Tuple[Tensor, Tensor, Tensor]:
    """(rev_nbr [N_in, kp], rev_mask, rev_perm) for dgrad; cached on the map.

    Role of `_build_reverse_mask_data` (`mask_gemm.py:279-350`).
    """
    kmap.validate()
    twin = kmap._twin
    if kmap._rev is None and twin is not None and not twin._has_duplicates:
        twin.validate()
        if twin.has_tables and twin._num_out == num_in:  # (an exchanged map: the forward map's own tables)
            kmap._rev = (twin._nbr, twin._mask, twin._perm)
    if kmap._rev is None:
        dev = kmap.in_maps_device.device
        K = len(kmap)
        L = _lib.lib()
        kp, mw = L.wcn_kmap_row_pitch(K), L.wcn_kmap_mask_words(K)
        rev_nbr = torch.empty((num_in, kp), dtype=torch.int32, device=dev)
        rev_mask = torch.empty((num_in, mw), dtype=torch.int32, device=dev)
        if kmap._offsets_dev is None:
            kmap._offsets_dev = kmap.offsets.to(device=dev, dtype=torch.int32)
        _lib.check(
            L.wcn_kmap_reverse(_lib.ptr(kmap.in_maps_device), _lib.ptr(kmap.out_maps_device), _lib.ptr(kmap._offsets_dev), K,
                               kmap.in_maps_device.shape[0], num_in, _lib.ptr(rev_nbr), _lib.ptr(rev_mask),
                               _lib.stream_handle(dev)),
            "wcn_kmap_reverse",
        )
        # (the real pair count, not the length of the buffer: in_maps_device may be longer than the lists - an uninitialised tail
        # must never become row ids)
        if (getattr(kmap, "_stride_window", False) and int(kmap.offsets[-1]) == num_in
                and kmap.in_maps_device.shape[0] == num_in):
            # a stride-window map (kernel_size == stride) pairs every input row with exactly one (output row, offset): the
            # reverse masks are one-hot, and the input side of the pair lists - rows grouped by offset, CSR order - already IS
            # a permutation of the input rows in which a tile meets one offset: no sort (3 launches per strided layer)
            rev_perm = kmap.in_maps_device
        else:
            rev_perm = mask_argsort(rev_mask, K)
        kmap._rev = (rev_nbr, rev_mask, rev_perm)
    return kmap._rev


def _first_half(full: IntSearchResult) -> IntSearchResult:
    """Buckets 0 .. K//2 - 1 of a full odd-kernel map as their own result (plain CSR container, identity index K//2)."""
    c = len(full) // 2
    offs = full.offsets[: c + 1].clone()
    n = int(offs[-1])
    return IntSearchResult(full.in_maps_device[:n].clone(), full.out_maps_device[:n].clone(), offs, identity_map_index=c)


_SPIN_POLLS = 4000  # polls of the pinned READY word, ~0.1 us each: up to ~0.4 ms of spinning before the ordinary event wait


class BuildHints:
    """What earlier builds taught about the scenes of this job - sizing guesses only, results never depend on them:

    * ``div``: first-try bound of the binned builder's block table, N / div occupied 8^3 blocks (0.15 GB of workspace per
      million voxels at 16); the device reports TABLE_FULL for sparser scenes and the bound moves to N / 4, then N;
    * ``pairs_per_row``: pairs per output row of the last maps - sizes the pair lists an OPTIMISTIC build writes before the
      host has read the pair count (an underestimate is caught on the host: the pair count exceeds the capacity, the
      kernel wrote nothing past it, and the lists are written again at their exact length).  Remembered PER KERNEL VOLUME: a
      network alternates 3 x 3 x 3 maps (9 - 18 pairs per row) with stride-window ones (4 - 6), and one shared estimate was
      short for every map that followed a sparser kind - four lists written twice per MinkUNet iteration.

    ``generate_kernel_map(..., hints=...)`` takes an explicit object (tests pass fresh ones so that the rebuild branches
    are reached deterministically); the default is one object per process, `default_hints()`."""

    def __init__(self, div: int = 16, pairs_per_row: float = 12.0):
        self.div = int(div)
        self.pairs_per_row = float(pairs_per_row)  # kernel volumes not seen yet
        self._by_volume = {}                       # kernel volume -> pairs per row of its last maps
        self._dense_volumes = set()                # kernel volumes whose scenes overflowed a compact row (> 15 neighbours)
        # what the guesses cost so far: builds the device sent back (block table too small, strict insert, dense rows) and
        # speculative pair lists that were too short and written again at their exact length
        self.stats = {"builds": 0, "rebuilds": 0, "pair_rewrites": 0}

    def compact_rows(self, num_offsets: int) -> bool:
        """First try of a submanifold build: compact 64-B table rows?  (Dense from then on once a scene has had a row with more
        than 15 neighbours - the device reports ROW_OVERFLOW and the build is redone dense.)"""
        return int(num_offsets) not in self._dense_volumes

    def observe_row_overflow(self, num_offsets: int) -> None:
        self._dense_volumes.add(int(num_offsets))

    def reset(self) -> None:
        self.__init__()

    def max_blocks(self, n: int) -> int:
        return max(1024, n // self.div) if self.div > 1 else max(n, 1)

    def pair_capacity(self, rows: int, num_offsets: int) -> int:
        per_row = self._by_volume.get(int(num_offsets), self.pairs_per_row)
        return int(min(num_offsets * rows, rows * per_row * 1.25 + 4096))

    def observe_pairs(self, rows: int, pairs: int, num_offsets: int = 0) -> None:
        if rows > 0:
            seen = pairs / rows
            self.pairs_per_row = seen if seen > self.pairs_per_row else 0.5 * (self.pairs_per_row + seen)
            if num_offsets > 0:
                cur = self._by_volume.get(int(num_offsets))
                self._by_volume[int(num_offsets)] = seen if cur is None or seen > cur else 0.5 * (cur + seen)


_DEFAULT_HINTS = BuildHints()


def default_hints() -> BuildHints:
    return _DEFAULT_HINTS


@eager_unless_compiling
@torch.no_grad()
def generate_kernel_map(
    batch_indexed_in_coords: Tensor,
    batch_indexed_out_coords: Tensor,
    in_to_out_stride_ratio: Tuple[int, ...],
    kernel_size: Tuple[int, ...],
    kernel_dilation: Optional[Tuple[int, ...]] = None,
    kernel_center_offset: Optional[Tuple[int, ...]] = None,
    method: Literal["offset", "size"] = "size",
    skip_symmetric_kernel_map: bool = False,
    need_pairs: bool = True,
    optimistic: bool = False,
    hints: Optional[BuildHints] = None,
    **kwargs,
) -> IntSearchResult:
    """Kernel map between integer coordinate sets: ``in = out * stride + offset[k]``.

    Returns an ``IntSearchResult`` whose buckets are ordered by output row (deterministic).  ``optimistic``: return as
    soon as the build is queued - the device tables (``_nbr / _mask / _perm``) may be used for launches right away, the
    status word (range / capacity errors, pair count, duplicate flags) is read by ``result.validate()``, which every
    host-side accessor calls first and which returns True if it had to rebuild the tables (see the builder's tail).
    """
    dev = batch_indexed_in_coords.device
    assert dev == batch_indexed_out_coords.device
    assert batch_indexed_in_coords.dtype == torch.int32 and batch_indexed_out_coords.dtype == torch.int32
    if not batch_indexed_in_coords.is_cuda:
        raise RuntimeError(
            "generate_kernel_map runs on the GPU through libwcn_hip.so; got CPU coordinates (no CPU fallback)"
        )
    if method not in ("offset", "size"):
        raise ValueError(f"Invalid method: {method}. Choose 'offset', or 'size'.")
    odd_kernel = all(int(k) % 2 == 1 for k in kernel_size)
    if skip_symmetric_kernel_map:  # reference torch_discrete.py:319-326, 383-385
        assert len(batch_indexed_in_coords) == len(batch_indexed_out_coords), (
            "You can only skip symmetric kernel map if the input and output coordinates are the same.")
        assert odd_kernel, "Kernel size must be odd for symmetric skipping."
    if method == "size":  # (reference :403-409; dilation itself IS supported by this build's size path)
        assert kernel_center_offset is None, (
            "Custom kernel_center_offset is not supported with method='size'. Use method='offset' instead.")
    # The reference's `offset` method on an odd kernel over equally sized coordinate sets, and `skip_symmetric_kernel_map`,
    # return only the FIRST HALF of the buckets (offsets 0 .. K//2 - 1, identity_map_index = K//2: the second half is the
    # mirror image; `torch_discrete.py:363-370, 387-401, 211-219`).  Here the full map is built once and cut.
    half_map = (skip_symmetric_kernel_map or method == "offset") and odd_kernel and (
        len(batch_indexed_in_coords) == len(batch_indexed_out_coords))
    if kernel_center_offset is not None:
        # offsets with a custom centre are the default offsets plus a constant: in = out*s + off_default[k] + delta with
        # delta = (c_default - c_custom) * dilation, i.e. the default-centre map of the input coordinates shifted by -delta
        # (row indices unchanged).  The shifted set is a different tensor, so the general (hash) builder runs.
        nd = len(kernel_size)
        dil = ntuple(kernel_dilation if kernel_dilation is not None else 1, nd)
        c_def = [(int(k) - 1) // 2 if int(k) % 2 == 1 else 0 for k in kernel_size]
        delta = [(cd - int(cu)) * int(d) for cd, cu, d in zip(c_def, kernel_center_offset, dil)]
        shift = torch.tensor([0] + delta, dtype=torch.int32, device=dev)
        full = generate_kernel_map(batch_indexed_in_coords - shift, batch_indexed_out_coords, in_to_out_stride_ratio,
                                   kernel_size, kernel_dilation, None, "size", False)
        return _first_half(full) if half_map else full
    if half_map:
        return _first_half(generate_kernel_map(batch_indexed_in_coords, batch_indexed_out_coords, in_to_out_stride_ratio,
                                               kernel_size, kernel_dilation, None, "size", False))
    same_tensor = (
        batch_indexed_in_coords.data_ptr() == batch_indexed_out_coords.data_ptr()
        and batch_indexed_in_coords.shape == batch_indexed_out_coords.shape
    )
    # 2-D convolution: pad [b, x, y] -> [b, x, y, 0] (reference torch_discrete.py:328-342)
    if batch_indexed_in_coords.shape[1] == 3:
        batch_indexed_in_coords = torch.nn.functional.pad(batch_indexed_in_coords, (0, 1), value=0)
        batch_indexed_out_coords = (
            batch_indexed_in_coords if same_tensor else torch.nn.functional.pad(batch_indexed_out_coords, (0, 1), value=0)
        )
        kernel_size = tuple(kernel_size) + (1,)
        in_to_out_stride_ratio = tuple(in_to_out_stride_ratio) + (1,)
        if kernel_dilation is not None:
            kernel_dilation = tuple(kernel_dilation) + (1,)
    assert batch_indexed_in_coords.shape[1] == 4, "expected batch-indexed 3-D coordinates [N, 4]"
    ksize = ntuple(kernel_size, 3)
    stride = ntuple(in_to_out_stride_ratio, 3)
    dilation = ntuple(kernel_dilation if kernel_dilation is not None else 1, 3)
    K = ksize[0] * ksize[1] * ksize[2]

    in_coords = batch_indexed_in_coords.contiguous()
    out_coords = in_coords if same_tensor else batch_indexed_out_coords.contiguous()
    N, M = in_coords.shape[0], out_coords.shape[0]
    L = _lib.lib()
    stream = _lib.stream_handle(dev)
    kp, mw = L.wcn_kmap_row_pitch(K), L.wcn_kmap_mask_words(K)
    nblk = L.wcn_kmap_num_blocks(M)

    # meta[0:K+1] = offsets, meta[K+1] = status flags -> one host read (written to pinned memory by the scan kernel)
    unit_stride = all(s == 1 for s in stride)
    method_env = os.environ.get("WARPCONVNET_AMD_KMAP_METHOD", "auto").strip().lower()
    use_binned = (
        method_env != "hash" and same_tensor and unit_stride and N > 0
        and bool(L.wcn_kmap_binned_supported(_lib.i3(ksize), _lib.i3(dilation)))
    )
    if method_env == "binned" and not use_binned and N > 0:
        raise RuntimeError("WARPCONVNET_AMD_KMAP_METHOD=binned needs a submanifold map (same coordinate tensor, stride 1, "
                           "halo <= 8, K % 32 != 0)")
    hints = hints if hints is not None else _DEFAULT_HINTS
    table_capacity = _next_power_of_2(max(16, 2 * N))
    # binned path: capacity of the block table.  Most scenes have >= 4 voxels per occupied 8^3 block (uniform 12 % occupancy:
    # 28, surfaces: ~64); sparser ones raise TABLE_FULL on the device and are rebuilt with one block per voxel (always
    # enough).  `strict`: see wcn.h.
    # compact table rows (csrc/kmap_cells.h): the binned builder's own format where the kernel volume allows (17 <= K <= 31)
    compact_ok = use_binned and bool(L.wcn_kmap_compact_supported(K)) and os.environ.get("WARPCONVNET_AMD_KMAP_COMPACT", "1") != "0"
    state = {"max_blocks": hints.max_blocks(N), "strict": 0, "compact": bool(compact_ok and hints.compact_rows(K))}
    odd = all(k % 2 == 1 for k in ksize)
    # general maps: the cell table an earlier (validated) submanifold build left on the input coordinate tensor, or the
    # table the down-sampling pass wrote next to the output coordinates
    cells = prebuilt = None
    if not use_binned and N > 0 and M > 0 and os.environ.get("WARPCONVNET_AMD_KMAP_METHOD", "auto").strip().lower() != "hash":
        cells = cells_of(batch_indexed_in_coords)  # ignored once the tensor has been edited in place (`cell_handle.py`)
        pm = stride_map_of(batch_indexed_out_coords, batch_indexed_in_coords, in_coords)
        if (pm is not None and pm[0] == tuple(stride) == tuple(ksize) and all(d == 1 for d in dilation)
                and pm[1].shape[0] == M):
            prebuilt = (pm[1], pm[2])

    def launch(spec_capacity: int = 0):
        """Queue one build (tables + tally + scans + mask sort) on the current stream; nothing waits.  ``spec_capacity`` > 0: the
        pair lists too, at that (speculative) capacity, INSIDE the sort's launches (`wcn_kmap_tally_sort`)."""
        max_blocks, strict, compact = state["max_blocks"], state["strict"], state["compact"]
        nbr = torch.empty((M, 16 if compact else kp), dtype=torch.int32, device=dev)
        mask = torch.empty((M, mw), dtype=torch.int32, device=dev)
        block_counts = torch.empty(L.wcn_kmap_counts_bytes(M, K) // 4, dtype=torch.int32, device=dev)
        perm = torch.empty(M, dtype=torch.int32, device=dev)
        meta = (torch.empty if use_binned else torch.zeros)(K + 2, dtype=torch.int32, device=dev)
        table, bin_ws = None, None
        stream = _lib.stream_handle(dev)
        if use_binned:
            # LDS-binned path: block-level hash + cell table + per-block LDS grid probes (csrc/kmap_binned.hip)
            ws_bytes = L.wcn_kmap_binned_workspace(N, max_blocks)
            bin_ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
            _lib.check(
                L.wcn_kmap_build_binned(_lib.ptr(in_coords), N, _lib.i3(ksize), _lib.i3(dilation), max_blocks, strict, int(compact),
                                        _lib.ptr(bin_ws), ws_bytes, _lib.ptr(nbr), _lib.ptr(mask), _lib.ptr(meta[K + 1 :]),
                                        stream),
                "wcn_kmap_build_binned",
            )
        elif prebuilt is not None:
            # strided layer whose kernel is its stride window: the down-sampling pass (coords/ops/stride.py) read these very
            # cells and wrote the table with the output coordinates
            nbr, mask = prebuilt
        elif cells is not None:
            # the input set has the cell table of its submanifold layers (validated): every probe is a block-table lookup +
            # one cell read, nothing is inserted (csrc/kmap_stride.hip)
            _lib.check(
                L.wcn_kmap_probe_cells(_lib.ptr(cells[0]), N, cells[2], _lib.ptr(out_coords), M, _lib.i3(ksize), _lib.i3(stride),
                                       _lib.i3(dilation), _lib.ptr(nbr), _lib.ptr(mask), stream),
                "wcn_kmap_probe_cells",
            )
        else:
            table = PackedHashTable(max(16, 2 * N), device=dev)
            table._launch_insert(in_coords, meta[K + 1 :])
            _lib.check(
                L.wcn_kmap_probe(_lib.ptr(table.slots_tensor), table.capacity, _lib.ptr(out_coords), M, _lib.i3(ksize),
                                 _lib.i3(stride), _lib.i3(dilation), _lib.ptr(nbr), _lib.ptr(mask), stream),
                "wcn_kmap_probe",
            )
        # tally (pair counts per tile + first sort digit + duplicate repair) -> scans (offsets + status flags written to
        # pinned host memory by the kernel itself, no copy command) -> mask argsort; the sort does not depend on the pair
        # count and keeps the GPU busy during the host round trip
        meta_host = torch.empty(K + 3, dtype=torch.int32, pin_memory=True)
        ready = ctypes.c_int32.from_address(meta_host.data_ptr() + 4 * (K + 2))
        ready.value = 0
        sort_bytes = L.wcn_kmap_tally_sort_workspace(M)
        sort_ws = torch.empty(sort_bytes, dtype=torch.uint8, device=dev)
        spec_capacity = max(int(spec_capacity), 0) if M > 0 else 0
        spec_in = torch.empty(spec_capacity, dtype=torch.int32, device=dev) if spec_capacity else None
        spec_out = torch.empty(spec_capacity, dtype=torch.int32, device=dev) if spec_capacity else None
        _lib.check(
            L.wcn_kmap_tally_sort(_lib.ptr(mask), _lib.ptr(nbr), M, K, _lib.ptr(block_counts), _lib.ptr(meta),
                                  _lib.ptr(meta[K + 1 :]), ctypes.c_void_p(meta_host.data_ptr()), _lib.ptr(perm),
                                  _lib.ptr(sort_ws), sort_bytes, _lib.ptr(in_coords) if use_binned else None,
                                  _lib.ptr(bin_ws), N if use_binned else 0, max_blocks if use_binned else 0, int(compact),
                                  _lib.ptr(spec_in), _lib.ptr(spec_out), spec_capacity, stream),
            "wcn_kmap_tally_sort",
        )
        event = torch.cuda.Event()
        event.record(torch.cuda.current_stream(dev))
        return dict(nbr=nbr, mask=mask, perm=perm, block_counts=block_counts, meta=meta, meta_host=meta_host, ready=ready,
                    event=event, table=table, keep=(bin_ws, sort_ws, cells), max_blocks=max_blocks, compact=compact,
                    spec_pairs=(spec_in, spec_out, spec_capacity) if spec_capacity else None)

    def scatter(b, capacity):
        """Pair lists of build `b` (buckets ordered by output row), `capacity` entries each; the kernel writes nothing past
        the capacity."""
        capacity = max(int(capacity), 0)
        in_maps = torch.empty(capacity, dtype=torch.int32, device=dev)
        out_maps = torch.empty(capacity, dtype=torch.int32, device=dev)
        _lib.check(
            L.wcn_kmap_scatter(_lib.ptr(b["nbr"]), _lib.ptr(b["mask"]), M, K, _lib.ptr(b["block_counts"]), _lib.ptr(b["meta"]),
                               _lib.ptr(in_maps), _lib.ptr(out_maps), capacity, _lib.ptr(b["meta"][K + 1 :]), int(b["compact"]),
                               _lib.stream_handle(dev)),
            "wcn_kmap_scatter",
        )
        return in_maps, out_maps, capacity

    def settle(b):
        """Wait for the status word of build `b`; rebuild while the device asks for it.  -> (build, flags, rebuilt)"""
        rebuilt = False
        while True:
            # the scan kernel raises the READY word of the pinned mirror behind a system-scope fence: spinning on it costs a
            # few microseconds where the event wait costs a thread wake-up; bounded, then the ordinary wait (long queues, M = 0)
            if M > 0:
                for _ in range(_SPIN_POLLS):
                    if b["ready"].value:
                        break
            if not b["ready"].value:
                b["event"].synchronize()
            flags = int(b["meta_host"][K + 1])
            if use_binned and (flags & _lib.WCN_FLAG_TABLE_FULL) and state["max_blocks"] < N:
                # these scenes are sparser than the bound assumed (>= 16, then >= 4 voxels per occupied 8^3 block): start
                # with the next larger table from now on
                hints.div = 4 if hints.div > 4 else 1
                state["max_blocks"] = max(1024, N // 4) if hints.div == 4 and state["max_blocks"] < max(1024, N // 4) else N
            elif use_binned and (flags & _lib.WCN_FLAG_ROW_OVERFLOW) and state["compact"]:
                # a row with more than 15 neighbours (dense volumetric data): dense table rows for this kernel volume from now on
                hints.observe_row_overflow(K)
                state["compact"] = False
            elif use_binned and (flags & _lib.WCN_FLAG_NEED_STRICT) and not state["strict"]:
                state["strict"] = 1
            else:
                return b, flags, rebuilt
            hints.stats["rebuilds"] += 1
            b, rebuilt = launch(), True

    def attach_tables(result, b):
        result._nbr, result._mask, result._perm = None, b["mask"], b["perm"]
        if b["compact"]:
            result._nbrc = b["nbr"]  # (the dense table is expanded from it on first use: IntSearchResult._nbr)
        else:
            result._nbr = b["nbr"]
        result._offsets_dev = b["meta"][: K + 1]
        result._hashtable = b["table"]
        result._keepalive = b  # (workspaces the queued kernels still read)

    def finalize(result, b, flags):
        """Status word in hand: raise what the reference raises at build time, then offsets, identities, pair lists.
        (The forward / dgrad kernels read the neighbour table only, so nothing queued on the tables has to be repeated
        when the speculative pair lists turn out too short.)"""
        PackedHashTable.raise_for_flags(flags, N, table_capacity)
        # (numpy copy, not Tensor.clone(): copying OUT of pinned memory goes through a stream-synchronising memcpy in torch -
        # measured 380 us per call with work queued, i.e. the host waited for the forward kernel it had just launched)
        offsets_host = torch.from_numpy(b["meta_host"].numpy()[: K + 1].copy())
        pair_capacity = int(offsets_host[-1])
        has_duplicates = bool(flags & _lib.WCN_FLAG_DUPLICATE_COORD)
        identity = K // 2 if (odd and unit_stride and N == M) else None
        if has_duplicates and same_tensor:
            identity = None  # "output row i == input row i at the centre offset" fails for the rows that lost their coordinate
        hints.observe_pairs(M, pair_capacity, K)
        if use_binned and not (flags & (_lib.WCN_FLAG_TABLE_FULL | _lib.WCN_FLAG_NEED_STRICT | _lib.WCN_FLAG_ROW_OVERFLOW)):
            # the cell table of this coordinate set is complete and keeps the smallest row of every coordinate: strided
            # layers on the same tensor reuse it (down-sampling and their kernel maps, coords/ops/stride.py)
            attach_cells(in_coords, b["keep"][0], N, b["max_blocks"])
            if batch_indexed_in_coords is not in_coords and batch_indexed_in_coords.data_ptr() == in_coords.data_ptr():
                attach_cells(batch_indexed_in_coords, b["keep"][0], N, b["max_blocks"])
        attach_tables(result, b)
        result._offsets = offsets_host
        result.identity_map_index = identity
        result._device = torch.device(dev)
        spec = b.get("spec_pairs")
        if spec is not None and spec[2] < pair_capacity:
            spec = None  # the optimistic capacity (pairs per row of earlier maps) was short: exact lists below
            hints.stats["pair_rewrites"] += 1
        if spec is not None:
            # written speculatively right behind the mask sort: the first L entries are the lists.  A guess far above the
            # need (first build of a process) is copied to exact-size buffers instead of pinning the memory for the lifetime
            # of the cached map.
            in_full, out_full, cap = spec
            if cap > 2 * pair_capacity + 65536:
                in_full, out_full = in_full[:pair_capacity].clone(), out_full[:pair_capacity].clone()
            result._in_maps, result._out_maps = in_full[:pair_capacity], out_full[:pair_capacity]
            result._lazy_pairs = None
            b["spec_pairs"] = None
        elif need_pairs:
            # training: the weight gradient needs the pair lists, and written NOW - while the neighbour table is still in the
            # Infinity Cache - they cost 35-45 us less than between dgrad and wgrad
            result._in_maps, result._out_maps = scatter(b, pair_capacity)[:2]
            result._lazy_pairs = None
        else:
            # the forward and dgrad kernels read the neighbour table; only wgrad and the container API need the lists (CSR by
            # offset), so a caller that will not run a weight gradient (the convolution under no_grad) skips the 49 us scatter
            result._in_maps = result._out_maps = None
            result._lazy_pairs = lambda: scatter(b, pair_capacity)[:2]
        # Duplicate input rows break the k-flip identity the dgrad shortcut relies on (rev[n][k] == nbr[n][K-1-k] holds only
        # when every coordinate is one row: a non-winner duplicate has neighbours but is nobody's neighbour): such maps take
        # the explicit reverse table instead.
        result._symmetric = bool(same_tensor and odd and unit_stride and not has_duplicates)
        result._self_exact = result._symmetric
        # duplicate OUTPUT rows (a submanifold map over repeated coordinates) share their input rows per offset: the
        # [N_in, K] reverse table has one slot per (input row, offset), so dgrad then goes through the pair lists
        result._has_duplicates = bool(same_tensor and has_duplicates)
        result._dup_symmetric = bool(result._has_duplicates and odd and unit_stride)  # dgrad: `hip_gemm._dgrad_duplicates`

    result = IntSearchResult._blank(K, dev)
    result._num_in, result._num_out = N, M
    result._kernel_size = ksize
    result._stride_window = prebuilt is not None  # (tables written by the down-sampling pass: every input row in one window)
    # optimistic builds that need the pair lists (the weight gradient's input) write them without knowing the pair count:
    # capacity from the pairs per row of earlier maps (`hints`; the kernel writes nothing past it, validate() rewrites a short
    # guess at the exact length), inside the launches of the mask sort (a stride-window map pairs every input row with at most
    # one cell: N bounds its lists exactly)
    spec = 0
    if optimistic and need_pairs and M > 0:
        spec = min(N, K * M) if prebuilt is not None else hints.pair_capacity(M, K)
    hints.stats["builds"] += 1
    first = launch(spec)
    if optimistic:
        # The caller launches its forward kernel on these tables BEFORE the status word is read (`IntSearchResult.validate`
        # afterwards): the host never stands between the scan kernel and the forward, so the GPU runs the mask sort and the
        # forward back to back instead of idling for the host's round trip.  Every scene a voxeliser produces passes; a
        # build the device rejects (block table too small, duplicate coordinates that need the strict insert) is redone
        # inside validate(), which then reports that the tables changed.
        if need_pairs and first.get("spec_pairs") is None:  # (M == 0)
            first["spec_pairs"] = scatter(first, 0)
        attach_tables(result, first)

        def validate_fn(res, b=first):
            b2, flags, rebuilt = settle(b)
            finalize(res, b2, flags)
            return rebuilt

        result._validate_fn = validate_fn
        return result
    b, flags, _ = settle(first)
    finalize(result, b, flags)
    return result
