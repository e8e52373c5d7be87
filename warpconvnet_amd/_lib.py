"""ctypes binding of the C-ABI library ``libwcn_hip.so`` (declared in ``include/wcn.h``).

The library is built in-tree by ``make -C warpconvnet_amd/csrc`` (``__graft_entry__.build()``); this
module never falls back to another implementation: if the shared object is missing or a symbol is
absent, ``lib()`` raises.  Tensors cross the boundary as raw device pointers + sizes, the stream as
``torch.cuda.current_stream().cuda_stream`` (a ``hipStream_t``).
"""
import ctypes
import os
import subprocess
import threading
from ctypes import c_char_p, c_int, c_int32, c_int64, c_size_t, c_void_p
from typing import Optional

import torch

_CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
# WARPCONVNET_AMD_LIB: alternative build of the same library (e.g. the phase-stamp build `make prof`)
LIB_PATH = os.environ.get("WARPCONVNET_AMD_LIB") or os.path.join(_CSRC, "libwcn_hip.so")

WCN_F32, WCN_F16, WCN_BF16 = 0, 1, 2
WCN_ALGO_AUTO, WCN_ALGO_REF, WCN_ALGO_MFMA = 0, 1, 2
WCN_FLAG_TABLE_FULL, WCN_FLAG_COORD_RANGE, WCN_FLAG_PAIR_OVERFLOW, WCN_FLAG_DUPLICATE_COORD = 1, 2, 4, 8
WCN_FLAG_NEED_STRICT = 16
WCN_FLAG_ROW_OVERFLOW = 32

_I32P = c_void_p  # all pointers travel as void*
_3I = c_int32 * 3

# name -> (restype, argtypes); one line per declaration in include/wcn.h
SIGNATURES = {
    "wcn_abi_version": (c_int, []),
    "wcn_status_string": (c_char_p, [c_int]),
    "wcn_hash_prepare": (c_int, [c_void_p, c_int64, c_void_p]),
    "wcn_hash_insert": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p]),
    "wcn_hash_search": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p]),
    "wcn_kmap_row_pitch": (c_int32, [c_int32]),
    "wcn_kmap_mask_words": (c_int32, [c_int32]),
    "wcn_kmap_num_blocks": (c_int64, [c_int64]),
    "wcn_kmap_probe": (
        c_int,
        [c_void_p, c_int64, c_void_p, c_int64, _3I, _3I, _3I, c_void_p, c_void_p, c_void_p],
    ),
    "wcn_batch_indexed_coords": (c_int, [c_void_p, c_int64, c_int32, c_void_p, c_int32, c_void_p, c_void_p]),
    "wcn_bn_workspace": (c_size_t, [c_int32]),
    "wcn_bn_stats": (c_int, [c_void_p, c_int64, c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "wcn_bn_stats_fold": (c_int, [c_void_p, c_int64, c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_void_p, ctypes.c_float,
                                  ctypes.c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t,
                                  c_void_p]),
    "wcn_bn_fold": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, ctypes.c_float, c_int32, c_void_p, c_void_p, c_void_p,
                            c_void_p, c_void_p]),
    "wcn_bn_apply": (c_int, [c_void_p, c_int64, c_int32, c_int32, c_void_p, c_void_p, c_int32, c_void_p, c_void_p]),
    "wcn_bn_backward_reduce": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_int32, c_void_p, c_void_p,
                                       c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "wcn_bn_backward_apply": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_int32, c_void_p, c_void_p,
                                      c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "wcn_bn_apply_residual": (c_int, [c_void_p, c_void_p, c_int64, c_int32, c_int32, c_void_p, c_void_p, c_int32, c_void_p,
                                      c_void_p]),
    "wcn_bn_backward_reduce_masked": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_int32, c_void_p, c_void_p,
                                              c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "wcn_bn_backward_apply_masked": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_int32, c_void_p, c_void_p,
                                             c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "wcn_bn_train_forward": (c_int, [c_void_p, c_void_p, c_int64, c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_void_p,
                                     ctypes.c_float, ctypes.c_float, c_void_p, c_int32, c_void_p, c_void_p, c_void_p, c_size_t,
                                     c_void_p]),
    "wcn_bn_train_backward": (c_int, [c_void_p, c_void_p, c_void_p, c_int32, c_int64, c_int32, c_int32, c_void_p, c_void_p,
                                      c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "wcn_bn_train_backward_ld": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_int32, c_int64, c_int32, c_int32, c_void_p, c_void_p,
                                         c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "wcn_conv_bn_backward_ld": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_int32, c_void_p, c_void_p, c_int32, c_void_p,
                                        c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_void_p, c_void_p, c_void_p,
                                        c_void_p, c_void_p, c_void_p, c_size_t, c_int64, c_int64, c_int32, c_int32, c_int32, c_int32,
                                        c_void_p, c_size_t, c_void_p]),
    "wcn_conv_bn_backward": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_void_p, c_void_p, c_int32, c_void_p, c_void_p,
                                     c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_void_p, c_void_p, c_void_p, c_void_p,
                                     c_void_p, c_void_p, c_size_t, c_int64, c_int64, c_int32, c_int32, c_int32, c_int32, c_void_p,
                                     c_size_t, c_void_p]),
    "wcn_pool_gather": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int32, c_int32, c_int32, c_int32, c_void_p, c_void_p,
                                c_void_p, c_void_p]),
    "wcn_pool_select": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int32, c_int32, c_int32, c_void_p, c_void_p]),
    "wcn_morton_code": (c_int, [c_void_p, c_int64, c_int32, c_void_p, c_void_p, c_void_p, c_void_p]),
    "wcn_kmap_counts_bytes": (c_size_t, [c_int64, c_int32]),
    "wcn_kmap_binned_workspace": (c_size_t, [c_int64, c_int64]),
    "wcn_kmap_binned_supported": (c_int, [_3I, _3I]),
    "wcn_kmap_build_binned": (
        c_int,
        [c_void_p, c_int64, _3I, _3I, c_int64, c_int32, c_int32, c_void_p, c_size_t, c_void_p, c_void_p, c_void_p, c_void_p],
    ),
    "wcn_kmap_compact_supported": (c_int, [c_int32]),
    "wcn_kmap_densify": (c_int, [c_void_p, c_int64, c_int32, c_void_p, c_void_p]),
    "wcn_kmap_tally_sort_workspace": (c_size_t, [c_int64]),
    "wcn_kmap_tally_sort": (
        c_int,
        [c_void_p, c_void_p, c_int64, c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t,
         c_void_p, c_void_p, c_int64, c_int64, c_int32, c_void_p, c_void_p, c_int64, c_void_p],
    ),
    "wcn_kmap_count": (c_int, [c_void_p, c_int64, c_int32, c_void_p, c_void_p]),
    "wcn_kmap_scan": (c_int, [c_void_p, c_int64, c_int32, c_void_p, c_void_p]),
    "wcn_kmap_scan_to_host": (c_int, [c_void_p, c_int64, c_int32, c_void_p, c_void_p, c_void_p, c_void_p]),
    "wcn_kmap_scatter": (
        c_int,
        [c_void_p, c_void_p, c_int64, c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_int32, c_void_p],
    ),
    "wcn_kmap_cells_build": (c_int, [c_void_p, c_int64, c_int64, c_void_p, c_size_t, c_void_p, c_void_p, c_void_p]),
    "wcn_cells_stride_supported": (c_int, [_3I]),
    "wcn_cells_stride_tiles": (c_int64, [c_int64]),
    "wcn_cells_stride_count": (c_int, [c_void_p, c_int64, c_int64, c_void_p, _3I, c_void_p, c_void_p, c_int32, c_void_p,
                                       c_void_p]),
    "wcn_cells_stride_emit": (c_int, [c_void_p, c_int64, c_int64, c_void_p, _3I, c_void_p, c_void_p, c_void_p, c_void_p,
                                      c_void_p, c_void_p, c_void_p]),
    "wcn_kmap_probe_cells": (c_int, [c_void_p, c_int64, c_int64, c_void_p, c_int64, _3I, _3I, _3I, c_void_p, c_void_p,
                                     c_void_p]),
    "wcn_kmap_transpose": (c_int, [c_void_p, c_int64, c_int32, c_void_p, c_void_p]),
    "wcn_kmap_reverse": (
        c_int,
        [c_void_p, c_void_p, c_void_p, c_int32, c_int64, c_int64, c_void_p, c_void_p, c_void_p],
    ),
    "wcn_kmap_from_csr": (
        c_int,
        [c_void_p, c_void_p, c_void_p, c_int32, c_int64, c_int64, c_void_p, c_void_p, c_void_p],
    ),
    "wcn_mask_argsort_workspace": (c_size_t, [c_int64]),
    "wcn_mask_argsort": (c_int, [c_void_p, c_int32, c_int32, c_int64, c_void_p, c_void_p, c_size_t, c_void_p]),
    "wcn_mask_tile_order": (c_int, [c_void_p, c_int32, c_int32, c_int64, c_void_p, c_void_p, c_size_t, c_void_p]),
    "wcn_mfma_gather_supported": (c_int, [c_int32, c_int32, c_int32, c_int32]),
    "wcn_mfma_wgrad_supported": (c_int, [c_int32, c_int32, c_int32]),
    "wcn_conv_identity_supported": (c_int, [c_int32, c_int32, c_int32]),
    "wcn_dense_rows_supported": (c_int, [c_int32, c_int32, c_int32]),
    "wcn_dense_rows": (
        c_int,
        [c_void_p, c_int32, c_void_p, c_int32, c_int32, c_void_p, c_void_p, c_int64, c_int32, c_int32, c_int32, c_void_p],
    ),
    "wcn_packed_weight_bytes": (c_size_t, [c_int32, c_int32, c_int32, c_int32, c_int32]),
    "wcn_pack_weight": (
        c_int,
        [c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, c_void_p, c_size_t, c_void_p],
    ),
    "wcn_pack_weight_f32": (
        c_int,
        [c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, c_void_p, c_size_t, c_void_p],
    ),
    "wcn_pack_weight_pair_supported": (c_int, [c_int32, c_int32, c_int32, c_int32]),
    "wcn_conv_compact_table_supported": (c_int, [c_int32, c_int32, c_int32, c_int32]),
    "wcn_pack_weight_f32_pair": (c_int, [c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32, c_void_p, c_size_t, c_void_p, c_size_t,
                                         c_void_p]),
    "wcn_conv_gather_gemm": (
        c_int,
        [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int32, c_int32,
         c_int32, c_int32, c_int32, c_int32, c_int32, c_void_p],
    ),
    "wcn_mfma_grouped_supported": (c_int, [c_int32, c_int32, c_int32, c_int32]),
    "wcn_pack_weight_grouped": (
        c_int,
        [c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, c_void_p, c_void_p],
    ),
    "wcn_conv_gather_gemm_grouped": (
        c_int,
        [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int32, c_int32,
         c_int32, c_int32, c_int32, c_void_p],
    ),
    "wcn_conv_gather_gemm_f32out": (
        c_int,
        [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int32, c_int32,
         c_int32, c_int32, c_void_p],
    ),
    "wcn_conv_gather_gemm_fused": (
        c_int,
        [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32,
         c_int64, c_int64, c_int32, c_int32, c_int32, c_int32, c_void_p],
    ),
    "wcn_colsum_workspace": (c_size_t, [c_int32]),
    "wcn_colsum": (c_int, [c_void_p, c_int64, c_int32, c_int32, c_void_p, c_void_p, c_size_t, c_void_p]),
    "wcn_conv_wgrad_workspace": (c_size_t, [c_int32, c_int32, c_int32, c_int32]),
    "wcn_conv_wgrad": (
        c_int,
        [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int32, c_int32, c_int32,
         c_int32, c_int32, c_void_p, c_size_t, c_void_p],
    ),
    "wcn_dwconv_gather": (
        c_int,
        [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int32, c_int32, c_int32, c_int32, c_void_p],
    ),
    "wcn_dwconv_wgrad_workspace": (c_size_t, [c_int32, c_int32]),
    "wcn_dwconv_wgrad": (
        c_int,
        [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int32, c_int32, c_int32,
         c_void_p, c_size_t, c_void_p],
    ),
    "wcn_radius_grid_count": (
        c_int,
        [c_void_p, c_void_p, c_void_p, ctypes.c_float * 3, ctypes.c_float, c_int32 * 3, c_void_p, c_int64, ctypes.c_float,
         c_void_p, c_void_p],
    ),
    "wcn_radius_grid_write": (
        c_int,
        [c_void_p, c_void_p, c_void_p, ctypes.c_float * 3, ctypes.c_float, c_int32 * 3, c_void_p, c_int64, ctypes.c_float,
         c_void_p, c_void_p, c_void_p, c_void_p],
    ),
    "wcn_knn_grid": (
        c_int,
        [c_void_p, c_void_p, c_void_p, ctypes.c_float * 3, ctypes.c_float, c_int32 * 3, c_void_p, c_int64, c_int32, c_void_p,
         c_void_p, c_void_p],
    ),
    "wcn_segment_reduce": (c_int, [c_void_p, c_void_p, c_int64, c_int32, c_int32, c_int32, c_void_p, c_void_p, c_void_p]),
    "wcn_pointconv_supported": (c_int, [c_int32] * 7),
    "wcn_pointconv_packed_floats": (c_int64, [c_int32] * 3),
    "wcn_pointconv_grad_floats": (c_int64, [c_int32] * 4),
    "wcn_pointconv_backward_workspace": (c_size_t, [c_int64, c_int32, c_int32, c_int32, c_int32, c_int32]),
    "wcn_pointconv_pack": (c_int, [c_void_p] * 10 + [c_int32] * 3 + [c_void_p, c_void_p]),
    "wcn_pointconv_edge_forward": (
        c_int,
        [c_void_p] * 5 + [c_int64] + [c_int32] * 4 + [c_void_p, c_int32, c_int32, ctypes.c_float, ctypes.c_float, c_int32,
                                                      c_int32, c_void_p, c_void_p],
    ),
    "wcn_pointconv_edge_backward": (
        c_int,
        [c_void_p] * 5 + [c_int64] + [c_int32] * 4 + [c_void_p, c_int32, c_int32, ctypes.c_float, ctypes.c_float, c_int32,
                                                      c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t,
                                                      c_void_p],
    ),
    "wcn_pointconv_edge_backward_peredge": (
        c_int,
        [c_void_p] * 5 + [c_int64] + [c_int32] * 4 + [c_void_p, c_int32, c_int32, ctypes.c_float, ctypes.c_float, c_int32,
                                                      c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t,
                                                      c_void_p],
    ),
    "wcn_pointconv_edge_forward_ragged": (
        c_int,
        [c_void_p] * 7 + [c_int64, c_int64] + [c_int32] * 3 + [c_void_p, c_int32, c_int32, ctypes.c_float, ctypes.c_float,
                                                               c_int32, c_void_p, c_void_p],
    ),
    "wcn_pointconv_edge_backward_ragged": (
        c_int,
        [c_void_p] * 7 + [c_int64, c_int64] + [c_int32] * 3 + [c_void_p, c_int32, c_int32, ctypes.c_float, ctypes.c_float,
                                                               c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                                               c_size_t, c_void_p],
    ),
    "wcn_mfma_wgrad_bias_supported": (c_int, [c_int32, c_int32, c_int32]),
    "wcn_conv_wgrad_bias": (
        c_int,
        [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int32, c_int32, c_int32,
         c_int32, c_int32, c_void_p, c_void_p, c_size_t, c_void_p],
    ),
}

_LIB = None  # _GuardedLib


class _GuardedLib:
    """The loaded library with the reference bindings' device guard (`c10::cuda::CUDAGuard` in every pybind entry point):
    the C-ABI takes raw pointers and a stream, launches go to the CURRENT device, so a call whose tensors live on another
    GPU (model on cuda:1 while cuda:0 is current, several GPUs driven from one process) switches the device for the call.
    The device of a call is the one its stream handle was asked for (`stream_handle(device)`, evaluated as the last
    argument of every launching call); pure host entry points never pass through the switch."""

    def __init__(self, handle: ctypes.CDLL):
        self._h = handle

    def __getattr__(self, name):
        fn = getattr(self._h, name)
        argtypes = fn.argtypes or []
        if not argtypes or argtypes[-1] is not c_void_p or name in _HOST_ONLY or _single_gpu():
            setattr(self, name, fn)  # (one visible GPU: there is no other device to be current)
            return fn

        def guarded(*args, _fn=fn):
            dev = getattr(_CALL_DEVICE, "dev", None)
            if dev is None or dev == torch.cuda.current_device():
                return _fn(*args)
            with torch.cuda.device(dev):
                return _fn(*args)

        guarded.__name__ = name
        setattr(self, name, guarded)
        return guarded


def _single_gpu() -> bool:
    try:
        return torch.cuda.is_available() and torch.cuda.device_count() == 1
    except Exception:
        return False


_HOST_ONLY = {"wcn_status_string"}
_CALL_DEVICE = threading.local()  # .dev = device index of the stream this THREAD asked for last (evaluated per call)


def build(verbose: bool = False) -> str:
    """Compile the HIP sources for gfx950 (cross-compiles without a GPU). Returns the .so path."""
    cmd = ["make", "-C", _CSRC, "-j", str(min(8, os.cpu_count() or 1))]
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError(f"building libwcn_hip.so failed:\n{res.stdout[-4000:]}\n{res.stderr[-4000:]}")
    if verbose:
        print(res.stdout[-2000:])
    return LIB_PATH


def lib() -> ctypes.CDLL:
    """Load (once) and return the C-ABI library; raise loudly if it is not there."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} not found: the HIP extension is not built. Run "
                "`python -c 'import __graft_entry__ as g; g.build()'` (or `make -C warpconvnet_amd/csrc`). "
                "There is no CPU fallback for the HIP paths."
            )
        handle = ctypes.CDLL(LIB_PATH)
        for name, (restype, argtypes) in SIGNATURES.items():
            fn = getattr(handle, name)  # AttributeError if the symbol is missing
            fn.restype = restype
            fn.argtypes = argtypes
        _LIB = _GuardedLib(handle)
    return _LIB


def status_string(code: int) -> str:
    return lib().wcn_status_string(int(code)).decode()


def check(code: int, what: str) -> None:
    """Non-zero C status -> RuntimeError (reference convention: `backends.py:489-510`)."""
    if code != 0:
        raise RuntimeError(f"{what} error: {status_string(code)} ({code})")


def ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    if t is None:
        return None
    return t.data_ptr()


_RAW_STREAM = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def stream_handle(device: torch.device) -> int:
    """hipStream_t of the current PyTorch stream on ``device``.  Every C-ABI call needs it (~900 per MinkUNet iteration):
    the raw accessor skips building a ``torch.cuda.Stream`` object (≈ 6 us -> 0.3 us per call)."""
    idx = device.index
    if idx is None:
        idx = torch.cuda.current_device()
    _CALL_DEVICE.dev = idx
    if _RAW_STREAM is not None:
        return _RAW_STREAM(idx)
    return torch.cuda.current_stream(device).cuda_stream


def dtype_code(dtype: torch.dtype) -> int:
    if dtype == torch.float32:
        return WCN_F32
    if dtype == torch.float16:
        return WCN_F16
    if dtype == torch.bfloat16:
        return WCN_BF16
    raise TypeError(f"unsupported feature dtype {dtype} (supported: float32, float16, bfloat16)")


def torch_dtype(code: int) -> torch.dtype:
    return {WCN_F32: torch.float32, WCN_F16: torch.float16, WCN_BF16: torch.bfloat16}[code]


def i3(v) -> "ctypes.Array":
    return _3I(int(v[0]), int(v[1]), int(v[2]))


def require_gpu_tensor(t: torch.Tensor, name: str) -> None:
    if not t.is_cuda:
        raise RuntimeError(f"{name} must live on a GPU for the HIP path (got {t.device}); there is no CPU fallback")
    if not t.is_contiguous():
        raise RuntimeError(f"{name} must be contiguous")
