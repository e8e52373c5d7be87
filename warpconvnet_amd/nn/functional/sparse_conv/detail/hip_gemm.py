"""HIP backends of the three sparse-conv GEMMs (through the C-ABI in ``include/wcn.h``).

* ``hip_mfma`` - fused gather -> MFMA -> store kernels (`csrc/conv_mfma.hip`, `csrc/wgrad_mfma.hip`)
* ``hip_ref``  - simple kernels for any channel count / fp32 (`csrc/conv_ref.hip`)
* ``auto``     - ``hip_mfma`` when the shape is covered, else ``hip_ref``; a pure function of
  (C_in, C_out, K, dtype), so every rank of a data-parallel job takes the same path.

Role of the reference's `_mask_gemm_forward_logic` / `_mask_gemm_backward_logic`
(`warpconvnet/nn/functional/sparse_conv/detail/mask_gemm.py:661-745, 818-963`).
"""
from typing import Optional, Tuple

import torch
from torch import Tensor

from warpconvnet_amd import _lib
from warpconvnet_amd.geometry.coords.search.search_results import IntSearchResult
from warpconvnet_amd.geometry.coords.search.torch_discrete import attach_tables_from_csr, reverse_tables


def _prep(t: Tensor, name: str) -> Tensor:
    t = t.contiguous()
    _lib.require_gpu_tensor(t, name)
    return t


def resolve_gather_algo(algo: str, cin: int, cout: int, K: int, dtype: torch.dtype) -> int:
    L = _lib.lib()
    ok = bool(L.wcn_mfma_gather_supported(cin, cout, K, _lib.dtype_code(dtype)))
    if algo == "hip_ref":
        return _lib.WCN_ALGO_REF
    if algo == "hip_mfma":
        if not ok:
            raise RuntimeError(f"hip_mfma error: {_lib.status_string(-4)} (cin={cin}, cout={cout}, K={K}, {dtype})")
        return _lib.WCN_ALGO_MFMA
    return _lib.WCN_ALGO_MFMA if ok else _lib.WCN_ALGO_REF


def resolve_wgrad_algo(algo: str, cin: int, cout: int, dtype: torch.dtype) -> int:
    L = _lib.lib()
    ok = bool(L.wcn_mfma_wgrad_supported(cin, cout, _lib.dtype_code(dtype)))
    if algo == "hip_ref":
        return _lib.WCN_ALGO_REF
    if algo == "hip_mfma":
        if not ok:
            raise RuntimeError(f"hip_mfma wgrad error: {_lib.status_string(-4)} (cin={cin}, cout={cout}, {dtype})")
        return _lib.WCN_ALGO_MFMA
    return _lib.WCN_ALGO_MFMA if ok else _lib.WCN_ALGO_REF


def pack_weight(weight: Tensor, transpose: bool, flip: bool) -> Tensor:
    """Fragment-ordered weight image for the MFMA gather-GEMM.  ``weight`` is the forward [K, Cin, Cout]."""
    K, c_in, c_out = weight.shape
    kin, kout = (c_out, c_in) if transpose else (c_in, c_out)  # kernel-side channel roles
    packed = torch.empty(weight.numel(), dtype=weight.dtype, device=weight.device)
    _lib.check(
        _lib.lib().wcn_pack_weight(_lib.ptr(weight), K, kin, kout, _lib.dtype_code(weight.dtype), int(transpose),
                                   int(flip), _lib.ptr(packed), _lib.stream_handle(weight.device)),
        "wcn_pack_weight",
    )
    return packed


def _gather_gemm(inp: Tensor, weight: Tensor, nbr: Tensor, mask: Tensor, perm: Optional[Tensor], n_out: int,
                 cin: int, cout: int, K: int, algo_code: int, transposed: bool, flip: bool,
                 bias: Optional[Tensor] = None) -> Tensor:
    out = torch.empty((n_out, cout), dtype=inp.dtype, device=inp.device)
    if n_out == 0:
        return out
    if algo_code == _lib.WCN_ALGO_MFMA:
        w_arg = pack_weight(weight, transposed, flip)
    else:
        w_arg = weight
    _lib.check(
        _lib.lib().wcn_conv_gather_gemm(
            _lib.ptr(inp), _lib.ptr(w_arg), _lib.ptr(out), _lib.ptr(nbr), _lib.ptr(mask), _lib.ptr(perm), _lib.ptr(bias),
            inp.shape[0], n_out, cin, cout, K, _lib.dtype_code(inp.dtype), algo_code, int(transposed), int(flip),
            _lib.stream_handle(inp.device)),
        "wcn_conv_gather_gemm",
    )
    return out


def hip_forward(in_features: Tensor, weight: Tensor, kernel_map: IntSearchResult, num_out_coords: int,
                algo: str = "auto", bias: Optional[Tensor] = None) -> Tensor:
    """y[m] = sum_k x[nbr[m][k]] @ w[k] (+ bias, fused into the epilogue in fp32)."""
    if bias is not None:
        bias = bias.detach()
        if bias.dtype != torch.float32:
            bias = bias.float()
        bias = _prep(bias, "bias")
    x, w = _prep(in_features, "in_features"), _prep(weight, "weight")
    if x.dtype != w.dtype:
        raise RuntimeError(f"hip forward error: {_lib.status_string(-6)} ({x.dtype} vs {w.dtype})")
    K, cin, cout = w.shape
    assert K == len(kernel_map) and cin == x.shape[1]
    kernel_map.poll()
    attach_tables_from_csr(kernel_map, x.shape[0], num_out_coords)
    code = resolve_gather_algo(algo, cin, cout, K, x.dtype)
    return _gather_gemm(x, w, kernel_map._nbr, kernel_map._mask, kernel_map._perm, num_out_coords, cin, cout, K, code,
                        transposed=False, flip=False, bias=bias)


def hip_colsum(t: Tensor) -> Tensor:
    """fp32 column sums of a [N, C] tensor (bias gradient), deterministic."""
    t = _prep(t, "grad_output")
    n, c = t.shape
    out = torch.empty(c, dtype=torch.float32, device=t.device)
    L = _lib.lib()
    ws_bytes = L.wcn_colsum_workspace(c)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=t.device)
    _lib.check(
        L.wcn_colsum(_lib.ptr(t), n, c, _lib.dtype_code(t.dtype), _lib.ptr(out), _lib.ptr(ws), ws_bytes,
                     _lib.stream_handle(t.device)),
        "wcn_colsum",
    )
    return out


def hip_dgrad(grad_output: Tensor, weight: Tensor, kernel_map: IntSearchResult, num_in_coords: int,
              algo: str = "auto") -> Tensor:
    """dx[n] = sum_k dy[rev[n][k]] @ w[k]^T; a submanifold map reuses the forward table with k reversed."""
    dy, w = _prep(grad_output, "grad_output"), _prep(weight, "weight")
    if dy.dtype != w.dtype:
        raise RuntimeError(f"hip dgrad error: {_lib.status_string(-6)} ({dy.dtype} vs {w.dtype})")
    K, cin, cout = w.shape
    kernel_map.poll()
    attach_tables_from_csr(kernel_map, num_in_coords, dy.shape[0])
    if kernel_map._symmetric:
        tbl, mask, perm, flip = kernel_map._nbr, kernel_map._mask, kernel_map._perm, True
    else:
        tbl, mask, perm = reverse_tables(kernel_map, num_in_coords)
        flip = False
    code = resolve_gather_algo(algo, cout, cin, K, dy.dtype)
    return _gather_gemm(dy, w, tbl, mask, perm, num_in_coords, cout, cin, K, code, transposed=True, flip=flip)


def hip_wgrad(in_features: Tensor, grad_output: Tensor, kernel_map: IntSearchResult, weight_shape, algo: str = "auto",
              want_bias_grad: bool = False):
    """dw[k] = x[in_k]^T @ dy[out_k], fp32 [K, Cin, Cout].

    ``want_bias_grad``: returns ``(dw, bias_grad_or_None)``; the fp32 column sums of ``grad_output`` come out of the same
    kernel (ones-row MFMA on the centre bucket) when the map is a submanifold map over distinct coordinates and the
    MFMA tile layout supports it, else ``None`` (the caller then uses :func:`hip_colsum`).
    """
    x, dy = _prep(in_features, "in_features"), _prep(grad_output, "grad_output")
    if x.dtype != dy.dtype:
        raise RuntimeError(f"hip wgrad error: {_lib.status_string(-6)} ({x.dtype} vs {dy.dtype})")
    K, cin, cout = weight_shape
    dev = x.device
    kernel_map.poll()
    dw = torch.empty((K, cin, cout), dtype=torch.float32, device=dev)
    if kernel_map._offsets_dev is None:
        kernel_map._offsets_dev = kernel_map.offsets.to(device=dev, dtype=torch.int32)
    L = _lib.lib()
    code = resolve_wgrad_algo(algo, cin, cout, x.dtype)
    ws_bytes = L.wcn_conv_wgrad_workspace(K, cin, cout, code)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    fuse = (want_bias_grad and code == _lib.WCN_ALGO_MFMA and getattr(kernel_map, "_self_exact", False)
            and x.shape[0] == dy.shape[0] and dy.shape[0] > 0
            and bool(L.wcn_mfma_wgrad_bias_supported(cin, cout, _lib.dtype_code(x.dtype))))
    if fuse:
        db = torch.empty(cout, dtype=torch.float32, device=dev)
        _lib.check(
            L.wcn_conv_wgrad_bias(_lib.ptr(x), _lib.ptr(dy), _lib.ptr(dw), _lib.ptr(kernel_map.in_maps_device),
                                  _lib.ptr(kernel_map.out_maps_device), _lib.ptr(kernel_map._offsets_dev), x.shape[0],
                                  dy.shape[0], cin, cout, K, _lib.dtype_code(x.dtype), K // 2, _lib.ptr(db), _lib.ptr(ws),
                                  ws_bytes, _lib.stream_handle(dev)),
            "wcn_conv_wgrad_bias",
        )
        return dw, db
    _lib.check(
        L.wcn_conv_wgrad(_lib.ptr(x), _lib.ptr(dy), _lib.ptr(dw), _lib.ptr(kernel_map.in_maps_device),
                         _lib.ptr(kernel_map.out_maps_device), _lib.ptr(kernel_map._offsets_dev), x.shape[0], dy.shape[0], cin,
                         cout, K, _lib.dtype_code(x.dtype), code, _lib.ptr(ws), ws_bytes, _lib.stream_handle(dev)),
        "wcn_conv_wgrad",
    )
    return (dw, None) if want_bias_grad else dw
