"""Inference-time sparse convolution with the ConvBlock tail folded into the GEMM store.

The reference's blocks run ``SparseConv3d -> BatchNorm1d -> ReLU`` (and the residual add of a BasicBlock) as separate
elementwise kernels over the ``[N, C]`` feature tensor (`models/mink_unet.py:31-53, 161-175`): after a bandwidth-bound
GEMM that is three more read+write passes.  `wcn_conv_gather_gemm_fused` applies

    y = act((conv + bias) * scale + shift + residual)

in fp32 inside the epilogue of the MFMA gather GEMM, so the features are written once.  Forward only (no autograd):
training-mode BatchNorm needs batch statistics of the convolution output, which is a different fusion.
"""
from typing import Optional, Tuple

import torch
from torch import Tensor

from warpconvnet_amd import _lib
from warpconvnet_amd.geometry.coords.integer import IntCoords
from warpconvnet_amd.geometry.coords.search.torch_discrete import attach_tables_from_csr
from warpconvnet_amd.geometry.types.voxels import Voxels
from warpconvnet_amd.nn.functional.sparse_conv.detail import hip_gemm
from warpconvnet_amd.utils.ntuple import ntuple


def fold_batchnorm(norm: torch.nn.modules.batchnorm._BatchNorm) -> Tuple[Tensor, Tensor]:
    """(scale, shift) fp32 of a BatchNorm in inference mode: y = x * scale + shift."""
    if norm.running_mean is None or norm.running_var is None:
        raise ValueError("fold_batchnorm needs running statistics (track_running_stats=True)")
    var, mean = norm.running_var.float(), norm.running_mean.float()
    gamma = norm.weight.float() if norm.weight is not None else torch.ones_like(var)
    beta = norm.bias.float() if norm.bias is not None else torch.zeros_like(var)
    scale = gamma * torch.rsqrt(var + norm.eps)
    return scale.contiguous(), (beta - mean * scale).contiguous()


def _f32(t: Optional[Tensor], name: str, cout: int, dev) -> Optional[Tensor]:
    if t is None:
        return None
    t = t.detach().to(device=dev, dtype=torch.float32).contiguous()
    if t.shape != (cout,):
        raise ValueError(f"{name} must have shape ({cout},), got {tuple(t.shape)}")
    return t


@torch.no_grad()
def hip_forward_fused(in_features: Tensor, weight: Tensor, kernel_map, num_out_coords: int, bias: Optional[Tensor] = None,
                      scale: Optional[Tensor] = None, shift: Optional[Tensor] = None, residual: Optional[Tensor] = None,
                      relu: bool = False) -> Tensor:
    """One fused launch when the MFMA kernel covers the shape; otherwise the HIP forward followed by the same chain as
    torch elementwise ops (identical math, more passes)."""
    x, w = in_features.contiguous(), weight.to(in_features.dtype).contiguous()
    _lib.require_gpu_tensor(x, "in_features")
    K, cin, cout = w.shape
    dev = x.device
    bias, scale, shift = _f32(bias, "bias", cout, dev), _f32(scale, "scale", cout, dev), _f32(shift, "shift", cout, dev)
    if (scale is None) != (shift is None):
        raise ValueError("scale and shift go together")
    if residual is not None:
        if residual.shape != (num_out_coords, cout):
            raise ValueError(f"residual must be [{num_out_coords}, {cout}], got {tuple(residual.shape)}")
        residual = residual.detach().to(x.dtype).contiguous()
    L = _lib.lib()
    fused_ok = x.dtype in (torch.float16, torch.bfloat16) and bool(
        L.wcn_mfma_gather_supported(cin, cout, K, _lib.dtype_code(x.dtype)))
    if not fused_ok:
        y = hip_gemm.hip_forward(x, w, kernel_map, num_out_coords, "auto", bias=bias).float()
        if scale is not None:
            y = y * scale + shift
        if residual is not None:
            y = y + residual.float()
        return (torch.relu(y) if relu else y).to(x.dtype)
    kernel_map.validate()
    attach_tables_from_csr(kernel_map, x.shape[0], num_out_coords)
    out = torch.empty((num_out_coords, cout), dtype=x.dtype, device=dev)
    if num_out_coords == 0:
        return out
    wp = hip_gemm.pack_weight(w, False, False)
    tb, mk = hip_gemm.own_tables(kernel_map, cin, cout, K, x.dtype)
    _lib.check(
        L.wcn_conv_gather_gemm_fused(_lib.ptr(x), _lib.ptr(wp), _lib.ptr(out), _lib.ptr(tb),
                                     _lib.ptr(mk), _lib.ptr(kernel_map._perm), _lib.ptr(bias), _lib.ptr(scale),
                                     _lib.ptr(shift), _lib.ptr(residual), int(relu), x.shape[0], num_out_coords, cin, cout, K,
                                     _lib.dtype_code(x.dtype), _lib.stream_handle(dev)),
        "wcn_conv_gather_gemm_fused",
    )
    return out


@torch.no_grad()
def fused_sparse_conv_inference(input_sparse_tensor: Voxels, weight: Tensor, kernel_size, stride=1, kernel_dilation=1,
                                bias: Optional[Tensor] = None, scale: Optional[Tensor] = None,
                                shift: Optional[Tensor] = None, residual: Optional[Voxels] = None, relu: bool = False,
                                compute_dtype: Optional[torch.dtype] = None, order=None) -> Voxels:
    """``spatially_sparse_conv`` (non-transposed, groups = 1) + per-channel affine + residual add + ReLU in one kernel.
    Output coordinates, tensor stride and kernel-map caching follow `spatially_sparse_conv`."""
    from warpconvnet_amd.nn.functional.sparse_conv.helper import generate_output_coords_and_kernel_map

    if weight.ndim != 3:
        raise ValueError("fused_sparse_conv_inference takes an ungrouped weight [K, Cin, Cout]")
    nd = input_sparse_tensor.num_spatial_dims
    ks, st, dl = ntuple(kernel_size, nd), ntuple(stride, nd), ntuple(kernel_dilation, nd)
    if compute_dtype is None:
        compute_dtype = torch.get_autocast_dtype("cuda") if torch.is_autocast_enabled() else input_sparse_tensor.feature_tensor.dtype
    feats = input_sparse_tensor.feature_tensor.detach().to(compute_dtype)
    bcoords_out, out_offsets, kernel_map = generate_output_coords_and_kernel_map(input_sparse_tensor, ks, dl, st, order=order)
    res = None
    if residual is not None:
        res = residual.feature_tensor
        assert res.shape[0] == bcoords_out.shape[0], "residual must live on the output coordinates"
    out = hip_forward_fused(feats, weight.detach(), kernel_map, bcoords_out.shape[0], bias, scale, shift, res, relu)
    in_ts = input_sparse_tensor.tensor_stride or ntuple(1, nd)
    return input_sparse_tensor.replace(
        batched_coordinates=IntCoords(bcoords_out[:, 1:], offsets=out_offsets.cpu().int()),
        batched_features=out,
        tensor_stride=tuple(o * s for o, s in zip(st, in_ts)),
    )
