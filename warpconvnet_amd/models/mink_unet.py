"""MinkUNet family on the build's sparse convolutions (BASELINE config 3).

Mirrors the part of the reference's model zoo that the hot path's configurations exercise (`warpconvnet/models/mink_unet.py`:
``ConvBlock`` `:31-53`, ``ConvTrBlock`` `:55-91`, ``BasicBlock`` `:115-175`, ``BottleneckBlock`` `:178-250`, ``MinkUNetBase``
`:253-404`, ``MinkUNet18/34/50/101`` `:407-454`): same constructor arguments, same module tree - so the ``state_dict`` keys and
shapes are the reference's and a checkpoint of one loads into the other (pinned by `tests/golden/mink_unet_state.json`).
``MinkUNet14`` is the MinkowskiEngine convention ``layers=(1,) * 8, planes=(32, 64, 128, 256, 128, 128, 96, 96)``: the reference
has no class of that name, BASELINE config 3 names the network.

What is different is how a block runs, not what it computes:

* ``ConvBlock`` is the package's ``Sequential``: conv -> BatchNorm (-> ReLU) is ONE autograd node with direct launches
  (`nn/functional/sparse_conv/block.py`).
* the tail of a residual block - ``out += identity; out = relu(out)`` behind conv2's BatchNorm (`mink_unet.py:160-172`) - rides
  on that node's BatchNorm passes (``conv_bn_act(..., residual=identity)``): the forward adds the identity while it applies
  scale / shift, the backward masks with the stored output and hands the masked gradient to the identity branch.  Three
  streaming passes over the feature tensor and two autograd nodes fewer per block, values bit-identical.
* ``use_checkpoint``: activation checkpointing per block through ``torch.utils.checkpoint`` (non-reentrant) with the BatchNorm
  buffers of the block restored after the recomputation, as the reference's mixin does (`mink_unet.py:92-112`).
"""
from contextlib import contextmanager, nullcontext
from typing import Optional, Union

import torch
import torch.nn as nn

from warpconvnet_amd.geometry.types.voxels import Voxels
from warpconvnet_amd.nn.functional.sparse_conv.block import conv_bn_act
from warpconvnet_amd.nn.modules.sequential import Sequential, _has_hooks
from warpconvnet_amd.nn.modules.sparse_conv import SparseConv3d

__all__ = ["ConvBlock", "ConvTrBlock", "BasicBlock", "BottleneckBlock", "MinkUNetBase", "MinkUNet14", "MinkUNet18", "MinkUNet34",
           "MinkUNet50", "MinkUNet101", "cat"]

_RELU = object()  # default activation marker (a fresh ReLU per block instead of one shared default instance)


def cat(a: Voxels, b: Voxels) -> Voxels:
    """Channel concatenation of two tensors on the same coordinates (the skip connections of the decoder)."""
    return a.replace(batched_features=torch.cat([a.feature_tensor, b.feature_tensor], dim=1))


def _activation(activation):
    if activation is _RELU:
        return nn.ReLU(inplace=True)
    return nn.Identity() if activation is None else activation


class ConvBlock(Sequential):
    """SparseConv3d -> BatchNorm1d -> activation (reference `mink_unet.py:31-53`)."""

    def __init__(self, in_channels: int, out_channels: int, kernel_size: int = 3, stride: int = 1, activation=_RELU,
                 bias: bool = False, compute_dtype: Optional[torch.dtype] = None):
        super().__init__(SparseConv3d(in_channels, out_channels, kernel_size, stride, bias=bias, compute_dtype=compute_dtype),
                         nn.BatchNorm1d(out_channels), _activation(activation))


class ConvTrBlock(nn.Module):
    """Transposed SparseConv3d onto the coordinates of an encoder tensor -> BatchNorm1d -> activation (`mink_unet.py:55-91`)."""

    def __init__(self, in_channels: int, out_channels: int, kernel_size: int = 3, stride: int = 1, activation=_RELU,
                 bias: bool = False, compute_dtype: Optional[torch.dtype] = None):
        super().__init__()
        self.conv_tr = SparseConv3d(in_channels, out_channels, kernel_size, stride, transposed=True, bias=bias,
                                    compute_dtype=compute_dtype)
        self.norm_act = Sequential(nn.BatchNorm1d(out_channels), _activation(activation))

    def forward(self, x: Voxels, out_spatial_sparsity: Voxels) -> Voxels:
        norm, act = self.norm_act[0], self.norm_act[1]
        if type(act) in (nn.ReLU, nn.Identity) and not (_has_hooks(self.conv_tr) or _has_hooks(norm) or _has_hooks(act) or
                                                       _has_hooks(self.norm_act)):
            y = conv_bn_act(x, self.conv_tr, norm, type(act) is nn.ReLU, out_spatial=out_spatial_sparsity)
            if y is not None:  # one autograd node, tables shared with the strided layer whose map this one exchanges
                return y
        return self.norm_act(self.conv_tr(x, out_spatial_sparsity))


@contextmanager
def _restore_batchnorm_buffers(module: nn.Module):
    """Recomputation must not advance the running statistics a second time (reference `mink_unet.py:92-112`)."""
    saved = [(b, b.detach().clone()) for m in module.modules() if isinstance(m, nn.modules.batchnorm._BatchNorm)
             for b in (m.running_mean, m.running_var, m.num_batches_tracked) if b is not None]
    try:
        yield
    finally:
        with torch.no_grad():
            for b, v in saved:
                b.copy_(v)


class _Checkpointed(nn.Module):
    use_checkpoint = False

    def _run(self, fn, x):
        if self.use_checkpoint and self.training and torch.is_grad_enabled():
            from torch.utils.checkpoint import checkpoint

            return checkpoint(fn, x, use_reentrant=False,
                              context_fn=lambda: (nullcontext(), _restore_batchnorm_buffers(self)))
        return fn(x)

    def _residual_tail(self, out: Voxels, last: ConvBlock, identity: Voxels) -> Voxels:
        """``relu(last(out) + identity)``: through the fused node when ``last`` is conv -> BatchNorm without activation."""
        conv, norm, act = last[0], last[1], last[2]
        if type(act) is nn.Identity and type(self.relu) is nn.ReLU and not (_has_hooks(conv) or _has_hooks(norm) or
                                                                          _has_hooks(self.relu) or _has_hooks(last)):
            y = conv_bn_act(out, conv, norm, True, residual=identity)
            if y is not None:
                return y
        out = last(out)
        return out.replace(batched_features=self.relu(out.feature_tensor + identity.feature_tensor))


class BasicBlock(_Checkpointed):
    """Two 3x3x3 ConvBlocks with an identity (or 1x1x1-projected) shortcut (reference `mink_unet.py:115-175`)."""

    expansion = 1

    def __init__(self, in_channels: int, out_channels: int, stride: int = 1, bias: bool = False,
                 compute_dtype: Optional[torch.dtype] = None, use_checkpoint: bool = False):
        super().__init__()
        self.use_checkpoint = use_checkpoint
        self.conv1 = ConvBlock(in_channels, out_channels, kernel_size=3, stride=stride, bias=bias, compute_dtype=compute_dtype)
        self.conv2 = ConvBlock(out_channels, out_channels, kernel_size=3, activation=None, bias=bias, compute_dtype=compute_dtype)
        self.downsample = None
        if stride != 1 or in_channels != out_channels:
            self.downsample = ConvBlock(in_channels, out_channels, kernel_size=1, stride=stride, activation=None, bias=bias,
                                        compute_dtype=compute_dtype)
        self.relu = nn.ReLU(inplace=True)

    def _forward(self, x: Voxels) -> Voxels:
        out = self.conv1(x)
        identity = x if self.downsample is None else self.downsample(x)
        return self._residual_tail(out, self.conv2, identity)

    def forward(self, x: Voxels) -> Voxels:
        return self._run(self._forward, x)


class BottleneckBlock(_Checkpointed):
    """1x1x1 -> 3x3x3 -> 1x1x1 with a four-fold channel reduction in the middle (reference `mink_unet.py:178-250`)."""

    expansion = 4

    def __init__(self, in_channels: int, out_channels: int, stride: int = 1, bias: bool = False,
                 compute_dtype: Optional[torch.dtype] = None, use_checkpoint: bool = False):
        super().__init__()
        self.use_checkpoint = use_checkpoint
        mid = out_channels // self.expansion
        self.conv1 = ConvBlock(in_channels, mid, kernel_size=1, bias=bias, compute_dtype=compute_dtype)
        self.conv2 = ConvBlock(mid, mid, kernel_size=3, stride=stride, bias=bias, compute_dtype=compute_dtype)
        self.conv3 = ConvBlock(mid, out_channels, kernel_size=1, activation=None, bias=bias, compute_dtype=compute_dtype)
        self.downsample = None
        if stride != 1 or in_channels != out_channels:
            self.downsample = ConvBlock(in_channels, out_channels, kernel_size=1, stride=stride, activation=None, bias=bias,
                                        compute_dtype=compute_dtype)
        self.relu = nn.ReLU(inplace=True)

    def _forward(self, x: Voxels) -> Voxels:
        out = self.conv2(self.conv1(x))
        identity = x if self.downsample is None else self.downsample(x)
        return self._residual_tail(out, self.conv3, identity)

    def forward(self, x: Voxels) -> Voxels:
        return self._run(self._forward, x)


_BLOCKS = {"BasicBlock": BasicBlock, "BottleneckBlock": BottleneckBlock}


class MinkUNetBase(nn.Module):
    """U-Net of four stride-2 encoder stages and four transposed decoder stages with channel concatenation onto the encoder
    tensors (reference `mink_unet.py:253-404`, after MinkowskiEngine's examples/minkunet.py)."""

    def __init__(self, in_channels: int, out_channels: int, planes: tuple, layers: tuple, init_dim: int = 32,
                 BLOCK: Union[str, type] = BasicBlock, init_kernel_size: int = 1, use_checkpoint: bool = False, **kwargs):
        super().__init__()
        assert len(planes) == len(layers) == 8, "eight stages: four down, four up"
        if isinstance(BLOCK, str):
            BLOCK = _BLOCKS[BLOCK]
        self.PLANES, self.LAYERS, self.INIT_DIM = planes, layers, init_dim
        p = planes
        self.conv0 = ConvBlock(in_channels, init_dim, kernel_size=init_kernel_size, bias=False)
        widths_in = [init_dim, p[0], p[1], p[2]]
        for i in range(4):  # encoder: conv{i+1} halves the resolution, block{i+1} works at it
            setattr(self, f"conv{i + 1}", ConvBlock(widths_in[i], widths_in[i], kernel_size=2, stride=2))
            setattr(self, f"block{i + 1}", self._make_layer(BLOCK, widths_in[i], p[i], layers[i]))
        skips = [p[2], p[1], p[0], p[0]]  # (the last skip is conv0's output: init_dim channels - the reference sizes it p[0])
        for i in range(4):  # decoder: convtr{4+i} doubles the resolution onto a skip tensor, block{5+i} follows the concatenation
            setattr(self, f"convtr{4 + i}", ConvTrBlock(p[3 + i], p[4 + i], kernel_size=2, stride=2))
            setattr(self, f"block{5 + i}", self._make_layer(BLOCK, p[4 + i] + skips[i], p[4 + i], layers[4 + i]))
        self.final = SparseConv3d(p[7], out_channels, kernel_size=1, bias=True)
        if use_checkpoint:
            self.gradient_checkpointing_enable()

    def _make_layer(self, BLOCK, in_channels: int, out_channels: int, blocks: int, block_kwargs: Optional[dict] = None,
                    compute_dtype: Optional[torch.dtype] = None) -> nn.Sequential:
        kw = dict(block_kwargs or {})
        mods = [BLOCK(in_channels, out_channels, compute_dtype=compute_dtype, **kw)]
        mods += [BLOCK(out_channels, out_channels, compute_dtype=compute_dtype, **kw) for _ in range(1, blocks)]
        return nn.Sequential(*mods)

    def gradient_checkpointing_enable(self, enabled: bool = True):
        for m in self.modules():
            if isinstance(m, _Checkpointed):
                m.use_checkpoint = enabled

    def gradient_checkpointing_disable(self):
        self.gradient_checkpointing_enable(False)

    def forward(self, x: Voxels) -> Voxels:
        p1 = self.conv0(x)
        b1 = self.block1(self.conv1(p1))
        b2 = self.block2(self.conv2(b1))
        b3 = self.block3(self.conv3(b2))
        out = self.block4(self.conv4(b3))
        for i, skip in enumerate((b3, b2, b1, p1)):
            out = getattr(self, f"convtr{4 + i}")(out, skip)
            out = getattr(self, f"block{5 + i}")(cat(out, skip))
        return self.final(out)


_PLANES_18 = (32, 64, 128, 256, 256, 128, 96, 96)


class MinkUNet14(MinkUNetBase):
    def __init__(self, in_channels: int, out_channels: int, **kwargs):
        super().__init__(in_channels, out_channels, planes=(32, 64, 128, 256, 128, 128, 96, 96), layers=(1,) * 8, init_dim=32,
                         BLOCK=BasicBlock, **kwargs)


class MinkUNet18(MinkUNetBase):
    def __init__(self, in_channels: int, out_channels: int, **kwargs):
        super().__init__(in_channels, out_channels, planes=_PLANES_18, layers=(2,) * 8, init_dim=32, BLOCK=BasicBlock, **kwargs)


class MinkUNet34(MinkUNetBase):
    def __init__(self, in_channels: int, out_channels: int, **kwargs):
        super().__init__(in_channels, out_channels, planes=_PLANES_18, layers=(2, 3, 4, 6, 2, 2, 2, 2), init_dim=32,
                         BLOCK=BasicBlock, **kwargs)


class MinkUNet50(MinkUNetBase):
    def __init__(self, in_channels: int, out_channels: int, **kwargs):
        super().__init__(in_channels, out_channels, planes=_PLANES_18, layers=(2, 3, 4, 6, 2, 2, 2, 2), init_dim=32,
                         BLOCK=BottleneckBlock, **kwargs)


class MinkUNet101(MinkUNetBase):
    def __init__(self, in_channels: int, out_channels: int, **kwargs):
        super().__init__(in_channels, out_channels, planes=_PLANES_18, layers=(2, 3, 4, 23, 2, 2, 2, 2), init_dim=32,
                         BLOCK=BottleneckBlock, **kwargs)
