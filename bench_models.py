"""Benchmark / parity-test models (beside bench.py; not product code).

MinkUNet-14 (BASELINE config 3) assembled from the build's SparseConv3d.

Structure of the reference's `MinkUNetBase(planes=(32,64,128,256,128,128,96,96), layers=(1,)*8)`
(`warpconvnet/models/mink_unet.py:31-405`; the reference has no MinkUNet14 class, this is the MinkowskiEngine
convention): 1x1 stem, four k=2/s=2 down-convolutions each followed by a residual block, four transposed k=2/s=2
up-convolutions onto the encoder tensors with channel concatenation, 1x1 head.
"""
import torch
import torch.nn as nn

from warpconvnet_amd.geometry.types.voxels import Voxels
from warpconvnet_amd.nn.modules.sequential import Sequential
from warpconvnet_amd.nn.modules.sparse_conv import SparseConv3d

PLANES = (32, 64, 128, 256, 128, 128, 96, 96)


def cat(a: Voxels, b: Voxels) -> Voxels:
    return a.replace(batched_features=torch.cat([a.feature_tensor, b.feature_tensor], dim=1))


class ConvBlock(Sequential):
    def __init__(self, cin, cout, kernel_size=3, stride=1, act=True):
        super().__init__(SparseConv3d(cin, cout, kernel_size, stride, bias=False), nn.BatchNorm1d(cout),
                         nn.ReLU(inplace=True) if act else nn.Identity())


class ConvTrBlock(nn.Module):
    def __init__(self, cin, cout, kernel_size=2, stride=2):
        super().__init__()
        self.conv_tr = SparseConv3d(cin, cout, kernel_size, stride, transposed=True, bias=False)
        self.norm_act = Sequential(nn.BatchNorm1d(cout), nn.ReLU(inplace=True))

    def forward(self, x, skip):
        return self.norm_act(self.conv_tr(x, skip))


class BasicBlock(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.conv1 = ConvBlock(cin, cout, 3)
        self.conv2 = ConvBlock(cout, cout, 3, act=False)
        self.down = None if cin == cout else Sequential(SparseConv3d(cin, cout, 1, bias=False), nn.BatchNorm1d(cout))
        self.relu = nn.ReLU(inplace=True)

    def forward(self, x):
        out = self.conv2(self.conv1(x))
        res = x if self.down is None else self.down(x)
        return out.replace(batched_features=self.relu(out.feature_tensor + res.feature_tensor))


class MinkUNet14(nn.Module):
    def __init__(self, in_channels=3, out_channels=20, planes=PLANES):
        super().__init__()
        p = planes
        self.conv0 = ConvBlock(in_channels, 32, kernel_size=1)
        self.conv1, self.block1 = ConvBlock(32, 32, 2, 2), BasicBlock(32, p[0])
        self.conv2, self.block2 = ConvBlock(p[0], p[0], 2, 2), BasicBlock(p[0], p[1])
        self.conv3, self.block3 = ConvBlock(p[1], p[1], 2, 2), BasicBlock(p[1], p[2])
        self.conv4, self.block4 = ConvBlock(p[2], p[2], 2, 2), BasicBlock(p[2], p[3])
        self.convtr4, self.block5 = ConvTrBlock(p[3], p[4]), BasicBlock(p[4] + p[2], p[4])
        self.convtr5, self.block6 = ConvTrBlock(p[4], p[5]), BasicBlock(p[5] + p[1], p[5])
        self.convtr6, self.block7 = ConvTrBlock(p[5], p[6]), BasicBlock(p[6] + p[0], p[6])
        self.convtr7, self.block8 = ConvTrBlock(p[6], p[7]), BasicBlock(p[7] + 32, p[7])
        self.final = SparseConv3d(p[7], out_channels, kernel_size=1, bias=True)

    def forward(self, x: Voxels) -> Voxels:
        p1 = self.conv0(x)
        b1 = self.block1(self.conv1(p1))
        b2 = self.block2(self.conv2(b1))
        b3 = self.block3(self.conv3(b2))
        out = self.block4(self.conv4(b3))
        out = self.block5(cat(self.convtr4(out, b3), b3))
        out = self.block6(cat(self.convtr5(out, b2), b2))
        out = self.block7(cat(self.convtr6(out, b1), b1))
        out = self.block8(cat(self.convtr7(out, p1), p1))
        return self.final(out)

    def set_algo(self, algo: str):
        for m in self.modules():
            if isinstance(m, SparseConv3d):
                m.fwd_algo = m.dgrad_algo = type(m.fwd_algo)(algo)
                m.wgrad_algo = type(m.wgrad_algo)(algo)


class ConvLayerRecorder:
    """Forward hooks on every SparseConv3d of a model: (cin, cout, K, N_in, N_out, pairs) per call, for byte models of a
    whole network (bench.py secondary roofline).  `pairs` comes from the kernel map the layer left in the input's cache."""

    def __init__(self, model):
        self.records = []
        self._handles = [m.register_forward_hook(self._hook) for m in model.modules() if isinstance(m, SparseConv3d)]

    def _hook(self, mod, inputs, output):
        x = inputs[0]
        n_in, n_out = int(x.coordinate_tensor.shape[0]), int(output.coordinate_tensor.shape[0])
        K = 1
        for k in mod.kernel_size:
            K *= int(k)
        pairs = n_out if K == 1 else None
        for src in (x, output) + tuple(inputs[1:2]):
            cache = getattr(src, "cache", None)
            if pairs is not None or cache is None:
                continue
            for key, km in cache.items():
                if tuple(key.kernel_size) == tuple(mod.kernel_size) and {int(key.in_offsets[-1]), int(key.out_offsets[-1])} == {n_in, n_out}:
                    pairs = int(km.offsets[-1])
                    break
        self.records.append(dict(cin=mod.in_channels, cout=mod.out_channels, K=K, n_in=n_in, n_out=n_out,
                                 pairs=pairs if pairs is not None else 0))

    def close(self):
        for h in self._handles:
            h.remove()
