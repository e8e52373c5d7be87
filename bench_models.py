"""Benchmark / parity-test helpers around the product model (beside bench.py; not product code).

MinkUNet-14 (BASELINE config 3) is `warpconvnet_amd.models.mink_unet.MinkUNet14`: the reference's
`MinkUNetBase(planes=(32,64,128,256,128,128,96,96), layers=(1,)*8)` (`warpconvnet/models/mink_unet.py:31-405`; the reference
has no MinkUNet14 class, this is the MinkowskiEngine convention): 1x1 stem, four k=2/s=2 down-convolutions each followed by a
residual block, four transposed k=2/s=2 up-convolutions onto the encoder tensors with channel concatenation, 1x1 head.
"""
from warpconvnet_amd.models.mink_unet import MinkUNet14 as _MinkUNet14
from warpconvnet_amd.nn.modules.sparse_conv import SparseConv3d

PLANES = (32, 64, 128, 256, 128, 128, 96, 96)


class MinkUNet14(_MinkUNet14):
    """The product model (`warpconvnet_amd/models/mink_unet.py`) plus a test knob: one algorithm for every convolution."""

    def set_algo(self, algo: str):
        for m in self.modules():
            if isinstance(m, SparseConv3d):
                m.fwd_algo = m.dgrad_algo = type(m.fwd_algo)(algo)
                m.wgrad_algo = type(m.wgrad_algo)(algo)
                m.__dict__.pop("_wcn_block_ok", None)  # (memo of the fused-block eligibility: depends on the algorithms)


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
