"""GPU: torch.compile over models that contain sparse convolutions (reference tests/nn/test_torch_compile.py:93-235):
tracing must not crash, results and gradients must equal eager, strided layers and changing inputs must work."""
import pytest
import torch
import torch.nn as nn

from tests.util import rel_max_err, scene_u

pytestmark = pytest.mark.gpu


def _vox(n=3000, seed=0, C=16):
    from warpconvnet_amd.geometry.types.voxels import Voxels

    p = [torch.from_numpy(scene_u(n, seed + b)[:, 1:]) for b in range(2)]
    g = torch.Generator().manual_seed(seed)
    return Voxels(p, [torch.randn(len(c), C, generator=g) for c in p], device=torch.device("cuda:0"))


class _Net(nn.Module):
    def __init__(self, stride=1):
        super().__init__()
        from warpconvnet_amd.nn.modules.sparse_conv import SparseConv3d

        self.c1 = SparseConv3d(16, 32, 3)
        self.c2 = SparseConv3d(32, 24, 2 if stride == 2 else 3, stride=stride)

    def forward(self, x):
        h = self.c1(x)
        h = h.replace(batched_features=torch.relu(h.feature_tensor) * 1.5)  # dense work between the sparse layers
        return self.c2(h)


@pytest.mark.parametrize("backend", ["eager", "aot_eager"])
@pytest.mark.parametrize("stride", [1, 2])
def test_compile_matches_eager_forward_backward(stride, backend):
    torch.manual_seed(0)
    net = _Net(stride).cuda()
    compiled = torch.compile(net, fullgraph=False, backend=backend)
    outs = []
    for fn in (net, compiled):
        net.zero_grad()
        vox = _vox()
        x = vox.replace(batched_features=vox.feature_tensor.clone().requires_grad_(True))
        y = fn(x)
        y.feature_tensor.square().sum().backward()
        outs.append((y.feature_tensor.detach(), x.feature_tensor.grad.clone(), net.c1.weight.grad.clone(), net.c2.weight.grad.clone()))
    for a, b in zip(*outs):
        assert rel_max_err(b, a) < 1e-5
    # a different scene through the same compiled module
    y2 = compiled(_vox(2000, seed=7))
    assert y2.feature_tensor.shape[1] == 24 and torch.isfinite(y2.feature_tensor).all()


def test_compile_default_backend_no_crash():
    """The default (inductor) backend compiles the dense pieces; the sparse layers stay opaque graph breaks."""
    torch.manual_seed(0)
    net = _Net().cuda()
    vox = _vox()
    want = net(vox).feature_tensor
    try:
        got = torch.compile(net, fullgraph=False)(vox).feature_tensor
    except Exception as e:  # inductor needs a working triton on the box; its absence is not this package's failure
        if "triton" in str(e).lower() or "inductor" in str(e).lower():
            pytest.skip(f"inductor unavailable: {type(e).__name__}")
        raise
    assert rel_max_err(got, want) < 1e-4
