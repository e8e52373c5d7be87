"""CPU: `warpconvnet_amd.models.mink_unet` has the reference's module tree - the ordered (key, shape, dtype) list of
``state_dict()`` equals the one captured from the reference's own classes (`tests/golden/mink_unet_state.json`, written by
`tests/golden/make_mink_unet_state.py` through the stubbed import), so checkpoints move between the two."""
import json
import os

import pytest
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "mink_unet_state.json")


def _entries(net):
    return [[k, list(v.shape), str(v.dtype).replace("torch.", "")] for k, v in net.state_dict().items()]


@pytest.mark.parametrize("name,args", [("MinkUNet14", (3, 20)), ("MinkUNet18", (3, 20)), ("MinkUNet34", (4, 13)),
                                       ("MinkUNet50", (3, 20))])
def test_state_dict_layout_is_the_references(name, args):
    from warpconvnet_amd.models import mink_unet

    with open(GOLDEN) as f:
        want = json.load(f)[name]
    got = _entries(getattr(mink_unet, name)(*args))
    assert len(got) == len(want)
    for g, w in zip(got, want):
        assert g == w


def test_minkunet14_is_minkunetbase_with_one_block_per_stage():
    from warpconvnet_amd.models.mink_unet import BasicBlock, MinkUNet14, MinkUNetBase

    a = MinkUNet14(3, 20)
    b = MinkUNetBase(3, 20, planes=(32, 64, 128, 256, 128, 128, 96, 96), layers=(1,) * 8, BLOCK="BasicBlock")
    assert _entries(a) == _entries(b)
    assert all(len(getattr(a, f"block{i}")) == 1 and isinstance(getattr(a, f"block{i}")[0], BasicBlock) for i in range(1, 9))
    # a checkpoint round trip between two instances
    b.load_state_dict(a.state_dict())
    for (k, va), (_, vb) in zip(a.state_dict().items(), b.state_dict().items()):
        assert torch.equal(va, vb), k


def test_gradient_checkpointing_switch_reaches_every_block():
    from warpconvnet_amd.models.mink_unet import MinkUNet18, _Checkpointed

    net = MinkUNet18(3, 5, use_checkpoint=True)
    blocks = [m for m in net.modules() if isinstance(m, _Checkpointed)]
    assert len(blocks) == 16 and all(m.use_checkpoint for m in blocks)
    net.gradient_checkpointing_disable()
    assert not any(m.use_checkpoint for m in blocks)
