"""Generate tests/golden/mink_unet_state.json by IMPORTING THE REFERENCE (authoring container only).

    python tests/golden/make_mink_unet_state.py          # needs /root/reference

For each network the ordered list of (state_dict key, shape) of the reference's own module tree
(`warpconvnet/models/mink_unet.py`): what a checkpoint of the reference contains.  `warpconvnet_amd.models.mink_unet` must
produce exactly these lists (tests/test_models.py).  Only data is written - no reference source.
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import import_reference  # noqa: E402


def main():
    import_reference()
    from warpconvnet.models import mink_unet as ref

    nets = {
        "MinkUNet14": lambda: ref.MinkUNetBase(3, 20, planes=(32, 64, 128, 256, 128, 128, 96, 96), layers=(1,) * 8),
        "MinkUNet18": lambda: ref.MinkUNet18(3, 20),
        "MinkUNet34": lambda: ref.MinkUNet34(4, 13),
        "MinkUNet50": lambda: ref.MinkUNet50(3, 20),
    }
    out = {}
    for name, make in nets.items():
        sd = make().state_dict()
        out[name] = [[k, list(v.shape), str(v.dtype).replace("torch.", "")] for k, v in sd.items()]
        print(name, len(out[name]), "entries")
    with open(os.path.join(HERE, "mink_unet_state.json"), "w") as f:
        json.dump(out, f, indent=0, separators=(",", ":"))


if __name__ == "__main__":
    main()
