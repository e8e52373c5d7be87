# dev soak: MinkUNet-14 with the fused nodes vs WARPCONVNET_AMD_FUSED_BLOCK=0 (module by module) on random batched scenes -
# logits, every gradient and every BatchNorm buffer bit for bit (GPU box): python tools/soak_models.py
import copy, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tests.util import scene_u
from warpconvnet_amd.geometry.types.voxels import Voxels
from warpconvnet_amd.models import mink_unet
dev = torch.device("cuda:0")
bad = 0
for seed in range(int(sys.argv[1]) if len(sys.argv) > 1 else 10):
    rng = np.random.default_rng(seed)
    parts = [scene_u(int(rng.integers(3000, 40000)), 100 + 7 * seed + b)[:, 1:] for b in range(int(rng.integers(1, 4)))]
    torch.manual_seed(seed)
    name = ("MinkUNet14", "MinkUNet14", "MinkUNet18", "MinkUNet50")[(seed // 2) % 4]
    net = getattr(mink_unet, name)(3, 11).to(dev)
    feats = [torch.randn(len(p), 3) for p in parts]
    def run(model):
        model.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.bfloat16 if seed % 2 == 0 else torch.float16):
            y = model(Voxels([torch.from_numpy(p) for p in parts], feats, device=dev))
        y.feature_tensor.float().square().mean().backward()
        return [y.feature_tensor.detach().clone()] + [p.grad.clone() for p in model.parameters()] + [b.clone() for b in model.buffers()]
    a = run(copy.deepcopy(net))
    os.environ["WARPCONVNET_AMD_FUSED_BLOCK"] = "0"
    try: b = run(copy.deepcopy(net))
    finally: del os.environ["WARPCONVNET_AMD_FUSED_BLOCK"]
    eq = [torch.equal(u, v) or bool(((u == v) | (u.isnan() & v.isnan())).all()) for u, v in zip(a, b)]
    fin = all(torch.isfinite(u.float()).all() for u in a)
    ok = all(eq)
    names = ["logits"] + [n for n, _ in net.named_parameters()] + [n for n, _ in net.named_buffers()]
    print(f"seed {seed} {name}: scenes {[len(p) for p in parts]} -> {'ok' if ok else 'MISMATCH'} finite={bool(fin)}",
          [names[i] for i, e in enumerate(eq) if not e][:6])
    bad += not ok
print("soak done, failures:", bad)
