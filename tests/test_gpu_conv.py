"""GPU: the three sparse-conv GEMMs (HIP, through the C-ABI) vs the CPU oracle and the golden vectors."""
import glob
import os

import numpy as np
import pytest
import torch

from oracle import conv as oconv
from oracle import kmap as okmap
from tests.util import rel_max_err, scene_surface, scene_u

pytestmark = pytest.mark.gpu

# Tolerances of the reference's own tests (tests/nn/test_kernel_correctness.py:64-65, 139-145):
# max|d| / max|ref| < 1e-3 for fp32, < 2e-2 for fp16 / bf16.
TOL = {torch.float32: 1e-3, torch.float16: 2e-2, torch.bfloat16: 2e-2}


def _dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _kmap(in_np, out_np, ksize, stride=(1, 1, 1), same=False):
    from warpconvnet_amd.geometry.coords.search.torch_discrete import generate_kernel_map

    a = torch.from_numpy(in_np).to(_dev())
    b = a if same else torch.from_numpy(out_np).to(_dev())
    return generate_kernel_map(a, b, stride, ksize)


def _run_all(km, X, W, dY, algo, num_in, num_out):
    from warpconvnet_amd.nn.functional.sparse_conv.detail import hip_gemm

    Y = hip_gemm.hip_forward(X, W, km, num_out, algo)
    dX = hip_gemm.hip_dgrad(dY, W, km, num_in, algo)
    dW = hip_gemm.hip_wgrad(X, dY, km, tuple(W.shape), algo)
    return Y, dX, dW


def _oracle(r, X, W, dY, num_out, iden=None):
    """fp64 oracle on the values the GPU actually saw (inputs already rounded to the storage dtype)."""
    Xd, Wd, dYd = X.double().cpu(), W.double().cpu(), dY.double().cpu()
    Y = oconv.forward(Xd, Wd, r["in_maps"], r["out_maps"], r["offsets"], num_out, iden)
    dX, dW = oconv.backward(dYd, Xd, Wd, r["in_maps"], r["out_maps"], r["offsets"], iden)
    return Y, dX, dW


@pytest.mark.parametrize("name", sorted(os.path.basename(p) for p in glob.glob(os.path.join(os.path.dirname(__file__), "golden", "explicit_*f32*.npz"))))
@pytest.mark.parametrize("algo", ["hip_ref", "auto"])
def test_golden_vectors_fp32(golden_dir, name, algo):
    g = np.load(os.path.join(golden_dir, name))
    same = g["in_coords"].shape == g["out_coords"].shape and (g["in_coords"] == g["out_coords"]).all()
    km = _kmap(g["in_coords"], g["out_coords"], tuple(g["ksize"]), tuple(g["stride"]), same=same)
    np.testing.assert_array_equal(km.in_maps.cpu().numpy(), g["in_maps"])
    np.testing.assert_array_equal(km.out_maps.cpu().numpy(), g["out_maps"])
    dev = _dev()
    X, W, dY = (torch.from_numpy(g[k]).to(dev) for k in ("X", "W", "dY"))
    Y, dX, dW = _run_all(km, X, W, dY, algo, len(g["in_coords"]), len(g["out_coords"]))
    for got, want in ((Y, g["Y"]), (dX, g["dX"]), (dW, g["dW"])):
        assert rel_max_err(got, torch.from_numpy(want)) < TOL[torch.float32]


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("cin,cout", [(64, 128), (128, 64), (32, 32), (16, 32), (96, 96), (64, 256), (256, 64), (128, 128),
                                      (256, 256), (192, 128), (96, 320), (128, 384)])
def test_mfma_vs_oracle_submanifold(dtype, cin, cout):
    """(outputs wider than 128 channels run as 128 / 96 / 64-wide column blocks of the channel-split kernel: 256 = 2 x 128,
    192 = 2 x 96 (the dgrad of 192 -> 128), 320 = 5 x 64, 384 = 3 x 128)"""
    s = np.concatenate([scene_u(3000, 21, 0), scene_u(1500, 22, 1)], 0)
    km = _kmap(s, s, (3, 3, 3), same=True)
    r = okmap.kernel_map(s, s, (3, 3, 3))
    dev = _dev()
    g = torch.Generator(device="cpu").manual_seed(cin * 1000 + cout)
    X = torch.randn(len(s), cin, generator=g).to(dev, dtype)
    W = (torch.randn(27, cin, cout, generator=g) * 0.05).to(dev, dtype)
    dY = torch.randn(len(s), cout, generator=g).to(dev, dtype)
    Y, dX, dW = _run_all(km, X, W, dY, "hip_mfma" if (cin % 64 == 0 and cout % 64 == 0) else "auto", len(s), len(s))
    assert Y.dtype == dtype and dX.dtype == dtype and dW.dtype == torch.float32
    Yr, dXr, dWr = _oracle(r, X, W, dY, len(s))
    assert rel_max_err(Y, Yr) < TOL[dtype]
    assert rel_max_err(dX, dXr) < TOL[dtype]
    assert rel_max_err(dW, dWr) < TOL[dtype]
    # hip_ref on the same inputs agrees too (fp32 accumulate, storage-dtype rounding only at the end)
    Y2, dX2, dW2 = _run_all(km, X, W, dY, "hip_ref", len(s), len(s))
    assert rel_max_err(Y2, Yr) < TOL[dtype] and rel_max_err(dX2, dXr) < TOL[dtype] and rel_max_err(dW2, dWr) < 1e-3

@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("cin,cout", [(64, 128), (128, 64), (32, 32), (64, 64), (96, 96), (256, 64), (128, 256)])
def test_wgrad_fused_bias_grad(dtype, cin, cout):
    """wcn_conv_wgrad_bias: dw bit-identical to wcn_conv_wgrad, bias gradient = column sums of grad_output (fp64 sum of
    the values the GPU saw; fp32 accumulation => 1e-4 relative), deterministic, and NOT offered when coordinates repeat."""
    from warpconvnet_amd import _lib
    from warpconvnet_amd.nn.functional.sparse_conv.detail import hip_gemm

    s = np.concatenate([scene_u(3000, 31, 0), scene_u(1234, 32, 1)], 0)  # 4234 rows: ragged last chunk
    km = _kmap(s, s, (3, 3, 3), same=True)
    assert km._symmetric and km._self_exact
    dev = _dev()
    g = torch.Generator(device="cpu").manual_seed(cin * 7 + cout)
    X = torch.randn(len(s), cin, generator=g).to(dev, dtype)
    dY = torch.randn(len(s), cout, generator=g).to(dev, dtype)
    dW0 = hip_gemm.hip_wgrad(X, dY, km, (27, cin, cout), "auto")
    dW1, db = hip_gemm.hip_wgrad(X, dY, km, (27, cin, cout), "auto", want_bias_grad=True)
    assert torch.equal(dW0, dW1)
    if not _lib.lib().wcn_mfma_wgrad_bias_supported(cin, cout, _lib.dtype_code(dtype)):
        assert db is None  # odd number of 32-channel blocks: the caller falls back to wcn_colsum
        return
    assert db is not None and db.dtype == torch.float32 and db.shape == (cout,)
    want = dY.double().sum(0)
    assert ((db.double() - want).abs().max() / want.abs().max()).item() < 1e-4
    _, db2 = hip_gemm.hip_wgrad(X, dY, km, (27, cin, cout), "auto", want_bias_grad=True)
    assert torch.equal(db, db2)
    assert rel_max_err(db, hip_gemm.hip_colsum(dY)) < 1e-5
    # duplicate coordinates: bucket K//2 is no longer "every row once" -> no fused bias gradient
    sd = np.concatenate([s[:2000], s[:17]], 0)
    kmd = _kmap(sd, sd, (3, 3, 3), same=True)
    assert not kmd._symmetric and not kmd._self_exact  # duplicates: dgrad takes the explicit reverse table
    _, dbd = hip_gemm.hip_wgrad(X[: len(sd)].contiguous(), dY[: len(sd)].contiguous(), kmd, (27, cin, cout), "auto",
                                want_bias_grad=True)
    assert dbd is None

@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("cin,cout", [(64, 128), (128, 64), (96, 96), (64, 256), (96, 320)])
@pytest.mark.parametrize("flip", [False, True])
def test_forward_and_dgrad_weight_images_in_one_launch(dtype, cin, cout, flip):
    """wcn_pack_weight_f32_pair: the two images of a training step equal the ones wcn_pack_weight_f32 writes one by one, and
    `pack_weight(..., dgrad_flip=)` leaves both in the parameter's cache (the backward then packs nothing)."""
    from warpconvnet_amd import _lib
    from warpconvnet_amd.nn.functional.sparse_conv.detail import hip_gemm

    dev = _dev()
    torch.manual_seed(cin + cout)
    w = torch.nn.Parameter(torch.randn(27, cin, cout, device=dev))
    assert _lib.lib().wcn_pack_weight_pair_supported(27, cin, cout, _lib.dtype_code(dtype)) == 1
    fwd = hip_gemm.pack_weight(w, False, False, dtype=dtype, dgrad_flip=flip)
    cache = w._wcn_packed
    assert (dtype, False, False) in cache and (dtype, True, flip) in cache
    bwd = hip_gemm.pack_weight(w, True, flip, dtype=dtype)
    assert bwd is cache[(dtype, True, flip)][1]  # served from the cache
    w2 = torch.nn.Parameter(w.detach().clone())
    assert torch.equal(fwd.view(torch.int16), hip_gemm.pack_weight(w2, False, False, dtype=dtype).view(torch.int16))
    assert torch.equal(bwd.view(torch.int16), hip_gemm.pack_weight(w2, True, flip, dtype=dtype).view(torch.int16))
    # a wrong prediction costs one ordinary pack, never a wrong image
    other = hip_gemm.pack_weight(w, True, not flip, dtype=dtype)
    assert torch.equal(other.view(torch.int16), hip_gemm.pack_weight(w2, True, not flip, dtype=dtype).view(torch.int16))
    with torch.no_grad():
        w.add_(1.0)  # an optimizer step: both cached images are stale
    assert not torch.equal(hip_gemm.pack_weight(w, False, False, dtype=dtype, dgrad_flip=flip).view(torch.int16), fwd.view(torch.int16))

@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("cin,cout", [(64, 128), (128, 64), (96, 96), (64, 256)])
def test_compact_table_rows_equal_the_dense_table(dtype, cin, cout):
    """Round 6: the binned builder writes COMPACT rows (mask + the neighbours of the set offsets, 64 B) and the channel-split
    kernels expand them into their index slab (`mask` = NULL).  Same tiles, same steps: forward and dgrad bit-identical to the
    launches on the dense table + mask array expanded from them; duplicates and out-of-table rows included."""
    from warpconvnet_amd import _lib
    from warpconvnet_amd.nn.functional.sparse_conv.detail import hip_gemm

    s = np.concatenate([scene_u(3000, 51, 0), scene_u(1100, 52, 1)], 0)
    s = np.concatenate([s, s[:40]], 0)  # 40 duplicated coordinates: repaired rows copy their winner's compact row
    km = _kmap(s, s, (3, 3, 3), same=True)
    assert km._nbrc is not None and km._nbr_dense is None and km.has_tables
    tb, mk = hip_gemm.own_tables(km, cin, cout, 27, dtype)
    assert tb is km._nbrc and mk is None
    np.testing.assert_array_equal(km._nbrc[:, 0].cpu().numpy().view(np.uint32), km._mask[:, 0].cpu().numpy().view(np.uint32))
    dense = km._nbr  # expanded on first use
    assert dense.shape == (len(s), 32) and km._nbr_dense is dense
    r = okmap.kernel_map(s, s, (3, 3, 3))
    np.testing.assert_array_equal(dense.cpu().numpy()[:, :27].T, r["found"])
    assert (dense[:, 27:] == -1).all()
    dev = _dev()
    g = torch.Generator().manual_seed(cin + cout)
    X = torch.randn(len(s), cin, generator=g).to(dev, dtype)
    W = (torch.randn(27, cin, cout, generator=g) * 0.05).to(dev, dtype)
    y_tbl = hip_gemm._gather_gemm(X, W, km._nbrc, None, km._perm, len(s), cin, cout, 27, _lib.WCN_ALGO_MFMA, False, False)
    y_arr = hip_gemm._gather_gemm(X, W, dense, km._mask, km._perm, len(s), cin, cout, 27, _lib.WCN_ALGO_MFMA, False, False)
    assert torch.equal(y_tbl, y_arr)
    dY = torch.randn(len(s), cout, generator=g).to(dev, dtype)
    d_tbl = hip_gemm._gather_gemm(dY, W, km._nbrc, None, km._perm, len(s), cout, cin, 27, _lib.WCN_ALGO_MFMA, True, True)
    d_arr = hip_gemm._gather_gemm(dY, W, dense, km._mask, km._perm, len(s), cout, cin, 27, _lib.WCN_ALGO_MFMA, True, True)
    assert torch.equal(d_tbl, d_arr)
    # a dense table (hash builder) keeps its mask argument; shapes outside the channel-split family get the dense table
    import os
    os.environ["WARPCONVNET_AMD_KMAP_METHOD"] = "hash"
    try:
        kh = _kmap(s[:-40], s[:-40], (3, 3, 3), same=True)
    finally:
        del os.environ["WARPCONVNET_AMD_KMAP_METHOD"]
    assert kh._nbrc is None and hip_gemm.own_tables(kh, cin, cout, 27, dtype) == (kh._nbr, kh._mask)
    tb32, mk32 = hip_gemm.own_tables(km, 32, 32, 27, dtype)
    assert tb32 is dense and mk32 is km._mask


def test_a_row_with_more_than_15_neighbours_falls_back_to_dense_rows():
    """A dense volume (every cell of a 6^3 cube occupied: interior rows have 27 neighbours) does not fit compact rows: the device
    raises ROW_OVERFLOW, the build is redone with dense rows and the hints remember it for that kernel volume."""
    from warpconvnet_amd.geometry.coords.search.torch_discrete import BuildHints, generate_kernel_map

    g = np.stack(np.meshgrid(np.arange(6), np.arange(6), np.arange(6), indexing="ij"), -1).reshape(-1, 3)
    rng = np.random.default_rng(5)
    rng.shuffle(g)
    c = np.concatenate([np.zeros((len(g), 1), np.int64), g], 1).astype(np.int32)
    t = torch.from_numpy(c).to(_dev())
    hints = BuildHints()
    assert hints.compact_rows(27)
    km = generate_kernel_map(t, t, (1, 1, 1), (3, 3, 3), hints=hints)
    assert km._nbrc is None and km._nbr_dense is not None and not hints.compact_rows(27) and hints.compact_rows(25)
    r = okmap.kernel_map(c, c, (3, 3, 3))
    np.testing.assert_array_equal(km._nbr.cpu().numpy()[:, :27].T, r["found"])
    np.testing.assert_array_equal(km.in_maps.cpu().numpy(), r["in_maps"])
    np.testing.assert_array_equal(km.out_maps.cpu().numpy(), r["out_maps"])
    # optimistic build of the same scene with fresh hints: the forward queued on the rejected tables is harmless and repeated
    hints2 = BuildHints()
    km2 = generate_kernel_map(t, t, (1, 1, 1), (3, 3, 3), optimistic=True, hints=hints2)
    assert km2._nbrc is not None
    assert km2.validate() is True and km2._nbrc is None
    np.testing.assert_array_equal(km2._nbr.cpu().numpy()[:, :27].T, r["found"])
    # ... and a scene at the limit (exactly 15 neighbours in some rows is fine): a 2-D sheet of 3 layers has rows of 18, a
    # single layer rows of 9
    sheet = np.stack(np.meshgrid(np.arange(12), np.arange(12), np.arange(1), indexing="ij"), -1).reshape(-1, 3)
    cs = np.concatenate([np.zeros((len(sheet), 1), np.int64), sheet], 1).astype(np.int32)
    ts = torch.from_numpy(cs).to(_dev())
    ks = generate_kernel_map(ts, ts, (1, 1, 1), (3, 3, 3), hints=BuildHints())
    assert ks._nbrc is not None
    rs = okmap.kernel_map(cs, cs, (3, 3, 3))
    np.testing.assert_array_equal(ks._nbr.cpu().numpy()[:, :27].T, rs["found"])


def test_rows_with_exactly_15_neighbours_keep_their_last_id():
    """The limit case of compact rows: 15 neighbours, the last of them NOT at the last kernel offset (so absent offsets are
    probed after the row is full - an append that writes before it advances must not land on word 15), over many random
    neighbour sets, several voxels per 8^3 block and blocks of more than 64 voxels (two probe passes)."""
    from warpconvnet_amd.geometry.coords.search.torch_discrete import BuildHints, generate_kernel_map

    rng = np.random.default_rng(15)
    offs = np.stack(np.meshgrid(np.arange(-1, 2), np.arange(-1, 2), np.arange(-1, 2), indexing="ij"), -1).reshape(-1, 3)
    pts = []
    for i in range(200):  # isolated centres 4 apart, each with 14 random neighbours besides itself (14 + 1 = 15)
        centre = np.array([4 * (i % 10), 4 * ((i // 10) % 10), 4 * (i // 100)]) + 2
        others = rng.choice(np.delete(np.arange(27), 13), size=14, replace=False)
        pts.append(centre[None])
        pts.append(centre[None] + offs[others])
    g = np.unique(np.concatenate(pts), axis=0)
    rng.shuffle(g)
    c = np.concatenate([np.zeros((len(g), 1), np.int64), g], 1).astype(np.int32)
    r = okmap.kernel_map(c, c, (3, 3, 3))
    counts = (r["found"] >= 0).sum(0)
    assert counts.max() == 15 and (counts == 15).sum() >= 100
    t = torch.from_numpy(c).to(_dev())
    km = generate_kernel_map(t, t, (1, 1, 1), (3, 3, 3), hints=BuildHints())
    assert km._nbrc is not None
    np.testing.assert_array_equal(km._nbr.cpu().numpy()[:, :27].T, r["found"])
    np.testing.assert_array_equal(km.in_maps.cpu().numpy(), r["in_maps"])
    np.testing.assert_array_equal(km.out_maps.cpu().numpy(), r["out_maps"])


@pytest.mark.parametrize("fused_block", [False, True])
def test_the_training_forward_packs_both_weight_images(fused_block):
    """The module path (and the fused conv -> BatchNorm node) must reach the pair packer: grad mode is OFF inside
    `Function.forward`, so the decision rides on `ctx.needs_input_grad`, not on `torch.is_grad_enabled()` - after the forward
    of a step whose input needs a gradient the parameter's cache already holds the (transposed, k-flipped) dgrad image."""
    from warpconvnet_amd.geometry.types.voxels import Voxels
    from warpconvnet_amd.nn.modules.sparse_conv import SparseConv3d

    dev = _dev()
    c = torch.from_numpy(scene_u(4000, 77)[:, 1:]).to(dev)
    n = c.shape[0]
    feats = torch.randn(n, 64, device=dev).requires_grad_(True)
    torch.manual_seed(0)
    conv = SparseConv3d(64, 128, 3, bias=not fused_block).to(dev)
    x = Voxels(c, feats, offsets=torch.tensor([0, n], dtype=torch.int32))
    with torch.autocast("cuda", dtype=torch.bfloat16):
        if fused_block:
            from warpconvnet_amd.nn.functional.sparse_conv.block import conv_bn_act

            y = conv_bn_act(x, conv, torch.nn.BatchNorm1d(128).to(dev), relu=True)
            assert y is not None
        else:
            y = conv(x)
    cache = conv.weight._wcn_packed
    assert (torch.bfloat16, False, False) in cache and (torch.bfloat16, True, True) in cache, list(cache)
    before = cache[(torch.bfloat16, True, True)][1]
    y.feature_tensor.float().sum().backward()
    assert conv.weight._wcn_packed[(torch.bfloat16, True, True)][1] is before  # the backward packed nothing
    assert feats.grad is not None and conv.weight.grad is not None
    # inference: no dgrad image
    conv2 = SparseConv3d(64, 128, 3).to(dev)
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        conv2(Voxels(c, feats.detach(), offsets=torch.tensor([0, n], dtype=torch.int32)))
    assert list(conv2.weight._wcn_packed) == [(torch.bfloat16, False, False)]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("method", ["binned", "hash"])
def test_duplicate_rows_all_three_gemms_vs_oracle(dtype, method, monkeypatch):
    """Duplicate input coordinates (the smallest row wins every probe): the k-flip shortcut of dgrad does not hold - a
    non-winner row has neighbours but is nobody's neighbour - so the map must route dgrad through the reverse table.
    Map bit-exact vs the C oracle, forward / dgrad / wgrad vs the fp64 pair-list oracle (round-1 ADVICE)."""
    monkeypatch.setenv("WARPCONVNET_AMD_KMAP_METHOD", method)
    base = scene_u(2500, 41, 0)
    s = np.concatenate([base, base[100:400], base[:50]], 0)  # 350 duplicate rows, some coordinates three times
    rng = np.random.default_rng(5)
    s = s[rng.permutation(len(s))]
    km = _kmap(s, s, (3, 3, 3), same=True)
    r = okmap.kernel_map(s, s, (3, 3, 3))
    np.testing.assert_array_equal(km.in_maps.cpu().numpy(), r["in_maps"])
    np.testing.assert_array_equal(km.out_maps.cpu().numpy(), r["out_maps"])
    assert not km._symmetric and not km._self_exact
    dev = _dev()
    g = torch.Generator(device="cpu").manual_seed(77)
    cin, cout = 64, 128
    X = torch.randn(len(s), cin, generator=g).to(dev, dtype)
    W = (torch.randn(27, cin, cout, generator=g) * 0.05).to(dev, dtype)
    dY = torch.randn(len(s), cout, generator=g).to(dev, dtype)
    from warpconvnet_amd.nn.functional.sparse_conv.detail import hip_gemm

    loop = hip_gemm._dgrad_pair_lists
    hip_gemm._dgrad_pair_lists = None  # an odd-kernel submanifold map must stay on the gather kernels (`_dgrad_duplicates`)
    try:
        Y, dX, dW = _run_all(km, X, W, dY, "auto", len(s), len(s))
    finally:
        hip_gemm._dgrad_pair_lists = loop
    Yr, dXr, dWr = _oracle(r, X, W, dY, len(s))
    assert rel_max_err(Y, Yr) < TOL[dtype] and rel_max_err(dX, dXr) < TOL[dtype] and rel_max_err(dW, dWr) < TOL[dtype]
    # rows that lose their coordinate to a smaller row receive no gradient at all
    winners = np.unique(r["in_maps"])
    losers = np.setdiff1d(np.arange(len(s)), winners)
    assert len(losers) > 0 and float(dX[torch.from_numpy(losers).to(dev)].abs().max()) == 0.0


@pytest.mark.parametrize("cout", [128, 256])
@pytest.mark.parametrize("scale", [1.0, 3.0e5])
def test_fp32_features_take_fp16_operand_kernels(scale, cout):
    """fp32 features under `auto`: fp16 operands (exact power-of-two rescale beyond the fp16 range), fp32 accumulation and
    output - the reference's production treatment (mask_gemm.py:72-103); tolerance of its fp32 tests (1e-3).
    grad_output at 3e5 is the GradScaler case the reference documents: finite in, finite out."""
    from warpconvnet_amd.nn.functional.sparse_conv.detail import hip_gemm

    s = np.concatenate([scene_u(3000, 71, 0), scene_u(900, 72, 1)], 0)
    km = _kmap(s, s, (3, 3, 3), same=True)
    r = okmap.kernel_map(s, s, (3, 3, 3))
    dev = _dev()
    g = torch.Generator(device="cpu").manual_seed(9)
    X = (torch.randn(len(s), 64, generator=g) * scale).to(dev)
    W = (torch.randn(27, 64, cout, generator=g) * 0.05).to(dev)
    dY = (torch.randn(len(s), cout, generator=g) * scale).to(dev)
    Y, dX, dW = _run_all(km, X, W, dY, "auto", len(s), len(s))
    assert Y.dtype == torch.float32 and dX.dtype == torch.float32 and dW.dtype == torch.float32
    assert torch.isfinite(Y).all() and torch.isfinite(dX).all() and torch.isfinite(dW).all()
    Yr, dXr, dWr = _oracle(r, X, W, dY, len(s))
    assert rel_max_err(Y, Yr) < 1e-3 and rel_max_err(dX, dXr) < 1e-3 and rel_max_err(dW, dWr) < 1e-3
    # hip_ref keeps fp32 operands: an order of magnitude closer
    Y2, dX2, dW2 = _run_all(km, X, W, dY, "hip_ref", len(s), len(s))
    assert rel_max_err(Y2, Yr) < 1e-5 and rel_max_err(dX2, dXr) < 1e-5 and rel_max_err(dW2, dWr) < 1e-4
    # the scale is a power of two, computed on the device
    x16, sc = hip_gemm.fp16_safe_cast(X)
    assert x16.dtype == torch.float16 and float(torch.log2(sc)) == round(float(torch.log2(sc)))
    assert torch.equal(x16.float() * sc, (X / sc).half().float() * sc)


@pytest.mark.parametrize("ksize,stride,cin,cout", [((5, 5, 5), (1, 1, 1), 32, 64), ((5, 5, 5), (1, 1, 1), 64, 128),
                                                    ((4, 4, 4), (2, 2, 2), 64, 64), ((7, 7, 7), (1, 1, 1), 32, 32)])
def test_mfma_large_kernel_volumes(ksize, stride, cin, cout):
    """K = 125 / 64 / 343 > 32: multi-word masks, one pass of the fused kernel per mask word (hip_mfma forced)."""
    from warpconvnet_amd import _lib

    dtype = torch.bfloat16
    s_in = np.concatenate([scene_u(1500, 61, 0), scene_u(700, 62, 1)], 0)
    same = all(v == 1 for v in stride)
    s_out = s_in if same else okmap.stride_coords(s_in, stride)[0]
    km = _kmap(s_in, s_out, ksize, stride, same=same)
    r = okmap.kernel_map(s_in, s_out, ksize, stride)
    K = int(np.prod(ksize))
    np.testing.assert_array_equal(km.offsets.numpy(), r["offsets"])
    assert _lib.lib().wcn_mfma_gather_supported(cin, cout, K, _lib.WCN_BF16) and km._mask.shape[1] == (K + 31) // 32
    dev = _dev()
    g = torch.Generator(device="cpu").manual_seed(K + cin)
    X = torch.randn(len(s_in), cin, generator=g).to(dev, dtype)
    W = (torch.randn(K, cin, cout, generator=g) * 0.03).to(dev, dtype)
    dY = torch.randn(len(s_out), cout, generator=g).to(dev, dtype)
    Y, dX, dW = _run_all(km, X, W, dY, "hip_mfma", len(s_in), len(s_out))
    Yr, dXr, dWr = _oracle(r, X, W, dY, len(s_out))
    assert rel_max_err(Y, Yr) < TOL[dtype]
    assert rel_max_err(dX, dXr) < TOL[dtype]
    assert rel_max_err(dW, dWr) < TOL[dtype]


@pytest.mark.parametrize("dtype,cin,cout,groups", [(torch.bfloat16, 128, 256, 2), (torch.float16, 64, 64, 4),
                                                    (torch.float32, 24, 36, 3), (torch.bfloat16, 32, 32, 32),
                                                    (torch.bfloat16, 64, 128, 2), (torch.float16, 128, 128, 4)])
def test_grouped_conv_hip_vs_oracle(dtype, cin, cout, groups):
    """Channel groups through the module API on the GPU (auto backend: MFMA where the per-group shape allows, hip_ref
    otherwise, e.g. groups == channels) == G independent convolutions on channel slices (oracle)."""
    from tests.test_host_api import _grouped_oracle
    from warpconvnet_amd.geometry.types.voxels import Voxels
    from warpconvnet_amd.nn.modules.sparse_conv import SparseConv3d

    dev = _dev()
    s = scene_u(2500, 51, 0)
    r = okmap.kernel_map(s, s, (3, 3, 3))
    torch.manual_seed(5)
    conv = SparseConv3d(cin, cout, 3, groups=groups).to(dev)
    X = torch.randn(len(s), cin, device=dev).to(dtype).requires_grad_(True)
    x = Voxels(torch.from_numpy(s[:, 1:]).to(dev), X, offsets=torch.tensor([0, len(s)]))
    y = conv(x) if dtype == torch.float32 else SparseConv3d.forward(conv.to(dtype), x)
    dY = torch.randn(len(s), cout, device=dev).to(dtype)
    y.feature_tensor.backward(dY)
    W4 = conv.weight.detach().double().cpu()
    Yr, dXr, dWr = _grouped_oracle(X.detach().double().cpu(), W4, dY.double().cpu(), r, len(s), conv.bias.detach().double().cpu())
    tol = TOL[dtype]
    assert y.feature_tensor.shape == (len(s), cout) and conv.weight.grad.shape == (27, groups, cin // groups, cout // groups)
    assert rel_max_err(y.feature_tensor.detach(), Yr) < tol
    assert rel_max_err(X.grad, dXr) < tol
    assert rel_max_err(conv.weight.grad, dWr) < tol
    assert rel_max_err(conv.bias.grad, dY.double().sum(0).cpu()) < tol


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("cin,cout", [(3, 32), (7, 13), (23, 33), (33, 65), (65, 7), (16, 48), (48, 160)])
def test_channel_counts_outside_the_mfma_tiles(dtype, cin, cout, monkeypatch):
    """C in {3, 7, 13, 23, 33, 65} (the reference's scalar-load / K-tail tile cases, `mask_gemm.py:495-541`,
    `tests/nn/test_kernel_deterministic.py:10-16`) and widths only the 16x16x32 family has (48, 160): under `auto` all
    three products run on the matrix cores - zero-padded channels where needed - never on the one-thread-per-element
    kernels, and agree with the fp64 oracle."""
    from warpconvnet_amd import _lib

    L = _lib.lib()
    ref_calls = []
    for name in ("wcn_conv_gather_gemm", "wcn_conv_wgrad"):
        real = getattr(L, name)
        idx = {"wcn_conv_gather_gemm": 13, "wcn_conv_wgrad": 12}[name]  # position of the `algo` argument
        monkeypatch.setattr(L, name, (lambda *a, _r=real, _i=idx, _n=name: (ref_calls.append(_n) if a[_i] == _lib.WCN_ALGO_REF else None, _r(*a))[1]))
    s = scene_u(2200, 71, 0)
    km = _kmap(s, s, (3, 3, 3), same=True)
    r = okmap.kernel_map(s, s, (3, 3, 3))
    dev = _dev()
    g = torch.Generator(device="cpu").manual_seed(cin * 100 + cout)
    X = torch.randn(len(s), cin, generator=g).to(dev, dtype)
    W = (torch.randn(27, cin, cout, generator=g) * 0.05).to(dev, dtype)
    dY = torch.randn(len(s), cout, generator=g).to(dev, dtype)
    Y, dX, dW = _run_all(km, X, W, dY, "auto", len(s), len(s))
    assert Y.shape == (len(s), cout) and dX.shape == (len(s), cin) and dW.shape == (27, cin, cout)
    assert Y.is_contiguous() and dX.is_contiguous() and dW.is_contiguous()
    Yr, dXr, dWr = _oracle(r, X, W, dY, len(s))
    tol = TOL[dtype]
    assert rel_max_err(Y, Yr) < tol and rel_max_err(dX, dXr) < tol and rel_max_err(dW, dWr) < tol
    assert not ref_calls, ref_calls


def test_grouped_conv_is_one_launch_per_direction(monkeypatch):
    """Groups whose per-group widths are an MFMA shape: ONE grouped gather-GEMM launch for the forward pass and one for
    dgrad (group index on grid.y) - no per-group launches, no channel-slice copies, no concatenation; `groups == channels`
    takes the depthwise kernels (one launch each way) instead of C one-channel problems."""
    from warpconvnet_amd import _lib
    from warpconvnet_amd.geometry.types.voxels import Voxels
    from warpconvnet_amd.nn.modules.sparse_conv import SparseConv3d

    dev = _dev()
    L = _lib.lib()
    calls = {"grouped": 0, "plain": 0, "dw": 0}
    for name, key in (("wcn_conv_gather_gemm_grouped", "grouped"), ("wcn_conv_gather_gemm", "plain"), ("wcn_dwconv_gather", "dw")):
        real = getattr(L, name)
        monkeypatch.setattr(L, name, (lambda *a, _r=real, _k=key: (calls.__setitem__(_k, calls[_k] + 1), _r(*a))[1]))
    s = scene_u(2000, 52, 0)
    coords = torch.from_numpy(s[:, 1:]).to(dev)
    for cin, cout, groups, want in ((128, 256, 4, {"grouped": 2, "plain": 0, "dw": 0}), (64, 64, 64, {"grouped": 0, "plain": 0, "dw": 2})):
        for k in calls:
            calls[k] = 0
        torch.manual_seed(6)
        conv = SparseConv3d(cin, cout, 3, groups=groups).to(dev).to(torch.bfloat16)
        X = torch.randn(len(s), cin, device=dev).to(torch.bfloat16).requires_grad_(True)
        y = conv(Voxels(coords, X, offsets=torch.tensor([0, len(s)])))
        y.feature_tensor.backward(torch.randn_like(y.feature_tensor))
        assert calls == want, (cin, cout, groups, calls)


@pytest.mark.parametrize("ksize,stride", [((2, 2, 2), (2, 2, 2)), ((3, 3, 3), (2, 2, 2))])
def test_mfma_strided_and_transposed(ksize, stride):
    from warpconvnet_amd.geometry.coords.search.search_results import IntSearchResult

    s = scene_u(6000, 31)
    coarse, _ = okmap.stride_coords(s, stride)
    km = _kmap(s, coarse, ksize, stride)
    r = okmap.kernel_map(s, coarse, ksize, stride)
    dev, dtype = _dev(), torch.bfloat16
    K = int(np.prod(ksize))
    g = torch.Generator().manual_seed(5)
    X = torch.randn(len(s), 64, generator=g).to(dev, dtype)
    W = (torch.randn(K, 64, 128, generator=g) * 0.05).to(dev, dtype)
    dY = torch.randn(len(coarse), 128, generator=g).to(dev, dtype)
    Y, dX, dW = _run_all(km, X, W, dY, "hip_mfma", len(s), len(coarse))
    Yr, dXr, dWr = _oracle(r, X, W, dY, len(coarse))
    assert rel_max_err(Y, Yr) < 2e-2 and rel_max_err(dX, dXr) < 2e-2 and rel_max_err(dW, dWr) < 2e-2
    # transposed conv = same map with in/out swapped (coarse -> fine), weight [K, 128, 64]
    sw = IntSearchResult(km.out_maps, km.in_maps, km.offsets)
    rs = dict(in_maps=r["out_maps"], out_maps=r["in_maps"], offsets=r["offsets"])
    Xt = torch.randn(len(coarse), 128, generator=g).to(dev, dtype)
    Wt = (torch.randn(K, 128, 64, generator=g) * 0.05).to(dev, dtype)
    dYt = torch.randn(len(s), 64, generator=g).to(dev, dtype)
    Y, dX, dW = _run_all(sw, Xt, Wt, dYt, "hip_mfma", len(coarse), len(s))
    Yr, dXr, dWr = _oracle(rs, Xt, Wt, dYt, len(s))
    assert rel_max_err(Y, Yr) < 2e-2 and rel_max_err(dX, dXr) < 2e-2 and rel_max_err(dW, dWr) < 2e-2


def test_known_answer_patterns_gpu(golden_dir):
    """Hand-made weight / feature patterns (reference tests/nn/test_kernel_deterministic.py:128-183):
    fp32 rtol 1e-4 / atol 1e-3, fp16 rtol 8e-3 / atol 5e-2 (:81-84)."""
    from tests.golden.make_golden import make_feats, make_grad_out, make_weight

    g = np.load(os.path.join(golden_dir, "known_answer.npz"))
    s = g["coords"]
    km = _kmap(s, s, (3, 3, 3), same=True)
    np.testing.assert_array_equal(km.in_maps.cpu().numpy(), g["in_maps"])
    dev, n = _dev(), len(s)
    for cin, cout in [(8, 8), (7, 13), (32, 16)]:
        for wp in ["ones", "triu", "tril", "eye", "center_eye"]:
            for fp in ["ones", "range", "row_index"]:
                tag = f"{cin}x{cout}_{wp}_{fp}"
                W = make_weight(27, cin, cout, wp, torch.float32).to(dev)
                X = make_feats(n, cin, fp, torch.float32).to(dev)
                dY = make_grad_out(n, cout, torch.float32).to(dev)
                # fp32 tolerances: the full-fp32 kernels (under `auto`, fp32 features on an MFMA-native shape such as 32 -> 16
                # take fp16 operands like the reference's production path and are held to the 16-bit tolerance elsewhere)
                Y, dX, dW = _run_all(km, X, W, dY, "hip_ref", n, n)
                torch.testing.assert_close(Y.cpu(), torch.from_numpy(g[f"Y_{tag}"]), rtol=1e-4, atol=1e-3)
                torch.testing.assert_close(dX.cpu(), torch.from_numpy(g[f"dX_{tag}"]), rtol=1e-4, atol=1e-3)
                torch.testing.assert_close(dW.cpu(), torch.from_numpy(g[f"dW_{tag}"]), rtol=1e-4, atol=2e-2)
    # half precision on an MFMA-covered shape: a centre-identity weight is an exact pass-through
    W = make_weight(27, 64, 64, "center_eye", torch.float32).to(dev, torch.float16)
    X = make_feats(n, 64, "row_index", torch.float32).to(dev, torch.float16)
    Y, _, _ = _run_all(km, X, W, make_grad_out(n, 64, torch.float32).to(dev, torch.float16), "hip_mfma", n, n)
    assert torch.equal(Y, X)
    W = make_weight(27, 64, 64, "triu", torch.float32).to(dev, torch.float16)
    Y, dX, dW = _run_all(km, X, W, make_grad_out(n, 64, torch.float32).to(dev, torch.float16), "hip_mfma", n, n)
    r = okmap.kernel_map(s, s, (3, 3, 3))
    Yr, dXr, dWr = _oracle(r, X, W, make_grad_out(n, 64, torch.float32).to(dev, torch.float16), n)
    torch.testing.assert_close(Y.double().cpu(), Yr, rtol=8e-3, atol=5e-2)
    torch.testing.assert_close(dX.double().cpu(), dXr, rtol=8e-3, atol=5e-2)


def test_module_forward_backward_amp():
    """SparseConv3d under autocast(bf16): module API end to end, dtype policy, cache reuse, grads vs oracle
    (mean relative diff < 0.02, reference tests/nn/test_sparse_conv.py:921-923)."""
    from warpconvnet_amd.geometry.types.voxels import Voxels
    from warpconvnet_amd.nn.modules.sparse_conv import SparseConv3d

    dev = _dev()
    parts = [scene_u(4000, 41)[:, 1:], scene_u(3000, 42)[:, 1:]]
    feats = [torch.randn(len(p), 64) for p in parts]
    vox = Voxels([torch.from_numpy(p) for p in parts], feats, device=dev)
    torch.manual_seed(0)
    conv = SparseConv3d(64, 128, 3).to(dev)
    x = vox.replace(batched_features=vox.feature_tensor.detach().clone().requires_grad_(True))
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y = conv(x)
    assert y.feature_tensor.dtype == torch.bfloat16 or y.batched_features.dtype == torch.bfloat16
    assert len(x.cache) == 1 and y.cache is x.cache
    loss = (y.batched_features.batched_tensor.float() ** 2).mean()
    loss.backward()
    assert conv.weight.grad.dtype == torch.float32 and conv.weight.grad.shape == conv.weight.shape
    assert x.batched_features.batched_tensor.grad.shape == (7000, 64)
    # oracle in fp64 on the fp32 master values
    bc = vox.batch_indexed_coordinates.cpu().numpy().astype(np.int32)
    r = okmap.kernel_map(bc, bc, (3, 3, 3))
    X = vox.feature_tensor.detach().double().cpu()
    Wd = conv.weight.detach().double().cpu()
    Yr = oconv.forward(X, Wd, r["in_maps"], r["out_maps"], r["offsets"], len(bc)) + conv.bias.detach().double().cpu()
    got = y.batched_features.batched_tensor.detach().double().cpu()
    assert ((got - Yr).abs().mean() / Yr.abs().mean()).item() < 0.02
    dY = (2.0 / Yr.numel()) * got  # d(mean(y^2))/dy at the GPU's y
    dXr, dWr = oconv.backward(dY, X, Wd, r["in_maps"], r["out_maps"], r["offsets"])
    gw = conv.weight.grad.double().cpu()
    assert ((gw - dWr).abs().mean() / dWr.abs().mean()).item() < 0.02
    gx = x.batched_features.batched_tensor.grad.double().cpu()
    assert ((gx - dXr).abs().mean() / dXr.abs().mean()).item() < 0.02
    assert ((conv.bias.grad.double().cpu() - dY.sum(0)).abs().max() / dY.sum(0).abs().max()).item() < 0.02
    # second layer at the same resolution reuses the cached map (no rebuild)
    conv2 = SparseConv3d(128, 64, 3).to(dev)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        z = conv2(y)
    assert len(x.cache) == 1 and z.num_channels == 64


def test_unet_style_down_up():
    """Strided conv then transposed conv back onto the encoder tensor (MinkUNet ConvBlock / ConvTrBlock,
    reference models/mink_unet.py:83-90, 286-339): shapes, tensor strides, map reuse, explicit-vs-HIP agreement."""
    from warpconvnet_amd.geometry.types.voxels import Voxels
    from warpconvnet_amd.nn.modules.sparse_conv import SparseConv3d

    dev = _dev()
    p = scene_surface(100, 3)[:, 1:]
    vox = Voxels([torch.from_numpy(p)], [torch.randn(len(p), 32)], device=dev)
    torch.manual_seed(1)
    down = SparseConv3d(32, 64, 2, stride=2).to(dev)
    up = SparseConv3d(64, 32, 2, stride=2, transposed=True).to(dev)
    outs = {}
    for algo in ("explicit_gemm", "auto"):
        for m in (down, up):
            m.fwd_algo = m.dgrad_algo = type(m.fwd_algo)(algo)
            m.wgrad_algo = type(m.wgrad_algo)(algo)
            m.zero_grad()
        x = vox.replace(batched_features=vox.feature_tensor.detach().clone().to(torch.bfloat16).requires_grad_(True))
        with torch.autocast("cuda", dtype=torch.bfloat16):
            d = down(x)
            u = up(d, x)
        assert d.tensor_stride == (2, 2, 2) and u.tensor_stride == (1, 1, 1)
        assert u.batched_features.batched_tensor.shape == (len(p), 32)
        assert torch.equal(u.coordinate_tensor, x.coordinate_tensor)
        u.batched_features.batched_tensor.float().square().sum().backward()
        outs[algo] = (u.batched_features.batched_tensor.detach().float(), down.weight.grad.clone(), up.weight.grad.clone(),
                      x.batched_features.batched_tensor.grad.float())
    for a, b in zip(outs["explicit_gemm"], outs["auto"]):
        assert rel_max_err(b, a) < 3e-2


def test_empty_inputs_and_errors():
    from warpconvnet_amd.nn.functional.sparse_conv.detail import hip_gemm
    from warpconvnet_amd.nn.functional.sparse_conv.detail.backends import FwdCtx, run_forward

    dev = _dev()
    e = np.zeros((0, 4), np.int32)
    km = _kmap(e, e, (3, 3, 3), same=True)
    W = torch.randn(27, 64, 128, device=dev, dtype=torch.bfloat16)
    Y = hip_gemm.hip_forward(torch.zeros(0, 64, device=dev, dtype=torch.bfloat16), W, km, 0)
    assert Y.shape == (0, 128)
    dW = hip_gemm.hip_wgrad(torch.zeros(0, 64, device=dev, dtype=torch.bfloat16), torch.zeros(0, 128, device=dev, dtype=torch.bfloat16), km, (27, 64, 128))
    assert dW.shape == (27, 64, 128) and float(dW.abs().max()) == 0.0
    s = scene_u(100, 1)
    km = _kmap(s, s, (3, 3, 3), same=True)
    with pytest.raises(RuntimeError):  # explicit request for the MFMA path on an uncovered shape fails loudly
        hip_gemm.hip_forward(torch.zeros(100, 7, device=dev, dtype=torch.bfloat16), torch.zeros(27, 7, 13, device=dev, dtype=torch.bfloat16), km, 100, "hip_mfma")
    with pytest.raises(RuntimeError):  # mixed precision inputs
        hip_gemm.hip_forward(torch.zeros(100, 64, device=dev), W, km, 100)
    with pytest.raises(ValueError):
        run_forward("no_such_algo", FwdCtx(torch.zeros(1, 1), torch.zeros(1, 1, 1), km, 1, None, {}))
    with pytest.raises(RuntimeError):  # CPU tensors never reach a HIP kernel
        hip_gemm.hip_forward(torch.zeros(100, 64, dtype=torch.bfloat16), W.cpu(), km, 100)


def test_full_size_properties():
    """BASELINE config 2 shape (1M voxels, 64 -> 128, bf16): size-independent properties instead of the serial
    oracle - linearity in X, pass-through of a centre-identity weight, dW consistency with an fp32
    torch reference on one offset, sub-sampled rows against the oracle."""
    from warpconvnet_amd.nn.functional.sparse_conv.detail import hip_gemm

    dev, dtype = _dev(), torch.bfloat16
    s = scene_u(1_000_000, 0)
    km = _kmap(s, s, (3, 3, 3), same=True)
    N = len(s)
    g = torch.Generator().manual_seed(0)
    X = torch.randn(N, 64, generator=g).to(dev, dtype)
    W = (torch.randn(27, 64, 128, generator=g) * 0.05).to(dev, dtype)
    dY = torch.randn(N, 128, generator=g).to(dev, dtype)
    Y = hip_gemm.hip_forward(X, W, km, N, "hip_mfma")
    # (1) linearity: conv(2X) == 2 conv(X) exactly (power-of-two scaling commutes with rounding)
    assert torch.equal(hip_gemm.hip_forward(X * 2, W, km, N, "hip_mfma"), Y * 2)
    # (2) centre identity weight is a pass-through
    We = torch.zeros(27, 64, 128, device=dev, dtype=dtype)
    We[13, :, :64] = torch.eye(64, device=dev, dtype=dtype)
    Ye = hip_gemm.hip_forward(X, We, km, N, "hip_mfma")
    assert torch.equal(Ye[:, :64], X) and float(Ye[:, 64:].abs().max()) == 0.0
    # (3) sub-sampled rows against the fp64 oracle (explicit formula per row from the pair table)
    pt = km._pair_table
    rows = torch.randint(0, N, (512,), generator=g).to(dev)
    ref = torch.zeros(512, 128, dtype=torch.float64, device=dev)
    for k in range(27):
        idx = pt[k][rows].long()
        valid = (idx >= 0).double().unsqueeze(1)
        ref += (X[idx.clamp_min(0)].double() * valid) @ W[k].double()
    assert rel_max_err(Y[rows], ref) < 2e-2
    # (4) dgrad is the adjoint of forward: <conv(X), dY> == <X, dgrad(dY)>
    dX = hip_gemm.hip_dgrad(dY, W, km, N, "hip_mfma")
    lhs = (Y.double() * dY.double()).sum().item()
    rhs = (X.double() * dX.double()).sum().item()
    assert abs(lhs - rhs) / abs(lhs) < 2e-2
    # (5) wgrad: <W, dW> == <conv(X), dY>; and bucket 13 equals the dense X^T dY
    dW = hip_gemm.hip_wgrad(X, dY, km, (27, 64, 128), "hip_mfma")
    assert abs((W.double() * dW.double()).sum().item() - lhs) / abs(lhs) < 2e-2
    dense = X.float().T @ dY.float()
    assert rel_max_err(dW[13], dense) < 2e-2
    # (5b) sub-sampled rows of dgrad against fp64: for a submanifold map the pair (in n, out m, k) exists iff
    # (in m, out n, K-1-k) does, so dX[n] = sum_k dY[pair_table[K-1-k][n]] @ W[k]^T
    refx = torch.zeros(512, 64, dtype=torch.float64, device=dev)
    for k in range(27):
        idx = pt[26 - k][rows].long()
        valid = (idx >= 0).double().unsqueeze(1)
        refx += (dY[idx.clamp_min(0)].double() * valid) @ W[k].double().T
    assert rel_max_err(dX[rows], refx) < 2e-2
    # (5c) whole buckets of wgrad against fp64 (corner, edge, centre, opposite corner): dW[k] = X[in_k]^T dY[out_k]
    off = km.offsets.tolist()
    for k in (0, 7, 13, 26):
        i, o = km.in_maps[off[k] : off[k + 1]].long(), km.out_maps[off[k] : off[k + 1]].long()
        want = X[i].double().T @ dY[o].double()
        assert rel_max_err(dW[k], want) < 2e-2, k
    # (6) run-to-run determinism of all three kernels
    assert torch.equal(hip_gemm.hip_forward(X, W, km, N, "hip_mfma"), Y)
    assert torch.equal(hip_gemm.hip_dgrad(dY, W, km, N, "hip_mfma"), dX)
    assert torch.equal(hip_gemm.hip_wgrad(X, dY, km, (27, 64, 128), "hip_mfma"), dW)


def test_feature_tensors_above_2gib_use_64bit_offsets():
    """A [4.6 M, 256] bf16 feature tensor is 2.36 GB: row * C * 2 exceeds 2^31 for the upper rows.  The reference computes
    that gather address in 32-bit int (warpgemm_a_loader_precomputed.cuh:16-17, 103); here every row offset is 64-bit.
    1x1x1 kernel (K = 1) over all rows, so the expected output is a plain matmul; forward, dgrad and wgrad are checked on
    the rows beyond the 2 GiB boundary."""
    from warpconvnet_amd.nn.functional.sparse_conv.detail import hip_gemm

    dev, dtype = _dev(), torch.bfloat16
    N, cin, cout = 4_600_000, 256, 64
    # distinct coordinates on a line; the map of a 1x1x1 kernel is the identity on rows
    coords = torch.zeros(N, 4, dtype=torch.int32)
    idx = torch.arange(N, dtype=torch.int64)
    coords[:, 1], coords[:, 2], coords[:, 3] = (idx % 4096).int(), ((idx // 4096) % 4096).int(), (idx // (4096 * 4096)).int()
    from warpconvnet_amd.geometry.coords.search.torch_discrete import generate_kernel_map

    c = coords.to(dev)
    km = generate_kernel_map(c, c, (1, 1, 1), (1, 1, 1))
    assert int(km.offsets[-1]) == N
    g = torch.Generator().manual_seed(7)
    X = torch.empty(N, cin, dtype=dtype, device=dev)
    assert X.numel() * X.element_size() > 2**31
    X.copy_(torch.randn(1024, cin, generator=g).to(dev, dtype).repeat(N // 1024 + 1, 1)[:N])
    X[-1000:] = torch.randn(1000, cin, generator=g).to(dev, dtype)  # the tail is not a repeat of the head
    W = (torch.randn(1, cin, cout, generator=g) * 0.05).to(dev, dtype)
    Y = hip_gemm.hip_forward(X, W, km, N, "hip_mfma")
    tail = slice(N - 4096, N)
    assert rel_max_err(Y[tail], X[tail].float() @ W[0].float()) < 2e-2
    dY = torch.empty(N, cout, dtype=dtype, device=dev)
    dY.copy_(torch.randn(1024, cout, generator=g).to(dev, dtype).repeat(N // 1024 + 1, 1)[:N])
    dX = hip_gemm.hip_dgrad(dY, W, km, N, "hip_mfma")
    assert rel_max_err(dX[tail], dY[tail].float() @ W[0].float().T) < 2e-2
    dW = hip_gemm.hip_wgrad(X, dY, km, (1, cin, cout), "hip_mfma")
    ref = torch.zeros(cin, cout, dtype=torch.float64, device=dev)
    for a in range(0, N, 1 << 20):
        ref += X[a:a + (1 << 20)].double().T @ dY[a:a + (1 << 20)].double()
    assert rel_max_err(dW[0], ref) < 2e-2


@pytest.mark.parametrize("transposed,ksize,stride", [(False, 3, 1), (False, 2, 2), (True, 2, 2)])
def test_generative_convolution_module(transposed, ksize, stride):
    """generative=True: output coordinates = expanded (strided / up-scaled) inputs, kernel map as the reference builds it
    (helper.py:58-146, 512-540); features vs the oracle on the oracle's own map of the same coordinate sets."""
    from warpconvnet_amd.geometry.types.voxels import Voxels
    from warpconvnet_amd.nn.modules.sparse_conv import SparseConv3d

    dev = _dev()
    s = scene_u(1200, 91, 0)
    torch.manual_seed(2)
    conv = SparseConv3d(16, 32, ksize, stride=stride, transposed=transposed, generative=True).to(dev)
    X = torch.randn(len(s), 16, device=dev, requires_grad=True)
    # a transposed (up-sampling) layer consumes a coarse tensor: tensor stride = conv stride there
    x = Voxels(torch.from_numpy(s[:, 1:]).to(dev), X, offsets=torch.tensor([0, len(s)]),
               tensor_stride=(stride,) * 3 if transposed else None)
    y = conv(x)
    out_c = y.batch_indexed_coordinates.cpu().numpy()
    ks3, st3 = (ksize,) * 3, (stride,) * 3
    from oracle import brute

    offs = brute.kernel_offsets(ks3, (1, 1, 1))
    if transposed:
        base = s * np.array([1, stride, stride, stride], np.int32)
        r = okmap.kernel_map(out_c, base, ks3, (1, 1, 1))  # built out -> in, then swapped
        in_maps, out_maps = r["out_maps"], r["in_maps"]
    else:
        base = s if stride == 1 else okmap.stride_coords(s, st3)[0]
        r = okmap.kernel_map(s, out_c, ks3, st3)
        in_maps, out_maps = r["in_maps"], r["out_maps"]
    want_set = np.unique(np.concatenate([base] + [base + np.concatenate([[0], o]) for o in offs], 0), axis=0)
    np.testing.assert_array_equal(np.unique(out_c, axis=0), want_set)
    assert y.tensor_stride == ((1, 1, 1) if (transposed or stride == 1) else (stride,) * 3)
    dY = torch.randn(len(out_c), 32, device=dev)
    y.feature_tensor.backward(dY)
    Xd, Wd = X.detach().double().cpu(), conv.weight.detach().double().cpu()
    Yr = oconv.forward(Xd, Wd, in_maps, out_maps, r["offsets"], len(out_c)) + conv.bias.detach().double().cpu()
    dXr, dWr = oconv.backward(dY.double().cpu(), Xd, Wd, in_maps, out_maps, r["offsets"])
    assert rel_max_err(y.feature_tensor.detach(), Yr) < 1e-3
    assert rel_max_err(X.grad, dXr) < 1e-3 and rel_max_err(conv.weight.grad, dWr) < 1e-3


@pytest.mark.parametrize("n", [1, 127, 70001])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("cin,cout", [(64, 64), (96, 128), (128, 96), (256, 128), (160, 64), (128, 256), (192, 192), (64, 80),
                                      (128, 320)])
def test_dense_rows_through_the_identity_map(cin, cout, dtype, n):
    """wcn_conv_gather_gemm with nbr = mask = NULL and one offset (include/wcn.h: wcn_conv_identity_supported) is the dense
    product of a 1 x 1 x 1 convolution: x @ w and dy @ w.T vs fp64 on the same 16-bit values, within one rounding of the
    output type (fp32 accumulation); shapes outside the kernel's return None (the caller keeps the vendor GEMM)."""
    from warpconvnet_amd.nn.functional.sparse_conv.pointwise import dense_rows

    dev = _dev()
    torch.manual_seed(n + cin)
    w = torch.randn(1, cin, cout, device=dev) / cin ** 0.5
    x = torch.randn(n, cin, device=dev).to(dtype)
    dy = torch.randn(n, cout, device=dev).to(dtype)
    bias = torch.randn(cout, device=dev)
    wq = w[0].to(dtype).double().cpu()
    from warpconvnet_amd import _lib

    code = _lib.dtype_code(dtype)
    eps = 2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11
    L = _lib.lib()
    y = dense_rows(x, w, False, bias)
    if L.wcn_conv_identity_supported(cin, cout, code) or L.wcn_dense_rows_supported(cin, cout, code):
        # cout: 64 / 96 / 128, or wider in blocks of those; up to 128 -> 96 through the narrow-layer kernel
        assert y is not None and y.dtype == dtype and y.shape == (n, cout)
        want = x.double().cpu() @ wq + bias.double().cpu()
        assert float((y.double().cpu() - want).abs().max()) <= eps * float(want.abs().max()) + 1e-6
    else:
        assert y is None
    dx = dense_rows(dy, w, True)
    if L.wcn_conv_identity_supported(cout, cin, code) or L.wcn_dense_rows_supported(cout, cin, code):
        want = dy.double().cpu() @ wq.t()
        assert dx is not None and float((dx.double().cpu() - want).abs().max()) <= eps * float(want.abs().max()) + 1e-6
    else:
        assert dx is None
    assert (cin, cout) != (160, 64) or dx is None  # 64 -> 160: neither kernel's shape, the caller keeps the vendor GEMM
    assert dense_rows(x.float(), w, False) is None


@pytest.mark.parametrize("n", [1, 33, 70001])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("cin,cout", [(3, 32), (96, 20), (20, 96), (5, 7), (1, 1), (100, 36), (127, 93), (128, 96), (16, 64), (48, 64)])
def test_narrow_layers_as_one_streaming_launch(cin, cout, dtype, n):
    """wcn_dense_rows (include/wcn.h; reference shortcut helper.py:206-213 `feats @ weight[0]`): x @ W (+ bias) and dy @ W^T for
    the stem / head shapes the gather kernels do not take - fp32 or 16-bit weights read in place in either orientation, 16-bit
    or fp32 rows - vs fp64 on the same rounded values, within one rounding of the output type (fp32 accumulation)."""
    from warpconvnet_amd import _lib
    from warpconvnet_amd.nn.functional.sparse_conv.pointwise import narrow_rows

    dev = _dev()
    torch.manual_seed(n * 131 + cin * 7 + cout)
    w = torch.randn(1, cin, cout, device=dev) / cin ** 0.5
    x32 = torch.randn(n, cin, device=dev)
    dy = torch.randn(n, cout, device=dev).to(dtype)
    bias = torch.randn(cout, device=dev)
    eps = 2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11
    assert _lib.lib().wcn_dense_rows_supported(cin, cout, _lib.dtype_code(dtype)) == 1
    wq = w[0].to(dtype).double().cpu()
    xq = x32.to(dtype).double().cpu()
    want = xq @ wq + bias.double().cpu()
    tol = eps * float(want.abs().max()) + 1e-6
    for rows, weight in ((x32.to(dtype), w), (x32.to(dtype), w.to(dtype)), (x32, w), (x32, w.to(dtype))):
        y = narrow_rows(rows, weight, False, bias, dtype=dtype)
        assert y is not None and y.dtype == dtype and y.shape == (n, cout)
        assert float((y.double().cpu() - want).abs().max()) <= tol
    y0 = narrow_rows(x32.to(dtype), w, False)  # (no bias)
    assert float((y0.double().cpu() - xq @ wq).abs().max()) <= tol
    wantx = dy.double().cpu() @ wq.t()
    for weight in (w, w.to(dtype)):
        dx = narrow_rows(dy, weight, True)
        if cin > 96:  # (the input gradient would have more than 96 columns)
            assert dx is None
            continue
        assert dx is not None and dx.shape == (n, cin)
        assert float((dx.double().cpu() - wantx).abs().max()) <= eps * float(wantx.abs().max()) + 1e-6
    # outside the kernel: more than 128 input / 96 output channels, fp32 arithmetic, a weight of a third type
    assert narrow_rows(x32.to(dtype), torch.randn(1, cin, 97, device=dev), False) is None
    assert narrow_rows(x32, w, False) is None
    other = torch.float16 if dtype == torch.bfloat16 else torch.bfloat16
    assert narrow_rows(x32.to(dtype), w.to(other), False) is None


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16, torch.float32])
@pytest.mark.parametrize("cin,cout", [(96, 20), (32, 64), (3, 16), (256, 256), (96, 64), (128, 128), (256, 96)])
def test_pointwise_conv_gradients(cin, cout, dtype):
    """kernel_size = 1: forward / dX dense products, dW through the sparse AtB kernel with the identity pair list
    (channel counts outside the MFMA tiles zero-padded), bias gradient by column sum - vs fp64 on the same values."""
    from warpconvnet_amd.geometry.types.voxels import Voxels
    from warpconvnet_amd.nn.modules.sparse_conv import SparseConv3d

    dev = _dev()
    p = scene_u(20000, 12)[:, 1:]
    torch.manual_seed(cin * 7 + cout)
    conv = SparseConv3d(cin, cout, 1, bias=True).to(dev)
    x = torch.randn(len(p), cin, device=dev).to(dtype).requires_grad_(True)
    vox = Voxels([torch.from_numpy(p)], [x.detach().cpu().float()], device=dev).replace(batched_features=x)
    g = torch.randn(len(p), cout, device=dev).to(dtype)
    with torch.autocast("cuda", dtype=dtype, enabled=dtype != torch.float32):
        y = conv(vox)
    assert y.feature_tensor.dtype == dtype and torch.equal(y.coordinate_tensor, vox.coordinate_tensor)
    y.feature_tensor.backward(g)
    xr = x.detach().double().cpu().requires_grad_(True)
    wq = conv.weight.detach()[0].to(dtype).double().cpu().requires_grad_(True)
    br = conv.bias.detach().double().cpu().requires_grad_(True)
    yr = xr @ wq + br
    yr.backward(g.double().cpu())
    tol = 1e-4 if dtype == torch.float32 else 1.5e-2
    assert rel_max_err(y.feature_tensor.detach(), yr.detach()) < tol
    assert rel_max_err(x.grad, xr.grad) < tol
    assert rel_max_err(conv.weight.grad[0], wq.grad) < tol
    assert rel_max_err(conv.bias.grad, br.grad) < tol
    assert conv.weight.grad.dtype == torch.float32 and conv.weight.grad.shape == (1, cin, cout)


def test_config4_batch_of_lidar_scale_scenes():
    """BASELINE config 4 shape per GPU: 8 scenes x ~35 k voxels with anisotropic extent (wide in x / y, shallow in z) in one
    batch - kernel map bit-exact (no pair crosses scenes), SparseConv3d forward / backward vs the oracle."""
    from warpconvnet_amd.geometry.types.voxels import Voxels
    from warpconvnet_amd.nn.modules.sparse_conv import SparseConv3d

    dev = _dev()
    rng = np.random.default_rng(44)
    parts = []
    for b in range(8):
        n = int(rng.integers(30000, 40000))
        c = np.stack([rng.integers(0, 400, size=2 * n), rng.integers(0, 400, size=2 * n), rng.integers(0, 12, size=2 * n)], 1)
        _, first = np.unique(c, axis=0, return_index=True)
        parts.append(c[np.sort(first)][:n].astype(np.int32))
    feats = [torch.randn(len(p), 32) for p in parts]
    vox = Voxels([torch.from_numpy(p) for p in parts], feats, device=dev)
    torch.manual_seed(4)
    conv = SparseConv3d(32, 64, 3).to(dev)
    x = vox.replace(batched_features=vox.feature_tensor.detach().clone().requires_grad_(True))
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y = conv(x)
    g = torch.randn_like(y.feature_tensor)
    y.feature_tensor.backward(g)
    bc = vox.batch_indexed_coordinates.cpu().numpy().astype(np.int32)
    r = okmap.kernel_map(bc, bc, (3, 3, 3))
    km = next(iter(x.cache.values()))
    np.testing.assert_array_equal(km.in_maps.cpu().numpy(), r["in_maps"])
    np.testing.assert_array_equal(km.out_maps.cpu().numpy(), r["out_maps"])
    assert (bc[r["in_maps"], 0] == bc[r["out_maps"], 0]).all()  # every pair stays inside its scene
    X = vox.feature_tensor.detach().to(torch.bfloat16).double().cpu()
    Wd = conv.weight.detach().to(torch.bfloat16).double().cpu()
    Yr = oconv.forward(X, Wd, r["in_maps"], r["out_maps"], r["offsets"], len(bc)) + conv.bias.detach().double().cpu()
    assert rel_max_err(y.feature_tensor.detach(), Yr) < 2e-2
    dXr, dWr = oconv.backward(g.double().cpu(), X, Wd, r["in_maps"], r["out_maps"], r["offsets"])
    assert rel_max_err(x.feature_tensor.grad, dXr) < 2e-2 and rel_max_err(conv.weight.grad, dWr) < 2e-2
    assert y.offsets.tolist() == vox.offsets.tolist() and len(vox.offsets) == 9


@pytest.mark.parametrize("case", ["table_full", "duplicates_out_of_order"])
def test_module_path_relaunches_after_a_rejected_optimistic_build(case):
    """`SparseConv3d` under autocast queues its forward kernel on the tables of an OPTIMISTIC build and validates afterwards
    (`hip_gemm.hip_forward`): when the device rejects the first build - (i) block table too small for a very sparse scene,
    (ii) duplicate coordinates whose later copy won a cell - the map is rebuilt and the forward repeated.  Hints reset, so
    the branch is taken whatever ran before; Y / dX / dW against the fp64 oracle on the oracle's map."""
    from warpconvnet_amd.geometry.coords.search.search_results import IntSearchResult
    from warpconvnet_amd.geometry.coords.search.torch_discrete import default_hints
    from warpconvnet_amd.geometry.types.voxels import Voxels
    from warpconvnet_amd.nn.modules.sparse_conv import SparseConv3d

    dev = _dev()
    rng = np.random.default_rng(17)
    if case == "table_full":
        cells = rng.permutation(36 * 36 * 36)[:24000]  # one voxel per 8^3 block, neighbours across block faces
        c = np.stack([cells // 1296 * 8 + 7 * (cells % 2), cells // 36 % 36 * 8, cells % 36 * 8], 1).astype(np.int32)
    else:
        base = scene_u(5000, 31)[:, 1:]
        c = np.concatenate([base[:400][::-1], base], 0).astype(np.int32)  # the LATER copy of 400 coordinates comes first
    default_hints().reset()
    seen = []
    real_validate = IntSearchResult.validate

    def spy(self):
        r = real_validate(self)
        seen.append(r)
        return r

    torch.manual_seed(0)
    conv = SparseConv3d(64, 128, 3).to(dev)
    feats = torch.randn(len(c), 64)
    vox = Voxels([torch.from_numpy(c)], [feats], device=dev)
    x = vox.replace(batched_features=vox.feature_tensor.detach().clone().requires_grad_(True))
    IntSearchResult.validate = spy
    try:
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y = conv(x)
        g = torch.randn(len(c), 128, device=dev)
        y.feature_tensor.backward(g.to(y.feature_tensor.dtype))
    finally:
        IntSearchResult.validate = real_validate
    if case == "table_full":
        assert True in seen, "the sparse scene must have taken the TABLE_FULL rebuild"
    bc = np.concatenate([np.zeros((len(c), 1), np.int32), c], 1)
    r = okmap.kernel_map(bc, bc, (3, 3, 3))
    km = next(iter(x.cache.values()))
    np.testing.assert_array_equal(km._pair_table.cpu().numpy(), r["found"])
    np.testing.assert_array_equal(km.in_maps.cpu().numpy(), r["in_maps"])
    Wd = conv.weight.detach().to(torch.bfloat16).double().cpu()
    X = feats.to(torch.bfloat16).double()
    Yr = oconv.forward(X, Wd, r["in_maps"], r["out_maps"], r["offsets"], len(bc)) + conv.bias.detach().double().cpu()
    assert rel_max_err(y.feature_tensor, Yr) < 2e-2
    dY = g.to(torch.bfloat16).double().cpu()
    dXr, dWr = oconv.backward(dY, X, Wd, r["in_maps"], r["out_maps"], r["offsets"])
    assert rel_max_err(x.feature_tensor.grad, dXr) < 2e-2
    assert rel_max_err(conv.weight.grad, dWr) < 2e-2


def test_rejected_build_is_not_served_from_the_cache():
    """A coordinate outside the packed range: the convolution raises ValueError (reference `_packed_base.py:102-122`), the
    map it had cached optimistically is evicted, and a second call on the same tensor raises again instead of running on
    half-built tables."""
    from warpconvnet_amd.geometry.types.voxels import Voxels
    from warpconvnet_amd.nn.modules.sparse_conv import SparseConv3d

    dev = _dev()
    c = scene_u(3000, 41)[:, 1:].copy()
    c[7, 0] = 131072
    conv = SparseConv3d(64, 128, 3).to(dev)
    vox = Voxels([torch.from_numpy(c)], [torch.randn(len(c), 64)], device=dev)
    for _ in range(2):
        with pytest.raises(ValueError):
            with torch.autocast("cuda", dtype=torch.bfloat16):
                conv(vox)
        assert vox.cache is None or len(vox.cache) == 0


def test_weight_gradient_lands_in_the_gradient_bucket_slot():
    """`dist.GradientBuckets` publishes every parameter's view of the flat bucket as `_wcn_grad_slot`; with `.grad` None
    the convolution's weight-gradient kernel writes there and autograd adopts the alias - same values as the plain path,
    `.grad` storage inside the bucket, a second (accumulating) backward still adds."""
    from warpconvnet_amd.dist import GradientBuckets
    from warpconvnet_amd.geometry.types.voxels import Voxels
    from warpconvnet_amd.nn.modules.sparse_conv import SparseConv3d

    dev = _dev()
    c = scene_u(4000, 51)[:, 1:]
    torch.manual_seed(0)
    conv = SparseConv3d(64, 128, 3).to(dev)
    feats = torch.randn(len(c), 64, device=dev)
    g = torch.randn(len(c), 128, device=dev).bfloat16()

    def run():
        x = Voxels([torch.from_numpy(c)], [feats], device=dev)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y = conv(x)
        y.feature_tensor.backward(g)

    run()
    ref_w, ref_b = conv.weight.grad.clone(), conv.bias.grad.clone()
    buckets = GradientBuckets(conv.parameters())
    flat = buckets._buckets[0]["flat"]
    buckets.zero_grad()
    assert conv.weight.grad is None
    flat.fill_(float("nan"))  # whatever is not written by this backward would show
    run()
    buckets.finish()
    lo, hi = flat.data_ptr(), flat.data_ptr() + flat.numel() * 4
    assert lo <= conv.weight.grad.data_ptr() < hi and lo <= conv.bias.grad.data_ptr() < hi
    assert torch.equal(conv.weight.grad, ref_w) and torch.equal(conv.bias.grad, ref_b)
    with buckets.no_sync():
        run()  # .grad is set now: the ordinary accumulation path
    assert torch.allclose(conv.weight.grad, 2 * ref_w, rtol=1e-6, atol=0)
    buckets.remove()
    assert not hasattr(conv.weight, "_wcn_grad_slot")


@pytest.mark.parametrize("fused", [False, True])
def test_a_weight_used_twice_in_one_graph_sums_its_gradients_under_gradient_buckets(fused, monkeypatch):
    """One conv (plain module, and inside the fused conv -> BN -> ReLU node) applied TWICE in a graph, gradients None,
    buckets attached: `.grad` is None when both weight-gradient nodes run, so only the first may write into the bucket
    slot (`dist.claim_grad_slot`); the second writes a fresh tensor and the engine adds them.  Expected = the same graph
    without buckets (advisor finding, round 4: N x the last gradient instead of the sum)."""
    from warpconvnet_amd.dist import GradientBuckets
    from warpconvnet_amd.geometry.types.voxels import Voxels
    from warpconvnet_amd.nn.modules.sequential import Sequential
    from warpconvnet_amd.nn.modules.sparse_conv import SparseConv3d

    monkeypatch.setenv("WARPCONVNET_AMD_FUSED_BLOCK", "1" if fused else "0")
    dev = _dev()
    c = scene_u(3000, 77)[:, 1:]
    torch.manual_seed(1)
    conv = SparseConv3d(64, 64, 3, bias=False).to(dev)
    net = Sequential(conv, torch.nn.BatchNorm1d(64), torch.nn.ReLU()).to(dev) if fused else conv
    feats = torch.randn(len(c), 64, device=dev)
    g = torch.randn(len(c), 64, device=dev).bfloat16()

    def run():
        x = Voxels([torch.from_numpy(c)], [feats], device=dev)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y = net(net(x))  # the same weight feeds two nodes of this graph
        y.feature_tensor.backward(g.to(y.feature_tensor.dtype))

    run()
    ref = conv.weight.grad.clone()
    assert ref.abs().max() > 0
    for p in net.parameters():
        p.grad = None
    buckets = GradientBuckets(net.parameters())
    buckets.zero_grad()
    buckets._buckets[0]["flat"].fill_(float("nan"))
    run()
    buckets.finish()
    assert torch.allclose(conv.weight.grad, ref, rtol=1e-5, atol=1e-6 * float(ref.abs().max()))
    assert not getattr(conv.weight, "_wcn_grad_claimed", False)
    buckets.remove()
