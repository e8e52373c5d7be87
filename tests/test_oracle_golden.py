"""CPU: the oracle against the golden vectors produced by the reference itself, and against itself."""
import glob
import os

import numpy as np
import pytest
import torch

from oracle import brute, conv, kmap
from tests.util import scene_u


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def test_offset_tables_match_reference(golden_dir):
    g = _load(golden_dir, "offset_tables.npz")
    assert len(g.files) == 10
    for name in g.files:
        ks = tuple(int(c) for c in name[1:4])
        dl = tuple(int(c) for c in name[6:9])
        ref = g[name]
        assert ref.shape == (int(np.prod(ks)), 4)
        assert (ref[:, 0] == 0).all()
        np.testing.assert_array_equal(brute.kernel_offsets(ks, dl), ref[:, 1:])
    t = g["k333_d111"]
    assert t[0].tolist() == [0, -1, -1, -1] and t[13].tolist() == [0, 0, 0, 0] and t[26].tolist() == [0, 1, 1, 1]
    assert g["k222_d111"][-1].tolist() == [0, 1, 1, 1]  # even kernels probe {0, 1}


@pytest.mark.parametrize("name", sorted(os.path.basename(p) for p in glob.glob(os.path.join(os.path.dirname(__file__), "golden", "explicit_*.npz"))))
def test_conv_oracle_matches_reference_explicit(golden_dir, name):
    g = _load(golden_dir, name)
    iden = int(g["identity"])
    iden = None if iden < 0 else iden
    Y = conv.forward(g["X"], g["W"], g["in_maps"], g["out_maps"], g["offsets"], g["out_coords"].shape[0], iden)
    dX, dW = conv.backward(g["dY"], g["X"], g["W"], g["in_maps"], g["out_maps"], g["offsets"], iden)
    tol = 1e-12 if g["X"].dtype == np.float64 else 1e-6
    for got, want in ((Y, g["Y"]), (dX, g["dX"]), (dW, g["dW"])):
        want = torch.from_numpy(want)
        assert got.dtype == want.dtype and got.shape == want.shape
        assert (got - want).abs().max().item() <= tol * max(1.0, want.abs().max().item())


@pytest.mark.parametrize("name", sorted(os.path.basename(p) for p in glob.glob(os.path.join(os.path.dirname(__file__), "golden", "depthwise_*.npz"))))
def test_depthwise_oracle_matches_reference_explicit(golden_dir, name):
    """oracle.conv.depthwise_* vs the reference's `_explicit_depthwise_{forward,backward}_logic` outputs."""
    g = _load(golden_dir, name)
    iden = int(g["identity"])
    iden = None if iden < 0 else iden
    Y = conv.depthwise_forward(g["X"], g["W"], g["in_maps"], g["out_maps"], g["offsets"], g["out_coords"].shape[0], iden)
    dX, dW = conv.depthwise_backward(g["dY"], g["X"], g["W"], g["in_maps"], g["out_maps"], g["offsets"], iden)
    tol = 1e-12 if g["X"].dtype == np.float64 else 1e-6
    for got, want in ((Y, g["Y"]), (dX, g["dX"]), (dW, g["dW"])):
        want = torch.from_numpy(want)
        assert got.dtype == want.dtype and got.shape == want.shape
        assert (got - want).abs().max().item() <= tol * max(1.0, want.abs().max().item())


def test_kmap_oracle_reproduces_golden_maps(golden_dir):
    """The maps stored in the fixtures were made by the brute-force builder; the C restatement must agree."""
    for name in ("explicit_u2048_16x32_f32.npz", "explicit_b2_7x13_f32_noiden.npz", "explicit_stride2_k2_16x32_f32.npz"):
        g = _load(golden_dir, name)
        r = kmap.kernel_map(g["in_coords"], g["out_coords"], g["ksize"], g["stride"])
        np.testing.assert_array_equal(r["offsets"], g["offsets"])
        np.testing.assert_array_equal(r["in_maps"], g["in_maps"])
        np.testing.assert_array_equal(r["out_maps"], g["out_maps"])


def test_known_answer_patterns(golden_dir):
    from tests.golden.make_golden import make_feats, make_grad_out, make_weight

    g = _load(golden_dir, "known_answer.npz")
    n = g["coords"].shape[0]
    for cin, cout in [(8, 8), (7, 13), (32, 16)]:
        for wp in ["ones", "triu", "tril", "eye", "center_eye"]:
            for fp in ["ones", "range", "row_index"]:
                W = make_weight(27, cin, cout, wp, torch.float64)
                X = make_feats(n, cin, fp, torch.float64)
                dY = make_grad_out(n, cout, torch.float64)
                Y = conv.forward(X, W, g["in_maps"], g["out_maps"], g["offsets"], n, 13)
                dX, dW = conv.backward(dY, X, W, g["in_maps"], g["out_maps"], g["offsets"], 13)
                tag = f"{cin}x{cout}_{wp}_{fp}"
                np.testing.assert_allclose(Y.numpy(), g[f"Y_{tag}"], rtol=1e-6, atol=1e-6)
                np.testing.assert_allclose(dX.numpy(), g[f"dX_{tag}"], rtol=1e-6, atol=1e-6)
                np.testing.assert_allclose(dW.numpy(), g[f"dW_{tag}"], rtol=1e-6, atol=1e-5)
    # analytic check: centre-identity weight on a submanifold map is a pass-through
    Y = conv.forward(make_feats(n, 8, "row_index", torch.float64), make_weight(27, 8, 8, "center_eye", torch.float64),
                     g["in_maps"], g["out_maps"], g["offsets"], n, 13)
    np.testing.assert_allclose(Y.numpy(), make_feats(n, 8, "row_index", torch.float64).numpy())


@pytest.mark.parametrize("ksize,stride", [((3, 3, 3), (1, 1, 1)), ((2, 2, 2), (2, 2, 2)), ((3, 3, 3), (2, 2, 2)), ((5, 5, 5), (1, 1, 1)), ((3, 1, 2), (1, 1, 1))])
def test_kmap_oracle_vs_bruteforce_and_invariants(ksize, stride):
    """No missing / no spurious pairs, invariant in = stride*out + offset[k], exactly-once coverage
    (reference tests/coords/test_kernel_map_invariants.py:94-277)."""
    a = np.concatenate([scene_u(400, 7, 0), scene_u(350, 8, 1)], 0)
    out = a if all(s == 1 for s in stride) else kmap.stride_coords(a, stride)[0]
    r = kmap.kernel_map(a, out, ksize, stride)
    f, o, i, om = brute.kernel_map(a, out, ksize, stride)
    np.testing.assert_array_equal(r["found"], f)
    np.testing.assert_array_equal(r["offsets"], o)
    np.testing.assert_array_equal(r["in_maps"], i)
    np.testing.assert_array_equal(r["out_maps"], om)
    offs = brute.kernel_offsets(ksize)
    K = len(offs)
    assert r["offsets"][-1] == len(r["in_maps"]) == len(r["out_maps"])
    st = np.asarray(stride)
    for k in range(K):
        s, e = r["offsets"][k], r["offsets"][k + 1]
        ii, oo = r["in_maps"][s:e], r["out_maps"][s:e]
        np.testing.assert_array_equal(a[ii, 1:], out[oo, 1:] * st + offs[k])
        np.testing.assert_array_equal(a[ii, 0], out[oo, 0])  # never across batch elements
        assert len(np.unique(oo)) == len(oo)  # an output row appears at most once per offset
        assert (np.diff(oo) > 0).all()  # canonical order
    # mask and reverse table are consistent with found
    for k in range(K):
        np.testing.assert_array_equal((r["mask"][:, k // 32] >> (k % 32)) & 1, (r["found"][k] >= 0).astype(np.uint32))
        rows = np.nonzero(r["found"][k] >= 0)[0]
        np.testing.assert_array_equal(r["rev"][k][r["found"][k][rows]], rows)
    if all(s == 1 for s in stride) and all(k % 2 == 1 for k in ksize):
        np.testing.assert_array_equal(r["rev"], r["found"][::-1])  # submanifold symmetry used by dgrad
        np.testing.assert_array_equal(r["found"][K // 2], np.arange(len(a)))


def test_hash_table_contract():
    """Value = insertion row index, -1 on miss, duplicates keep the first, bounds, capacity rule
    (reference tests/coords/test_packed_hashmap.py:70-133, 283-340)."""
    c = scene_u(500, 3)
    t = kmap.HashTable(c)
    assert t.capacity == 1024 and (t.capacity & (t.capacity - 1)) == 0
    np.testing.assert_array_equal(t.search(c), np.arange(len(c)))
    miss = c.copy()
    miss[:, 1] += 1000
    assert (t.search(miss) == -1).all()
    dup = np.concatenate([c[:10], c[:10], c[10:20]], 0)
    td = kmap.HashTable(dup)
    np.testing.assert_array_equal(td.search(dup), np.concatenate([np.arange(10), np.arange(10), np.arange(20, 30)]))
    edge = np.array([[511, 131071, -131072, 0], [0, -131072, 131071, 5]], np.int32)
    np.testing.assert_array_equal(kmap.HashTable(edge).search(edge), [0, 1])
    for bad in ([[512, 0, 0, 0]], [[-1, 0, 0, 0]], [[0, 131072, 0, 0]], [[0, 0, -131073, 0]]):
        with pytest.raises(ValueError):
            kmap.HashTable(np.asarray(bad, np.int32))
    with pytest.raises(RuntimeError):
        kmap.HashTable(c[:40], capacity=32)  # 40 distinct keys do not fit 32 slots


def test_stride_coords_oracle():
    a = np.concatenate([scene_u(300, 1, 0), scene_u(300, 2, 1)], 0)
    a[:, 1:] -= 4  # negative coordinates: floor, not truncation
    out, first = kmap.stride_coords(a, (2, 2, 2))
    want = np.floor_divide(a[:, 1:], 2)
    np.testing.assert_array_equal(out[:, 1:], want[first])
    full = np.concatenate([a[:, :1], want], 1)
    assert len(np.unique(full, axis=0)) == len(out)
    assert (np.diff(first) > 0).all() and (np.diff(out[:, 0]) >= 0).all()


def test_morton_and_pool_oracle_known_answers():
    """The numpy restatement of the z-order codes / orderings / sparse pooling against hand-computed values and the
    properties the reference's tests pin (tests/coords/test_serialization.py:67-135, 193-240)."""
    from oracle import serialization as oser

    pts = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1], [1, 1, 1], [2, 0, 0], [3, 5, 7]], np.int32)
    np.testing.assert_array_equal(oser.morton_code(pts), [0, 1, 2, 4, 7, 8, 431])  # x lowest bit; (3,5,7) = 7 + 5*8 + 6*64
    np.testing.assert_array_equal(oser.morton_code(pts + 11), oser.morton_code(pts))  # per-column minimum removed
    # axis permutation == encoding the permuted columns with xyz (test_serialization.py:116-135)
    np.testing.assert_array_equal(oser.morton_code(pts, "morton_yxz"), oser.morton_code(pts[:, [1, 0, 2]]))
    np.testing.assert_array_equal(oser.morton_code(pts, "morton_zyx"), oser.morton_code(pts[:, [2, 1, 0]]))
    # batched: (b << 48) | 16-bit interleave
    b = np.array([[0, 0, 0, 0], [2, 1, 1, 1], [1, 65535, 65535, 65535]], np.int32)
    np.testing.assert_array_equal(oser.morton_code(b), [0, (2 << 48) | 7, (1 << 48) | ((1 << 48) - 1)])
    # 21-bit single-batch range: the largest coordinate fills all 63 bits
    big = np.array([[0, 0, 0], [(1 << 21) - 1] * 3], np.int32)
    assert oser.morton_code(big)[1] == (1 << 63) - 1
    # per-batch permutation: sorted inside every segment, segments stay put, distinct coordinates -> distinct codes
    rng = np.random.default_rng(0)
    c = np.unique(rng.integers(0, 40, size=(500, 3)), axis=0).astype(np.int32)
    rng.shuffle(c)
    offs = np.array([0, 100, 101, len(c)])
    codes, perm = oser.encode_perm(c, offs)
    assert len(np.unique(codes)) == len(c) and sorted(perm.tolist()) == list(range(len(c)))
    for i in range(3):
        seg = perm[offs[i] : offs[i + 1]]
        assert ((seg >= offs[i]) & (seg < offs[i + 1])).all() and (np.diff(codes[seg]) > 0).all()
    # pooling: two windows, one empty output row
    f = np.array([[1.0, -2.0], [3.0, 5.0], [-4.0, 0.5]])
    im, om = np.array([0, 1, 2]), np.array([0, 0, 2])
    np.testing.assert_array_equal(oser.sparse_reduce(f, im, om, 3, "max"), [[3, 5], [0, 0], [-4, 0.5]])
    np.testing.assert_array_equal(oser.sparse_reduce(f, im, om, 3, "min"), [[1, -2], [0, 0], [-4, 0.5]])
    np.testing.assert_array_equal(oser.sparse_reduce(f, im, om, 3, "sum"), [[4, 3], [0, 0], [-4, 0.5]])
    np.testing.assert_array_equal(oser.sparse_reduce(f, im, om, 3, "mean"), [[2, 1.5], [0, 0], [-4, 0.5]])


def test_compact_rows_restatement_round_trips():
    """`oracle.kmap.compact_rows` (the format of the product's compact neighbour table, checked against it on the GPU in
    tests/test_gpu_kmap.py) is the inverse of `densify_rows` on rows that fit, flags the ones that do not, and agrees with the
    oracle's own masks."""
    from oracle import kmap as okmap
    from tests.util import scene_u

    c = scene_u(4000, 3)
    r = okmap.kernel_map(c, c, (3, 3, 3))
    rows, mask, fits = okmap.compact_rows(r["found"])
    assert fits.all() and rows.shape == (len(c), 16)
    np.testing.assert_array_equal(mask, r["mask"][:, 0])
    np.testing.assert_array_equal(okmap.densify_rows(rows, 27).T, r["found"])
    # a dense cube: interior rows have 27 neighbours and do not fit
    g = np.stack(np.meshgrid(np.arange(5), np.arange(5), np.arange(5), indexing="ij"), -1).reshape(-1, 3)
    cc = np.concatenate([np.zeros((len(g), 1), np.int64), g], 1).astype(np.int32)
    rd = okmap.kernel_map(cc, cc, (3, 3, 3))
    rows_d, mask_d, fits_d = okmap.compact_rows(rd["found"])
    # 27 interior voxels (27 neighbours incl. themselves) + 6 x 9 face voxels (18) do not fit; edges (12) and corners (8) do
    assert int((~fits_d).sum()) == 27 + 54
    np.testing.assert_array_equal(mask_d, rd["mask"][:, 0])
    ok_rows = np.nonzero(fits_d)[0]
    np.testing.assert_array_equal(okmap.densify_rows(rows_d[ok_rows], 27), rd["found"].T[ok_rows])
