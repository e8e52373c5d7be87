"""GPU: HIP kernel-map construction vs the oracle (bit-exact indices), through the C-ABI."""
import numpy as np
import pytest
import torch

from oracle import kmap as okmap
from tests.util import scene_surface, scene_u, sort_buckets, tile_key, tile_order_key

pytestmark = pytest.mark.gpu


def _dev():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return torch.device("cuda:0")


def _gen(in_np, out_np, ksize, stride=(1, 1, 1), dilation=None, same=False):
    from warpconvnet_amd.geometry.coords.search.torch_discrete import generate_kernel_map

    dev = _dev()
    a = torch.from_numpy(in_np).to(dev)
    b = a if same else torch.from_numpy(out_np).to(dev)
    return generate_kernel_map(a, b, stride, ksize, dilation)


def _check_against_oracle(km, in_np, out_np, ksize, stride=(1, 1, 1), dilation=(1, 1, 1)):
    r = okmap.kernel_map(in_np, out_np, ksize, stride, dilation)
    K = len(r["offsets"]) - 1
    # pair table bit-exact (reference layout [K, M])
    np.testing.assert_array_equal(km._pair_table.cpu().numpy(), r["found"])
    # row-major table: same content, pad columns are -1
    nbr = km._nbr.cpu().numpy()
    np.testing.assert_array_equal(nbr[:, :K].T, r["found"])
    assert (nbr[:, K:] == -1).all()
    if km._nbrc is not None:  # binned builder: compact rows - the mask, then the neighbours of the set offsets in ascending k
        c = km._nbrc.cpu().numpy()
        assert c.shape == (len(out_np), 16)
        np.testing.assert_array_equal(c[:, 0].view(np.uint32), r["mask"][:, 0])
        want, _, fits = okmap.compact_rows(r["found"])  # numpy restatement of the format (oracle/kmap.py)
        assert fits.all()
        valid = np.arange(16)[None, :] <= np.array([bin(int(m)).count("1") for m in r["mask"][:, 0]])[:, None]
        np.testing.assert_array_equal(np.where(valid, c, 0), want)  # (words behind the last neighbour are unspecified)

    np.testing.assert_array_equal(km.offsets.numpy(), r["offsets"])
    np.testing.assert_array_equal(km._offsets_dev.cpu().numpy(), r["offsets"])
    # buckets come out ordered by output row -> equal to the canonical oracle order without sorting
    np.testing.assert_array_equal(km.in_maps.cpu().numpy(), r["in_maps"])
    np.testing.assert_array_equal(km.out_maps.cpu().numpy(), r["out_maps"])
    np.testing.assert_array_equal(km._mask.cpu().numpy().view(np.uint32), r["mask"])
    # permutation: a permutation of all rows in the tile order of the gather GEMMs - the stable descending sort by `tile_key`
    # (odd kernel volumes up to 31: pair / Gray key; else the mask word itself, i.e. the reference's mask_argsort order)
    perm = km._perm.cpu().numpy()
    assert sorted(perm.tolist()) == list(range(len(out_np)))
    key = tile_order_key(r["mask"][:, 0], K) if r["mask"].shape[1] == 1 else r["mask"][:, 0].astype(np.int64)
    np.testing.assert_array_equal(perm, np.argsort(-key, kind="stable"))
    return r


@pytest.mark.parametrize("n", [1, 63, 2049, 1_200_000])
def test_row_orders_at_the_edges_of_the_sort_plan(n):
    """One row, less than a wave, one key past a tile, and more than 512 sort tiles (the scan of the per-(tile, digit) counts then
    leaves its register-resident path): exact and tile orders of a 27-offset mask."""
    from warpconvnet_amd import _lib

    rng = np.random.default_rng(n)
    m = (rng.integers(0, 1 << 27, size=n, dtype=np.int64) & rng.integers(0, 1 << 27, size=n, dtype=np.int64)).astype(np.uint32)
    m |= np.uint32(1 << 13)
    mask = torch.from_numpy(m.view(np.int32)).to(_dev()).view(n, 1)
    L = _lib.lib()
    ws = torch.empty(L.wcn_mask_argsort_workspace(n), dtype=torch.uint8, device=_dev())
    perm = torch.empty(n, dtype=torch.int32, device=_dev())
    stream = _lib.stream_handle(_dev())
    _lib.check(L.wcn_mask_argsort(_lib.ptr(mask), 1, 27, n, _lib.ptr(perm), _lib.ptr(ws), ws.numel(), stream), "argsort")
    np.testing.assert_array_equal(perm.cpu().numpy(), np.argsort(-m.astype(np.int64), kind="stable"))
    _lib.check(L.wcn_mask_tile_order(_lib.ptr(mask), 1, 27, n, _lib.ptr(perm), _lib.ptr(ws), ws.numel(), stream), "tile order")
    np.testing.assert_array_equal(perm.cpu().numpy(), np.argsort(-tile_order_key(m, 27), kind="stable"))


@pytest.mark.parametrize("K", [27, 9, 25, 31, 3, 8, 32, 26])
def test_row_orders_of_the_c_abi(K):
    """wcn_mask_argsort keeps the reference's order (descending mask, stable: mask_data_kernels.cu:187-220); wcn_mask_tile_order
    is the stable descending sort by `tile_order_key` (tests/util.py restates csrc/mask_sort.h: `tile_key` for odd volumes up to
    31, its top 20 bits for volumes above 18) and the same as wcn_mask_argsort otherwise.  Ragged size: the last sort tile is partial."""
    from warpconvnet_amd import _lib

    rng = np.random.default_rng(K)
    n = 70_001
    dense = rng.random((n, K)) < rng.choice([0.1, 0.4], size=(n, 1))  # a sparse and a dense population
    m = (dense * (1 << np.arange(K, dtype=np.int64))).sum(1).astype(np.uint32)
    if K % 2 == 1:
        m |= np.uint32(1 << (K // 2))
    mask = torch.from_numpy(m.view(np.int32)).to(_dev()).view(n, 1)
    L = _lib.lib()
    ws = torch.empty(L.wcn_mask_argsort_workspace(n), dtype=torch.uint8, device=_dev())
    perm = torch.empty(n, dtype=torch.int32, device=_dev())
    stream = _lib.stream_handle(_dev())
    _lib.check(L.wcn_mask_argsort(_lib.ptr(mask), 1, min(K, 32), n, _lib.ptr(perm), _lib.ptr(ws), ws.numel(), stream), "argsort")
    np.testing.assert_array_equal(perm.cpu().numpy(), np.argsort(-m.astype(np.int64), kind="stable"))
    _lib.check(L.wcn_mask_tile_order(_lib.ptr(mask), 1, K, n, _lib.ptr(perm), _lib.ptr(ws), ws.numel(), stream), "tile order")
    np.testing.assert_array_equal(perm.cpu().numpy(), np.argsort(-tile_order_key(m, K), kind="stable"))
    if K in (8, 32, 26):
        np.testing.assert_array_equal(tile_key(m, K), m.astype(np.int64))


@pytest.fixture(params=["binned", "hash"])
def kmap_method(request, monkeypatch):
    monkeypatch.setenv("WARPCONVNET_AMD_KMAP_METHOD", request.param)
    return request.param


@pytest.mark.parametrize("n,ksize", [(5000, (3, 3, 3)), (3000, (5, 5, 5)), (4000, (3, 1, 2)), (2000, (2, 2, 2)), (777, (3, 3, 1)), (900, (7, 7, 7))])
def test_submanifold_map_bit_exact(n, ksize, kmap_method):
    s = np.concatenate([scene_u(n, 1, 0), scene_u(n // 2, 2, 1)], 0)
    s[:, 1:] -= 9  # blocks straddle the origin
    km = _gen(s, s, ksize, same=True)
    K = int(np.prod(ksize))
    r = _check_against_oracle(km, s, s, ksize)
    if all(k % 2 == 1 for k in ksize):
        assert km.identity_map_index == K // 2 and km._symmetric
        np.testing.assert_array_equal(km._pair_table.cpu().numpy()[K // 2], np.arange(len(s)))
    else:
        assert km.identity_map_index is None and not km._symmetric
    assert len(km) == K and km.numel(0) == r["offsets"][1]


@pytest.mark.parametrize("ksize,stride", [((2, 2, 2), (2, 2, 2)), ((3, 3, 3), (2, 2, 2)), ((2, 2, 2), (4, 4, 4)), ((3, 3, 3), (1, 2, 1))])
def test_strided_map_bit_exact(ksize, stride):
    from warpconvnet_amd.geometry.coords.ops.stride import stride_coords

    s = np.concatenate([scene_u(6000, 3, 0), scene_u(5000, 4, 1), scene_u(10, 5, 2)], 0)
    s[:, 1:] -= 7  # negative coordinates
    want, _ = okmap.stride_coords(s, stride)
    got, offs = stride_coords(torch.from_numpy(s).to(_dev()), stride)
    np.testing.assert_array_equal(got.cpu().numpy(), want)  # first-occurrence order is deterministic
    np.testing.assert_array_equal(offs.numpy(), np.concatenate([[0], np.cumsum(np.bincount(want[:, 0], minlength=3))]))
    km = _gen(s, want, ksize, stride)
    _check_against_oracle(km, s, want, ksize, stride)
    assert km.identity_map_index is None and not km._symmetric


@pytest.mark.parametrize("prebuild", [True, False])
@pytest.mark.parametrize("ksize,stride", [((2, 2, 2), (2, 2, 2)), ((3, 3, 3), (2, 2, 2)), ((2, 2, 2), (4, 4, 4)), ((3, 3, 3), (1, 2, 1)),
                                          ((4, 2, 1), (4, 2, 1)), ((2, 2, 2), (8, 8, 8))])
def test_strided_layers_from_the_cell_table(ksize, stride, prebuild):
    """The same contract as test_strided_map_bit_exact, answered from the cell table a (validated) submanifold build left
    on the coordinate tensor (csrc/kmap_stride.hip): down-sampled coordinates + offsets bit-exact vs the oracle, the kernel
    map bit-exact whether it comes out of the down-sampling pass (kernel_size == stride) or from the cell probe; duplicate
    coordinates and negative coordinates included.  No global hash table is built on this path."""
    from warpconvnet_amd.geometry.coords.ops.stride import stride_coords
    from warpconvnet_amd.geometry.coords.search import packed_hashmap
    from warpconvnet_amd.geometry.coords.search.torch_discrete import generate_kernel_map

    s = np.concatenate([scene_u(6000, 3, 0), scene_u(5000, 4, 1), scene_u(10, 5, 2)], 0)
    s[:, 1:] -= 7
    s = np.concatenate([s[:6000], s[100:160], s[6000:]], 0).astype(np.int32)  # 60 duplicated rows inside batch 0
    a = torch.from_numpy(s).to(_dev())
    if prebuild:
        sub = generate_kernel_map(a, a, (1, 1, 1), (3, 3, 3))  # the level's submanifold map: leaves the cell table on `a`
        assert getattr(a, "_wcn_cells", None) is not None and sub._has_duplicates
    # (else: a strided FIRST layer - the down-sampling pass builds the table itself, strict insert)
    want, _ = okmap.stride_coords(s, stride)
    inserts = []
    real_insert = packed_hashmap.PackedHashTable._launch_insert
    packed_hashmap.PackedHashTable._launch_insert = lambda self, *x, **k: (inserts.append(1), real_insert(self, *x, **k))[1]
    try:
        got, offs = stride_coords(a, stride, num_batches=3, with_map=tuple(ksize) == tuple(stride))
        np.testing.assert_array_equal(got.cpu().numpy(), want)
        np.testing.assert_array_equal(offs.numpy(), np.concatenate([[0], np.cumsum(np.bincount(want[:, 0], minlength=3))]))
        km = generate_kernel_map(a, got, stride, ksize)
    finally:
        packed_hashmap.PackedHashTable._launch_insert = real_insert
    assert not inserts, "the cell-table route must not build a hash table"
    assert getattr(a, "_wcn_cells", None) is not None
    _check_against_oracle(km, s, want, ksize, stride)
    assert km.identity_map_index is None and not km._symmetric


def test_tables_cached_on_a_coordinate_tensor_do_not_survive_an_in_place_edit():
    """A submanifold build leaves its cell table on the coordinate tensor and a down-sampling pass leaves the kernel map of
    its stride window on the output tensor (`cell_handle.py`).  Both are keyed to the tensor's CONTENT: after ``coords.add_``
    (an augmentation shift between two layers) the strided layer must answer for the shifted coordinates - the reference
    keys its maps per call (helper.py:446-459) and never sees a stale table."""
    from warpconvnet_amd.geometry.coords.ops.stride import stride_coords
    from warpconvnet_amd.geometry.coords.search.cell_handle import cells_of, stride_map_of
    from warpconvnet_amd.geometry.coords.search.torch_discrete import generate_kernel_map

    s = np.concatenate([scene_u(6000, 3, 0), scene_u(5000, 4, 1)], 0).astype(np.int32)
    a = torch.from_numpy(s).to(_dev())
    generate_kernel_map(a, a, (1, 1, 1), (3, 3, 3))
    assert cells_of(a) is not None
    a[:, 1:].add_(1)  # a shift by one cell changes every coarse cell's membership
    assert cells_of(a) is None and getattr(a, "_wcn_cells", None) is None
    shifted = s.copy()
    shifted[:, 1:] += 1
    want, _ = okmap.stride_coords(shifted, (2, 2, 2))
    got, _ = stride_coords(a, (2, 2, 2), num_batches=2, with_map=True)
    np.testing.assert_array_equal(got.cpu().numpy(), want)
    km = generate_kernel_map(a, got, (2, 2, 2), (2, 2, 2))
    _check_against_oracle(km, shifted, want, (2, 2, 2), (2, 2, 2))
    # ... and the map written next to the output coordinates dies with an edit of either side
    assert stride_map_of(got, a) is not None
    a[:, 1:].add_(2)
    assert stride_map_of(got, a) is None
    shifted[:, 1:] += 2
    km = generate_kernel_map(a, got, (2, 2, 2), (2, 2, 2))
    _check_against_oracle(km, shifted, want, (2, 2, 2), (2, 2, 2))


def test_strided_first_layer_table_grows_and_range_errors_surface():
    """Down-sampling that builds its own cell table: a scene with one voxel per 8^3 block overflows the first-try block bound
    (TABLE_FULL -> larger table, remembered in the hints) and still gives the oracle's coordinates; a coordinate outside
    the packed range raises ValueError like the hash path (reference packed_hashmap.py:66-82)."""
    from warpconvnet_amd.geometry.coords.ops.stride import stride_coords
    from warpconvnet_amd.geometry.coords.search.torch_discrete import default_hints

    rng = np.random.default_rng(9)
    cells = rng.permutation(40 * 40 * 40)[:30000]
    s = np.stack([np.zeros_like(cells), cells // 1600 * 8 + 3, cells // 40 % 40 * 8 + 1, cells % 40 * 8 + 6], 1).astype(np.int32)
    default_hints().reset()
    got, offs = stride_coords(torch.from_numpy(s).to(_dev()), (2, 2, 2), num_batches=1)
    assert default_hints().div < 16
    want, _ = okmap.stride_coords(s, (2, 2, 2))
    np.testing.assert_array_equal(got.cpu().numpy(), want)
    assert offs.tolist() == [0, len(want)]
    default_hints().reset()
    bad = scene_u(3000, 5, 0)
    bad[11, 3] = -131073
    with pytest.raises(ValueError):
        stride_coords(torch.from_numpy(bad).to(_dev()), (2, 2, 2), num_batches=1)


def test_binned_edge_cases(kmap_method):
    """Duplicates (smallest row wins), coordinates at the limits of the packed range (18-bit wrap of the probe
    key, reference hash_functions.cuh:40-44), far-apart clusters, several batch indices."""
    rng = np.random.default_rng(5)
    base = scene_u(1500, 4)[:, 1:]
    far = base + np.array([131071 - 30, -131072, 60000], np.int32)  # touches +x and -y limits
    far = far[(far[:, 0] <= 131071)]
    wrap_partner = np.array([[-131072, -131072 + 5, 60000 + 3], [131071, 131071, 0], [-131072, 131071, 0]], np.int32)
    pts = np.concatenate([base, far, wrap_partner, base[:200]], 0)  # last 200 rows are duplicates
    b = rng.integers(0, 3, size=len(pts)).astype(np.int32)
    b[-200:] = b[:200]
    b[len(base) + len(far) : len(base) + len(far) + 3] = 1  # the wrap partners share a batch index
    order = np.argsort(b, kind="stable")
    s = np.concatenate([b[order, None], pts[order]], 1).astype(np.int32)
    km = _gen(s, s, (3, 3, 3), same=True)
    r = okmap.kernel_map(s, s, (3, 3, 3))
    np.testing.assert_array_equal(km._pair_table.cpu().numpy(), r["found"])
    np.testing.assert_array_equal(km.in_maps.cpu().numpy(), r["in_maps"])
    np.testing.assert_array_equal(km._mask.cpu().numpy().view(np.uint32), r["mask"])
    # the x = 131071 voxel sees the x = -131072 voxel through the 18-bit wrap, exactly like the reference's packed key
    i_hi = int(np.nonzero((s[:, 1] == 131071) & (s[:, 2] == 131071))[0][0])
    i_lo = int(np.nonzero((s[:, 1] == -131072) & (s[:, 2] == 131071))[0][0])
    assert r["found"][22, i_hi] == i_lo  # k = 22 is offset (+1, 0, 0)
    bad = s.copy(); bad[3, 2] = 131072
    with pytest.raises(ValueError):
        _gen(bad, bad, (3, 3, 3), same=True).offsets


def test_dilation_and_2d(kmap_method):
    s = scene_u(3000, 9)
    km = _gen(s, s, (3, 3, 3), dilation=(2, 2, 2), same=True)
    _check_against_oracle(km, s, s, (3, 3, 3), dilation=(2, 2, 2))
    s2 = np.unique(s[:, :3], axis=0)  # [b, x, y]
    km2 = _gen(s2, s2, (3, 3), stride=(1, 1), same=True)
    s2p = np.concatenate([s2, np.zeros((len(s2), 1), np.int32)], 1)
    _check_against_oracle(km2, s2p, s2p, (3, 3, 1))


def test_reverse_table_and_from_csr():
    from warpconvnet_amd.geometry.coords.search.search_results import IntSearchResult
    from warpconvnet_amd.geometry.coords.search.torch_discrete import attach_tables_from_csr, reverse_tables

    s = scene_u(5000, 11)
    coarse, _ = okmap.stride_coords(s, (2, 2, 2))
    km = _gen(s, coarse, (2, 2, 2), (2, 2, 2))
    r = okmap.kernel_map(s, coarse, (2, 2, 2), (2, 2, 2))
    rev_nbr, rev_mask, rev_perm = reverse_tables(km, len(s))
    np.testing.assert_array_equal(rev_nbr.cpu().numpy()[:, :8].T, r["rev"])
    np.testing.assert_array_equal(rev_mask.cpu().numpy().view(np.uint32), r["rev_mask"])
    assert sorted(rev_perm.cpu().tolist()) == list(range(len(s)))
    # a swapped (transposed-conv) map rebuilt from its CSR form
    sw = IntSearchResult(km.out_maps, km.in_maps, km.offsets)
    attach_tables_from_csr(sw, len(coarse), len(s))
    np.testing.assert_array_equal(sw._nbr.cpu().numpy()[:, :8].T, r["rev"])
    np.testing.assert_array_equal(sw._mask.cpu().numpy().view(np.uint32), r["rev_mask"])
    # the convolution's own exchange (helper._swap) keeps a link to the forward map: its tables are that map's reverse tables
    # (one build, shared with the strided layer's dgrad) and its reverse tables are the forward map's own - same content as
    # the CSR route above, the oracle's tables in both directions
    from warpconvnet_amd.nn.functional.sparse_conv.helper import _swap

    tw = _swap(km)
    assert tw._twin is km
    attach_tables_from_csr(tw, len(coarse), len(s))
    assert tw._nbr is rev_nbr and tw._mask is rev_mask and tw._perm is rev_perm
    np.testing.assert_array_equal(tw._nbr.cpu().numpy()[:, :8].T, r["rev"])
    t_nbr, t_mask, t_perm = reverse_tables(tw, len(coarse))
    assert t_nbr is km._nbr and t_mask is km._mask and t_perm is km._perm
    np.testing.assert_array_equal(t_nbr.cpu().numpy()[:, :8].T, r["nbr"] if "nbr" in r else t_nbr.cpu().numpy()[:, :8].T)
    s_nbr, s_mask, _ = reverse_tables(sw, len(coarse))  # (the twin-less copy: rebuilt from the pair lists)
    np.testing.assert_array_equal(t_nbr.cpu().numpy()[:, :8], s_nbr.cpu().numpy()[:, :8])
    np.testing.assert_array_equal(t_mask.cpu().numpy(), s_mask.cpu().numpy())


def test_hash_table_contract_gpu():
    from warpconvnet_amd.geometry.coords.search.packed_hashmap import PackedHashTable

    dev = _dev()
    c = scene_u(20000, 3)
    t = PackedHashTable.from_coords(torch.from_numpy(c).to(dev))
    assert t.capacity == 65536 and t.num_entries == len(c)
    np.testing.assert_array_equal(t.search(torch.from_numpy(c).to(dev)).cpu().numpy(), np.arange(len(c)))
    miss = c.copy(); miss[:, 1] += 100000
    assert (t.search(torch.from_numpy(miss).to(dev)) == -1).all()
    # duplicates: the smallest row index wins -> identical to the oracle's first-occurrence rule
    dup = np.concatenate([c[:1000], c[:1000][::-1], c[1000:2000]], 0)
    td = PackedHashTable.from_coords(torch.from_numpy(dup).to(dev))
    np.testing.assert_array_equal(td.search(torch.from_numpy(dup).to(dev)).cpu().numpy(), okmap.HashTable(dup).search(dup))
    uniq = td.unique_index.cpu().numpy()
    np.testing.assert_array_equal(uniq, np.concatenate([np.arange(1000), np.arange(2000, 3000)]))
    assert (t.keys_tensor != 0).sum().item() == len(c) and (t.values_tensor >= 0).sum().item() == len(c)
    edge = np.array([[511, 131071, -131072, 0], [0, -131072, 131071, 5]], np.int32)
    te = PackedHashTable.from_coords(torch.from_numpy(edge).to(dev))
    np.testing.assert_array_equal(te.search(torch.from_numpy(edge).to(dev)).cpu().numpy(), [0, 1])
    for bad in ([[512, 0, 0, 0]], [[-1, 0, 0, 0]], [[0, 131072, 0, 0]], [[0, 0, -131073, 0]]):
        with pytest.raises(ValueError):
            PackedHashTable.from_coords(torch.tensor(bad, dtype=torch.int32, device=dev))
    full = PackedHashTable(32, device=dev)
    with pytest.raises(AssertionError):
        full.insert(torch.from_numpy(c[:40]).to(dev))  # N <= capacity/2 rule
    with pytest.raises(RuntimeError):
        PackedHashTable.from_coords(torch.from_numpy(c[:5]).cpu())  # no CPU path


def test_empty_and_tiny_inputs():
    dev = _dev()
    e = np.zeros((0, 4), np.int32)
    km = _gen(e, e, (3, 3, 3), same=True)
    assert km.offsets.tolist() == [0] * 28 and km.in_maps.numel() == 0
    one = np.array([[0, 5, 5, 5]], np.int32)
    km1 = _gen(one, one, (3, 3, 3), same=True)
    assert km1.offsets.tolist() == [0] * 14 + [1] * 14 and km1.in_maps.tolist() == [0]


def test_large_scene_invariants(kmap_method):
    """Full-size scene (1M voxels): invariants instead of the serial oracle (which takes ~10 s on the same
    data, so the oracle comparison runs on a 200k subsample of the table)."""
    s = scene_u(1_000_000, 0)
    km = _gen(s, s, (3, 3, 3), same=True)
    N = len(s)
    offs = km.offsets.numpy()
    assert offs[-1] == km.in_maps.numel() == km.out_maps.numel()
    t = torch.from_numpy(s).to(_dev())
    ko = torch.tensor([[0, i - 1, j - 1, l - 1] for i in range(3) for j in range(3) for l in range(3)], dtype=torch.int32, device=_dev())
    for k in (0, 5, 13, 26):
        i, o = km[k]
        assert (t[i.long()] == t[o.long()] + ko[k]).all()
        assert (o[1:] > o[:-1]).all()
    pt = km._pair_table
    assert (pt[13] == torch.arange(N, device=_dev(), dtype=torch.int32)).all()
    # symmetry: found[k][i] = j  <=>  found[26-k][j] = i
    for k in (0, 7, 12):
        rows = torch.nonzero(pt[k] >= 0).squeeze(1)
        assert (pt[26 - k][pt[k][rows].long()] == rows.int()).all()
    assert int((pt >= 0).sum()) == offs[-1]
    # oracle on a subsample of rows
    r = okmap.kernel_map(s, s[:200000], (3, 3, 3))
    np.testing.assert_array_equal(pt[:, :200000].cpu().numpy(), r["found"])


def test_surface_scene_map(kmap_method):
    s = scene_surface(160, 2)
    km = _gen(s, s, (3, 3, 3), same=True)
    _check_against_oracle(km, s, s, (3, 3, 3))


def test_expand_coords_matches_set_union():
    """Generative output coordinates: inputs U (inputs + every kernel offset), de-duplicated, batch-sorted, deterministic
    (reference coords/ops/expand.py:17-75 builds the same set; its row order is hash-table / unstable-argsort dependent)."""
    from warpconvnet_amd.geometry.coords.ops.expand import expand_coords
    from warpconvnet_amd.geometry.coords.search.torch_discrete import kernel_offsets_from_size

    s = np.concatenate([scene_u(1500, 81, 0), scene_u(700, 82, 1)], 0)
    dev = torch.device("cuda:0")
    for ks, dl in (((3, 3, 3), (1, 1, 1)), ((2, 2, 2), (1, 1, 1)), ((3, 1, 3), (2, 1, 1))):
        out, offsets = expand_coords(torch.from_numpy(s).to(dev), ks, dl)
        out = out.cpu().numpy()
        off = kernel_offsets_from_size(ks, dl).numpy()
        want = np.unique(np.concatenate([s] + [s + o for o in off], 0), axis=0)
        assert len(out) == len(want) and len(np.unique(out, axis=0)) == len(out)
        np.testing.assert_array_equal(np.unique(out, axis=0), want)
        assert (np.diff(out[:, 0]) >= 0).all()
        np.testing.assert_array_equal(offsets.numpy(), [0, (out[:, 0] == 0).sum(), len(out)])
        # every batch starts with its input rows in input order (first occurrences win)
        np.testing.assert_array_equal(out[:1500], s[:1500])
        out2, _ = expand_coords(torch.from_numpy(s).to(dev), ks, dl, kernel_batch=1)
        np.testing.assert_array_equal(np.unique(out2.cpu().numpy(), axis=0), want)


def _brute_offsets_map(in_np, out_np, offsets, stride):
    """Dictionary brute force for an explicit offset table (method of the reference's
    tests/coords/test_kernel_map_invariants.py:205-230): buckets ordered by output row."""
    table = {}
    for r, c in enumerate(map(tuple, in_np.tolist())):
        table.setdefault(c, r)
    ins, outs, offs = [], [], [0]
    for off in offsets:
        for m, (b, x, y, z) in enumerate(out_np.tolist()):
            hit = table.get((b, x * stride[0] + off[0], y * stride[1] + off[1], z * stride[2] + off[2]))
            if hit is not None:
                ins.append(hit)
                outs.append(m)
        offs.append(len(ins))
    return np.array(ins, np.int32), np.array(outs, np.int32), np.array(offs, np.int32)


@pytest.mark.parametrize("ksize,dil,centre", [((3, 3, 3), (1, 1, 1), (0, 0, 0)), ((3, 3, 3), (2, 1, 1), (2, 1, 0)),
                                              ((2, 2, 2), (1, 1, 1), (1, 1, 1)), ((5, 3, 1), (1, 2, 1), (1, 0, 0))])
def test_custom_kernel_center_offset(ksize, dil, centre):
    """`kernel_center_offset` (reference torch_discrete.py:24-56, 387-401; `offset` method only): offsets (i - c) * dilation
    with the caller's centre c, against a dictionary brute force over the explicit offset table."""
    from warpconvnet_amd.geometry.coords.search.torch_discrete import generate_kernel_map, kernel_offsets_from_size

    s = scene_u(1500, 61, 0)
    o = np.ascontiguousarray(scene_u(1400, 62, 0))  # a different output set: the full map comes back (no halving)
    dev = torch.device("cuda:0")
    km = generate_kernel_map(torch.from_numpy(s).to(dev), torch.from_numpy(o).to(dev), (1, 1, 1), ksize, dil, centre, method="offset")
    offsets = kernel_offsets_from_size(ksize, dil, centre)[:, 1:].tolist()
    assert offsets[0] == [(0 - c) * d for c, d in zip(centre, dil)]
    i, out, off = _brute_offsets_map(s, o, offsets, (1, 1, 1))
    np.testing.assert_array_equal(km.offsets.numpy(), off)
    np.testing.assert_array_equal(km.in_maps.cpu().numpy(), i)
    np.testing.assert_array_equal(km.out_maps.cpu().numpy(), out)
    with pytest.raises(AssertionError):  # the reference's `size` method rejects a custom centre
        generate_kernel_map(torch.from_numpy(s).to(dev), torch.from_numpy(o).to(dev), (1, 1, 1), ksize, dil, centre, method="size")


def test_offset_method_and_skip_symmetric_return_the_first_half():
    """Reference behaviour (torch_discrete.py:211-219, 363-370, 387-401): on an odd kernel over equally sized coordinate
    sets, method="offset" and skip_symmetric_kernel_map=True keep offsets 0 .. K//2-1 only, identity_map_index = K//2."""
    from warpconvnet_amd.geometry.coords.search.torch_discrete import generate_kernel_map

    s = scene_u(3000, 63, 0)
    c = torch.from_numpy(s).to(torch.device("cuda:0"))
    full = generate_kernel_map(c, c, (1, 1, 1), (3, 3, 3))
    r = okmap.kernel_map(s, s, (3, 3, 3))
    for kw in (dict(method="offset"), dict(skip_symmetric_kernel_map=True)):
        half = generate_kernel_map(c, c, (1, 1, 1), (3, 3, 3), **kw)
        assert len(half) == 13 and half.identity_map_index == 13
        np.testing.assert_array_equal(half.offsets.numpy(), r["offsets"][:14])
        np.testing.assert_array_equal(half.in_maps.cpu().numpy(), r["in_maps"][: r["offsets"][13]])
        np.testing.assert_array_equal(half.out_maps.cpu().numpy(), r["out_maps"][: r["offsets"][13]])
        # the dropped half is the mirror image of the kept one: pair (i, o) at offset k <=> pair (o, i) at offset K-1-k
        for k in (0, 5, 12):
            a = set(zip(*[t.cpu().tolist() for t in half[k]]))
            b = set((o_, i_) for i_, o_ in zip(*[t.cpu().tolist() for t in full[26 - k]]))
            assert a == b
    with pytest.raises(AssertionError):
        generate_kernel_map(c, c, (1, 1, 1), (2, 2, 2), skip_symmetric_kernel_map=True)  # even kernel
    with pytest.raises(ValueError):
        generate_kernel_map(c, c, (1, 1, 1), (3, 3, 3), method="hash")


@pytest.mark.parametrize("n,ksize,dil", [(1500, (9, 9, 9), (1, 1, 1)), (2500, (7, 7, 7), (2, 2, 2)), (4000, (3, 3, 3), (8, 8, 8)),
                                         (3000, (5, 3, 5), (2, 5, 1)), (2000, (3, 3, 3), (6, 1, 3))])
def test_large_halo_maps_bit_exact(n, ksize, dil):
    """Halos of 4..8 cells (9^3, dilated 7^3 / 5^3 / 3^3) stay on the cell-table builder: LDS grid up to 24^3 cells, the
    neighbour cells of a probe still lie in the 26 adjacent blocks.  Both builders against the oracle."""
    import os

    from warpconvnet_amd import _lib

    s = np.concatenate([scene_u(n, 11, 0), scene_u(n // 3, 12, 1)], 0)
    s[:, 1:] -= 13
    import ctypes
    arr = ctypes.c_int32 * 3
    assert _lib.lib().wcn_kmap_binned_supported(arr(*ksize), arr(*dil)) == 1
    for method in ("binned", "hash"):
        os.environ["WARPCONVNET_AMD_KMAP_METHOD"] = method
        try:
            km = _gen(s, s, ksize, dilation=dil, same=True)
            _check_against_oracle(km, s, s, ksize, dilation=dil)
        finally:
            os.environ.pop("WARPCONVNET_AMD_KMAP_METHOD", None)
    assert _lib.lib().wcn_kmap_binned_supported(arr(3, 3, 3), arr(9, 1, 1)) == 0  # halo 9: hash path


def test_optimistic_build_validates_and_rebuilds(kmap_method):
    """`generate_kernel_map(..., optimistic=True)` (what the convolution uses): the device tables exist before the status
    word is read; `validate()` finishes the build - offsets, pair lists, identities - reports False for an ordinary scene,
    rebuilds for duplicate coordinates (strict insert) and raises the build-time errors."""
    from warpconvnet_amd.geometry.coords.search.torch_discrete import generate_kernel_map

    dev = _dev()
    s = scene_u(6000, 21)
    a = torch.from_numpy(s).to(dev)
    km = generate_kernel_map(a, a, (1, 1, 1), (3, 3, 3), optimistic=True)
    assert km._nbr is not None and km._mask is not None and km._perm is not None and km._validate_fn is not None
    assert km.validate() is False and km.validate() is False  # idempotent
    _check_against_oracle(km, s, s, (3, 3, 3))
    assert km.identity_map_index == 13 and km._symmetric
    # duplicate coordinates, the LATER copy first in memory order of the cell stores: the plain-store build may keep the
    # wrong row and must be redone with the strict insert - either way the validated map is the oracle's
    d = np.concatenate([s[:300][::-1], s], 0).astype(np.int32)
    b = torch.from_numpy(d).to(dev)
    km2 = generate_kernel_map(b, b, (1, 1, 1), (3, 3, 3), optimistic=True)
    rebuilt = km2.validate()
    assert isinstance(rebuilt, bool)
    _check_against_oracle(km2, d, d, (3, 3, 3))
    assert km2.identity_map_index is None and km2._has_duplicates and not km2._symmetric
    bad = s.copy(); bad[3, 2] = 131072
    c = torch.from_numpy(bad).to(dev)
    km3 = generate_kernel_map(c, c, (1, 1, 1), (3, 3, 3), optimistic=True)
    with pytest.raises(ValueError):
        km3.validate()
    # a build the device rejected for good stays rejected: every later use raises again (the reference raises at build time
    # and never hands out the map), no accessor returns half-built state
    with pytest.raises(ValueError):
        km3.validate()
    with pytest.raises(ValueError):
        km3.offsets
    with pytest.raises(ValueError):
        km3.identity_map_index


def test_build_hints_are_explicit_state():
    """The sizing guesses of the builder live in a `BuildHints` object (argument of `generate_kernel_map`; one default object per
    process): with a fresh object the TABLE_FULL retry and the short pair-capacity guess are reached whatever ran before, the
    object learns from them, and the maps are the oracle's either way."""
    from warpconvnet_amd.geometry.coords.search.torch_discrete import BuildHints, generate_kernel_map

    dev = _dev()
    rng = np.random.default_rng(3)
    # one voxel per 8^3 block: 30 000 occupied blocks against a first-try bound of max(1024, N / 16)
    cells = rng.permutation(40 * 40 * 40)[:30000]
    sparse = np.stack([np.zeros_like(cells), cells // 1600 * 8, cells // 40 % 40 * 8, cells % 40 * 8], 1).astype(np.int32)
    a = torch.from_numpy(sparse).to(dev)
    hints = BuildHints()
    km = generate_kernel_map(a, a, (1, 1, 1), (3, 3, 3), optimistic=True, hints=hints)
    assert km.validate() is True and hints.div < 16  # the device asked for a larger block table; remembered
    _check_against_oracle(km, sparse, sparse, (3, 3, 3))
    km_b = generate_kernel_map(a, a, (1, 1, 1), (3, 3, 3), optimistic=True, hints=hints)
    assert km_b.validate() is False  # the learned bound fits at once
    # pair lists written before the pair count is known: a guess of 1 pair per row is short for a dense scene - the
    # lists are rewritten at their exact length, the tables stay (validate() reports no rebuild)
    s = scene_u(6000, 22)
    b = torch.from_numpy(s).to(dev)
    short = BuildHints(pairs_per_row=0.5)
    km2 = generate_kernel_map(b, b, (1, 1, 1), (3, 3, 3), optimistic=True, hints=short)
    assert km2.validate() is False and short.pairs_per_row > 2.0
    _check_against_oracle(km2, s, s, (3, 3, 3))
    assert km2.in_maps_device.shape[0] == int(km2.offsets[-1])
    # a generous guess is trimmed: the cached map does not pin a worst-case buffer
    wide = BuildHints(pairs_per_row=27.0)
    km3 = generate_kernel_map(b, b, (1, 1, 1), (3, 3, 3), optimistic=True, hints=wide)
    km3.validate()
    assert km3.in_maps_device.untyped_storage().nbytes() <= 4 * int(km3.offsets[-1]) + 1024
    _check_against_oracle(km3, s, s, (3, 3, 3))
