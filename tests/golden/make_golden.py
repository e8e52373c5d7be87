"""Generate the golden fixtures in this directory by IMPORTING THE REFERENCE (authoring container only).

    python tests/golden/make_golden.py          # needs /root/reference; writes tests/golden/*.npz

The reference's native extension cannot be built here (CUDA + CUTLASS), so three stub modules are injected
before the import: ``jaxtyping`` (annotation dummies), ``torch_scatter`` and an empty
``warpconvnet._C`` package.  Only reference PYTHON runs: ``kernel_offsets_from_size``, the explicit
gather-matmul-scatter logic, ``IntSearchResult``, ``IntSearchCacheKey`` and ``SparseConv3d.__init__``.
Kernel maps fed to the explicit logic come from ``oracle.brute`` (dictionary enumeration), because the
reference's own map builder needs CUDA.  Only data (inputs / expected outputs) is written - no reference source.
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)


def import_reference():
    os.environ.setdefault("WARPCONVNET_BENCHMARK_CACHE_DIR", "/tmp/wcn_ref_cache")
    os.makedirs(os.environ["WARPCONVNET_BENCHMARK_CACHE_DIR"], exist_ok=True)

    class _Sub:
        def __getitem__(self, k): return self
        def __call__(self, *a, **k): return self
        def __or__(self, o): return self
        def __ror__(self, o): return self

    jt = types.ModuleType("jaxtyping")
    for n in ["Float", "Int", "Bool", "Shaped", "Num", "Array", "Integer", "UInt8", "Int32", "Int64", "Float32"]:
        setattr(jt, n, _Sub())
    sys.modules["jaxtyping"] = jt
    ts = types.ModuleType("torch_scatter")
    ts.segment_csr = lambda *a, **k: None
    sys.modules["torch_scatter"] = ts
    C = types.ModuleType("warpconvnet._C")
    for s in ["gemm", "fma", "utils", "sampling", "coords", "cuhash", "mask_gemm", "fused_rope"]:
        m = types.ModuleType("warpconvnet._C." + s)
        setattr(C, s, m)
        sys.modules["warpconvnet._C." + s] = m
    sys.modules["warpconvnet._C"] = C
    sys.path.insert(0, "/root/reference")
    import warpconvnet  # noqa: F401

    warpconvnet._C = C


def scene_u(n, seed, batch=0):
    """Reference-style uniform scene (scripts/populate_benchmark_cache.py:283-309): extent 2*ceil(n^(1/3))."""
    rng = np.random.default_rng(seed)
    extent = 2 * int(np.ceil(n ** (1.0 / 3.0)))
    c = rng.integers(0, extent, size=(int(1.3 * n), 3))
    _, first = np.unique(c, axis=0, return_index=True)
    c = c[np.sort(first)][:n].astype(np.int32)
    return np.concatenate([np.full((len(c), 1), batch, np.int32), c], axis=1)


# known-answer patterns of the reference's tests/nn/test_kernel_deterministic.py:129-183
def make_weight(K, cin, cout, pattern, dtype):
    if pattern == "ones":
        return torch.ones(K, cin, cout, dtype=dtype)
    if pattern in ("triu", "tril"):
        m = torch.ones(cin, cout, dtype=dtype)
        m = m.triu() if pattern == "triu" else m.tril()
        return m.unsqueeze(0).expand(K, -1, -1).contiguous()
    d = min(cin, cout)
    eye = torch.zeros(cin, cout, dtype=dtype)
    eye[:d, :d] = torch.eye(d, dtype=dtype)
    if pattern == "eye":
        return eye.unsqueeze(0).expand(K, -1, -1).contiguous()
    if pattern == "center_eye":
        w = torch.zeros(K, cin, cout, dtype=dtype)
        w[K // 2] = eye
        return w
    raise ValueError(pattern)


def make_feats(n, cin, pattern, dtype):
    col = torch.arange(cin, dtype=dtype) / max(cin, 1)
    if pattern == "ones":
        return torch.ones(n, cin, dtype=dtype)
    if pattern == "range":
        return col.unsqueeze(0).expand(n, -1).contiguous()
    if pattern == "row_index":
        row = (torch.arange(n, dtype=dtype) % 16) / 16.0
        return (row.unsqueeze(1) + col.unsqueeze(0)) / 2.0
    raise ValueError(pattern)


def make_grad_out(n, cout, dtype):
    col = torch.arange(cout, dtype=dtype) / max(cout, 1)
    row = (torch.arange(n, dtype=dtype) % 8) / 8.0
    return (row.unsqueeze(1) + col.unsqueeze(0)) / 2.0


def main_depthwise():
    """(g) depthwise sparse convolution: explicit forward/backward and seeded module init (reference:
    nn/functional/sparse_conv_depth.py:227-306, nn/modules/sparse_conv_depth.py:43-183)."""
    import_reference()
    from warpconvnet.geometry.coords.search.search_results import IntSearchResult
    from warpconvnet.nn.functional.sparse_conv_depth import (
        _explicit_depthwise_backward_logic,
        _explicit_depthwise_forward_logic,
    )
    from warpconvnet.nn.modules.sparse_conv_depth import SparseDepthwiseConv3d

    from oracle import brute, kmap

    def run_case(name, bc_in, bc_out, ksize, stride, C, dtype, use_identity, seed):
        found, offsets, in_maps, out_maps = brute.kernel_map(bc_in, bc_out, ksize, stride)
        K = len(offsets) - 1
        iden = K // 2 if use_identity else None
        g = torch.Generator().manual_seed(seed)
        X = torch.randn(len(bc_in), C, generator=g, dtype=dtype)
        W = torch.randn(K, C, generator=g, dtype=dtype) * 0.2
        dY = torch.randn(len(bc_out), C, generator=g, dtype=dtype)
        km = IntSearchResult(torch.from_numpy(in_maps), torch.from_numpy(out_maps), torch.from_numpy(offsets), iden)
        Y = _explicit_depthwise_forward_logic(X, W, km, len(bc_out))
        dX, dW = _explicit_depthwise_backward_logic(dY, X, W, km)
        np.savez_compressed(
            os.path.join(HERE, f"depthwise_{name}.npz"),
            in_coords=bc_in, out_coords=bc_out, ksize=np.asarray(ksize, np.int32), stride=np.asarray(stride, np.int32),
            in_maps=in_maps, out_maps=out_maps, offsets=offsets, identity=np.asarray(-1 if iden is None else iden),
            X=X.numpy(), W=W.numpy(), dY=dY.numpy(), Y=Y.numpy(), dX=dX.numpy(), dW=dW.numpy(),
        )
        print("depthwise", name, "pairs", offsets[-1])

    s = scene_u(600, 1)
    run_case("u600_c64_f32", s, s, (3, 3, 3), (1, 1, 1), 64, torch.float32, True, 21)
    run_case("u600_c64_f64", s, s, (3, 3, 3), (1, 1, 1), 64, torch.float64, True, 21)
    s2 = np.concatenate([scene_u(300, 2, 0), scene_u(400, 3, 1)], 0)
    run_case("b2_c13_f32_noiden", s2, s2, (3, 3, 3), (1, 1, 1), 13, torch.float32, False, 22)
    coarse, _ = kmap.stride_coords(s, (2, 2, 2))
    run_case("stride2_k2_c32_f32", s, coarse, (2, 2, 2), (2, 2, 2), 32, torch.float32, False, 23)

    init = {}
    for name, kwargs in [("c64_k3", dict(channels=64, kernel_size=3)),
                         ("c32_k2_s2_tr", dict(channels=32, kernel_size=2, stride=2, transposed=True))]:
        torch.manual_seed(0)
        m = SparseDepthwiseConv3d(**kwargs)
        init[name + "_weight"] = m.weight.detach().numpy()
        init[name + "_bias"] = m.bias.detach().numpy()
    np.savez_compressed(os.path.join(HERE, "module_init_depthwise.npz"), **init)
    print("done (depthwise)")


def main_pointconv():
    """(h) PointConv (knn) forward / backward on CPU with seeded weights (reference nn/modules/point_conv.py:36-282).
    torch_scatter is a third-party dependency absent from this image: segment_csr is stubbed with a plain loop."""
    import_reference()
    import torch_scatter

    def segment_csr(src, indptr, reduce="sum"):
        rows = []
        for i in range(indptr.numel() - 1):
            seg = src[int(indptr[i]) : int(indptr[i + 1])]
            rows.append({"sum": lambda t: t.sum(0), "mean": lambda t: t.mean(0), "max": lambda t: t.max(0).values,
                         "min": lambda t: t.min(0).values}[reduce](seg))
        return torch.stack(rows)

    torch_scatter.segment_csr = segment_csr
    import warpconvnet.ops.reductions as R

    R.segment_csr = segment_csr
    from warpconvnet.geometry.coords.search.search_configs import RealSearchConfig
    from warpconvnet.geometry.types.points import Points
    from warpconvnet.nn.modules.point_conv import PointConv

    out = {}
    for name, kw in [("knn8_relpos_mean_max", dict(use_rel_pos=True, reductions=("mean", "max"))),
                     ("knn8_plain_sum", dict(reductions=("sum",)))]:
        g = torch.Generator().manual_seed(31)
        c0, c1 = torch.rand(180, 3, generator=g), torch.rand(140, 3, generator=g) * 2.0
        f0, f1 = torch.randn(180, 8, generator=g), torch.randn(140, 8, generator=g)
        feats = torch.cat([f0, f1]).requires_grad_(True)
        pc = Points(torch.cat([c0, c1]), feats, offsets=torch.tensor([0, 180, 320]))
        torch.manual_seed(0)
        conv = PointConv(8, 16, RealSearchConfig(mode="knn", knn_k=8), **kw)
        y = conv(pc).feature_tensor
        dY = torch.randn(y.shape, generator=g)
        y.backward(dY)
        out[name + "_coords"] = torch.cat([c0, c1]).numpy()
        out[name + "_feats"] = feats.detach().numpy()
        out[name + "_dY"] = dY.numpy()
        out[name + "_Y"] = y.detach().numpy()
        out[name + "_dX"] = feats.grad.numpy()
        for k, v in conv.state_dict().items():
            out[f"{name}_param_{k}"] = v.numpy()
        for k, p in conv.named_parameters():
            out[f"{name}_grad_{k}"] = p.grad.numpy()
    np.savez_compressed(os.path.join(HERE, "pointconv.npz"), **out)
    print("done (pointconv)")


def main_radius():
    """(i) radius search: the reference's own CPU path (`geometry/coords/search/radius.py:127-225`, chunked cdist) and
    `batched_radius_search`'s bookkeeping restated on its output (the batched entry point itself asserts CUDA)."""
    import_reference()
    from warpconvnet.geometry.coords.search.radius import radius_search

    g = torch.Generator().manual_seed(11)
    pts = torch.rand(700, 3, generator=g) * torch.tensor([4.0, 4.0, 1.0])
    qry = (torch.rand(260, 3, generator=g) * 1.2 - 0.1) * torch.tensor([4.0, 4.0, 1.0])
    out = {"points": pts.numpy(), "queries": qry.numpy()}
    for tag, radius in (("r030", 0.30), ("r075", 0.75)):
        idx, dist, split = radius_search(pts.contiguous(), qry.contiguous(), radius)
        out[f"{tag}_radius"] = np.float32(radius)
        out[f"{tag}_index"] = idx.numpy()
        out[f"{tag}_distance"] = dist.numpy()
        out[f"{tag}_split"] = split.numpy()
    np.savez_compressed(os.path.join(HERE, "radius_search.npz"), **out)
    print("radius_search.npz", {k: v.shape for k, v in out.items() if hasattr(v, "shape")})


def main():
    import_reference()
    from warpconvnet.geometry.coords.search.cache import IntSearchCacheKey
    from warpconvnet.geometry.coords.search.search_results import IntSearchResult
    from warpconvnet.geometry.coords.search.torch_discrete import kernel_offsets_from_size
    from warpconvnet.nn.functional.sparse_conv.detail.explicit import (
        _explicit_gemm_backward_logic,
        _explicit_gemm_forward_logic,
    )
    from warpconvnet.nn.modules.sparse_conv import SparseConv3d

    from oracle import brute, kmap

    # ---- (a) offset tables ---------------------------------------------------------------------
    tables = {}
    for ks in [(3, 3, 3), (2, 2, 2), (5, 5, 5), (3, 3, 1), (1, 3, 5)]:
        for dl in [(1, 1, 1), (2, 2, 2)]:
            name = "k%d%d%d_d%d%d%d" % (*ks, *dl)
            tables[name] = kernel_offsets_from_size(ks, dl).numpy()
    np.savez_compressed(os.path.join(HERE, "offset_tables.npz"), **tables)

    # ---- (b) explicit forward / backward on seeded inputs -----------------------------------------
    def run_case(name, bc_in, bc_out, ksize, stride, cin, cout, dtype, use_identity, seed):
        found, offsets, in_maps, out_maps = brute.kernel_map(bc_in, bc_out, ksize, stride)
        K = len(offsets) - 1
        iden = K // 2 if use_identity else None
        g = torch.Generator().manual_seed(seed)
        X = torch.randn(len(bc_in), cin, generator=g, dtype=dtype)
        W = (torch.randn(K, cin, cout, generator=g, dtype=dtype) * 0.05)
        dY = torch.randn(len(bc_out), cout, generator=g, dtype=dtype)
        km = IntSearchResult(torch.from_numpy(in_maps), torch.from_numpy(out_maps), torch.from_numpy(offsets), iden)
        Y = _explicit_gemm_forward_logic(X, W, km, len(bc_out))
        dX, dW = _explicit_gemm_backward_logic(dY, X, W, km)
        np.savez_compressed(
            os.path.join(HERE, f"explicit_{name}.npz"),
            in_coords=bc_in, out_coords=bc_out, ksize=np.asarray(ksize, np.int32), stride=np.asarray(stride, np.int32),
            in_maps=in_maps, out_maps=out_maps, offsets=offsets, identity=np.asarray(-1 if iden is None else iden),
            X=X.numpy(), W=W.numpy(), dY=dY.numpy(), Y=Y.numpy(), dX=dX.numpy(), dW=dW.numpy(),
        )
        print(name, "pairs", offsets[-1], "N_in", len(bc_in), "N_out", len(bc_out))

    s = scene_u(2048, 1)
    run_case("u2048_16x32_f32", s, s, (3, 3, 3), (1, 1, 1), 16, 32, torch.float32, True, 11)
    run_case("u2048_16x32_f64", s, s, (3, 3, 3), (1, 1, 1), 16, 32, torch.float64, True, 11)
    s2 = np.concatenate([scene_u(700, 2, 0), scene_u(800, 3, 1)], 0)
    run_case("b2_7x13_f32_noiden", s2, s2, (3, 3, 3), (1, 1, 1), 7, 13, torch.float32, False, 12)
    coarse, _ = kmap.stride_coords(s, (2, 2, 2))
    run_case("stride2_k2_16x32_f32", s, coarse, (2, 2, 2), (2, 2, 2), 16, 32, torch.float32, False, 13)
    s3 = scene_u(512, 4)
    run_case("u512_64x128_f32", s3, s3, (3, 3, 3), (1, 1, 1), 64, 128, torch.float32, True, 14)

    # ---- (c) known-answer patterns (outputs only; inputs are rebuilt from the pattern names) -----------
    s4 = scene_u(300, 5)
    found, offsets, in_maps, out_maps = brute.kernel_map(s4, s4, (3, 3, 3))
    km = IntSearchResult(torch.from_numpy(in_maps), torch.from_numpy(out_maps), torch.from_numpy(offsets), 13)
    known = dict(coords=s4, in_maps=in_maps, out_maps=out_maps, offsets=offsets)
    for (cin, cout) in [(8, 8), (7, 13), (32, 16)]:
        for wp in ["ones", "triu", "tril", "eye", "center_eye"]:
            for fp in ["ones", "range", "row_index"]:
                W = make_weight(27, cin, cout, wp, torch.float64)
                X = make_feats(len(s4), cin, fp, torch.float64)
                dY = make_grad_out(len(s4), cout, torch.float64)
                Y = _explicit_gemm_forward_logic(X, W, km, len(s4))
                dX, dW = _explicit_gemm_backward_logic(dY, X, W, km)
                tag = f"{cin}x{cout}_{wp}_{fp}"
                known[f"Y_{tag}"] = Y.numpy().astype(np.float32)
                known[f"dX_{tag}"] = dX.numpy().astype(np.float32)
                known[f"dW_{tag}"] = dW.numpy().astype(np.float32)
    np.savez_compressed(os.path.join(HERE, "known_answer.npz"), **known)

    # ---- (d) module init (seeded state_dict) -------------------------------------------------------
    init = {}
    for name, kwargs in [("c16x32_k3", dict(in_channels=16, out_channels=32, kernel_size=3)),
                         ("c64x128_k3_g4", dict(in_channels=64, out_channels=128, kernel_size=3, groups=4)),
                         ("c32x16_k2_s2_tr", dict(in_channels=32, out_channels=16, kernel_size=2, stride=2, transposed=True))]:
        torch.manual_seed(0)
        m = SparseConv3d(**kwargs)
        init[name + "_weight"] = m.weight.detach().numpy()
        init[name + "_bias"] = m.bias.detach().numpy()
    np.savez_compressed(os.path.join(HERE, "module_init.npz"), **init)

    # ---- (e) IntSearchResult container API on a toy map --------------------------------------------
    toy = IntSearchResult(torch.tensor([0, 2, 1, 3, 0], dtype=torch.int32), torch.tensor([1, 0, 2, 2, 3], dtype=torch.int32),
                          torch.tensor([0, 2, 2, 5], dtype=torch.int32))
    csr_in, csr_out, csr_off = toy.to_csr()
    np.savez_compressed(
        os.path.join(HERE, "search_result_api.npz"),
        in_maps=toy.in_maps.numpy(), out_maps=toy.out_maps.numpy(), offsets=toy.offsets.numpy(),
        length=np.asarray(len(toy)), numel=np.asarray([toy.numel(i) for i in range(3)]),
        item1_in=toy[1][0].numpy(), item2_in=toy[2][0].numpy(), item2_out=toy[2][1].numpy(),
        csr_in=csr_in.numpy(), csr_out=csr_out.numpy(), csr_off=csr_off.numpy(),
        counts=toy.neighbor_count_per_output(5).numpy(),
    )

    # ---- (f) cache-key equality truth table ----------------------------------------------------------
    o1, o2 = torch.tensor([0, 5, 9]), torch.tensor([0, 5, 10])
    base = dict(kernel_size=(3, 3, 3), kernel_dilation=(1, 1, 1), transposed=False, generative=False,
                stride_mode="stride_only", skip_symmetric_kernel_map=False, in_offsets=o1, out_offsets=o1)
    variants = [dict(), dict(kernel_size=(2, 2, 2)), dict(kernel_dilation=(2, 2, 2)), dict(transposed=True),
                dict(in_offsets=o2), dict(out_offsets=o2)]
    k0 = IntSearchCacheKey(**base)
    eq = [bool(k0 == IntSearchCacheKey(**{**base, **v})) for v in variants]
    hs = [bool(hash(k0) == hash(IntSearchCacheKey(**{**base, **v}))) for v in variants[:1]]
    np.savez_compressed(os.path.join(HERE, "cache_key.npz"), equal=np.asarray(eq), hash_equal_same=np.asarray(hs))
    print("done")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "depthwise":
        main_depthwise()  # only the (g) fixtures; the others are left untouched
    elif len(sys.argv) > 1 and sys.argv[1] == "pointconv":
        main_pointconv()  # only the (h) fixture
    elif len(sys.argv) > 1 and sys.argv[1] == "radius":
        main_radius()  # only the (i) fixture
    else:
        main()
        main_depthwise()
        main_pointconv()
        main_radius()
