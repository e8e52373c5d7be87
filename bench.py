#!/usr/bin/env python
"""Headline benchmark: M active voxels / s, forward + backward of SparseConv3d(64 -> 128, k=3), bf16.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One step = the whole hot path on one resident ~1M-voxel scene per GPU (BASELINE.json configs[1]):
kernel-map build (hash + probe + per-offset bucketing + mask sort) + AB forward + ABt dgrad + AtB wgrad,
through the SparseConv3d module under bf16 autocast, then (N > 1) one RCCL all-reduce of the weight/bias
gradients.  Inputs (coordinates, features, grad_out, fp32 master weights) are resident in HBM before the
timed region.  Prints ONE JSON line on rank 0 (see README / DESIGN.md §Measurement).
"""
import argparse
import gc
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy ceiling
CIN, COUT, KVOL = 64, 128, 27


def scene_u(n, seed):
    """Reference-style uniform scene (scripts/populate_benchmark_cache.py:283-309): extent 2*ceil(n^(1/3)),
    draw 1.3n coordinates, unique, truncate to n.  Occupancy ~0.15, L/N ~4.7."""
    rng = np.random.default_rng(seed)
    extent = 2 * int(np.ceil(n ** (1.0 / 3.0)))
    c = rng.integers(0, extent, size=(int(1.3 * n), 3))
    _, first = np.unique(c, axis=0, return_index=True)
    return c[np.sort(first)][:n].astype(np.int32)


def scene_surface(n, seed):
    """Surface-like scene (SURVEY.md §8d generator S): two height-field sheets over a side x side grid,
    side chosen so that ~n voxels come out (L/N ~ 9).  Secondary workload (`--scene surface`), never the headline."""
    rng = np.random.default_rng(seed)
    side = int(np.ceil(np.sqrt(n / 2.0)))
    xs, ys = np.meshgrid(np.arange(side), np.arange(side), indexing="ij")
    sheets = []
    for l in range(2):
        a1, a2 = rng.uniform(3, 9), rng.uniform(1, 4)
        f = rng.uniform(0.02, 0.15, size=4)
        ph = rng.uniform(0, 6.28, size=3)
        z = np.round(40 * l + a1 * np.sin(f[0] * xs + ph[0]) * np.cos(f[1] * ys + ph[1]) + a2 * np.sin(f[2] * xs + f[3] * ys + ph[2]))
        sheets.append(np.stack([xs.ravel(), ys.ravel(), z.ravel().astype(np.int64)], 1))
    c = np.unique(np.concatenate(sheets, 0), axis=0).astype(np.int32)
    rng.shuffle(c)
    return c[:n]


def algorithmic_bytes(N, L, cin=CIN, cout=COUT, K=KVOL, e=2):
    """SURVEY.md §8(d) per-iteration algorithmic bytes (N_in = N_out = N)."""
    cap = 1 << int(np.ceil(np.log2(max(16, 2 * N))))
    kmap = 16 * N + 12 * cap + 12 * N + 16 * N + 8 * K * N + 4 * L + 4 * K * N + 8 * L
    fwd = L * cin * e + K * cin * cout * e + N * cout * e + 4 * K * N
    dgrad = L * cout * e + K * cin * cout * e + N * cin * e + 4 * K * N
    wgrad = L * (cin + cout) * e + 4 * K * cin * cout + 4 * K * N
    return dict(kmap=kmap, fwd=fwd, dgrad=dgrad, wgrad=wgrad)


def time_events(fn, iters, warmup=2):
    """Average milliseconds per call measured with HIP events on the current stream."""
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    gc.collect()  # (garbage of the warm-up and of earlier sections is collected here, not inside the timed window)
    start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    start.record()
    for _ in range(iters):
        fn()
    end.record()
    end.synchronize()
    return start.elapsed_time(end) / iters


def cpu_baseline(n_sample, seed):
    """The oracle (CPU port of the reference's explicit path) timed on the host cores: C kernel-map restatement
    (1 thread) + torch fp32 gather-matmul-scatter forward/backward (all threads) on a bounded sample."""
    from oracle import conv as oconv
    from oracle import kmap as okmap

    c = scene_u(n_sample, seed)
    bc = np.concatenate([np.zeros((len(c), 1), np.int32), c], 1)
    g = torch.Generator().manual_seed(seed)
    X = torch.randn(len(c), CIN, generator=g)
    W = torch.randn(KVOL, CIN, COUT, generator=g) * 0.05
    dY = torch.randn(len(c), COUT, generator=g)
    t0 = time.perf_counter()
    r = okmap.kernel_map(bc, bc, (3, 3, 3))
    t1 = time.perf_counter()
    oconv.forward(X, W, r["in_maps"], r["out_maps"], r["offsets"], len(c), 13)  # warm-up (thread pool, pages)
    t2 = time.perf_counter()
    oconv.forward(X, W, r["in_maps"], r["out_maps"], r["offsets"], len(c), 13)
    oconv.backward(dY, X, W, r["in_maps"], r["out_maps"], r["offsets"], 13)
    t3 = time.perf_counter()
    total = (t1 - t0) + (t3 - t2)
    return dict(value=round(len(c) / total / 1e6, 4), unit="M voxels/s", cores=torch.get_num_threads(), kind="port",
                sample=f"{len(c)} voxels of the same generator, 64->128 k=3 fp32: kernel map {t1 - t0:.2f}s (C, 1 thread) "
                       f"+ fwd+bwd {t3 - t2:.2f}s (torch CPU, {torch.get_num_threads()} threads)")


def conv_bytes(rec, e=2):
    """SURVEY.md §8(d) byte model of one convolution layer, forward + dgrad + wgrad (16-bit features, fp32 weight gradient)."""
    cin, cout, K, n_in, n_out, L = rec["cin"], rec["cout"], rec["K"], rec["n_in"], rec["n_out"], rec["pairs"]
    w = K * cin * cout
    fwd = L * cin * e + w * e + n_out * cout * e + (4 * K * n_out if K > 1 else 0)
    dgrad = L * cout * e + w * e + n_in * cin * e + (4 * K * n_in if K > 1 else 0)
    wgrad = L * (cin + cout) * e + 4 * w + (8 * L if K > 1 else 0)
    return fwd + dgrad + wgrad


def secondary_configs(dev, args):
    """BASELINE configs 3 and 5 and the S scene of config 2 as secondary figures of the same run (never the headline):
    the headline step on a surface-like 1 M-voxel scene; MinkUNet-14 forward + backward on surface scenes of 200 k and 1 M
    voxels; PointConv(32->64, kNN 16) on 200 k points followed by voxelisation and a depthwise k=3 convolution, forward +
    backward (with the one-kernel edge pipeline, and composed from separate kernels).  HIP events, warm-up + timed."""
    from bench_models import ConvLayerRecorder, MinkUNet14
    from warpconvnet_amd.geometry.coords.search.search_configs import RealSearchConfig
    from warpconvnet_amd.geometry.types.points import Points
    from warpconvnet_amd.geometry.types.voxels import Voxels
    from warpconvnet_amd.nn.modules import PointConv, SparseDepthwiseConv3d
    from warpconvnet_amd.nn.modules.sparse_conv import SparseConv3d

    out = {}
    # ---- config 2 on generator S (ScanNet-like sheets, ~9 pairs per voxel): the headline step, module API, fresh map per step
    cs = torch.from_numpy(scene_surface(args.voxels, seed=1000)).to(dev)
    ns = cs.shape[0]
    g = torch.Generator().manual_seed(7)
    fs = torch.randn(ns, CIN, generator=g).to(dev, torch.bfloat16).requires_grad_(True)
    gs = torch.randn(ns, COUT, generator=g).to(dev, torch.bfloat16)
    offs = torch.tensor([0, ns], dtype=torch.int32)
    torch.manual_seed(0)
    conv_s = SparseConv3d(CIN, COUT, 3, bias=True).to(dev)
    pairs = [0]

    def surface_step():
        for p in conv_s.parameters():
            p.grad = None
        fs.grad = None
        x = Voxels(cs, fs, offsets=offs)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y = conv_s(x)
        y.batched_features.batched_tensor.backward(gs)
        if not pairs[0]:
            pairs[0] = int(next(iter(x.cache.values())).offsets[-1])

    ms = time_events(surface_step, 50, warmup=10)
    ab = algorithmic_bytes(ns, pairs[0])
    out["surface_1M"] = {"value": round(ns / (ms * 1e-3) / 1e6, 3), "unit": "M voxels/s", "ms_per_step": round(ms, 4),
                         "voxels": ns, "pairs_per_voxel": round(pairs[0] / ns, 2),
                         "whole_step_hbm_frac": round(sum(ab.values()) / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}

    # ---- the headline step over a ROTATING set of resident U scenes of different size (0.6 / 0.8 / 1.0 / 1.2 x the headline):
    # every build meets sizing hints (block-table bound, speculative pair capacity) learned on a DIFFERENT scene
    from warpconvnet_amd.geometry.coords.search.torch_discrete import default_hints

    rot = []
    for i, frac in enumerate((0.6, 0.8, 1.0, 1.2)):
        cr = torch.from_numpy(scene_u(int(args.voxels * frac), seed=2000 + i)).to(dev)
        gr = torch.Generator().manual_seed(20 + i)
        rot.append((cr, torch.randn(cr.shape[0], CIN, generator=gr).to(dev, torch.bfloat16).requires_grad_(True),
                    torch.randn(cr.shape[0], COUT, generator=gr).to(dev, torch.bfloat16), torch.tensor([0, cr.shape[0]], dtype=torch.int32)))
    turn = [0]

    def rotating_step():
        cr, fr, gr_, offr = rot[turn[0] % len(rot)]
        turn[0] += 1
        for p in conv_s.parameters():
            p.grad = None
        fr.grad = None
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y = conv_s(Voxels(cr, fr, offsets=offr))
        y.batched_features.batched_tensor.backward(gr_)

    for _ in range(8):
        rotating_step()
    stats0 = dict(default_hints().stats)
    turn[0] = 0
    ms = time_events(rotating_step, 48, warmup=0)
    stats1 = default_hints().stats
    vox_per_step = sum(r[0].shape[0] for r in rot) / len(rot)
    out["rotating_scenes"] = {"value": round(vox_per_step / (ms * 1e-3) / 1e6, 3), "unit": "M voxels/s", "ms_per_step": round(ms, 4),
                              "scene_voxels": [int(r[0].shape[0]) for r in rot], "steps": 48,
                              "builds": stats1["builds"] - stats0["builds"], "rebuilds": stats1["rebuilds"] - stats0["rebuilds"],
                              "pair_list_rewrites": stats1["pair_rewrites"] - stats0["pair_rewrites"],
                              "note": "headline step cycling over four resident U scenes: sizing hints always come from another scene"}
    del rot

    # ---- large kernel volumes on the 200 k surface scene, 32 -> 64: map build alone and the whole step ----
    from warpconvnet_amd.geometry.coords.search.torch_discrete import generate_kernel_map

    c3 = torch.from_numpy(scene_surface(200_000, seed=3)).to(dev)
    n3 = c3.shape[0]
    b3 = torch.cat([torch.zeros(n3, 1, dtype=torch.int32, device=dev), c3], 1).contiguous()
    g3 = torch.Generator().manual_seed(9)
    f3 = torch.randn(n3, 32, generator=g3).to(dev, torch.bfloat16).requires_grad_(True)
    d3 = torch.randn(n3, 64, generator=g3).to(dev, torch.bfloat16)
    off3 = torch.tensor([0, n3], dtype=torch.int32)
    big = {}
    for label, ks, dil in (("k3", 3, 1), ("k5", 5, 1), ("k7", 7, 1), ("k3_dilation9_hash", 3, 9)):
        torch.manual_seed(0)
        conv3 = SparseConv3d(32, 64, ks, dilation=dil, bias=True).to(dev)
        K3 = ks ** 3
        km3 = generate_kernel_map(b3, b3, (1, 1, 1), (ks,) * 3, (dil,) * 3)
        L3 = int(km3.offsets[-1])
        t_map = time_events(lambda: generate_kernel_map(b3, b3, (1, 1, 1), (ks,) * 3, (dil,) * 3).in_maps_device, 10, warmup=2)

        def big_step():
            for p in conv3.parameters():
                p.grad = None
            f3.grad = None
            with torch.autocast("cuda", dtype=torch.bfloat16):
                y = conv3(Voxels(c3, f3, offsets=off3))
            y.batched_features.batched_tensor.backward(d3)

        t_step = time_events(big_step, 10, warmup=3)
        ab3 = algorithmic_bytes(n3, L3, 32, 64, K3)
        big[label] = {"num_offsets": K3, "pairs": L3, "map_build_ms": round(t_map, 4), "step_ms": round(t_step, 4),
                      "ns_per_probe": round(t_map * 1e6 / (K3 * n3), 3),
                      "map_frac": round(ab3["kmap"] / (t_map * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                      "step_frac": round(sum(ab3.values()) / (t_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                      "builder": "hash" if dil * (ks // 2) > 8 else "cell table"}
        del km3
    out["large_kernels_200k"] = {"voxels": n3, "channels": "32->64", **big,
                                 "note": "map build (incl. pair lists) and fwd+bwd step with a fresh map; frac = SURVEY 8d algorithmic bytes / time / 8 TB/s; "
                                         "k3_dilation9 has a halo of 9 cells: beyond the cell table's 8, the global hash path"}

    # ---- config 3: MinkUNet-14 ----
    torch.manual_seed(0)
    net = MinkUNet14(3, 20).to(dev)
    for label, n_vox in (("200k", 200_000), ("1M", 1_000_000)):
        c = torch.from_numpy(scene_surface(n_vox, seed=3)).to(dev)
        n = c.shape[0]
        f = torch.randn(n, 3, device=dev)
        off = torch.tensor([0, n], dtype=torch.int32)

        def unet_step():
            net.zero_grad(set_to_none=True)
            with torch.autocast("cuda", dtype=torch.bfloat16):
                y = net(Voxels(c, f, offsets=off))
            y.feature_tensor.float().square().mean().backward()

        rec = ConvLayerRecorder(net)
        unet_step()
        rec.close()
        ms = time_events(unet_step, 10, warmup=3)
        out[f"minkunet14_{label}_ms"] = round(ms, 3)
        out[f"minkunet14_{label}_voxels"] = n
        layer_bytes = [conv_bytes(r) for r in rec.records]
        top = max(range(len(layer_bytes)), key=lambda i: layer_bytes[i])
        tr = rec.records[top]
        out[f"minkunet14_{label}_roofline"] = {
            "bound": "hbm", "scope": "all convolution layers, forward + dgrad + wgrad (SURVEY 8d byte model per layer; BatchNorm / ReLU "
                                     "passes and map builds not counted), over the whole iteration's time",
            "algorithmic_bytes": int(sum(layer_bytes)), "achieved": round(sum(layer_bytes) / (ms * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS,
            "unit": "GB/s", "frac": round(sum(layer_bytes) / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "conv_layers": len(layer_bytes),
            "dominant_layer": f"{tr['cin']}->{tr['cout']} K={tr['K']} N_out={tr['n_out']} pairs={tr['pairs']}",
            "dominant_layer_bytes": int(layer_bytes[top]),
        }

    # ---- config 5: PointConv + depthwise ----
    g = torch.Generator().manual_seed(5)
    n = 200_000
    pts = (torch.rand(n, 3, generator=g) * torch.tensor([50.0, 50.0, 4.0])).to(dev)
    pf = torch.randn(n, 32, generator=g).to(dev)
    torch.manual_seed(0)
    knn_k = 16
    pconv = PointConv(32, 64, RealSearchConfig(mode="knn", knn_k=knn_k)).to(dev)
    dw = SparseDepthwiseConv3d(64, 3).to(dev)

    def point_step():
        pconv.zero_grad(set_to_none=True)
        dw.zero_grad(set_to_none=True)
        o = pconv(Points(pts, pf, offsets=torch.tensor([0, n])))
        y = dw(o.to_voxels(0.25))
        y.feature_tensor.sum().backward()

    ms = time_events(point_step, 5, warmup=3)
    out["pointconv_dw_ms"] = round(ms, 3)
    out["pointconv_dw_points"] = n
    # edge MLP on the fp32 matrix cores: Linear(e_in -> hidden) + Linear(hidden -> out) per edge; the backward recomputes the
    # forward and adds the data-gradient and weight-gradient products (3x the forward's multiply-adds)
    lin = [m for m in pconv.edge_transform_mlp.modules() if isinstance(m, torch.nn.Linear)] if hasattr(pconv, "edge_transform_mlp") else []
    macs = sum(m.in_features * m.out_features for m in lin)
    flops = 2.0 * n * knn_k * macs * 4 if macs else None
    if flops:
        out["pointconv_dw_roofline"] = {"bound": "mfma", "kernel": "pointconv edge kernels (forward + backward), fp32 v_mfma_f32_32x32x2_f32",
                                        "flops": flops, "achieved": round(flops / (ms * 1e-3) / 1e12, 2), "peak": 157.3, "unit": "TFLOP/s",
                                        "frac": round(flops / (ms * 1e-3) / 1e12 / 157.3, 4),
                                        "note": "edge-MLP products only, over the whole step's time (kNN, voxelisation, depthwise conv included in the time)"}
    # the same step with the edge pipeline composed from separate kernels (what the reference's op sequence costs here)
    from warpconvnet_amd.nn.functional import point_conv as fpc
    fpc._ENABLED = False
    try:
        out["pointconv_dw_composed_ms"] = round(time_events(point_step, 3, warmup=1), 3)
    finally:
        fpc._ENABLED = True
    return out


def report(args, dev, world, coords, feats, grad_out, offsets, conv, params, N, value, ms_per_step, second_half):
    """Rank 0: per-kernel timing IN THE STEP (HIP events on the launch stream between the four phases of an unrolled step,
    so every kernel meets the cache state it meets in the module step), roofline figures, secondary configs, CPU baseline."""
    from warpconvnet_amd import _lib
    from warpconvnet_amd.geometry.coords.search.torch_discrete import generate_kernel_map
    from warpconvnet_amd.geometry.types.voxels import Voxels
    from warpconvnet_amd.nn.functional.sparse_conv.detail import hip_gemm

    bcoords = torch.cat([torch.zeros(N, 1, dtype=torch.int32, device=dev), coords], 1).contiguous()
    km = generate_kernel_map(bcoords, bcoords, (1, 1, 1), (3, 3, 3))
    L = int(km.offsets[-1])
    X = feats.detach()
    W = conv.weight.detach().to(torch.bfloat16)
    it = max(5, min(args.steps, 40))
    Lc = _lib.lib()
    stream = _lib.stream_handle(dev)
    wp_f = hip_gemm.pack_weight(W, False, False)
    wp_d = hip_gemm.pack_weight(W, True, True)
    y_buf = torch.empty(N, COUT, dtype=torch.bfloat16, device=dev)
    dx_buf = torch.empty(N, CIN, dtype=torch.bfloat16, device=dev)
    dw_buf = torch.empty(KVOL, CIN, COUT, dtype=torch.float32, device=dev)
    db_buf = torch.empty(COUT, dtype=torch.float32, device=dev)
    ws_bytes = Lc.wcn_conv_wgrad_workspace(KVOL, CIN, COUT, _lib.WCN_ALGO_MFMA)
    ws_buf = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    bias = conv.bias.detach().float()

    def k_fwd(m):
        # (as the module launches it: the binned builder's compact rows and no mask array - `hip_gemm.own_tables`)
        tb, mk = hip_gemm.own_tables(m, CIN, COUT, KVOL, torch.bfloat16)
        Lc.wcn_conv_gather_gemm(_lib.ptr(X), _lib.ptr(wp_f), _lib.ptr(y_buf), _lib.ptr(tb), _lib.ptr(mk),
                                _lib.ptr(m._perm), _lib.ptr(bias), N, N, CIN, COUT, KVOL, _lib.WCN_BF16, _lib.WCN_ALGO_MFMA, 0, 0, stream)

    def k_dgrad(m):
        tb, mk = hip_gemm.own_tables(m, COUT, CIN, KVOL, torch.bfloat16)
        Lc.wcn_conv_gather_gemm(_lib.ptr(grad_out), _lib.ptr(wp_d), _lib.ptr(dx_buf), _lib.ptr(tb), _lib.ptr(mk),
                                _lib.ptr(m._perm), None, N, N, COUT, CIN, KVOL, _lib.WCN_BF16, _lib.WCN_ALGO_MFMA, 1, 1, stream)

    def k_wgrad(m):
        # the weight-gradient entry point as the training step calls it (bias gradient fused): main kernel + the two
        # small fixed-order reduce kernels
        Lc.wcn_conv_wgrad_bias(_lib.ptr(X), _lib.ptr(grad_out), _lib.ptr(dw_buf), _lib.ptr(m.in_maps_device),
                               _lib.ptr(m.out_maps_device), _lib.ptr(m._offsets_dev), N, N, CIN, COUT, KVOL, _lib.WCN_BF16,
                               KVOL // 2, _lib.ptr(db_buf), _lib.ptr(ws_buf), ws_bytes, stream)

    # ---- unrolled step: map build -> forward -> wgrad -> dgrad back to back, one event between the phases ----
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(5)] for _ in range(it)]
    for rep in range(-2, it):
        e = ev[max(rep, 0)]
        e[0].record()
        # as the module does it: the build is queued, the forward kernel right behind it, THEN the host reads the build's
        # status word and queues the pair-list scatter (wgrad's input) - charged to the forward phase it overlaps with
        m = generate_kernel_map(bcoords, bcoords, (1, 1, 1), (3, 3, 3), optimistic=True)
        e[1].record()
        k_fwd(m)
        assert not m.validate()
        e[2].record()
        k_wgrad(m)  # (the module's backward order since round 5: weight gradient first - `detail/backends.py`)
        e[3].record()
        k_dgrad(m)
        e[4].record()
    torch.cuda.synchronize()
    in_step = [float(np.mean([ev[r][p].elapsed_time(ev[r][p + 1]) for r in range(it)])) for p in range(4)]
    t_kmap, tk_fwd, tk_wgrad, tk_dgrad = in_step
    # isolated figures (back-to-back launches of one kernel find part of their operands in the 256 MiB Infinity Cache):
    # reported next to the in-step ones, never used for the roofline
    iso = {"fwd": time_events(lambda: k_fwd(km), it), "dgrad": time_events(lambda: k_dgrad(km), it),
           "wgrad": time_events(lambda: k_wgrad(km), it),
           "kmap": time_events(lambda: generate_kernel_map(bcoords, bcoords, (1, 1, 1), (3, 3, 3)).in_maps_device, it)}
    # (isolated "kmap" includes the pair-list scatter; in the step the scatter is queued behind the forward kernel)

    ab = algorithmic_bytes(N, L)
    e = 2
    comp = {  # compulsory traffic: every L*C term replaced by N*C - what an ideal cache would leave
        "kmap": ab["kmap"],
        "fwd": N * CIN * e + KVOL * CIN * COUT * e + N * COUT * e + 4 * KVOL * N,
        "dgrad": N * COUT * e + KVOL * CIN * COUT * e + N * CIN * e + 4 * KVOL * N,
        "wgrad": N * (CIN + COUT) * e + 4 * KVOL * CIN * COUT + 4 * KVOL * N,
    }
    names = {
        "fwd": "gather_gemm_cs_kernel<bf16,CO=128> (fwd)",
        "dgrad": "gather_gemm_cs_kernel<bf16,CO=64> (dgrad)",
        "wgrad": "wgrad_mfma_kernel<bf16,64,128> (+ wgrad_reduce)",
        "kmap": "kernel map build (all launches)",
    }
    times = {"fwd": tk_fwd, "dgrad": tk_dgrad, "wgrad": tk_wgrad, "kmap": t_kmap}

    def entry(key):
        ms = times[key]
        return {"achieved": round(ab[key] / (ms * 1e-3) / 1e9, 1), "frac": round(ab[key] / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                "compulsory_frac": round(comp[key] / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "avg_launch_ms": round(ms, 4),
                "isolated_ms": round(iso[key], 4), "algorithmic_bytes_per_launch": int(ab[key])}

    # the dominant PHASE = the one with the largest IN-STEP time, the kernel-map build (a chain of launches) included
    dom_key = max(("fwd", "dgrad", "wgrad", "kmap"), key=lambda k: times[k])
    dom = entry(dom_key)
    # HBM bytes per launch from the PMC passes (separate rocprofv3 runs of this same command, tools/collect_profiles.sh): a
    # STATIC figure read from the newest committed summary, not a measurement of this run - labelled as such in the line
    traffic, traffic_src = None, None
    prof_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles")
    pmc_files = sorted(f for f in (os.listdir(prof_dir) if os.path.isdir(prof_dir) else []) if f.endswith("_pmc_traffic.json"))
    if N == 1_000_000 and args.scene == "uniform" and pmc_files:
        with open(os.path.join(prof_dir, pmc_files[-1])) as f:
            pmc = json.load(f)
        from warpconvnet_amd.utils.codesig import phase_signatures

        running = phase_signatures(_lib.LIB_PATH).get(dom_key)
        stamped = (pmc.get("kernel_signatures") or {}).get(dom_key)
        if dom_key in pmc and stamped is not None and stamped == running:
            traffic = pmc[dom_key]["hbm_bytes"]
            traffic_src = (f"profiles/{pmc_files[-1]} (2 x FETCH_SIZE + WRITE_SIZE, separate rocprofv3 --pmc passes of this "
                           f"command; static: read from the committed summary, whose kernel signature {stamped} equals the running "
                           "library's)")
        elif dom_key in pmc:
            traffic_src = (f"profiles/{pmc_files[-1]} was collected on other kernels (signature {stamped} vs the running library's "
                           f"{running}): no traffic figure until tools/collect_profiles.sh has been rerun")

    # map-cached variant (networks amortise the map over the layers of a resolution level)
    x_cached = Voxels(coords, feats, offsets=offsets)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        conv(x_cached)  # fills x_cached.cache

    def cached_step():
        for p in params:
            p.grad = None
        feats.grad = None
        with torch.autocast("cuda", dtype=torch.bfloat16):
            yc = conv(x_cached)
        yc.batched_features.batched_tensor.backward(grad_out)

    t_cached = time_events(cached_step, it)
    total_bytes = sum(ab.values())
    result = {
        "metric": "M active voxels/sec fwd+bwd, SparseConv3d 64\u2192128 k=3, 1/2/4/8 MI355X",  # = BASELINE.json:metric
        "value": round(value, 3),
        "unit": "M voxels/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "settle_steps": args.settle,
        "ms_per_step": round(ms_per_step, 4),
        "second_half": second_half,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "bf16",
        "data": "synthetic",
        "config": {
            "workload": f"configs[1]: one {N}-voxel {'uniform (U)' if args.scene == 'uniform' else 'surface-like (S, secondary)'} scene per GPU, SparseConv3d 64->128 k=3, bf16 autocast, "
                        "kernel-map build + AB fwd + ABt dgrad + AtB wgrad per step",
            "voxels_per_gpu": N, "pairs_per_scene": L, "coord_order": args.coord_order, "parallelism": f"dp{world} (scene-sharded, grad all-reduce)" + ("" if world == 1 else "; N > 1 has run under gloo on CPU only before this launch (no earlier RCCL measurement exists)"),
        },
        "roofline": {
            "bound": "hbm", "kernel": names[dom_key], "phase": dom_key, "achieved": dom["achieved"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": dom["frac"], "compulsory_frac": dom["compulsory_frac"], "traffic": traffic, "traffic_static": traffic is not None,
            "traffic_source": traffic_src,
            "algorithmic_bytes_per_launch": dom["algorithmic_bytes_per_launch"], "avg_launch_ms": dom["avg_launch_ms"],
            "timing": "HIP events on the launch stream between the phases of an unrolled step (map build -> fwd -> wgrad -> dgrad), "
                      "i.e. in-step; the rocprofv3 trace of this command is the newest profiles/r*_kernel_trace_stats.md",
        },
        "step_includes": "kernel-map build, forward, dgrad, wgrad + bias gradient, SGD parameter update (so both packed bf16 "
                         "weight images are rebuilt every step)",
        "roofline_all": {names[k]: entry(k) for k in ("fwd", "dgrad", "wgrad", "kmap")},
        "phases_ms": {"kmap": round(t_kmap, 4), "fwd_kernel": round(tk_fwd, 4), "dgrad_kernel": round(tk_dgrad, 4),
                      "wgrad_kernels": round(tk_wgrad, 4), "sum": round(sum(in_step), 4)},
        "whole_step_hbm_frac": round(total_bytes / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
        "whole_step_compulsory_hbm_frac": round(sum(comp.values()) / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
        "map_cached": {"value": round(N / (t_cached * 1e-3) / 1e6, 3), "unit": "M voxels/s", "ms_per_step": round(t_cached, 4),
                       "note": "same step with the kernel map taken from the geometry's cache (fwd + dgrad + wgrad only)"},
    }
    if world == 1 and not args.no_secondary:
        try:
            result["secondary"] = secondary_configs(dev, args)
        except Exception as exc:  # the headline line must come out even if a secondary workload fails
            result["secondary"] = {"error": f"{type(exc).__name__}: {exc}"}
    if world == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(args.cpu_sample, 1000)
    return result


def build_workload(args, dev, rank, conv_kwargs=None):
    """Resident inputs: one scene per rank (weak scaling), identical weights on every rank."""
    from warpconvnet_amd.nn.modules.sparse_conv import SparseConv3d

    c_np = (scene_u if args.scene == "uniform" else scene_surface)(args.voxels, seed=1000 + rank)
    if args.coord_order == "block":
        key = (((c_np[:, 0] >> 4) * 4096 + (c_np[:, 1] >> 4)) * 4096 + (c_np[:, 2] >> 4)).astype(np.int64)
        c_np = c_np[np.lexsort((c_np[:, 2], c_np[:, 1], c_np[:, 0], key))]
    coords = torch.from_numpy(np.ascontiguousarray(c_np)).to(dev)
    N = coords.shape[0]
    g = torch.Generator().manual_seed(rank)
    fdtype = torch.bfloat16 if dev.type == "cuda" else torch.float32
    # features resident in device memory; a leaf that requires grad, so the backward pass runs ABt (dgrad) as well as AtB
    feats = torch.randn(N, CIN, generator=g).to(dev, fdtype).requires_grad_(True)
    grad_out = torch.randn(N, COUT, generator=g).to(dev, fdtype)
    offsets = torch.tensor([0, N], dtype=torch.int32)
    torch.manual_seed(0)
    conv = SparseConv3d(CIN, COUT, 3, bias=True, **(conv_kwargs or {})).to(dev)
    return coords, feats, grad_out, offsets, conv, [p for p in conv.parameters()]


def make_step(dev, world, coords, feats, grad_out, offsets, conv, params, attach=None):
    """The timed step, exactly as `main` runs it: fresh geometry -> kernel map rebuilt -> forward -> backward (dgrad + wgrad) ->
    (N > 1) the flat-bucket gradient all-reduce launched from the autograd hook of the last gradient.  `attach(voxels)` lets
    the CPU test (tests/test_dist_gloo.py: gloo, explicit backend) supply an oracle-built kernel map; None on the GPU."""
    from warpconvnet_amd.dist import GradientBuckets
    from warpconvnet_amd.geometry.types.voxels import Voxels

    buckets = GradientBuckets(params) if world > 1 else None

    def step():
        if buckets is not None:
            buckets.zero_grad()
        else:
            for p in params:
                p.grad = None
        feats.grad = None
        x = Voxels(coords, feats, offsets=offsets)  # fresh geometry -> kernel map rebuilt every step
        if attach is not None:
            attach(x)
        if dev.type == "cuda":
            with torch.autocast("cuda", dtype=torch.bfloat16):
                y = conv(x)
        else:
            y = conv(x)
        y.batched_features.batched_tensor.backward(grad_out)
        if buckets is not None:
            buckets.finish()
        # the parameter update of a training step (plain SGD, lr 1e-6): bumps the parameters' version counters, so the packed
        # bf16 weight images are rebuilt next step like in any real training loop (two pack launches per step)
        # (two elementwise launches, 4.5 us each; the multi-tensor form torch.optim.SGD uses - `torch._foreach_add_` - is ONE
        # launch of 18 us for these two small tensors on this stack: measured, profiles/r05_kernel_trace_stats.md history)
        with torch.no_grad():
            for p in params:
                if p.grad is not None:
                    p.add_(p.grad, alpha=-1e-6)

    return step, buckets


def timed_loop(step, args, dev, world):
    """W warm-up steps, then EXACTLY K timed steps between barrier + synchronize on both sides; the maximum over ranks.
    A fresh process (and a fresh box) runs its first ~60 steps at cold clocks with first-use allocator / pinned-memory /
    kernel-module costs (830-860 instead of 960-1 020 M voxels/s over a 20-step window).  Nothing hidden: `--settle` (default
    0, printed as settle_steps) adds untimed steps, the default K = 200 timed steps carry the cold start inside the timed
    region, and `second_half` reports the rate of the last K/2 steps (HIP events, no extra synchronisation)."""
    cuda = dev.type == "cuda"
    sync = torch.cuda.synchronize if cuda else (lambda: None)
    for _ in range(args.settle):
        step()
    sync()
    for _ in range(args.warmup):
        step()
    sync()
    if world > 1:
        dist.barrier()
    sync()
    ev_mid = torch.cuda.Event(enable_timing=True) if cuda else None
    ev_end = torch.cuda.Event(enable_timing=True) if cuda else None
    half = args.steps // 2
    t0 = time.perf_counter()
    t_mid = t0
    for i in range(args.steps):
        if i == half:
            if cuda:
                ev_mid.record()
            t_mid = time.perf_counter()
        step()
    if cuda:
        ev_end.record()
    sync()
    if world > 1:
        dist.barrier()
    sync()
    elapsed = time.perf_counter() - t0
    ms_half = (ev_mid.elapsed_time(ev_end) if cuda else (time.perf_counter() - t_mid) * 1e3) / max(1, args.steps - half)
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    return elapsed, ms_half


def multi_gpu_diag(dev, rank, world, params, iters=20):
    """The first multi-GPU run is also the first RCCL run: make it self-diagnosing - which ranks took part (an all-gather of the
    ranks), and what ONE gradient all-reduce of this step's size costs on its own (the step hides it behind the backward).
    Collective: every rank calls it.  Also runs under gloo on CPU (`tests/test_dist_gloo.py`)."""
    cuda = dev.type == "cuda"
    seen = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(seen, torch.tensor([rank], dtype=torch.int64, device=dev))
    flat = torch.zeros(sum(p.numel() for p in params), dtype=torch.float32, device=dev)
    for _ in range(3):
        dist.all_reduce(flat)
    if cuda:
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            dist.all_reduce(flat)
        e1.record()
        e1.synchronize()
        ms = e0.elapsed_time(e1) / iters
    else:
        t0 = time.perf_counter()
        for _ in range(iters):
            dist.all_reduce(flat)
        ms = (time.perf_counter() - t0) * 1e3 / iters
    return {"ranks_seen": sorted(int(t.item()) for t in seen), "grad_allreduce_ms": round(ms, 4),
            "grad_allreduce_bytes": int(flat.numel() * 4), "backend": str(dist.get_backend())}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--settle", type=int, default=0,
                    help="extra untimed steps in front of the warm-up (reported as settle_steps): a fresh process on a fresh box "
                         "runs its first ~60 steps at cold clocks / first-use allocator cost")
    ap.add_argument("--voxels", type=int, default=1_000_000)
    ap.add_argument("--cpu-sample", type=int, default=1_000_000)  # the whole configs[1] scene: ~10-15 s of CPU work
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the surface-scene / MinkUNet-14 / PointConv secondary timings")
    ap.add_argument("--coord-order", choices=["generator", "block"], default="generator",
                    help="generator: rows in the order the reference's generator emits them (the headline workload); "
                         "block: the same scene with rows sorted by 16^3 block then x,y,z (what a voxelised scan looks like) - "
                         "a locality study, never the headline")
    ap.add_argument("--scene", choices=["uniform", "surface"], default="uniform",
                    help="uniform: the reference-style generator U the metric is quoted on; surface: generator S of "
                         "SURVEY.md §8d (ScanNet-like sheets, ~2x the pairs per voxel) - a secondary workload")
    return ap.parse_args(argv)


def main():
    args = parse_args()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    from warpconvnet_amd import _lib

    _lib.lib()  # fail loudly if the HIP extension is missing

    coords, feats, grad_out, offsets, conv, params = build_workload(args, dev, rank)
    N = coords.shape[0]
    # Python's cyclic collector walks every tracked object of the process on a full collection - after `import torch` that is
    # a 100-140 ms host stall whenever one happens to fall into a timed window (measured: the first timed step of the
    # surface-scene section, 3.0 instead of 1.4 ms per step over its 50 steps, depending on nothing but how the script was
    # started).  The collector stays ON; the objects that exist now (modules, the workload) move to the permanent generation,
    # so later collections only walk what the steps themselves allocate.
    gc.collect()
    gc.freeze()
    # N > 1: persistent flat gradient bucket, the all-reduce is launched from the autograd hook of the last gradient
    step, _ = make_step(dev, world, coords, feats, grad_out, offsets, conv, params)
    elapsed, ms_half = timed_loop(step, args, dev, world)
    second_half = {"value": round(N * world / (ms_half * 1e-3) / 1e6, 3), "ms_per_step": round(ms_half, 4),
                   "steps": args.steps - args.steps // 2,
                   "note": "rank 0's last K/2 timed steps between two HIP events (steady state; the headline value covers all K)"}
    ms_per_step = elapsed / args.steps * 1e3
    value = N * world * args.steps / elapsed / 1e6

    diag = multi_gpu_diag(dev, rank, world, params) if world > 1 else None
    result = None
    if rank == 0:
        result = report(args, dev, world, coords, feats, grad_out, offsets, conv, params, N, value, ms_per_step, second_half)
        if diag is not None:
            result["multi_gpu"] = diag
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(result))


if __name__ == "__main__":
    main()
