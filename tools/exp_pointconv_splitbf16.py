"""Numerical experiment (CPU, no GPU needed): what three-term split-bf16 products would do to the PointConv edge MLP.

The round-5 review's route for the PointConv backward is `v_mfma_f32_32x32x16_bf16` on operands split as x = hi + lo
(hi = bf16(x), lo = bf16(x - hi)), products hi.hi + hi.lo + lo.hi accumulated in fp32 - 4 x fewer matrix cycles than
`v_mfma_f32_32x32x2_f32`.  This script restates the edge pipeline of `csrc/pointconv.hip` (Linear - LayerNorm - ReLU - Linear -
LayerNorm + identity shortcut, sum over the k edges of a query; reference `nn/modules/mlp.py:124-177`, `point_conv.py:231-273`) with
every matrix product replaced by that arithmetic, and compares forward output, input gradient and parameter gradients with fp64 -
on the shapes of `tests/test_gpu_points.py::test_pointconv_fused_edge_kernel_vs_fp64`, with its tolerances.

    python tools/exp_pointconv_splitbf16.py            # prints the table of profiles/r06_pointconv_splitbf16.md
"""
import sys

import torch


def split(x):
    hi = x.to(torch.bfloat16).to(torch.float32)
    lo = (x - hi).to(torch.bfloat16).to(torch.float32)
    return hi, lo


def mm_split(a, b, terms=3):
    """a @ b with bf16-split operands, fp32 accumulation (the MFMA accumulates in fp32; torch's fp32 matmul of bf16-exact
    values is exact in every product, only the summation order differs)."""
    ah, al = split(a)
    bh, bl = split(b)
    out = ah @ bh
    if terms >= 2:
        out = out + ah @ bl
    if terms >= 3:
        out = out + al @ bh
    return out


def mm_f32(a, b, terms=0):
    return a @ b


def ln(x, g, b, eps=1e-5):
    mu = x.mean(1, keepdim=True)
    var = x.var(1, unbiased=False, keepdim=True)
    xh = (x - mu) / torch.sqrt(var + eps)
    return xh * g + b, xh, torch.rsqrt(var + eps)


def run(mm, mm_fwd1, x, W1, b1, g1, be1, W2, b2, g2, be2, dy, terms):
    """forward + hand-written backward of the edge chain with `mm` as the matrix product (`mm_fwd1`: the product of GEMM1,
    whose sign decides the ReLU)."""
    hpre = mm_fwd1(x, W1.t(), terms) + b1
    h1, xh1, rstd1 = ln(hpre, g1, be1)
    H = torch.relu(h1)
    opre = mm(H, W2.t(), terms) + b2
    o, xh2, rstd2 = ln(opre, g2, be2)
    y = o + x[:, : o.shape[1]]
    # backward
    C2, C1 = o.shape[1], H.shape[1]
    gd = dy * g2
    dopre = rstd2 * (gd - gd.mean(1, keepdim=True) - xh2 * (gd * xh2).mean(1, keepdim=True))
    dW2 = mm(dopre.t(), H, terms)
    dH = mm(dopre, W2, terms)
    g = dH * (h1 > 0)
    gg = g * g1
    dhpre = rstd1 * (gg - gg.mean(1, keepdim=True) - xh1 * (gg * xh1).mean(1, keepdim=True))
    dW1 = mm(dhpre.t(), x, terms)
    dx = mm(dhpre, W1, terms)
    dx[:, :C2] += dy
    return dict(y=y, dx=dx, dW1=dW1, dW2=dW2, dg1=(g * xh1).sum(0), dg2=(dy * xh2).sum(0), relu=(h1 > 0))


def main():
    torch.manual_seed(0)
    rows = []
    for (E, ein, hid, co) in ((48000, 64, 128, 64), (24000, 32, 64, 32), (51200, 64, 128, 64)):
        g = torch.Generator().manual_seed(E + ein)
        x = torch.randn(E, ein, generator=g)
        W1 = torch.randn(hid, ein, generator=g) / ein ** 0.5
        W2 = torch.randn(co, hid, generator=g) / hid ** 0.5
        b1, b2 = torch.randn(hid, generator=g) * 0.1, torch.randn(co, generator=g) * 0.1
        g1, g2 = torch.rand(hid, generator=g) + 0.5, torch.rand(co, generator=g) + 0.5
        be1, be2 = torch.rand(hid, generator=g) - 0.5, torch.rand(co, generator=g) - 0.5
        dy = torch.randn(E, co, generator=g)
        args = (x, W1, b1, g1, be1, W2, b2, g2, be2, dy)
        ref = run(mm_f32, mm_f32, *(t.double() for t in args), 0)
        variants = {
            "fp32 products (the shipped kernel's arithmetic)": (mm_f32, mm_f32, 0),
            "split-bf16, 3 terms, all six GEMMs": (mm_split, mm_split, 3),
            "split-bf16, 3 terms, GEMM1 kept in fp32": (mm_split, mm_f32, 3),
            "split-bf16, 2 terms (hi.hi + hi.lo), all six": (mm_split, mm_split, 2),
            "plain bf16 products (1 term)": (mm_split, mm_split, 1),
        }
        for name, (mm, mm1, terms) in variants.items():
            got = run(mm, mm1, *args, terms)
            flips = float((got["relu"] != ref["relu"]).double().mean())
            edge_flip = float((got["relu"] != ref["relu"]).any(1).double().mean())
            def rel(k):
                return float((got[k].double() - ref[k]).abs().max() / ref[k].abs().max())
            dxe = (got["dx"].double() - ref["dx"]).abs()
            bad = float((dxe > 1e-4 + 1e-3 * ref["dx"].abs()).double().mean())
            rows.append((f"{E} x {ein}->{hid}->{co}", name, rel("y"), rel("dW1"), rel("dW2"), rel("dg1"), rel("dg2"),
                         float((got["dx"].double() - ref["dx"]).norm() / ref["dx"].norm()), bad, flips, edge_flip))
    print("| edges x shape | products | out | dW1 | dW2 | dg1 | dg2 | dX rel. norm | dX entries off (test: < 2e-3) | ReLU flips / activation | edges with a flip |")
    print("|---|---|---:|---:|---:|---:|---:|---:|---:|---:|---:|")
    for r in rows:
        print(f"| {r[0]} | {r[1]} | " + " | ".join(f"{v:.1e}" for v in r[2:]) + " |")


if __name__ == "__main__":
    sys.exit(main())
