#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd SQLite database (``--kernel-trace --stats``) as a per-kernel table (dev tool).

    python tools/rocpd_stats.py gpurun_out/prof/r1_results.db > profiles/r01_kernel_stats.md
"""
import re
import sqlite3
import sys


def short(name: str) -> str:
    name = re.sub(r"\[clone .*?\]", "", name)
    name = name.replace("void ", "")
    return name if len(name) <= 110 else name[:107] + "..."


def main(path):
    db = sqlite3.connect(path)
    cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
    rows = db.execute("select name, (end - start) from kernels").fetchall()
    agg = {}
    for name, dur in rows:
        a = agg.setdefault(name, [0, 0, 1 << 62, 0])
        a[0] += 1
        a[1] += dur
        a[2] = min(a[2], dur)
        a[3] = max(a[3], dur)
    total = sum(a[1] for a in agg.values())
    print(f"# rocprofv3 kernel-trace summary ({path})\n")
    print(f"total kernel time {total / 1e6:.3f} ms over {len(rows)} dispatches\n")
    print("| kernel | calls | total ms | avg us | min us | max us | % |")
    print("|---|---:|---:|---:|---:|---:|---:|")
    for name, (n, t, lo, hi) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"| `{short(name)}` | {n} | {t / 1e6:.3f} | {t / n / 1e3:.1f} | {lo / 1e3:.1f} | {hi / 1e3:.1f} | {100 * t / total:.1f} |")




def gaps(path, tail_frac=0.5, top=40):
    """Idle time of the GPU between consecutive kernels over the last ``tail_frac`` of the trace (steady state): total, and the
    largest gaps with the kernels on either side - where a host-bound stretch or a host read of device state stalls the queue."""
    db = sqlite3.connect(path)
    rows = db.execute("select name, start, end from kernels order by start").fetchall()
    rows = rows[int(len(rows) * (1 - tail_frac)):]
    busy = sum(e - s for _, s, e in rows)
    span = rows[-1][2] - rows[0][1]
    print(f"# GPU idle gaps ({path}; last {len(rows)} dispatches)\n")
    print(f"span {span / 1e6:.3f} ms, kernels {busy / 1e6:.3f} ms, idle {(span - busy) / 1e6:.3f} ms ({100 * (span - busy) / span:.1f} %)\n")
    g = []
    last_end = rows[0][2]
    for i in range(1, len(rows)):
        gap = rows[i][1] - last_end
        if gap > 0:
            g.append((gap, i))
        last_end = max(last_end, rows[i][2])
    hist = {}
    for gap, i in g:
        b = "<5us" if gap < 5e3 else "5-20us" if gap < 2e4 else "20-100us" if gap < 1e5 else ">100us"
        h = hist.setdefault(b, [0, 0])
        h[0] += 1
        h[1] += gap
    for b in ("<5us", "5-20us", "20-100us", ">100us"):
        if b in hist:
            print(f"- gaps {b}: {hist[b][0]} totalling {hist[b][1] / 1e6:.3f} ms")
    after = {}
    for gap, i in g:
        a = after.setdefault(short(rows[i][0])[:90], [0, 0])
        a[0] += 1
        a[1] += gap
    print("\n| kernel AFTER the gap | gaps | total idle ms | avg us |")
    print("|---|---:|---:|---:|")
    for k, (n, t) in sorted(after.items(), key=lambda kv: -kv[1][1])[:top]:
        print(f"| `{k}` | {n} | {t / 1e6:.3f} | {t / n / 1e3:.1f} |")


def pmc(path):
    """Per-kernel average of every collected counter (``--pmc`` runs)."""
    db = sqlite3.connect(path)
    rows = db.execute("select kernel_name, counter_name, value from counters_collection").fetchall()
    agg = {}
    for k, c, v in rows:
        a = agg.setdefault((k, c), [0, 0.0])
        a[0] += 1
        a[1] += v
    print(f"# rocprofv3 PMC summary ({path})\n")
    print("| kernel | counter | dispatches | avg value (KB) | avg MB |")
    print("|---|---|---:|---:|---:|")
    for (k, c), (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        if t / n < 64:
            continue
        print(f"| `{short(k)}` | {c} | {n} | {t / n:.1f} | {t / n / 1024:.2f} |")


def traffic(fetch_db, write_db):
    """HBM bytes per launch of the three GEMM kernels from two separate --pmc passes (FETCH_SIZE / WRITE_SIZE, both in
    KiB).  FETCH_SIZE is doubled: on gfx950 it tallies 128-B fabric requests at 64 B (MI355X_MICROARCH.md, HBM section);
    WRITE_SIZE matched the known output size exactly on this workload (250000 KiB for 1 M x 128 bf16) and is used as is."""
    import json

    def avg(db, counter):
        rows = sqlite3.connect(db).execute(
            "select kernel_name, value from counters_collection where counter_name = ?", (counter,)).fetchall()
        agg = {}
        for k, v in rows:
            a = agg.setdefault(k, [0, 0.0])
            a[0] += 1
            a[1] += v
        return {k: t / n for k, (n, t) in agg.items()}

    f, w = avg(fetch_db, "FETCH_SIZE"), avg(write_db, "WRITE_SIZE")
    # (round 3: the 64 -> 128 forward and the 128 -> 64 dgrad run the channel-split kernels, template <T, CO, RBW, WR, MINW>)
    keys = {"fwd": "gather_gemm_cs_kernelIDF16bLi128E", "dgrad": "gather_gemm_cs_kernelIDF16bLi64E",
            "wgrad": "wgrad_mfma_kernelIDF16bLi64ELi128ELb1"}
    out = {"unit": "bytes per launch", "fetch_correction": 2.0, "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes of `python bench.py --steps 20 --warmup 5 --no-cpu-baseline`"}
    for name, sub in keys.items():
        fk = [v for k, v in f.items() if sub in k]
        wk = [v for k, v in w.items() if sub in k]
        fetch = 2.0 * 1024 * (fk[0] if fk else 0.0)
        write = 1024 * (wk[0] if wk else 0.0)
        out[name] = {"fetch_bytes": int(fetch), "write_bytes": int(write), "hbm_bytes": int(fetch + write)}
    # kernel-map build: per BUILD = sum over its kernels of (average per dispatch x dispatches per build)
    def per_build(db, counter, table):
        rows = sqlite3.connect(db).execute(
            "select kernel_name, value from counters_collection where counter_name = ?", (counter,)).fetchall()
        per_kernel = {}
        for k, v in rows:
            if any(t in k for t in ("cell_", "kmap_", "rs_kernel", "rs_pairs")):
                a = per_kernel.setdefault(k, [0, 0.0])
                a[0] += 1
                a[1] += v
        if not per_kernel:
            return 0.0, {}
        builds = min(n for k, (n, t) in per_kernel.items() if "cell_prepare" in k) if any("cell_prepare" in k for k in per_kernel) else 1
        detail = {short(k): round(t / builds, 1) for k, (n, t) in per_kernel.items()}
        return sum(t for n, t in per_kernel.values()) / builds, detail
    kf, kfd = per_build(fetch_db, "FETCH_SIZE", f)
    kw, kwd = per_build(write_db, "WRITE_SIZE", w)
    out["kmap"] = {"fetch_bytes": int(2.0 * 1024 * kf), "write_bytes": int(1024 * kw), "hbm_bytes": int(2.0 * 1024 * kf + 1024 * kw),
                   "fetch_kib_per_build_by_kernel": kfd, "write_kib_per_build_by_kernel": kwd}
    # stamp: the kernels these counters were collected on (bench.py reports the figures only while the library still has them)
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from warpconvnet_amd.utils.codesig import phase_signatures
    from warpconvnet_amd import _lib
    out["kernel_signatures"] = phase_signatures(_lib.LIB_PATH)
    print(json.dumps(out, indent=1))


def roofline(trace_db, fetch_db, write_db, sq_db=None):
    """Per kernel INSTANCE (template instantiation) of a whole workload: time from the kernel trace, HBM bytes from two separate
    --pmc passes (FETCH_SIZE x 2 on gfx950, WRITE_SIZE; KiB), achieved HBM GB/s = bytes / time, fraction of 8 TB/s, and -
    with an SQ pass - the share of wave-cycles parked in s_waitcnt / barriers."""
    def per_kernel(path, counter):
        db = sqlite3.connect(path)
        agg = {}
        for k, v in db.execute("select kernel_name, value from counters_collection where counter_name = ?", (counter,)):
            a = agg.setdefault(k, [0, 0.0])
            a[0] += 1
            a[1] += v
        return agg
    t = sqlite3.connect(trace_db)
    times = {}
    for name, dur in t.execute("select name, (end - start) from kernels"):
        a = times.setdefault(name, [0, 0, 0])
        a[0] += 1
        a[1] += dur
        a[2] = max(a[2], dur)
    fetch, write = per_kernel(fetch_db, "FETCH_SIZE"), per_kernel(write_db, "WRITE_SIZE")
    parked = {}
    if sq_db:
        wc, wa = per_kernel(sq_db, "SQ_WAVE_CYCLES"), per_kernel(sq_db, "SQ_WAIT_ANY")
        parked = {k: wa[k][1] / wc[k][1] for k in wc if k in wa and wc[k][1] > 0}
    total = sum(a[1] for a in times.values())
    print(f"# per-kernel-instance HBM roofline ({trace_db}; FETCH_SIZE x 2 + WRITE_SIZE from separate PMC passes)\n")
    print(f"total kernel time {total / 1e6:.3f} ms over {sum(a[0] for a in times.values())} dispatches\n")
    print("| kernel instance | calls | total ms | % of kernels | avg us | max us | HBM MB / call | GB/s | frac of 8 TB/s | parked |")
    print("|---|---:|---:|---:|---:|---:|---:|---:|---:|---:|")
    for name, (n, tt, hi) in sorted(times.items(), key=lambda kv: -kv[1][1]):
        if tt / total < 0.004:
            continue
        f, w = fetch.get(name), write.get(name)
        kib = (2.0 * f[1] / f[0] if f else 0.0) + (w[1] / w[0] if w else 0.0)
        gbs = kib * 1024 / (tt / n) if tt else 0.0  # bytes per ns = GB/s
        pk = f"{parked[name]:.2f}" if name in parked else ""
        print(f"| `{short(name)}` | {n} | {tt / 1e6:.3f} | {100 * tt / total:.1f} | {tt / n / 1e3:.1f} | {hi / 1e3:.1f} | "
              f"{kib / 1024:.1f} | {gbs:.0f} | {gbs / 8000:.3f} | {pk} |")


def mfma(db):
    """Matrix-core utilisation per kernel from a `--pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES` pass:
    MFMA busy cycles (summed over SIMDs) / (4 SIMDs x CU busy cycles)."""
    rows = sqlite3.connect(db).execute("select kernel_name, counter_name, value from counters_collection").fetchall()
    agg = {}
    for k, c, v in rows:
        a = agg.setdefault((k, c), [0, 0.0])
        a[0] += 1
        a[1] += v
    avg = {kc: t / n for kc, (n, t) in agg.items()}
    names = sorted({k for k, _ in avg}, key=lambda k: -avg.get((k, "SQ_BUSY_CU_CYCLES"), 0))
    print(f"# rocprofv3 PMC summary: matrix-core utilisation ({db})\n")
    print("`SQ_VALU_MFMA_BUSY_CYCLES` is summed over the SIMDs (32 cycles per `v_mfma_f32_32x32x16_bf16`), `SQ_BUSY_CU_CYCLES` over")
    print("the CUs; MFMA utilisation = MFMA busy cycles / (4 SIMDs x CU busy cycles).\n")
    print("| kernel | dispatches | MFMA busy cycles (avg) | CU busy cycles (avg) | MFMA utilisation |\n|---|---:|---:|---:|---:|")
    for k in names[:14]:
        m, b = avg.get((k, "SQ_VALU_MFMA_BUSY_CYCLES"), 0.0), avg.get((k, "SQ_BUSY_CU_CYCLES"), 0.0)
        if b > 0:
            name = k if len(k) < 110 else k[:107] + "..."
            print(f"| `{name}` | {agg[(k, 'SQ_BUSY_CU_CYCLES')][0]} | {m:,.0f} | {b:,.0f} | {m / (4 * b):.3f} |")


def sq(db):
    """Where the wave-cycles go, per kernel, from a `--pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
    SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT` pass (values in quad-cycles, summed over all waves of a dispatch)."""
    rows = sqlite3.connect(db).execute("select kernel_name, counter_name, value from counters_collection").fetchall()
    agg = {}
    for k, c, v in rows:
        a = agg.setdefault((k, c), [0, 0.0])
        a[0] += 1
        a[1] += v
    avg = {kc: t / n for kc, (n, t) in agg.items()}
    names = sorted({k for k, _ in avg}, key=lambda k: -avg.get((k, "SQ_WAVE_CYCLES"), 0) * agg.get((k, "SQ_WAVE_CYCLES"), [0])[0])
    print(f"# rocprofv3 PMC summary: wave-cycle breakdown ({db})\n")
    print("Per dispatch, quad-cycles summed over all waves.  parked = `SQ_WAIT_ANY` (waves in `s_waitcnt` / `s_barrier`), issue stall =")
    print("`SQ_WAIT_INST_ANY` (an instruction is ready but cannot issue), executing = `SQ_ACTIVE_INST_ANY`; LDS bank conflict cycles on the right.\n")
    print("| kernel | dispatches | wave quad-cycles | parked | issue stall | executing | of which LDS | LDS bank conflicts |\n|---|---:|---:|---:|---:|---:|---:|---:|")
    for k in names[:16]:
        wv = avg.get((k, "SQ_WAVE_CYCLES"), 0.0)
        if wv <= 0:
            continue
        f = lambda c: avg.get((k, c), 0.0) / wv
        name = k if len(k) < 100 else k[:97] + "..."
        print(f"| `{name}` | {agg[(k, 'SQ_WAVE_CYCLES')][0]} | {wv:,.0f} | {f('SQ_WAIT_ANY'):.2f} | {f('SQ_WAIT_INST_ANY'):.2f} | "
              f"{f('SQ_ACTIVE_INST_ANY'):.2f} | {f('SQ_ACTIVE_INST_LDS'):.2f} | {f('SQ_LDS_BANK_CONFLICT'):.2f} |")


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[2] == "sq":
        sq(sys.argv[1])
        sys.exit(0)
    if len(sys.argv) > 2 and sys.argv[2] == "mfma":
        mfma(sys.argv[1])
        sys.exit(0)
    if len(sys.argv) > 4 and sys.argv[1] == "roofline":
        roofline(*sys.argv[2:6])
        sys.exit(0)
    if len(sys.argv) > 3 and sys.argv[1] == "traffic":
        traffic(sys.argv[2], sys.argv[3])
        sys.exit(0)
    if len(sys.argv) > 2 and sys.argv[2] == "gaps":
        gaps(sys.argv[1])
        sys.exit(0)
    if len(sys.argv) > 2 and sys.argv[2] == "pmc":
        pmc(sys.argv[1])
    else:
        main(sys.argv[1])
