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


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[2] == "pmc":
        pmc(sys.argv[1])
    else:
        main(sys.argv[1])
