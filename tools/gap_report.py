"""Dev tool: GPU idle gaps of a kernel trace (rocprofv3 --kernel-trace --output-format csv).  For a host-bound network
iteration it says between which kernels the device waits for the host, and for how long in total.

    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/gap -- python tools/host_profile.py 200000 noprof
    python tools/gap_report.py gpurun_out/gap [skip_fraction]
"""
import csv
import glob
import os
import sys
from collections import defaultdict


def short(name, n=60):
    name = name.replace("void ", "").replace("wcn::", "").replace("(anonymous namespace)::", "")
    return name[:n]


def main():
    root = sys.argv[1]
    skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
    files = glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True)
    if not files:
        sys.exit(f"no *kernel_trace.csv under {root}")
    rows = []
    with open(files[0]) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    rows = rows[int(len(rows) * skip):]
    span = rows[-1][1] - rows[0][0]
    busy = sum(e - s for s, e, _ in rows)
    gaps = defaultdict(lambda: [0, 0])
    hist = defaultdict(lambda: [0, 0])
    end = rows[0][1]
    prev = rows[0][2]
    for s, e, name in rows[1:]:
        g = s - end
        if g > 0:
            b = 1 if g < 2000 else 2 if g < 5000 else 5 if g < 10000 else 10 if g < 30000 else 30 if g < 100000 else 100
            hist[b][0] += 1
            hist[b][1] += g
            if g >= 5000:
                k = (short(prev, 44), short(name, 44))
                gaps[k][0] += 1
                gaps[k][1] += g
        if e > end:
            end, prev = e, name
    print(f"{len(rows)} kernels over {span / 1e6:.2f} ms: busy {busy / 1e6:.2f} ms ({busy / span:.3f}), idle {(span - busy) / 1e6:.2f} ms")
    print("gap size (us, lower edge) : count, total ms")
    for b in sorted(hist):
        print(f"  >= {b:4d} : {hist[b][0]:6d}  {hist[b][1] / 1e6:8.3f}")
    print("gaps >= 5 us by (kernel before -> kernel after): count, total us, mean us")
    for k, (c, t) in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:40]:
        print(f"  {c:5d} {t / 1e3:9.1f} {t / 1e3 / c:7.1f}  {k[0]} -> {k[1]}")


if __name__ == "__main__":
    main()
