"""Dev tool: per-step time of the first steps of the headline loop in a fresh process (HIP events around every step, no host
synchronisation in between) - how long the cold start lasts and what it costs.  GPU box only.   python tools/cold_steps.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench

args = bench.parse_args(["--no-secondary", "--no-cpu-baseline"])
dev = torch.device("cuda:0")
coords, feats, grad_out, offsets, conv, params = bench.build_workload(args, dev, 0)
step, _ = bench.make_step(dev, 1, coords, feats, grad_out, offsets, conv, params)
n = 60
ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
import time
host = []
ev[0].record()
for i in range(n):
    t = time.perf_counter()
    step()
    host.append((time.perf_counter() - t) * 1e3)
    ev[i + 1].record()
torch.cuda.synchronize()
ms = [ev[i].elapsed_time(ev[i + 1]) for i in range(n)]
for i in range(0, n, 10):
    print("steps %2d-%2d  gpu ms: %s" % (i, i + 9, " ".join(f"{v:6.3f}" for v in ms[i:i + 10])))
    print("             host ms: %s" % " ".join(f"{v:6.3f}" for v in host[i:i + 10]))
time.sleep(2.0)
ev2 = [torch.cuda.Event(enable_timing=True) for _ in range(21)]
ev2[0].record()
for i in range(20):
    step()
    ev2[i + 1].record()
torch.cuda.synchronize()
print("after 2 s idle: %s" % " ".join(f"{ev2[i].elapsed_time(ev2[i + 1]):6.3f}" for i in range(20)))
# a burst of unrelated GPU work first (clocks up), then steps
a = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16)
time.sleep(2.0)
for _ in range(40):
    a @ a
ev3 = [torch.cuda.Event(enable_timing=True) for _ in range(11)]
ev3[0].record()
for i in range(10):
    step()
    ev3[i + 1].record()
torch.cuda.synchronize()
print("after idle + 40 dense GEMMs: %s" % " ".join(f"{ev3[i].elapsed_time(ev3[i + 1]):6.3f}" for i in range(10)))
print("allocated MB", torch.cuda.memory_allocated() / 1e6, "reserved MB", torch.cuda.memory_reserved() / 1e6)
