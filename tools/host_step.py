"""Dev: host time of the headline bench step (build_workload / make_step of bench.py) - enqueue time per step and cProfile."""
import argparse, cProfile, pstats, sys, time
sys.path.insert(0, ".")
import torch
import bench

args = bench.parse_args(["--no-secondary", "--no-cpu-baseline"])
dev = torch.device("cuda:0")
coords, feats, grad_out, offsets, conv, params = bench.build_workload(args, dev, 0)
step, _ = bench.make_step(dev, 1, coords, feats, grad_out, offsets, conv, params)
for _ in range(30): step()
torch.cuda.synchronize()
t = time.perf_counter()
for i in range(200):
    step()
host = (time.perf_counter() - t) / 200
torch.cuda.synchronize()
wall = (time.perf_counter() - t) / 200
print(f"step: host enqueue {host*1e3:.3f} ms, wall {wall*1e3:.3f} ms")
# host time alone: a sync after every step keeps the queue empty (no back-pressure)
t = time.perf_counter(); h = 0.0
for i in range(100):
    t0 = time.perf_counter(); step(); h += time.perf_counter() - t0
    torch.cuda.synchronize()
print(f"step: host time with an empty queue {h/100*1e3:.3f} ms")
pr = cProfile.Profile()
with torch.autograd.set_multithreading_enabled(False):
    pr.enable()
    for i in range(100):
        step()
        if i % 20 == 19: torch.cuda.synchronize()
    pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(38)
