#!/bin/bash
# Runs ON the GPU box: rocprofv3 kernel trace of config 5 (tools/bench_pointconv.py), per-kernel summary to
# gpurun_out/trace_pointconv.md
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/prof
python tools/bench_pointconv.py --iters 20 | tail -1
rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o pconv -- python tools/bench_pointconv.py --iters 10 > gpurun_out/prof/pconv.log 2>&1
python tools/rocpd_stats.py gpurun_out/prof/pconv_results.db > gpurun_out/trace_pointconv.md
rm -f gpurun_out/prof/pconv_results.db
tail -1 gpurun_out/prof/pconv.log
head -40 gpurun_out/trace_pointconv.md | cut -c1-200
