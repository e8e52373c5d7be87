#!/bin/bash
# Runs ON the GPU box: rocprofv3 kernel trace of a short headline bench, per-kernel summary to gpurun_out/trace_<tag>.md
TAG=${1:-x}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/prof
rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o t_${TAG} -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > gpurun_out/prof/t_${TAG}.log 2>&1
python tools/rocpd_stats.py gpurun_out/prof/t_${TAG}_results.db > gpurun_out/trace_${TAG}.md
rm -f gpurun_out/prof/t_${TAG}_results.db
head -40 gpurun_out/trace_${TAG}.md
grep -h '"metric"' gpurun_out/prof/t_${TAG}.log | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['phases_ms'])"
