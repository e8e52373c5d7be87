#!/bin/bash
# Runs ON the GPU box: rocprofv3 kernel trace of MinkUNet-14 forward + backward (tools/bench_minkunet.py) at the given size,
# per-kernel summary to gpurun_out/trace_unet_<voxels>.md
V=${1:-1000000}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/prof
rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o unet_${V} -- python tools/bench_minkunet.py --voxels $V --iters 10 > gpurun_out/prof/unet_${V}.log 2>&1
python tools/rocpd_stats.py gpurun_out/prof/unet_${V}_results.db > gpurun_out/trace_unet_${V}.md
python tools/rocpd_stats.py gpurun_out/prof/unet_${V}_results.db gaps > gpurun_out/gaps_unet_${V}.md
rm -f gpurun_out/prof/unet_${V}_results.db
tail -2 gpurun_out/prof/unet_${V}.log
head -${2:-48} gpurun_out/trace_unet_${V}.md | cut -c1-170
head -40 gpurun_out/gaps_unet_${V}.md | cut -c1-150
