#!/bin/bash
# Runs ON the GPU box: kernel trace of MinkUNet-14 iterations alone (tools/host_profile.py <voxels> noprof).   tools/trace_unet.sh [voxels]
N=${1:-1000000}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/prof
rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o unet_$N -- python tools/host_profile.py $N noprof > gpurun_out/prof/unet_$N.log 2>&1
python tools/rocpd_stats.py gpurun_out/prof/unet_${N}_results.db > gpurun_out/prof/unet_${N}_kernel_trace_stats.md
rm -f gpurun_out/prof/unet_${N}_results.db
tail -3 gpurun_out/prof/unet_$N.log
head -45 gpurun_out/prof/unet_${N}_kernel_trace_stats.md | cut -c1-90,150-240
