#!/bin/bash
# Runs ON the GPU box: per-kernel-instance HBM roofline of MinkUNet-14 forward + backward (tools/bench_minkunet.py): kernel
# trace + separate FETCH_SIZE / WRITE_SIZE / SQ passes -> gpurun_out/unet_<voxels>_roofline.md
V=${1:-1000000}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/prof
CMD="python tools/bench_minkunet.py --voxels $V --iters 6"
rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o ur_t -- $CMD > gpurun_out/prof/ur_t.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/prof -o ur_f -- $CMD > gpurun_out/prof/ur_f.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/prof -o ur_w -- $CMD > gpurun_out/prof/ur_w.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY -d gpurun_out/prof -o ur_s -- $CMD > gpurun_out/prof/ur_s.log 2>&1
python tools/rocpd_stats.py roofline gpurun_out/prof/ur_t_results.db gpurun_out/prof/ur_f_results.db gpurun_out/prof/ur_w_results.db gpurun_out/prof/ur_s_results.db > gpurun_out/unet_${V}_roofline.md
rm -f gpurun_out/prof/ur_*_results.db
head -40 gpurun_out/unet_${V}_roofline.md
