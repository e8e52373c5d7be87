#!/bin/bash
# Runs ON the GPU box (through gpurun): rocprofv3 kernel trace + separate PMC passes of the default bench command.
# Summaries are produced afterwards with tools/rocpd_stats.py from the merged gpurun_out/prof/*.db files.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/prof
CMD="python bench.py --steps 20 --warmup 5 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o r01_trace -- $CMD > gpurun_out/prof/r01_trace.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/prof -o r01_fetch -- $CMD > gpurun_out/prof/r01_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/prof -o r01_write -- $CMD > gpurun_out/prof/r01_write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES -d gpurun_out/prof -o r01_mfma -- $CMD > gpurun_out/prof/r01_mfma.log 2>&1
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum -d gpurun_out/prof -o r01_l2 -- $CMD > gpurun_out/prof/r01_l2.log 2>&1
grep -h metric gpurun_out/prof/r01_trace.log | tail -1
ls -la gpurun_out/prof | grep r01_
