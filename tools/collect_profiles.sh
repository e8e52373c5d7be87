#!/bin/bash
# Runs ON the GPU box (through gpurun): rocprofv3 kernel trace + separate PMC passes of the default bench command
# (counters in their own runs, never combined with runtime / marker traces), summaries into gpurun_out/prof/ as text.
# Usage: bash tools/collect_profiles.sh [round-tag, default r06]
R=${1:-r06}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/prof
CMD="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary"
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o ${R}_trace -- $CMD > gpurun_out/prof/${R}_trace.log 2>&1
grep -h '"metric"' gpurun_out/prof/${R}_trace.log | tail -1 > gpurun_out/prof/${R}_bench_line.json
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/prof -o ${R}_fetch -- $CMD > gpurun_out/prof/${R}_fetch.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/prof -o ${R}_write -- $CMD > gpurun_out/prof/${R}_write.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES -d gpurun_out/prof -o ${R}_mfma -- $CMD > gpurun_out/prof/${R}_mfma.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT -d gpurun_out/prof -o ${R}_sq -- $CMD > gpurun_out/prof/${R}_sq.log 2>&1
# round 5: L2 hit rate of every kernel
# (every pass under its own `timeout`: a counter set the hardware cannot take in one pass must not eat the box's time budget)
# (a TA_* counter pass - TA_TA_BUSY, TA_ADDR_STALLED_BY_TC_CYCLES, TA_DATA_STALLED_BY_TC_CYCLES - never finished on this stack: 50 minutes of box time in round 5; left out)
timeout 300 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum -d gpurun_out/prof -o ${R}_l2 -- $CMD > gpurun_out/prof/${R}_l2.log 2>&1
# secondary workloads (MinkUNet-14 at 200 k / 1 M voxels, PointConv + depthwise): kernel trace of the full default command
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o ${R}_secondary -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/prof/${R}_secondary.log 2>&1
python tools/rocpd_stats.py gpurun_out/prof/${R}_trace_results.db > gpurun_out/prof/${R}_kernel_trace_stats.md
python tools/rocpd_stats.py gpurun_out/prof/${R}_secondary_results.db > gpurun_out/prof/${R}_secondary_kernel_trace_stats.md
python tools/rocpd_stats.py gpurun_out/prof/${R}_fetch_results.db pmc > gpurun_out/prof/${R}_pmc_fetch_size.md
python tools/rocpd_stats.py gpurun_out/prof/${R}_write_results.db pmc > gpurun_out/prof/${R}_pmc_write_size.md
python tools/rocpd_stats.py gpurun_out/prof/${R}_mfma_results.db mfma > gpurun_out/prof/${R}_pmc_mfma_busy.md
python tools/rocpd_stats.py gpurun_out/prof/${R}_sq_results.db sq > gpurun_out/prof/${R}_pmc_wave_cycles.md
[ -f gpurun_out/prof/${R}_l2_results.db ] && python tools/rocpd_stats.py gpurun_out/prof/${R}_l2_results.db pmc > gpurun_out/prof/${R}_pmc_l2.md
python tools/rocpd_stats.py traffic gpurun_out/prof/${R}_fetch_results.db gpurun_out/prof/${R}_write_results.db > gpurun_out/prof/${R}_pmc_traffic.json
# the summaries travel back; of the databases only the kernel trace of the headline command does (the others do not fit the
# 64 MiB return budget)
mv gpurun_out/prof/${R}_trace_results.db gpurun_out/prof/${R}_trace_results.keep
rm -f gpurun_out/prof/*_results.db
mv gpurun_out/prof/${R}_trace_results.keep gpurun_out/prof/${R}_trace_results.db
head -30 gpurun_out/prof/${R}_kernel_trace_stats.md
cat gpurun_out/prof/${R}_bench_line.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['phases_ms'], d['roofline'])"
grep -h "\"metric\"" gpurun_out/prof/${R}_secondary.log | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d.get('secondary'))"
