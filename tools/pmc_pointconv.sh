#!/bin/bash
# Runs ON the GPU box: PMC passes over config 5 (tools/bench_pointconv.py): instruction mix, wave cycles, MFMA busy, HBM bytes
# of the PointConv kernels -> gpurun_out/pmc_pointconv_*.md
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/prof
CMD="python tools/bench_pointconv.py --iters 3"
run() { n=$1; shift; rocprofv3 --kernel-trace --pmc "$@" -d gpurun_out/prof -o pc_$n -- $CMD > gpurun_out/prof/pc_$n.log 2>&1; }
run insts SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVES
run sq SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT
run mfma SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA
run fetch FETCH_SIZE
run write WRITE_SIZE
for n in insts sq mfma fetch write; do python tools/rocpd_stats.py gpurun_out/prof/pc_${n}_results.db pmc > gpurun_out/pmc_pointconv_$n.md 2>&1; rm -f gpurun_out/prof/pc_${n}_results.db; done
for n in insts sq mfma fetch write; do echo "== $n"; grep -i "pointconv\|^| kernel\|^|---" gpurun_out/pmc_pointconv_$n.md | cut -c1-60,120-400 | head -8; done
