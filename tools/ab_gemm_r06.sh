#!/bin/bash
# Runs ON the GPU box: the round-6 variants of the channel-split gather GEMM (tools/build_abl.sh conv_mfma_cs.hip pair:"-DWCN_CS_PAIR"
# wfirst:"-DWCN_CS_WFIRST") against the shipped build: parity, in-step times (two rounds), and SQ / MFMA counters per variant.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/prof
[ -n "$COUNTERS_ONLY" ] || for lib in libwcn_abl_pair.so libwcn_abl_wfirst.so; do
  echo "== parity $lib"; WARPCONVNET_AMD_LIB=$PWD/warpconvnet_amd/csrc/$lib python -m pytest tests/test_gpu_conv.py -x -q -k "mfma or oracle or golden or compact" 2>&1 | tail -2
done
[ -n "$COUNTERS_ONLY" ] || bash tools/ab_libs.sh "uniform surface" 2 - libwcn_abl_pair.so libwcn_abl_wfirst.so
CMD="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary"
for lib in - libwcn_abl_pair.so libwcn_abl_wfirst.so; do
  tag=$(basename $lib .so); [ "$lib" = "-" ] && tag=shipped
  if [ "$lib" = "-" ]; then unset WARPCONVNET_AMD_LIB; else export WARPCONVNET_AMD_LIB=$PWD/warpconvnet_amd/csrc/$lib; fi
  rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT -d gpurun_out/prof -o g_sq_$tag -- $CMD > gpurun_out/prof/g_sq_$tag.log 2>&1
  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES -d gpurun_out/prof -o g_mf_$tag -- $CMD > gpurun_out/prof/g_mf_$tag.log 2>&1
  echo "== counters $tag"
  python tools/rocpd_stats.py gpurun_out/prof/g_sq_${tag}_results.db sq | grep -i "gather_gemm_cs\|^| kernel"
  python tools/rocpd_stats.py gpurun_out/prof/g_mf_${tag}_results.db mfma | grep -i "gather_gemm_cs\|^| kernel"
  rm -f gpurun_out/prof/g_*_results.db
done
