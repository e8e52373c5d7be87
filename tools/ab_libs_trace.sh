#!/bin/bash
# Runs ON the GPU box: kernel traces of the headline bench for several builds of the library on ONE box.
#   tools/ab_libs_trace.sh <grep pattern> <rounds> lib1.so lib2.so ...      ("-" = the shipped build)
PAT=$1; ROUNDS=$2; shift 2
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/prof
for r in $(seq $ROUNDS); do for lib in "$@"; do
  tag=$(basename $lib .so); [ "$lib" = "-" ] && tag=shipped
  if [ "$lib" = "-" ]; then unset WARPCONVNET_AMD_LIB; else export WARPCONVNET_AMD_LIB=$PWD/warpconvnet_amd/csrc/$lib; fi
  rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o abt_$tag -- python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-secondary > gpurun_out/prof/abt_$tag.log 2>&1
  echo "== $tag (round $r)"
  python tools/rocpd_stats.py gpurun_out/prof/abt_${tag}_results.db | grep -E "$PAT" | sed 's/(.*)` /` /' | cut -c1-150
  grep -h '"metric"' gpurun_out/prof/abt_$tag.log | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['phases_ms'])"
  rm -f gpurun_out/prof/abt_${tag}_results.db
done; done
