#!/bin/bash
# Runs ON the GPU box: kernel traces of the headline bench under the A/B switches of the round-6 kernel-map work
#   WARPCONVNET_AMD_KMAP_COMPACT=0|1 (compact table rows), WARPCONVNET_AMD_KMAP_COSCHED="0" | "s,h,c" (pair scatter inside the sort)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/prof
run() {
  TAG=$1; shift
  env "$@" rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o t_${TAG} -- python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-secondary > gpurun_out/prof/t_${TAG}.log 2>&1
  python tools/rocpd_stats.py gpurun_out/prof/t_${TAG}_results.db > gpurun_out/trace_${TAG}.md
  rm -f gpurun_out/prof/t_${TAG}_results.db
  echo "== ${TAG}: $*"
  head -32 gpurun_out/trace_${TAG}.md
  grep -h '"metric"' gpurun_out/prof/t_${TAG}.log | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['phases_ms'])"
}
for spec in "$@"; do
  TAG=${spec%%:*}
  ENVS=${spec#*:}
  run "$TAG" $(echo "$ENVS" | tr ';' ' ')
done
