#!/bin/bash
# Runs ON the GPU box: kernel traces of the headline bench under several settings of ONE environment variable, on one box.
#   tools/ab_env_trace.sh <grep pattern> <rounds> VAR v1 v2 ...
PAT=$1; ROUNDS=$2; VAR=$3; shift 3
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/prof
for r in $(seq $ROUNDS); do for v in "$@"; do
  export $VAR=$v
  rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o abe_$v -- python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-secondary > gpurun_out/prof/abe_$v.log 2>&1
  echo "== $VAR=$v (round $r)"
  python tools/rocpd_stats.py gpurun_out/prof/abe_${v}_results.db | grep -E "$PAT" | sed 's/(.*)` /` /' | cut -c1-150
  grep -h '"metric"' gpurun_out/prof/abe_$v.log | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['phases_ms'])"
  rm -f gpurun_out/prof/abe_${v}_results.db
done; done
