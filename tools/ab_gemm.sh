#!/bin/bash
# A/B on one box: parity tests for the gather GEMMs, then bench with the gather-GEMM variants given as arguments.
# usage: tools/ab_gemm.sh "<test mode>" "<modes to bench>" [scenes]
TM=${1:-1}; MODES=${2:-"1 0"}; SCENES=${3:-"uniform surface"}
mkdir -p gpurun_out/ab
if [ "$TM" != "none" ]; then
WARPCONVNET_AMD_GEMM_CS=$TM python -m pytest tests/test_gpu_conv.py tests/test_gpu_fused_epilogue.py tests/test_gpu_minkunet.py -x -q -m gpu 2>&1 | tail -15 > gpurun_out/ab/tests.log
cat gpurun_out/ab/tests.log
fi
for scene in $SCENES; do
for cs in $MODES; do
  f=gpurun_out/ab/bench_${scene}_cs$cs.json
  WARPCONVNET_AMD_GEMM_CS=$cs python bench.py --steps 40 --warmup 5 --no-secondary --no-cpu-baseline --scene $scene > $f 2> gpurun_out/ab/bench_${scene}_cs$cs.err
  python - "$f" "$scene cs=$cs" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    ra=d.get('roofline_all',{})
    ks=list(ra.values())
    print(sys.argv[2], 'Mvox/s', d['value'], 'ms', d['ms_per_step'], 'phases', d.get('phases_ms'), 'isolated fwd/dgrad', ks[0].get('isolated_ms'), ks[1].get('isolated_ms'))
except Exception as e: print('ERR', e, open(sys.argv[1]).read()[-500:], open(sys.argv[1].replace('.json','.err')).read()[-1500:])
PY
done; done
