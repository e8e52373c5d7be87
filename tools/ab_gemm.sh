#!/bin/bash
# A/B on one box: parity tests for the gather GEMMs, then bench with the channel-split family on / off.
mkdir -p gpurun_out/ab
python -m pytest tests/test_gpu_conv.py tests/test_gpu_fused_epilogue.py tests/test_gpu_minkunet.py -x -q -m gpu 2>&1 | tail -15 > gpurun_out/ab/tests.log
for scene in uniform surface; do
for cs in 1 0; do
  WARPCONVNET_AMD_GEMM_CS=$cs python bench.py --steps 40 --warmup 5 --no-secondary --no-cpu-baseline --scene $scene > gpurun_out/ab/bench_${scene}_cs$cs.json 2> gpurun_out/ab/bench_${scene}_cs$cs.err
done; done
cat gpurun_out/ab/tests.log
for f in gpurun_out/ab/bench_*.json; do echo $f; python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(d['value'], d['ms_per_step'], d.get('phases_ms'))
    ra=d.get('roofline_all',{})
    for k,v in ra.items(): print('  ',k, {kk:(round(vv,4) if isinstance(vv,float) else vv) for kk,vv in v.items()} if isinstance(v,dict) else v)
except Exception as e: print('ERR', e, open(sys.argv[1]).read()[-500:])
PY
done
