#!/bin/bash
# dev: A/B of two builds of the library on one box: tools/ab_lib.sh <base.so> [scenes] [rounds]  (the other build is libwcn_hip.so)
BASE=$1; SCENES=${2:-"uniform surface"}; ROUNDS=${3:-2}
mkdir -p gpurun_out/ab
for scene in $SCENES; do for r in $(seq $ROUNDS); do for which in base new; do
  f=gpurun_out/ab/lib_${scene}_${which}_$r.json
  if [ $which = base ]; then export WARPCONVNET_AMD_LIB=$BASE; else unset WARPCONVNET_AMD_LIB; fi
  python bench.py --steps 60 --warmup 5 --no-secondary --no-cpu-baseline --scene $scene > $f 2> ${f%.json}.err
  python - "$f" "$scene $which" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    ks=list(d.get('roofline_all',{}).values())
    print(sys.argv[2], 'Mvox/s', d['value'], 'phases', d.get('phases_ms'), 'isolated fwd/dgrad', ks[0].get('isolated_ms'), ks[1].get('isolated_ms'))
except Exception as e: print('ERR', e, open(sys.argv[1].replace('.json','.err')).read()[-800:])
PY
done; done; done
