#!/bin/bash
# dev: bench several builds of the library on one box: tools/ab_libs.sh "<scenes>" <rounds> lib1.so lib2.so ...  ("-" = the shipped build)
SCENES=$1; ROUNDS=$2; shift 2
mkdir -p gpurun_out/ab
for scene in $SCENES; do for r in $(seq $ROUNDS); do for lib in "$@"; do
  tag=$(basename $lib .so)
  f=gpurun_out/ab/libs_${scene}_${tag}_$r.json
  if [ "$lib" = "-" ]; then unset WARPCONVNET_AMD_LIB; else export WARPCONVNET_AMD_LIB=$PWD/warpconvnet_amd/csrc/$lib; fi
  python bench.py --steps 60 --warmup 5 --no-secondary --no-cpu-baseline --scene $scene > $f 2> ${f%.json}.err
  python - "$f" "$scene $tag" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    ks=list(d.get('roofline_all',{}).values())
    print(sys.argv[2], 'Mvox/s', d['value'], 'phases', d.get('phases_ms'), 'isolated fwd/dgrad', ks[0].get('isolated_ms'), ks[1].get('isolated_ms'))
except Exception as e: print('ERR', e, open(sys.argv[1].replace('.json','.err')).read()[-800:])
PY
done; done; done
