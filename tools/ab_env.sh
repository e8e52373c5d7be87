#!/bin/bash
# dev: bench under several environment settings on one box.  usage: tools/ab_env.sh "<scenes>" "ENV1=a ENV2=b" "ENV1=c" ...
SCENES=$1; shift
mkdir -p gpurun_out/ab
for scene in $SCENES; do
for envs in "$@"; do
  tag=$(echo "$envs" | tr ' =/.' '____' | tail -c 60)
  f=gpurun_out/ab/bench_${scene}_$tag.json
  env $envs python bench.py --steps 40 --warmup 5 --no-secondary --no-cpu-baseline --scene $scene > $f 2> ${f%.json}.err
  python - "$f" "$scene [$envs]" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    ks=list(d.get('roofline_all',{}).values())
    print(sys.argv[2], 'Mvox/s', d['value'], 'phases', d.get('phases_ms'), 'isolated fwd/dgrad', ks[0].get('isolated_ms'), ks[1].get('isolated_ms'))
except Exception as e: print('ERR', e, open(sys.argv[1]).read()[-300:], open(sys.argv[1].replace('.json','.err')).read()[-1200:])
PY
done; done
