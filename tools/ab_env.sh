#!/bin/bash
# A/B helper (development only): same build, different environment settings: tools_ab_env.sh "VAR=val VAR2=val" ...
for v in "$@"; do
  env $v timeout -k 10 200 python3 bench.py --spp ${SPP:-32} --steps ${STEPS:-2} --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('$v', 'Msamples/s', j['value'], 'Gseg/s', j['roofline']['gsegments_per_s'], 'avg_launch_ms', j['roofline']['avg_launch_ms'])
"
done
