#!/bin/bash
# dragon stand-in at 1920x1080, few spp, under different environment settings
for v in "$@"; do
  env $v timeout -k 10 300 python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --scene cornell_dragon.json --spp ${SPP:-8} 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('$v', 'Msamples/s', j['value'], 'Gseg/s', j['roofline']['gsegments_per_s'])
"
done
