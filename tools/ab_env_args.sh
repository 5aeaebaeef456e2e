#!/bin/bash
# A/B helper (development only): same build, same bench arguments, different environment settings
#   tools/ab_env_args.sh "<bench args>" "VAR=val VAR2=val" ...
args=$1; shift
for v in "$@"; do
  env $v timeout -k 10 400 python3 bench.py --steps ${STEPS:-1} --warmup 1 --no-cpu-baseline $args 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('$v', 'Msamples/s', j['value'], 'Gseg/s', j['roofline']['gsegments_per_s'], 'avg_launch_ms', j['roofline']['avg_launch_ms'], j['roofline']['kernel'])
"
done
