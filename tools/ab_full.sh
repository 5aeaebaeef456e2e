#!/bin/bash
# development helper: the shipped defaults against forced settings on the full headline config, same box, alternating
run() { env $1 timeout -k 10 300 python3 bench.py --steps $2 --warmup $3 --no-cpu-baseline $4 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('$1 steps $2 warmup $3', 'Msamples/s', j['value'], 'Gseg/s', j['roofline']['gsegments_per_s'], j['roofline']['kernel'])
"; }
for rep in 1 2; do
for s in "PRT_WAVES=6 PRT_WALK_MIN_LANES=8" "PRT_WAVES=6 PRT_WALK_MIN_LANES=12" "PRT_WAVES=5 PRT_WALK_MIN_LANES=8" "PRT_WAVES=5 PRT_WALK_MIN_LANES=12"; do
run "$s" 2 1 "$@"
done; done
