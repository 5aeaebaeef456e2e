#!/bin/bash
# development helper: launch structure x wave-count build on the headline config, to tell the two kinds of box apart
run() { env $1 timeout -k 10 300 python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); r=j['roofline']; print('$1', 'Msamples/s', j['value'], 'avg_launch_ms', r['avg_launch_ms'], 'launches', r['launches'], 'wall', r['kernel_wall_ms'], r['kernel'])
"; }
for s in "PRT_STREAMS=2 PRT_WAVES=6" "PRT_STREAMS=2 PRT_WAVES=5" "PRT_STREAMS=1 PRT_WAVES=6" "PRT_STREAMS=3 PRT_WAVES=6" "PRT_WAVES=6 PRT_TILE_ORDER=0"; do run "$s"; done
