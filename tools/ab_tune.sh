#!/bin/bash
# development helper: what the wave-count tuner sees and decides, against forced builds, on the box it runs on
run() { env $1 timeout -k 10 300 python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline $2 2>gpurun_out/tune_err.log | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('$1', 'Msamples/s', j['value'], 'Gseg/s', j['roofline']['gsegments_per_s'], j['roofline']['kernel'])
"; grep "autotune" gpurun_out/tune_err.log | head -3; grep -c tuning gpurun_out/tune_err.log; }
if [ "$1" = "short" ]; then set -- "" "--scene cornell_roughcond.json --env sky"; else set -- "" "--scene cornell_roughcond.json --env sky" "--scene cornell_roughdiel.json --env sky" "--scene cornell_coat.json" "--scene cornell_media.json --env sky"; fi
for a in "$@"; do
echo "== $a"
run "PRT_LAUNCH_LOG=1" "$a"
run "PRT_AUTOTUNE=0 PRT_WAVES=6" "$a"
run "PRT_AUTOTUNE=0 PRT_WAVES=5" "$a"
done
