#!/bin/bash
# runs the BASELINE.json configs 1-5 on one GPU at reduced spp (development helper; numbers go to BASELINE.md)
run() { name=$1; shift; timeout -k 10 300 python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline "$@" 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('$name', 'Msamples/s', j['value'], 'Gseg/s', j['roofline']['gsegments_per_s'], 'L', j['config']['mean_path_length'], 'ms/step', j['ms_per_step'])
"; }
run coat --scene cornell_coat.json --spp ${SPP:-64}
run diffuse --scene cornell_diffuse.json --spp ${SPP:-64}
run roughcond_env --scene cornell_roughcond.json --env sky --spp ${SPP:-64}
run roughdiel_env --scene cornell_roughdiel.json --env sky --spp ${SPP:-64}
run media_iso --scene cornell_media.json --env sky --spp ${SPP:-64}
run media_hg --scene cornell_media.json --env sky --phase hg --spp ${SPP:-64}
run dragon_4k --scene cornell_dragon.json --width 3840 --height 2160 --spp ${SPPD:-16}
run sdf_env --scene cornell_sdf.json --env sky --spp ${SPP:-64}
