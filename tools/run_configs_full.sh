#!/bin/bash
# the BASELINE.json configs at their stated resolution and spp on ONE GPU (config 5: 1/16 of its spp); one cold render each, except
# config 1 (a 23 ms render: 5 steps after a warm-up, or the number is the first launch's start-up)
run() { name=$1; shift; timeout -k 10 900 python3 bench.py --steps ${STEPS:-1} --warmup ${WARM:-0} --no-cpu-baseline "$@" 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('$name', 'Msamples/s', j['value'], 'Gseg/s', j['roofline']['gsegments_per_s'], 'L', j['config']['mean_path_length'], 's/step', round(j['ms_per_step']/1000,2), 'frac', j['roofline']['frac'])
"; }
STEPS=5 WARM=1 run c1_coat_512x512_64spp --scene cornell_coat.json --width 512 --height 512 --spp 64
run c2_diffuse_1080p_1024spp --scene cornell_diffuse.json --spp 1024
run c3a_roughcond_env_1080p_4096spp --scene cornell_roughcond.json --env sky --spp 4096
run c3b_roughdiel_env_1080p_4096spp --scene cornell_roughdiel.json --env sky --spp 4096
run c4_media_iso_1080p_4096spp --scene cornell_media.json --env sky --spp 4096
run c4_media_hg_1080p_4096spp --scene cornell_media.json --env sky --phase hg --spp 4096
run c5_dragon_4k_512spp --scene cornell_dragon.json --width 3840 --height 2160 --spp 512
