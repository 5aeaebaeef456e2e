#!/bin/bash
# A/B helper (development only): the same build under different environment settings over a small suite of workloads
#   SUITE="dragon4k diffuse" tools/ab_suite_env.sh "PRT_TRI_Q=0" "PRT_TRI_Q=4" "PRT_LIB=$PWD/<package>/variants/libprt_x.so" ...
for w in ${SUITE:-dragon4k dragon1080 diffuse}; do
  case $w in
    dragon4k) args="--scene cornell_dragon.json --width 3840 --height 2160 --spp ${SPPD:-16}";;
    dragon1080) args="--scene cornell_dragon.json --spp ${SPPD:-16}";;
    diffuse) args="--scene cornell_diffuse.json --spp ${SPP:-64}";;
    roughdiel) args="--scene cornell_roughdiel.json --env sky --spp ${SPP:-64}";;
    roughcond) args="--scene cornell_roughcond.json --env sky --spp ${SPP:-64}";;
    coat512) args="--scene cornell_coat.json --width 512 --height 512 --spp 64";;
    coat) args="--scene cornell_coat.json --spp ${SPP:-64}";;
    media) args="--scene cornell_media.json --env sky --phase hg --spp ${SPP:-64}";;
    sdf) args="--scene cornell_sdf.json --env sky --spp ${SPP:-64}";;
  esac
  echo "== $w"
  for v in "$@"; do
    env $v timeout -k 10 300 python3 bench.py --steps ${STEPS:-1} --warmup 1 --no-cpu-baseline $args 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('$v', 'Msamples/s', j['value'], 'Gseg/s', j['roofline']['gsegments_per_s'], 'avg_launch_ms', j['roofline']['avg_launch_ms'])
"
  done
done
