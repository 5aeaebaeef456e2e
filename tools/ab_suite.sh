#!/bin/bash
# A/B helper (development only): several builds of libprt over a small suite of workloads in one gpurun call
#   tools/ab_suite.sh base rec0 rec2 ...      (SUITE="dragon4k dragon1080 diffuse roughdiel coat512 media" to choose)
for w in ${SUITE:-dragon4k dragon1080 diffuse}; do
  case $w in
    dragon4k) args="--scene cornell_dragon.json --width 3840 --height 2160 --spp ${SPPD:-16}";;
    dragon1080) args="--scene cornell_dragon.json --spp ${SPPD:-16}";;
    diffuse) args="--scene cornell_diffuse.json --spp ${SPP:-64}";;
    roughdiel) args="--scene cornell_roughdiel.json --env sky --spp ${SPP:-64}";;
    roughcond) args="--scene cornell_roughcond.json --env sky --spp ${SPP:-64}";;
    coat512) args="--scene cornell_coat.json --width 512 --height 512 --spp 64";;
    media) args="--scene cornell_media.json --env sky --phase hg --spp ${SPP:-64}";;
  esac
  echo "== $w"
  tools/ab_scene.sh "$args" "$@"
done
