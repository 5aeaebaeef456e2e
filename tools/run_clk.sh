#!/bin/bash
# development helper: per-phase wave cycles (clk build) and wave-level execution counts (wst build) of a few workloads
P="photorealistic-rendering-using-opencl_amd"
for v in ${VARIANTS:-clk wst}; do
for a in "--scene cornell_dragon.json --width 3840 --height 2160 --spp 8" "--scene cornell_diffuse.json --spp 32"; do
echo "== $v $a"
PRT_LIB=$PWD/$P/variants/libprt_$v.so timeout -k 10 600 python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline $a 2>&1 | grep -E "phase clocks" | cut -c1-200
done; done
