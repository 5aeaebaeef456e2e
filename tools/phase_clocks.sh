#!/bin/bash
# development helper: where a wave's cycles go, per phase of render_kernel's iteration (a build with -DPT_PHASE_CLOCKS:
# tools/build_variant.sh clk -DPT_PHASE_CLOCKS).  Cycles are wall cycles of the wave, shared with the other waves of its SIMD.
P="photorealistic-rendering-using-opencl_amd"
PRT_LIB=$PWD/$P/variants/libprt_clk.so timeout -k 10 300 python3 bench.py --spp ${SPP:-256} --steps 1 --warmup 0 --no-cpu-baseline "$@" 2>&1 | grep -E "phase clocks|^\{" | cut -c1-200
