#!/bin/bash
# development helper: tools/isa.sh <out.s> [-D...]  -> listing (with -g line info) of pt_kernels.hip built for the one-variant
# development configuration, and the resource usage of render_kernel (for tools/asm_loops.py / asm_attrib.py)
out=$1; shift
H="$(cd "$(dirname "$0")/.." && pwd)/photorealistic-rendering-using-opencl_amd/csrc/hip"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -fno-slp-vectorize -DPT_UNITY -g -S --cuda-device-only -DPT_DEV_ONE_VARIANT "$@" \
  -I$H/../../../include -I$H -Rpass-analysis=kernel-resource-usage -o $out $H/pt_kernels.hip 2>&1 \
  | grep -E "error|Function Name: .*render_kernel" -A9 | grep -E "error|SGPRs:|VGPRs:|ScratchSize|Occupancy"
