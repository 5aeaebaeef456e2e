#!/bin/bash
# development helper: what distinguishes one GPU box from the next (clocks, power cap, partition modes, firmware)
rocm-smi --showclocks --showpower --showmaxpower --showperflevel --showmemuse --showcomputepartition --showmemorypartition --showvbios --showfwinfo 2>&1 | grep -v "^$" | head -80
rocminfo 2>/dev/null | grep -E "Marketing Name|Compute Unit|Max Clock|Name: +gfx|amdgcn-amd-amdhsa|Cacheline|L2:|L3:|Wavefront" | head -24
env | grep -E "^HSA_|^HIP_|^ROCR|^GPU_" | head
cat /sys/class/drm/card*/device/pp_dpm_sclk 2>/dev/null | head -12
nproc; head -1 /proc/loadavg
