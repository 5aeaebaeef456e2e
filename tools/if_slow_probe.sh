#!/bin/bash
# development helper: if this box is one of the slow kind (headline below 1 850 Msamples/s), collect what tells the kinds apart
v=$(python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'): print(int(json.loads(l)['value']))")
echo "headline $v Msamples/s"
if [ "${v:-0}" -lt 1850 ]; then
  echo "slow box: probing"
  bash tools/box_info.sh > gpurun_out/slow_box_info.log 2>&1
  bash tools/pmc_probe.sh > /dev/null 2>&1
  mkdir -p gpurun_out/pmc_probe_slow && cp gpurun_out/pmc_probe/*.json gpurun_out/pmc_probe_slow/
  bash tools/clk_under_load.sh > gpurun_out/slow_clk.log 2>&1
  bash tools/ab_box.sh > gpurun_out/slow_ab_box.log 2>&1
fi
