#!/bin/bash
# development helper: shader clock and package power WHILE the headline bench runs (is a slow box a throttled one?)
python3 bench.py --steps 8 --warmup 1 --no-cpu-baseline > gpurun_out/clk_bench.json 2>/dev/null &
BP=$!
sleep 6
for i in 1 2 3 4 5 6; do
  rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|fclk|socclk|Package Power" | sed 's/GPU\[0\]\s*: //' | tr '\n' ';'; echo
  sleep 0.7
done
wait $BP
python3 -c "
import json
j=json.loads(open('gpurun_out/clk_bench.json').read().strip().split('\n')[-1]); print('Msamples/s', j['value'], 'avg_launch_ms', j['roofline']['avg_launch_ms'])"
