#!/bin/bash
# A/B helper (development only): runs the bench on several builds of libprt in one gpurun call
P="photorealistic-rendering-using-opencl_amd"
for v in "$@"; do
  if [ "$v" = "base" ]; then lib="$P/libprt.so"; else lib="$P/variants/libprt_$v.so"; fi
  PRT_LIB=$PWD/$lib timeout -k 10 200 python3 bench.py --spp ${SPP:-32} --steps ${STEPS:-2} --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('$v', 'Msamples/s', j['value'], 'Gseg/s', j['roofline']['gsegments_per_s'], 'avg_launch_ms', j['roofline']['avg_launch_ms'], 'build', j.get('build_id'))
"
done
