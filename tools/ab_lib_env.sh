#!/bin/bash
# A/B helper (development only): builds x environments:  tools/ab_lib_env.sh "<bench args>" "lib:ENV=val ..." ...
P="photorealistic-rendering-using-opencl_amd"
args=$1; shift
for v in "$@"; do
  lib=${v%%:*}; envs=${v#*:}
  if [ "$lib" = "base" ]; then path="$P/libprt.so"; else path="$P/variants/libprt_$lib.so"; fi
  env PRT_LIB=$PWD/$path $envs timeout -k 10 300 python3 bench.py --steps ${STEPS:-1} --warmup 1 --no-cpu-baseline $args 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('$v', 'Msamples/s', j['value'], 'Gseg/s', j['roofline']['gsegments_per_s'], 'avg_launch_ms', j['roofline']['avg_launch_ms'])
"
done
