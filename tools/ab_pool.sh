#!/bin/bash
# development helper: the pool kernel (PRT_POOL=1) of several variant builds on one workload, with their pool statistics when built with -DPT_POOL_STATS
#   tools/ab_pool.sh "<bench args>" name[:env=val] ...      ("base" = the in-tree library)
P="photorealistic-rendering-using-opencl_amd"
args=$1; shift
for spec in "$@"; do
  v=${spec%%:*}; envs=""; [ "$spec" != "$v" ] && envs=${spec#*:}
  if [ "$v" = "base" ]; then lib="$P/libprt.so"; else lib="$P/variants/libprt_$v.so"; fi
  env PRT_LIB=$PWD/$lib $envs timeout -k 10 300 python3 bench.py --steps ${STEPS:-1} --warmup 1 --no-cpu-baseline $args 2>/tmp/ab_pool.err | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('$spec', 'Msamples/s', j['value'], 'Gseg/s', j['roofline']['gsegments_per_s'], j['roofline']['kernel'], 'build', j.get('build_id'))
"
  grep "pool stats" /tmp/ab_pool.err | tail -3 | cut -c1-260
done
